"""Frame timeline from a rocprofv3 --kernel-trace CSV: busy time per kernel, idle gaps and the kernel each gap precedes."""
import collections
import csv
import glob
import re
import sys


def short(name):
    return re.sub(r"\(.*", "", name.replace("void ", "")).strip()


rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2] == "k_motion"]
# frames of the 20-step timed window: skip the warm-up frames, keep frames that contain no census launch
frames = []
for a, b in zip(starts, starts[1:]):
    fr = rows[a:b]
    if any("true>" in r[2] and "k_score_kd" in r[2] for r in fr):
        continue   # a frame of bench.py's census replay (the counting instantiation runs behind its scan-match kernel)
    if not any("k_score_kd" in r[2] for r in fr):
        continue   # a bare pfslam_motion_update (bench.py disperses the cloud five times before the first frame)
    frames.append(fr + [rows[b]])
walls = sorted(fr[-1][0] - fr[0][0] for fr in frames)
median = walls[len(walls) // 2] if walls else 0
frames = [fr for fr in frames if fr[-1][0] - fr[0][0] < 1.5 * median][-20:]  # steady state: no allocation / census / balance frame
busy = collections.defaultdict(float)
gap_before = collections.defaultdict(float)
count = collections.defaultdict(int)
wall = idle = 0.0
for fr in frames:
    t0, end = fr[0][0], fr[0][0]
    for s, e, k in fr[:-1]:
        if s > end:
            gap_before[k] += s - end
            idle += s - end
        busy[k] += e - s
        count[k] += 1
        end = max(end, e)
    nxt = fr[-1][0]
    if nxt > end:
        gap_before["(next frame's k_motion)"] += nxt - end
        idle += nxt - end
    wall += nxt - t0
n = max(len(frames), 1)
print("frames %d: wall %.1f us/frame, idle %.1f us/frame" % (len(frames), wall / n / 1e3, idle / n / 1e3))
print("%-44s %8s %10s %12s" % ("kernel", "calls/fr", "busy us/fr", "gap before us/fr"))
keys = sorted(set(busy) | set(gap_before), key=lambda k: -(busy[k] + gap_before[k]))
for k in keys:
    print("%-44s %8.2f %10.2f %12.2f" % (k[:44], count[k] / n, busy[k] / n / 1e3, gap_before[k] / n / 1e3))
