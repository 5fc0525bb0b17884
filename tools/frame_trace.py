"""tools/frame_trace.py <rocprofv3 kernel-trace dir> [first last]  -- a frame's launches, when they start and end relative to the END of
the frame's scan-match kernel (k_score_kd_cells<false, ...>), mean over the frames of bench.py's timed window (frames with a census
launch are dropped).  Also: launches per frame, and the time from the end of one scan-match kernel to the start of the next."""
import collections, csv, glob, re, sys

def short(name):
    return re.sub(r"\(.*", "", name.replace("void ", "")).strip()

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
sm = [i for i, r in enumerate(rows) if r[2].startswith("k_score_kd_cells<false")]
frames = []
for a, b in zip(sm, sm[1:]):
    fr = rows[a:b + 1]
    if any(r[2].startswith("k_score_kd_cells<true") for r in fr):
        continue
    frames.append(fr)
gaps = sorted(fr[-1][0] - fr[0][1] for fr in frames)
med = gaps[len(gaps) // 2]
frames = [fr for fr in frames if fr[-1][0] - fr[0][1] < 2.0 * med]
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (5, 25)
frames = frames[lo:hi]
acc = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0])
for fr in frames:
    end0 = fr[0][1]
    for s, e, k in fr[1:-1]:
        a = acc[k]
        a[0] += s - end0; a[1] += e - end0; a[2] += e - s; a[3] += 1
n = len(frames)
print("frames %d; scan-match kernel %.1f us; end of scan-match -> start of next %.1f us (min %.1f max %.1f); launches per frame %.1f" % (
    n, sum(fr[0][1] - fr[0][0] for fr in frames) / n / 1e3, sum(fr[-1][0] - fr[0][1] for fr in frames) / n / 1e3,
    min(fr[-1][0] - fr[0][1] for fr in frames) / 1e3, max(fr[-1][0] - fr[0][1] for fr in frames) / 1e3, sum(len(fr) - 1 for fr in frames) / n))
print("%-40s %6s %9s %9s %8s" % ("kernel (relative to the END of the scan-match kernel)", "per fr", "start us", "end us", "dur us"))
for k, a in sorted(acc.items(), key=lambda kv: kv[1][0] / kv[1][3]):
    c = a[3]
    print("%-40s %6.2f %+9.1f %+9.1f %8.1f" % (k[:40], c / n, a[0] / c / 1e3, a[1] / c / 1e3, a[2] / c / 1e3))
