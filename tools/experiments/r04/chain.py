"""chain.py TRACE_DIR: the frame's tail from a rocprofv3 kernel trace -- for the steady-state frames, mean start / end of every kernel
relative to the END of the frame's scan-match kernel (us), its queue, and the start of the next scan-match kernel."""
import csv, glob, re, sys
from collections import defaultdict
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "")), r.get("Queue_Id", "?")))
rows.sort()
sc = [r for r in rows if r[2].startswith("k_score_kd_cells<false")]
sc = sc[8:24]
acc = defaultdict(lambda: [0.0, 0.0, 0, set()])
nxt = 0.0
for k in range(len(sc) - 1):
    e0, s1 = sc[k][1], sc[k + 1][0]
    nxt += (s1 - e0) / 1e3
    for s, e, n, q in rows:
        if s >= e0 - 400000 and s < s1 and not n.startswith("k_score_kd_cells"):
            a = acc[n]
            a[0] += (s - e0) / 1e3; a[1] += (e - e0) / 1e3; a[2] += 1; a[3].add(q)
m = len(sc) - 1
print("next scan-match kernel starts %.1f us after the end of this one (frame = kernel + that)" % (nxt / m))
for n, a in sorted(acc.items(), key=lambda kv: kv[1][0] / kv[1][2]):
    print("%-34s q%-6s x%.2f  start %+8.1f  end %+8.1f  (%.1f us)" % (n[:34], ",".join(sorted(map(str, a[3]))), a[2] / m, a[0] / a[2], a[1] / a[2], (a[1] - a[0]) / a[2]))
