"""overlaps.py TRACE_DIR: for every timed scan-match launch of a rocprofv3 kernel trace, its duration and which other kernels ran during it
(name: overlap in us, start relative to the launch's start)."""
import csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "")), r.get("Queue_Id", "?")))
rows.sort()
sc = [r for r in rows if r[2].startswith("k_score_kd_cells<false")][5:25]
for s, e, name, q in sc:
    ov = []
    for s2, e2, n2, q2 in rows:
        if n2 == name and s2 == s: continue
        o = min(e, e2) - max(s, s2)
        if o > 0: ov.append("%s[q%s] %.0f@%+.0f" % (n2[:22], q2, o / 1e3, (s2 - s) / 1e3))
    print("%.0f us [q%s]: %s" % ((e - s) / 1e3, q, "; ".join(ov)))
