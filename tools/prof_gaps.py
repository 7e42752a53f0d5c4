"""Largest idle gaps between consecutive kernels of the last optimizer step in a rocprofv3 kernel trace:
python tools/prof_gaps.py <kernel_trace.csv> [n].  Prints gap length and the kernels on either side."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 15
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
ends = [i for i, e in enumerate(ev) if "adam_step" in e[2]]
if len(ends) >= 2:
    ev = ev[ends[-2] + 1:ends[-1] + 1]
gaps = []
last_end = ev[0][1]
for i in range(1, len(ev)):
    g = ev[i][0] - last_end
    if g > 0:
        gaps.append((g, ev[i - 1][2][:60], ev[i][2][:60]))
    last_end = max(last_end, ev[i][1])
gaps.sort(reverse=True)
tot = sum(g for g, _, _ in gaps)
print("kernels %d, total idle %.3f ms in %d gaps" % (len(ev), tot / 1e6, len(gaps)))
for g, a, b in gaps[:n]:
    print("%8.1f us  after %-60s before %s" % (g / 1e3, a, b))
