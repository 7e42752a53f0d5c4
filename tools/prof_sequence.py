"""The LAST optimizer step of a rocprofv3 kernel trace as an ordered list (index, start offset us, duration us, full kernel name):
where in the step the small launches sit.  usage: prof_sequence.py <run_kernel_trace.csv> [name filter regex]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "")) for r in rows)
ad = [i for i, e in enumerate(ev) if e[2].startswith("adam_step")]
seg = ev[ad[-2] + 1:ad[-1] + 1]
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
t0 = seg[0][0]
for i, (s, e, n) in enumerate(seg):
    short = n if len(n) < 150 else n[:150]
    if pat is None or pat.search(n) or (i > 0 and pat.search(seg[i - 1][2])) or (i + 1 < len(seg) and pat.search(seg[i + 1][2])):
        print("%4d %9.1f %7.1f  %s" % (i, (s - t0) / 1e3, (e - s) / 1e3, short))
