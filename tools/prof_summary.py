"""Condense a rocprofv3 *_kernel_stats.csv into ms per step by kernel (usage: prof_summary.py stats.csv nsteps)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = 0.0
out = []
for r in rows:
    name = r["Name"].split("(")[0].replace("void ", "")
    if len(name) > 70:
        name = name[:70]
    ms = float(r["TotalDurationNs"]) / 1e6 / nsteps
    tot += ms
    out.append((ms, int(r["Calls"]) / nsteps, float(r["AverageNs"]) / 1e3, name))
out.sort(reverse=True)
print("%-72s %9s %8s %10s" % ("kernel", "ms/step", "calls", "avg us"))
for ms, calls, avg, name in out[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print("%-72s %9.3f %8.1f %10.1f" % (name, ms, calls, avg))
print("total kernel time per step: %.2f ms" % tot)
