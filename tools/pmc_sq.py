"""Per-kernel SQ counters of a rocprofv3 --pmc pass (counter_collection.csv): sums per kernel name and per step.
usage: pmc_sq.py <counter_collection.csv> <steps_profiled>"""
import collections
import csv
import sys

per = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
names = []
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:56]
    c = r["Counter_Name"]
    if c not in names:
        names.append(c)
    per[k][c] += float(r["Counter_Value"])
    if c == names[0]:
        cnt[k] += 1
steps = float(sys.argv[2])
print("%-56s %7s " % ("kernel", "calls") + " ".join("%22s" % n[:22] for n in names))
key = names[0]
for k, v in sorted(per.items(), key=lambda kv: -kv[1].get(key, 0))[:30]:
    print("%-56s %7.1f " % (k, cnt[k] / steps) + " ".join("%22.4g" % (v.get(n, 0.0) / steps) for n in names))
