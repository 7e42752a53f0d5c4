"""Per-kernel table of the LAST optimizer step in a rocprofv3 kernel trace (steps are delimited by adam_step_kernel; a trace
without an optimizer -- extract_features calls -- by the extractor's first kernel: the last COMPLETE call),
plus the GPU idle time inside that step.  usage: prof_laststep.py <run_kernel_trace.csv> [rows]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0][:64]) for r in rows)
ad = [i for i, e in enumerate(ev) if e[2].startswith("adam_step")]
if len(ad) >= 2:
    seg = ev[ad[-2] + 1:ad[-1] + 1]
else:
    # no optimizer in the trace (bench.py --config extract: forward calls only): a call starts with the extractor's first kernel
    first = [i for i, e in enumerate(ev) if e[2].startswith(("conv0_gram", "conv0_ln_gram"))]
    if len(first) < 2:
        sys.exit("prof_laststep.py: no step boundary in this trace (neither adam_step_kernel nor two conv0 Gram kernels)")
    seg = ev[first[-2]:first[-1]]
c = collections.defaultdict(lambda: [0, 0])
for s, e, n in seg:
    c[n][0] += 1
    c[n][1] += e - s
busy = sum(v[1] for v in c.values())
span = seg[-1][1] - seg[0][0]
print("last step: %d kernels, span %.2f ms, busy %.2f ms, idle %.2f ms" % (len(seg), span / 1e6, busy / 1e6, (span - busy) / 1e6))
print("%-66s %5s %9s %9s" % ("kernel", "calls", "ms/step", "avg us"))
for n, (k, t) in sorted(c.items(), key=lambda x: -x[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("%-66s %5d %9.3f %9.1f" % (n, k, t / 1e6, t / k / 1e3))
