"""HBM traffic of the GEMM kernels from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, each with --kernel-trace)
over `bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline`.
usage: pmc_traffic.py <dir_with_FETCH_SIZE_pass> <dir_with_WRITE_SIZE_pass> <steps_profiled> [out.json]
Corrections (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): both counters are in KiB-like units of 1024 B as
reported by rocprofv3; FETCH_SIZE on gfx950 counts 128-B requests at 64 B, i.e. HALF of a wide coalesced read -> x2.
WRITE_SIZE is uncalibrated on gfx950 (taken as reported)."""
import collections
import csv
import json
import sys


def load(d, counter):
    per = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(d + "/run_counter_collection.csv")):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        per[name][0] += 1
        per[name][1] += float(r["Counter_Value"])
    return per


fd, wd, steps = sys.argv[1], sys.argv[2], float(sys.argv[3])
F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
rows = []
for k in sorted(set(F) | set(W)):
    n = F.get(k, [0, 0])[0] or W.get(k, [0, 0])[0]
    fetch = 2.0 * F.get(k, [0, 0.0])[1] * 1024.0   # gfx950: x2
    write = W.get(k, [0, 0.0])[1] * 1024.0
    rows.append((fetch + write, k, n, fetch, write))
rows.sort(reverse=True)
print("%-60s %8s %12s %12s %12s" % ("kernel", "launches", "fetch MB/st", "write MB/st", "MB/launch"))
for tot, k, n, f, w in rows[:25]:
    print("%-60s %8.1f %12.1f %12.1f %12.2f" % (k[:60], n / steps, f / steps / 1e6, w / steps / 1e6, tot / max(n, 1) / 1e6))
# EVERY kernel of the GEMM family (gemm_pp / gemm_pp3 / gemm_w4 / gemm_h2 / gemm_bf16 + the split-K reductions that serve
# them): selected by the common prefix, not by a list of names -- round 4's list missed gemm_w4_kernel, the step's largest
# GEMM kernel.  The launch count (split-K reductions excluded: they are part of their GEMM's launch
# in the library's own record) must equal bench.py's roofline.launches_per_step; bench.py checks that.
g = [(k, n, f, w) for _, k, n, f, w in rows if k.startswith("gemm_")]
n = sum(x[1] for x in g if not x[0].startswith("gemm_splitk")); f = sum(x[2] for x in g); w = sum(x[3] for x in g)
out = {"gemm_launches_per_step": n / steps, "gemm_fetch_bytes_per_step": f / steps, "gemm_write_bytes_per_step": w / steps,
       "gemm_hbm_bytes_per_launch": (f + w) / max(n, 1),
       "kernels": {k: {"launches_per_step": n_ / steps, "hbm_bytes_per_launch": (f_ + w_) / max(n_, 1)} for k, n_, f_, w_ in g},
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B); "
               "every kernel whose name starts with gemm_; split-K reduce kernels' traffic is included in the bytes, not in the launch count"}
print(json.dumps(out))
if len(sys.argv) > 4:
    json.dump(out, open(sys.argv[4], "w"), indent=1)
