cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/w4_ab.py all > gpurun_out/w4_ab.txt 2>&1; grep -v "ok$" gpurun_out/w4_ab.txt | tail -40
WAVLM_GEMM_W4=1 timeout 300 python tools/w4_ab.py none > /dev/null 2>&1
for m in 0 1; do WAVLM_GEMM_W4=$m timeout 300 python - <<'PY' 2>&1 | tail -3
import os, sys
sys.argv = ["x", "none"]
sys.path.insert(0, "tools")
import torch
exec(open("tools/w4_ab.py").read().split("if what in (\"check\"")[0])
n = 32 * 749
items = [(rnd(n, N), rnd(n, K), torch.zeros(N, K, device=dev, dtype=bf)) for (N, K) in [(768, 3072), (3072, 768), (768, 768), (2304, 768)]]
fl = sum(2.0 * n * a.shape[1] * b.shape[1] for a, b, _ in items)
ms, tf = timeit(lambda: ops.gemm_wgrad_grouped(items, bf), fl, "grouped")
print("grouped dW of a block, WAVLM_GEMM_W4=%s: %.1f us = %.1f TF/s" % (os.environ.get("WAVLM_GEMM_W4"), ms * 1e3, tf))
ref = [(a.double().t() @ b.double()) for a, b, _ in items]
for (a, b, o), r in zip(items, ref):
    o.zero_()
ops.gemm_wgrad_grouped(items, bf)
torch.cuda.synchronize()
print("max rel err", max(((o.double() - r).abs().max() / r.abs().max()).item() for (_, _, o), r in zip(items, ref)))
PY
done
