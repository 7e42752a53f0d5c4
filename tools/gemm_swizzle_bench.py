"""Tile order inside an XCD (gemm_common.hpp: gemm_tile_rc): times the wide-N activation launches of the Base and Large encoder
blocks; run once with WAVLM_GEMM_SWIZZLE=0 (row-major tile ids) and once without.  Under `rocprofv3 --pmc FETCH_SIZE` the
per-launch fetch shows what the order does to the operand traffic."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unispeech_amd import ops  # noqa: E402

dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


tag = "row-major" if os.environ.get("WAVLM_GEMM_SWIZZLE") == "0" else "XCD blocks"
for (n, N, K, what) in ((23968, 3072, 768, "Base fc1 / fc2-dX"), (23968, 2304, 768, "Base q|k|v"), (31968, 4096, 1024, "Large fc1 / fc2-dX"),
                        (31968, 3072, 1024, "Large q|k|v"), (31968, 1024, 4096, "Large fc2 / fc1-dX")):
    x = (0.5 * torch.randn(n, K, device=dev)).to(bf)
    W = (K ** -0.5 * torch.randn(N, K, device=dev)).to(bf)
    Wt = W.t().contiguous()
    b = (0.1 * torch.randn(N, device=dev)).to(bf)
    y = torch.empty(n, N, device=dev, dtype=bf)
    g = torch.empty(n, N, device=dev, dtype=bf)
    cs = torch.zeros(N, device=dev, dtype=bf)
    t1 = timeit(lambda: ops.gemm(x, W, y, n, N, K, lda=K, ldb=K, ldc=N, bias=b))
    t2 = timeit(lambda: ops.gemm(x, W, y, n, N, K, lda=K, ldb=K, ldc=N, bias=b, epi=3, aux=g, ld_aux=N))
    t3 = timeit(lambda: ops.gemm(x, Wt, y, n, N, K, lda=K, ldb=N, ldc=N, transB=True, epi=4, aux=g, ld_aux=N, colsum=cs, colsum_accumulate=True))
    fl = 2.0 * n * N * K
    print("%-10s %-20s %5d x %4d x %4d: plain %6.1f us (%4.0f TF/s) | gelu + g' %6.1f us | NT x aux + csum %6.1f us"
          % (tag, what, n, N, K, t1, fl / t1 / 1e6, t2, t3), flush=True)
