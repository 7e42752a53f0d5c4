"""K sweep at fixed M, N: separates the per-K-tile cost from the per-output-tile (prologue + epilogue) cost.
usage: python tools/gemm_ksweep.py <variant> [N]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unispeech_amd import ops  # noqa: E402

dev, bf = "cuda", torch.bfloat16
ops.gemm_set_variant(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2304
M = 256 * 94
for K in (64, 256, 768, 1536, 3072, 6144):
    x = torch.randn(M, K, device=dev, dtype=bf)
    W = torch.randn(N, K, device=dev, dtype=bf)
    y = torch.empty(M, N, device=dev, dtype=bf)
    f = lambda: ops.gemm(x, W, y, M, N, K, lda=K, ldb=K, ldc=N)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("M=%d N=%d K=%5d  %8.3f ms  %7.1f TF/s" % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
