"""Epilogue probe: one round of 256x256 tiles at K = 64 with 8 / 64 / 256 CUs active -- is a tile's epilogue bound by the
CU itself or by the chip-wide HBM write burst?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unispeech_amd import ops  # noqa: E402

dev, bf = "cuda", torch.bfloat16
ops.gemm_set_variant(3)
for (M, N, K) in [(256, 2048, 64), (2048, 2048, 64), (8192, 2048, 64), (256, 2048, 768), (8192, 2048, 768)]:
    x = torch.randn(M, K, device=dev, dtype=bf)
    W = torch.randn(N, K, device=dev, dtype=bf)
    y = torch.empty(M, N, device=dev, dtype=bf)
    u = torch.empty(M, N, device=dev, dtype=bf)
    b = torch.randn(N, device=dev, dtype=bf)
    for name, f in (("plain", lambda: ops.gemm(x, W, y, M, N, K, lda=K, ldb=K, ldc=N)),
                    ("bias+gelu+aux", lambda: ops.gemm(x, W, y, M, N, K, lda=K, ldb=K, ldc=N, bias=b, epi=1, aux=u, ld_aux=N))):
        for _ in range(5):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        print("tiles=%4d K=%4d %-14s %7.1f us per launch" % ((M // 256) * (N // 256), K, name, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
