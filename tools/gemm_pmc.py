"""Launches each GEMM layout of the WavLM-Base step a few times (for `rocprofv3 --pmc ...` counter passes).
usage: python tools/gemm_pmc.py [variant]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unispeech_amd import ops  # noqa: E402

dev, bf = "cuda", torch.bfloat16
ops.gemm_set_variant(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = 32 * 749
rnd = lambda *s: torch.randn(*s, device=dev, dtype=bf)
x, W, y, b = rnd(n, 768), rnd(3072, 768), torch.empty(n, 3072, device=dev, dtype=bf), rnd(3072)
x2, W2, y2 = rnd(n, 3072), rnd(768, 3072), torch.empty(n, 768, device=dev, dtype=bf)
dW = torch.empty(3072, 768, device=dev, dtype=bf)
for _ in range(3):
    ops.gemm(x, W, y, n, 3072, 768, lda=768, ldb=768, ldc=3072, bias=b)                       # NN  K=768
    ops.gemm(x2, W2, y2, n, 768, 3072, lda=3072, ldb=3072, ldc=768)                           # NN  K=3072
    ops.gemm(y, W, y2, n, 768, 3072, lda=3072, ldb=768, ldc=768, transB=True)                 # NT  dX fc1
    ops.gemm(y, x, dW, 3072, 768, n, lda=3072, ldb=768, ldc=768, transA=True, transB=True, split_k=6)  # TT dW
torch.cuda.synchronize()
