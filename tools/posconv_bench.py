"""Times pos_conv (weight-norm grouped Conv1d k=128, 16 groups + GELU + residual; WavLM/WavLM.py:514-527, 577-579) forward and
backward at the WavLM-Base step shape (B=32, T=749, D=768) or, with `large`, the Large one (B=32, T=999, D=1024).  For
per-kernel numbers run under `rocprofv3 --kernel-trace --stats`."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unispeech_amd import functional as F  # noqa: E402

dev = "cuda"
large = len(sys.argv) > 1 and sys.argv[1] == "large"
B, T, D, K, G = (32, 999, 1024, 128, 16) if large else (32, 749, 768, 128, 16)
Cg = D // G
bf = torch.bfloat16
x = torch.randn(B, T, D, device=dev).to(bf).requires_grad_(True)
v = (torch.randn(D, Cg, K, device=dev) * math.sqrt(4.0 / (K * D))).to(bf).requires_grad_(True)
g = v.detach().float().norm(dim=(0, 1), keepdim=True).to(bf).requires_grad_(True)
bias = torch.zeros(D, device=dev, dtype=bf, requires_grad=True)
dy = torch.randn(B, T, D, device=dev).to(bf)


def step():
    y = F.PosConvFn.apply(x, v, g, bias, G)
    y.backward(dy)


for _ in range(3):
    step()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    step()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
fl = 3 * 2.0 * B * T * D * Cg * K
print("pos_conv fwd+bwd [%dx%dx%d]  %.3f ms  %.1f TF/s" % (B, T, D, ms, fl / ms / 1e9))
