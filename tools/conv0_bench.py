"""conv0 + GroupNorm + GELU forward / backward at the bench shape (32 x 240 000 samples, 512 channels, bf16), timed with
HIP events: python tools/conv0_bench.py [iters].  WAVLM_HIP_LIB selects another build of the library (A/B of variants)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unispeech_amd import functional as F  # noqa: E402

it = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, T, C = 32, 240000, 512
dev = "cuda"
g = torch.Generator().manual_seed(0)
wav = torch.randn(B, T, generator=g).to(dev).to(torch.bfloat16)
W = (torch.randn(C, 1, 10, generator=g) * 0.3).to(dev).to(torch.bfloat16).requires_grad_(True)
gm = torch.ones(C, device=dev, dtype=torch.bfloat16, requires_grad=True)
bt = torch.zeros(C, device=dev, dtype=torch.bfloat16, requires_grad=True)
y = F.Conv0Fn.apply(wav, W, gm, bt, 5, 1e-5, torch.bfloat16)
dy = torch.randn_like(y)
y.backward(dy)
for _ in range(2):  # (first calls load the code objects and set the LDS attributes: tens of milliseconds)
    y = F.Conv0Fn.apply(wav, W, gm, bt, 5, 1e-5, torch.bfloat16)
    y.backward(dy)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for _ in range(it):
    e[0].record()
    y = F.Conv0Fn.apply(wav, W, gm, bt, 5, 1e-5, torch.bfloat16)
    e[1].record()
    y.backward(dy)
    e[2].record()
    torch.cuda.synchronize()
    tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
nbytes = y.numel() * 2
print("lib %s: conv0 fwd %.3f ms (%.2f TB/s written), bwd %.3f ms (%.2f TB/s read)"
      % (os.environ.get("WAVLM_HIP_LIB", "default"), tf / it, nbytes / (tf / it * 1e-3) / 1e12, tb / it, nbytes / (tb / it * 1e-3) / 1e12))
