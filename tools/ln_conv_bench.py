"""The LayerNorm of layer_norm-mode extractors (WavLM-Large: conv output -> LayerNorm(512) -> GELU, no residual) at Large's
largest shape (32 x 31999 rows x 512), HIP-event timing.  WAVLM_LN_FWD_BLOCKS / WAVLM_LN_BWD_BLOCKS select the launch geometry
(one process per setting: the library reads them once)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unispeech_amd import ops  # noqa: E402

n, D = int(os.environ.get("LN_ROWS", 32 * 31999)), 512
dev = "cuda"
x = torch.randn(n, D, device=dev).to(torch.bfloat16)
g = torch.ones(D, device=dev, dtype=torch.bfloat16)
b = torch.zeros(D, device=dev, dtype=torch.bfloat16)
dy = torch.randn(n, D, device=dev).to(torch.bfloat16)
dg = torch.zeros(D, device=dev, dtype=torch.bfloat16)
db = torch.zeros(D, device=dev, dtype=torch.bfloat16)
y, s, mean, rstd = ops.layernorm_fwd(x, None, g, b, 1e-5, act=1)
ops.layernorm_bwd(dy, s, mean, rstd, g, b, act=1, dgamma=dg, dbeta=db)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
it = 20
for _ in range(it):
    e[0].record()
    y, s, mean, rstd = ops.layernorm_fwd(x, None, g, b, 1e-5, act=1)
    e[1].record()
    ops.layernorm_bwd(dy, s, mean, rstd, g, b, act=1, dgamma=dg, dbeta=db)
    e[2].record()
    torch.cuda.synchronize()
    tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
gb = n * D * 2 / 1e9
print("fwd blocks %s bwd blocks %s: fwd %.0f us (%.2f TB/s on 2 passes), bwd + finish %.0f us (%.2f TB/s on 3 passes)"
      % (os.environ.get("WAVLM_LN_FWD_BLOCKS", "default"), os.environ.get("WAVLM_LN_BWD_BLOCKS", "default"),
         tf / it * 1e3, 2 * gb / (tf / it * 1e-3) / 1e3, tb / it * 1e3, 3 * gb / (tb / it * 1e-3) / 1e3))
