"""LayerNorm (+ residual + dropout) forward / backward at the encoder's shape (23 968 x 768 bf16), HIP-event timing.
WAVLM_HIP_LIB selects another build (A/B of launch geometries)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unispeech_amd import ops  # noqa: E402

n, D = int(os.environ.get('LN_ROWS', 23968)), int(os.environ.get('LN_D', 768))
dev = "cuda"
x = torch.randn(n, D, device=dev).to(torch.bfloat16)
r = torch.randn(n, D, device=dev).to(torch.bfloat16)
g = torch.ones(D, device=dev, dtype=torch.bfloat16)
b = torch.zeros(D, device=dev, dtype=torch.bfloat16)
dy = torch.randn(n, D, device=dev).to(torch.bfloat16)
dg = torch.zeros(D, device=dev, dtype=torch.bfloat16)
db = torch.zeros(D, device=dev, dtype=torch.bfloat16)
dc = torch.zeros(D, device=dev, dtype=torch.bfloat16)
for p in (0.0, 0.1):
    y, s, mean, rstd = ops.layernorm_fwd(x, r, g, b, 1e-5, p_in=p, seed_in=7)
    ops.layernorm_bwd(dy, s, mean, rstd, g, b, p_in=p, seed_in=7, need_dr=p > 0, dgamma=dg, dbeta=db, dr_colsum=dc)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    it = 50
    for _ in range(it):
        e[0].record()
        y, s, mean, rstd = ops.layernorm_fwd(x, r, g, b, 1e-5, p_in=p, seed_in=7)
        e[1].record()
        ops.layernorm_bwd(dy, s, mean, rstd, g, b, p_in=p, seed_in=7, need_dr=p > 0, dgamma=dg, dbeta=db, dr_colsum=dc)
        e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    mb = n * D * 2 / 1e6
    print("full=%s blocks=%s D=%d lib %s p=%.1f: fwd %.1f us (%.2f TB/s on 4 passes), bwd+finish %.1f us"
          % (os.environ.get("WAVLM_LN_FULL", "1"), os.environ.get("WAVLM_LN_FWD_BLOCKS", "-"), D, os.path.basename(os.environ.get("WAVLM_HIP_LIB", "default")), p, tf / it * 1e3, 4 * mb / (tf / it * 1e3), tb / it * 1e3))
