"""usage (GPU box): python tools/conv0_ln_bench.py [B] [T]  -- the LayerNorm-mode conv0 block (WavLM-Large's block 0) at the Large
step's shape (32 x 20 s): forward and backward, us per launch and the gradient's bytes per second.  WAVLM_CONV0_BWD_MFMA=0 times
the VALU form of the backward."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unispeech_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 320000
C = 512
g = torch.Generator().manual_seed(0)
wav = torch.randn(B, T, generator=g).cuda().bfloat16()
W = (0.4 * torch.randn(C, 1, 10, generator=g)).cuda().bfloat16()
gm = (1 + 0.1 * torch.randn(C, generator=g)).cuda().bfloat16()
bt = (0.1 * torch.randn(C, generator=g)).cuda().bfloat16()
cb = (0.2 * torch.randn(C, generator=g)).cuda().bfloat16()
T0 = (T - 10) // 5 + 1
dy = torch.randn(B, T0, C, generator=torch.Generator(device="cuda").manual_seed(1), device="cuda").bfloat16()


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


gb = B * T0 * C * 2 / 1e9
f = timed(lambda: ops.conv0_ln_gelu_fwd(wav, W, gm, bt, 5, 1e-5, torch.bfloat16, bias=cb))
b = timed(lambda: ops.conv0_ln_gelu_bwd(wav, W, gm, bt, dy, 5, 1e-5, bias=cb))
print(f"conv0+LN+GELU B={B} T={T} frames={B * T0} ({gb:.2f} GB): forward {f:.0f} us ({gb / f * 1e3:.2f} TB/s), "
      f"backward {b:.0f} us ({gb / b * 1e3:.2f} TB/s) [WAVLM_CONV0_BWD_MFMA={os.environ.get('WAVLM_CONV0_BWD_MFMA', '1')}]")
# the GroupNorm-mode block (WavLM-Base) at the headline step's shape: 32 x 15 s
Bg, Tg = 32, 240000
wg = torch.randn(Bg, Tg, generator=g).cuda().bfloat16()
T0g = (Tg - 10) // 5 + 1
gbg = Bg * T0g * C * 2 / 1e9
fg = timed(lambda: ops.conv0_gn_gelu_fwd(wg, W, gm, bt, 5, 1e-5, torch.bfloat16))
print(f"conv0+GN+GELU B={Bg} T={Tg} frames={Bg * T0g} ({gbg:.2f} GB): forward (Gram + statistics + apply) {fg:.0f} us "
      f"({gbg / fg * 1e3:.2f} TB/s) [WAVLM_CONV0_FWD_MFMA={os.environ.get('WAVLM_CONV0_FWD_MFMA', '1')}]")
