"""Time of the grouped weight-gradient launch of one WavLM-Base encoder block (q|k|v, out_proj, fc1, fc2 over 32 x 749 frames)
under the current environment (WAVLM_WGRAD_STREAMK, WAVLM_SK_SEG_COST, WAVLM_SK_SPREAD, WAVLM_GEMM_W4).
usage (GPU box): python tools/wgrad_grouped_bench.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unispeech_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32 * 749
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)
items = []
for N, K in [(768, 3072), (3072, 768), (768, 768), (2304, 768)]:
    items.append((torch.randn(n, N, device=dev).to(bf), torch.randn(n, K, device=dev).to(bf), torch.zeros(N, K, device=dev, dtype=bf)))
for _ in range(5):
    ops.gemm_wgrad_grouped(items, bf)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(3):
    e0.record()
    for _ in range(20):
        ops.gemm_wgrad_grouped(items, bf)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
fl = sum(2.0 * n * a.shape[1] * b.shape[1] for a, b, _ in items)
print("%-60s %7.1f us  %6.0f TF/s" % (" ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("WAVLM_")) or "(default)", best * 1e3, fl / best / 1e9))
