"""Every wavlm_gemm launch of ONE real training step (bench.py's model, batch and path) with its HIP-event duration: which
shapes sit where against the 2.5 PF/s MFMA peak.  usage (GPU box): python tools/gemm_step_table.py [config=base]"""
import collections
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from unispeech_amd import _lib, hostenv, ops  # noqa: E402
from unispeech_amd.optim import FusedAdam  # noqa: E402
from unispeech_amd.pretrain import WavLMCriterion, WavLMPretrainModel  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else "base"
hostenv.cap_threads(4)
dev = torch.device("cuda", 0)
cfg = bench.base_cfg(True, config)
torch.manual_seed(0)
model = WavLMPretrainModel(cfg, None, [range(bench.V)]).to(dev).to(torch.bfloat16).train()
opt = FusedAdam(model.parameters(), lr=5e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01, clip_norm=10.0, model=model)
crit = WavLMCriterion(None, 1.0, 0.0, loss_weights=[10.0], defer_logging=True)
sec = bench.CONFIGS[config]["seconds"]
B, T = bench.BATCH_PER_GPU, int(sec * bench.SR)
g = torch.Generator().manual_seed(1234)
wav = torch.randn(B, T, generator=g).to(dev).to(torch.bfloat16)
pm_cpu = torch.zeros(B, T, dtype=torch.bool)
sample = {"id": torch.arange(B), "net_input": {"source": wav, "padding_mask": pm_cpu.to(dev), "padding_mask_cpu": pm_cpu},
          "target_list": [torch.randint(4, bench.V, (B, int(50 * sec)), generator=g).to(dev)]}
np.random.seed(1337)


def step():
    opt.zero_grad()
    loss, ss, _ = crit(model, sample)
    loss.backward()
    opt.step(grad_mult=1.0 / max(float(ss), 1.0))


for _ in range(6):
    step()
torch.cuda.synchronize()
best = None
for _ in range(3):   # three profiled steps, the one with the smallest GEMM total is tabulated
    ops.prof_enable(True)
    step()
    path = "/tmp/gemm_step_%d.txt" % os.getpid()
    n = _lib.lib().wavlm_prof_dump(path.encode())
    ops.prof_enable(False)
    rows = [l.split() for l in open(path)]
    tot = sum(float(r[8]) for r in rows)
    if best is None or tot < best[0]:
        best = (tot, rows)
tot, rows = best
agg = collections.OrderedDict()
for r in rows:
    key = tuple(r[:8])
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1; a[1] += float(r[8]); a[2] += float(r[9])
EPI = {0: "-", 1: "gelu", 2: "*gelu'", 3: "gelu+g'", 4: "*aux"}
print("%d launches, %.3f ms, %.1f TFLOP -> %.0f TF/s (%.3f of 2500)" % (len(rows), tot, sum(a[2] for a in agg.values()) / 1e3,
                                                                        sum(a[2] for a in agg.values()) / tot, sum(a[2] for a in agg.values()) / tot / 2500))
print("%6s %6s %6s %4s %3s %-12s %3s %4s | %3s %9s %9s %8s %6s" % ("M", "N", "K", "KB", "tr", "epilogue", "spl", "bat", "n", "ms total", "us each", "TF/s", "frac"))
for key, (n, ms, gf) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, N, K, KB, tr, epi, spl, bat = (int(v) for v in key)
    e = EPI[epi & 7] + ("+res" if epi & 16 else "") + ("+csum" if epi & 32 else "")
    print("%6d %6d %6d %4d %3s %-12s %3d %4d | %3d %9.3f %9.1f %8.0f %6.3f" % (
        M, N, K, KB, ("T" if tr & 1 else "N") + ("T" if tr & 2 else "N"), e if M >= 0 else "grouped dW", spl, bat, n, ms, ms / n * 1e3, gf / ms, gf / ms / 2500))
