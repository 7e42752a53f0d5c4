"""How far ahead of the GPU is the launch thread?  Per step: host time to ENQUEUE the step (step() called on an idle GPU,
no synchronisation inside) against the time until the GPU has finished it.  usage: python tools/host_time.py [steps] [batch]"""
import gc
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from unispeech_amd.optim import FusedAdam  # noqa: E402
from unispeech_amd.pretrain import WavLMCriterion, WavLMPretrainModel  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
cfg = bench.base_cfg(True)
torch.manual_seed(0)
model = WavLMPretrainModel(cfg, None, [range(bench.V)]).to(dev).to(torch.bfloat16).train()
opt = FusedAdam(model.parameters(), lr=5e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01, clip_norm=10.0, model=model)
crit = WavLMCriterion(None, 1.0, 0.0, loss_weights=[10.0], defer_logging=True)
B = int(sys.argv[2]) if len(sys.argv) > 2 else bench.BATCH_PER_GPU  # batch 1: the same launches with ~no GPU work
T = int(bench.SECONDS * bench.SR)
g = torch.Generator().manual_seed(1234)
wav = torch.randn(B, T, generator=g).to(dev).to(torch.bfloat16)
pm_cpu = torch.zeros(B, T, dtype=torch.bool)
sample = {"id": torch.arange(B), "net_input": {"source": wav, "padding_mask": pm_cpu.to(dev), "padding_mask_cpu": pm_cpu},
          "target_list": [torch.randint(4, bench.V, (B, int(50 * bench.SECONDS)), generator=g).to(dev)]}
np.random.seed(1337)


def step():
    opt.zero_grad()
    loss, ss, _ = crit(model, sample)
    loss.backward()
    opt.step(grad_mult=1.0 / max(float(ss), 1.0))


for _ in range(3):
    step()
gc.collect()
gc.freeze()
host, total = [], []
for _ in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3)
    total.append((t2 - t0) * 1e3)
print("host enqueue ms per step:", " ".join("%.1f" % h for h in host))
print("until GPU done   ms     :", " ".join("%.1f" % t for t in total))
print("median host %.1f ms, median total %.1f ms" % (sorted(host)[len(host) // 2], sorted(total)[len(total) // 2]))
