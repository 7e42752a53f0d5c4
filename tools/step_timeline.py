"""Per-step timeline of a free-running training loop (no synchronisation inside): for each of N steps the host's enqueue
time and the GPU-side duration (delta of events recorded at the step boundaries), the host's lead over the GPU at every
step start, and the steps in which either side stalls.  This is the measurement behind `ms_per_step` vs
`gpu_busy_ms_per_step`: a GPU-side step longer than busy = the GPU waited for the launch thread there.
usage (GPU box): python tools/step_timeline.py [steps=60] [ENV=VALUE ...]"""
import gc
import os
import sys
import time

for a in sys.argv[2:]:
    k, v = a.split("=", 1)
    os.environ[k] = v

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from unispeech_amd.optim import FusedAdam  # noqa: E402
from unispeech_amd.pretrain import WavLMCriterion, WavLMPretrainModel  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0)
cfg = bench.base_cfg(True)
torch.manual_seed(0)
model = WavLMPretrainModel(cfg, None, [range(bench.V)]).to(dev).to(torch.bfloat16).train()
opt = FusedAdam(model.parameters(), lr=5e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01, clip_norm=10.0, model=model)
crit = WavLMCriterion(None, 1.0, 0.0, loss_weights=[10.0], defer_logging=True)
B, T = bench.BATCH_PER_GPU, int(bench.SECONDS * bench.SR)
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


for _ in range(8):
    step()
torch.cuda.synchronize()
gc.collect()
gc.freeze()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
host = []
THROTTLE = int(os.environ.get("THROTTLE", "0"))   # > 0: at most that many steps in flight (wait for step i - THROTTLE before enqueueing step i)
torch.cuda.synchronize()
t_start = time.perf_counter()
ev[0].record()
for i in range(steps):
    if THROTTLE and i >= THROTTLE:
        ev[i + 1 - THROTTLE].synchronize()
    t0 = time.perf_counter()
    step()
    ev[i + 1].record()
    host.append((t0 - t_start, time.perf_counter() - t0))
torch.cuda.synchronize()
wall = time.perf_counter() - t_start
gpu = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
gpu_end = np.cumsum(gpu)   # GPU-side end of step i, ms since ev[0]
med_g, med_h = float(np.median(gpu)), float(np.median([h for _, h in host])) * 1e3
print("steps %d: wall %.2f ms/step; GPU-side step median %.2f ms (min %.2f, max %.2f); host enqueue median %.2f ms (max %.2f)"
      % (steps, wall / steps * 1e3, med_g, min(gpu), max(gpu), med_h, max(h for _, h in host) * 1e3))
nst = sum(1 for _, h in host if h * 1e3 > 30.0)
print("host stalls > 30 ms: %d in %d steps (total %.0f ms); GPU-side steps > 1.08 x median: %d (excess %.0f ms)"
      % (nst, steps, sum(h * 1e3 for _, h in host if h * 1e3 > 30.0), sum(1 for g_ in gpu if g_ > 1.08 * med_g),
         sum(g_ - med_g for g_ in gpu if g_ > 1.08 * med_g)))
if os.environ.get("QUIET") == "1":
    sys.exit(0)
print("%4s %10s %10s %12s" % ("step", "host ms", "gpu ms", "host lead ms"))
for i, ((ts, th), gd) in enumerate(zip(host, gpu)):
    lead = (gpu_end[i - 1] if i else 0.0) - ts * 1e3   # > 0: the GPU is still busy with earlier steps when the host starts step i
    flag = ("  <-- host stall" if th * 1e3 > 3 * med_h else "") + ("  <-- GPU-side long" if gd > 1.08 * med_g else "")
    if flag or i < 4 or i % 10 == 0:
        print("%4d %10.2f %10.2f %12.1f%s" % (i, th * 1e3, gd, lead, flag))
