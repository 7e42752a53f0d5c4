"""Where the launch thread's time goes, by section of a pre-training step: perf_counter around the extractor + encoder
forward, the loss head, backward, the optimizer; then cProfile sorted by cumulative time.  The GPU is parked on a spin
kernel while a step is enqueued, so nothing here waits for the device.
usage (GPU box): python tools/host_sections.py [steps] [batch]"""
import cProfile
import gc
import io
import os
import pstats
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
B, T = (int(sys.argv[2]) if len(sys.argv) > 2 else bench.BATCH_PER_GPU), int(bench.SECONDS * bench.SR)
g = torch.Generator().manual_seed(1234)
wav = torch.randn(B, T, generator=g).to(dev).to(torch.bfloat16)
pm_cpu = torch.zeros(B, T, dtype=torch.bool)
sample = {"id": torch.arange(B), "net_input": {"source": wav, "padding_mask": pm_cpu.to(dev), "padding_mask_cpu": pm_cpu},
          "target_list": [torch.randint(4, bench.V, (B, int(50 * bench.SECONDS)), generator=g).to(dev)]}
np.random.seed(1337)
sec = {"zero_grad": [], "model_fwd": [], "criterion": [], "backward": [], "opt_step": []}


def step(rec=None):
    t = [time.perf_counter()]
    opt.zero_grad(); t.append(time.perf_counter())
    net = model(target_list=sample["target_list"], **sample["net_input"]); t.append(time.perf_counter())
    loss, ss, _ = crit.get_loss(model, sample, net); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt.step(grad_mult=1.0 / max(float(ss), 1.0)); t.append(time.perf_counter())
    if rec is not None:
        for k, a, b in zip(sec, t[:-1], t[1:]):
            rec[k].append((b - a) * 1e3)


for _ in range(4):
    step()
torch.cuda.synchronize()
gc.collect()
gc.freeze()
for _ in range(steps):
    torch.cuda.synchronize()
    torch.cuda._sleep(200_000_000)
    step(sec)
torch.cuda.synchronize()
tot = 0.0
for k, v in sec.items():
    v.sort()
    print("%-10s median %.2f ms  (min %.2f max %.2f)" % (k, v[len(v) // 2], v[0], v[-1]))
    tot += v[len(v) // 2]
print("sum of medians %.2f ms" % tot)
pr = cProfile.Profile()
torch.cuda.synchronize()
torch.cuda._sleep(400_000_000)
pr.enable()
for _ in range(3):
    step()
pr.disable()
torch.cuda.synchronize()
for key in ("cumulative", "tottime"):
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats(key).print_stats(32)
    print("\n".join(l[:160] for l in out.getvalue().split("\n")))
