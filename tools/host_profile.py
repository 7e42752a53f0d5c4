"""cProfile of the launch thread: where does the host time of a pre-training step go?  Batch 1 (the same launches with almost
no GPU work), GC frozen, 6 profiled steps; prints the 40 most expensive functions by internal time.
usage (GPU box): python tools/host_profile.py [steps] [batch]"""
import cProfile
import gc
import io
import os
import pstats
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from unispeech_amd.optim import FusedAdam  # noqa: E402
from unispeech_amd.pretrain import WavLMCriterion, WavLMPretrainModel  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
cfg = bench.base_cfg(True)
torch.manual_seed(0)
model = WavLMPretrainModel(cfg, None, [range(bench.V)]).to(dev).to(torch.bfloat16).train()
opt = FusedAdam(model.parameters(), lr=5e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01, clip_norm=10.0, model=model)
crit = WavLMCriterion(None, 1.0, 0.0, loss_weights=[10.0], defer_logging=True)
B, T = (int(sys.argv[2]) if len(sys.argv) > 2 else 1), int(bench.SECONDS * bench.SR)
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


for _ in range(4):
    step()
torch.cuda.synchronize()
gc.collect()
gc.freeze()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
out = io.StringIO()
st = pstats.Stats(pr, stream=out).sort_stats("tottime")
st.print_stats(28)
txt = out.getvalue()
print("\n".join(l[:170] for l in txt.split("\n")))
