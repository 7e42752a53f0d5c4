"""Which call is the launch thread sitting in when it stalls?  sys.setprofile / threading.setprofile on every thread (the
autograd engine runs backward on its own): any Python or C call that takes longer than 20 ms is recorded with its name and the
names of its callers.  usage (GPU box): python tools/host_stall_trace.py [steps=30]"""
import collections
import gc
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from unispeech_amd.optim import FusedAdam  # noqa: E402
from unispeech_amd.pretrain import WavLMCriterion, WavLMPretrainModel  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
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

slow = []
local = threading.local()


def prof(frame, event, arg):
    st = getattr(local, "st", None)
    if st is None:
        st = local.st = []
    if event in ("call", "c_call"):
        name = frame.f_code.co_name if event == "call" else getattr(arg, "__qualname__", None) or getattr(arg, "__name__", repr(arg))
        st.append((name, time.perf_counter()))
    elif event in ("return", "c_return", "c_exception") and st:
        name, t0 = st.pop()
        dt = time.perf_counter() - t0
        if dt > 0.020:
            slow.append((dt, name, [n for n, _ in st[-6:]], threading.current_thread().name))


threading.setprofile(prof)
sys.setprofile(prof)
for _ in range(steps):
    step()
sys.setprofile(None)
threading.setprofile(None)
torch.cuda.synchronize()
# innermost frames only: drop an entry if a later-recorded (i.e. outer) one contains it -- keep the leaves
leaves = collections.Counter()
tot = collections.Counter()
for dt, name, stack, th in slow:
    is_leaf = not any(name in s2 and dt2 < dt * 1.001 and n2 != name for dt2, n2, s2, _ in slow if dt2 <= dt and (dt - dt2) < 0.002 and name in s2)
    key = "%s  <- %s  [%s]" % (name, " <- ".join(reversed(stack[-4:])), th)
    leaves[key] += 1
    tot[key] += dt
print("calls > 20 ms over %d steps (count, total ms, name <- callers [thread]):" % steps)
for key, n in sorted(leaves.items(), key=lambda kv: -tot[kv[0]])[:25]:
    print("%4d %9.1f  %s" % (n, tot[key] * 1e3, key))
