"""Where do the launch thread's occasional 50 ms stalls come from?  Times the phases of a step on the host (no syncs
inside) and the calls suspected of blocking.  usage: python tools/host_spikes.py [steps] [batch]"""
import gc
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import unispeech_amd.functional as F  # noqa: E402
from unispeech_amd.optim import FusedAdam  # noqa: E402
from unispeech_amd.pretrain import WavLMCriterion, WavLMPretrainModel  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
cfg = bench.base_cfg(True)
torch.manual_seed(0)
model = WavLMPretrainModel(cfg, None, [range(bench.V)]).to(dev).to(torch.bfloat16).train()
opt = FusedAdam(model.parameters(), lr=5e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01, clip_norm=10.0, model=model)
crit = WavLMCriterion(None, 1.0, 0.0, loss_weights=[10.0], defer_logging=True)
T = int(bench.SECONDS * bench.SR)
g = torch.Generator().manual_seed(1234)
wav = torch.randn(B, T, generator=g).to(dev).to(torch.bfloat16)
pm_cpu = torch.zeros(B, T, dtype=torch.bool)
sample = {"id": torch.arange(B), "net_input": {"source": wav, "padding_mask": pm_cpu.to(dev), "padding_mask_cpu": pm_cpu},
          "target_list": [torch.randint(4, bench.V, (B, int(50 * bench.SECONDS)), generator=g).to(dev)]}
np.random.seed(1337)

acc = {}
orig_h2d = F.h2d


def timed_h2d(a, d):
    t = time.perf_counter()
    r = orig_h2d(a, d)
    acc["h2d"] = acc.get("h2d", 0.0) + (time.perf_counter() - t) * 1e3
    return r


F.h2d = timed_h2d
# every ops.* entry point timed: the slowest call of a step names the launch (or allocation) that blocked
import types
import unispeech_amd.ops as OPS  # noqa: E402
slow = {}


timeline = []


def _wrap(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        timeline.append((t, name))
        r = fn(*a, **k)
        d = (time.perf_counter() - t) * 1e3
        if d > slow.get("max", (0.0, ""))[0]:
            slow["max"] = (d, name)
        return r
    return w


for _n, _f in list(vars(OPS).items()):
    if isinstance(_f, types.FunctionType) and not _n.startswith("_") and _n not in ("dt", "ptr", "stream", "workspace", "check"):
        setattr(OPS, _n, _wrap(_n, _f))
_orig_empty = torch.empty


def _empty(*a, **k):
    t = time.perf_counter()
    r = _orig_empty(*a, **k)
    d = (time.perf_counter() - t) * 1e3
    if d > slow.get("max", (0.0, ""))[0]:
        slow["max"] = (d, "torch.empty%s" % (tuple(a[0]) if a and isinstance(a[0], (tuple, list)) else a[:1],))
    return r


torch.empty = _empty
gcs = []
gc.callbacks.append(lambda phase, info: gcs.append((phase, info["generation"], time.perf_counter())))
for _ in range(3):
    opt.zero_grad(); loss, ss, _ = crit(model, sample); loss.backward(); opt.step(grad_mult=1.0 / max(float(ss), 1.0))
gc.collect(); gc.freeze()
for it in range(steps):
    torch.cuda.synchronize()
    acc.clear(); gcs.clear(); slow.clear(); timeline.clear()
    ms0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    opt.zero_grad()
    t1 = time.perf_counter()
    loss, ss, _ = crit(model, sample)
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    opt.step(grad_mult=1.0 / max(float(ss), 1.0))
    t4 = time.perf_counter()
    gct = sum(b[2] - a[2] for a, b in zip(gcs[::2], gcs[1::2])) * 1e3
    ms1 = torch.cuda.memory_stats()
    dm = {k: ms1.get(k, 0) - ms0.get(k, 0) for k in ("num_device_alloc", "num_device_free", "num_alloc_retries")}
    print("step %2d host ms: zero_grad %.1f  fwd+loss %.1f (h2d %.1f)  bwd %.1f  opt %.1f | gc %.1f ms/%d | hipMalloc %d hipFree %d retries %d reserved %.0f MB"
          % (it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, acc.get("h2d", 0.0), (t3 - t2) * 1e3, (t4 - t3) * 1e3, gct, len(gcs) // 2,
             dm["num_device_alloc"], dm["num_device_free"], dm["num_alloc_retries"], ms1.get("reserved_bytes.all.current", 0) / 1e6), flush=True)
    print("        slowest host call: %.1f ms in %s" % slow.get("max", (0.0, "-")), flush=True)
    if os.environ.get("TIMELINE") and it == steps - 1:
        tl = [(t, n) for t, n in timeline if t >= t2]
        gaps = sorted(((b[0] - a[0]) * 1e3, a[1], b[1], (a[0] - t2) * 1e3) for a, b in zip(tl, tl[1:]))[-12:]
        print("        backward: %d ops calls; largest gaps between consecutive calls (ms, after -> before, at ms):" % len(tl))
        for gdt, an, bn, at in sorted(gaps, key=lambda x: x[3]):
            print("          %.2f  %s -> %s  @%.1f" % (gdt, an, bn, at))
