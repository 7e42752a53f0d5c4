#!/usr/bin/env python
"""BUILD CONTAINER ONLY (needs /root/reference): what `cpu_baseline.kind: "port"` stands in for.

bench.py's cpu_baseline leg times the oracle (oracle/wavlm_oracle.py, plain fp32 PyTorch on the host cores) because the GPU box has
no /root/reference.  This script times BOTH on the same host, same threads, same sample -- the reference's own fairseq `WavLMModel` +
`WavLMCriterion` (imported from /root/reference through oracle/ref_shim.py) and the oracle, forward + loss + backward of ONE 15 s
utterance through the 12-layer Base model, 1 warm-up + 3 timed runs each, median -- so that the port's cost can be read against the
reference's.  Output: one JSON line; committed as profiles/r06/ref_vs_oracle_cpu.txt.
"""
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ref_shim  # noqa: E402
from oracle import wavlm_oracle as O  # noqa: E402
from test_oracle_vs_reference import BASE, _Dict  # noqa: E402


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
    torch.set_num_threads(threads)
    WavLMModel, WavLMConfig, WavLMCriterion, _, cmi = ref_shim.fairseq_wavlm()
    cfg = WavLMConfig()
    for k, v in BASE.items():
        setattr(cfg, k, v)
    V = 504
    torch.manual_seed(0)
    model = WavLMModel(cfg, SimpleNamespace(sample_rate=16000), [_Dict(V)])
    model.train()
    crit = WavLMCriterion(SimpleNamespace(), 1.0, 0.0, loss_weights=[10.0])
    B, T = 1, int(16000 * seconds)
    g = torch.Generator().manual_seed(2)
    wav = torch.randn(B, T, generator=g)
    pm = torch.zeros(B, T, dtype=torch.bool)
    target = torch.randint(4, V, (B, int(50 * seconds)), generator=g)
    sample = {"id": torch.arange(B), "net_input": {"source": wav, "padding_mask": pm}, "target_list": [target]}

    def run_ref():
        for p in model.parameters():
            p.grad = None
        np.random.seed(123)
        t0 = time.time()
        loss, _, _ = crit(model, sample)
        loss.backward()
        return time.time() - t0, loss.item()

    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
    ocfg = SimpleNamespace(**BASE)
    Tp = T
    for _, k, s in eval(BASE["conv_feature_layers"]):
        Tp = (Tp - k) // s + 1
    Tp = min(Tp, target.shape[1])

    def run_oracle():
        for p in sd.values():
            p.grad = None
        np.random.seed(123)
        t0 = time.time()
        m = cmi((B, Tp), torch.zeros(B, Tp, dtype=torch.bool), cfg.mask_prob, cfg.mask_length, cfg.mask_selection, cfg.mask_other,
                min_masks=2, no_overlap=False, min_space=1)
        net = O.pretrain_forward(sd, ocfg, wav, [target], pm, torch.from_numpy(m), [V])
        loss, _, _ = O.criterion(net, 1.0, 0.0, [10.0])
        loss.backward()
        return time.time() - t0, loss.item()

    out = {}
    for name, fn in (("reference", run_ref), ("oracle", run_oracle), ("reference_again", run_ref)):
        fn()
        ts, ls = zip(*[fn() for _ in range(3)])
        out[name] = {"seconds_per_run": [round(t, 3) for t in ts], "median_s": round(sorted(ts)[1], 3),
                     "audio_s_per_s": round(seconds / sorted(ts)[1], 2), "loss": ls[0]}
    out["oracle_over_reference_time"] = round(out["oracle"]["median_s"] / out["reference"]["median_s"], 3)
    out["sample"] = "fwd + loss + bwd, WavLM-Base 12 layers fp32, B=1 x %g s, %d threads of %d CPUs (build container)" % (
        seconds, threads, os.cpu_count())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
