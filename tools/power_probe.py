"""Socket power and shader clock while one kernel family runs back to back (rocm-smi sampled from a second thread): is the
GEMM rate a per-CU matter or a chip-wide (power / clock) one?  Scenarios: idle; the grouped weight-gradient launch of a Base
block with the one-round split (216 of 256 CUs) and balanced (256 CUs, child process: the switch is read once per process);
fc2 forward (gemm_pp3, K = 3072); fc1 + GELU (heavy epilogue); the vendor library's 8192^3 (torch.matmul); fused attention
forward.  usage (GPU box): python tools/power_probe.py [seconds per scenario]"""
import json
import os
import re
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unispeech_amd import ops  # noqa: E402

SECS = float(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else 4.0
dev, bf = "cuda", torch.bfloat16


def smi():
    try:
        out = subprocess.run(["rocm-smi", "-P", "-c", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        card = d[sorted(k for k in d if k.startswith("card"))[0]]
        pw = next((float(v) for k, v in card.items() if "ower" in k and re.match(r"^[0-9.]+$", str(v))), None)
        sclk = next((float(re.sub(r"[^0-9.]", "", str(v))) for k, v in card.items() if k.lower().startswith("sclk")), None)
        return pw, sclk
    except Exception:
        return None, None


def run(name, fn, flops):
    fn()
    torch.cuda.synchronize()
    samples, stop = [], threading.Event()

    def sampler():
        time.sleep(0.5)
        while not stop.is_set():
            samples.append(smi())
            time.sleep(0.1)
    th = threading.Thread(target=sampler)
    th.start()
    n, t0 = 0, time.time()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < SECS:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    us = e0.elapsed_time(e1) * 1e3 / max(n, 1)
    pw = [p for p, _ in samples if p is not None]
    ck = [c for _, c in samples if c is not None]
    print("%-58s %8.1f us %7.0f TF/s   power %s W   sclk %s MHz   (%d samples)" % (
        name, us, flops / us / 1e6 if flops else 0.0, "%.0f" % (sum(pw) / len(pw)) if pw else "n/a",
        "%.0f" % (sum(ck) / len(ck)) if ck else "n/a", len(samples)), flush=True)


n = 32 * 749
torch.manual_seed(0)
if "--grouped-only" in sys.argv:
    items = [(torch.randn(n, N, device=dev).to(bf), torch.randn(n, K, device=dev).to(bf), torch.zeros(N, K, device=dev, dtype=bf))
             for N, K in [(768, 3072), (3072, 768), (768, 768), (2304, 768)]]
    fl = sum(2.0 * n * a.shape[1] * b.shape[1] for a, b, _ in items)
    run("grouped dW of a block, WAVLM_WGRAD_STREAMK=%s" % os.environ.get("WAVLM_WGRAD_STREAMK", "0"), lambda: ops.gemm_wgrad_grouped(items, bf), fl)
    sys.exit(0)

print("raw sample:", smi())
t_idle = time.time()
time.sleep(1.0)
print("%-58s %s" % ("idle", smi()))
for sk in ("0", "1"):
    env = dict(os.environ, WAVLM_WGRAD_STREAMK=sk)
    sys.stdout.write(subprocess.run([sys.executable, os.path.abspath(__file__), str(SECS), "--grouped-only"], env=env, capture_output=True, text=True).stdout)
    sys.stdout.flush()


def lin(N, K, epi=0, tB=False):
    x, W, b = torch.randn(n, K, device=dev).to(bf), torch.randn(N, K, device=dev).to(bf), torch.randn(N, device=dev).to(bf)
    y = torch.empty(n, N, device=dev, dtype=bf)
    aux = torch.empty(n, N, device=dev, dtype=bf) if epi == 3 else None
    return lambda: ops.gemm(x, W, y, n, N, K, lda=K, ldb=K, ldc=N, bias=b, epi=epi, aux=aux, ld_aux=N)


run("fc2 forward [23968 x 768 x 3072] (gemm_pp3)", lin(768, 3072), 2.0 * n * 768 * 3072)
run("fc1 + GELU + GELU' store [23968 x 3072 x 768] (gemm_pp3)", lin(3072, 768, epi=3), 2.0 * n * 3072 * 768)
a8, b8 = torch.randn(8192, 8192, device=dev).to(bf), torch.randn(8192, 8192, device=dev).to(bf)
c8 = torch.empty(8192, 8192, device=dev, dtype=bf)
run("vendor library 8192^3 (torch.matmul)", lambda: torch.matmul(a8, b8, out=c8), 2.0 * 8192 ** 3)
ops.gemm_set_variant(3)
run("this library 8192^3 NT (gemm_pp)", lambda: ops.gemm(a8, b8, c8, 8192, 8192, 8192, lda=8192, ldb=8192, ldc=8192), 2.0 * 8192 ** 3)
ops.gemm_set_variant(5)
run("this library 8192^3 NT (gemm_w4)", lambda: ops.gemm(a8, b8, c8, 8192, 8192, 8192, lda=8192, ldb=8192, ldc=8192), 2.0 * 8192 ** 3)
ops.gemm_set_variant(0)
# fused attention and a LayerNorm at the step's shapes: which of the non-GEMM families sit at the power limit too
B_, T_, H_, hd_ = 32, 749, 12, 64
qkv = (0.5 * torch.randn(B_, T_, 3 * H_ * hd_, device=dev)).to(bf)
gate = 1 + 0.5 * torch.rand(B_, H_, T_, device=dev)
tab = 0.5 * torch.randn(H_, 2 * T_ - 1, device=dev)
dO = torch.randn(B_, T_, H_ * hd_, device=dev).to(bf)
fl = 4.0 * B_ * H_ * T_ * T_ * hd_
O_, lse_, _ = ops.attn_fused_fwd(qkv, gate, tab, None, H_, hd_ ** -0.5, 0.1, 1234)
run("fused attention forward (12 heads, T = 749, dropout 0.1)", lambda: ops.attn_fused_fwd(qkv, gate, tab, None, H_, hd_ ** -0.5, 0.1, 1234), fl)
run("fused attention backward (dQ + dK/dV + reduction)", lambda: ops.attn_fused_bwd(qkv, O_, dO, lse_, gate, tab, None, H_, hd_ ** -0.5, 0.1, 1234), 2.5 * fl)
x_ = torch.randn(n, 768, device=dev).to(bf)
r_ = torch.randn(n, 768, device=dev).to(bf)
gm_, bt_ = torch.ones(768, device=dev).to(bf), torch.zeros(768, device=dev).to(bf)
run("LayerNorm forward + residual [23968 x 768] (HBM-bound)", lambda: ops.layernorm_fwd(x_, r_, gm_, bt_, 1e-5), 0.0)
