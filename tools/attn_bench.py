"""Times the fused attention forward / backward at the WavLM-Base step shape (B=32, T=749, H=12, hd=64) with the
gated relative-position bias and attention dropout on.  For per-kernel numbers run under
`rocprofv3 --kernel-trace --stats`."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unispeech_amd import ops  # noqa: E402

dev = "cuda"
B, T, H, hd = 32, 749, 12, 64
D = H * hd
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
qkv = (0.5 * torch.randn(B, T, 3 * D, device=dev)).to(torch.bfloat16)
gate = 1 + 0.5 * torch.rand(B, H, T, device=dev)
tab = 0.5 * torch.randn(H, 2 * T - 1, device=dev)
dO = torch.randn(B, T, D, device=dev).to(torch.bfloat16)


def timeit(fn, name, flops):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%-28s %8.3f ms  %7.1f TF/s" % (name, ms, flops / ms / 1e9), flush=True)


fl = 4.0 * B * H * T * T * hd
O, lse, _ = ops.attn_fused_fwd(qkv, gate, tab, None, H, hd ** -0.5, p, 1234)
timeit(lambda: ops.attn_fused_fwd(qkv, gate, tab, None, H, hd ** -0.5, p, 1234), "attention fwd (p=%.2f)" % p, fl)
timeit(lambda: ops.attn_fused_bwd(qkv, O, dO, lse, gate, tab, None, H, hd ** -0.5, p, 1234), "attention bwd (dq+dkv+red)", 2.5 * fl)
# stored probabilities (round 5): the forward writes P (fp16, sign = dropped), the backward kernels read it.  The pstore is
# allocated once here (the training path allocates it inside the block's `saved` region).
L_ = ops._lib.lib()
nps = int(L_.wavlm_attn_fused_pstore_bytes(B, H, T))
pst = torch.empty(nps, dtype=torch.uint8, device=dev)
O_s, lse_s = torch.empty_like(O), torch.empty_like(lse)


def fwd_store():
    ops.check(L_.wavlm_attn_fused_fwd_p(ops.ptr(qkv), ops.ptr(O_s), ops.ptr(lse_s), ops.ptr(gate), ops.ptr(tab), None, ops.ptr(pst), nps,
                                        B, H, T, hd, hd ** -0.5, p, 1234, ops.stream()), "fwd_p")


fwd_store()
torch.cuda.synchronize()
print("stored-P forward == plain forward: O max |diff| %.3e, lse max |diff| %.3e (pstore %.1f MB)"
      % ((O_s.float() - O.float()).abs().max().item(), (lse_s - lse).abs().max().item(), nps / 1e6), flush=True)
timeit(fwd_store, "fwd + P store (p=%.2f)" % p, fl)
timeit(lambda: ops.attn_fused_bwd(qkv, O_s, dO, lse_s, gate, tab, None, H, hd ** -0.5, p, 1234, pstore=pst), "bwd from stored P", 2.5 * fl)
# stored dropout BITS (round 6, WAVLM_ATTN_STORE_P=bits): the forward writes its decisions (27 MB), both backward kernels read them
O_b, lse_b, pb = ops.attn_fused_fwd(qkv, gate, tab, None, H, hd ** -0.5, p, 1234, store_p="bits")
torch.cuda.synchronize()
print("bit-storing forward == plain forward: O %s, lse %s (%.1f MB of bit words)" % (torch.equal(O_b, O), torch.equal(lse_b, lse), pb.numel() / 1e6), flush=True)
timeit(lambda: ops.attn_fused_fwd(qkv, gate, tab, None, H, hd ** -0.5, p, 1234, store_p="bits"), "fwd + bit store (p=%.2f)" % p, fl)
timeit(lambda: ops.attn_fused_bwd(qkv, O_b, dO, lse_b, gate, tab, None, H, hd ** -0.5, p, 1234, pstore=pb), "bwd from stored bits", 2.5 * fl)
g_b = ops.attn_fused_bwd(qkv, O_b, dO, lse_b, gate, tab, None, H, hd ** -0.5, p, 1234, pstore=pb)
g_r = ops.attn_fused_bwd(qkv, O, dO, lse, gate, tab, None, H, hd ** -0.5, p, 1234)
print("stored-bits backward == recompute backward (bit for bit): %s" % all(torch.equal(a, b_) for a, b_ in zip(g_b, g_r)), flush=True)
g_s = ops.attn_fused_bwd(qkv, O_s, dO, lse_s, gate, tab, None, H, hd ** -0.5, p, 1234, pstore=pst)
for nm, a, b_ in zip(("dqkv", "dgate", "dtab"), g_s, g_r):
    print("stored-P backward vs recompute backward, %s: max |diff| / max |ref| = %.3e" % (nm, (a.float() - b_.float()).abs().max().item() / b_.float().abs().max().item()), flush=True)
if os.environ.get("ATTN_BENCH_DBIAS", "0") == "1":  # with the q|k|v bias gradient delivered by the backward kernels
    db = torch.zeros(3 * D, device=dev)
    timeit(lambda: ops.attn_fused_bwd(qkv, O, dO, lse, gate, tab, None, H, hd ** -0.5, p, 1234, dbias=db, dbias_accumulate=True),
           "attention bwd + dbias", 2.5 * fl)
timeit(lambda: ops.attn_fused_fwd(qkv, None, None, None, H, hd ** -0.5, 0.0, 0), "fwd, no bias, no dropout", fl)

# backward variants: which part of the dQ / dK-dV kernels costs what (per-kernel times: rocprofv3 --kernel-trace --stats,
# the template arguments in the kernel names are <DROP, TAB> / <DROP>)
for (use_tab, pp) in ((True, 0.0), (False, p), (False, 0.0)):
    g_, t_ = (gate, tab) if use_tab else (None, None)
    O2, lse2, _ = ops.attn_fused_fwd(qkv, g_, t_, None, H, hd ** -0.5, pp, 1234)
    timeit(lambda: ops.attn_fused_bwd(qkv, O2, dO, lse2, g_, t_, None, H, hd ** -0.5, pp, 1234),
           "bwd tab=%d p=%.2f" % (use_tab, pp), 2.5 * fl)
