"""What a narrower (8-bit) or absent GELU' store could buy the two heavy-epilogue launches of an encoder block (VERDICT r4,
item 3a / 3b) -- measured as bounds with the product library, no new kernel:
  fc1 forward  23968 x 3072 x 768: epilogue 3 (bias + GELU + GELU' -> aux)   vs the same WITHOUT the aux store   vs plain + bias
  fc2 dX       23968 x 3072 x 768 (NT): epilogue 4 (x aux) + column sums      vs epilogue 0 + column sums (no aux read) vs plain
The middle column is the upper bound for ANY change to the aux tensor's width: nothing narrower can beat not writing /
reading it at all.  usage (GPU box): python tools/gemm_aux_bound.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unispeech_amd import ops  # noqa: E402

dev, bf = "cuda", torch.bfloat16
n, D, F = 23968, 768, 3072
torch.manual_seed(0)
x = (0.5 * torch.randn(n, D, device=dev)).to(bf)
W1 = (D ** -0.5 * torch.randn(F, D, device=dev)).to(bf)
b1 = (0.1 * torch.randn(F, device=dev)).to(bf)
h = torch.empty(n, F, device=dev, dtype=bf)
g = torch.empty(n, F, device=dev, dtype=bf)
df = (0.5 * torch.randn(n, D, device=dev)).to(bf)
W2 = (F ** -0.5 * torch.randn(D, F, device=dev)).to(bf)     # fc2.weight [D, F]: dX = df @ W2 -> B is K-strided (transB)
du = torch.empty(n, F, device=dev, dtype=bf)
db1 = torch.zeros(F, device=dev, dtype=bf)


def timeit(fn, name, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    fl = 2.0 * n * D * F
    print("%-64s %7.1f us  %6.0f TF/s" % (name, best, fl / best / 1e6), flush=True)
    return best


print("fc1 forward, 23968 x 3072 x 768")
a = timeit(lambda: ops.gemm(x, W1, h, n, F, D, lda=D, ldb=D, ldc=F, bias=b1, epi=3, aux=g, ld_aux=F), "  bias + GELU + GELU' store (the step's launch)")
b = timeit(lambda: ops.gemm(x, W1, h, n, F, D, lda=D, ldb=D, ldc=F, bias=b1, epi=3), "  bias + GELU, NO GELU' store (bound for any narrower store)")
c = timeit(lambda: ops.gemm(x, W1, h, n, F, D, lda=D, ldb=D, ldc=F, bias=b1), "  bias only (plain epilogue)")
c0 = timeit(lambda: ops.gemm(x, W1, h, n, F, D, lda=D, ldb=D, ldc=F), "  no bias, plain")
print("  -> the aux store costs %.1f us of %.1f; the GELU arithmetic + table %.1f us; the bias load %.1f us" % (a - b, a, b - c, c - c0))
print("fc2 dX, 23968 x 3072 x 768 (B K-strided)")
a2 = timeit(lambda: ops.gemm(df, W2, du, n, F, D, lda=D, ldb=F, ldc=F, transB=True, epi=4, aux=g, ld_aux=F, colsum=db1, colsum_accumulate=True),
            "  x GELU' + column sums (the step's launch)")
b2 = timeit(lambda: ops.gemm(df, W2, du, n, F, D, lda=D, ldb=F, ldc=F, transB=True, colsum=db1, colsum_accumulate=True),
            "  column sums, NO aux read (bound for any narrower aux)")
c2 = timeit(lambda: ops.gemm(df, W2, du, n, F, D, lda=D, ldb=F, ldc=F, transB=True), "  plain")
print("  -> the aux read costs %.1f us of %.1f; the fused column sums %.1f us" % (a2 - b2, a2, b2 - c2))
print("per step (12 blocks): aux store + aux read = %.2f ms -- the most an 8-bit GELU' (half the bytes) could save is about half of that"
      % (12 * ((a - b) + (a2 - b2)) / 1e3))
# the residual read of the dX launches that add the gradient of the tensor's other consumer (fc1's dX + LayerNorm dx; q|k|v's dX + dx)
print("fc1 dX, 23968 x 768 x 3072 (B K-strided), + bf16 residual")
W1t = (F ** -0.5 * torch.randn(F, D, device=dev)).to(bf)    # fc1.weight [F, D]: dX = du @ W1
duu = (0.5 * torch.randn(n, F, device=dev)).to(bf)
dres = (0.5 * torch.randn(n, D, device=dev)).to(bf)
dh = torch.empty(n, D, device=dev, dtype=bf)
r1 = timeit(lambda: ops.gemm(duu, W1t, dh, n, D, F, lda=F, ldb=D, ldc=D, transB=True, res=dres, ld_res=D), "  + residual (the step's launch)")
r0 = timeit(lambda: ops.gemm(duu, W1t, dh, n, D, F, lda=F, ldb=D, ldc=D, transB=True), "  plain")
print("  -> the residual read costs %.1f us of %.1f (24 such launches per step: %.2f ms)" % (r1 - r0, r1, 24 * (r1 - r0) / 1e3))
