"""A/B of the two 256 x 256 kernels (eight-wave ping-pong gemm_pp.hip = variant 3, four-wave gemm_w4.hip = variant 5) on the
long-K shapes of the WavLM-Base step + large squares; correctness of variant 5 first (tests/gpu_checks.check_gemm_pp).
usage (GPU box): python tools/w4_ab.py [check|time|all]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from unispeech_amd import ops  # noqa: E402

dev, bf = "cuda", torch.bfloat16
what = sys.argv[1] if len(sys.argv) > 1 else "all"


def rnd(*shape):
    return torch.randn(*shape, device=dev, dtype=bf)


def timeit(fn, flops, name, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, flops / ms / 1e9


if what in ("check", "all"):
    import gpu_checks as G
    bad = 0
    for name, e, tol in G.check_gemm_pp(5, "gemm_w4"):
        ok = e <= tol
        bad += not ok
        print("%-70s %.3e %s" % (name, e, "ok" if ok else "FAIL"))
    # fused column sums + GELU' epilogue through variant 5
    ops.gemm_set_variant(5)
    try:
        for name, e, tol in G.check_gemm_colsum():
            ok = e <= tol
            bad += not ok
            print("%-70s %.3e %s" % ("[v5] " + name, e, "ok" if ok else "FAIL"))
    finally:
        ops.gemm_set_variant(0)
    print("CHECK", "FAILED" if bad else "PASSED", bad)

if what in ("time", "all"):
    n = 32 * 749
    shapes = []
    for S in (4096, 8192):
        A, B, C = rnd(S, S), rnd(S, S), torch.empty(S, S, device=dev, dtype=bf)
        shapes.append(("square NN %d^3" % S, 2.0 * S ** 3, lambda A=A, B=B, C=C, S=S: ops.gemm(A, B, C, S, S, S, lda=S, ldb=S, ldc=S)))
        shapes.append(("square TT %d^3" % S, 2.0 * S ** 3, lambda A=A, B=B, C=C, S=S: ops.gemm(A, B, C, S, S, S, lda=S, ldb=S, ldc=S, transA=True, transB=True)))
    Bz, Tin, Cc, Tout = 32, 47999, 512, 23999
    x, Wf = rnd(Bz, Tin, Cc), rnd(Cc, 3 * Cc)
    y, u = torch.empty(Bz, Tout, Cc, device=dev, dtype=bf), torch.empty(Bz, Tout, Cc, device=dev, dtype=bf)
    shapes.append(("conv1 fwd + gelu + gelu' store", 2.0 * Bz * Tout * Cc * 3 * Cc,
                   lambda: ops.gemm(x, Wf, y, Tout, Cc, 3 * Cc, lda=2 * Cc, ldb=3 * Cc, ldc=Cc, batch=(Bz, 1), sA=(Tin * Cc, 0),
                                    sC=(Tout * Cc, 0), epi=3, aux=u, ld_aux=Cc, sAux=(Tout * Cc, 0))))
    shapes.append(("conv1 fwd plain", 2.0 * Bz * Tout * Cc * 3 * Cc,
                   lambda: ops.gemm(x, Wf, y, Tout, Cc, 3 * Cc, lda=2 * Cc, ldb=3 * Cc, ldc=Cc, batch=(Bz, 1), sA=(Tin * Cc, 0),
                                    sC=(Tout * Cc, 0))))
    dWf = torch.empty(Cc, 3 * Cc, device=dev, dtype=bf)
    sp = ops.pick_split(Cc, 3 * Cc, Bz * ((Tout + 63) // 64))
    shapes.append(("conv1 dW split=%d" % sp, 2.0 * Bz * Tout * Cc * 3 * Cc,
                   lambda: ops.gemm(y, x, dWf, Cc, 3 * Cc, Tout, lda=Cc, ldb=2 * Cc, ldc=3 * Cc, transA=True, transB=True, KB=Bz,
                                    sA_kb=Tout * Cc, sB_kb=Tin * Cc, split_k=sp)))
    for (N, K, nm) in [(3072, 768, "fc1"), (768, 3072, "fc2"), (2304, 768, "qkv")]:
        dy, xx, dW = rnd(n, N), rnd(n, K), torch.empty(N, K, device=dev, dtype=bf)
        sp = ops.pick_split(N, K, (n + 63) // 64)
        shapes.append(("dW %s [%dx%dx%d] split=%d" % (nm, N, K, n, sp), 2.0 * n * N * K,
                       lambda dy=dy, xx=xx, dW=dW, N=N, K=K, sp=sp: ops.gemm(dy, xx, dW, N, K, n, lda=N, ldb=K, ldc=K, transA=True,
                                                                              transB=True, split_k=sp)))
    # N = 2048-class forward (stays on the 256-wide kernel)
    xx, W, yy = rnd(n, 768), rnd(2048, 768), torch.empty(n, 2048, device=dev, dtype=bf)
    shapes.append(("fwd [%dx2048x768]" % n, 2.0 * n * 2048 * 768, lambda: ops.gemm(xx, W, yy, n, 2048, 768, lda=768, ldb=768, ldc=2048)))
    print("%-40s %10s %10s   %s" % ("shape", "pp8 TF/s", "w4 TF/s", "w4 / pp8 time"))
    for name, fl, fn in shapes:
        res = {}
        for v in (3, 5, 3, 5):
            ops.gemm_set_variant(v)
            ms, tf = timeit(fn, fl, name)
            res.setdefault(v, []).append((ms, tf))
        ops.gemm_set_variant(0)
        a = min(m for m, _ in res[3]); b = min(m for m, _ in res[5])
        print("%-40s %10.1f %10.1f   %.3f   (%.1f / %.1f us)" % (name, fl / a / 1e9, fl / b / 1e9, b / a, a * 1e3, b * 1e3), flush=True)
    # grouped weight gradients of one block
    items = [(rnd(n, N), rnd(n, K), torch.zeros(N, K, device=dev, dtype=bf)) for (N, K) in [(768, 3072), (3072, 768), (768, 768), (2304, 768)]]
    fl = sum(2.0 * n * a.shape[1] * b.shape[1] for a, b, _ in items)
    ms, tf = timeit(lambda: ops.gemm_wgrad_grouped(items, bf), fl, "grouped")
    print("grouped dW of a block (WAVLM_GEMM_W4=%s): %.1f us = %.1f TF/s" % (os.environ.get("WAVLM_GEMM_W4", "unset"), ms * 1e3, tf))
