"""LAB: needs the lab library (`python tools/probe/build_probe.py lab`, then WAVLM_HIP_LIB=tools/probe/lib/libwavlm_hip_lab.so).
The two-workgroups-per-CU GEMM (tools/probe/gemm_h2.hip, variant 6) against the default dispatch (variant 0: gemm_pp3 / gemm_pp) on
the transformer's shapes at 24 k rows: correctness against an fp32 torch reference first (ragged M / N edges, both B layouts,
the three fast epilogues incl. residual and fused column sums), then timing.  usage (GPU box): python tools/probe/h2_ab.py [check|time|all]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from unispeech_amd import ops  # noqa: E402

dev, bf = "cuda", torch.bfloat16
what = sys.argv[1] if len(sys.argv) > 1 else "all"
torch.manual_seed(0)


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(bf)


def run(x, W, tB, epi, bias=None, aux_in=None, res=None, colsum=False, variant=6):
    """y[n, N] = epi(x[n, K] @ (W[N, K]^T or W[K, N]) + bias) (* aux) (+ res); returns (y, aux_out, colsum)"""
    n, K = x.shape
    N = W.shape[1] if tB else W.shape[0]
    y = torch.full((n, N), float("nan"), device=dev, dtype=bf)
    aux = aux_in if epi == 4 else (torch.full((n, N), float("nan"), device=dev, dtype=bf) if epi == 3 else None)
    cs = torch.zeros(N, device=dev, dtype=bf) if colsum else None
    ops.gemm_set_variant(variant)
    try:
        ops.gemm(x, W, y, n, N, K, lda=K, ldb=N if tB else K, ldc=N, transB=tB, bias=bias, epi=epi, aux=aux, ld_aux=N, res=res, ld_res=N,
                 colsum=cs, colsum_accumulate=False)
    finally:
        ops.gemm_set_variant(0)
    return y, aux, cs


def ref(x, W, tB, epi, bias=None, aux_in=None, res=None):
    a = x.float() @ (W.float() if tB else W.float().t())
    if bias is not None:
        a = a + bias.float()
    g = None
    if epi == 3:
        g = 0.5 * (1 + torch.erf(a * 0.7071067811865476)) + a * torch.exp(-0.5 * a * a) * 0.3989422804014327
        a = torch.nn.functional.gelu(a)
    if epi == 4:
        a = a * aux_in.float()
    if res is not None:
        a = a + res.float()
    return a, g


def err(a, b):
    return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-20)).item()


if what in ("check", "all"):
    bad = 0
    for (n, N, K) in [(2000, 768, 768), (1000, 200, 96), (392, 2304, 64), (777, 3072, 768), (1921, 392, 3072)]:
        for tB in (False, True):
            if tB and N % 8:
                continue
            x = rnd(n, K)
            W = rnd(K, N, scale=K ** -0.5) if tB else rnd(N, K, scale=K ** -0.5)
            b = rnd(N)
            cases = [("bias", 0, dict(bias=b)), ("gelu+g'", 3, dict(bias=b)), ("plain+res", 0, dict(res=rnd(n, N))),
                     ("*aux+res+csum", 4, dict(aux_in=rnd(n, N), res=rnd(n, N), colsum=True)), ("plain+csum", 0, dict(colsum=True))]
            for name, epi, kw in cases:
                y, aux, cs = run(x, W, tB, epi, **kw)
                kr = {k: v for k, v in kw.items() if k != "colsum"}
                r, g = ref(x, W, tB, epi, **kr)
                e = [err(y, r)]
                if epi == 3:
                    e.append(err(aux, g))
                if kw.get("colsum"):
                    e.append(err(cs, r.sum(0)))
                ok = all(v < 1.5e-2 for v in e) and bool(torch.isfinite(y.float()).all())
                bad += not ok
                print("%-5d x %-5d x %-5d %s %-14s %s %s" % (n, N, K, "NT" if tB else "NN", name, " ".join("%.2e" % v for v in e), "ok" if ok else "FAIL"))
    print("CHECK", "FAILED %d" % bad if bad else "PASSED")

if what in ("time", "all"):
    n = 32 * 749

    def timeit(fn, reps=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    shapes = []
    for (N, K, nm) in [(2304, 768, "qkv"), (768, 768, "out_proj"), (768, 3072, "fc2")]:
        x, W, b = rnd(n, K), rnd(N, K), rnd(N)
        shapes.append(("fwd %s [%d x %d x %d] + bias" % (nm, n, N, K), 2.0 * n * N * K, x, W, False, 0, dict(bias=b)))
    x, W, b = rnd(n, 768), rnd(3072, 768), rnd(3072)
    shapes.append(("fwd fc1 + bias + gelu + gelu' store", 2.0 * n * 3072 * 768, x, W, False, 3, dict(bias=b)))
    for (N, K, nm, extra) in [(768, 768, "out_proj dX", {}), (768, 2304, "qkv dX + res", dict(res=True)), (768, 3072, "fc1 dX + res", dict(res=True))]:
        dy, W = rnd(n, K), rnd(K, N)
        kw = dict(res=rnd(n, N)) if extra.get("res") else {}
        shapes.append(("%s [%d x %d x %d]" % (nm, n, N, K), 2.0 * n * N * K, dy, W, True, 0, kw))
    dy, W = rnd(n, 768), rnd(768, 3072)
    shapes.append(("fc2 dX * gelu' + colsum [%d x 3072 x 768]" % n, 2.0 * n * 3072 * 768, dy, W, True, 4, dict(aux_in=rnd(n, 3072), colsum=True)))
    print("%-52s %10s %10s   %s" % ("shape", "dflt TF/s", "h2 TF/s", "h2 / default time"))
    for name, fl, x, W, tB, epi, kw in shapes:
        # outputs allocated once: the timed region holds the GEMM launches only
        nn, K = x.shape
        N = W.shape[1] if tB else W.shape[0]
        y = torch.empty((nn, N), device=dev, dtype=bf)
        aux = kw.get("aux_in") if epi == 4 else (torch.empty((nn, N), device=dev, dtype=bf) if epi == 3 else None)
        cs = torch.zeros(N, device=dev, dtype=bf) if kw.get("colsum") else None

        def go():
            ops.gemm(x, W, y, nn, N, K, lda=K, ldb=N if tB else K, ldc=N, transB=tB, bias=kw.get("bias"), epi=epi, aux=aux, ld_aux=N,
                     res=kw.get("res"), ld_res=N, colsum=cs, colsum_accumulate=False)
        res = {0: [], 6: []}
        for v in (0, 6, 0, 6):
            ops.gemm_set_variant(v)
            res[v].append(timeit(go))
            ops.gemm_set_variant(0)
        a, b2 = min(res[0]), min(res[6])
        print("%-52s %10.1f %10.1f   %.3f   (%.1f / %.1f us)" % (name, fl / a / 1e9, fl / b2 / 1e9, b2 / a, a * 1e3, b2 * 1e3), flush=True)
