"""Kernel-level parity checks: each HIP entry point / autograd function against a plain PyTorch fp32 (fp64 where
cheap) CPU reference of the same op on seeded inputs.  Used by tests/test_kernels_gpu.py (pytest -m gpu) and by
tools/gpu_report.py (crash-isolated report).  Every check returns a list of (name, error, tolerance).

Error metric: max |a - b| / max(|b|_max, tiny) -- "relative to the tensor scale" -- unless noted.
Tolerances: fp32 mode 1e-4 (BASELINE.json north_star) except long fp32 reductions (2e-4); bf16 mode 2e-2.
"""
import math
import sys

import numpy as np
import torch
import torch.nn.functional as TF

sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
from unispeech_amd import functional as F  # noqa: E402
from unispeech_amd import ops  # noqa: E402

DEV = "cuda"
ATTN_STORE_P_DEFAULT = F.ATTN_STORE_P   # restored after the checks that switch the attention backward mode
TOL32, TOLBF = 1e-4, 2e-2


def err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    if a.shape != b.shape:
        return float("inf")
    if b.numel() == 0:
        return 0.0
    if not torch.isfinite(a).all():
        return float("inf")
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def gen(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def tol_for(dtype):
    return TOL32 if dtype == torch.float32 else TOLBF


def q(t, dtype):
    """round a CPU fp32 tensor through `dtype` so that device and reference see identical inputs"""
    return t.to(dtype).float()


# ------------------------------------------------------------------------------------------------------ GEMM
def check_gemm():
    out = []
    # the 256x128 / 8-wave tile, forced, on ragged sizes in all four layouts (+ split-K)
    ops.gemm_set_variant(2)
    try:
        dtype, tol = torch.bfloat16, TOLBF
        for (M, N, K, tA, tB) in [(700, 200, 264, 0, 0), (513, 136, 749, 0, 1), (300, 260, 200, 1, 0), (258, 130, 333, 1, 1)]:
            Kp, Mp, Np = (K + 7) // 8 * 8, (M + 7) // 8 * 8, (N + 7) // 8 * 8
            A, B = q(gen(M, K, seed=21), dtype), q(gen(N, K, seed=22), dtype)
            ref = A.double() @ B.double().t()
            if tA:
                Ad = torch.zeros(K, Mp); Ad[:, :M] = A.t(); lda = Mp
            else:
                Ad = torch.full((M, Kp), float("nan")); Ad[:, :K] = A; lda = Kp
            if tB:
                Bd = torch.zeros(K, Np); Bd[:, :N] = B.t(); ldb = Np
            else:
                Bd = torch.full((N, Kp), float("nan")); Bd[:, :K] = B; ldb = Kp
            Ad, Bd = Ad.to(dtype).to(DEV), Bd.to(dtype).to(DEV)
            for split in (1, 3):
                C = torch.full((M, Np), float("nan"), dtype=dtype, device=DEV)
                ops.gemm(Ad, Bd, C, M, N, K, lda=lda, ldb=ldb, ldc=Np, transA=tA, transB=tB, split_k=split)
                out.append((f"gemm256[{dtype}] {M}x{N}x{K} tA={tA} tB={tB} split={split}", err(C[:, :N], ref), tol))
    finally:
        ops.gemm_set_variant(0)
    for dtype in (torch.float32, torch.bfloat16):
        tol = tol_for(dtype)
        for (M, N, K, tA, tB) in [(200, 136, 72, 0, 0), (129, 64, 264, 0, 1), (260, 130, 200, 1, 0), (77, 48, 333, 1, 1),
                                  (512, 256, 512, 0, 0), (48, 640, 749, 1, 1), (100, 72, 749, 0, 0), (100, 64, 749, 0, 1)]:
            Kp = (K + 7) // 8 * 8
            Mp = (M + 7) // 8 * 8
            Np = (N + 7) // 8 * 8
            A = q(gen(M, K, seed=1), dtype)
            B = q(gen(N, K, seed=2), dtype)
            ref = A.double() @ B.double().t()
            # storage: K-contiguous [rows, Kp] or K-strided [K, rows_p]
            if tA:
                Ad = torch.zeros(K, Mp); Ad[:, :M] = A.t(); lda = Mp
            else:
                Ad = torch.zeros(M, Kp); Ad[:, :K] = A; lda = Kp
            if tB:
                Bd = torch.zeros(K, Np); Bd[:, :N] = B.t(); ldb = Np
            else:
                Bd = torch.zeros(N, Kp); Bd[:, :K] = B; ldb = Kp
            # poison the padding with NaN where it must be ignored (K tail of K-contiguous operands)
            if not tA and Kp > K:
                Ad[:, K:] = float("nan")
            if not tB and Kp > K:
                Bd[:, K:] = float("nan")
            Ad = Ad.to(dtype).to(DEV); Bd = Bd.to(dtype).to(DEV)
            C = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
            ops.gemm(Ad, Bd, C, M, N, K, lda=lda, ldb=ldb, ldc=N, transA=tA, transB=tB)
            out.append((f"gemm[{dtype}] {M}x{N}x{K} tA={tA} tB={tB}", err(C, ref), tol))
            for split in (3,):
                C2 = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
                ops.gemm(Ad, Bd, C2, M, N, K, lda=lda, ldb=ldb, ldc=N, transA=tA, transB=tB, split_k=split)
                out.append((f"gemm[{dtype}] {M}x{N}x{K} tA={tA} tB={tB} split={split}", err(C2, ref), tol))
        # batched + K-batch + epilogues
        Bo, Bi, KB, M, N, K = 2, 3, 2, 70, 40, 64
        A = q(gen(Bo, Bi, KB, M, K, seed=3), dtype)
        B = q(gen(Bo, Bi, KB, N, K, seed=4), dtype)
        bias = q(gen(Bi, N, seed=5), dtype)
        res = q(gen(Bo, Bi, M, N, seed=6), dtype)
        pre = torch.einsum("oikmc,oiknc->oimn", A.double(), B.double()) * 0.5 + bias.double()[None, :, None, :]
        ref = TF.gelu(pre) + res.double()
        Ad, Bd = A.to(dtype).to(DEV), B.to(dtype).to(DEV)
        C = torch.empty(Bo, Bi, M, N, dtype=dtype, device=DEV)
        aux = torch.empty(Bo, Bi, M, N, dtype=dtype, device=DEV)
        ops.gemm(Ad, Bd, C, M, N, K, lda=K, ldb=K, ldc=N, KB=KB, sA_kb=M * K, sB_kb=N * K, batch=(Bo, Bi),
                 sA=(Bi * KB * M * K, KB * M * K), sB=(Bi * KB * N * K, KB * N * K), sC=(Bi * M * N, M * N), alpha=0.5,
                 bias=bias.to(dtype).to(DEV), sBias=(0, N), epi=1, aux=aux, ld_aux=N, sAux=(Bi * M * N, M * N),
                 res=res.to(dtype).to(DEV), ld_res=N, sRes=(Bi * M * N, M * N))
        out.append((f"gemm[{dtype}] batched+KB+bias+gelu+res", err(C, ref), tol))
        out.append((f"gemm[{dtype}] aux(pre-activation)", err(aux, pre), tol))
        # epi 2: multiply by gelu'(aux)
        u = q(gen(M, N, seed=7), dtype)
        A2, B2 = q(gen(M, K, seed=8), dtype), q(gen(N, K, seed=9), dtype)
        ud = u.double().requires_grad_(True)
        gp = torch.autograd.grad(TF.gelu(ud).sum(), ud)[0]
        ref2 = (A2.double() @ B2.double().t()) * gp
        C = torch.empty(M, N, dtype=dtype, device=DEV)
        ops.gemm(A2.to(dtype).to(DEV), B2.to(dtype).to(DEV), C, M, N, K, lda=K, ldb=K, ldc=N, epi=2,
                 aux=u.to(dtype).to(DEV), ld_aux=N)
        out.append((f"gemm[{dtype}] epi=gelu'", err(C, ref2), tol))
        # overlapping rows == strided conv1d over a channel-last activation
        Bb, Tin, Cin, Cout, k, s = 2, 41, 16, 24, 3, 2
        x = q(gen(Bb, Tin, Cin, seed=10), dtype)
        w = q(gen(Cout, Cin, k, seed=11, scale=0.2), dtype)
        refc = TF.conv1d(x.double().transpose(1, 2), w.double(), stride=s).transpose(1, 2)
        Tout = refc.shape[1]
        Wf = w.permute(0, 2, 1).reshape(Cout, k * Cin).contiguous()
        y = torch.empty(Bb, Tout, Cout, dtype=dtype, device=DEV)
        ops.gemm(x.to(dtype).to(DEV), Wf.to(dtype).to(DEV), y, Tout, Cout, k * Cin, lda=s * Cin, ldb=k * Cin, ldc=Cout,
                 batch=(Bb, 1), sA=(Tin * Cin, 0), sC=(Tout * Cout, 0))
        out.append((f"gemm[{dtype}] overlapping-row conv1d k3 s2", err(y, refc), tol))
    return out


def check_gemm_pp(variant=3, name="gemm_pp"):
    """256x256 (variant 3) / 192x384 (variant 4) ping-pong kernels: all four operand layouts, ragged M/N edges, K tails
    (zero-page path), K batches with a tail in every batch, split-K, and the fused epilogues."""
    out = []
    dtype, tol = torch.bfloat16, TOLBF
    ops.gemm_set_variant(variant)
    try:
        cases = [  # M, N, K, tA, tB, KB, split
            (600, 520, 256, 0, 0, 1, 1), (600, 520, 200, 0, 0, 1, 1), (1000, 296, 1000, 0, 1, 1, 1),
            (512, 256, 777, 1, 1, 1, 1), (264, 392, 333, 1, 0, 1, 1), (520, 264, 150, 1, 1, 3, 1),
            (768, 512, 1400, 1, 1, 2, 3), (300, 260, 128, 0, 0, 1, 2), (2000, 768, 64, 0, 0, 1, 1),
            (256, 128, 64, 0, 1, 1, 1), (776, 1032, 520, 0, 0, 2, 1),
        ]
        for (M, N, K, tA, tB, KB, split) in cases:
            A = q(gen(KB, M, K, seed=31), dtype)
            B = q(gen(KB, N, K, seed=32), dtype)
            ref = torch.einsum("kmc,knc->mn", A.double(), B.double())
            Kp, Mp, Np = (K + 7) // 8 * 8 + 8, (M + 7) // 8 * 8, (N + 7) // 8 * 8
            if tA:
                Ad = torch.full((KB, K, Mp), float("nan")); Ad[:, :, :M] = A.transpose(1, 2); lda, sa = Mp, K * Mp
            else:
                Ad = torch.full((KB, M, Kp), float("nan")); Ad[:, :, :K] = A; lda, sa = Kp, M * Kp
            if tB:
                Bd = torch.full((KB, K, Np), float("nan")); Bd[:, :, :N] = B.transpose(1, 2); ldb, sb = Np, K * Np
            else:
                Bd = torch.full((KB, N, Kp), float("nan")); Bd[:, :, :K] = B; ldb, sb = Kp, N * Kp
            Ad, Bd = Ad.to(dtype).to(DEV), Bd.to(dtype).to(DEV)
            C = torch.full((M, Np), float("nan"), dtype=dtype, device=DEV)
            ops.gemm(Ad, Bd, C, M, N, K, lda=lda, ldb=ldb, ldc=Np, transA=tA, transB=tB, KB=KB, sA_kb=sa, sB_kb=sb,
                     split_k=split)
            out.append((f"{name} {M}x{N}x{K} tA={tA} tB={tB} KB={KB} split={split}", err(C[:, :N], ref), tol))
        # batched + bias + gelu + aux + residual through the ping-pong kernel
        Bo, Bi, M, N, K = 2, 2, 300, 264, 192
        A = q(gen(Bo, Bi, M, K, seed=33), dtype)
        B = q(gen(Bo, Bi, N, K, seed=34), dtype)
        bias = q(gen(Bi, N, seed=35), dtype)
        res = q(gen(Bo, Bi, M, N, seed=36), dtype)
        pre = torch.einsum("oimc,oinc->oimn", A.double(), B.double()) * 0.5 + bias.double()[None, :, None, :]
        ref = TF.gelu(pre) + res.double()
        C = torch.empty(Bo, Bi, M, N, dtype=dtype, device=DEV)
        aux = torch.empty(Bo, Bi, M, N, dtype=dtype, device=DEV)
        ops.gemm(A.to(dtype).to(DEV), B.to(dtype).to(DEV), C, M, N, K, lda=K, ldb=K, ldc=N, batch=(Bo, Bi),
                 sA=(Bi * M * K, M * K), sB=(Bi * N * K, N * K), sC=(Bi * M * N, M * N), alpha=0.5,
                 bias=bias.to(dtype).to(DEV), sBias=(0, N), epi=1, aux=aux, ld_aux=N, sAux=(Bi * M * N, M * N),
                 res=res.to(dtype).to(DEV), ld_res=N, sRes=(Bi * M * N, M * N))
        out.append((f"{name} batched+bias+gelu+res", err(C, ref), tol))
        out.append((f"{name} aux(pre-activation)", err(aux, pre), tol))
        # overlapping rows (strided conv) through the ping-pong kernel
        Bb, Tin, Cin, Cout, k, s = 2, 701, 64, 256, 3, 2
        x = q(gen(Bb, Tin, Cin, seed=37), dtype)
        w = q(gen(Cout, Cin, k, seed=38, scale=0.2), dtype)
        refc = TF.conv1d(x.double().transpose(1, 2), w.double(), stride=s).transpose(1, 2)
        Tout = refc.shape[1]
        Wf = w.permute(0, 2, 1).reshape(Cout, k * Cin).contiguous()
        y = torch.empty(Bb, Tout, Cout, dtype=dtype, device=DEV)
        ops.gemm(x.to(dtype).to(DEV), Wf.to(dtype).to(DEV), y, Tout, Cout, k * Cin, lda=s * Cin, ldb=k * Cin, ldc=Cout,
                 batch=(Bb, 1), sA=(Tin * Cin, 0), sC=(Tout * Cout, 0))
        out.append((f"{name} overlapping-row conv1d k3 s2", err(y, refc), tol))
    finally:
        ops.gemm_set_variant(0)
    return out


def check_gemm_pp3():
    return check_gemm_pp(4, "gemm_pp3")


def check_gemm_w4():
    """the four-wave 256 x 256 kernel (csrc/gemm_w4.hip, variant 5: 128 x 128 accumulators per wave, fragment reads
    software-pipelined between the MFMAs) on the same cases as the eight-wave kernel it shares its LDS image with, plus the
    fused column sums / GELU' epilogue and the grouped launch"""
    out = check_gemm_pp(5, "gemm_w4")
    ops.gemm_set_variant(5)
    try:
        out += [("[w4] " + n, e, t) for n, e, t in check_gemm_colsum()]
    finally:
        ops.gemm_set_variant(0)
    return out


def check_gemm_race():
    """the same ping-pong launch repeated gives bit-identical output (an LDS hazard between the asynchronous operand DMA
    and the fragment reads would show as run-to-run differences); step-sized shapes, both tile shapes, split-K"""
    out = []
    n = 32 * 749
    for name, M, N, K, tA, tB, split in [("192x384 NN", n, 2304, 768, 0, 0, 1), ("256x256 NN", n, 2048, 768, 0, 0, 1),
                                          ("256x256 TT split 7", 3072, 768, n, 1, 1, 7), ("192x384 NT", n, 768, 3072, 0, 1, 1)]:
        A = q(gen(*((K, M) if tA else (M, K)), seed=91), torch.bfloat16).to(torch.bfloat16).to(DEV)
        B = q(gen(*((K, N) if tB else (N, K)), seed=92), torch.bfloat16).to(torch.bfloat16).to(DEV)
        C0, diff = None, 0
        for _ in range(8):
            C = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
            ops.gemm(A, B, C, M, N, K, lda=M if tA else K, ldb=N if tB else K, ldc=N, transA=tA, transB=tB, split_k=split)
            if C0 is None:
                C0 = C
            else:
                diff += int((C.view(torch.int16) != C0.view(torch.int16)).sum().item())
        out.append((f"gemm_race {name} {M}x{N}x{K}: differing elements over 8 launches", float(diff), 0.0))
    return out


def check_gemm_grouped():
    """wavlm_gemm_grouped: weight gradients of several linears over the same rows in one grouped split-K launch,
    accumulated into existing (non-zero) outputs; ragged rows (K tail), ragged M/N tiles, 2-4 members; plus a group the
    256-wide kernel does not take (sequential fall-back inside the library)."""
    out = []
    dtype, tol = torch.bfloat16, TOLBF
    # (n = 5003 with the Base shapes, n = 9000 with ragged tiles: balanced launches -- main workgroups + tail workgroups, see
    #  gemm_common.hpp: gemm_sk_plan; the others: the one-round split)
    for n, shapes in [(1000, [(520, 264), (256, 768)]), (2500, [(2304, 768), (768, 768), (3072, 768), (768, 3072)]),
                      (5003, [(2304, 768), (768, 768), (3072, 768), (768, 3072)]), (9000, [(2000, 1032), (1288, 1032)]),
                      (777, [(264, 392), (512, 256), (304, 520)]), (300, [(64, 48), (96, 64)])]:
        items, refs = [], []
        for k, (N, K) in enumerate(shapes):
            dy, x = q(gen(n, N, seed=50 + k), dtype), q(gen(n, K, seed=60 + k), dtype)
            o0 = q(gen(N, K, seed=70 + k), dtype)
            refs.append(o0.double() + dy.double().t() @ x.double())
            items.append((dy.to(dtype).to(DEV), x.to(dtype).to(DEV), o0.to(dtype).to(DEV).clone()))
        ops.gemm_wgrad_grouped(items, dtype)
        for k, ((N, K), it, ref) in enumerate(zip(shapes, items, refs)):
            out.append((f"gemm_grouped n={n} member {k} [{N}x{K}]", err(it[2], ref), tol))
    return out


# --------------------------------------------------------------------------------------------------- row ops
def check_layernorm():
    out = []
    for dtype in (torch.float32, torch.bfloat16):
        tol = tol_for(dtype)
        for D in (64, 512, 768, 1024):
            rows = 37
            x, r = q(gen(rows, D, seed=1), dtype), q(gen(rows, D, seed=2), dtype)
            g, b = q(1 + 0.1 * gen(D, seed=3), dtype), q(0.1 * gen(D, seed=4), dtype)
            dy = q(gen(rows, D, seed=5), dtype)
            for act in (0, 1):
                xr, rr, gr, br = [t.clone().requires_grad_(True) for t in (x, r, g, b)]
                s = xr + rr
                if dtype == torch.bfloat16:
                    s = s + (s.detach().to(dtype).float() - s.detach())  # straight-through bf16 rounding of the sum
                z = TF.layer_norm(s, (D,), gr, br, 1e-5)
                yr = TF.gelu(z) if act else z
                (yr * dy).sum().backward()
                xd, rd, gd, bd = [t.to(dtype).to(DEV).requires_grad_(True) for t in (x, r, g, b)]
                y, s_out = F.LayerNormFn.apply(xd, rd, gd, bd, 1e-5, act, 0.0, 0, 0.0, 0, 1.0)
                y.backward(dy.to(dtype).to(DEV))
                tag = f"layernorm[{dtype}] D={D} act={act}"
                out.append((tag + " y", err(y, yr), tol))
                out.append((tag + " s", err(s_out, s), tol))
                out.append((tag + " dx", err(xd.grad, xr.grad), tol))
                out.append((tag + " dr", err(rd.grad, rr.grad), tol))
                out.append((tag + " dgamma", err(gd.grad, gr.grad), tol * 2))
                out.append((tag + " dbeta", err(bd.grad, br.grad), tol * 2))
        # dropout consistency: y(p) on kept elements == y(0)/(1-p); grads use the same mask
        D, rows, p = 768, 64, 0.25
        x = gen(rows, D, seed=1).to(dtype).to(DEV).requires_grad_(True)
        r = gen(rows, D, seed=2).to(dtype).to(DEV).requires_grad_(True)
        g = torch.ones(D, dtype=dtype, device=DEV, requires_grad=True)
        b = torch.zeros(D, dtype=dtype, device=DEV, requires_grad=True)
        y0, _ = F.LayerNormFn.apply(x, None, g, b, 1e-5, 0, 0.0, 0, 0.0, 0, 1.0)
        yp, _ = F.LayerNormFn.apply(x, None, g, b, 1e-5, 0, 0.0, 0, p, 777, 1.0)
        keep = (yp != 0)
        frac = keep.float().mean().item()
        out.append((f"layernorm[{dtype}] out-dropout keep fraction", abs(frac - (1 - p)), 0.02))
        out.append((f"layernorm[{dtype}] out-dropout values", err(yp[keep], (y0 / (1 - p))[keep]), tol))
        yp.backward(torch.ones_like(yp))
        # in-dropout: s - x must be r/(1-p) on kept elements, and dr must carry the same mask
        y2, s2 = F.LayerNormFn.apply(x.detach(), r, g, b, 1e-5, 0, p, 999, 0.0, 0, 1.0)
        dlt = (s2.float() - x.detach().float())
        keep2 = dlt.abs() > 1e-6
        out.append((f"layernorm[{dtype}] in-dropout keep fraction", abs(keep2.float().mean().item() - (1 - p)), 0.02))
        y2.backward(gen(rows, D, seed=9).to(dtype).to(DEV))
        out.append((f"layernorm[{dtype}] in-dropout grad mask", float(((r.grad != 0) != keep2).float().mean().item()), 0.01))
        # by-product of the backward pass: column sums of dr (the bias gradient of the linear that produced r), fresh
        # and accumulated into existing tensors, against an explicit column sum of the dr the same call returned
        dyb = gen(rows, D, seed=19).to(dtype).to(DEV)
        _y, sv, mean, rstd = ops.layernorm_fwd(x.detach(), r.detach(), g.detach(), b.detach(), 1e-5, act=0, p_in=p, seed_in=999,
                                               p_out=0.0, seed_out=0, save=True)
        for p_in in (p, 0.0):
            dx_, dr_, _, _, cs = ops.layernorm_bwd(dyb, sv, mean, rstd, g.detach(), b.detach(), p_in=p_in, seed_in=999,
                                                   need_dr=p_in > 0, dr_colsum=True)
            want = (dr_ if dr_ is not None else dx_).double().sum(0)
            out.append((f"layernorm[{dtype}] dr colsum by-product (p_in={p_in})", err(cs, want), tol * 4))
        dg0 = gen(D, seed=21).to(dtype).to(DEV)
        dg, db_, dc = dg0.clone(), dg0.clone(), dg0.clone()
        dx_, dr_, _, _, _ = ops.layernorm_bwd(dyb, sv, mean, rstd, g.detach(), b.detach(), p_in=p, seed_in=999, need_dr=True,
                                              dgamma=dg, dbeta=db_, dr_colsum=dc)
        out.append((f"layernorm[{dtype}] dr colsum accumulated", err(dc, dg0.double() + dr_.double().sum(0)), tol * 4))
        # residual-stream gradient added inside the backward kernel (pre-LN blocks): dx gains dx_add, dr / colsum do not
        extra = gen(rows, D, seed=23).to(dtype).to(DEV)
        dx0, dr0, _, _, cs0 = ops.layernorm_bwd(dyb, sv, mean, rstd, g.detach(), b.detach(), p_in=p, seed_in=999,
                                                need_dr=True, dr_colsum=True)
        dx1, dr1, _, _, cs1 = ops.layernorm_bwd(dyb, sv, mean, rstd, g.detach(), b.detach(), p_in=p, seed_in=999,
                                                need_dr=True, dr_colsum=True, dx_add=extra)
        out.append((f"layernorm[{dtype}] dx_add", err(dx1, dx0.double() + extra.double()), tol))
        # (same values; the two template instantiations may contract the fp32 expression differently: last-bit slack)
        out.append((f"layernorm[{dtype}] dx_add leaves dr", err(dr1, dr0), 1e-6 if dtype == torch.float32 else 0.0))
        out.append((f"layernorm[{dtype}] dx_add leaves colsum", err(cs1, cs0), tol))
        # the same through autograd: y = LN(x) + x with x handed through as an alias
        xa = gen(rows, D, seed=31).to(dtype).to(DEV).requires_grad_(True)
        ya, _s, xal = F.LayerNormFn.apply(xa, None, g.detach(), b.detach(), 1e-5, 0, 0.0, 0, 0.0, 0, 1.0, None, True)
        (ya.float() * dyb.float() + xal.float() * extra.float()).sum().backward()
        xb = xa.detach().clone().requires_grad_(True)
        yb, _s = F.LayerNormFn.apply(xb, None, g.detach(), b.detach(), 1e-5, 0, 0.0, 0, 0.0, 0, 1.0)
        (yb.float() * dyb.float() + xb.float() * extra.float()).sum().backward()
        out.append((f"layernorm[{dtype}] pass_x autograd", err(xa.grad, xb.grad), tol))
    return out


def check_rowops():
    out = []
    for dtype in (torch.float32, torch.bfloat16):
        tol = tol_for(dtype)
        rows, D = 301, 768
        x = q(gen(rows, D, seed=1), dtype)
        xd = x.to(dtype).to(DEV)
        inc = (torch.arange(rows) % 3 == 0)
        exc = (torch.arange(rows) % 5 == 0)
        cs = ops.colsum(xd, torch.float32)
        out.append((f"colsum[{dtype}]", err(cs, x.double().sum(0)), tol))
        cs2 = ops.colsum(xd, torch.float32, include=inc.to(torch.uint8).to(DEV), exclude=exc.to(torch.uint8).to(DEV))
        out.append((f"colsum[{dtype}] masked", err(cs2, x.double()[inc & ~exc].sum(0)), tol))
        emb = q(gen(D, seed=2), dtype)
        y = ops.select_rows(xd, inc.to(torch.uint8).to(DEV), emb.to(dtype).to(DEV), exc.to(torch.uint8).to(DEV))
        ref = x.clone(); ref[inc] = emb; ref[exc] = 0
        out.append((f"select_rows[{dtype}]", err(y, ref), 1e-6))
        idx = torch.tensor([5, -1, 0, 300, 17], dtype=torch.int32)
        gth = ops.gather_rows(xd, idx.to(DEV), 5)
        ref = torch.stack([x[5], torch.zeros(D), x[0], x[300], x[17]])
        out.append((f"gather_rows[{dtype}]", err(gth, ref), 1e-6))
        yv = q(gen(rows, D, seed=3), dtype)
        yd = yv.to(dtype).to(DEV)
        ops.axpby_(yd, xd, 0.5, 2.0)
        out.append((f"axpby[{dtype}]", err(yd, 0.5 * x + 2.0 * yv), tol))
        d1 = ops.dropout(xd, 0.1, 42)
        d2 = ops.dropout(xd, 0.1, 42)
        keep = d1 != 0
        out.append((f"dropout[{dtype}] deterministic", float((d1 != d2).float().mean().item()), 0.0))
        out.append((f"dropout[{dtype}] keep fraction", abs(keep.float().mean().item() - 0.9), 0.01))
        out.append((f"dropout[{dtype}] scale", err(d1[keep], (xd.float() / 0.9)[keep]), tol))
        ss = ops.sumsq(xd, 0.25)
        out.append((f"sumsq[{dtype}]", err(ss, (x.double() ** 2).sum().reshape(1) * 0.25), tol))
        sc = torch.tensor([3.0], device=DEV)
        z = xd.clone(); ops.scale_dev_(z, sc, 0.5)
        out.append((f"scale_dev[{dtype}]", err(z, x * 1.5), tol))
    lr = gen(1000, seed=4).to(DEV)
    out.append(("sum_f32", err(ops.sum_f32(lr), lr.double().sum().reshape(1)), 1e-5))
    return out


# ----------------------------------------------------------------------------------------------------- conv0
def check_conv0():
    out = []
    for dtype in (torch.float32, torch.bfloat16):
        tol = tol_for(dtype)
        for (B, T, C) in [(2, 16000, 512), (3, 4005, 32)]:
            wav = q(gen(B, T, seed=1), dtype)
            W = q(gen(C, 1, 10, seed=2, scale=0.4), dtype)
            g, b = q(1 + 0.1 * gen(C, seed=3), dtype), q(0.1 * gen(C, seed=4), dtype)
            wr, Wr, gr, br = wav.clone(), W.clone().requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
            y = TF.gelu(TF.group_norm(TF.conv1d(wr.unsqueeze(1), Wr, stride=5), C, gr, br, 1e-5)).transpose(1, 2)
            dy = q(gen(*y.shape, seed=5), dtype)
            (y * dy).sum().backward()
            Wd, gd, bd = [t.to(dtype).to(DEV).requires_grad_(True) for t in (W, g, b)]
            yd = F.Conv0Fn.apply(wav.to(dtype).to(DEV), Wd, gd, bd, 5, 1e-5, dtype)
            yd.backward(dy.to(dtype).to(DEV))
            tag = f"conv0[{dtype}] B={B} T={T} C={C}"
            out.append((tag + " y", err(yd, y), tol))
            out.append((tag + " dW", err(Wd.grad, Wr.grad), tol * 3))
            out.append((tag + " dgamma", err(gd.grad, gr.grad), tol * 3))
            out.append((tag + " dbeta", err(bd.grad, br.grad), tol * 3))
            # extractor_mode "layer_norm": conv0 -> LayerNorm over channels -> GELU
            Wr2, gr2, br2 = W.clone().requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
            y2 = TF.gelu(TF.layer_norm(TF.conv1d(wav.unsqueeze(1), Wr2, stride=5).transpose(1, 2), (C,), gr2, br2, 1e-5))
            (y2 * dy).sum().backward()
            Wd, gd, bd = [t.to(dtype).to(DEV).requires_grad_(True) for t in (W, g, b)]
            yd = F.Conv0LNFn.apply(wav.to(dtype).to(DEV), Wd, gd, bd, 5, 1e-5, dtype)
            yd.backward(dy.to(dtype).to(DEV))
            tag = f"conv0+LN[{dtype}] B={B} T={T} C={C}"
            out.append((tag + " y", err(yd, y2), tol))
            out.append((tag + " dW", err(Wd.grad, Wr2.grad), tol * 3))
            out.append((tag + " dgamma", err(gd.grad, gr2.grad), tol * 3))
            out.append((tag + " dbeta", err(bd.grad, br2.grad), tol * 3))
    # other strides through the GroupNorm-mode kernels (the waveform segment a workgroup stages grows with the stride)
    for st, T in ((8, 33000), (3, 9000)):
        dtype, C, B = torch.bfloat16, 512, 2
        wav = q(gen(B, T, seed=21), dtype)
        W = q(gen(C, 1, 10, seed=22, scale=0.4), dtype)
        g, b = q(1 + 0.1 * gen(C, seed=23), dtype), q(0.1 * gen(C, seed=24), dtype)
        ts = [t.double().clone().requires_grad_(True) for t in (W, g, b)]
        y = TF.gelu(TF.group_norm(TF.conv1d(wav.double().unsqueeze(1), ts[0], stride=st), C, ts[1], ts[2], 1e-5)).transpose(1, 2)
        dy = q(gen(*y.shape, seed=25), dtype)
        gr = torch.autograd.grad(y, ts, dy.double())
        Wd, gd, bd = [t.to(dtype).to(DEV).requires_grad_(True) for t in (W, g, b)]
        yd = F.Conv0Fn.apply(wav.to(dtype).to(DEV), Wd, gd, bd, st, 1e-5, dtype)
        yd.backward(dy.to(dtype).to(DEV))
        tag = f"conv0[{dtype}] stride={st} T={T}"
        out.append((tag + " y", err(yd, y), TOLBF))
        for nm, a, r in zip(("dW", "dgamma", "dbeta"), (Wd.grad, gd.grad, bd.grad), gr):
            out.append((tag + " " + nm, err(a, r), TOLBF * 3))
    return out


def check_conv0_ln():
    """extractor_mode 'layer_norm' block 0 at the width of the real models (C = 512: with bf16 operands the backward runs on the
    matrix cores, conv0_bwd_mfma.hip) with conv bias, ragged chunk ends (512 frames per workgroup, 32 per tile), fewer frames
    than one tile, a bias that moves the frame mean away from zero, and feature_grad_mult-style scaling."""
    out = []
    C = 512
    for dtype in (torch.bfloat16, torch.float32):
        tol = tol_for(dtype)
        for (B, T, boff, gs, st) in [(2, 16000, 0.0, 1.0, 5), (1, 400, 0.0, 1.0, 5), (3, 2565, 0.5, 0.1, 5), (1, 5175, -1.0, 1.0, 5),
                                     (2, 95, 0.0, 1.0, 5), (2, 4106, 0.0, 1.0, 8), (1, 3001, 0.2, 1.0, 3), (2, 10, 0.0, 1.0, 5)]:
            if dtype == torch.float32 and T > 3000:
                continue
            wav = q(gen(B, T, seed=1), dtype)
            W = q(gen(C, 1, 10, seed=2, scale=0.4), dtype)
            g, b = q(1 + 0.1 * gen(C, seed=3), dtype), q(0.1 * gen(C, seed=4), dtype)
            cb = q(boff + 0.2 * gen(C, seed=6), dtype)
            ts = [t.double().clone().requires_grad_(True) for t in (W, g, b, cb)]
            y = TF.gelu(TF.layer_norm(TF.conv1d(wav.double().unsqueeze(1), ts[0], ts[3], stride=st).transpose(1, 2), (C,), ts[1], ts[2], 1e-5))
            dy = q(gen(*y.shape, seed=5), dtype)
            gr = torch.autograd.grad(y, ts, dy.double() * gs)
            Wd, gd, bd, cd = [t.to(dtype).to(DEV) for t in (W, g, b, cb)]
            wd = wav.to(dtype).to(DEV)
            yd = ops.conv0_ln_gelu_fwd(wd, Wd, gd, bd, st, 1e-5, dtype, bias=cd)
            dW, dg, db, dcb = ops.conv0_ln_gelu_bwd(wd, Wd, gd, bd, dy.to(dtype).to(DEV), st, 1e-5, gscale=gs, bias=cd)
            tag = f"conv0+LN+bias[{dtype}] B={B} T={T} off={boff} stride={st}"
            out.append((tag + " y", err(yd, y), tol))
            out.append((tag + " dW", err(dW, gr[0]), tol))
            out.append((tag + " dgamma", err(dg, gr[1]), tol))
            out.append((tag + " dbeta", err(db, gr[2]), tol))
            out.append((tag + " dbias", err(dcb, gr[3]), tol))
    # batch linearity at a size where a workgroup of the matrix-core form walks two chunks (64 rows x 8 chunks = 512 chunks on
    # 256 CUs) and one where it walks one (32 x 8): the gradients of the whole batch are the sums over its two halves
    B, T = 64, 4000 * 5 + 5
    wd = gen(B, T, seed=11).to(torch.bfloat16).to(DEV)
    Wd = gen(C, 1, 10, seed=12, scale=0.4).to(torch.bfloat16).to(DEV)
    gd, bd, cd = [(o + 0.1 * gen(C, seed=13 + i)).to(torch.bfloat16).to(DEV) for i, o in enumerate((1.0, 0.0, 0.3))]
    dy = torch.randn(B, 4000, C, generator=torch.Generator(device=DEV).manual_seed(7), device=DEV).to(torch.bfloat16)
    full = ops.conv0_ln_gelu_bwd(wd, Wd, gd, bd, dy, 5, 1e-5, bias=cd)
    ha = ops.conv0_ln_gelu_bwd(wd[:32].contiguous(), Wd, gd, bd, dy[:32].contiguous(), 5, 1e-5, bias=cd)
    hb = ops.conv0_ln_gelu_bwd(wd[32:].contiguous(), Wd, gd, bd, dy[32:].contiguous(), 5, 1e-5, bias=cd)
    for nm, f, a, b2 in zip(("dW", "dgamma", "dbeta", "dbias"), full, ha, hb):
        out.append((f"conv0+LN batch linearity {nm}", err(f, a.float() + b2.float()), 1e-2))
    return out


def check_conv_ln_block():
    """A conv block of the layer_norm extractor mode (conv -> LayerNorm over channels -> GELU, WavLM/WavLM.py:403-418) at the real
    width: the LayerNorm's backward writes its input gradient straight into the zero-padded layout the conv's backward reads
    (LayerNormFn grad_pad / wavlm_layernorm_bwd_seg).  Against the fp64 reference, and bit-identical to the path with the padded
    copy (ops.LN_SEG_OK = False)."""
    out = []
    C = 512
    dtype = torch.bfloat16
    for (B, T_in, k, s_) in [(3, 101, 3, 2), (2, 64, 2, 2), (2, 37, 3, 2), (1, 200, 2, 2)]:
        x = q(gen(B, T_in, C, seed=1), dtype)
        W = q(gen(C, C, k, seed=2, scale=1.0 / math.sqrt(C * k)), dtype)
        cb = q(0.1 * gen(C, seed=3), dtype)
        g, b = q(1 + 0.1 * gen(C, seed=4), dtype), q(0.1 * gen(C, seed=5), dtype)
        ts = [t.double().clone().requires_grad_(True) for t in (x, W, cb, g, b)]
        yr = TF.gelu(TF.layer_norm(TF.conv1d(ts[0].transpose(1, 2), ts[1], ts[2], stride=s_).transpose(1, 2), (C,), ts[3], ts[4], 1e-5))
        dy = q(gen(*yr.shape, seed=6), dtype)
        gr = torch.autograd.grad(yr, ts, dy.double())
        res = {}
        for seg in (True, False):
            saved = ops.LN_SEG_OK
            ops.LN_SEG_OK = seg
            try:
                td = [t.to(dtype).to(DEV).requires_grad_(True) for t in (x, W, cb, g, b)]
                v = F.ConvStackFn.apply(td[0], ((k, s_),), False, td[1], td[2])
                y, _ = F.layer_norm(v, td[3], td[4], 1e-5, act=1, grad_pad=F.conv_grad_pad(T_in, k, s_))
                res[seg] = (y,) + torch.autograd.grad(y, td, dy.to(dtype).to(DEV))
            finally:
                ops.LN_SEG_OK = saved
        tag = f"conv+LN block B={B} T={T_in} k={k} s={s_}"
        out.append((tag + " y", err(res[True][0], yr), TOLBF))
        for nm, a, r in zip(("dx", "dW", "dbias", "dgamma", "dbeta"), res[True][1:], gr):
            out.append((tag + " " + nm, err(a, r), TOLBF))
        for nm, a, c in zip(("y", "dx", "dW", "dbias", "dgamma", "dbeta"), res[True], res[False]):
            out.append((tag + " " + nm + " == the padded-copy path", 0.0 if torch.equal(a, c) else 1.0, 0.0))
    return out


def check_convstack():
    out = []
    for dtype in (torch.float32, torch.bfloat16):
        tol = tol_for(dtype)
        B, T0, C = 2, 403, 32
        specs = ((3, 2), (3, 2), (2, 2), (2, 2))
        x = q(gen(B, T0, C, seed=1), dtype)
        Ws = [q(gen(C, C, k, seed=10 + i, scale=1.0 / math.sqrt(C * k)), dtype) for i, (k, s) in enumerate(specs)]
        xr = x.clone().requires_grad_(True)
        Wr = [w.clone().requires_grad_(True) for w in Ws]
        h = xr.transpose(1, 2)
        for (k, s), w in zip(specs, Wr):
            h = TF.gelu(TF.conv1d(h, w, stride=s))
        yr = h.transpose(1, 2)
        dy = q(gen(*yr.shape, seed=5), dtype)
        (yr * dy).sum().backward()
        xd = x.to(dtype).to(DEV).requires_grad_(True)
        Wd = [w.to(dtype).to(DEV).requires_grad_(True) for w in Ws]
        yd = F.ConvStackFn.apply(xd, specs, True, *Wd)
        yd.backward(dy.to(dtype).to(DEV))
        tag = f"convstack[{dtype}]"
        out.append((tag + " y", err(yd, yr), tol))
        out.append((tag + " dx", err(xd.grad, xr.grad), tol * 2))
        for i in range(len(specs)):
            out.append((tag + f" dW{i}", err(Wd[i].grad, Wr[i].grad), tol * 2))
    return out


# -------------------------------------------------------------------------------------------------- attention
def _ref_attention(qkv, gate, tab, kpm, H, scale):
    B, T, D3 = qkv.shape
    D = D3 // 3
    hd = D // H
    qh = qkv[..., :D].view(B, T, H, hd).permute(0, 2, 1, 3)
    kh = qkv[..., D:2 * D].view(B, T, H, hd).permute(0, 2, 1, 3)
    vh = qkv[..., 2 * D:].view(B, T, H, hd).permute(0, 2, 1, 3)
    s = (qh @ kh.transpose(-1, -2)) * scale
    if tab is not None:
        i = torch.arange(T)[:, None]
        j = torch.arange(T)[None, :]
        rel = tab[:, (j - i) + T - 1]  # [H, T, T]
        s = s + gate.unsqueeze(-1) * rel.unsqueeze(0)
    if kpm is not None:
        s = s.masked_fill(kpm.bool()[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    return (p @ vh).permute(0, 2, 1, 3).reshape(B, T, D)


def check_attention():
    out = []
    for dtype in (torch.float32, torch.bfloat16):
        tol = tol_for(dtype)
        for (B, T, H, hd, use_pad) in [(2, 49, 2, 32, False), (2, 131, 4, 64, True)]:
            D = H * hd
            qkv = q(gen(B, T, 3 * D, seed=1), dtype)
            gate = 1 + 0.5 * gen(B, H, T, seed=2)
            tab = 0.5 * gen(H, 2 * T - 1, seed=3)
            kpm = None
            if use_pad:
                kpm = torch.zeros(B, T, dtype=torch.uint8)
                kpm[1, T - 20:] = 1
            dO = q(gen(B, T, D, seed=4), dtype)
            qr, gr, tr = qkv.clone().double().requires_grad_(True), gate.clone().double().requires_grad_(True), tab.clone().double().requires_grad_(True)
            Or = _ref_attention(qr, gr, tr, kpm, H, hd ** -0.5)
            (Or * dO.double()).sum().backward()
            qd = qkv.to(dtype).to(DEV).requires_grad_(True)
            gd = gate.to(DEV).requires_grad_(True)
            td = tab.to(DEV).requires_grad_(True)
            kd = kpm.to(DEV) if kpm is not None else None
            Od = F.AttnCoreFn.apply(qd, gd, td, kd, H, hd ** -0.5, 0.0, 0)
            Od.backward(dO.to(dtype).to(DEV))
            tag = f"attn[{dtype}] B={B} T={T} H={H} hd={hd} pad={use_pad}"
            out.append((tag + " O", err(Od, Or), tol))
            out.append((tag + " dqkv", err(qd.grad, qr.grad), tol * 2))
            out.append((tag + " dgate", err(gd.grad, gr.grad), tol * 2))
            out.append((tag + " dtab", err(td.grad, tr.grad), tol * 2))
        # no-bias path
        B, T, H, hd = 1, 40, 2, 32
        qkv = q(gen(B, T, 3 * H * hd, seed=7), dtype)
        Or = _ref_attention(qkv.double(), None, None, None, H, hd ** -0.5)
        Od = F.AttnCoreFn.apply(qkv.to(dtype).to(DEV), None, None, None, H, hd ** -0.5, 0.0, 0)
        out.append((f"attn[{dtype}] no bias O", err(Od, Or), tol))
    # fused kernels (bf16, head_dim 64) at the real frame count, with padding, vs the fp64 reference
    # (1, 999, 16): WavLM-Large's frame count and head count (20 s utterances); (2, 1000, 2, 'ragged'): a full-length row next
    # to one with 63 valid frames (937 padded keys: whole key tiles masked, dead query rows in the same block)
    for (B, T, H, use_pad) in [(2, 749, 3, True), (1, 300, 2, False), (1, 999, 16, False), (2, 1000, 2, 'ragged')]:
        hd, D, dtype, tol = 64, 64 * H, torch.bfloat16, TOLBF
        qkv = q(gen(B, T, 3 * D, seed=11), dtype)
        gate = 1 + 0.5 * gen(B, H, T, seed=12)
        tab = 0.5 * gen(H, 2 * T - 1, seed=13)
        kpm = None
        if use_pad:
            kpm = torch.zeros(B, T, dtype=torch.uint8)
            kpm[1, (63 if use_pad == 'ragged' else T - 100):] = 1
        dO = q(gen(B, T, D, seed=14), dtype)
        qr, gr, tr = qkv.double().requires_grad_(True), gate.double().requires_grad_(True), tab.double().requires_grad_(True)
        Or = _ref_attention(qr, gr, tr, kpm, H, hd ** -0.5)
        (Or * dO.double()).sum().backward()
        qd = qkv.to(dtype).to(DEV).requires_grad_(True)
        gd, td = gate.to(DEV).requires_grad_(True), tab.to(DEV).requires_grad_(True)
        kd = kpm.to(DEV) if kpm is not None else None
        # fused with stored probabilities (WAVLM_ATTN_STORE_P=1), with recomputation (=0, the default), with stored dropout bits
        # (=bits; no dropout here: nothing is stored), unfused composition
        for fused, store in ((True, True), (True, False), (True, "bits"), (False, False)):
            F.USE_FUSED_ATTENTION = fused
            F.ATTN_STORE_P = store
            for t in (qd, gd, td):
                t.grad = None
            Od = F.AttnCoreFn.apply(qd, gd, td, kd, H, hd ** -0.5, 0.0, 0)
            Od.backward(dO.to(dtype).to(DEV))
            tag = f"attn[{('fused, stored bits' if store == 'bits' else 'fused, stored P' if store else 'fused, recompute') if fused else 'unfused'} bf16] B={B} T={T} H={H} pad={use_pad}"
            out.append((tag + " O", err(Od, Or), tol))
            out.append((tag + " dqkv", err(qd.grad, qr.grad), tol * 2))
            out.append((tag + " dgate", err(gd.grad, gr.grad), tol * 2))
            out.append((tag + " dtab", err(td.grad, tr.grad), tol * 2))
        F.USE_FUSED_ATTENTION = True
        F.ATTN_STORE_P = ATTN_STORE_P_DEFAULT
    # fused dropout: deterministic per seed, keep fraction, and forward/backward agree on the mask
    B, T, H, hd = 1, 256, 2, 64
    D = H * hd
    qkv = (0.5 * gen(B, T, 3 * D, seed=21)).to(torch.bfloat16).to(DEV)
    ones_v = qkv.clone()
    ones_v[..., 2 * D:] = 1.0  # V = 1 -> O = sum_j P_drop = (kept mass) / (1 - p)
    O1 = F.AttnCoreFn.apply(ones_v, None, None, None, H, hd ** -0.5, 0.25, 77)
    O2 = F.AttnCoreFn.apply(ones_v, None, None, None, H, hd ** -0.5, 0.25, 77)
    O3 = F.AttnCoreFn.apply(ones_v, None, None, None, H, hd ** -0.5, 0.25, 78)
    out.append(("attn[fused] dropout deterministic", float((O1 != O2).float().mean().item()), 0.0))
    out.append(("attn[fused] dropout seed changes mask", 0.0 if (O1 != O3).any().item() else 1.0, 0.0))
    out.append(("attn[fused] dropout E[kept mass] ~ 1", abs(O1.float().mean().item() - 1.0), 0.02))
    # (forward / backward mask agreement and the numerics under dropout: check_dropout_exact -- exact, not statistical)
    # gate
    # (3, 64): 16-byte fast path with idle lanes, even row count; (12, 64) x 67 rows: Base geometry, odd row count (half
    # step at the end); (2, 32): generic kernel
    for dtype, (B, T, H, hd) in [(dt_, g_) for dt_ in (torch.float32, torch.bfloat16)
                                 for g_ in ((2, 33, 3, 64), (1, 67, 12, 64), (2, 19, 2, 32), (3, 41, 16, 64), (2, 3001, 12, 64))]:
        tol = tol_for(dtype)
        x = q(gen(B, T, H * hd, seed=1), dtype)
        W, b = q(0.2 * gen(8, hd, seed=2), dtype), q(0.1 * gen(8, seed=3), dtype)
        a = q(1 + 0.1 * gen(1, H, 1, 1, seed=4), dtype)
        xr, Wr, br, ar = [t.clone().requires_grad_(True) for t in (x, W, b, a)]
        ql = xr.view(B, T, H, hd).permute(0, 2, 1, 3)
        ga, gb = torch.sigmoid(TF.linear(ql, Wr, br).view(B, H, T, 2, 4).sum(-1)).chunk(2, dim=-1)
        gr = (ga * (gb * ar - 1.0) + 2.0).squeeze(-1)
        dg = gen(B, H, T, seed=5)
        (gr * dg).sum().backward()
        xd, Wd, bd, ad = [t.to(dtype).to(DEV).requires_grad_(True) for t in (x, W, b, a)]
        gdv = F.GateFn.apply(xd, Wd, bd, ad, H)
        gdv.backward(dg.to(DEV))
        tag = f"gate[{dtype}] H={H} hd={hd} rows={B * T}"
        out.append((tag + " gate", err(gdv, gr), tol))
        out.append((tag + " dx", err(xd.grad, xr.grad), tol * 2))
        out.append((tag + " dW", err(Wd.grad, Wr.grad), tol * 2))
        out.append((tag + " dbias", err(bd.grad, br.grad), tol * 2))
        out.append((tag + " dgrep_a", err(ad.grad, ar.grad), tol * 2))
        if hd == 64:  # accumulate form: the gate's gradient of x is added into a buffer that already holds another consumer's
            base = q(gen(B, T, H * hd, seed=6), dtype)
            acc = base.to(dtype).to(DEV).clone()
            _, ga_d, gb_d = ops.gate_fwd(xd.detach(), Wd.detach(), bd.detach(), ad.detach().view(-1), H)
            dx2, _, _, _ = ops.gate_bwd(dg.to(DEV), xd.detach(), Wd.detach(), bd.detach(), ad.detach().view(-1), ga_d, gb_d, H,
                                        dx_accumulate=acc)
            out.append((tag + " dx (accumulate)", err(dx2, xr.grad + base), tol * 2))
    # relpos table
    emb = gen(32, 4, seed=1).requires_grad_(True)
    bucket = torch.randint(0, 32, (97,), generator=torch.Generator().manual_seed(0)).to(torch.int32)
    tr = emb[bucket.long()].t()
    dt_ = gen(4, 97, seed=2)
    (tr * dt_).sum().backward()
    ed = emb.detach().to(DEV).requires_grad_(True)
    td = F.RelPosTableFn.apply(ed, bucket.to(DEV))
    td.backward(dt_.to(DEV))
    out.append(("relpos table", err(td, tr), 1e-6))
    out.append(("relpos table grad", err(ed.grad, emb.grad), 1e-5))
    return out


# ------------------------------------------------------------------------------------------- exact dropout parity
def _attn_kernel_masks(B, H, T, p_drop, seed, gate, tab, kpm, qseed=41, store_p=False):
    """The keep mask each of the three fused attention kernels ACTUALLY applies, read out of the kernels' own results
    (no debug entry point, no re-implementation of the hash): bool [B, H, T, T] (query, key) each.
      forward   V = one-hot over a 64-key chunk  ->  O[i, u] = P_drop[i, j0 + u]: kept iff non-zero;
      dK/dV     dO = one-hot over a 64-query chunk  ->  dV[j, u] = P_drop[i0 + u, j]: kept iff non-zero;
      dQ        K = one-hot over a 64-key chunk, V = dO = e_0 (so dP = 1 everywhere)  ->  dQ[i, u] = scale * dS[i, j0 + u]
                with dS = P sc (keep - kappa_i), kappa_i = kept probability mass of row i in (0, 1): kept iff positive.
    Padded keys (P = 0) carry no information and are reported as False by all three."""
    hd = 64
    D = H * hd
    scale = hd ** -0.5
    qv = (0.5 * gen(B, T, D, seed=qseed)).to(torch.bfloat16).to(DEV)
    kv = (0.5 * gen(B, T, D, seed=qseed + 1)).to(torch.bfloat16).to(DEV)
    keep_f = torch.zeros(B, H, T, T, dtype=torch.bool)
    keep_q = torch.zeros(B, H, T, T, dtype=torch.bool)
    keep_kv = torch.zeros(B, H, T, T, dtype=torch.bool)
    hsel = torch.arange(H, device=DEV) * hd
    for c0 in range(0, T, 64):
        n = min(64, T - c0)
        u = torch.arange(n, device=DEV)
        onehot = torch.zeros(B, T, D, dtype=torch.bfloat16, device=DEV)
        for h in range(H):
            onehot[:, c0 + u, h * hd + u] = 1.0
        # forward: V one-hot on the key chunk
        qkv = torch.cat([qv, kv, onehot], dim=-1).contiguous()
        O, lse, _ = ops.attn_fused_fwd(qkv, gate, tab, kpm, H, scale, p_drop, seed, store_p=store_p)
        keep_f[:, :, :, c0:c0 + n] = (O.view(B, T, H, hd)[..., :n] != 0).permute(0, 2, 1, 3).cpu()
        # dK/dV: dO one-hot on the query chunk (any V)
        qkv2 = torch.cat([qv, kv, kv], dim=-1).contiguous()
        O2, lse2, ps2 = ops.attn_fused_fwd(qkv2, gate, tab, kpm, H, scale, p_drop, seed, store_p=store_p)
        dqkv, _, _ = ops.attn_fused_bwd(qkv2, O2, onehot, lse2, gate, tab, kpm, H, scale, p_drop, seed, pstore=ps2)
        dV = dqkv[..., 2 * D:].view(B, T, H, hd)[..., :n]                       # [b, j, h, u] = P_drop[i0 + u, j]
        keep_kv[:, :, c0:c0 + n, :] = (dV != 0).permute(0, 2, 3, 1).cpu()
        # dQ: K one-hot on the key chunk, V = dO = e_0
        e0 = torch.zeros(B, T, D, dtype=torch.bfloat16, device=DEV)
        e0[..., hsel] = 1.0
        qkv3 = torch.cat([qv, onehot, e0], dim=-1).contiguous()
        O3, lse3, ps3 = ops.attn_fused_fwd(qkv3, gate, tab, kpm, H, scale, p_drop, seed, store_p=store_p)
        dqkv3, _, _ = ops.attn_fused_bwd(qkv3, O3, e0, lse3, gate, tab, kpm, H, scale, p_drop, seed, pstore=ps3)
        dQ = dqkv3[..., :D].view(B, T, H, hd)[..., :n]
        keep_q[:, :, :, c0:c0 + n] = (dQ > 0).permute(0, 2, 1, 3).cpu()
    if kpm is not None:
        valid = ~kpm.bool().cpu()[:, None, None, :]
        keep_f, keep_q, keep_kv = keep_f & valid, keep_q & valid, keep_kv & valid
    return keep_f, keep_q, keep_kv


def _ref_attention_masked(qkv, gate, tab, kpm, H, scale, keep, sc):
    """_ref_attention with an explicit dropout keep mask [B, H, T, T] and scale sc = 1 / (1 - p) on the probabilities"""
    B, T, D3 = qkv.shape
    D = D3 // 3
    hd = D // H
    qh = qkv[..., :D].view(B, T, H, hd).permute(0, 2, 1, 3)
    kh = qkv[..., D:2 * D].view(B, T, H, hd).permute(0, 2, 1, 3)
    vh = qkv[..., 2 * D:].view(B, T, H, hd).permute(0, 2, 1, 3)
    s = (qh @ kh.transpose(-1, -2)) * scale
    if tab is not None:
        i = torch.arange(T)[:, None]
        j = torch.arange(T)[None, :]
        s = s + gate.unsqueeze(-1) * tab[:, (j - i) + T - 1].unsqueeze(0)
    if kpm is not None:
        s = s.masked_fill(kpm.bool()[:, None, None, :], float("-inf"))
    pr = torch.softmax(s, dim=-1) * keep.to(s.dtype) * sc
    return (pr @ vh).permute(0, 2, 1, 3).reshape(B, T, D)


def check_dropout_exact():
    """Dropout-on is the benchmarked mode: its parity must not be statistical.  (1) The keep masks the forward, dQ and dK/dV
    attention kernels apply are read out of the kernels themselves and must be IDENTICAL -- in the recompute mode three kernels
    regenerate the mask independently (the dK/dV one with a different word-sharing scheme), in the stored-probability mode
    (the default) the backward kernels take the decision from the sign bit the forward stored, through two different
    read paths (register fragments / LDS gather + transposing read); and the storing forward must drop exactly what the
    plain forward drops.  (2) With that mask the fused forward and backward
    are compared with the fp64 reference at the usual bf16 tolerance (multihead_attention.py:278-300 with
    dropout_p = attention_dropout).  (3) The same for the dropouts fused into the LayerNorm kernels (residual-branch
    dropout of post- and pre-LN blocks incl. the fused pre-LN residual stream, output dropout) and the dropout-add."""
    out = []
    p_drop = 0.25   # 16384 / 65536: the kernels' 16-bit threshold represents it exactly, sc = 4/3
    sc = 1.0 / (1.0 - p_drop)
    cases = [(1, 2, 256, False, False, 1234567), (2, 2, 200, True, True, 0x9E3779B97F4A7C15), (1, 16, 331, True, False, 77)]
    for (B, H, T, use_tab, use_pad, seed) in cases:
        gate = tab = kpm = None
        if use_tab:
            gate = (1 + 0.5 * gen(B, H, T, seed=2)).to(DEV)
            tab = (0.5 * gen(H, 2 * T - 1, seed=3)).to(DEV)
        if use_pad:
            kpm = torch.zeros(B, T, dtype=torch.uint8)
            kpm[1, T - 37:] = 1
            kpm = kpm.to(DEV)
        kf, kq, kkv = _attn_kernel_masks(B, H, T, p_drop, seed, gate, tab, kpm)
        tag = f"dropout-exact attn B={B} H={H} T={T} tab={use_tab} pad={use_pad}"
        nvalid = float((~kpm.bool().cpu()).float().mean().item()) if kpm is not None else 1.0
        out.append((tag + " keep fraction", abs(kf.float().mean().item() / nvalid - (1 - p_drop)), 0.01))
        out.append((tag + " mask fwd == dQ (mismatching elements)", float((kf != kq).sum().item()), 0.0))
        out.append((tag + " mask fwd == dK/dV (mismatching elements)", float((kf != kkv).sum().item()), 0.0))
        sf, sq, skv = _attn_kernel_masks(B, H, T, p_drop, seed, gate, tab, kpm, store_p=True)
        out.append((tag + " stored P: storing fwd == plain fwd (mismatching elements)", float((sf != kf).sum().item()), 0.0))
        out.append((tag + " stored P: mask fwd == dQ (mismatching elements)", float((sf != sq).sum().item()), 0.0))
        out.append((tag + " stored P: mask fwd == dK/dV (mismatching elements)", float((sf != skv).sum().item()), 0.0))
        # stored dropout BITS (round 6, WAVLM_ATTN_STORE_P=bits): the forward writes its decisions as bit words, the dQ kernel reads them per
        # row (v_bfe_i32 at the forward's bit order), the dK/dV kernel through its per-row LDS arrays (bit of the lane's key)
        bf, bq, bkv = _attn_kernel_masks(B, H, T, p_drop, seed, gate, tab, kpm, store_p="bits")
        out.append((tag + " stored bits: storing fwd == plain fwd (mismatching elements)", float((bf != kf).sum().item()), 0.0))
        out.append((tag + " stored bits: mask fwd == dQ (mismatching elements)", float((bf != bq).sum().item()), 0.0))
        out.append((tag + " stored bits: mask fwd == dK/dV (mismatching elements)", float((bf != bkv).sum().item()), 0.0))
        # (2) numerics with the forward's own mask
        D = 64 * H
        qkv = q(gen(B, T, 3 * D, seed=11), torch.bfloat16)
        dO = q(gen(B, T, D, seed=14), torch.bfloat16)
        gc, tc = (gate.cpu(), tab.cpu()) if use_tab else (None, None)
        qr = qkv.double().requires_grad_(True)
        gr = gc.double().requires_grad_(True) if use_tab else None
        tr = tc.double().requires_grad_(True) if use_tab else None
        Or = _ref_attention_masked(qr, gr, tr, kpm.cpu() if kpm is not None else None, H, 64 ** -0.5, kf, sc)
        (Or * dO.double()).sum().backward()
        for store in (True, False, "bits"):
            F.ATTN_STORE_P = store
            tg2 = tag + (" [stored bits]" if store == "bits" else " [stored P]" if store else " [recompute]")
            qd = qkv.to(torch.bfloat16).to(DEV).requires_grad_(True)
            gd = gate.clone().requires_grad_(True) if use_tab else None
            td = tab.clone().requires_grad_(True) if use_tab else None
            Od = F.AttnCoreFn.apply(qd, gd, td, kpm, H, 64 ** -0.5, p_drop, seed)
            Od.backward(dO.to(torch.bfloat16).to(DEV))
            out.append((tg2 + " O vs fp64 with the kernel's mask", err(Od, Or), TOLBF))
            out.append((tg2 + " dqkv vs fp64 with the kernel's mask", err(qd.grad, qr.grad), TOLBF * 2))
            if use_tab:
                out.append((tg2 + " dgate", err(gd.grad, gr.grad), TOLBF * 2))
                out.append((tg2 + " dtab", err(td.grad, tr.grad), TOLBF * 2))
        F.ATTN_STORE_P = ATTN_STORE_P_DEFAULT
    # (3) LayerNorm-fused dropouts
    for dtype in (torch.float32, torch.bfloat16):
        tol = tol_for(dtype)
        for D in (768, 1024):
            rows, p = 75, 0.25
            x, r = q(gen(rows, D, seed=1), dtype), q(gen(rows, D, seed=2), dtype)
            g, b = q(1 + 0.1 * gen(D, seed=3), dtype), q(0.1 * gen(D, seed=4), dtype)
            dy, ds = q(gen(rows, D, seed=5), dtype), q(gen(rows, D, seed=6), dtype)
            dev = lambda t: t.to(dtype).to(DEV)
            # the forward's residual-dropout mask: x = 0, r = 1  ->  s = keep / (1 - p)
            _, s_probe = F.LayerNormFn.apply(dev(torch.zeros(rows, D)), dev(torch.ones(rows, D)), dev(g), dev(b), 1e-5, 0, p, 4711,
                                             0.0, 0, 1.0)
            keep = (s_probe != 0).cpu()
            tag = f"dropout-exact layernorm[{dtype}] D={D}"
            out.append((tag + " residual-dropout keep fraction", abs(keep.float().mean().item() - (1 - p)), 0.02))
            for s_grad in (False, True):
                xr, rr, gr, br = [t.clone().double().requires_grad_(True) for t in (x, r, g, b)]
                s_ref = xr + rr * keep.double() / (1 - p)
                s_q = s_ref + (s_ref.detach().to(dtype).double() - s_ref.detach())  # the kernel rounds s to `dtype`
                yr = TF.layer_norm(s_q, (D,), gr, br, 1e-5)
                ((yr * dy.double()).sum() + ((s_q * ds.double()).sum() if s_grad else 0.0)).backward()
                xd, rd, gd, bd = [dev(t).requires_grad_(True) for t in (x, r, g, b)]
                res = F.LayerNormFn.apply(xd, rd, gd, bd, 1e-5, 0, p, 4711, 0.0, 0, 1.0, None, False, s_grad)
                y, s_out = res[0], res[1]
                if s_grad:
                    (y.float() * dev(dy).float()).sum().add((s_out.float() * dev(ds).float()).sum()).backward()
                else:
                    y.backward(dev(dy))
                t2 = tag + (" fused residual stream" if s_grad else "")
                out.append((t2 + " backward mask == forward mask (mismatching elements)",
                            float((((rd.grad != 0).cpu() != keep) & (xd.grad != 0).cpu()).sum().item()), 0.0))
                out.append((t2 + " y", err(y, yr), tol))
                out.append((t2 + " dx", err(xd.grad, xr.grad), tol))
                out.append((t2 + " dr", err(rd.grad, rr.grad), tol))
                out.append((t2 + " dgamma", err(gd.grad, gr.grad), tol * 2))
            # output dropout (the encoder's first LayerNorm): mask from the forward, gradient through the same mask
            xd = dev(x).requires_grad_(True)
            yp, _ = F.LayerNormFn.apply(xd, None, dev(g), dev(b), 1e-5, 0, 0.0, 0, p, 999, 1.0)
            keep_o = (yp != 0).cpu()
            yp.backward(dev(dy))
            xr = x.clone().double().requires_grad_(True)
            yr = TF.layer_norm(xr, (D,), g.double(), b.double(), 1e-5) * keep_o.double() / (1 - p)
            (yr * dy.double()).sum().backward()
            out.append((tag + " output-dropout y", err(yp, yr), tol))
            out.append((tag + " output-dropout dx with the forward's mask", err(xd.grad, xr.grad), tol))
        # dropout-add of the unfused pre-LN block: y = x + dropout(r); backward dr = dropout(dy) with the same mask
        from unispeech_amd.wavlm import ResidualAddFn
        rows, D, p = 64, 1024, 0.1
        xz = torch.zeros(rows, D, dtype=dtype, device=DEV)
        r1 = torch.ones(rows, D, dtype=dtype, device=DEV, requires_grad=True)
        ya = ResidualAddFn.apply(xz, r1, p, 31337)
        ya.backward(torch.ones_like(ya))
        out.append((f"dropout-exact dropout_add[{dtype}] backward mask == forward mask",
                    float(((ya != 0) != (r1.grad != 0)).sum().item()), 0.0))
        out.append((f"dropout-exact dropout_add[{dtype}] keep fraction", abs((ya != 0).float().mean().item() - (1 - p)), 0.02))
    return out


# ---------------------------------------------------------------------------------------- pos_conv / FFN / linear
def check_posconv():
    out = []
    for dtype in (torch.float32, torch.bfloat16):
        tol = tol_for(dtype)
        # (.., 768, 128, 16) / (.., 1024, 128, 16) in bf16 run the direct-convolution kernel (Cg = 48 / 64), at one, two
        # frame segments and every tile height; the rest the overlapping-row GEMM form
        cases = [(2, 49, 64, 16, 4), (2, 75, 768, 128, 16)]
        if dtype == torch.bfloat16:
            cases += [(1, 749, 768, 128, 16), (2, 400, 768, 128, 16), (1, 999, 1024, 128, 16), (2, 330, 1024, 128, 16)]
        for (B, T, D, K, G) in cases:
            Cg = D // G
            x = q(gen(B, T, D, seed=1), dtype)
            v = q(gen(D, Cg, K, seed=2, scale=math.sqrt(4.0 / (K * D))), dtype)
            g = q(v.norm(dim=(0, 1), keepdim=True) * (1 + 0.1 * gen(1, 1, K, seed=3)), dtype)
            bias = q(0.1 * gen(D, seed=4), dtype)
            xr, vr, gr, br = [t.clone().requires_grad_(True) for t in (x, v, g, bias)]
            w = gr * vr / vr.norm(dim=(0, 1), keepdim=True)
            yc = TF.conv1d(xr.transpose(1, 2), w, br, padding=K // 2, groups=G)[:, :, :T]
            yr = xr + TF.gelu(yc).transpose(1, 2)
            dy = q(gen(B, T, D, seed=5), dtype)
            (yr * dy).sum().backward()
            xd, vd, gd, bd = [t.to(dtype).to(DEV).requires_grad_(True) for t in (x, v, g, bias)]
            yd = F.PosConvFn.apply(xd, vd, gd, bd, G)
            yd.backward(dy.to(dtype).to(DEV))
            tag = f"posconv[{dtype}] D={D} K={K} G={G}"
            out.append((tag + " y", err(yd, yr), tol))
            out.append((tag + " dx", err(xd.grad, xr.grad), tol * 2))
            out.append((tag + " dv", err(vd.grad, vr.grad), tol * 3))
            out.append((tag + " dg", err(gd.grad, gr.grad), tol * 3))
            out.append((tag + " dbias", err(bd.grad, br.grad), tol * 2))
            if dtype == torch.bfloat16 and ops.posconv_direct_supported(dtype, Cg, K, T):
                # the two activation-side implementations against each other (same bf16 inputs, fp32 accumulation)
                F.POSCONV_DIRECT = False
                try:
                    x2 = xd.detach().clone().requires_grad_(True)
                    v2 = vd.detach().clone().requires_grad_(True)
                    y2 = F.PosConvFn.apply(x2, v2, gd.detach(), bd.detach(), G)
                    y2.backward(dy.to(dtype).to(DEV))
                finally:
                    F.POSCONV_DIRECT = True
                out.append((tag + " direct vs gemm y", err(yd, y2), 1.0e-2))
                out.append((tag + " direct vs gemm dx", err(xd.grad, x2.grad), 1.0e-2))
                out.append((tag + " direct vs gemm dv", err(vd.grad, v2.grad), 1.0e-2))
    return out


def check_gemm_colsum():
    """column sums of C out of the GEMM epilogue (192 x 384 and 256 x 256 ping-pong kernels) and by the fall-back pass
    (128-wide kernel, fp32), against an explicit sum over the rows of the C the same call stored"""
    out = []
    bf = torch.bfloat16
    for (n, N, K, epi, dtype) in [(1000, 3072, 768, 4, bf), (1000, 768, 3072, 0, bf), (777, 2048, 512, 4, bf),
                                  (300, 2048, 512, 0, bf), (500, 48, 256, 0, bf), (300, 256, 128, 0, torch.float32)]:
        # dx[n, K'] = dy[n, N'] @ W[N', K'] in the dX form the model uses (B K-strided); here N plays K'
        dy = q(gen(n, K, seed=1), dtype).to(dtype).to(DEV)
        W = q(gen(K, N, seed=2, scale=1.0 / math.sqrt(K)), dtype).to(dtype).to(DEV)
        aux = q(gen(n, N, seed=3), dtype).to(dtype).to(DEV) if epi == 4 else None
        Cc = torch.empty((n, N), dtype=dtype, device=DEV)
        base = q(gen(N, seed=4), torch.float32).to(DEV)
        cs = base.clone()
        ops.gemm(dy, W, Cc, n, N, K, lda=K, ldb=N, ldc=N, transB=True, epi=epi, aux=aux, ld_aux=N, colsum=cs,
                 colsum_accumulate=True)
        ref = dy.double() @ W.double()
        if epi == 4:
            ref = ref * aux.double()
        tag = f"gemm colsum[{dtype}] n={n} N={N} K={K} epi={epi}"
        out.append((tag + " C", err(Cc, ref), tol_for(dtype)))
        out.append((tag + " colsum vs stored C", err(cs - base, Cc.double().sum(0)), 2e-3 if dtype == bf else 1e-5))
        out.append((tag + " colsum vs exact", err(cs - base, ref.sum(0)), 5e-3 if dtype == bf else 1e-5))
    return out


def check_linear_ffn():
    out = []
    for dtype in (torch.float32, torch.bfloat16):
        tol = tol_for(dtype)
        n, D, Fd = 300, 768, 3072
        x = q(gen(2, n // 2, D, seed=1), dtype)
        W1, b1 = q(0.03 * gen(Fd, D, seed=2), dtype), q(0.1 * gen(Fd, seed=3), dtype)
        W2, b2 = q(0.03 * gen(D, Fd, seed=4), dtype), q(0.1 * gen(D, seed=5), dtype)
        dy = q(gen(2, n // 2, D, seed=6), dtype)
        xr, W1r, b1r, W2r, b2r = [t.clone().requires_grad_(True) for t in (x, W1, b1, W2, b2)]
        yr = TF.linear(TF.gelu(TF.linear(xr, W1r, b1r)), W2r, b2r)
        (yr * dy).sum().backward()
        xd, W1d, b1d, W2d, b2d = [t.to(dtype).to(DEV).requires_grad_(True) for t in (x, W1, b1, W2, b2)]
        yd = F.FFNFn.apply(xd, W1d, b1d, W2d, b2d, 0.0, 0)
        yd.backward(dy.to(dtype).to(DEV))
        tag = f"ffn[{dtype}]"
        for nm, a, b in [("y", yd, yr), ("dx", xd.grad, xr.grad), ("dW1", W1d.grad, W1r.grad), ("db1", b1d.grad, b1r.grad),
                         ("dW2", W2d.grad, W2r.grad), ("db2", b2d.grad, b2r.grad)]:
            out.append((f"{tag} {nm}", err(a, b), tol * (1 if nm == "y" else 2)))
        xr2, Wr2, br2 = x.clone().requires_grad_(True), W1.clone().requires_grad_(True), b1.clone().requires_grad_(True)
        y2 = TF.linear(xr2, Wr2, br2)
        dy2 = q(gen(*y2.shape, seed=7), dtype)
        (y2 * dy2).sum().backward()
        xd2, Wd2, bd2 = [t.to(dtype).to(DEV).requires_grad_(True) for t in (x, W1, b1)]
        yd2 = F.LinearFn.apply(xd2, Wd2, bd2)
        yd2.backward(dy2.to(dtype).to(DEV))
        for nm, a, b in [("y", yd2, y2), ("dx", xd2.grad, xr2.grad), ("dW", Wd2.grad, Wr2.grad), ("db", bd2.grad, br2.grad)]:
            out.append((f"linear[{dtype}] {nm}", err(a, b), tol * (1 if nm == "y" else 2)))
    return out


# ------------------------------------------------------------------------------------------------------- loss
def check_loss():
    from oracle import wavlm_oracle as O
    out = []
    for dtype in (torch.float32, torch.bfloat16):
        tol = tol_for(dtype)
        for (S, V, Fd) in [(57, 23, 32), (700, 504, 256)]:
            proj = q(gen(S, Fd, seed=1), dtype)
            emb = q(torch.rand(V, Fd, generator=torch.Generator().manual_seed(2)), dtype)
            tgt = torch.randint(0, V, (S,), generator=torch.Generator().manual_seed(3))
            pr, er = proj.clone().requires_grad_(True), emb.clone().requires_grad_(True)
            pos = er[tgt]
            negs = er.unsqueeze(1).expand(-1, S, -1)
            logits = O.compute_nce(pr, pos, negs, 0.1)
            lossr = TF.cross_entropy(logits.float(), torch.zeros(S, dtype=torch.long), reduction="sum")
            corr = ((logits.argmax(-1) == 0) & ~(logits.argmin(-1) == 0)).sum()
            (lossr * 1.7).backward()
            pd, ed = proj.to(dtype).to(DEV).requires_grad_(True), emb.to(dtype).to(DEV).requires_grad_(True)
            loss, nc = F.MaskedPredLossFn.apply(pd, ed, tgt.to(torch.int32).to(DEV), 0.1, True)
            (loss * 1.7).sum().backward()
            tag = f"loss[{dtype}] S={S} V={V}"
            out.append((tag + " loss", err(loss, lossr.reshape(1)), tol))
            out.append((tag + " correct", abs(nc.item() - corr.item()), 0.0 if dtype == torch.float32 else max(2.0, 0.02 * S)))
            out.append((tag + " dproj", err(pd.grad, pr.grad), tol * 3))
            out.append((tag + " demb", err(ed.grad, er.grad), tol * 3))
    f = gen(2, 49, 32, seed=1)
    fr = f.clone().requires_grad_(True)
    (fr.pow(2).mean() * 3.0).backward()
    fd = f.to(DEV).requires_grad_(True)
    pen = F.FeaturesPenFn.apply(fd)
    (pen * 3.0).sum().backward()
    out.append(("features_pen", err(pen, f.pow(2).mean().reshape(1)), 1e-5))
    out.append(("features_pen grad", err(fd.grad, fr.grad), 1e-5))
    # sampled-instance cosine logits + BCE (UniSpeech-SAT utterance-contrastive head): gathered rows, duplicates, the
    # row itself as column 0
    for dtype in (torch.float32, torch.bfloat16):
        tol = tol_for(dtype)
        S, N, C = 301, 9, 256
        y = q(gen(S, C, seed=41), dtype)
        gi = torch.Generator().manual_seed(5)
        idx = torch.cat([torch.arange(S).view(S, 1), torch.randint(0, S, (S, N), generator=gi)], dim=1)
        tg = torch.cat([torch.ones(S, 1, dtype=torch.bool), torch.rand(S, N, generator=gi) < 0.3], dim=1)
        yr = y.clone().double().requires_grad_(True)
        cand = yr[idx.view(-1)].view(S, N + 1, C)
        lg = torch.cosine_similarity(yr.unsqueeze(1), cand, dim=-1) / 0.1
        lr_ = TF.binary_cross_entropy_with_logits(lg, tg.double(), reduction="none").mean()
        lr_.backward()
        yd = y.to(dtype).to(DEV).requires_grad_(True)
        ld, acc = F.UttContrastiveLossFn.apply(yd, idx.to(torch.int32).to(DEV), tg.to(torch.uint8).to(DEV), 0.1)
        ld.sum().backward()
        tag = f"utt_contrastive[{dtype}]"
        out.append((tag + " loss", abs(ld.item() - lr_.item()) / abs(lr_.item()), tol))
        out.append((tag + " accuracy", abs(acc.item() - ((lg >= 0) == tg).double().mean().item()), 1e-6 if dtype == torch.float32 else 0.02))
        out.append((tag + " dproj", err(yd.grad, yr.grad), tol * 3))
    return out


def check_adam():
    from oracle import wavlm_oracle as O
    out = []
    n = 10007
    p, g = gen(n, seed=1), gen(n, seed=2, scale=0.1)
    m, v = torch.zeros(n), torch.zeros(n)
    pd, md, vd = p.clone().to(DEV), m.clone().to(DEV), v.clone().to(DEV)
    plow = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    gd = g.to(torch.bfloat16).to(DEV)
    gq = g.to(torch.bfloat16).float()
    pr, mr, vr = p.clone(), m.clone(), v.clone()
    gn = ops.sumsq(gd)
    mult, max_norm = 0.5, 1.0
    total = gq.norm().item() * mult
    clip = min(1.0, max_norm / (total + 1e-6))
    for step in (1, 2, 3):
        ops.adam_step(pd, md, vd, gd, plow, lr=5e-4, beta1=0.9, beta2=0.98, eps=1e-6, weight_decay=0.01, step=step,
                      grad_mult=mult, gnorm_sq=gn, max_norm=max_norm)
        pr, mr, vr = O.adam_reference_step(pr, gq * mult * clip, mr, vr, step, 5e-4, 0.9, 0.98, 1e-6, 0.01)
    out.append(("adam p", err(pd, pr), 1e-5))
    out.append(("adam m", err(md, mr), 1e-5))
    out.append(("adam v", err(vd, vr), 1e-5))
    out.append(("adam low-precision copy", err(plow, pr), 1e-2))
    return out


def check_activations():
    """elementwise feed-forward activations and gated linear units (wavlm_act_* / wavlm_glu_*) against torch in fp64, forward
    and backward, both dtypes; and FFNFn with every activation_fn against the same composition in torch"""
    import torch.nn.functional as tF
    from unispeech_amd import functional as Fn
    out = []
    refs = {"relu": torch.relu, "tanh": torch.tanh, "gelu": lambda x: tF.gelu(x),
            "gelu_accurate": lambda x: 0.5 * x * (1 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * x ** 3)))}
    gates = {"sigmoid": torch.sigmoid, "swish": lambda b: b * torch.sigmoid(b), "relu": torch.relu, "gelu": lambda b: tF.gelu(b),
             "bilinear": lambda b: b}
    for dtype in (torch.float32, torch.bfloat16):
        tol = tol_for(dtype)
        x = q(2.0 * gen(301, 96, seed=5), dtype)
        dy = q(gen(301, 96, seed=6), dtype)
        for kind, f in refs.items():
            xr = x.double().requires_grad_(True)
            yr = f(xr)
            (dxr,) = torch.autograd.grad(yr, xr, dy.double())
            y = ops.act_fwd(x.to('cuda', dtype), kind)
            dx = ops.act_bwd(x.to('cuda', dtype), dy.to('cuda', dtype), kind)
            out.append((f"act[{dtype}] {kind} fwd", err(y.float().cpu(), yr.detach().float()), tol))
            out.append((f"act[{dtype}] {kind} bwd", err(dx.float().cpu(), dxr.float()), tol))
        dyh = q(gen(301, 48, seed=7), dtype)
        for gate, g in gates.items():
            xr = x.double().requires_grad_(True)
            yr = xr[:, :48] * g(xr[:, 48:])
            (dxr,) = torch.autograd.grad(yr, xr, dyh.double())
            y = ops.glu_fwd(x.to('cuda', dtype), gate)
            dx = ops.glu_bwd(x.to('cuda', dtype), dyh.to('cuda', dtype), gate)
            out.append((f"glu[{dtype}] {gate} fwd", err(y.float().cpu(), yr.detach().float()), tol))
            out.append((f"glu[{dtype}] {gate} bwd", err(dx.float().cpu(), dxr.float()), tol))
        # the feed-forward block end to end, every activation_fn (no dropout)
        n, D, Fd = 200, 64, 128
        for act in ("gelu", "relu", "gelu_accurate", "tanh", "linear", "glu"):
            xin = q(gen(n, D, seed=11), dtype)
            W1 = q(0.2 * gen(Fd * (2 if act == "glu" else 1), D, seed=12), dtype)
            b1 = q(0.1 * gen(W1.shape[0], seed=13), dtype)
            W2, b2 = q(0.2 * gen(D, Fd, seed=14), dtype), q(0.1 * gen(D, seed=15), dtype)
            dyo = q(gen(n, D, seed=16), dtype)
            ts = [t.double().requires_grad_(True) for t in (xin, W1, b1, W2, b2)]
            u = tF.linear(ts[0], ts[1], ts[2])
            if act == "glu":
                h = u[:, :Fd] * (u[:, Fd:] * torch.sigmoid(u[:, Fd:]))
            elif act == "linear":
                h = u
            else:
                h = refs[act](u)
            yr = tF.linear(h, ts[3], ts[4])
            gr = torch.autograd.grad(yr, ts, dyo.double())
            td = [t.to('cuda', dtype).requires_grad_(True) for t in (xin, W1, b1, W2, b2)]
            y = Fn.FFNFn.apply(td[0], td[1], td[2], td[3], td[4], 0.0, 0, None, None, False, act)
            gd = torch.autograd.grad(y, td, dyo.to('cuda', dtype))
            out.append((f"ffn[{dtype}] {act} y", err(y.detach().float().cpu(), yr.detach().float()), tol))
            for nm, a, b in zip(("dx", "dW1", "db1", "dW2", "db2"), gd, gr):
                out.append((f"ffn[{dtype}] {act} {nm}", err(a.float().cpu(), b.float()), tol))
    return out


GROUPS = {
    "gemm": check_gemm, "gemm_pp": check_gemm_pp, "gemm_pp3": check_gemm_pp3, "gemm_w4": check_gemm_w4, "gemm_grouped": check_gemm_grouped, "gemm_race": check_gemm_race, "layernorm": check_layernorm, "rowops": check_rowops, "conv0": check_conv0, "conv0_ln": check_conv0_ln, "conv_ln_block": check_conv_ln_block,
    "convstack": check_convstack, "attention": check_attention, "posconv": check_posconv, "gemm_colsum": check_gemm_colsum,
    "linear_ffn": check_linear_ffn, "activations": check_activations, "loss": check_loss, "adam": check_adam, "dropout_exact": check_dropout_exact,
}

if __name__ == "__main__":
    import json
    name = sys.argv[1]
    res = GROUPS[name]()
    bad = 0
    for (nm, e, t) in res:
        ok = e <= t
        bad += (not ok)
        print(("ok   " if ok else "FAIL ") + f"{nm}: err={e:.3e} tol={t:.1e}")
    print(json.dumps({"group": name, "n": len(res), "failed": bad}))
    sys.exit(1 if bad else 0)
