"""Thin functional layer over the C ABI: torch tensors in, kernel launches on torch's current HIP stream.

torch is used here for device memory (allocation, views), the current stream and nothing else: every function
below ends in exactly one (or a fixed short sequence of) libwavlm_hip.so entry points.  There is no CPU path --
a CPU tensor raises.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import BF16, F32, ConvRelayoutDesc, GemmDesc, check

_WS = {}


def dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError("unsupported dtype %s (float32 / bfloat16 only)" % t.dtype)


def _dev(t):
    if not t.is_cuda:
        raise _lib.WavlmHipError("unispeech_amd kernels need a HIP device tensor (no CPU fallback)")
    return t.device


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """torch's current HIP stream of the current device as a raw handle (one per launch: the Stream-object route costs
    ~1.5 us of the launch thread per call)"""
    if _RAW_STREAM is not None:
        return C.c_void_p(_RAW_STREAM(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, offset=0):
    """device address of element `offset` of tensor t (None -> NULL)"""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr() + offset * t.element_size())


def workspace(device, nbytes, tag="main"):
    """grow-only scratch buffer per (device, tag, stream): reuse is safe because the launches of one stream are ordered"""
    key = (device.index, tag, _RAW_STREAM(device.index) if _RAW_STREAM is not None else torch.cuda.current_stream(device).cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


def _contig(t):
    if t is not None and not t.is_contiguous():
        raise ValueError("expected a contiguous tensor")
    return t


# --------------------------------------------------------------------------------------------- GEMM
def gemm(A, B, Cc, M, N, K, *, lda, ldb, ldc, transA=False, transB=False, KB=1, sA_kb=0, sB_kb=0, batch=(1, 1),
         sA=(0, 0), sB=(0, 0), sC=(0, 0), a_off=0, b_off=0, c_off=0, alpha=1.0, bias=None, bias_off=0, sBias=(0, 0),
         epi=0, aux=None, aux_off=0, ld_aux=0, sAux=(0, 0), res=None, res_off=0, ld_res=0, sRes=(0, 0),
         accumulate=False, split_k=1, colsum=None, colsum_accumulate=False):
    """C = epi(alpha * sum_kb A.B^T + bias) (+res) (+C); see include/wavlm_hip.h for the addressing rules.
    colsum [N] (optional): (+)= column sums of C over its rows (a bias gradient), fused into the epilogue where possible."""
    dev = _dev(A)
    if A.dtype != B.dtype:
        raise TypeError("A and B must share a dtype")
    d = GemmDesc()
    d.dtype, d.c_dtype = dt(A), dt(Cc)
    d.M, d.N, d.K, d.KB = int(M), int(N), int(K), int(KB)
    d.transA, d.transB = int(bool(transA)), int(bool(transB))
    d.lda, d.ldb, d.ldc = int(lda), int(ldb), int(ldc)
    d.sA_kb, d.sB_kb = int(sA_kb), int(sB_kb)
    d.batch_o, d.batch_i = int(batch[0]), int(batch[1])
    d.sA_o, d.sA_i = int(sA[0]), int(sA[1])
    d.sB_o, d.sB_i = int(sB[0]), int(sB[1])
    d.sC_o, d.sC_i = int(sC[0]), int(sC[1])
    d.A, d.B, d.C = ptr(A, a_off), ptr(B, b_off), ptr(Cc, c_off)
    d.alpha, d.epi = float(alpha), int(epi)
    d.bias = ptr(bias, bias_off)
    d.bias_dtype = dt(bias) if bias is not None else 0
    d.sBias_o, d.sBias_i = int(sBias[0]), int(sBias[1])
    d.aux = ptr(aux, aux_off)
    d.aux_dtype = dt(aux) if aux is not None else 0
    d.ld_aux, d.sAux_o, d.sAux_i = int(ld_aux), int(sAux[0]), int(sAux[1])
    d.res = ptr(res, res_off)
    d.res_dtype = dt(res) if res is not None else 0
    d.ld_res, d.sRes_o, d.sRes_i = int(ld_res), int(sRes[0]), int(sRes[1])
    d.accumulate = int(bool(accumulate))
    d.split_k = int(split_k)
    d.colsum = ptr(colsum)
    d.colsum_dtype = dt(colsum) if colsum is not None else 0
    d.colsum_accumulate = int(bool(colsum_accumulate))
    L = _lib.lib()
    if d.split_k > 1 or colsum is not None:
        need = L.wavlm_gemm_workspace_bytes(C.byref(d))
        ws = workspace(dev, need, "gemm")
        d.workspace, d.ws_bytes = ptr(ws), need
    else:
        d.workspace, d.ws_bytes = None, 0
    check(L.wavlm_gemm(C.byref(d), stream()), "wavlm_gemm[M=%d N=%d K=%d KB=%d tA=%d tB=%d]" % (M, N, K, KB, transA, transB))


def gemm_wgrad_grouped(items, w_dtype):
    """The weight gradients dW_i[N_i, K_i] (+)= dy_i[n, N_i]^T @ x_i[n, K_i] of several linears that share the row count
    n, as one grouped split-K launch (wavlm_gemm_grouped).  items: [(dy2d, x2d, out)] with `out` accumulated into."""
    dev = _dev(items[0][0])
    n = items[0][0].shape[0]
    tiles = sum(((dy.shape[1] + 255) // 256) * ((x.shape[1] + 255) // 256) for dy, x, _ in items)
    split = grouped_slabs(tiles, (n + 63) // 64)  # (callers that only group when one round fits: WgradGroup.fire)
    L = _lib.lib()
    descs = (GemmDesc * len(items))()
    need = []
    for d, (dy, x, out) in zip(descs, items):
        if dy.shape[0] != n or x.shape[0] != n:
            raise ValueError("grouped weight gradients need a common row count")
        _contig(dy); _contig(x); _contig(out)
        N, K = dy.shape[1], x.shape[1]
        d.dtype, d.c_dtype = dt(dy), dt(out)
        d.M, d.N, d.K, d.KB = N, K, n, 1
        d.transA, d.transB = 1, 1
        d.lda, d.ldb, d.ldc = N, K, K
        d.batch_o, d.batch_i = 1, 1
        d.A, d.B, d.C = ptr(dy), ptr(x), ptr(out)
        d.alpha, d.epi = 1.0, 0
        d.accumulate = 1
        d.split_k = split
        need.append((int(L.wavlm_gemm_workspace_bytes(C.byref(d))) + 255) // 256 * 256)
    ws = workspace(dev, sum(need), "gemm")
    off = 0
    for d, nb in zip(descs, need):
        d.workspace, d.ws_bytes = ptr(ws, off), nb
        off += nb
    check(L.wavlm_gemm_grouped(descs, len(items), stream()), "wavlm_gemm_grouped[%d]" % len(items))


def grouped_split(tiles, ktiles, grid=None):
    """split-K factor of a grouped weight-gradient launch, or 0 when the members should run as single launches.
    Measured (profiles/r03/envab_wg.txt, Base: 108 tiles x 375 K-steps per layer, same box): ONE round of tiles * split work
    items is what pays -- split 2 (216 items) 335 us per layer against 349 us for the four single launches + 28 us less slab
    reduction; split 7 (2.95 rounds) 371 us and split 14 423 us although they balance the K-steps better: every extra round
    costs a slab store and a pipeline refill per CU.  So: the largest split that still fits one round, and no grouping when
    even split 2 does not (Large: 192 tiles).  WAVLM_WGRAD_SPLIT overrides (A/B measurements)."""
    forced = os.environ.get("WAVLM_WGRAD_SPLIT")
    if forced:
        return max(2, int(forced))
    grid = grid or grid_blocks()
    s = min(grid // max(tiles, 1), ktiles // 8, 64)
    return s if s >= 2 else 0


def grid_blocks():
    """blocks of a persistent GEMM grid: one per CU minus what the data-parallel reducer keeps free for the RCCL kernels
    (wavlm_set_reserved_cus).  Every split-K choice aims at ONE round of THIS many blocks: with 8 CUs reserved, the 252 work
    items a 256-CU split produces run as a full round plus a round of four (measured: +17 % step time,
    profiles/r04/reserved_cus_base_before.txt)."""
    return 256 - get_reserved_cus()


def lab_build():
    """the loaded library is the lab build (tools/probe/build_probe.py lab: -DWAVLM_EXPERIMENTAL): it alone exports
    `wavlm_lab_build`, and it alone carries the balanced grouped launch -- asked of the library itself, so that the slab count
    chosen here always equals csrc/layer.hip: grouped_slabs of the same library"""
    return hasattr(_lib.lib(), "wavlm_lab_build")


def grouped_slabs(tiles, ktiles, grid=None):
    """`split_k` of the members of a grouped weight-gradient launch = fp32 slabs each member's workspace holds: the one-round
    split.  Lab library only (lab_build(): WAVLM_HIP_LIB = tools/probe/lib/libwavlm_hip_lab.so) with WAVLM_WGRAD_STREAMK=1: one more, so that
    the library can hand the CUs that split leaves idle (Base: 108 tiles x 2 = 216 of 256) the K tail of every tile
    (csrc/gemm_common.hpp: gemm_sk_plan) -- measured neutral (profiles/r04/ab_wgrad_balanced_*.txt: the launch is not bound by
    how many CUs take part), so libwavlm_hip.so does not carry that path."""
    grid = grid or grid_blocks()
    split = max(2, grouped_split(tiles, ktiles, grid))
    if (os.environ.get("WAVLM_WGRAD_STREAMK", "0") == "1" and lab_build() and not os.environ.get("WAVLM_WGRAD_SPLIT")
            and tiles < grid and tiles * ktiles >= 8 * grid):
        return max(split, grid // tiles + 1)
    return split


def pick_split(M, N, ktiles, nbatch=1, target_blocks=768):
    """split-K factor so that a small-output / long-reduction GEMM still fills 256 CUs.  Problems the 256 x 256
    ping-pong kernel takes (one block per CU) aim at one full round of 256 blocks; the 128-wide kernel (two to three
    blocks per CU) at `target_blocks`."""
    if M >= 256 and N >= 256:
        tiles = ((M + 255) // 256) * ((N + 255) // 256) * nbatch
        s = max(1, grid_blocks() // tiles)
        return max(1, min(s, ktiles // 8, 64))
    tiles = ((M + 127) // 128) * ((N + 127) // 128) * nbatch
    s = max(1, min(ktiles, (target_blocks + tiles - 1) // tiles))
    return min(s, 64)


# ---------------------------------------------------------------------------------------- row kernels
def layernorm_fwd(x, r, gamma, beta, eps, *, act=0, p_in=0.0, seed_in=0, p_out=0.0, seed_out=0, save=True):
    """returns (y, s, mean, rstd); s is the pre-norm sum (== x when r is None)"""
    dev = _dev(x)
    _contig(x); _contig(r)
    D = x.shape[-1]
    rows = x.numel() // D
    y = torch.empty_like(x)
    s = None
    mean = rstd = None
    if save:
        mean = torch.empty(rows, dtype=torch.float32, device=dev)
        rstd = torch.empty(rows, dtype=torch.float32, device=dev)
        s = torch.empty_like(x) if r is not None else x
    check(_lib.lib().wavlm_layernorm_fwd(ptr(x), ptr(r), ptr(y), ptr(s) if (save and r is not None) else None,
                                         ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), rows, D, float(eps), dt(x),
                                         dt(gamma), int(act), float(p_in), int(seed_in), float(p_out), int(seed_out),
                                         stream()), "wavlm_layernorm_fwd")
    return y, s, mean, rstd


# WAVLM_LN_SEG=0: LayerNorm backward never writes the padded conv-gradient layout itself (A/B: the padded copy comes back);
# WAVLM_LN_FULL=0 (the general kernels) has no segmented form either
LN_SEG_OK = os.environ.get("WAVLM_LN_SEG", "1") != "0" and os.environ.get("WAVLM_LN_FULL", "1") != "0"


def layernorm_bwd(dy, s, mean, rstd, gamma, beta, *, act=0, p_in=0.0, seed_in=0, p_out=0.0, seed_out=0,
                  grad_scale=1.0, need_dr=False, dgamma=None, dbeta=None, dr_colsum=None, dx_add=None, dr_incl_add=False,
                  dx_pad=None):
    """returns (dx, dr, dgamma, dbeta, dr_colsum); given dgamma / dbeta (/ dr_colsum) tensors are accumulated into (+=).
    dr_colsum: True -> also return the column sums of dr (fresh tensor); a tensor -> accumulate into it (only together
    with dgamma / dbeta tensors: the three share the accumulate flag).  dx_add: added into dx (not into dr).
    dx_pad = (fp, bp) with dy of shape [B, T, D]: dx is written into a zero-padded [B, fp + T + bp, D] buffer (the layout the
    data-gradient GEMMs of the conv layer in front of this LayerNorm read); returned dx = the [B, T, D] interior VIEW of it,
    whose `_padded` attribute is the whole buffer."""
    dev = _dev(dy)
    _contig(dy); _contig(s)
    if dx_add is not None:
        _contig(dx_add)
        if dx_add.shape != dy.shape or dx_add.dtype != dy.dtype:
            raise ValueError("dx_add must match dy in shape and dtype")
    D = dy.shape[-1]
    rows = dy.numel() // D
    seg_rows = seg_gap = 0
    padded = None
    if dx_pad is not None and (dx_pad[0] or dx_pad[1]):
        fp, bp = dx_pad
        Bn, Tn, _ = dy.shape
        Tp = fp + Tn + bp
        # all pads are the B + 1 equally spaced gaps of one allocation with bp spare rows in front and fp behind (see
        # ConvStackFn.backward): one strided fill
        big = torch.empty((Bn * Tp + fp + bp, D), dtype=dy.dtype, device=dev)
        big.as_strided((Bn + 1, fp + bp, D), (Tp * D, D, 1)).zero_()
        padded = big[bp:bp + Bn * Tp].view(Bn, Tp, D)
        dx = padded[:, fp:fp + Tn]
        seg_rows, seg_gap = Tn, fp + bp
    else:
        dx = torch.empty_like(dy)
    dr = torch.empty_like(dy) if need_dr else None
    acc = dgamma is not None and dbeta is not None
    if not acc:
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(beta)
    if dr_colsum is True or (dr_colsum is not None and not acc):
        if torch.is_tensor(dr_colsum):
            raise ValueError("dr_colsum sink needs dgamma / dbeta sinks")
        dr_colsum = torch.empty_like(gamma) if not acc else torch.zeros_like(gamma)
    L = _lib.lib()
    need = L.wavlm_layernorm_bwd_workspace_bytes(D)
    ws = workspace(dev, need)
    check(L.wavlm_layernorm_bwd_seg(ptr(dy), ptr(s), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(dx), ptr(dr),
                                    ptr(dx_add), ptr(dgamma), ptr(dbeta), ptr(dr_colsum), rows, D, dt(dy), dt(gamma), int(act),
                                    float(p_in), int(seed_in), float(p_out), int(seed_out), float(grad_scale), int(acc),
                                    int(bool(dr_incl_add)), seg_rows, seg_gap, ptr(ws), need, stream()), "wavlm_layernorm_bwd_seg")
    if padded is not None:
        dx._padded = padded
    return dx, dr, dgamma, dbeta, dr_colsum


def colsum(x2d, out_dtype, *, include=None, exclude=None, rows=None, N=None, ld=None, out=None, accumulate=False):
    """column sums of a [rows, N] matrix (row stride ld) -> [N] tensor of out_dtype (or (+)= into `out`)"""
    dev = _dev(x2d)
    if rows is None:
        N = x2d.shape[-1]
        rows = x2d.numel() // N
        ld = N
        _contig(x2d)
    if out is None:
        out = torch.empty(N, dtype=out_dtype, device=dev)
        accumulate = False
    L = _lib.lib()
    need = L.wavlm_colsum_workspace_bytes(N)
    ws = workspace(dev, need)
    check(L.wavlm_colsum(ptr(x2d), rows, N, ld, dt(x2d), ptr(include), ptr(exclude), ptr(out), dt(out),
                         int(bool(accumulate)), ptr(ws), need, stream()), "wavlm_colsum")
    return out


def select_rows(x, sel, emb, zero):
    """y[row] = zero[row] ? 0 : sel[row] ? emb : x[row]   (sel / zero: uint8 [rows] or None; emb None -> 0)"""
    _dev(x); _contig(x)
    D = x.shape[-1]
    rows = x.numel() // D
    y = torch.empty_like(x)
    check(_lib.lib().wavlm_select_rows(ptr(x), ptr(y), ptr(sel), ptr(emb), ptr(zero), rows, D, dt(x),
                                       dt(emb) if emb is not None else dt(x), stream()), "wavlm_select_rows")
    return y


def gather_rows(src2d, idx, n_out):
    """dst[i] = src[idx[i]] (idx int32; -1 -> zero row)"""
    dev = _dev(src2d); _contig(src2d)
    D = src2d.shape[-1]
    dst = torch.empty((n_out, D), dtype=src2d.dtype, device=dev)
    check(_lib.lib().wavlm_gather_rows(ptr(src2d), ptr(idx), ptr(dst), n_out, D, dt(src2d), stream()),
          "wavlm_gather_rows")
    return dst


def _conv_desc(weights, specs):
    d = ConvRelayoutDesc()
    if len(weights) > 8:
        raise ValueError("at most 8 convolution layers per call")
    d.n_layers, d.dtype = len(weights), dt(weights[0])
    for l, (W, (k, s_)) in enumerate(zip(weights, specs)):
        _dev(W); _contig(W)
        Cout, Cin, kk = W.shape
        if kk != k:
            raise ValueError("kernel width of layer %d does not match its spec" % l)
        d.Cout[l], d.Cin[l], d.k[l], d.s[l] = Cout, Cin, k, s_
        d.W[l] = W.data_ptr()
    return d


def conv_weights_relayout(weights, specs, want_bwd):
    """GEMM operand images of the extractor's Conv1d weights, all layers in one launch.  weights: [Cout, Cin, k] tensors,
    specs: [(k, stride)].  Returns (Wf list [Cout, k * Cin], Wb flat buffers [Cin * Cout * k] or None): the stride-phase
    images of layer l are views of its flat buffer (conv_phase_views)."""
    d = _conv_desc(weights, specs)
    Wf, Wb = [], []
    for l, (W, (k, s_)) in enumerate(zip(weights, specs)):
        Cout, Cin, _ = W.shape
        f = torch.empty((Cout, k * Cin), dtype=W.dtype, device=W.device)
        d.Wf[l] = f.data_ptr()
        Wf.append(f)
        if want_bwd:
            flat = torch.empty(Cin * Cout * k, dtype=W.dtype, device=W.device)
            d.Wb[l] = flat.data_ptr()
            Wb.append(flat)
        else:
            d.Wb[l] = None
    check(_lib.lib().wavlm_conv_weights_relayout(C.byref(d), stream()), "wavlm_conv_weights_relayout")
    return Wf, (Wb if want_bwd else None)


def conv_phase_views(flat, Cout, Cin, k, s_):
    """the s_ stride-phase images [Cin, J_r * Cout] (taps of phase r newest first) inside a layer's flat Wb buffer"""
    views, off = [], 0
    for r in range(s_):
        Jr = len(range(r, k, s_))
        views.append(flat[off:off + Cin * Jr * Cout].view(Cin, Jr * Cout))
        off += Cin * Jr * Cout
    return views


def conv_wgrad_scatter(grads, dWf, specs, accumulate):
    """grads[l][co][ci][kk] (+)= dWf[l][co][kk * Cin + ci] for all layers in one launch"""
    d = _conv_desc(grads, specs)
    for l, f in enumerate(dWf):
        _contig(f)
        d.Wf[l] = f.data_ptr()
        d.Wb[l] = None
    check(_lib.lib().wavlm_conv_wgrad_scatter(C.byref(d), int(bool(accumulate)), stream()), "wavlm_conv_wgrad_scatter")


def axpby_(y, x, a, b):
    """y = a*x + b*y in place"""
    _dev(y); _contig(y); _contig(x)
    check(_lib.lib().wavlm_axpby(ptr(x), dt(x), ptr(y), dt(y), y.numel(), float(a), float(b), stream()), "wavlm_axpby")
    return y


def scale_dev_(y, scalar, extra=1.0):
    """y *= scalar[0] * extra with `scalar` a 1-element fp32 device tensor"""
    _dev(y); _contig(y)
    check(_lib.lib().wavlm_scale_dev(ptr(y), dt(y), y.numel(), ptr(scalar), float(extra), stream()), "wavlm_scale_dev")
    return y


def dropout(x, p, seed):
    _dev(x); _contig(x)
    y = torch.empty_like(x)
    check(_lib.lib().wavlm_dropout(ptr(x), ptr(y), x.numel(), float(p), int(seed), dt(x), stream()), "wavlm_dropout")
    return y


def dropout_add(x, r, p, seed):
    """x + dropout(r, p, seed) in one pass (p = 0: plain add)"""
    _dev(x); _contig(x); _contig(r)
    y = torch.empty_like(x)
    check(_lib.lib().wavlm_dropout_add(ptr(x), ptr(r), ptr(y), x.numel(), float(p), int(seed), dt(x), stream()),
          "wavlm_dropout_add")
    return y


def sumsq(x, scale=1.0, out=None):
    """scale * sum(x^2) as a 1-element fp32 device tensor (no host sync)"""
    dev = _dev(x); _contig(x)
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=dev)
    L = _lib.lib()
    need = L.wavlm_sumsq_workspace_bytes()
    ws = workspace(dev, need, "reduce")
    check(L.wavlm_sumsq(ptr(x), dt(x), x.numel(), float(scale), ptr(out), ptr(ws), need, stream()), "wavlm_sumsq")
    return out


def sum_f32(x, out=None):
    dev = _dev(x); _contig(x)
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=dev)
    L = _lib.lib()
    need = L.wavlm_sum_workspace_bytes()
    ws = workspace(dev, need, "reduce")
    check(L.wavlm_sum_f32(ptr(x), x.numel(), ptr(out), ptr(ws), need, stream()), "wavlm_sum_f32")
    return out


# -------------------------------------------------------------------------------------------- conv0
def conv0_gn_gelu_fwd(wav, W, gamma, beta, stride, eps, out_dtype):
    dev = _dev(wav); _contig(wav); _contig(W)
    B, T = wav.shape
    Cc, _, kw = W.shape
    T0 = (T - kw) // stride + 1
    out = torch.empty((B, T0, Cc), dtype=out_dtype, device=dev)
    stats = torch.empty((B, Cc, 2), dtype=torch.float32, device=dev)
    L = _lib.lib()
    need = L.wavlm_conv0_gn_workspace_bytes(B, T, Cc, stride)
    ws = workspace(dev, need)
    check(L.wavlm_conv0_gn_gelu_fwd(ptr(wav), dt(wav), ptr(W), ptr(gamma), ptr(beta), dt(W), ptr(out), dt(out),
                                    ptr(stats), B, T, Cc, kw, stride, float(eps), ptr(ws), need, stream()),
          "wavlm_conv0_gn_gelu_fwd")
    return out, stats


def conv0_gn_gelu_bwd(wav, W, gamma, beta, g, stats, stride, gscale=1.0):
    dev = _dev(wav); _contig(g)
    B, T = wav.shape
    Cc, _, kw = W.shape
    dW = torch.empty_like(W)
    dgamma = torch.empty_like(gamma)
    dbeta = torch.empty_like(beta)
    L = _lib.lib()
    need = L.wavlm_conv0_gn_bwd_workspace_bytes(B, T, Cc, stride)
    ws = workspace(dev, need)
    check(L.wavlm_conv0_gn_gelu_bwd(ptr(wav), dt(wav), ptr(W), ptr(gamma), ptr(beta), dt(W), ptr(g), dt(g),
                                    ptr(stats), ptr(dW), ptr(dgamma), ptr(dbeta), B, T, Cc, kw, stride,
                                    float(gscale), ptr(ws), need, stream()), "wavlm_conv0_gn_gelu_bwd")
    return dW, dgamma, dbeta


def conv0_ln_gelu_fwd(wav, W, gamma, beta, stride, eps, out_dtype, bias=None):
    """extractor_mode 'layer_norm', block 0: gelu(LayerNorm_C(conv0(wav) + bias)) -> [B, T0, C]"""
    dev = _dev(wav); _contig(wav); _contig(W)
    B, T = wav.shape
    Cc, _, kw = W.shape
    T0 = (T - kw) // stride + 1
    out = torch.empty((B, T0, Cc), dtype=out_dtype, device=dev)
    L = _lib.lib()
    need = L.wavlm_conv0_ln_fwd_workspace_bytes()
    ws = workspace(dev, need)
    check(L.wavlm_conv0_ln_gelu_fwd(ptr(wav), dt(wav), ptr(W), ptr(bias), ptr(gamma), ptr(beta), dt(W), ptr(out), dt(out), B, T,
                                    Cc, kw, stride, float(eps), ptr(ws), need, stream()), "wavlm_conv0_ln_gelu_fwd")
    return out


def conv0_ln_gelu_bwd(wav, W, gamma, beta, g, stride, eps, gscale=1.0, bias=None):
    """returns (dW, dgamma, dbeta, dbias); dbias is None without a conv bias"""
    dev = _dev(wav); _contig(g)
    B, T = wav.shape
    Cc, _, kw = W.shape
    dW = torch.empty_like(W)
    dgamma = torch.empty_like(gamma)
    dbeta = torch.empty_like(beta)
    dbias = torch.empty_like(bias) if bias is not None else None
    L = _lib.lib()
    need = L.wavlm_conv0_ln_bwd_workspace_bytes(B, T, Cc, stride)
    ws = workspace(dev, need)
    check(L.wavlm_conv0_ln_gelu_bwd(ptr(wav), dt(wav), ptr(W), ptr(bias), ptr(gamma), ptr(beta), dt(W), ptr(g), dt(g),
                                    ptr(dW), ptr(dbias), ptr(dgamma), ptr(dbeta), B, T, Cc, kw, stride, float(eps),
                                    float(gscale), ptr(ws), need, stream()), "wavlm_conv0_ln_gelu_bwd")
    return dW, dgamma, dbeta, dbias


# ---------------------------------------------------------------------------------------- attention
def relpos_gather(emb, bucket, H, L_):
    dev = _dev(emb)
    tab = torch.empty((H, L_), dtype=torch.float32, device=dev)
    check(_lib.lib().wavlm_relpos_gather(ptr(emb), dt(emb), ptr(bucket), ptr(tab), H, L_, stream()),
          "wavlm_relpos_gather")
    return tab


def relpos_scatter(dtab, bucket, like_emb):
    H, L_ = dtab.shape
    demb = torch.empty_like(like_emb)
    check(_lib.lib().wavlm_relpos_scatter(ptr(dtab), ptr(bucket), ptr(demb), dt(demb), H, L_, like_emb.shape[0],
                                          stream()), "wavlm_relpos_scatter")
    return demb


def gate_fwd(x, W, bias, grep_a, H):
    dev = _dev(x); _contig(x)
    B, T, D = x.shape
    hd = D // H
    gate = torch.empty((B, H, T), dtype=torch.float32, device=dev)
    ga = torch.empty_like(gate)
    gb = torch.empty_like(gate)
    check(_lib.lib().wavlm_gate_fwd(ptr(x), ptr(W), ptr(bias), ptr(grep_a), ptr(gate), ptr(ga), ptr(gb), B, T, H, hd,
                                    dt(x), dt(W), stream()), "wavlm_gate_fwd")
    return gate, ga, gb


def gate_bwd(dgate, x, W, bias, grep_a, ga, gb, H, sinks=None, dx_accumulate=None):
    """returns (dx, dW, dbias, da); sinks = (dW, dbias, da) gradient-sink views -> accumulated in place, returned as None;
    dx_accumulate: a [B, T, D] tensor already holding another consumer's gradient of x -> the gate's is added into it"""
    dev = _dev(x)
    B, T, D = x.shape
    hd = D // H
    dx = dx_accumulate if dx_accumulate is not None else torch.empty_like(x)
    _contig(dx)
    if sinks is not None:
        dW, dbias, da = sinks
    else:
        dW = torch.empty_like(W)
        dbias = torch.empty_like(bias)
        da = torch.empty_like(grep_a)
    L = _lib.lib()
    need = L.wavlm_gate_bwd_workspace_bytes(H, hd)
    ws = workspace(dev, need)
    check(L.wavlm_gate_bwd(ptr(dgate), ptr(x), ptr(W), ptr(grep_a), ptr(ga), ptr(gb), ptr(dx), ptr(dW), ptr(dbias),
                           ptr(da), B, T, H, hd, dt(x), dt(W), int(sinks is not None) | (2 if dx_accumulate is not None else 0),
                           ptr(ws), need, stream()),
          "wavlm_gate_bwd")
    if sinks is not None:
        return dx, None, None, None
    return dx, dW, dbias, da


def attn_softmax_fwd(S, P, lse, gate, tab, kpm, B, H, T, ldS, ldP, p_drop, seed):
    check(_lib.lib().wavlm_attn_softmax_fwd(ptr(S), ptr(P), ptr(lse), ptr(gate), ptr(tab), ptr(kpm), B, H, T, ldS, ldP,
                                            dt(S), dt(P), float(p_drop), int(seed), stream()),
          "wavlm_attn_softmax_fwd")


def attn_softmax_bwd(S, dP, lse, gate, tab, kpm, dS, dgate, dtab, B, H, T, ldS, ldP, p_drop, seed):
    dev = _dev(S)
    L = _lib.lib()
    need = L.wavlm_attn_softmax_bwd_workspace_bytes(B, H, T) if tab is not None else 0
    ws = workspace(dev, need) if need else None
    check(L.wavlm_attn_softmax_bwd(ptr(S), ptr(dP), ptr(lse), ptr(gate), ptr(tab), ptr(kpm), ptr(dS), ptr(dgate),
                                   ptr(dtab), 0, B, H, T, ldS, ldP, dt(S), dt(dS), float(p_drop), int(seed), ptr(ws),
                                   need, stream()), "wavlm_attn_softmax_bwd")


def attn_fused_fwd(qkv, gate, tab, kpm, H, scale, p_drop, seed, store_p=False):
    """fused bf16 attention forward (head_dim 64): returns (O [B,T,D], lse [B*H,T], pstore).  store_p True: the forward also
    writes its probabilities into `pstore` (opaque uint8 buffer) for attn_fused_bwd; "bits": only its dropout decisions (one
    bit per element; nothing without dropout); False: pstore is None"""
    dev = _dev(qkv); _contig(qkv)
    B, T, D3 = qkv.shape
    D = D3 // 3
    O = torch.empty((B, T, D), dtype=qkv.dtype, device=dev)
    lse = torch.empty((B * H, T), dtype=torch.float32, device=dev)
    L = _lib.lib()
    pstore, nps = None, 0
    if store_p == "bits":
        if p_drop > 0:
            nps = int(L.wavlm_attn_fused_dbits_bytes(B, H, T))
            pstore = torch.empty(nps, dtype=torch.uint8, device=dev)
    elif store_p:
        nps = int(L.wavlm_attn_fused_pstore_bytes(B, H, T))   # 0: this T is not supported by the stored form -> recompute
        pstore = torch.empty(nps, dtype=torch.uint8, device=dev) if nps else None
    check(L.wavlm_attn_fused_fwd_p(ptr(qkv), ptr(O), ptr(lse), ptr(gate), ptr(tab), ptr(kpm), ptr(pstore), nps, B, H, T, D // H,
                                   float(scale), float(p_drop), int(seed), stream()), "wavlm_attn_fused_fwd_p")
    return O, lse, pstore


def attn_fused_bwd(qkv, O, dO, lse, gate, tab, kpm, H, scale, p_drop, seed, dbias=None, dbias_accumulate=False, pstore=None):
    """returns (dqkv, dgate, dtab); dbias [3D] (optional, any float dtype): (+)= column sums of dqkv (q|k|v bias gradient);
    pstore: what attn_fused_fwd(store_p=True) returned for the same arguments (None: probabilities are recomputed)"""
    dev = _dev(qkv); _contig(dO)
    B, T, D3 = qkv.shape
    D = D3 // 3
    dqkv = torch.empty_like(qkv)
    dgate = dtab = None
    if tab is not None:
        dgate = torch.empty((B, H, T), dtype=torch.float32, device=dev)
        dtab = torch.empty_like(tab)
    L = _lib.lib()
    need = L.wavlm_attn_fused_bwd_workspace_bytes(B, H, T)
    ws = workspace(dev, need, "attn")
    check(L.wavlm_attn_fused_bwd_p(ptr(qkv), ptr(O), ptr(dO), ptr(lse), ptr(gate), ptr(tab), ptr(kpm), ptr(pstore),
                                   pstore.numel() if pstore is not None else 0, ptr(dqkv),
                                   ptr(dgate), ptr(dtab), ptr(dbias), dt(dbias) if dbias is not None else 0,
                                   int(bool(dbias_accumulate)), B, H, T, D // H, float(scale), float(p_drop), int(seed),
                                   ptr(ws), need, stream()), "wavlm_attn_fused_bwd_p")
    return dqkv, dgate, dtab


# ------------------------------------------------------------------------------------------ pos_conv
def posconv_weight_fwd(v, g, out_dtype, layout=0):
    """layout 0: operand images of the GEMM form; 1: of the direct convolution (posconv_direct)"""
    dev = _dev(v); _contig(v)
    D, Cg, K = v.shape
    G = D // Cg
    Wf = torch.empty((G, Cg, K * Cg), dtype=out_dtype, device=dev)
    Wb = torch.empty((G, Cg, K * Cg), dtype=out_dtype, device=dev)
    norm = torch.empty(K, dtype=torch.float32, device=dev)
    L = _lib.lib()
    need = L.wavlm_posconv_weight_workspace_bytes(D, Cg, K)
    ws = workspace(dev, need)
    check(L.wavlm_posconv_weight_fwd(ptr(v), ptr(g), dt(v), ptr(Wf), ptr(Wb), dt(Wf), ptr(norm), D, Cg, K, int(layout),
                                     ptr(ws), need, stream()), "wavlm_posconv_weight_fwd")
    return Wf, Wb, norm


def posconv_weight_bwd(dWf, v, g, norm, nsplit=1):
    """dWf: fp32 [nsplit, G, Cg, K*Cg] (slabs are summed)"""
    dev = _dev(v)
    D, Cg, K = v.shape
    dv = torch.empty_like(v)
    dg = torch.empty_like(g)
    L = _lib.lib()
    need = L.wavlm_posconv_weight_workspace_bytes(D, Cg, K)
    ws = workspace(dev, need)
    check(L.wavlm_posconv_weight_bwd(ptr(dWf), ptr(v), ptr(g), ptr(norm), dt(v), ptr(dv), ptr(dg), D, Cg, K, int(nsplit),
                                     ptr(ws), need, stream()), "wavlm_posconv_weight_bwd")
    return dv, dg


def group_major(x, aux, G, left_pad, Tp, want_nat=False, aux_is_grad=False):
    """x[B,T,D] (* gelu'(aux), or * aux if it already holds the derivative) -> [B,G,Tp,D/G], zero rows outside
    [left_pad, left_pad+T)"""
    dev = _dev(x); _contig(x); _contig(aux)
    B, T, D = x.shape
    out = torch.empty((B, G, Tp, D // G), dtype=x.dtype, device=dev)
    nat = torch.empty_like(x) if want_nat else None
    check(_lib.lib().wavlm_posconv_group_major(ptr(x), ptr(aux), ptr(out), ptr(nat), B, T, D, G, left_pad, Tp, dt(x),
                                               int(bool(aux_is_grad)), stream()), "wavlm_posconv_group_major")
    return out, nat


def posconv_dw_direct(xg, dug, du_off, T, K):
    """weight gradient of the grouped convolution from the two group-major copies -> (fp32 slabs [S, G, Cg, K*Cg], S)"""
    dev = _dev(xg); _contig(xg); _contig(dug)
    B, G, Tp, Cg = xg.shape
    L = _lib.lib()
    S = L.wavlm_posconv_dw_direct_splits(Cg, G)
    if S <= 0:
        raise ValueError("posconv_dw_direct: shape not covered")
    part = torch.empty((S, G, Cg, K * Cg), dtype=torch.float32, device=dev)
    check(L.wavlm_posconv_dw_direct(ptr(xg), ptr(dug), ptr(part), B, G, T, Tp, int(du_off), Cg, K, stream()),
          "wavlm_posconv_dw_direct")
    return part, S


def posconv_direct_supported(x_dtype, Cg, K, T):
    return x_dtype == torch.bfloat16 and bool(_lib.lib().wavlm_posconv_direct_supported(int(Cg), int(K), int(T)))


def posconv_direct(xg, W, out, T, K, *, bias=None, res=None, aux=None, gelu=False):
    """grouped convolution over the group-major, time-padded copy xg [B, G, Tp, Cg] with the weight image W [G, Cg, K*Cg]:
    out[B, T, G*Cg] = res + f(conv + bias), f = GELU (aux receives the pre-activation) or identity"""
    _dev(xg); _contig(xg); _contig(W); _contig(out); _contig(res); _contig(aux)
    B, G, Tp, Cg = xg.shape
    check(_lib.lib().wavlm_posconv_direct(ptr(xg), ptr(W), ptr(bias), ptr(res), ptr(out), ptr(aux), B, G, T, Tp, Cg, K,
                                          int(bool(gelu)), stream()), "wavlm_posconv_direct")
    return out


# ---------------------------------------------------------------------------------------------- loss
def l2norm_fwd(x, out_dtype, eps=1e-8):
    dev = _dev(x); _contig(x)
    rows, D = x.shape
    y = torch.empty((rows, D), dtype=out_dtype, device=dev)
    inv = torch.empty(rows, dtype=torch.float32, device=dev)
    check(_lib.lib().wavlm_l2norm_fwd(ptr(x), dt(x), ptr(y), dt(y), ptr(inv), rows, D, float(eps), stream()),
          "wavlm_l2norm_fwd")
    return y, inv


def l2norm_bwd(dy, y, inv, x_dtype):
    dev = _dev(dy); _contig(dy)
    rows, D = y.shape
    dx = torch.empty((rows, D), dtype=x_dtype, device=dev)
    check(_lib.lib().wavlm_l2norm_bwd(ptr(dy), ptr(y), dt(y), ptr(inv), ptr(dx), dt(dx), rows, D, stream()),
          "wavlm_l2norm_bwd")
    return dx


def ce_rows(logits, target, V, ld_logits, dlogits, ld_dlogits, weight):
    dev = _dev(logits)
    S = target.numel()
    loss_rows = torch.empty(max(S, 1), dtype=torch.float32, device=dev)
    correct_rows = torch.empty(max(S, 1), dtype=torch.float32, device=dev)
    check(_lib.lib().wavlm_ce_rows(ptr(logits), ptr(target), ptr(loss_rows), ptr(correct_rows), ptr(dlogits),
                                   dt(dlogits) if dlogits is not None else 0, S, V, ld_logits, ld_dlogits,
                                   float(weight), stream()), "wavlm_ce_rows")
    return loss_rows[:S], correct_rows[:S]


# --------------------------------------------------------------------------------------------- optim
def adam_step(p32, m, v, grad, p_lowp, *, lr, beta1, beta2, eps, weight_decay, step, grad_mult=1.0,
              grad_mult_dev=None, gnorm_sq=None, max_norm=0.0):
    _dev(p32)
    check(_lib.lib().wavlm_adam_step(ptr(p32), ptr(m), ptr(v), ptr(grad), dt(grad), ptr(p_lowp),
                                     dt(p_lowp) if p_lowp is not None else 0, p32.numel(), float(lr), float(beta1),
                                     float(beta2), float(eps), float(weight_decay), int(step), float(grad_mult),
                                     ptr(grad_mult_dev), ptr(gnorm_sq), float(max_norm), stream()), "wavlm_adam_step")


def gemm_set_variant(v):
    """0 auto | 1 force the 128x128 tile | 2 force the 256x128 tile (tests / A-B measurements)"""
    _lib.lib().wavlm_gemm_set_variant(int(v))


def set_reserved_cus(n):
    """leave n CUs out of every persistent GEMM grid (data-parallel runs: room for the RCCL kernels)"""
    _lib.lib().wavlm_set_reserved_cus(int(n))


def get_reserved_cus():
    return int(_lib.lib().wavlm_get_reserved_cus())


def prof_enable(on):
    _lib.lib().wavlm_prof_enable(1 if on else 0)


def prof_collect(dtype=-1):
    """(launches, total_ms, total_flops) of the GEMM launches recorded since prof_enable(True); synchronises"""
    ms, fl = C.c_double(0.0), C.c_double(0.0)
    n = _lib.lib().wavlm_prof_collect(int(dtype), C.byref(ms), C.byref(fl))
    return n, ms.value, fl.value


PROF_CLASSES = {"gemm": 0, "attn_fwd": 1, "attn_bwd": 2, "conv0_fwd": 3, "conv0_bwd": 4, "ln_fwd": 5, "ln_bwd": 6}


def prof_collect_class(name):
    """(calls, total_ms, algorithmic flops, algorithmic bytes) of one kernel class (PROF_CLASSES) recorded since
    prof_enable(True); synchronises"""
    ms, fl, by = C.c_double(0.0), C.c_double(0.0), C.c_double(0.0)
    n = _lib.lib().wavlm_prof_collect_class(PROF_CLASSES[name], C.byref(ms), C.byref(fl), C.byref(by))
    return n, ms.value, fl.value, by.value


def prof_collect_bytes(dtype=-1):
    """algorithmic HBM bytes (each operand / output / epilogue tensor once) of the launches prof_collect reports"""
    return float(_lib.lib().wavlm_prof_collect_bytes(int(dtype)))


# ------------------------------------------------------------------------- sampled-instance cosine head
GLU_GATES = {"sigmoid": 0, "swish": 1, "relu": 2, "gelu": 3, "bilinear": 4}
ACT_KINDS = {"relu": 1, "gelu_accurate": 2, "gelu_fast": 2, "tanh": 3, "gelu": 4}


def glu_fwd(x2d, gate="sigmoid"):
    """[rows, 2F] -> [rows, F]: x[:, :F] * g(x[:, F:]); gate "sigmoid" = nn.GLU(dim=-1), "swish" = GLU_Linear's"""
    _dev(x2d); _contig(x2d)
    rows, F2 = x2d.shape
    y = torch.empty((rows, F2 // 2), dtype=x2d.dtype, device=x2d.device)
    check(_lib.lib().wavlm_glu_fwd(ptr(x2d), ptr(y), rows, F2 // 2, dt(x2d), GLU_GATES[gate], stream()), "wavlm_glu_fwd")
    return y


def glu_bwd(x2d, dy, gate="sigmoid"):
    _dev(x2d); _contig(x2d); _contig(dy)
    rows, F2 = x2d.shape
    dx = torch.empty_like(x2d)
    check(_lib.lib().wavlm_glu_bwd(ptr(x2d), ptr(dy), ptr(dx), rows, F2 // 2, dt(x2d), GLU_GATES[gate], stream()), "wavlm_glu_bwd")
    return dx


def act_fwd(x, kind):
    """elementwise activation (ACT_KINDS) of a contiguous tensor"""
    _dev(x); _contig(x)
    y = torch.empty_like(x)
    check(_lib.lib().wavlm_act_fwd(ptr(x), ptr(y), x.numel(), dt(x), ACT_KINDS[kind], stream()), "wavlm_act_fwd")
    return y


def act_bwd(x, dy, kind):
    """dx = dy * act'(x), x the pre-activation"""
    _dev(x); _contig(x); _contig(dy)
    dx = torch.empty_like(x)
    check(_lib.lib().wavlm_act_bwd(ptr(x), ptr(dy), ptr(dx), x.numel(), dt(x), ACT_KINDS[kind], stream()), "wavlm_act_bwd")
    return dx


def gather_dot(X, Y, idx, scale, mask_equal=False):
    """out[s, n] = scale * <X[s], Y[idx[s, n]]>  (rows [., D], idx: int32 [S, N]); mask_equal: columns n >= 1 whose
    gathered row equals the row of column 0 become -inf"""
    dev = _dev(Y); _contig(X); _contig(Y); _contig(idx)
    S, N = idx.shape
    out = torch.empty((S, N), dtype=torch.float32, device=dev)
    check(_lib.lib().wavlm_gather_dot(ptr(X), ptr(Y), dt(Y), ptr(idx), ptr(out), S, N, Y.shape[1], float(scale),
                                      int(bool(mask_equal)), stream()), "wavlm_gather_dot")
    return out


def rows_wsum(Y, src, w, off, rows, out=None, accumulate=False):
    """out[j] (+)= sum_{e in [off[j], off[j+1])} w[e] * Y[src[e]]; out has Y's dtype"""
    dev = _dev(Y); _contig(Y); _contig(src); _contig(w); _contig(off)
    if out is None:
        out = torch.empty((rows, Y.shape[1]), dtype=Y.dtype, device=dev)
        accumulate = False
    check(_lib.lib().wavlm_rows_wsum(ptr(Y), dt(Y), ptr(src), ptr(w), ptr(off), ptr(out), dt(out), rows, Y.shape[1],
                                     int(bool(accumulate)), stream()), "wavlm_rows_wsum")
    return out


def bce_logits(logits, targets_u8, gscale, want_grad=True):
    """(out[2] = (mean BCE-with-logits, accuracy of logit >= 0), dlogits = gscale * (sigmoid - target) or None)"""
    dev = _dev(logits); _contig(logits); _contig(targets_u8)
    n = logits.numel()
    out = torch.empty(2, dtype=torch.float32, device=dev)
    dl = torch.empty_like(logits) if want_grad else None
    L = _lib.lib()
    need = L.wavlm_bce_workspace_bytes()
    ws = workspace(dev, need)
    check(L.wavlm_bce_logits(ptr(logits), ptr(targets_u8), ptr(dl), ptr(out), n, float(gscale), ptr(ws), need, stream()),
          "wavlm_bce_logits")
    return out, dl


# ------------------------------------------------------------------------------- Gumbel vector quantiser
def gumbel_vq_fwd(logits, G, V, tau, training, noise=None, seed=0):
    """logits [n, G*V] -> (idx int32 [n*G] (global code index g*V + k), ysoft fp32 [n*G, V] or None, out[2] =
    (prob_perplexity, code_perplexity), dA [G*V] = d prob_perplexity / d avg_probs)"""
    dev = _dev(logits); _contig(logits); _contig(noise)
    n = logits.shape[0]
    L = _lib.lib()
    rows = int(L.wavlm_gumbel_vq_partial_rows(n))
    part = torch.empty((rows, 2 * G * V), dtype=torch.float32, device=dev)
    idx = torch.empty(n * G, dtype=torch.int32, device=dev)
    ysoft = torch.empty((n * G, V), dtype=torch.float32, device=dev) if training else None
    check(L.wavlm_gumbel_vq_fwd(ptr(logits), dt(logits), ptr(noise), int(seed) & 0xFFFFFFFFFFFFFFFF, float(tau),
                                int(bool(training)), n, int(G), int(V), ptr(ysoft), ptr(idx), ptr(part), stream()),
          "wavlm_gumbel_vq_fwd")
    sums = colsum(part, torch.float32)
    out = torch.empty(2, dtype=torch.float32, device=dev)
    dA = torch.empty(G * V, dtype=torch.float32, device=dev)
    check(L.wavlm_vq_perplexity(ptr(sums), n, int(G), int(V), ptr(out), ptr(dA), stream()), "wavlm_vq_perplexity")
    return idx, ysoft, out, dA


def gumbel_vq_bwd(logits, ysoft, dret, dA, dppl, G, V, tau):
    """dlogits [n, G*V] (dtype of logits); ysoft / dret None in eval mode, dppl None when the perplexity carries no gradient"""
    _dev(logits); _contig(dret)
    n = logits.shape[0]
    dl = torch.empty_like(logits)
    check(_lib.lib().wavlm_gumbel_vq_bwd(ptr(logits), dt(logits), ptr(ysoft), ptr(dret), ptr(dA), ptr(dppl), float(tau), n,
                                         int(G), int(V), ptr(dl), stream()), "wavlm_gumbel_vq_bwd")
    return dl


# ------------------------------------------------------------------------------------- utterance mixing
def mix_utterances(src, ops_flat, n_ops, op_begin, noise, normalize, out_dtype=torch.float32, eps=1e-5):
    """src fp32 [B, T] (left untouched) -> mixed batch [B, T] in out_dtype (fp32, or bf16 = the Trainer's waveform cast);
    ops_flat int32 [n_ops * 8], op_begin int32 [B + 1], noise fp32 or None (see include/wavlm_hip.h)"""
    dev = _dev(src); _contig(src)
    if src.dtype != torch.float32:
        raise TypeError("mix_utterances works on the fp32 collated waveform")
    B, T = src.shape
    dst = torch.empty_like(src)
    low = torch.empty((B, T), dtype=torch.bfloat16, device=dev) if out_dtype == torch.bfloat16 else None
    L = _lib.lib()
    need = L.wavlm_mix_workspace_bytes(B, T)
    ws = workspace(dev, need, "mix")
    check(L.wavlm_mix_utterances(ptr(src), ptr(dst), ptr(low), B, T, ptr(ops_flat) if n_ops else None, int(n_ops),
                                 ptr(op_begin), ptr(noise), int(bool(normalize)), float(eps), ptr(ws), need, stream()),
          "wavlm_mix_utterances")
    return low if low is not None else dst
