"""Autograd boundary of the hot path: each torch.autograd.Function below is a fused forward/backward pair built
from the HIP entry points in ops.py.  torch supplies the graph, gradient accumulation and memory; all arithmetic
on activations happens in libwavlm_hip.so.  torch ops that remain are pure data movement on parameter-sized
tensors (cat / permute / contiguous of weights) and scalar bookkeeping.

Layout: activations are channel-last [B, T, C]; the reference's [T, B, C] tensors are transposed views of these.
"""
import os
import threading
import weakref

import torch

from . import ops

_SEED_CTR = [0]
USE_FUSED_ATTENTION = True  # bf16 + head_dim 64 -> fused kernels; otherwise GEMM + softmax-row-kernel composition
# Fused attention, training: WAVLM_ATTN_STORE_P=1 makes the forward keep its probabilities (fp16, dropout decision in the sign
# bit; 467 MB per Base layer at 32 x 15 s) and the two backward kernels read them instead of recomputing scores + bias +
# exponentials + dropout words (include/wavlm_hip.h: wavlm_attn_fused_fwd_p).  Built, parity-tested in both modes and OFF:
# measured on the same box the step is 0.1-0.5 ms SLOWER with it (forward +45 us per layer for the 467 MB write; backward
# kernels unchanged at ~210 + ~200 us although they issue 40 % fewer VALU and 25 % fewer MFMA instructions) -- the backward
# kernels are bound by LDS bandwidth, not by the element pass (profiles/r05/attn_stored_p.txt).
# Round 6, WAVLM_ATTN_STORE_P=bits: the forward keeps only its dropout DECISIONS (one bit per element, 27 MB per Base layer) and
# both backward kernels select with the stored bits instead of evaluating the dropout hash (wavlm_attn_fused_dbits_bytes).  Built,
# bit-identical, and OFF as well: the forward pays 11.5 us per layer for producing the words (it is VALU-bound: one more
# instruction per element) and the dQ kernel gains 12 (profiles/r06/ab_attn_fwd_bits.txt).  The DEFAULT ("0") keeps nothing and
# every kernel recomputes its decisions (WAVLM_ATTN_DBITS=1: the dQ kernel hands them to the dK/dV kernel -- neutral as well).
_ASP = os.environ.get("WAVLM_ATTN_STORE_P", "0")
ATTN_STORE_P = True if _ASP == "1" else ("bits" if _ASP in ("bits", "2") else False)


def next_seed():
    """64-bit dropout seed: torch's current seed mixed with a per-process call counter (the fairseq Trainer reseeds
    torch every update, trainer.py:1194-1198, so masks are reproducible per update)."""
    _SEED_CTR[0] += 1
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + _SEED_CTR[0] * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF


def _rup(x, m):
    return (x + m - 1) // m * m


# chained consumers (LinearFn / GateFn / FFNFn pass_x): removes the torch add kernels autograd would launch for tensors
# with several consumers (3 per post-LN encoder layer).  WAVLM_CHAIN_CONSUMERS=0 restores one Function per consumer.
CHAIN_CONSUMERS = os.environ.get("WAVLM_CHAIN_CONSUMERS", "1") == "1"
# Grouped weight-gradient launches (WgradGroup): the four dW of an encoder layer as ONE split-K launch when their tiles fit
# one round of the persistent grid with a split >= 2 (ops.grouped_split: Base yes, Large no).  Measured on one box, same
# build (profiles/r03/envab_wg.txt): per layer 335 us + 4 x 7 us of slab reduction against 349 us + 4 x 13 us for the single
# launches, -0.47 ms per step at Base.  (Round 2 had it off: its measurement compared 425 us against 401 us on another
# build.)  WAVLM_WGRAD_GROUPING=0 / 1 forces single / grouped launches.
_WG = os.environ.get("WAVLM_WGRAD_GROUPING", "auto")
WGRAD_GROUPING = _WG != "0"
SINK_LISTENERS = []  # callables(tensor): told which arena slice a backward kernel has just accumulated into (dp.GradReducer)


def h2d(a, dev):
    """host array / CPU tensor -> device without stalling the host: a pageable `.to(device)` synchronises the stream,
    i.e. the host waits for every kernel enqueued so far and the GPU then idles until the host is ahead again (measured:
    ~1 ms of launch gaps around the prediction head).  Staged through pinned memory and copied asynchronously."""
    t = torch.from_numpy(a) if not torch.is_tensor(a) else a
    dev = torch.device(dev)
    if dev.type != "cuda" or t.device.type != "cpu" or t.numel() == 0:
        return t.to(dev)
    return t.pin_memory().to(dev, non_blocking=True)


_SINK_USES = {}      # arena slice (data_ptr) -> forward uses not yet matched by a backward accumulation


def reset_sink_uses():
    """called by the optimizer's zero_grad(): forwards whose backward never ran must not leave counts behind"""
    _SINK_USES.clear()
    _PADDED_GRAD[0] = None   # (a LayerNorm backward whose conv layer's backward never ran must not keep ~1 GB alive)


def _sink_use(p, explicit=None):
    """forward side: this parameter (or explicit sink view) will receive one more accumulation in backward"""
    if not SINK_LISTENERS:  # (single process: nobody listens -- skip the p.grad look-up, ~170 calls per step)
        return
    t = explicit if explicit is not None else _sink(p)
    if t is not None:
        k = t.data_ptr()
        _SINK_USES[k] = _SINK_USES.get(k, 0) + 1


def _sink_written(t):
    """backward side: one accumulation into slice t is enqueued; listeners hear about the slice once its LAST pending
    accumulation of this backward is in (a parameter used twice -- ILS heads, tied projections -- must not be reduced early)"""
    if not SINK_LISTENERS:
        return
    k = t.data_ptr()
    left = _SINK_USES.get(k, 1) - 1
    if left > 0:
        _SINK_USES[k] = left
        return
    _SINK_USES.pop(k, None)
    for cb in SINK_LISTENERS:
        cb(t)


def _sink(p):
    """Gradient sink of a parameter: FusedAdam points `p.grad` at its slice of the flat gradient arena (zeroed every
    step) and marks the parameter.  Backward kernels then accumulate straight into the arena and the Function returns
    None for that input, instead of materialising a gradient tensor that autograd adds into `p.grad` with one tiny
    elementwise kernel per parameter (~200 launches, 1.2 ms per step at WavLM-Base)."""
    if p is None or not getattr(p, "_wl_sink", False):
        return None
    g = p.grad
    return g if (g is not None and g.is_contiguous()) else None


# ------------------------------------------------------------------------------------------------ Linear
def _linear_fwd(x2d, W, b, *, epi=0, aux=None, res=None, out_dtype=None):
    n, K = x2d.shape
    N = W.shape[0]
    y = torch.empty((n, N), dtype=out_dtype or x2d.dtype, device=x2d.device)
    if n == 0:  # empty frame selection (e.g. a batch with nothing masked): the reference yields empty logits / zero loss
        return y
    ops.gemm(x2d, W, y, n, N, K, lda=K, ldb=K, ldc=N, bias=b, epi=epi, aux=aux, ld_aux=N, res=res, ld_res=N)
    return y


def _linear_bwd_x(dy2d, W, *, epi=0, aux=None, res=None, colsum=None):
    """dx[n, K] = dy[n, N] @ W[N, K]  (W consumed K-strided: no transposed weight copy).
    colsum [K] (optional): += column sums of dx -- the bias gradient of the linear that produced this layer's input, out of
    the GEMM's epilogue"""
    n, N = dy2d.shape
    K = W.shape[1]
    dx = torch.empty((n, K), dtype=dy2d.dtype, device=dy2d.device)
    if n == 0:
        return dx
    ops.gemm(dy2d, W, dx, n, K, N, lda=N, ldb=K, ldc=K, transB=True, epi=epi, aux=aux, ld_aux=K, res=res, ld_res=K,
             colsum=colsum, colsum_accumulate=colsum is not None)
    return dx


def _linear_bwd_w(dy2d, x2d, w_dtype, out=None):
    """dW[N, K] = dy[n, N]^T @ x[n, K]: both operands K-strided, split-K over the n rows; `out` (+)= if given"""
    n, N = dy2d.shape
    K = x2d.shape[1]
    dW = out if out is not None else torch.empty((N, K), dtype=w_dtype, device=dy2d.device)
    if n == 0:  # no rows: the gradient contribution is zero
        return dW if out is not None else dW.zero_()
    split = ops.pick_split(N, K, (n + 63) // 64)
    ops.gemm(dy2d, x2d, dW, N, K, n, lda=N, ldb=K, ldc=K, transA=True, transB=True, split_k=split,
             accumulate=out is not None)
    return dW


class WgradGroup:
    """Weight gradients of one encoder layer collected during its backward and issued as ONE grouped split-K launch
    (ops.gemm_wgrad_grouped): the four dW of a layer share the reduction length B*T, and alone each needs a split of
    7-28 to fill the GPU (fp32 slabs, short K loops).  Members are added by `_param_grads` in backward order; the group
    fires when `expected` members have arrived, and `flush_wgrad_groups()` (optimizer step / reducer finish) fires
    whatever is left, so a member that never receives a gradient cannot strand the others."""
    __slots__ = ("expected", "items", "sinks", "fired")
    pending = []
    deferred = set()  # data_ptr of every sink slice whose accumulation is queued but not yet enqueued on the stream

    def __init__(self, expected):
        self.expected, self.items, self.sinks, self.fired = expected, [], [], False

    def add(self, dy2d, x2d, sink_view, sink):
        if not self.items:
            if not WgradGroup.pending:  # first deferred member of this backward pass: make sure the pass ends flushed
                torch.autograd.Variable._execution_engine.queue_callback(flush_wgrad_groups)
            WgradGroup.pending.append(self)
        self.items.append((dy2d, x2d, sink_view))
        self.sinks.append(sink)
        WgradGroup.deferred.add(sink.data_ptr())
        if len(self.items) >= self.expected:
            self.fire()

    def fire(self):
        if self.fired or not self.items:
            return
        self.fired = True
        if self in WgradGroup.pending:
            WgradGroup.pending.remove(self)
        # Members are packed, in arrival order, into sub-groups whose tiles still fit ONE round of the persistent grid
        # with a split >= 2 (ops.grouped_split): Base: all four dW of a layer (108 tiles); Large: fc2 | fc1 | out_proj +
        # q|k|v (64 tiles each) -- out_proj alone needs a split of 16 (16 fp32 slabs for a 1024 x 1024 output).
        n = self.items[0][0].shape[0]
        kt = (n + 63) // 64
        tl = [((dy.shape[1] + 255) // 256) * ((x.shape[1] + 255) // 256) for dy, x, _ in self.items]
        groups, cur, cur_t = [], [], 0
        for it, t in zip(self.items, tl):
            if cur and (n == 0 or ops.grouped_split(cur_t + t, kt) < 2 or len(cur) == 4):
                groups.append(cur)
                cur, cur_t = [], 0
            cur.append(it)
            cur_t += t
        if cur:
            groups.append(cur)
        for grp in groups:
            if len(grp) == 1 or n == 0:
                for dy2d, x2d, out in grp:
                    _linear_bwd_w(dy2d, x2d, out.dtype, out=out)
            else:
                ops.gemm_wgrad_grouped(grp, grp[0][2].dtype)
        for sk in self.sinks:
            WgradGroup.deferred.discard(sk.data_ptr())
            _sink_written(sk)
        self.items, self.sinks = [], []


def grad_write_deferred(grad_view):
    """True while the accumulation into this gradient slice is still queued in a WgradGroup: autograd's
    post-accumulate hook of the parameter fires when the linear's backward returns (with an undefined gradient), which
    is BEFORE the grouped launch -- the data-parallel reducer must not count the parameter as ready on that hook."""
    return grad_view is not None and grad_view.data_ptr() in WgradGroup.deferred


def flush_wgrad_groups():
    """fire what is still queued: called by every reader of the gradient arena (optimizer step / norm, reducer finish)"""
    for g in list(WgradGroup.pending):
        g.fire()


class BiasGradToken:
    """Hand-over of a bias gradient from the nn.Linear that owns the bias to the LayerNorm that consumes the linear's
    output as its residual branch: the LayerNorm backward already reduces dgamma / dbeta over rows, and the column sums
    of the residual-branch gradient (= that bias gradient) fall out of the same pass.  The linear keeps computing its
    own bias gradient unless the LayerNorm forward marked the token as taken (it does so only when the bias has a
    gradient sink), so a code path that never reaches the LayerNorm loses nothing."""
    __slots__ = ("param", "taken")

    def __init__(self, param):
        self.param, self.taken = param, False


def _param_grads(dy2d, x2d, W, b, has_bias, need_w, need_b, sink_w=None, sink_b=None, bias_tok=None, wgroup=None):
    """(dW, db) for a linear layer; a parameter with a gradient sink gets its gradient accumulated in place -> None"""
    if bias_tok is not None and bias_tok.taken:
        has_bias = False  # accumulated by the consuming LayerNorm's backward
    dW = db = None
    sw = sink_w if sink_w is not None else _sink(W)
    if sw is not None and wgroup is not None and WGRAD_GROUPING and dy2d.dtype == torch.bfloat16:
        wgroup.add(dy2d, x2d, sw.view(W.shape), sw)  # deferred: issued with the layer's other weight gradients
    elif sw is not None:
        _linear_bwd_w(dy2d, x2d, W.dtype, out=sw.view(W.shape))
        _sink_written(sw)
    elif need_w:
        dW = _linear_bwd_w(dy2d, x2d, W.dtype)
    if has_bias:
        sb = sink_b if sink_b is not None else _sink(b)
        if sb is not None:
            if dy2d.shape[0] > 0:
                ops.colsum(dy2d, W.dtype, out=sb.view(-1), accumulate=True)
            _sink_written(sb)
        elif need_b:
            db = ops.colsum(dy2d, W.dtype) if dy2d.shape[0] > 0 else torch.zeros(W.shape[0], dtype=W.dtype, device=W.device)
    return dW, db


class LinearFn(torch.autograd.Function):
    """y = x W^T + b  (nn.Linear: WavLM/WavLM.py:348 post_extract_proj, modules.py q/k/v/out_proj, final_proj).
    sink_w / sink_b: explicit gradient sinks for W / b given as plain views (packed q|k|v projections)."""

    @staticmethod
    def forward(ctx, x, W, b, sink_w=None, sink_b=None, bias_tok=None, wgroup=None, pass_x=False):
        """pass_x=True: also returns x itself (a view).  A tensor with several consumers costs one torch add kernel per
        extra consumer in backward (autograd sums their gradients).  When the consumers are CHAINED instead -- each takes
        x, hands it on, and the last one is the only real reader of the alias -- every consumer receives the gradient
        accumulated so far and folds it into its own kernel: here as the residual operand of the dX GEMM's epilogue."""
        x2d = x.reshape(-1, x.shape[-1])
        if not x2d.is_contiguous():
            x2d = x2d.contiguous()
        y = _linear_fwd(x2d, W, b)
        ctx.save_for_backward(x2d, W, b)
        ctx.sinks = (sink_w, sink_b)
        ctx.bias_tok = bias_tok
        ctx.wgroup = wgroup
        if ctx.needs_input_grad[1]:  # (all False under no_grad: no backward will match the use)
            _sink_use(W, sink_w)
            if b is not None:
                _sink_use(b, sink_b)
        ctx.xshape = x.shape
        ctx.pass_x = pass_x
        y = y.view(*x.shape[:-1], W.shape[0])
        return (y, x.view_as(x)) if pass_x else y

    @staticmethod
    def backward(ctx, dy, dx_pass=None):
        x2d, W, b = ctx.saved_tensors
        dx = None
        if dy is None:  # only the alias was used downstream
            return dx_pass, None, None, None, None, None, None, None
        dy2d = dy.reshape(-1, dy.shape[-1])
        if not dy2d.is_contiguous():
            dy2d = dy2d.contiguous()
        if ctx.needs_input_grad[0]:
            res = None
            if dx_pass is not None:
                res = dx_pass.reshape(-1, dx_pass.shape[-1])
                if not res.is_contiguous():
                    res = res.contiguous()
            dx = _linear_bwd_x(dy2d, W, res=res).view(ctx.xshape)
        dW, db = _param_grads(dy2d, x2d, W, b, b is not None, ctx.needs_input_grad[1], ctx.needs_input_grad[2],
                              ctx.sinks[0], ctx.sinks[1], ctx.bias_tok, ctx.wgroup)
        return dx, dW, db, None, None, None, None, None


class FFNFn(torch.autograd.Function):
    """y = fc2(dropout(gelu(fc1 x)))  (WavLM/WavLM.py:732-737; gelu = exact erf, WavLM/modules.py:140-141).
    GELU runs in fc1's GEMM epilogue (pre-activation kept for backward); GELU' runs in the epilogue of fc2's
    input-gradient GEMM."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, p_act, seed, b2_tok=None, wgroup=None, pass_x=False, act="gelu"):
        """act: the reference's activation_fn (utils.get_activation_fn, WavLM/modules.py:144-160).  "gelu" (every released
        model) is the fused path; "relu" / "gelu_accurate" (= "gelu_fast") / "tanh" / "linear" run fc1 as a plain GEMM and
        the activation as one elementwise pass each way; "glu" makes fc1 a GLU_Linear(D, F, "swish") (W1 [2F, D]:
        WavLM/modules.py:99-129, WavLM/WavLM.py:668-669, 707-708)."""
        x2d = x.reshape(-1, x.shape[-1])
        if not x2d.is_contiguous():
            x2d = x2d.contiguous()
        n = x2d.shape[0]
        F = W1.shape[0]
        ctx.act = act
        if act == "gelu" and _INFERENCE_CALL[0] and p_act <= 0:
            u = x2d.new_empty(0)   # inference (infer_apply under no_grad): nobody reads gelu'(pre-activation)
            h = _linear_fwd(x2d, W1, b1, epi=3, aux=None)
        elif act == "gelu":
            u = torch.empty((n, F), dtype=x2d.dtype, device=x2d.device)
            # without activation dropout the auxiliary tensor holds gelu'(pre-activation) (epi 3): backward multiplies
            # by it (epi 4) instead of re-evaluating erf / exp; with dropout it holds the pre-activation itself
            h = _linear_fwd(x2d, W1, b1, epi=3 if p_act <= 0 else 1, aux=u)
        else:
            u = _linear_fwd(x2d, W1, b1)  # the pre-activation ([n, 2F] for "glu")
            if act == "glu":
                h = ops.glu_fwd(u, "swish")
            elif act == "linear":
                h = u
            elif act in ops.ACT_KINDS:
                h = ops.act_fwd(u, act)
            else:
                raise RuntimeError("--activation-fn %s not supported" % act)  # the reference's message (utils.py:555)
        hd = ops.dropout(h, p_act, seed) if p_act > 0 else h
        y = _linear_fwd(hd, W2, b2)
        ctx.save_for_backward(x2d, W1, W2, u, hd, b1, b2)
        if ctx.needs_input_grad[1]:
            for t in (W1, b1, W2, b2):
                if t is not None:
                    _sink_use(t)
        ctx.p_act, ctx.seed, ctx.xshape = p_act, seed, x.shape
        ctx.b2_tok = b2_tok
        ctx.wgroup = wgroup
        y = y.view(*x.shape[:-1], W2.shape[0])
        return (y, x.view_as(x)) if pass_x else y   # pass_x: see LinearFn.forward

    @staticmethod
    def backward(ctx, dy, dx_pass=None):
        x2d, W1, W2, u, hd, b1, b2 = ctx.saved_tensors
        if dy is None:
            return (dx_pass,) + (None,) * 10
        dy2d = dy.reshape(-1, dy.shape[-1])
        if not dy2d.is_contiguous():
            dy2d = dy2d.contiguous()
        dW2, db2 = _param_grads(dy2d, hd, W2, b2, b2 is not None, True, True, bias_tok=ctx.b2_tok, wgroup=ctx.wgroup)
        fused_b1 = False
        if ctx.act != "gelu":
            dh = _linear_bwd_x(dy2d, W2)
            if ctx.p_act > 0:
                dh = ops.dropout(dh, ctx.p_act, ctx.seed)
            if ctx.act == "glu":
                du = ops.glu_bwd(u, dh, "swish")
            elif ctx.act == "linear":
                du = dh
            else:
                du = ops.act_bwd(u, dh, ctx.act)
        elif ctx.p_act > 0:
            dh = _linear_bwd_x(dy2d, W2)
            dh = ops.dropout(dh, ctx.p_act, ctx.seed)
            n, F = dh.shape
            du, _ = ops.group_major(dh.view(1, n, F), u.view(1, n, F), 1, 0, n)
            du = du.view(n, F)
        else:
            # fc1's bias gradient = column sums of du: out of the epilogue of the GEMM that produces du
            sb1 = _sink(b1) if (b1 is not None and FUSE_BIAS_COLSUM) else None
            du = _linear_bwd_x(dy2d, W2, epi=4, aux=u, colsum=sb1.view(-1) if sb1 is not None else None)
            if sb1 is not None:
                _sink_written(sb1)
                fused_b1 = True
        dW1, db1 = _param_grads(du, x2d, W1, b1, b1 is not None and not fused_b1, True, True, wgroup=ctx.wgroup)
        dx = None
        if ctx.needs_input_grad[0]:
            res = None
            if dx_pass is not None:
                res = dx_pass.reshape(-1, dx_pass.shape[-1])
                if not res.is_contiguous():
                    res = res.contiguous()
            dx = _linear_bwd_x(du, W1, res=res).view(ctx.xshape)
        return dx, dW1, db1, dW2, db2, None, None, None, None, None, None


# --------------------------------------------------------------------------------------------- LayerNorm
# ---- tensors derived from parameters only (packed q|k|v, GEMM images of the conv / pos_conv weights): under torch.no_grad()
# (inference: WavLM.extract_features per call, WavLM/WavLM.py:323-375) they MAY be kept between calls and rebuilt when a source
# parameter changes.  What the key can see: the storage address, torch's in-place version counter and PARAM_EPOCH (bumped by
# invalidate_derived()).  What it CANNOT see: a write through `p.data` (a fresh alias with a version counter of its own) --
# which is exactly how the reference's optimizers update parameters (optim/adam.py:172-226 `p.data.addcdiv_`,
# fp16_optimizer.py:155-165 `p.data.copy_`, nag.py) and how EMA / teacher copies are usually maintained.  Therefore the cache
# is OPT-IN (round 6; it was on by default in round 5 and served stale validation passes after such an update):
#   * set_eval_cache(True) / `with frozen_parameters():` / WAVLM_EVAL_CACHE=1 -- the caller states that between two inference
#     calls parameters change only through torch in-place ops on the parameter itself, this package's writers (FusedAdam,
#     load_state_dict hooks, the fairseq plugin's optimizer wrappers: all call invalidate_derived()) or not at all;
#   * anything else that writes through `.data` / raw pointers calls invalidate_derived() itself.
# Worth 2 % of an extract_features call (9.37 against 9.55 ms, profiles/r05): a frozen feature extractor opts in, training
# with periodic validation does not need to.  With autograd on nothing is ever cached.  One entry per (first source parameter,
# tag); entries die with that parameter.
_EVAL_DERIVED = {}   # id(first source parameter) -> (weak reference to it, {tag: (key, value)}); tensors compare elementwise, so no WeakKeyDictionary
EVAL_CACHE = os.environ.get("WAVLM_EVAL_CACHE", "0") == "1"
PARAM_EPOCH = [0]   # bumped by invalidate_derived(): every writer that changes parameters behind torch's back


def invalidate_derived():
    """every arena- / pointer- / `.data`-level parameter writer calls this (FusedAdam.step / load_state_dict / master sync,
    the fairseq plugin's optimizer wrappers, module load_state_dict and train() / eval() transitions)"""
    PARAM_EPOCH[0] += 1


def set_eval_cache(on):
    """opt into (or out of) keeping parameter-derived tensors between inference calls; returns the previous setting"""
    global EVAL_CACHE
    old, EVAL_CACHE = EVAL_CACHE, bool(on)
    if not on:
        _EVAL_DERIVED.clear()
    return old


class frozen_parameters:
    """`with frozen_parameters(): feats = model.extract_features(wav)` -- the cache is on inside and emptied of nothing on
    exit (entries stay valid for the next block as long as the keys match; invalidate_derived() drops them all)"""

    def __enter__(self):
        self._old = set_eval_cache(True)
        return self

    def __exit__(self, *exc):
        global EVAL_CACHE
        EVAL_CACHE = self._old
        return False


# Inside a Function.forward grad mode is always off and ctx.needs_input_grad follows requires_grad alone (True for parameters
# under torch.no_grad() as well): whether a call is inference is only visible at the call site.  infer_apply(Fn, ...) notes it
# for the forward it starts; a direct Fn.apply leaves the flag False (nothing kept, backward images built): always correct.
# Thread-local: a no_grad evaluation thread beside a training thread (or torch.nn.DataParallel's replicas) must not make a
# training forward skip the stores its backward reads.
class _InferenceFlag(threading.local):
    def __init__(self):
        self.on = False

    # list-style access kept for the Function.forward bodies and the tests: _INFERENCE_CALL[0]
    def __getitem__(self, i):
        return self.on

    def __setitem__(self, i, v):
        self.on = bool(v)


_INFERENCE_CALL = _InferenceFlag()


def infer_apply(fn, *args):
    _INFERENCE_CALL[0] = not torch.is_grad_enabled()
    try:
        return fn.apply(*args)
    finally:
        _INFERENCE_CALL[0] = False


def eval_derived(params, tag, build, inference=None):
    """inference: None = decide by torch.is_grad_enabled() (module code); a Function.forward passes _INFERENCE_CALL[0]"""
    if inference is None:
        inference = not torch.is_grad_enabled()
    if not inference or not EVAL_CACHE:
        return build()
    key = (PARAM_EPOCH[0],) + tuple((p.data_ptr(), p._version, p.dtype, p.device) for p in params)
    p0 = params[0]
    held = _EVAL_DERIVED.get(id(p0))
    if held is None or held[0]() is not p0:
        held = (weakref.ref(p0, lambda _r, k=id(p0): _EVAL_DERIVED.pop(k, None)), {})
        _EVAL_DERIVED[id(p0)] = held
    slot = held[1]
    ent = slot.get(tag)
    if ent is not None and ent[0] == key:
        return ent[1]
    val = build()
    slot[tag] = (key, val)
    return val


class LayerNormFn(torch.autograd.Function):
    """y = dropout_out(act(LN(x + dropout_in(r)))); returns (y, s) with s = the pre-norm sum (not differentiable).
    grad_scale multiplies the incoming gradient (GradMultiply at the extractor output).
    pass_x: also return an alias of x; the gradient that arrives at the alias (the residual stream of a pre-LN block) is
    added to dx inside the backward kernel instead of by an autograd add."""

    @staticmethod
    def forward(ctx, x, r, gamma, beta, eps, act, p_in, seed_in, p_out, seed_out, grad_scale, rbias_tok=None,
                pass_x=False, s_grad=False, grad_pad=None):
        """s_grad (with a residual r): the second output s = x + dropout(r) is differentiable -- it IS the residual stream
        of a pre-LN block whose residual add is fused into this LayerNorm; the gradient that arrives at s is added inside
        the backward kernel and reaches r through the dropout mask as well."""
        xc = x.contiguous()
        rc = r.contiguous() if r is not None else None
        y, s, mean, rstd = ops.layernorm_fwd(xc, rc, gamma, beta, eps, act=act, p_in=p_in, seed_in=seed_in,
                                             p_out=p_out, seed_out=seed_out, save=True)
        ctx.save_for_backward(s, mean, rstd, gamma, beta)
        ctx.cfg = (act, p_in, seed_in, p_out, seed_out, grad_scale, r is not None)
        ctx.s_grad = bool(s_grad and r is not None)
        # grad_pad = (fp, bp): x [B, T, D] is the output of a conv layer whose backward reads its incoming gradient with fp / bp
        # zero rows around every utterance (ConvStackFn.backward) -- dx is then written in that layout at once
        ctx.grad_pad = (tuple(grad_pad) if (grad_pad is not None and r is None and not pass_x and x.dim() == 3
                                            and x.shape[-1] in (512, 768, 1024) and ops.LN_SEG_OK) else None)
        ctx.rbias = None
        if (rbias_tok is not None and r is not None and _sink(rbias_tok.param) is not None
                and _sink(gamma) is not None and _sink(beta) is not None):
            rbias_tok.taken = True  # this backward delivers the bias gradient of the linear that produced r
            ctx.rbias = rbias_tok.param
        if ctx.needs_input_grad[2]:
            _sink_use(gamma)
            _sink_use(beta)
        ctx.set_materialize_grads(False)  # no zero-filled [rows, D] gradient for the second output
        if ctx.s_grad:
            return y, s
        s_out = s.detach()
        ctx.mark_non_differentiable(s_out)
        if pass_x:
            return y, s_out, x.view_as(x)
        return y, s_out

    @staticmethod
    def backward(ctx, dy, _ds, dx_pass=None):
        if ctx.s_grad:
            dx_pass = _ds  # the residual stream's gradient arrives at s
            if dy is None:
                raise NotImplementedError("fused pre-LN residual: the normalised output must be used")
        if dy is None:
            return (dx_pass,) + (None,) * 14
        if dx_pass is not None:
            dx_pass = dx_pass.contiguous()
            if dx_pass.dtype != dy.dtype:
                dx_pass = dx_pass.to(dy.dtype)
        s, mean, rstd, gamma, beta = ctx.saved_tensors
        act, p_in, seed_in, p_out, seed_out, grad_scale, has_r = ctx.cfg
        sg, sb = _sink(gamma), _sink(beta)
        if sg is None or sb is None:
            sg = sb = None
        sc = _sink(ctx.rbias) if (ctx.rbias is not None and sg is not None) else None
        if ctx.rbias is not None and sc is None:
            raise RuntimeError("gradient sink of a handed-over bias disappeared between forward and backward")
        dx, dr, dgamma, dbeta, _ = ops.layernorm_bwd(dy.contiguous(), s, mean, rstd, gamma, beta, act=act, p_in=p_in,
                                                     seed_in=seed_in, p_out=p_out, seed_out=seed_out,
                                                     grad_scale=grad_scale, need_dr=has_r and p_in > 0,
                                                     dgamma=sg, dbeta=sb,
                                                     dr_colsum=sc.view(-1) if sc is not None else None,
                                                     dx_add=dx_pass, dr_incl_add=ctx.s_grad,
                                                     dx_pad=ctx.grad_pad if dx_pass is None else None)
        padded = getattr(dx, "_padded", None)
        if padded is not None and ctx.needs_input_grad[0]:  # handed to the conv layer's backward, which runs next (one slot: see ConvStackFn.backward)
            _PADDED_GRAD[0] = (dx.data_ptr(), tuple(dx.shape), tuple(dx.stride()), padded, ctx.grad_pad)
        if sg is not None:
            dgamma = dbeta = None  # accumulated in place
            _sink_written(sg); _sink_written(sb)
        if sc is not None:
            _sink_written(sc)
        if has_r and dr is None:
            dr = dx
        return dx, (dr if has_r else None), dgamma, dbeta, None, None, None, None, None, None, None, None, None, None, None


def layer_norm(x, gamma, beta, eps=1e-5, *, residual=None, act=0, p_in=0.0, p_out=0.0, training=True,
               grad_scale=1.0, residual_bias_tok=None, pass_x=False, s_grad=False, grad_pad=None):
    p_in = p_in if training else 0.0
    p_out = p_out if training else 0.0
    return LayerNormFn.apply(x, residual, gamma, beta, eps, act, p_in, next_seed() if p_in > 0 else 0, p_out,
                             next_seed() if p_out > 0 else 0, grad_scale, residual_bias_tok, pass_x, s_grad, grad_pad)


# The one gradient a LayerNormFn.backward has just written in a conv layer's padded layout: (data_ptr, shape, strides, the whole
# padded buffer, (fp, bp)).  ConvStackFn.backward -- the only consumer of that gradient, and the next node autograd runs --
# takes it if it is handed exactly that view, and empties the slot either way.
_PADDED_GRAD = [None]


def conv_grad_pad(T_in, k, s):
    """(fp, bp) of the zero rows ConvStackFn.backward wants around every utterance of the gradient of a (k, s) conv layer's
    output, for LayerNormFn's grad_pad"""
    _, _, fp, bp = _conv_geometry(T_in, k, s)
    return fp, bp


# ------------------------------------------------------------------------------------- feature extractor
class Conv0Fn(torch.autograd.Function):
    """conv0 (k=10) + GroupNorm(C, C) + GELU -> [B, T0, C]  (WavLM/WavLM.py:420-426)"""

    @staticmethod
    def forward(ctx, wav, W, gamma, beta, stride, eps, out_dtype):
        wav = wav.contiguous()
        y, stats = ops.conv0_gn_gelu_fwd(wav, W.contiguous(), gamma, beta, stride, eps, out_dtype)
        ctx.save_for_backward(wav, W, gamma, beta, stats)
        ctx.stride = stride
        return y

    @staticmethod
    def backward(ctx, g):
        wav, W, gamma, beta, stats = ctx.saved_tensors
        dW, dgamma, dbeta = ops.conv0_gn_gelu_bwd(wav, W.contiguous(), gamma, beta, g.contiguous(), stats, ctx.stride)
        return None, dW, dgamma, dbeta, None, None, None


class Conv0LNFn(torch.autograd.Function):
    """conv0 (k=10) + LayerNorm over channels + GELU -> [B, T0, C]: block 0 of extractor_mode 'layer_norm'
    (WavLM-Large; WavLM/WavLM.py:403-418)"""

    @staticmethod
    def forward(ctx, wav, W, gamma, beta, stride, eps, out_dtype, bias=None):
        wav = wav.contiguous()
        y = ops.conv0_ln_gelu_fwd(wav, W.contiguous(), gamma, beta, stride, eps, out_dtype, bias=bias)
        ctx.save_for_backward(wav, W, gamma, beta, bias)
        ctx.stride, ctx.eps = stride, eps
        return y

    @staticmethod
    def backward(ctx, g):
        wav, W, gamma, beta, bias = ctx.saved_tensors
        dW, dgamma, dbeta, dbias = ops.conv0_ln_gelu_bwd(wav, W.contiguous(), gamma, beta, g.contiguous(), ctx.stride,
                                                         ctx.eps, bias=bias)
        return None, dW, dgamma, dbeta, None, None, None, dbias


def _conv_geometry(T_in, k, s):
    T_out = (T_in - k) // s + 1
    J = [len(range(r, k, s)) for r in range(s)]
    fp = max(J) - 1
    bp = (T_in + s - 1) // s - T_out
    return T_out, J, fp, max(bp, 0)


class ConvStackFn(torch.autograd.Function):
    """conv1..convL of the feature extractor in 'default' mode: Conv1d(no bias) -> GELU, channel-last.
    (WavLM/WavLM.py:428, 499-500).  Forward: one overlapping-row GEMM per layer (row t of A is the k*C_in
    contiguous values starting at frame s*t: lda = s*C_in < K), GELU in the epilogue.  Backward per layer: weight
    gradient = K-strided GEMM reduced over (batch, time); data gradient = one GEMM per stride phase writing
    du_{i-1} = dx * gelu'(u_{i-1}) straight into the zero-padded buffer the next (earlier) layer consumes."""

    @staticmethod
    def forward(ctx, x, specs, act, *params):
        """act=True: GELU fused (default mode); act=False: plain convolution (layer_norm mode: the LayerNorm + GELU
        that follows is a LayerNormFn).  params: one weight per layer, then (conv_bias=True) one bias per layer."""
        B = x.shape[0]
        nl = len(specs)
        weights = params[:nl]
        biases = params[nl:] if len(params) > nl else (None,) * nl
        ctx.has_bias = len(params) > nl
        xs, us = [], []
        cur = x.contiguous()
        # GEMM operand images of all layers' weights in ONE launch (forward layout and, when a backward will follow, the
        # stride-phase layouts of the data-gradient GEMMs): six permute copies + twelve flip / copy pairs of torch before
        infer = _INFERENCE_CALL[0]
        need_bwd = any(ctx.needs_input_grad) and not infer
        wfs, wbs = eval_derived(list(weights), ("conv_images", tuple(specs), need_bwd),
                                lambda: ops.conv_weights_relayout([W.contiguous() for W in weights], specs, need_bwd),
                                inference=infer)
        for (k, s), W, bias, Wf in zip(specs, weights, biases, wfs):
            Cout, Cin, _ = W.shape
            T_in = cur.shape[1]
            T_out = (T_in - k) // s + 1
            y = torch.empty((B, T_out, Cout), dtype=cur.dtype, device=cur.device)
            keep = act and not infer   # GELU' of the pre-activation: only a backward reads it
            u = torch.empty_like(y) if keep else y.new_empty(0)
            ops.gemm(cur, Wf, y, T_out, Cout, k * Cin, lda=s * Cin, ldb=k * Cin, ldc=Cout, batch=(B, 1),
                     sA=(T_in * Cin, 0), sC=(T_out * Cout, 0), epi=3 if act else 0, aux=u if keep else None, ld_aux=Cout,
                     sAux=(T_out * Cout, 0), bias=bias)
            xs.append(cur); us.append(u)
            cur = y
        ctx.specs = specs
        ctx.act = act
        ctx.nl = len(specs)
        ctx.has_wb = wbs is not None
        for i, W in enumerate(weights):
            if ctx.needs_input_grad[3 + i]:
                _sink_use(W)
        # the back-propagation weight images travel through autograd's saved-tensor machinery like every other saved tensor
        # (version checks, saved_tensors_hooks; released with the graph)
        ctx.save_for_backward(*xs, *us, *weights, *(wbs or ()))
        return cur

    @staticmethod
    def backward(ctx, dy):
        nl, specs = ctx.nl, ctx.specs
        saved = ctx.saved_tensors
        xs, us, weights = saved[:nl], saved[nl:2 * nl], saved[2 * nl:3 * nl]
        wb_flat = saved[3 * nl:] if ctx.has_wb else None
        B = dy.shape[0]
        dev = dy.device
        grads = [None] * nl
        bgrads = [None] * nl
        # du_L = dy * gelu'(u_L), laid out with the zero rows layer L's data-gradient GEMMs read
        k, s = specs[-1]
        T_out, J, fp, bp = _conv_geometry(xs[-1].shape[1], k, s)
        Cout = weights[-1].shape[0]
        slot, _PADDED_GRAD[0] = _PADDED_GRAD[0], None
        if (slot is not None and not ctx.act and slot[0] == dy.data_ptr() and slot[1] == tuple(dy.shape)
                and slot[2] == tuple(dy.stride()) and slot[4] == (fp, bp) and slot[3].shape == (B, fp + T_out + bp, Cout)):
            P = slot[3]   # the LayerNorm behind this layer wrote its input gradient in the padded layout: no copy
        else:
            P, _ = ops.group_major(dy.contiguous(), us[-1] if ctx.act else None, 1, fp, fp + T_out + bp, aux_is_grad=True)
            P = P.view(B, fp + T_out + bp, Cout)
        for i in range(nl - 1, -1, -1):
            k, s = specs[i]
            W = weights[i]
            Cout, Cin, _ = W.shape
            x = xs[i]
            T_in = x.shape[1]
            T_out, J, fp, bp = _conv_geometry(T_in, k, s)
            Tp = fp + T_out + bp
            # ---- weight gradient: dWf[co, (kk, ci)] = sum_{b,t} du[b,t,co] * x[b, s*t + kk, ci]
            if ctx.needs_input_grad[3 + i]:
                dWf = torch.empty((Cout, k * Cin), dtype=W.dtype, device=dev)
                split = ops.pick_split(Cout, k * Cin, B * ((T_out + 63) // 64))
                ops.gemm(P, x, dWf, Cout, k * Cin, T_out, lda=Cout, ldb=s * Cin, ldc=k * Cin, transA=True, transB=True,
                         a_off=fp * Cout, KB=B, sA_kb=Tp * Cout, sB_kb=T_in * Cin, split_k=split)
                grads[i] = dWf   # [Cout][k * Cin]: scattered into the parameter layout for all layers at once below
            if ctx.has_bias and ctx.needs_input_grad[3 + nl + i]:
                # bias gradient = column sums of du_i; the zero pad rows of the staged buffer add nothing
                bgrads[i] = ops.colsum(P.reshape(-1, Cout), W.dtype)
            # ---- data gradient
            need_dx = i > 0 or ctx.needs_input_grad[0]
            if not need_dx:
                break
            if i > 0:
                kp, sp = specs[i - 1]
                _, _, fpp, bpp = _conv_geometry(xs[i - 1].shape[1], kp, sp)
                Tpp = fpp + T_in + bpp
                # The zero pad rows in front of / behind every utterance's T_in rows: the back pad of utterance b and the front pad of
                # utterance b + 1 are adjacent, so with bpp spare rows in front of the buffer and fpp behind it ALL pads are the
                # B + 1 equally spaced gaps [g * Tpp, g * Tpp + bpp + fpp) of one allocation: ONE strided fill per layer
                # (two strided fills per layer before: 9 of the step's small torch launches).
                if fpp or bpp:
                    big = torch.empty((B * Tpp + fpp + bpp, Cin), dtype=dy.dtype, device=dev)
                    big.as_strided((B + 1, fpp + bpp, Cin), (Tpp * Cin, Cin, 1)).zero_()
                    nxt = big[bpp:bpp + B * Tpp].view(B, Tpp, Cin)
                else:
                    nxt = torch.empty((B, Tpp, Cin), dtype=dy.dtype, device=dev)
                aux = us[i - 1] if ctx.act else None
            else:
                fpp, Tpp = 0, T_in
                nxt = torch.empty((B, T_in, Cin), dtype=dy.dtype, device=dev)
                aux = None
            wb_views = ops.conv_phase_views(wb_flat[i], Cout, Cin, k, s)
            for r in range(s):
                Jr = J[r]
                if Jr == 0:
                    raise NotImplementedError("conv kernel narrower than its stride")
                # taps r + s*(Jr-1), ..., r + s, r of this stride phase, newest first.  As a strided slice + flip: indexing
                # with a Python list builds the index tensor on the host and copies it with a blocking H2D transfer, i.e.
                # a stream synchronisation in the middle of backward (measured: the launch thread stalled 23 ms here
                # every step and lost all its run-ahead for the rest of the step)
                Wb = wb_views[r]   # [Cin, Jr * Cout], taps of the phase newest first (ops.conv_weights_relayout)
                Mr = (T_in - r + s - 1) // s
                ops.gemm(P, Wb, nxt, Mr, Cin, Jr * Cout, lda=Cout, ldb=Jr * Cout, ldc=s * Cin, batch=(B, 1),
                         a_off=(fp - Jr + 1) * Cout, sA=(Tp * Cout, 0), c_off=(fpp + r) * Cin, sC=(Tpp * Cin, 0),
                         epi=4 if aux is not None else 0, aux=aux, aux_off=r * Cin, ld_aux=s * Cin,
                         sAux=(T_in * Cin, 0))
            P = nxt
        dx = P if ctx.needs_input_grad[0] else None
        # weight gradients [Cout][k * Cin] -> [Cout][Cin][k] for all layers in one launch: accumulated straight into the
        # gradient arena where the parameters have sinks (no per-layer permuted views for autograd to add), fresh
        # tensors otherwise
        have = [i for i in range(nl) if grads[i] is not None]
        if have:
            sinks = [_sink(weights[i]) for i in have]
            if all(t is not None for t in sinks):
                ops.conv_wgrad_scatter([t.view(weights[i].shape) for t, i in zip(sinks, have)], [grads[i] for i in have],
                                       [specs[i] for i in have], accumulate=True)
                for t, i in zip(sinks, have):
                    grads[i] = None
                    _sink_written(t)
            else:
                outs = [torch.empty_like(weights[i], memory_format=torch.contiguous_format) for i in have]
                ops.conv_wgrad_scatter(outs, [grads[i] for i in have], [specs[i] for i in have], accumulate=False)
                for o, i in zip(outs, have):
                    grads[i] = o
                    t = _sink(weights[i])   # (mixed case: a sink that autograd fills through the returned tensor)
                    if t is not None:
                        _SINK_USES.pop(t.data_ptr(), None)
        return (dx, None, None) + tuple(grads) + (tuple(bgrads) if ctx.has_bias else ())


# ---------------------------------------------------------------------------------------------- pos_conv
POSCONV_DIRECT = os.environ.get("WAVLM_POSCONV_DIRECT", "1") != "0"
FUSE_BIAS_COLSUM = os.environ.get("WAVLM_FUSE_BIAS_COLSUM", "1") != "0"  # fc1 bias gradient out of fc2's dX GEMM epilogue


class PosConvFn(torch.autograd.Function):
    """out = x + gelu(weight_norm_conv1d(x) + bias)[:, :T]  (WavLM/WavLM.py:514-527, 577-579; SamePad drops the last
    frame).  G*B overlapping-row GEMMs over a group-major, time-padded copy of x; bias, GELU and the residual add
    run in the GEMM epilogue."""

    @staticmethod
    def forward(ctx, x, v, g, bias, groups):
        B, T, D = x.shape
        K = v.shape[2]
        Cg = D // groups
        xc = x.contiguous()
        direct = POSCONV_DIRECT and ops.posconv_direct_supported(xc.dtype, Cg, K, T) and bias.dtype == xc.dtype
        Wf, Wb, norm = eval_derived([v, g], ("posconv_images", xc.dtype, direct),
                                    lambda: ops.posconv_weight_fwd(v.contiguous(), g.contiguous().view(-1), xc.dtype,
                                                                   layout=1 if direct else 0),
                                    inference=_INFERENCE_CALL[0])
        Tp = T + K - 1
        xg, _ = ops.group_major(xc, None, groups, K // 2, Tp)
        out = torch.empty_like(xc)
        keep = not (direct and _INFERENCE_CALL[0])   # the pre-activation: only a backward reads it
        u = torch.empty_like(xc) if keep else xc.new_empty(0)
        if direct:  # direct convolution: the input window stays in LDS (csrc/posconv_direct.hip)
            ops.posconv_direct(xg, Wf, out, T, K, bias=bias.contiguous(), res=xc, aux=u if keep else None, gelu=True)
        else:
            ops.gemm(xg, Wf, out, T, Cg, K * Cg, lda=Cg, ldb=K * Cg, ldc=D, batch=(B, groups),
                     sA=(groups * Tp * Cg, Tp * Cg), sB=(0, Cg * K * Cg), sC=(T * D, Cg), bias=bias, sBias=(0, Cg), epi=1,
                     aux=u, ld_aux=D, sAux=(T * D, Cg), res=xc, ld_res=D, sRes=(T * D, Cg))
        ctx.save_for_backward(xg, u, Wb, norm, v, g)
        ctx.dims = (B, T, D, K, Cg, groups, Tp)
        ctx.direct = direct
        return out

    @staticmethod
    def backward(ctx, dy):
        xg, u, Wb, norm, v, g = ctx.saved_tensors
        B, T, D, K, Cg, G, Tp = ctx.dims
        dyc = dy.contiguous()
        dug, du = ops.group_major(dyc, u, G, K // 2 - 1, Tp, want_nat=True)
        dbias = ops.colsum(du.view(B * T, D), v.dtype)
        if ctx.direct and ops._lib.lib().wavlm_posconv_dw_direct_splits(Cg, G) > 0:
            # direct form: fp32 partial sums over parts of the batch, added up by the weight-norm backward
            dWf, nsplit = ops.posconv_dw_direct(xg, dug, K // 2 - 1, T, K)
        else:
            # weight gradient in the forward GEMM layout, fp32: dWf[g][col][(tap, ci)]
            # computed transposed ([(tap, ci)][col]: M = 6144 rows, N = 48): the 48-wide side sits on the 64-column tile
            # edge (75 % MFMA use) instead of on a 128-row tile edge (37 %) -- 1.26 -> ~0.5 ms at cfg2
            dWfT = torch.empty((G, K * Cg, Cg), dtype=torch.float32, device=dy.device)
            ops.gemm(xg, dug, dWfT, K * Cg, Cg, T, lda=Cg, ldb=Cg, ldc=Cg, transA=True, transB=True,
                     b_off=(K // 2 - 1) * Cg, KB=B, sA_kb=G * Tp * Cg, sB_kb=G * Tp * Cg, batch=(1, G),
                     sA=(0, Tp * Cg), sB=(0, Tp * Cg), sC=(0, Cg * K * Cg))
            dWf, nsplit = dWfT.transpose(1, 2).contiguous(), 1
        dv, dg = ops.posconv_weight_bwd(dWf, v.contiguous(), g.contiguous().view(-1), norm, nsplit=nsplit)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(dyc)
            if ctx.direct:
                ops.posconv_direct(dug, Wb, dx, T, K, res=dyc)
            else:
                ops.gemm(dug, Wb, dx, T, Cg, K * Cg, lda=Cg, ldb=K * Cg, ldc=D, batch=(B, G),
                         sA=(G * Tp * Cg, Tp * Cg), sB=(0, Cg * K * Cg), sC=(T * D, Cg), res=dyc, ld_res=D,
                         sRes=(T * D, Cg))
        return dx, dv, dg.view_as(g), dbias, None


# --------------------------------------------------------------------------------------------- attention
class RelPosTableFn(torch.autograd.Function):
    """rel[h, d] = relative_attention_bias.weight[bucket[d], h]  (compute_bias, WavLM/modules.py:444-455): the
    [T, T] bias of head h is Toeplitz, so only its 2T-1 distinct values are kept."""

    @staticmethod
    def forward(ctx, emb, bucket):
        H = emb.shape[1]
        tab = ops.relpos_gather(emb.contiguous(), bucket, H, bucket.numel())
        ctx.save_for_backward(emb, bucket)
        return tab

    @staticmethod
    def backward(ctx, dtab):
        emb, bucket = ctx.saved_tensors
        return ops.relpos_scatter(dtab.contiguous(), bucket, emb), None


class GateFn(torch.autograd.Function):
    """gate[b,h,t] from the un-projected layer input (gru_rel_pos, WavLM/modules.py:523-533)"""

    @staticmethod
    def forward(ctx, x, W, bias, grep_a, H, pass_x=False):
        xc = x.contiguous()
        a = grep_a.contiguous().view(-1)
        gate, ga, gb = ops.gate_fwd(xc, W.contiguous(), bias.contiguous(), a, H)
        ctx.save_for_backward(xc, W, bias, grep_a, ga, gb)
        ctx.H = H
        if ctx.needs_input_grad[1]:
            for t in (W, bias, grep_a):
                _sink_use(t)
        return (gate, x.view_as(x)) if pass_x else gate   # pass_x: see LinearFn.forward

    @staticmethod
    def backward(ctx, dgate, dx_pass=None):
        xc, W, bias, grep_a, ga, gb = ctx.saved_tensors
        if dgate is None:
            return dx_pass, None, None, None, None, None
        sinks = (_sink(W), _sink(bias), _sink(grep_a))
        if any(t is None for t in sinks) or not (W.is_contiguous() and grep_a.is_contiguous()):
            sinks = None
        # chained consumer: the gate's gradient of x is ADDED into the gradient the later consumers already produced (in
        # place, inside the gate kernel).  The buffer may alias the LayerNorm's residual-branch gradient (dr is dx when
        # there is no residual dropout): every reader of that gradient (out_proj's backward -> attention backward ->
        # dgate) has been enqueued before this kernel by construction, so the in-place update is ordered after them.
        acc = None
        if dx_pass is not None and dx_pass.is_contiguous() and dx_pass.dtype == xc.dtype and ctx.needs_input_grad[0]:
            acc = dx_pass
        dx, dW, dbias, da = ops.gate_bwd(dgate.contiguous(), xc, W.contiguous(), bias, grep_a.contiguous().view(-1),
                                         ga, gb, ctx.H, sinks=None if sinks is None else tuple(t.view(-1) for t in sinks),
                                         dx_accumulate=acc)
        if acc is None and dx_pass is not None:
            dx = dx + dx_pass
        if sinks is not None:
            for t in sinks:
                _sink_written(t)
            return dx, None, None, None, None, None
        return dx, dW, dbias, da.view_as(grep_a), None, None


class AttnCoreFn(torch.autograd.Function):
    """O = dropout(softmax(scale * Q K^T + gate_i * rel[j-i] + key_padding)) V from the packed qkv [B, T, 3D].
    Round-1 form: batched MFMA GEMMs + one softmax row kernel that applies the Toeplitz bias on the fly (the
    [B*H, T, T] bias tensor of the reference is never built); S (fp32) and P are kept for backward."""

    @staticmethod
    def forward(ctx, qkv, gate, tab, kpm, H, scale, p_drop, seed, qkv_bias_tok=None, qkv_bias_sink=None):
        """qkv_bias_tok / qkv_bias_sink: hand-over of the packed q|k|v projection's bias gradient (BiasGradToken): the fused
        backward kernels deliver it (column sums of dq | dk | dv per block + one finishing launch) and the projection skips
        its own pass over dqkv."""
        B, T, D3 = qkv.shape
        D = D3 // 3
        hd = D // H
        dev = qkv.device
        qkvc = qkv.contiguous()
        ctx.fused = USE_FUSED_ATTENTION and qkv.dtype == torch.bfloat16 and hd == 64
        ctx.bias_sink = None
        if ctx.fused and qkv_bias_tok is not None and qkv_bias_sink is not None and ctx.needs_input_grad[0]:
            qkv_bias_tok.taken = True
            ctx.bias_sink = qkv_bias_sink
        if ctx.fused:
            store = ATTN_STORE_P if ctx.needs_input_grad[0] else False   # (grad mode is always off inside Function.forward: not a condition; `x and y` would turn "bits" into True)
            O, lse, pstore = ops.attn_fused_fwd(qkvc, gate, tab, kpm, H, scale, p_drop, seed, store_p=store)
            ctx.save_for_backward(qkvc, O, lse, gate, tab, kpm, pstore)
            ctx.cfg = (B, T, D, H, hd, 0, scale, p_drop, seed)
            return O
        ld = _rup(T, 8)
        S = torch.empty((B * H, T, ld), dtype=torch.float32, device=dev)
        ops.gemm(qkvc, qkvc, S, T, T, hd, lda=D3, ldb=D3, ldc=ld, batch=(B, H), sA=(T * D3, hd), sB=(T * D3, hd),
                 b_off=D, sC=(H * T * ld, T * ld), alpha=scale)
        P = torch.empty((B * H, T, ld), dtype=qkv.dtype, device=dev)
        lse = torch.empty((B * H, T), dtype=torch.float32, device=dev)
        ops.attn_softmax_fwd(S, P, lse, gate, tab, kpm, B, H, T, ld, ld, p_drop, seed)
        O = torch.empty((B, T, D), dtype=qkv.dtype, device=dev)
        ops.gemm(P, qkvc, O, T, hd, T, lda=ld, ldb=D3, ldc=D, transB=True, batch=(B, H), sA=(H * T * ld, T * ld),
                 sB=(T * D3, hd), b_off=2 * D, sC=(T * D, hd))
        ctx.save_for_backward(qkvc, S, P, lse, gate, tab, kpm)
        ctx.cfg = (B, T, D, H, hd, ld, scale, p_drop, seed)
        return O

    @staticmethod
    def backward(ctx, dO):
        if ctx.fused:
            qkvc, O, lse, gate, tab, kpm, pstore = ctx.saved_tensors
            B, T, D, H, hd, _, scale, p_drop, seed = ctx.cfg
            sb = ctx.bias_sink
            dqkv, dgate, dtab = ops.attn_fused_bwd(qkvc, O, dO.contiguous(), lse, gate, tab, kpm, H, scale, p_drop, seed,
                                                   dbias=sb.view(-1) if sb is not None else None, dbias_accumulate=True,
                                                   pstore=pstore)
            if sb is not None:
                _sink_written(sb)
            return dqkv, dgate, dtab, None, None, None, None, None, None, None
        qkvc, S, P, lse, gate, tab, kpm = ctx.saved_tensors
        B, T, D, H, hd, ld, scale, p_drop, seed = ctx.cfg
        D3 = 3 * D
        dev = dO.device
        dOc = dO.contiguous()
        dP = torch.empty((B * H, T, ld), dtype=P.dtype, device=dev)
        ops.gemm(dOc, qkvc, dP, T, T, hd, lda=D, ldb=D3, ldc=ld, batch=(B, H), sA=(T * D, hd), sB=(T * D3, hd),
                 b_off=2 * D, sC=(H * T * ld, T * ld))
        dS = torch.empty_like(dP)
        dgate = dtab = None
        if tab is not None:
            dgate = torch.empty((B, H, T), dtype=torch.float32, device=dev)
            dtab = torch.empty_like(tab)
        ops.attn_softmax_bwd(S, dP, lse, gate, tab, kpm, dS, dgate, dtab, B, H, T, ld, ld, p_drop, seed)
        dqkv = torch.empty_like(qkvc)
        # dQ = scale * dS K ; dK = scale * dS^T Q ; dV = P^T dO
        ops.gemm(dS, qkvc, dqkv, T, hd, T, lda=ld, ldb=D3, ldc=D3, transB=True, batch=(B, H),
                 sA=(H * T * ld, T * ld), sB=(T * D3, hd), b_off=D, sC=(T * D3, hd), c_off=0, alpha=scale)
        ops.gemm(dS, qkvc, dqkv, T, hd, T, lda=ld, ldb=D3, ldc=D3, transA=True, transB=True, batch=(B, H),
                 sA=(H * T * ld, T * ld), sB=(T * D3, hd), b_off=0, sC=(T * D3, hd), c_off=D, alpha=scale)
        ops.gemm(P, dOc, dqkv, T, hd, T, lda=ld, ldb=D, ldc=D3, transA=True, transB=True, batch=(B, H),
                 sA=(H * T * ld, T * ld), sB=(T * D, hd), sC=(T * D3, hd), c_off=2 * D)
        return dqkv, dgate, dtab, None, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------ masking / rows
class SelectRowsFn(torch.autograd.Function):
    """y = x; y[sel] = emb; y[zero] = 0  (apply_mask + the encoder's padding zero-fill, WavLM/WavLM.py:286, 574-575)"""

    @staticmethod
    def forward(ctx, x, sel, emb, zero):
        xc = x.contiguous()
        y = ops.select_rows(xc, sel, emb, zero)
        ctx.save_for_backward(sel, zero, emb)
        return y

    @staticmethod
    def backward(ctx, dy):
        sel, zero, emb = ctx.saved_tensors
        dyc = dy.contiguous()
        dx = ops.select_rows(dyc, sel, None, zero) if ctx.needs_input_grad[0] else None
        demb = None
        if emb is not None and sel is not None and ctx.needs_input_grad[2]:
            demb = ops.colsum(dyc.view(-1, dyc.shape[-1]), emb.dtype, include=sel, exclude=zero)
        return dx, None, demb, None


class GatherRowsFn(torch.autograd.Function):
    """y = x2d[idx]; backward scatters through the inverse index (rows are unique)"""

    @staticmethod
    def forward(ctx, x2d, idx, inv_idx):
        ctx.save_for_backward(inv_idx)
        ctx.n = x2d.shape[0]
        if idx.numel() == 0:
            return x2d.new_empty((0, x2d.shape[1]))
        return ops.gather_rows(x2d.contiguous(), idx, idx.numel())

    @staticmethod
    def backward(ctx, dy):
        (inv_idx,) = ctx.saved_tensors
        if dy.shape[0] == 0:
            return dy.new_zeros((ctx.n, dy.shape[1])), None, None
        return ops.gather_rows(dy.contiguous(), inv_idx, ctx.n), None, None


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.p, ctx.seed = p, seed
        return ops.dropout(x.contiguous(), p, seed)

    @staticmethod
    def backward(ctx, dy):
        return ops.dropout(dy.contiguous(), ctx.p, ctx.seed), None, None


def dropout(x, p, training=True):
    if not training or p <= 0:
        return x
    return DropoutFn.apply(x, p, next_seed())


# -------------------------------------------------------------------------------------------------- loss
class MaskedPredLossFn(torch.autograd.Function):
    """sum-reduced cross entropy of cos(proj_x, label_embs) / temp against the frame labels, plus the count of
    correct frames.  Identical to compute_nce + F.cross_entropy(logits, 0, 'sum') of the reference
    (src/fairseq/models/wavlm/wavlm.py:426-438; criterions/wavlm_criterion.py:68-71, 115-125) without the
    [V+1, S, 256] targets tensor.  Returns (loss_sum[1], n_correct[1]) as fp32 device tensors."""

    @staticmethod
    def forward(ctx, proj, label_embs, target, temp, need_grad):
        S, F = proj.shape
        V = label_embs.shape[0]
        dev = proj.device
        act_dtype = proj.dtype
        if S == 0:
            ctx.save_for_backward(proj, label_embs)
            ctx.dims = (0, V, F, 0, temp)
            z = torch.zeros(1, dtype=torch.float32, device=dev)
            nc = torch.zeros(1, dtype=torch.float32, device=dev)
            ctx.mark_non_differentiable(nc)
            return z, nc
        pn, inv_p = ops.l2norm_fwd(proj.contiguous(), act_dtype)
        en, inv_e = ops.l2norm_fwd(label_embs.contiguous(), act_dtype)
        logits = torch.empty((max(S, 1), V), dtype=torch.float32, device=dev)
        if S > 0:
            ops.gemm(pn, en, logits, S, V, F, lda=F, ldb=F, ldc=V, alpha=1.0 / temp)
        ldd = _rup(V, 8)
        dlog = torch.empty((max(S, 1), ldd), dtype=act_dtype, device=dev) if need_grad else None
        loss_rows, correct_rows = ops.ce_rows(logits, target, V, V, dlog, ldd, 1.0)
        loss = ops.sum_f32(loss_rows) if S > 0 else torch.zeros(1, dtype=torch.float32, device=dev)
        ncorrect = ops.sum_f32(correct_rows) if S > 0 else torch.zeros(1, dtype=torch.float32, device=dev)
        ctx.save_for_backward(pn, en, inv_p, inv_e, dlog, proj, label_embs)
        ctx.dims = (S, V, F, ldd, temp)
        ctx.mark_non_differentiable(ncorrect)
        return loss, ncorrect

    @staticmethod
    def backward(ctx, dloss, _dc):
        S, V, F, ldd, temp = ctx.dims
        if S == 0:
            proj, label_embs = ctx.saved_tensors
            return torch.zeros_like(proj), torch.zeros_like(label_embs), None, None, None
        pn, en, inv_p, inv_e, dlog, proj, label_embs = ctx.saved_tensors
        dev = pn.device
        g = dloss.reshape(1).to(torch.float32)
        # d pn = dlogits @ en / temp ; d en = dlogits^T @ pn / temp
        dpn = torch.empty((S, F), dtype=pn.dtype, device=dev)
        ops.gemm(dlog, en, dpn, S, F, V, lda=ldd, ldb=F, ldc=F, transB=True, alpha=1.0 / temp)
        den = torch.empty((V, F), dtype=en.dtype, device=dev)
        ops.gemm(dlog, pn, den, V, F, S, lda=ldd, ldb=F, ldc=F, transA=True, transB=True, alpha=1.0 / temp,
                 split_k=ops.pick_split(V, F, (S + 63) // 64))
        dproj = ops.l2norm_bwd(dpn, pn, inv_p, proj.dtype)
        demb = ops.l2norm_bwd(den, en, inv_e, label_embs.dtype)
        ops.scale_dev_(dproj, g)
        ops.scale_dev_(demb, g)
        return dproj, demb, None, None, None


class ActFn(torch.autograd.Function):
    """elementwise activation of ops.ACT_KINDS (erf gelu, relu, gelu_accurate, tanh) on its own -- the projection MLP of a deeper
    GumbelVectorQuantizer (modules/gumbel_vector_quantizer.py:54-66)"""

    @staticmethod
    def forward(ctx, x, kind):
        xc = x.contiguous()
        ctx.save_for_backward(xc)
        ctx.kind = kind
        return ops.act_fwd(xc, kind)

    @staticmethod
    def backward(ctx, dy):
        (xc,) = ctx.saved_tensors
        return ops.act_bwd(xc, dy.contiguous(), ctx.kind), None


class GLUFn(torch.autograd.Function):
    """nn.GLU over the last dimension (target_glu's second half, src/fairseq/models/wavlm/wavlm.py:322-327)"""

    @staticmethod
    def forward(ctx, x):
        x2d = x.reshape(-1, x.shape[-1]).contiguous()
        ctx.save_for_backward(x2d)
        ctx.xshape = x.shape
        return ops.glu_fwd(x2d).view(*x.shape[:-1], x.shape[-1] // 2)

    @staticmethod
    def backward(ctx, dy):
        (x2d,) = ctx.saved_tensors
        return ops.glu_bwd(x2d, dy.reshape(-1, dy.shape[-1]).contiguous()).view(ctx.xshape)


class FeaturesPenFn(torch.autograd.Function):
    """features.float().pow(2).mean()  (src/fairseq/models/wavlm/wavlm.py:486)"""

    @staticmethod
    def forward(ctx, feats, grad_scale=1.0):
        # grad_scale: the reference takes features_pen AFTER GradMultiply (wavlm.py:479,486), so the penalty's
        # gradient into the extractor is scaled by feature_grad_mult as well
        fc = feats.contiguous()
        ctx.save_for_backward(fc)
        ctx.grad_scale = grad_scale
        return ops.sumsq(fc, 1.0 / fc.numel())

    @staticmethod
    def backward(ctx, g):
        (fc,) = ctx.saved_tensors
        d = torch.empty_like(fc)
        ops.axpby_(d, fc, 2.0 * ctx.grad_scale / fc.numel(), 0.0)
        ops.scale_dev_(d, g.reshape(1).to(torch.float32))
        return d, None


# ------------------------------------------------------------------------ utterance-contrastive head (UniSpeech-SAT)
class UttContrastiveLossFn(torch.autograd.Function):
    """mean BCE-with-logits of cos(x_s, y_{idx[s, n]}) / temp against same-utterance indicators: compute_pred_spk of
    UniSpeech-SAT (src/fairseq/models/unispeech_sat/unispeech_sat.py:701-737) with compute_nce(replace_inf=False)
    (545-557).  idx[:, 0] is the row itself; y = the (Gumbel-quantised, projected) targets, or None = the projections
    themselves (no quantiser: the reference's positive is the projection itself).  The gathered [N, S, C] instance tensor
    of the reference is never built.  Returns (loss[1], accuracy[1])."""

    @staticmethod
    def forward(ctx, proj, idx, targets_u8, temp, y=None):
        S, N1 = idx.shape
        xn, inv = ops.l2norm_fwd(proj.contiguous(), proj.dtype)
        if y is not None:
            yn, inv_y = ops.l2norm_fwd(y.contiguous(), y.dtype)
        else:
            yn, inv_y = xn, inv
        logits = ops.gather_dot(xn, yn, idx, 1.0 / temp)
        out, dl = ops.bce_logits(logits, targets_u8, 1.0 / (temp * S * N1), want_grad=True)
        ctx.save_for_backward(xn, inv, yn, inv_y, dl, idx, proj, y if y is not None else proj)
        ctx.has_y = y is not None
        loss, acc = out[0:1].clone(), out[1:2].clone()
        ctx.mark_non_differentiable(acc)
        return loss, acc

    @staticmethod
    def backward(ctx, dloss, _dacc):
        xn, inv, yn, inv_y, dl, idx, proj, y = ctx.saved_tensors
        S, N1 = idx.shape
        dev = xn.device
        w = (dl * dloss.reshape(1).to(torch.float32)).view(-1)      # d loss / d logits, already / temp / numel
        flat = idx.view(-1)
        # direct half: d xn[s] = sum_n w[s, n] * yn[idx[s, n]]
        off = torch.arange(0, (S + 1) * N1, N1, dtype=torch.int32, device=dev)
        dxn = ops.rows_wsum(yn, flat, w, off, S)
        # transposed half: d yn[j] = sum_{(s, n): idx[s, n] = j} w[s, n] * xn[s]   (entries sorted by gathered row)
        src_t, order, off_t = _csr_by_target(flat, S, N1)
        w_t = w.index_select(0, order)
        if not ctx.has_y:
            ops.rows_wsum(xn, src_t, w_t, off_t, S, out=dxn, accumulate=True)
            return ops.l2norm_bwd(dxn, xn, inv, proj.dtype), None, None, None, None
        dyn = ops.rows_wsum(xn, src_t, w_t, off_t, S)
        return (ops.l2norm_bwd(dxn, xn, inv, proj.dtype), None, None, None, ops.l2norm_bwd(dyn, yn, inv_y, y.dtype))


def _csr_by_target(flat_idx, n_rows, per_row):
    """entries of a [S, per_row] index matrix grouped by the row they point to: (source row of each entry, order, offsets)"""
    order = torch.argsort(flat_idx.long(), stable=True)
    src_t = torch.div(order, per_row, rounding_mode="floor").to(torch.int32)
    off_t = torch.zeros(n_rows + 1, dtype=torch.int32, device=flat_idx.device)
    off_t[1:] = torch.cumsum(torch.bincount(flat_idx.long(), minlength=n_rows), 0).to(torch.int32)
    return src_t, order, off_t


class SampledNegativesLossFn(torch.autograd.Function):
    """InfoNCE over sampled negatives (wav2vec 2.0 / UniSpeech: Wav2Vec2Model.compute_preds, models/wav2vec/wav2vec2.py:
    533-553, + Wav2vecCriterion's cross_entropy(logits, 0, 'sum'), criterions/wav2vec_criterion.py:44-64).
    x [S, C]: context projections; y [S, C]: targets; idx [S, 1 + N] int32 rows of y (column 0 = the positive, i.e. s
    itself, then the sampled negatives); negatives equal to the positive are masked to -inf.  Returns
    (loss_sum[1], n_correct[1])."""

    @staticmethod
    def forward(ctx, x, y, idx, temp):
        S, N1 = idx.shape
        xn, inv_x = ops.l2norm_fwd(x.contiguous(), x.dtype)
        yn, inv_y = ops.l2norm_fwd(y.contiguous(), y.dtype)
        logits = ops.gather_dot(xn, yn, idx, 1.0 / temp, mask_equal=True)
        target = torch.zeros(S, dtype=torch.int32, device=x.device)
        dlog = torch.empty((S, N1), dtype=torch.float32, device=x.device)
        loss_rows, correct_rows = ops.ce_rows(logits, target, N1, N1, dlog, N1, 1.0)
        loss, ncorrect = ops.sum_f32(loss_rows), ops.sum_f32(correct_rows)
        ctx.save_for_backward(xn, yn, inv_x, inv_y, dlog, idx, x, y)
        ctx.temp = temp
        ctx.mark_non_differentiable(ncorrect)
        return loss, ncorrect

    @staticmethod
    def backward(ctx, dloss, _dc):
        xn, yn, inv_x, inv_y, dlog, idx, x, y = ctx.saved_tensors
        S, N1 = idx.shape
        dev = xn.device
        w = (dlog * (dloss.reshape(1).to(torch.float32) / ctx.temp)).view(-1)
        flat = idx.view(-1)
        off = torch.arange(0, (S + 1) * N1, N1, dtype=torch.int32, device=dev)
        dxn = ops.rows_wsum(yn, flat, w, off, S)                                   # d xn[s] = sum_n w[s,n] yn[idx[s,n]]
        src_t, order, off_t = _csr_by_target(flat, y.shape[0], N1)
        dyn = ops.rows_wsum(xn, src_t, w.index_select(0, order), off_t, y.shape[0])  # d yn[j] = sum_{idx=j} w xn[s]
        return ops.l2norm_bwd(dxn, xn, inv_x, x.dtype), ops.l2norm_bwd(dyn, yn, inv_y, y.dtype), None, None


def sample_negatives_indices(bsz, tsz, num, n_negatives, cross_sample_negatives, padding_count=None):
    """Index draws of Wav2Vec2Model.sample_negatives (wav2vec2.py:474-531): same torch.randint calls on the CPU RNG.
    Returns int64 [bsz, num * N]: negative n of frame t of utterance b is [b, t * N + n], indexing the [bsz * tsz] rows."""
    high, cross_high = tsz - (padding_count or 0), tsz * bsz
    assert high > 1
    neg = cross = None
    if n_negatives > 0:
        tszs = torch.arange(num).unsqueeze(-1).expand(-1, n_negatives).flatten()
        neg = torch.randint(low=0, high=high - 1, size=(bsz, n_negatives * num))
        neg[neg >= tszs] += 1
    if cross_sample_negatives > 0:
        tszs = torch.arange(num).unsqueeze(-1).expand(-1, cross_sample_negatives).flatten()
        cross = torch.randint(low=0, high=cross_high - 1, size=(bsz, cross_sample_negatives * num))
        cross[cross >= tszs] += 1
    if n_negatives > 0:
        for i in range(1, bsz):
            neg[i] += i * high
    else:
        neg = cross
    if cross_sample_negatives > 0 and n_negatives > 0:
        neg = torch.cat([neg, cross], dim=1)
    return neg


# ------------------------------------------------------------------------------------- Gumbel vector quantiser
class GumbelVQFn(torch.autograd.Function):
    """Per-row part of GumbelVectorQuantizer.forward (src/fairseq/modules/gumbel_vector_quantizer.py:157-213) on the
    projected logits [n, G*V]: hard / soft code statistics, Gumbel-softmax with the straight-through estimator, codebook
    product.  Returns (x [n, G*var_dim], prob_perplexity[1], code_perplexity[1]).  `noise`: the Gumbel draws [n*G, V]
    (host-drawn for RNG parity with the reference's F.gumbel_softmax) or None = drawn on the device from `seed`.
    The reference's [n, G, V] one-hot times codebook product is a gather of G code vectors per row; its backward a
    batched GEMM (d one-hot = dx . vars^T), the softmax backward kernel and a segmented row sum into the codebook."""

    @staticmethod
    def forward(ctx, logits, vars_, G, V, tau, training, noise, seed):
        n = logits.shape[0]
        lc = logits.contiguous()
        vd = vars_.shape[-1]
        idx, ysoft, ppl, dA = ops.gumbel_vq_fwd(lc, G, V, tau, training, noise, seed)
        codes = vars_.reshape(G * V, vd).contiguous()
        x = ops.gather_rows(codes, idx, n * G).view(n, G * vd)
        ctx.save_for_backward(lc, ysoft, idx, dA, codes)
        ctx.cfg = (n, G, V, vd, tau, training, vars_.shape)
        prob, code = ppl[0:1].clone(), ppl[1:2].clone()
        ctx.mark_non_differentiable(code)
        return x, prob, code

    @staticmethod
    def backward(ctx, dx, dprob, _dcode):
        lc, ysoft, idx, dA, codes = ctx.saved_tensors
        n, G, V, vd, tau, training, vshape = ctx.cfg
        dev = lc.device
        dret = None
        dvars = None
        if dx is not None:
            dxc = dx.contiguous()
            if training:
                # d one-hot[row, g, v] = <dx[row, g], vars[g, v]>  (fp32: feeds the softmax backward)
                dret = torch.empty((n, G * V), dtype=torch.float32, device=dev)
                ops.gemm(dxc, codes.to(dxc.dtype), dret, n, V, vd, lda=G * vd, ldb=vd, ldc=G * V, batch=(1, G),
                         sA=(0, vd), sB=(0, V * vd), sC=(0, V))
            # d vars[c] = sum of dx[row, g] over the (row, g) that selected code c (the one-hot's forward value is 1)
            flat = idx.view(-1)
            src_t, _order, off_t = _csr_by_target(flat, G * V, 1)
            ones = torch.ones(n * G, dtype=torch.float32, device=dev)
            dv = ops.rows_wsum(dxc.view(n * G, vd), src_t, ones, off_t, G * V)
            dvars = dv.to(codes.dtype).view(vshape)
        dppl = dprob.reshape(1).to(torch.float32).contiguous() if dprob is not None else None
        if dret is None and dppl is None:
            return None, dvars, None, None, None, None, None, None
        dl = ops.gumbel_vq_bwd(lc, ysoft if dret is not None else None, dret, dA, dppl, G, V, tau)
        return dl, dvars, None, None, None, None, None, None


def host_gumbel_noise(n_rows, V):
    """the draws of F.gumbel_softmax on the CPU generator (torch/nn/functional.py gumbel_softmax:
    -empty_like(logits).exponential_().log()), for RNG parity with the reference's CPU run"""
    return -torch.empty((n_rows, V), dtype=torch.float32).exponential_().log()
