"""One autograd node and one C call per transformer encoder block and direction (wavlm_encoder_layer_fwd / _bwd,
include/wavlm_hip.h): the training path of TransformerSentenceEncoderLayer (WavLM/WavLM.py:694-742) when the block runs in
its benchmarked configuration -- bf16, head_dim 64, erf GELU, no activation dropout, packed q|k|v bound by a flat-arena
optimizer, every parameter with a gradient sink.  The composed path (functional.LinearFn / AttnCoreFn / LayerNormFn ... one
node per kernel) stays for everything else (fp32 parity mode, other activations, hooks, no optimizer bound) and produces the
same numbers: both issue the same kernels in the same order (tests/test_layer_fused_gpu.py compares them bit for bit).

Why: the reference's trainer calls the model once per micro-batch (src/fairseq/trainer.py:697-760); at 35 ms of GPU work per
step the Python launch thread (~35 nodes, ~35 ctypes calls, ~60 tensor allocations per block) was one slow host away from
being the bottleneck.  Here a block costs the launch thread two allocations, one descriptor update and one call each way.
"""
import ctypes as C
import os
import weakref

import torch

from . import _lib, ops
from . import functional as F

LAYER_FUSED = os.environ.get("WAVLM_LAYER_FUSED", "1") != "0"


class TabGrad:
    """gradient of the relative-position table, shared by the blocks of one forward pass: every block's backward adds its
    share into ONE buffer (the first to run writes it), the block that owns the table hands the sum to autograd"""
    __slots__ = ("buf",)

    def __init__(self):
        self.buf = None


def _ptr(t):
    return t.data_ptr() if t is not None else None


class _Binding:
    """descriptor template of one block: parameter and gradient-sink addresses (stable while the optimizer's arenas are)"""

    def __init__(self, layer):
        at = layer.self_attn
        pk = at._packed
        fc1, fc2 = layer.fc1, layer.fc2
        ln1, ln2 = layer.self_attn_layer_norm, layer.final_layer_norm
        d = _lib.LayerDesc()
        d.D, d.H, d.F = layer.embedding_dim, at.num_heads, fc1.weight.shape[0]
        d.pre_ln = int(bool(layer.layer_norm_first))
        d.param_dtype = ops.dt(fc1.weight)
        d.eps1, d.eps2, d.scale = ln1.eps, ln2.eps, at.scaling
        self.sinks = []   # (parameter or None, sink view): what backward accumulates into, in notification order

        def bind(name, p, explicit=None):
            g = explicit if explicit is not None else F._sink(p)
            if g is None:
                raise RuntimeError("parameter without a gradient sink")
            setattr(d, name, p.data_ptr())
            setattr(d, "d" + name, g.data_ptr())
            self.sinks.append((p if explicit is None else None, g))
            return g

        bind("Wqkv", pk[0], pk[1]); bind("bqkv", pk[2], pk[3])
        bind("Wo", at.out_proj.weight); bind("bo", at.out_proj.bias)
        bind("W1", fc1.weight); bind("b1", fc1.bias); bind("W2", fc2.weight)
        self.b2_sink = bind("b2", fc2.bias)
        self.b2_index = len(self.sinks) - 1
        bind("ln1_g", ln1.weight); bind("ln1_b", ln1.bias); bind("ln2_g", ln2.weight); bind("ln2_b", ln2.bias)
        self.gate_from = len(self.sinks)
        if at.gru_rel_pos:
            self.gate_ptrs = (at.grep_linear.weight.data_ptr(), at.grep_linear.bias.data_ptr(), at.grep_a.data_ptr())
            self.gate_sinks = tuple(F._sink(p) for p in (at.grep_linear.weight, at.grep_linear.bias, at.grep_a))
            if any(g is None for g in self.gate_sinks):
                raise RuntimeError("parameter without a gradient sink")
            self.gate_grad_ptrs = tuple(g.data_ptr() for g in self.gate_sinks)
        else:
            self.gate_ptrs = self.gate_sinks = self.gate_grad_ptrs = None
        self.desc = d
        self.key = (pk[0].data_ptr(), pk[1].data_ptr(), self.b2_sink.data_ptr(), fc2.weight.data_ptr())

    def valid_for(self, layer):
        pk = layer.self_attn._packed
        g = layer.fc2.bias.grad
        return (pk is not None and g is not None
                and self.key == (pk[0].data_ptr(), pk[1].data_ptr(), g.data_ptr(), layer.fc2.weight.data_ptr()))


# layer -> _Binding.  Kept OUTSIDE the module's state: a _Binding holds a ctypes descriptor with pointer fields, which
# neither copy.deepcopy (EMA / teacher copies of a model that has already trained) nor pickle accept ("ctypes objects
# containing pointers cannot be pickled"); a copy of the model simply starts without bindings and builds its own.
_BINDINGS = weakref.WeakKeyDictionary()


def binding(layer):
    b = _BINDINGS.get(layer)
    if b is None or not b.valid_for(layer):
        b = _Binding(layer)
        _BINDINGS[layer] = b
    return b


def eligible(layer, x, table):
    """the block runs as ONE node when it is in the benchmarked configuration (module docstring)"""
    if not (LAYER_FUSED and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 3):
        return False
    at = layer.self_attn
    if at.head_dim != 64 or layer.activation_name != "gelu" or not F.USE_FUSED_ATTENTION:
        return False
    if layer.training and layer.activation_dropout > 0:
        return False
    if (table is not None or at.has_relative_attention_bias) and not at.gru_rel_pos:
        return False   # ungated relative position bias: composed path
    pk = at._packed
    if pk is None or pk[0].dtype != x.dtype or pk[0].data_ptr() != at.q_proj.weight.data_ptr() or pk[2].data_ptr() != at.q_proj.bias.data_ptr():
        return False
    ps = [at.out_proj.weight, at.out_proj.bias, layer.fc1.weight, layer.fc1.bias, layer.fc2.weight, layer.fc2.bias,
          layer.self_attn_layer_norm.weight, layer.self_attn_layer_norm.bias, layer.final_layer_norm.weight,
          layer.final_layer_norm.bias]
    if at.gru_rel_pos:
        ps += [at.grep_linear.weight, at.grep_linear.bias, at.grep_a]
    for p in ps:
        if not (p.requires_grad and getattr(p, "_wl_sink", False) and p.grad is not None and p.dtype == pk[0].dtype):
            return False
    return True


class EncoderLayerFn(torch.autograd.Function):
    """(y[, r_out]) = block(x[, r_in]); see wavlm_layer_desc.  `tab` carries a gradient only for the block that owns the
    table (TabGrad); `prev_tok`: pre-LN, the BiasGradToken of the linear that produced r_in; `out_tok`: pre-LN, the token
    of this block's fc2 bias (whoever consumes r_out through a LayerNorm takes it)."""

    @staticmethod
    def forward(ctx, x, r_in, tab, layer, kpm, tabgrad, owns_tab, prev_tok, out_tok):
        bd = binding(layer)
        B, T, D = x.shape
        xc = x if x.is_contiguous() else x.contiguous()
        rc = None
        if r_in is not None:
            rc = r_in if r_in.is_contiguous() else r_in.contiguous()
        training = layer.training
        p_drop = layer.dropout if training else 0.0
        p_attn = layer.self_attn.dropout_module.p if training else 0.0
        d = _lib.LayerDesc()
        C.memmove(C.byref(d), C.byref(bd.desc), C.sizeof(d))
        d.B, d.T = B, T
        d.p_drop, d.p_attn = p_drop, p_attn
        d.attn_store_p = 2 if F.ATTN_STORE_P == "bits" else int(bool(F.ATTN_STORE_P))   # what `saved` keeps of the attention (functional.ATTN_STORE_P)
        # seeds are drawn in the order the composed path draws them (same masks either way: the two paths can be compared
        # bit for bit with dropout on)
        if d.pre_ln and p_drop > 0 and rc is not None:
            d.seed_r1 = F.next_seed()
        if p_attn > 0:
            d.seed_attn = F.next_seed()
        if not d.pre_ln and p_drop > 0:
            d.seed_r1 = F.next_seed()
        if p_drop > 0:
            d.seed_r2 = F.next_seed()
        use_tab = tab is not None
        if use_tab:
            d.tab = tab.data_ptr()
            d.Wgate, d.bgate, d.grep_a = bd.gate_ptrs
        d.kpm = _ptr(kpm)
        d.x, d.r_in = xc.data_ptr(), _ptr(rc)
        y = torch.empty_like(xc)
        d.y = y.data_ptr()
        r_out = None
        if d.pre_ln:
            r_out = torch.empty_like(xc)
            d.r_out = r_out.data_ptr()
        L = _lib.lib()
        nsaved = int(L.wavlm_layer_saved_bytes(C.byref(d)))
        saved = torch.empty(nsaved, dtype=torch.uint8, device=x.device)
        d.saved, d.saved_bytes = saved.data_ptr(), nsaved
        need = int(L.wavlm_layer_fwd_workspace_bytes(C.byref(d)))
        ws = ops.workspace(x.device, need, "layer")
        d.workspace, d.ws_bytes = ws.data_ptr(), need
        _lib.check(L.wavlm_encoder_layer_fwd(C.byref(d), ops.stream()), "wavlm_encoder_layer_fwd")
        ctx.desc, ctx.bd, ctx.use_tab, ctx.tabgrad, ctx.owns_tab = d, bd, use_tab, tabgrad, owns_tab
        ctx.prev_sink = None
        if prev_tok is not None and rc is not None:
            ps = F._sink(prev_tok.param)
            if ps is not None:
                prev_tok.taken = True   # this block's LN1 backward delivers the bias gradient of the linear that produced r_in
                ctx.prev_sink = ps
        ctx.out_tok = out_tok
        if F.SINK_LISTENERS:
            for p, g in bd.sinks:
                F._sink_use(p, g)
            if use_tab:
                for g in bd.gate_sinks:
                    F._sink_use(None, g)
            # (no use count for prev_sink: the linear that owns that bias counted it in its own forward)
        ctx.save_for_backward(xc, rc, y if d.pre_ln else None, saved, tab, kpm)
        if d.pre_ln:
            return y, r_out
        return y

    @staticmethod
    def backward(ctx, dy, dr_out=None):
        xc, rc, y, saved, tab, kpm = ctx.saved_tensors
        d, bd = ctx.desc, ctx.bd
        dev = xc.device
        if d.pre_ln:
            if dy is None or dr_out is None:
                raise NotImplementedError("fused pre-LN block: both outputs must be used")
            dr_out = dr_out if dr_out.is_contiguous() else dr_out.contiguous()
            d.dr_out = dr_out.data_ptr()
            d.y = y.data_ptr()
        dy = dy if dy.is_contiguous() else dy.contiguous()
        d.dy = dy.data_ptr()
        dx = torch.empty_like(xc)
        d.dx = dx.data_ptr()
        dr_in = None
        if rc is not None:
            dr_in = torch.empty_like(xc)
            d.dr_in = dr_in.data_ptr()
        tg = ctx.tabgrad
        if ctx.use_tab:
            d.dWgate, d.dbgate, d.dgrep_a = bd.gate_grad_ptrs
            if tg.buf is None:
                tg.buf = torch.empty_like(tab)
                d.dtab_accumulate = 0
            else:
                d.dtab_accumulate = 1
            d.dtab = tg.buf.data_ptr()
        # pre-LN: the bias gradient of fc2 comes from the LayerNorm that consumed r_out, unless nobody took the token
        b2_here = True
        if d.pre_ln:
            b2_here = not (ctx.out_tok is not None and ctx.out_tok.taken)
            if not b2_here:
                d.db2 = None
            d.db2_prev = _ptr(ctx.prev_sink)
        L = _lib.lib()
        need = int(L.wavlm_layer_bwd_workspace_bytes(C.byref(d)))
        ws = ops.workspace(dev, need, "layer")
        d.workspace, d.ws_bytes = ws.data_ptr(), need
        _lib.check(L.wavlm_encoder_layer_bwd(C.byref(d), ops.stream()), "wavlm_encoder_layer_bwd")
        if F.SINK_LISTENERS:
            for i, (_p, g) in enumerate(bd.sinks):
                if i == bd.b2_index and not b2_here:
                    continue
                F._sink_written(g)
            if ctx.use_tab:
                for g in bd.gate_sinks:
                    F._sink_written(g)
            if ctx.prev_sink is not None:
                F._sink_written(ctx.prev_sink)
        dtab = tg.buf if (ctx.use_tab and ctx.owns_tab) else None
        return dx, dr_in, dtab, None, None, None, None, None, None


def run_block(layer, x, kpm, table, r_in=None, prev_tok=None, out_tok=None):
    """-> (y, r_out or None, table).  Creates the position table if this block owns the embedding."""
    at = layer.self_attn
    if at.has_relative_attention_bias and table is None:
        table = at.position_table(x.shape[1], x.device)
    tab_in, tg, owns = None, None, False
    if table is not None:
        tg = getattr(table, "_wl_tabgrad", None)
        if tg is None:
            tg = table._wl_tabgrad = TabGrad()
            owns = True
        tab_in = table if owns else table.detach()
    out = EncoderLayerFn.apply(x, r_in, tab_in, layer, kpm, tg, owns, prev_tok, out_tok)
    if layer.layer_norm_first:
        return out[0], out[1], table
    return out, None, table
