"""WavLM on MI355X: the standalone `WavLM` / `WavLMConfig` / `extract_features` surface of the reference
(WavLM/WavLM.py:162-375) over the HIP kernels in libwavlm_hip.so.

The module tree mirrors the reference one-to-one (same attribute names, same registration order, same
initialisers called in the same order), so that
  * `load_state_dict(checkpoint["model"])` works on released checkpoints unchanged, and
  * `torch.manual_seed(s); WavLM(cfg)` yields bit-identical parameters to the reference constructor
    (SURVEY.md 8(a) row Q).
The nn.Conv1d / nn.Linear / nn.LayerNorm objects are parameter containers only: their torch forward is never
called.  Every forward below routes through unispeech_amd.functional (HIP); there is no CPU execution path.

Activations are [B, T, C]; where the reference API exposes [T, B, C] tensors (layer_results) transposed views are
returned.
"""
import math
import os
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import functional as F
from . import layerfn
from .masking import compute_mask_indices, relative_position_buckets


class WavLMConfig:
    """Same fields and defaults as WavLM/WavLM.py:162-217."""

    def __init__(self, cfg=None):
        self.extractor_mode: str = "default"
        self.encoder_layers: int = 12
        self.encoder_embed_dim: int = 768
        self.encoder_ffn_embed_dim: int = 3072
        self.encoder_attention_heads: int = 12
        self.activation_fn: str = "gelu"
        self.layer_norm_first: bool = False
        self.conv_feature_layers: str = "[(512,10,5)] + [(512,3,2)] * 4 + [(512,2,2)] * 2"
        self.conv_bias: bool = False
        self.feature_grad_mult: float = 1.0
        self.normalize: bool = False
        self.dropout: float = 0.1
        self.attention_dropout: float = 0.1
        self.activation_dropout: float = 0.0
        self.encoder_layerdrop: float = 0.0
        self.dropout_input: float = 0.0
        self.dropout_features: float = 0.0
        self.mask_length: int = 10
        self.mask_prob: float = 0.65
        self.mask_selection: str = "static"
        self.mask_other: float = 0
        self.no_mask_overlap: bool = False
        self.mask_min_space: int = 1
        self.mask_channel_length: int = 10
        self.mask_channel_prob: float = 0.0
        self.mask_channel_selection: str = "static"
        self.mask_channel_other: float = 0
        self.no_mask_channel_overlap: bool = False
        self.mask_channel_min_space: int = 1
        self.conv_pos: int = 128
        self.conv_pos_groups: int = 16
        self.relative_position_embedding: bool = False
        self.num_buckets: int = 320
        self.max_distance: int = 1280
        self.gru_rel_pos: bool = False
        if cfg is not None:
            self.update(cfg)

    def update(self, cfg: dict):
        self.__dict__.update(cfg)


# ------------------------------------------------------------------------------------------ parameter init
def _bert_normal_(t):
    # drawn on the CPU generator exactly like the reference (init_bert_params.normal_)
    t.copy_(t.cpu().normal_(mean=0.0, std=0.02).to(t.device))


def init_bert_params(module):
    """BERT-style re-initialisation applied post-order over the encoder (WavLM/modules.py:168-200): every Linear
    and Embedding weight ~ N(0, 0.02), biases zero, then q/k/v of each attention block once more."""
    if isinstance(module, nn.Linear):
        _bert_normal_(module.weight.data)
        if module.bias is not None:
            module.bias.data.zero_()
    if isinstance(module, nn.Embedding):
        _bert_normal_(module.weight.data)
        if module.padding_idx is not None:
            module.weight.data[module.padding_idx].zero_()
    if isinstance(module, MultiheadAttention):
        _bert_normal_(module.q_proj.weight.data)
        _bert_normal_(module.k_proj.weight.data)
        _bert_normal_(module.v_proj.weight.data)


# ------------------------------------------------------------------------------------------ feature extractor
class _DerivedOwner:
    """mixin of the modules whose inference path keeps tensors derived from their parameters (functional.eval_derived, opt-in):
    every train() / eval() transition and every state-dict load drops what was kept.  The reference's optimizers write
    parameters through `p.data` (optim/adam.py:172-226, fp16_optimizer.py:155-165), which no version counter records; the
    Trainer calls model.train() before every update and model.eval() before every validation pass (trainer.py:697, 907), so a
    validation pass never sees images of the weights of an earlier step."""

    def train(self, mode=True):
        F.invalidate_derived()
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        F.invalidate_derived()
        return super()._load_from_state_dict(*args, **kwargs)


class ConvFeatureExtractionModel(_DerivedOwner, nn.Module):
    """7 x {Conv1d(no bias) -> [GroupNorm on block 0 | LayerNorm every block] -> GELU}
    (WavLM/WavLM.py:378-504, conv_type 'default').  State-dict keys: conv_layers.{i}.0.weight,
    conv_layers.0.2.{weight,bias} (GroupNorm) or conv_layers.{i}.2.1.{weight,bias} (layer_norm mode)."""

    def __init__(self, conv_layers: List[Tuple[int, int, int]], dropout: float = 0.0, mode: str = "default",
                 conv_bias: bool = False):
        super().__init__()
        assert mode in {"default", "layer_norm"}
        self.mode = mode
        self.specs = [(k, s) for (_, k, s) in conv_layers]
        self.conv_layers = nn.ModuleList()
        in_d = 1
        for i, (dim, k, stride) in enumerate(conv_layers):
            conv = nn.Conv1d(in_d, dim, k, stride=stride, bias=conv_bias)
            nn.init.kaiming_normal_(conv.weight)
            if mode == "layer_norm":
                norm = nn.Sequential(nn.Identity(), nn.LayerNorm(dim, elementwise_affine=True), nn.Identity())
                block = nn.Sequential(conv, nn.Dropout(p=dropout), norm, nn.GELU())
            elif i == 0:
                block = nn.Sequential(conv, nn.Dropout(p=dropout), nn.GroupNorm(dim, dim, affine=True), nn.GELU())
            else:
                block = nn.Sequential(conv, nn.Dropout(p=dropout), nn.GELU())
            self.conv_layers.append(block)
            in_d = dim

    def forward(self, x):
        """x: waveform [B, T] -> features [B, T', C] (channel-last; the reference returns [B, C, T'])"""
        blk0 = self.conv_layers[0]
        conv0 = blk0[0]
        has_bias = conv0.bias is not None  # conv_bias=True: Conv1d biases on every block
        wdt = conv0.weight.dtype
        if x.dtype != wdt:
            x = x.to(wdt)
        rest = list(self.conv_layers)[1:]
        if self.mode == "layer_norm":
            # every block: conv -> LayerNorm over channels -> GELU (WavLM/WavLM.py:403-418).  Block 0 is one fused
            # kernel; blocks 1.. are an overlapping-row GEMM (bias in its epilogue) followed by the fused LayerNorm +
            # GELU row kernel.
            ln0 = blk0[2][1]
            y = F.Conv0LNFn.apply(x, conv0.weight, ln0.weight, ln0.bias, self.specs[0][1], ln0.eps, wdt, conv0.bias)
            for blk, spec in zip(rest, self.specs[1:]):
                ln = blk[2][1]
                params = (blk[0].weight,) + ((blk[0].bias,) if has_bias else ())
                v = F.infer_apply(F.ConvStackFn, y, (spec,), False, *params)
                # (the LayerNorm's backward writes its input gradient in the zero-padded layout this conv's backward reads)
                y, _ = F.layer_norm(v, ln.weight, ln.bias, ln.eps, act=1, grad_pad=F.conv_grad_pad(y.shape[1], spec[0], spec[1]))
            return y
        # default mode: GroupNorm(C, C) normalises every channel over time, so block 0's Conv1d bias cancels exactly in
        # the forward and has a zero gradient; it is accepted (checkpoint compatibility) and left without a gradient
        gn = blk0[2]
        y = F.Conv0Fn.apply(x, conv0.weight, gn.weight, gn.bias, self.specs[0][1], gn.eps, wdt)
        params = [blk[0].weight for blk in rest] + ([blk[0].bias for blk in rest] if has_bias else [])
        if rest:
            y = F.infer_apply(F.ConvStackFn, y, tuple(self.specs[1:]), True, *params)
        return y


# ----------------------------------------------------------------------------------------------- attention
class MultiheadAttention(_DerivedOwner, nn.Module):
    """Self-attention with the gated relative position bias (WavLM/modules.py:303-563, the 'fast path' that all
    WavLM checkpoints were trained with).  Submodule names / order follow the reference so state dicts and seeded
    initialisation line up."""

    def __init__(self, embed_dim, num_heads, dropout=0.0, has_relative_attention_bias=False, num_buckets=32,
                 max_distance=128, gru_rel_pos=False):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.dropout_module = nn.Dropout(dropout)
        self.has_relative_attention_bias = has_relative_attention_bias
        self.num_buckets = num_buckets
        self.max_distance = max_distance
        if has_relative_attention_bias:
            self.relative_attention_bias = nn.Embedding(num_buckets, num_heads)
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == embed_dim, "embed_dim must be divisible by num_heads"
        self.scaling = self.head_dim ** -0.5
        self.k_proj = nn.Linear(embed_dim, embed_dim, bias=True)
        self.v_proj = nn.Linear(embed_dim, embed_dim, bias=True)
        self.q_proj = nn.Linear(embed_dim, embed_dim, bias=True)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=True)
        self._packed = None  # (W[3D, D], dW sink, b[3D], db sink): set by a flat-arena optimizer, see packed_param_groups
        self.gru_rel_pos = gru_rel_pos
        if gru_rel_pos:
            self.grep_linear = nn.Linear(self.head_dim, 8)
            self.grep_a = nn.Parameter(torch.ones(1, num_heads, 1, 1))
        self._tag_pack_owner()
        self.reset_parameters()

    def _tag_pack_owner(self):
        # an optimizer built from a bare parameter list (the fairseq Trainer's) finds the packed q|k|v groups through this tag
        for lin in (self.q_proj, self.k_proj, self.v_proj):
            lin.weight._wl_pack_owner = self
            lin.bias._wl_pack_owner = self

    def __setstate__(self, state):
        """copy.deepcopy / pickle (EMA copies, fairseq's model copying): nn.Parameter.__deepcopy__ drops Python attributes,
        so the copy's q|k|v parameters are re-tagged with THEIR owner, and views into the original's arenas are dropped
        (the copy packs again when an optimizer is built over it)"""
        super().__setstate__(state)
        self.__dict__.pop("_packed_w", None)
        self.__dict__.pop("_packed_b", None)
        self.__dict__.pop("_wl_binding", None)
        self._packed = None
        self._tag_pack_owner()

    def _apply(self, fn, recurse=True):
        """module conversion (.to / .cuda / .bfloat16) may replace the Parameter objects (overwrite_module_params_on_conversion):
        tag whatever objects are there afterwards"""
        out = super()._apply(fn, recurse)
        self._tag_pack_owner()
        return out

    def reset_parameters(self):
        g = 1 / math.sqrt(2)
        nn.init.xavier_uniform_(self.k_proj.weight, gain=g)
        nn.init.xavier_uniform_(self.v_proj.weight, gain=g)
        nn.init.xavier_uniform_(self.q_proj.weight, gain=g)
        nn.init.xavier_uniform_(self.out_proj.weight)
        nn.init.constant_(self.out_proj.bias, 0.0)
        if self.has_relative_attention_bias:
            nn.init.xavier_normal_(self.relative_attention_bias.weight)

    def packed_param_groups(self):
        """parameter groups a flat-arena optimizer should lay out contiguously, with the binder for the packed views"""
        D = self.embed_dim

        def bind_w(pv, gv):
            self._packed_w = (pv.view(3 * D, D), gv.view(3 * D, D))
            self._refresh_packed()

        def bind_b(pv, gv):
            self._packed_b = (pv.view(3 * D), gv.view(3 * D))
            self._refresh_packed()

        return [([self.q_proj.weight, self.k_proj.weight, self.v_proj.weight], bind_w),
                ([self.q_proj.bias, self.k_proj.bias, self.v_proj.bias], bind_b)]

    def _refresh_packed(self):
        w, b = getattr(self, "_packed_w", None), getattr(self, "_packed_b", None)
        self._packed = (w[0], w[1], b[0], b[1]) if (w is not None and b is not None) else None

    def position_table(self, T, device):
        """[H, 2T-1] fp32 Toeplitz generator of compute_bias(T, T) (modules.py:444-455)"""
        # the bucket line depends only on (T, num_buckets, max_distance): computed on the host exactly as the reference
        # does (bit-exact) once per sequence length and kept on the device (the reference rebuilds the [T, T] index
        # matrix on the CPU and uploads it every step, modules.py:179-190)
        key = (int(T), str(device))
        cache = self.__dict__.setdefault("_bucket_cache", {})
        bucket = cache.get(key)
        if bucket is None:
            if len(cache) > 64:
                cache.clear()
            bucket = relative_position_buckets(T, self.num_buckets, self.max_distance).to(device)
            cache[key] = bucket
        return F.RelPosTableFn.apply(self.relative_attention_bias.weight, bucket)

    def forward(self, x, key_padding_u8=None, position_table=None, out_bias_tok=None, wgroup=None, chain=False):
        """x [B, T, D] -> (attn_out [B, T, D], position_table).  chain=True: a third value, an alias of x that has passed
        through the gate and the q|k|v projection (functional.LinearFn pass_x): the caller uses it wherever it would have
        used x again (the residual), so that x has ONE consumer chain and backward needs no separate add kernels."""
        B, T, D = x.shape
        if self.has_relative_attention_bias and position_table is None:
            position_table = self.position_table(T, x.device)
        gate = None
        if position_table is not None:
            if self.gru_rel_pos:
                if chain:
                    gate, x = F.GateFn.apply(x, self.grep_linear.weight, self.grep_linear.bias, self.grep_a, self.num_heads, True)
                else:
                    gate = F.GateFn.apply(x, self.grep_linear.weight, self.grep_linear.bias, self.grep_a, self.num_heads)
            else:
                gate = torch.ones((B, self.num_heads, T), dtype=torch.float32, device=x.device)
        pk = self._packed
        if (pk is not None and torch.is_grad_enabled() and pk[0].data_ptr() == self.q_proj.weight.data_ptr()
                and pk[2].data_ptr() == self.q_proj.bias.data_ptr() and pk[0].dtype == x.dtype):
            # optimizer-bound packed views of q|k|v (no concatenation; the gradient lands packed in the arena)
            # the fused attention backward delivers the packed bias gradient (it produces dq | dk | dv anyway)
            tq = F.BiasGradToken(pk[2])
            qkv = F.LinearFn.apply(x, pk[0], pk[2], pk[1], pk[3], tq, wgroup, chain)
        else:
            tq = None
            srcs = [self.q_proj.weight, self.k_proj.weight, self.v_proj.weight, self.q_proj.bias, self.k_proj.bias, self.v_proj.bias]
            w, b = F.eval_derived(srcs, "qkv_packed", lambda: (torch.cat(srcs[:3], dim=0), torch.cat(srcs[3:], dim=0)))
            qkv = F.LinearFn.apply(x, w, b, None, None, None, None, chain)
        if chain:
            qkv, x = qkv
        p = self.dropout_module.p if self.training else 0.0
        o = F.AttnCoreFn.apply(qkv, gate, position_table, key_padding_u8, self.num_heads, self.scaling, p,
                               F.next_seed() if p > 0 else 0, tq, pk[3] if tq is not None else None)
        out = F.LinearFn.apply(o, self.out_proj.weight, self.out_proj.bias, None, None, out_bias_tok, wgroup)
        return (out, position_table, x) if chain else (out, position_table)


PRELN_FUSED = os.environ.get("WAVLM_PRELN_FUSED", "1") != "0"  # pre-LN blocks: residual adds inside the LayerNorms


class ResidualAddFn(torch.autograd.Function):
    """y = x + dropout(r) (pre-LN blocks, where no LayerNorm follows the add)"""

    @staticmethod
    def forward(ctx, x, r, p, seed):
        from . import ops
        xc, rc = x.contiguous(), r.contiguous()
        if xc.numel() % 8 == 0:
            y = ops.dropout_add(xc, rc, p, seed)   # one pass instead of clone + dropout + axpby
        else:
            y = xc.clone()
            ops.axpby_(y, ops.dropout(rc, p, seed) if p > 0 else rc, 1.0, 1.0)
        ctx.p, ctx.seed = p, seed
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        dr = ops.dropout(dy.contiguous(), ctx.p, ctx.seed) if ctx.p > 0 else dy
        return dy, dr, None, None


ACTIVATION_FNS = ("relu", "gelu", "gelu_accurate", "tanh", "linear", "glu")  # utils.get_available_activation_fns


class GLULinear(nn.Module):
    """Parameter holder of GLU_Linear(input_dim, output_dim, "swish") (WavLM/modules.py:99-129): `linear` maps to 2 x
    output_dim, the output is x[..., :F] * swish(x[..., F:]).  The arithmetic lives in functional.FFNFn (act="glu")."""

    def __init__(self, input_dim, output_dim):
        super().__init__()
        self.output_dim = output_dim
        self.linear = nn.Linear(input_dim, output_dim * 2)


class TransformerSentenceEncoderLayer(nn.Module):
    """post-LN (Base) / pre-LN (Large) encoder block (WavLM/WavLM.py:615-742)"""

    def __init__(self, embedding_dim=768, ffn_embedding_dim=3072, num_attention_heads=8, dropout=0.1,
                 attention_dropout=0.1, activation_dropout=0.1, activation_fn="relu", layer_norm_first=False,
                 has_relative_attention_bias=False, num_buckets=0, max_distance=0, gru_rel_pos=False):
        super().__init__()
        if activation_fn == "gelu_fast":  # deprecated name of gelu_accurate (src/fairseq/utils.py:541-545)
            activation_fn = "gelu_accurate"
        if activation_fn not in ACTIVATION_FNS:
            raise RuntimeError("--activation-fn {} not supported".format(activation_fn))  # utils.get_activation_fn
        self.activation_name = activation_fn
        self.embedding_dim = embedding_dim
        self.dropout = dropout
        self.activation_dropout = activation_dropout
        self.self_attn = MultiheadAttention(embedding_dim, num_attention_heads, dropout=attention_dropout,
                                            has_relative_attention_bias=has_relative_attention_bias,
                                            num_buckets=num_buckets, max_distance=max_distance, gru_rel_pos=gru_rel_pos)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(activation_dropout)
        self.dropout3 = nn.Dropout(dropout)
        self.layer_norm_first = layer_norm_first
        self.self_attn_layer_norm = nn.LayerNorm(embedding_dim)
        if activation_fn == "glu":  # WavLM/WavLM.py:668-669: fc1 = GLU_Linear(D, F, "swish"), parameters fc1.linear.*
            self.fc1 = GLULinear(embedding_dim, ffn_embedding_dim)
        else:
            self.fc1 = nn.Linear(embedding_dim, ffn_embedding_dim)
        self.fc2 = nn.Linear(ffn_embedding_dim, embedding_dim)
        self.final_layer_norm = nn.LayerNorm(embedding_dim)

    def _ffn(self, x, b2_tok=None, wgroup=None, chain=False):
        p = self.activation_dropout if self.training else 0.0
        fc1 = self.fc1.linear if self.activation_name == "glu" else self.fc1
        return F.infer_apply(F.FFNFn, x, fc1.weight, fc1.bias, self.fc2.weight, self.fc2.bias, p,
                             F.next_seed() if p > 0 else 0, b2_tok, wgroup, chain, self.activation_name)

    def forward_preln_fused(self, x, pending, key_padding_u8=None, position_table=None):
        """Pre-LN block with both residual adds fused into the LayerNorm that follows them (training path of the encoder,
        no taps): `pending` = (f, bias_token) is the previous block's feed-forward output, still to be added to x.
        LN1 computes x + dropout(f) and its normalisation in one kernel, LN2 does the same for the attention output; the
        LayerNorm backward kernels deliver the gradients of both branches through the dropout mask AND the out_proj / fc2
        bias gradients (column sums) -- the separate dropout-add, dropout and column-sum passes of the unfused form are
        gone.  Returns (x, pending', position_table) with this block's own feed-forward output pending."""
        ln1, ln2 = self.self_attn_layer_norm, self.final_layer_norm
        p = self.dropout if self.training else 0.0
        if layerfn.eligible(self, x, position_table):   # the whole block as one node / one C call each way
            tf = F.BiasGradToken(self.fc2.bias)
            x, f, position_table = layerfn.run_block(self, x, key_padding_u8, position_table,
                                                     r_in=None if pending is None else pending[0],
                                                     prev_tok=None if pending is None else pending[1], out_tok=tf)
            return x, (f, tf), position_table
        wg = self._wgrad_group()
        if pending is None:
            h, _, x = F.layer_norm(x, ln1.weight, ln1.bias, ln1.eps, pass_x=True)
        else:
            f_prev, tok_prev = pending
            h, x = F.layer_norm(x, ln1.weight, ln1.bias, ln1.eps, residual=f_prev, p_in=p, training=self.training,
                                residual_bias_tok=tok_prev, s_grad=True)
        ta = F.BiasGradToken(self.self_attn.out_proj.bias)
        a, position_table, _h = self.self_attn(h, key_padding_u8, position_table, out_bias_tok=ta, wgroup=wg, chain=True)
        h, x = F.layer_norm(x, ln2.weight, ln2.bias, ln2.eps, residual=a, p_in=p, training=self.training,
                            residual_bias_tok=ta, s_grad=True)
        tf = F.BiasGradToken(self.fc2.bias)
        f = self._ffn(h, tf, wg)
        return x, (f, tf), position_table

    def _wgrad_group(self):
        """the layer's weight gradients (q|k|v packed, out_proj, fc1, fc2) as one grouped launch in backward"""
        if not torch.is_grad_enabled():
            return None
        return F.WgradGroup(4 if self.self_attn._packed is not None else 3)

    def forward(self, x, key_padding_u8=None, position_table=None, tbc=False):
        """x [B, T, D] -> (x, None, position_table).  tbc=True: x comes and goes as the reference's [T, B, D] (a transposed
        VIEW of the channel-last tensor, no copy) -- the encoder calls the layer that way when forward hooks are registered
        on it, so that hook-based consumers written for the reference (s3prl's UpstreamExpert:
        downstreams/speaker_verification/models/utils.py:49-56 reads `input[0].transpose(0, 1)` of every layer) see the
        shapes they expect."""
        if tbc:
            x = x.transpose(0, 1)
        ln1, ln2 = self.self_attn_layer_norm, self.final_layer_norm
        p = self.dropout if self.training else 0.0
        if self.layer_norm_first:
            wg = self._wgrad_group()
            chain = torch.is_grad_enabled() and F.CHAIN_CONSUMERS
            if chain:
                # x feeds the LayerNorm and the residual add: the LayerNorm hands back an alias and adds the residual
                # stream's gradient inside its backward kernel.  h feeds the gate and the q|k|v projection: chained too
                # (the alias of h that comes back is not needed here)
                h, _, x = F.layer_norm(x, ln1.weight, ln1.bias, ln1.eps, pass_x=True)
                a, position_table, _h = self.self_attn(h, key_padding_u8, position_table, wgroup=wg, chain=True)
            else:
                h, _ = F.layer_norm(x, ln1.weight, ln1.bias, ln1.eps)
                a, position_table = self.self_attn(h, key_padding_u8, position_table, wgroup=wg)
            x = ResidualAddFn.apply(x, a, p, F.next_seed() if p > 0 else 0)
            if chain:
                h, _, x = F.layer_norm(x, ln2.weight, ln2.bias, ln2.eps, pass_x=True)
            else:
                h, _ = F.layer_norm(x, ln2.weight, ln2.bias, ln2.eps)
            f = self._ffn(h, wgroup=wg)
            x = ResidualAddFn.apply(x, f, p, F.next_seed() if p > 0 else 0)
        elif layerfn.eligible(self, x, position_table):   # the whole block as one node / one C call each way
            x, _, position_table = layerfn.run_block(self, x, key_padding_u8, position_table)
        else:
            # post-LN: the LayerNorm that follows a sub-layer also delivers the bias gradient of its last linear
            grad = torch.is_grad_enabled()
            wg = self._wgrad_group()
            ta = F.BiasGradToken(self.self_attn.out_proj.bias) if grad else None
            chain = grad and F.CHAIN_CONSUMERS  # x / the first LayerNorm's output: one consumer chain each (no add kernels)
            if chain:
                a, position_table, x = self.self_attn(x, key_padding_u8, position_table, out_bias_tok=ta, wgroup=wg, chain=True)
            else:
                a, position_table = self.self_attn(x, key_padding_u8, position_table, out_bias_tok=ta, wgroup=wg)
            x, _ = F.layer_norm(x, ln1.weight, ln1.bias, ln1.eps, residual=a, p_in=p, training=self.training,
                                residual_bias_tok=ta)
            tf = F.BiasGradToken(self.fc2.bias) if grad else None
            if chain:
                f, x = self._ffn(x, tf, wg, chain=True)
            else:
                f = self._ffn(x, tf, wg)
            x, _ = F.layer_norm(x, ln2.weight, ln2.bias, ln2.eps, residual=f, p_in=p, training=self.training,
                                residual_bias_tok=tf)
        return (x.transpose(0, 1) if tbc else x), None, position_table


class TransformerEncoder(_DerivedOwner, nn.Module):
    """pos_conv + N layers (WavLM/WavLM.py:507-612; fairseq twin src/fairseq/models/wavlm/wavlm.py:630-754)"""

    def __init__(self, args):
        super().__init__()
        self.dropout = args.dropout
        self.embedding_dim = args.encoder_embed_dim
        self.conv_pos = args.conv_pos
        self.conv_pos_groups = args.conv_pos_groups
        pos_conv = nn.Conv1d(self.embedding_dim, self.embedding_dim, kernel_size=args.conv_pos,
                             padding=args.conv_pos // 2, groups=args.conv_pos_groups)
        std = math.sqrt(4.0 / (args.conv_pos * self.embedding_dim))
        nn.init.normal_(pos_conv.weight, mean=0, std=std)
        nn.init.constant_(pos_conv.bias, 0)
        # old-style weight norm (weight_g / weight_v parameters) for checkpoint-key compatibility
        pos_conv = nn.utils.weight_norm(pos_conv, name="weight", dim=2)
        self.pos_conv = nn.Sequential(pos_conv, nn.Identity(), nn.GELU())
        self.relative_position_embedding = getattr(args, "relative_position_embedding", False)
        self.num_buckets = getattr(args, "num_buckets", 0) if self.relative_position_embedding else 0
        self.max_distance = getattr(args, "max_distance", 0) if self.relative_position_embedding else 0
        gru = getattr(args, "gru_rel_pos", False)
        self.layers = nn.ModuleList([
            TransformerSentenceEncoderLayer(
                embedding_dim=self.embedding_dim, ffn_embedding_dim=args.encoder_ffn_embed_dim,
                num_attention_heads=args.encoder_attention_heads, dropout=self.dropout,
                attention_dropout=args.attention_dropout, activation_dropout=args.activation_dropout,
                activation_fn=args.activation_fn, layer_norm_first=args.layer_norm_first,
                has_relative_attention_bias=(self.relative_position_embedding and i == 0),
                num_buckets=self.num_buckets, max_distance=self.max_distance, gru_rel_pos=gru)
            for i in range(args.encoder_layers)
        ])
        self.layer_norm_first = args.layer_norm_first
        self.layer_norm = nn.LayerNorm(self.embedding_dim)
        if self.layer_norm_first and getattr(args, "utterance_contrastive_loss", False):
            # UniSpeech-SAT's encoder (unispeech_sat.py:1195-1197): the speaker tap of a pre-LN stack gets its own final
            # LayerNorm.  Created only for SAT configs so that WavLM-Large state dicts still load strictly.
            self.layer_norm_for_extract = nn.LayerNorm(self.embedding_dim)
        self.layerdrop = args.encoder_layerdrop
        self.apply(init_bert_params)

    def forward(self, x, padding_mask=None, layer=None, fairseq_layer_results=False, prezeroed=False, extract_layer=None):
        """extract_layer (0-based): also return that layer's output [B, T, D] (UniSpeech-SAT's speaker tap,
        models/unispeech_sat/unispeech_sat.py:1202-1255) as a 4th value"""
        # pre-LN training path without taps: residual adds fused into the LayerNorms (forward_preln_fused); the last block's
        # feed-forward output is added by the final LayerNorm
        fuse = (self.layer_norm_first and layer is None  # (layer None: no per-layer results)
                and torch.is_grad_enabled() and F.CHAIN_CONSUMERS and PRELN_FUSED
                and not any(l._forward_hooks or l._forward_pre_hooks for l in self.layers))
        x, layer_results, pre_ln = self.extract_features(x, padding_mask, layer, fairseq_layer_results, prezeroed,
                                                         extract_layer=extract_layer, fuse_preln=fuse)
        if self.layer_norm_first and layer is None:
            pend = self._pending
            self._pending = None
            if pend is not None:
                x, _ = F.layer_norm(x, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps, residual=pend[0],
                                    p_in=self.dropout if self.training else 0.0, training=self.training,
                                    residual_bias_tok=pend[1])
            else:
                x, _ = F.layer_norm(x, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps)
        if extract_layer is None:
            return x, layer_results, pre_ln
        er = self._extract_result
        self._extract_result = None
        if er is not None and self.layer_norm_first and layer is None:
            ln = getattr(self, "layer_norm_for_extract", None)
            if ln is None:
                raise NotImplementedError("pre-LN encoder with a speaker tap needs layer_norm_for_extract parameters")
            er, _ = F.layer_norm(er, ln.weight, ln.bias, ln.eps)
        return x, layer_results, pre_ln, er

    _extract_result = None
    _pending = None

    def extract_features(self, x, padding_mask=None, tgt_layer=None, fairseq_layer_results=False, prezeroed=False,
                         extract_layer=None, fuse_preln=False):
        """x [B, T, D].  Returns (x, layer_results, conv_sum) where conv_sum = x + pos_conv(x): the tensor the
        reference's in-place `x += x_conv` leaves behind in `features` (WavLM/WavLM.py:579)."""
        kpm = None
        if padding_mask is not None:
            kpm = padding_mask.to(torch.uint8).contiguous()
            if not prezeroed:
                x = F.SelectRowsFn.apply(x, None, None, kpm.view(-1))
        conv = self.pos_conv[0]
        xs = F.infer_apply(F.PosConvFn, x, conv.weight_v, conv.weight_g, conv.bias, self.conv_pos_groups)
        if not self.layer_norm_first:
            ln = self.layer_norm
            x, _ = F.layer_norm(xs, ln.weight, ln.bias, ln.eps, p_out=self.dropout, training=self.training)
        else:
            x = F.dropout(xs, self.dropout, self.training)

        layer_results = []
        if tgt_layer is not None and not fairseq_layer_results:
            layer_results.append((x.transpose(0, 1), None))
        r = None
        table = None
        pending = None
        self._pending = None
        for i, layer in enumerate(self.layers):
            # one host draw per layer, training or not: keeps the numpy stream aligned with the reference
            dropout_probability = np.random.random()
            if not self.training or (dropout_probability > self.layerdrop):
                if fuse_preln:
                    x, pending, table = layer.forward_preln_fused(x, pending, kpm, table)
                    z = None
                elif layer._forward_hooks or layer._forward_pre_hooks:
                    xt, z, table = layer(x.transpose(0, 1), kpm, table, tbc=True)
                    x = xt.transpose(0, 1)
                else:
                    x, z, table = layer(x, kpm, table)
            else:
                z = None
            if extract_layer is not None and i == extract_layer:
                if pending is not None:
                    # fused pre-LN path: the last executed block's feed-forward branch is still pending.  The speaker tap
                    # is the residual stream AFTER that whole block, whether layer i itself ran or was dropped
                    # (unispeech_sat.py:1238-1247 records `x` after the layerdrop branch): add the branch here (its bias
                    # token stays untaken, fc2 computes its own bias gradient) and go on un-pended
                    pd = self.layers[0].dropout if self.training else 0.0
                    x = ResidualAddFn.apply(x, pending[0], pd, F.next_seed() if pd > 0 else 0)
                    pending = None
                self._extract_result = x
            if fairseq_layer_results:
                if isinstance(tgt_layer, list) and i + 1 in tgt_layer:
                    layer_results.append((x.transpose(0, 1), z))
                elif isinstance(tgt_layer, int) and i == tgt_layer:
                    r = x
                    break
            else:
                if tgt_layer is not None:
                    layer_results.append((x.transpose(0, 1), z))
                if i == tgt_layer:
                    r = x
                    break
        if r is not None:
            x = r
        self._pending = pending  # (fused pre-LN path: the caller adds it inside the final LayerNorm)
        return x, layer_results, xs


# -------------------------------------------------------------------------------------------------- model
class WavLM(nn.Module):
    """Drop-in for WavLM/WavLM.py:220-375 (`extract_features` API), HIP-only."""

    def __init__(self, cfg: WavLMConfig) -> None:
        super().__init__()
        self.cfg = cfg
        feature_enc_layers = eval(cfg.conv_feature_layers)
        self.embed = feature_enc_layers[-1][0]
        self.feature_extractor = ConvFeatureExtractionModel(conv_layers=feature_enc_layers, dropout=0.0,
                                                            mode=cfg.extractor_mode, conv_bias=cfg.conv_bias)
        self.post_extract_proj = (nn.Linear(self.embed, cfg.encoder_embed_dim)
                                  if self.embed != cfg.encoder_embed_dim else None)
        self.mask_prob = cfg.mask_prob
        self.mask_selection = cfg.mask_selection
        self.mask_other = cfg.mask_other
        self.mask_length = cfg.mask_length
        self.no_mask_overlap = cfg.no_mask_overlap
        self.mask_min_space = cfg.mask_min_space
        self.mask_channel_prob = cfg.mask_channel_prob
        self.mask_channel_selection = getattr(cfg, "mask_channel_selection", "static")
        self.mask_channel_other = getattr(cfg, "mask_channel_other", 0)
        self.mask_channel_length = getattr(cfg, "mask_channel_length", 10)
        self.no_mask_channel_overlap = getattr(cfg, "no_mask_channel_overlap", False)
        self.mask_channel_min_space = getattr(cfg, "mask_channel_min_space", 1)
        self.dropout_input = nn.Dropout(cfg.dropout_input)
        self.dropout_features = nn.Dropout(cfg.dropout_features)
        self.feature_grad_mult = cfg.feature_grad_mult
        self.mask_emb = nn.Parameter(torch.FloatTensor(cfg.encoder_embed_dim).uniform_())
        self.encoder = TransformerEncoder(cfg)
        self.layer_norm = nn.LayerNorm(self.embed)

    def half(self):
        """the reference recipes pass --fp16 (trainer.py:86-89 then calls model.half()); the gfx950 kernels compute in
        bf16 (MFMA, fp32 accumulate) or fp32.  Default: fail here, not at the first kernel launch.  With the explicit
        fp16-as-bf16 switch (unispeech_amd/precision.py) the model becomes bf16."""
        from . import precision
        if precision.fp16_as_bf16():
            return self.to(torch.bfloat16)
        raise NotImplementedError(precision.MESSAGE)

    # -- host-side pieces -------------------------------------------------------------------------------------
    def compute_mask(self, B, T, padding_mask):
        """bool numpy [B, T]; consumes the global numpy RNG exactly like apply_mask (WavLM.py:271-285)"""
        if self.mask_prob <= 0:
            return None
        return compute_mask_indices((B, T), padding_mask, self.mask_prob, self.mask_length, self.mask_selection,
                                    self.mask_other, min_masks=2, no_overlap=self.no_mask_overlap,
                                    min_space=self.mask_min_space)

    def apply_channel_mask(self, x):
        """apply_mask's second half (WavLM.py:287-304 / wavlm.py:405-422): a [B, C] span mask drawn from the same numpy
        stream right after the time mask, expanded over time; masked channels are zeroed (also where mask_emb sits)."""
        if self.mask_channel_prob <= 0:
            return x
        B, _, C = x.shape
        ch = compute_mask_indices((B, C), None, self.mask_channel_prob, self.mask_channel_length,
                                  self.mask_channel_selection, self.mask_channel_other,
                                  no_overlap=self.no_mask_channel_overlap, min_space=self.mask_channel_min_space)
        keep = F.h2d(np.logical_not(np.asarray(ch)).astype(np.float32), x.device).to(x.dtype)
        return x * keep.view(B, 1, C)

    def forward_padding_mask(self, n_frames: int, padding_mask: torch.Tensor) -> torch.Tensor:
        """frame f is padded iff all of its samples are (WavLM/WavLM.py:306-314)"""
        if padding_mask.device.type == "cpu" and padding_mask.dtype == torch.bool:
            # Host copy of the mask (the launch thread's own bookkeeping): numpy, single-threaded.  The torch CPU reduction of
            # a [32, 749, 320] mask is an OpenMP parallel region -- on a 256-core host it wakes 256 spinning threads every
            # step, and inside a container with a CPU quota (16 CPUs on the GPU boxes) that exhausts the cgroup's quota within
            # ~10 ms of every 100 ms period: the WHOLE process is then throttled for the rest of the period (measured: 90 ms
            # stalls of the launch thread every 2-3 steps, 52 instead of 34.5 ms per step; profiles/r04/host_stalls.txt).
            a = padding_mask.numpy()
            B, T = a.shape
            if not a.any():   # nothing padded (the common batch of a crop-to-shortest recipe): 0.4 ms instead of 0.9
                return torch.zeros((B, n_frames), dtype=torch.bool)
            k = T // n_frames
            if a.flags.c_contiguous and k % 8 == 0 and T % 8 == 0:
                # eight samples per 64-bit word: a frame is padded iff every byte of its k / 8 words is non-zero.  A bool
                # byte is 0x01 in anything numpy / torch produce, but a mask viewed from foreign uint8 storage may carry
                # other non-zero bytes: x | x >> 1 | ... | x >> 7 folds every byte's bits into its bit 0 (carries from the
                # byte above land in bits 1-7, which the 0x01 mask drops)
                w = a.view(np.uint64)[:, :n_frames * (k // 8)].reshape(B, n_frames, k // 8)
                if a.view(np.uint8).max() > 1:
                    w = w | (w >> np.uint64(1)); w = w | (w >> np.uint64(2)); w = w | (w >> np.uint64(4))
                out = ((w & np.uint64(0x0101010101010101)) == np.uint64(0x0101010101010101)).all(-1)
            else:
                out = a[:, :n_frames * k].reshape(B, n_frames, k).all(-1)
            return torch.from_numpy(out)
        extra = padding_mask.size(1) % n_frames
        if extra > 0:
            padding_mask = padding_mask[:, :-extra]
        padding_mask = padding_mask.view(padding_mask.size(0), n_frames, -1)
        return padding_mask.all(-1)

    @property
    def feat_grad_scale(self):
        """GradMultiply factor applied to every gradient entering the extractor (WavLM.py:333-336)"""
        return self.feature_grad_mult if (self.feature_grad_mult > 0 and self.feature_grad_mult != 1.0) else 1.0

    # -- device path -------------------------------------------------------------------------------------------
    def _features(self, source):
        """waveform -> (LayerNorm'ed, projected features [B, T', D], raw conv features [B, T', C])"""
        if self.feature_grad_mult > 0:
            feats = self.feature_extractor(source)
        else:
            with torch.no_grad():
                feats = self.feature_extractor(source)
        gscale = self.feat_grad_scale
        ln = self.layer_norm
        x, _ = F.layer_norm(feats, ln.weight, ln.bias, ln.eps, grad_scale=gscale)
        if self.post_extract_proj is not None:
            x = F.LinearFn.apply(x, self.post_extract_proj.weight, self.post_extract_proj.bias)
        return x, feats

    def extract_features(self, source: torch.Tensor, padding_mask: Optional[torch.Tensor] = None, mask: bool = False,
                         ret_conv: bool = False, output_layer: Optional[int] = None,
                         ret_layer_results: bool = False):
        x, _ = self._features(source)
        B, T, _ = x.shape
        if padding_mask is not None:
            padding_mask = self.forward_padding_mask(T, padding_mask)
        x = F.dropout(x, self.dropout_input.p, self.training)
        sel = None
        if mask:
            m = self.compute_mask(B, T, padding_mask)
            if m is not None:
                sel = F.h2d(np.asarray(m).astype(np.uint8), x.device).view(-1)
        kpm = padding_mask.to(torch.uint8).contiguous().view(-1) if padding_mask is not None else None
        if sel is not None or kpm is not None:
            x = F.SelectRowsFn.apply(x, sel, self.mask_emb if sel is not None else None, kpm)
        if mask:
            x = self.apply_channel_mask(x)
        x, layer_results, conv_sum = self.encoder(
            x, padding_mask=padding_mask, layer=None if output_layer is None else output_layer - 1, prezeroed=True)
        # the reference returns `features` after three in-place updates (mask, padding zero-fill, += pos_conv)
        feature = conv_sum if ret_conv else x
        if ret_layer_results:
            feature = (feature, layer_results)
        return feature, padding_mask
