"""wav2vec 2.0 / UniSpeech pre-training model and criterion on the MI355X kernels: the sampled-negatives variant of the
cosine contrastive loss.

Mirrors src/fairseq/models/wav2vec/wav2vec2.py:274-766 (`Wav2Vec2Model`: constructor order -> state-dict keys and seeded
initialisation, forward() arguments, result keys, get_logits / get_targets / get_extra_losses / quantize /
remove_pretraining_modules), src/fairseq/modules/gumbel_vector_quantizer.py (`GumbelVectorQuantizer`) and
src/fairseq/criterions/wav2vec_criterion.py:37-123 (`Wav2vecCriterion`, infonce).  The conv extractor, the transformer
encoder (no relative position bias) and every kernel are the ones of the WavLM path (unispeech_amd.wavlm).

What differs from the reference is how the numbers are produced: the [N+1, B, T_m, C] gathered-negatives tensor is never
built (gathered cosine logits + fused cross entropy, functional.SampledNegativesLossFn), the quantiser's [n, G, V]
one-hot x codebook product is a gather, and index lists come from the host-generated mask (no device nonzero()).

negatives_from_everywhere and codebook_negatives (wav2vec2.py:653-686) are built on the same gathered-row head (the candidate
table grows by one projected row per frame / per sampled code).  Options of the reference that no shipped recipe uses and
that raise NotImplementedError: transpose and the non-infonce (BCE)
criterion (it pairs with wav2vec 1.0's get_targets / get_target_weights, models/wav2vec/wav2vec.py, outside SURVEY.md 8).
"""
import ast
import math
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import functional as F
from .masking import compute_mask_indices
from .wavlm import ConvFeatureExtractionModel, TransformerEncoder


@dataclass
class Wav2Vec2Config:
    """field names and defaults of fairseq's Wav2Vec2Config (models/wav2vec/wav2vec2.py:71-271)"""
    extractor_mode: str = "default"
    encoder_layers: int = 12
    encoder_embed_dim: int = 768
    encoder_ffn_embed_dim: int = 3072
    encoder_attention_heads: int = 12
    activation_fn: str = "gelu"
    dropout: float = 0.1
    attention_dropout: float = 0.1
    activation_dropout: float = 0.0
    encoder_layerdrop: float = 0.0
    dropout_input: float = 0.0
    dropout_features: float = 0.0
    final_dim: int = 0
    layer_norm_first: bool = False
    conv_feature_layers: str = "[(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512,2,2)] + [(512,2,2)]"
    conv_bias: bool = False
    logit_temp: float = 0.1
    quantize_targets: bool = False
    quantize_input: bool = False
    same_quantizer: bool = False
    target_glu: bool = False
    feature_grad_mult: float = 1.0
    quantizer_depth: int = 1
    quantizer_factor: int = 3
    latent_vars: int = 320
    latent_groups: int = 2
    latent_dim: int = 0
    mask_length: int = 10
    mask_prob: float = 0.65
    mask_selection: str = "static"
    mask_other: float = 0
    no_mask_overlap: bool = False
    mask_min_space: int = 1
    mask_channel_length: int = 10
    mask_channel_prob: float = 0.0
    mask_channel_before: bool = False
    mask_channel_selection: str = "static"
    mask_channel_other: float = 0
    no_mask_channel_overlap: bool = False
    mask_channel_min_space: int = 1
    num_negatives: int = 100
    negatives_from_everywhere: bool = False
    cross_sample_negatives: int = 0
    codebook_negatives: int = 0
    conv_pos: int = 128
    conv_pos_groups: int = 16
    latent_temp: tuple = (2, 0.5, 0.999995)
    transpose: bool = False
    # accepted for the shared encoder class; wav2vec 2.0 has no relative position bias
    relative_position_embedding: bool = False
    num_buckets: int = 0
    max_distance: int = 0
    gru_rel_pos: bool = False


class GumbelVectorQuantizer(nn.Module):
    """GumbelVectorQuantizer(dim, num_vars, temp, groups, combine_groups=False, vq_dim, time_first=True) of the
    reference (modules/gumbel_vector_quantizer.py:13-213), weight_proj_depth 1.  State-dict keys: vars,
    weight_proj.{weight,bias}.  gumbel_noise: 'device' (counter hash, default) | 'host' (the reference's CPU draws)."""

    def __init__(self, dim, num_vars, temp, groups, combine_groups, vq_dim, time_first, weight_proj_depth=1,
                 weight_proj_factor=1):
        super().__init__()
        if combine_groups or not time_first:
            raise NotImplementedError("GumbelVectorQuantizer: only combine_groups=False, time_first=True")
        self.groups, self.combine_groups, self.input_dim, self.num_vars, self.time_first = groups, False, dim, num_vars, True
        assert vq_dim % groups == 0, f"dim {vq_dim} must be divisible by groups {groups} for concatenation"
        var_dim = vq_dim // groups
        self.vars = nn.Parameter(torch.FloatTensor(1, groups * num_vars, var_dim))
        nn.init.uniform_(self.vars)
        if weight_proj_depth > 1:  # Linear + GELU blocks in front of the logits projection (default nn.Linear init, as the reference)
            inner_dim = self.input_dim * weight_proj_factor
            self.weight_proj = nn.Sequential(
                *[nn.Sequential(nn.Linear(self.input_dim if i == 0 else inner_dim, inner_dim), nn.GELU())
                  for i in range(weight_proj_depth - 1)],
                nn.Linear(inner_dim, groups * num_vars))
        else:
            self.weight_proj = nn.Linear(self.input_dim, groups * num_vars)
            nn.init.normal_(self.weight_proj.weight, mean=0, std=1)
            nn.init.zeros_(self.weight_proj.bias)
        if isinstance(temp, str):
            temp = ast.literal_eval(temp)
        assert len(temp) == 3, f"{temp}, {len(temp)}"
        self.max_temp, self.min_temp, self.temp_decay = temp
        self.curr_temp = self.max_temp
        self.gumbel_noise = "device"

    def set_num_updates(self, num_updates):
        self.curr_temp = max(self.max_temp * self.temp_decay ** num_updates, self.min_temp)

    def forward(self, x, produce_targets=False):
        """x [B, T, C] -> {"x": [B, T, vq_dim], "prob_perplexity", "code_perplexity", "num_vars", "temp", ("targets")}"""
        B, T, C = x.shape
        G, V = self.groups, self.num_vars
        if isinstance(self.weight_proj, nn.Sequential):
            h = x.reshape(B * T, C)
            for blk in list(self.weight_proj)[:-1]:
                h = F.ActFn.apply(F.LinearFn.apply(h, blk[0].weight, blk[0].bias), "gelu")
            logits = F.LinearFn.apply(h, self.weight_proj[-1].weight, self.weight_proj[-1].bias)
        else:
            logits = F.LinearFn.apply(x.reshape(B * T, C), self.weight_proj.weight, self.weight_proj.bias)
        noise = None
        if self.training and self.gumbel_noise == "host":
            noise = F.h2d(F.host_gumbel_noise(B * T * G, V), x.device)
        q, prob, code = F.GumbelVQFn.apply(logits, self.vars, G, V, float(self.curr_temp), self.training, noise,
                                           F.next_seed() if (self.training and noise is None) else 0)
        res = {"num_vars": V * G, "temp": self.curr_temp, "prob_perplexity": prob.reshape(()),
               "code_perplexity": code.reshape(()), "x": q.view(B, T, -1)}
        if produce_targets:
            raise NotImplementedError("produce_targets (code indices) is only used by offline quantisation")
        return res

    def sample_from_codebook(self, b, n):
        """b x n random full codes (one index per group, groups concatenated) -> [b, n, vq_dim]
        (gumbel_vector_quantizer.py:90-128: the same torch.randint draw on the CPU generator; differentiable in `vars`)"""
        G, V = self.groups, self.num_vars
        cb_size = V ** G
        assert n < cb_size, f"sample size {n} is greater than size of codebook {cb_size}"
        sample_idx = torch.randint(low=0, high=cb_size, size=(b * n,))
        # row r of the reference's index table = digits of r in base V (itertools.product order) + V * group
        digits = torch.stack([(sample_idx // (V ** (G - 1 - g))) % V + V * g for g in range(G)], dim=1)
        flat = F.h2d(digits.reshape(-1).to(torch.int64), self.vars.device)
        return self.vars.squeeze(0).index_select(0, flat).view(b, n, -1)

    def burn_host_noise(self, n_rows):
        """advance the CPU generator as one more forward on n_rows rows would (the reference quantises the unmasked
        features a second time, wav2vec2.py:655-656, a result only its `transpose` option reads)"""
        if self.training and self.gumbel_noise == "host":
            F.host_gumbel_noise(n_rows * self.groups, self.num_vars)


class Wav2Vec2Model(nn.Module):
    def __init__(self, cfg: Wav2Vec2Config):
        super().__init__()
        if getattr(cfg, "transpose", False):
            raise NotImplementedError("wav2vec 2.0 option transpose is not supported by the HIP path")
        self.negatives_from_everywhere = bool(getattr(cfg, "negatives_from_everywhere", False))
        self.codebook_negatives = int(getattr(cfg, "codebook_negatives", 0))
        if self.codebook_negatives > 0 and not cfg.quantize_targets:
            raise ValueError("codebook_negatives needs quantize_targets (the reference dereferences self.quantizer)")
        self.cfg = cfg
        # parameter creation order of the reference constructor (wav2vec2.py:276-395): seeded-init parity
        layers = eval(cfg.conv_feature_layers)
        self.embed = layers[-1][0]
        self.feature_extractor = ConvFeatureExtractionModel(conv_layers=layers, dropout=0.0, mode=cfg.extractor_mode,
                                                            conv_bias=cfg.conv_bias)
        self.post_extract_proj = (nn.Linear(self.embed, cfg.encoder_embed_dim)
                                  if self.embed != cfg.encoder_embed_dim and not cfg.quantize_input else None)
        self.mask_prob, self.mask_selection, self.mask_other = cfg.mask_prob, cfg.mask_selection, cfg.mask_other
        self.mask_length, self.no_mask_overlap, self.mask_min_space = cfg.mask_length, cfg.no_mask_overlap, cfg.mask_min_space
        self.mask_channel_prob = cfg.mask_channel_prob
        self.mask_channel_before = getattr(cfg, "mask_channel_before", False)
        self.mask_channel_selection, self.mask_channel_other = cfg.mask_channel_selection, cfg.mask_channel_other
        self.mask_channel_length = cfg.mask_channel_length
        self.no_mask_channel_overlap, self.mask_channel_min_space = cfg.no_mask_channel_overlap, cfg.mask_channel_min_space
        self.dropout_input = nn.Dropout(cfg.dropout_input)
        self.dropout_features = nn.Dropout(cfg.dropout_features)
        self.feature_grad_mult = cfg.feature_grad_mult
        self.quantizer = None
        self.n_negatives = cfg.num_negatives
        self.cross_sample_negatives = cfg.cross_sample_negatives
        self.logit_temp = cfg.logit_temp
        final_dim = cfg.final_dim if cfg.final_dim > 0 else cfg.encoder_embed_dim
        if cfg.quantize_targets:
            vq_dim = cfg.latent_dim if cfg.latent_dim > 0 else final_dim
            self.quantizer = GumbelVectorQuantizer(dim=self.embed, num_vars=cfg.latent_vars, temp=cfg.latent_temp,
                                                   groups=cfg.latent_groups, combine_groups=False, vq_dim=vq_dim,
                                                   time_first=True, weight_proj_depth=cfg.quantizer_depth,
                                                   weight_proj_factor=cfg.quantizer_factor)
            self.project_q = nn.Linear(vq_dim, final_dim)
        else:
            self.project_q = nn.Linear(self.embed, final_dim)
        self.input_quantizer = None
        if cfg.quantize_input:  # wav2vec2.py:346-363: the encoder input is a (projected) codebook entry
            if cfg.same_quantizer and self.quantizer is not None:
                vq_dim = final_dim
                self.input_quantizer = self.quantizer
            else:
                vq_dim = cfg.latent_dim if cfg.latent_dim > 0 else cfg.encoder_embed_dim
                self.input_quantizer = GumbelVectorQuantizer(dim=self.embed, num_vars=cfg.latent_vars, temp=cfg.latent_temp,
                                                             groups=cfg.latent_groups, combine_groups=False, vq_dim=vq_dim,
                                                             time_first=True, weight_proj_depth=cfg.quantizer_depth,
                                                             weight_proj_factor=cfg.quantizer_factor)
            self.project_inp = nn.Linear(vq_dim, cfg.encoder_embed_dim)
        self.mask_emb = nn.Parameter(torch.FloatTensor(cfg.encoder_embed_dim).uniform_())
        self.encoder = TransformerEncoder(cfg)
        self.layer_norm = nn.LayerNorm(self.embed)
        self.target_glu = None
        if cfg.target_glu:  # wav2vec2.py:371-375
            self.target_glu = nn.Sequential(nn.Linear(final_dim, final_dim * 2), nn.GLU())
        self.final_proj = nn.Linear(cfg.encoder_embed_dim, final_dim)

    @classmethod
    def build_model(cls, cfg, task=None):
        return cls(cfg)

    def upgrade_state_dict_named(self, state_dict, name):
        return state_dict

    def set_num_updates(self, num_updates):
        self.num_updates = num_updates
        if self.quantizer is not None:
            self.quantizer.set_num_updates(num_updates)
        if self.input_quantizer is not None and self.input_quantizer is not self.quantizer:
            self.input_quantizer.set_num_updates(num_updates)

    def max_positions(self):
        return None

    def half(self):
        from . import precision
        if precision.fp16_as_bf16():   # explicit switch: an unmodified --fp16 recipe on bf16 kernels (precision.py)
            return self.to(torch.bfloat16)
        raise NotImplementedError(precision.MESSAGE)

    @property
    def feat_grad_scale(self):
        return self.feature_grad_mult if (self.feature_grad_mult > 0 and self.feature_grad_mult != 1.0) else 1.0

    def _channel_mask(self, x):
        B, _, C = x.shape
        ch = compute_mask_indices((B, C), None, self.mask_channel_prob, self.mask_channel_length,
                                  self.mask_channel_selection, self.mask_channel_other,
                                  no_overlap=self.no_mask_channel_overlap, min_space=self.mask_channel_min_space)
        keep = F.h2d(np.logical_not(np.asarray(ch)).astype(np.float32), x.device).to(x.dtype)
        return x * keep.view(B, 1, C)

    def forward(self, source, padding_mask=None, mask=True, features_only=False, layer=None, mask_indices=None,
                mask_channel_indices=None, padding_count=None, padding_mask_cpu=None) -> Dict[str, torch.Tensor]:
        if mask_channel_indices is not None:
            raise NotImplementedError("externally supplied channel masks")
        # -- extractor, features_pen, LayerNorm (wav2vec2.py:567-581)
        if self.feature_grad_mult > 0:
            feats = self.feature_extractor(source)
        else:
            with torch.no_grad():
                feats = self.feature_extractor(source)
        gscale = self.feat_grad_scale
        features_pen = F.FeaturesPenFn.apply(feats, gscale) if not features_only else None
        ln = self.layer_norm
        normed, _ = F.layer_norm(feats, ln.weight, ln.bias, ln.eps, grad_scale=gscale)
        B, T, _ = normed.shape
        dev = normed.device
        pad_cpu = None
        if padding_mask is not None:
            extra = padding_mask.size(1) % T
            pm = padding_mask[:, :-extra] if extra > 0 else padding_mask
            padding_mask = pm.view(pm.size(0), T, -1).all(-1)
            if padding_mask_cpu is not None:
                pc = padding_mask_cpu[:, :-extra] if extra > 0 else padding_mask_cpu
                pad_cpu = pc.view(pc.size(0), T, -1).all(-1)
            else:
                pad_cpu = padding_mask.cpu()
        x = normed
        if self.post_extract_proj is not None:
            x = F.LinearFn.apply(x, self.post_extract_proj.weight, self.post_extract_proj.bias)
        x = F.dropout(x, self.dropout_input.p, self.training)
        unmasked = F.dropout(normed, self.dropout_features.p, self.training)
        inq = None
        if self.input_quantizer is not None:  # wav2vec2.py:602-609
            inq = self.input_quantizer(x)
            x = F.LinearFn.apply(inq["x"].reshape(B * T, -1), self.project_inp.weight, self.project_inp.bias).view(B, T, -1)
        # -- masking (apply_mask, wav2vec2.py:405-472): same numpy draws in the same order
        mask_np = None
        if mask:
            if self.mask_channel_prob > 0 and self.mask_channel_before:
                x = self._channel_mask(x)
            if self.mask_prob > 0:
                if mask_indices is None:
                    mask_np = compute_mask_indices((B, T), pad_cpu, self.mask_prob, self.mask_length, self.mask_selection,
                                                   self.mask_other, min_masks=2, no_overlap=self.no_mask_overlap,
                                                   min_space=self.mask_min_space)
                else:
                    mask_np = np.asarray(mask_indices.cpu() if torch.is_tensor(mask_indices) else mask_indices, dtype=bool)
        sel = F.h2d(mask_np.astype(np.uint8), dev).view(-1) if mask_np is not None else None
        kpm = padding_mask.to(torch.uint8).contiguous().view(-1) if padding_mask is not None else None
        if sel is not None or kpm is not None:
            x = F.SelectRowsFn.apply(x, sel, self.mask_emb if sel is not None else None, kpm)
        if mask and self.mask_channel_prob > 0 and not self.mask_channel_before:
            x = self._channel_mask(x)
        x, layer_results, _ = self.encoder(x, padding_mask=padding_mask, layer=layer, fairseq_layer_results=True,
                                           prezeroed=True)
        if features_only:
            return {"x": x, "padding_mask": padding_mask, "features": unmasked, "layer_results": layer_results}
        if mask_np is None:
            raise NotImplementedError("the pre-training forward needs a time mask (mask=True, mask_prob > 0)")

        # -- targets y = (quantised) unmasked features at the masked frames (wav2vec2.py:617-664)
        idx_np = np.flatnonzero(mask_np.reshape(-1)).astype(np.int32)
        S = int(idx_np.size)
        Tm = S // B
        assert Tm * B == S, "rows must hold equal numbers of masked frames (compute_mask_indices guarantees it)"
        inv_np = np.full(B * T, -1, dtype=np.int32)
        inv_np[idx_np] = np.arange(S, dtype=np.int32)
        idx, inv = F.h2d(idx_np, dev), F.h2d(inv_np, dev)
        y = F.GatherRowsFn.apply(unmasked.reshape(B * T, -1), idx, inv)                # [S, C], row = b * Tm + t
        result = {"features": x, "feature_padding_mask": padding_mask}
        if inq is not None:  # overwritten by the target quantiser's values below, as in the reference
            result.update(prob_perplexity=inq["prob_perplexity"], code_perplexity=inq["code_perplexity"],
                          num_vars=inq["num_vars"], temp=inq["temp"])
        nfe, cbn = self.negatives_from_everywhere, self.codebook_negatives
        cand = None  # candidates of negatives_from_everywhere: one projected row per frame [B * T, F]
        if self.quantizer is not None:
            q = self.quantizer(y.view(B, Tm, -1))
            y = q["x"].reshape(S, -1)
            result.update(prob_perplexity=q["prob_perplexity"], code_perplexity=q["code_perplexity"],
                          num_vars=q["num_vars"], temp=q["temp"])
            if nfe:  # the reference's second quantiser pass over ALL frames supplies the candidates (wav2vec2.py:653-661)
                q_all = self.quantizer(unmasked.reshape(B, T, -1))["x"].reshape(B * T, -1)
                cand = F.LinearFn.apply(q_all, self.project_q.weight, self.project_q.bias)
            else:
                self.quantizer.burn_host_noise(B * T)
            y = F.LinearFn.apply(y, self.project_q.weight, self.project_q.bias)          # [S, F]
            own = torch.arange(S)
            table = y if cand is None else torch.cat([y, cand], dim=0)
            neg_base = 0 if cand is None else S
        elif nfe:
            # negatives from every frame of the un-quantised features (wav2vec2.py:679-686): project_q is applied per row, so
            # the positives are rows of the same projected table
            table = F.LinearFn.apply(unmasked.reshape(B * T, -1), self.project_q.weight, self.project_q.bias)
            own, neg_base = torch.from_numpy(idx_np.astype(np.int64)), 0
        else:
            y = F.LinearFn.apply(y, self.project_q.weight, self.project_q.bias)          # [S, F]
            own, table, neg_base = torch.arange(S), y, 0
        neg = F.sample_negatives_indices(B, T if nfe else Tm, Tm, self.n_negatives, self.cross_sample_negatives, padding_count)
        N = self.n_negatives + self.cross_sample_negatives
        cols = [own.view(S, 1), neg.view(B, Tm, N).reshape(S, N) + neg_base]
        if cbn > 0:
            # codebook negatives (wav2vec2.py:669-677): S * cbn random full codes through project_q; the reference views the
            # [S, cbn, .] samples as [cbn, B, Tm, .] ("order doesnt matter"): negative k of position s is flat sample k * S + s
            codes = self.quantizer.sample_from_codebook(S, cbn).reshape(S * cbn, -1)
            cb = F.LinearFn.apply(codes.to(table.dtype), self.project_q.weight, self.project_q.bias)
            cols.append(table.shape[0] + torch.arange(cbn).view(1, cbn) * S + torch.arange(S).view(S, 1))
            table = torch.cat([table, cb], dim=0)
        xs = F.GatherRowsFn.apply(x.reshape(B * T, -1), idx, inv)
        xs = F.LinearFn.apply(xs, self.final_proj.weight, self.final_proj.bias)      # [S, F]
        idx_full = F.h2d(torch.cat(cols, dim=1).to(torch.int32), dev)
        if self.target_glu is not None:  # y and the negatives alike (wav2vec2.py:697-699): per row of the candidate table
            tg = self.target_glu[0]
            table = F.GLUFn.apply(F.LinearFn.apply(table.contiguous(), tg.weight, tg.bias))
        y = table
        loss, ncorrect = F.SampledNegativesLossFn.apply(xs, y, idx_full, self.logit_temp)
        result["head"] = {"loss": loss, "correct": ncorrect, "count": S, "x": xs, "y": y, "idx": idx_full, "B": B, "Tm": Tm}
        result["x"] = None   # reference-shaped logits [N+1, B, Tm] are materialised on demand by get_logits()
        result["padding_mask"] = padding_mask
        result["features_pen"] = features_pen
        return result

    # -- reference surface ----------------------------------------------------------------------------------------
    def quantize(self, x):
        raise NotImplementedError("offline quantisation (forward_idx) is not part of the training hot path")

    def extract_features(self, source, padding_mask, mask=False, layer=None):
        return self.forward(source, padding_mask, mask=mask, features_only=True, layer=layer)

    def get_logits(self, net_output):
        """[T_m * B, N + 1] rows ordered (t, b) like the reference's logits.transpose(0, 2).reshape(-1, N + 1)
        (wav2vec2.py:738-741); -inf where a negative equals the positive"""
        h = net_output["head"]
        xn, _ = F.ops.l2norm_fwd(h["x"].detach().contiguous(), h["x"].dtype)
        yn, _ = F.ops.l2norm_fwd(h["y"].detach().contiguous(), h["y"].dtype)
        lg = F.ops.gather_dot(xn, yn, h["idx"], 1.0 / self.logit_temp, mask_equal=True)   # [S, N+1], row = b * Tm + t
        B, Tm = h["B"], h["Tm"]
        return lg.view(B, Tm, -1).transpose(0, 1).reshape(B * Tm, -1).float()

    def get_targets(self, sample, net_output, expand_steps=True):
        return torch.zeros(net_output["head"]["count"], dtype=torch.long, device=net_output["head"]["x"].device)

    def get_extra_losses(self, net_output):
        pen = []
        if "prob_perplexity" in net_output:
            pen.append((net_output["num_vars"] - net_output["prob_perplexity"]) / net_output["num_vars"])
        if "features_pen" in net_output:
            pen.append(net_output["features_pen"])
        return pen

    def remove_pretraining_modules(self):
        self.quantizer = None
        self.project_q = None
        self.target_glu = None
        self.final_proj = None


class Wav2vecCriterion(nn.Module):
    """criterion 'wav2vec' with --infonce (criterions/wav2vec_criterion.py:37-123): cross entropy against class 0 of the
    [S, N+1] logits, sum-reduced, + loss_weights x extra losses x sample_size; logging keys loss, ntokens, nsentences,
    sample_size, loss_0.., correct, count.  defer_logging keeps the values on the device."""

    def __init__(self, task=None, infonce=False, loss_weights=None, log_keys=None, defer_logging=False):
        if not hasattr(self, "_modules"):
            nn.Module.__init__(self)
        if not infonce:
            raise NotImplementedError("only the --infonce form of the wav2vec criterion is built (every wav2vec 2.0 recipe uses it)")
        self.task, self.infonce, self.loss_weights = task, infonce, loss_weights
        self.log_keys = [] if log_keys is None else log_keys
        self.defer_logging = defer_logging

    def forward(self, model, sample, reduce=True, log_pred=False):
        net_output = model(**sample["net_input"])
        return self.get_loss(model, sample, net_output, reduce, log_pred)

    def get_loss(self, model, sample, net_output, reduce=True, log_pred=False):
        if not reduce:
            raise NotImplementedError("the fused loss is sum-reduced")
        num = (lambda t: t) if self.defer_logging else (lambda t: t.item())
        h = net_output["head"]
        loss = h["loss"][0]
        sample_size = h["count"]
        losses = [loss.detach().clone()]
        if self.loss_weights is not None:
            extra = model.get_extra_losses(net_output)
            weights = list(self.loss_weights)
            if len(weights) == 1 and len(extra) != 1:
                weights = [weights[0]] * len(extra)
            assert len(extra) == len(weights), f"{len(extra)}, {len(weights)}"
            for p, coef in zip(extra, weights):
                if coef != 0 and p is not None:
                    p = coef * p.float().reshape(()) * sample_size
                    loss = loss + p
                    losses.append(p.detach())
        nsent = sample["id"].numel() if "id" in sample else sample["net_input"]["source"].size(0)
        log = {"loss": num(loss.detach()), "ntokens": sample_size, "nsentences": nsent, "sample_size": sample_size}
        for lk in self.log_keys:
            if lk in net_output and net_output[lk] is not None:
                log[lk] = float(net_output[lk])
        if len(losses) > 1:
            for i, l in enumerate(losses):
                log[f"loss_{i}"] = num(l)
        log["correct"] = num(h["correct"][0]) if self.defer_logging else int(h["correct"].item())
        log["count"] = sample_size
        if log_pred:
            log["logits"] = model.get_logits(net_output).cpu().numpy()
            log["target"] = np.zeros(sample_size, dtype=np.int64)
        return loss, sample_size, log

    @staticmethod
    def reduce_metrics(logging_outputs, log_scalar=None) -> Dict[str, float]:
        """aggregation of wav2vec_criterion.py:138-196; returns the scalars (forwarded to metrics.log_scalar if given)"""
        def val(v):
            return float(v.item()) if torch.is_tensor(v) else float(v)
        out = {}
        loss_sum = sum(val(l.get("loss", 0)) for l in logging_outputs)
        sample_size = sum(val(l.get("sample_size", 0)) for l in logging_outputs)
        out["loss"] = loss_sum / sample_size / math.log(2)
        out["ntokens"] = sum(val(l.get("ntokens", 0)) for l in logging_outputs)
        out["nsentences"] = sum(val(l.get("nsentences", 0)) for l in logging_outputs)
        correct = sum(val(l.get("correct", 0)) for l in logging_outputs)
        total = sum(val(l.get("count", 0)) for l in logging_outputs)
        out["_correct"], out["_total"] = correct, total
        if total > 0:
            out["accuracy"] = correct / total
        builtin = {"loss", "ntokens", "nsentences", "sample_size", "correct", "count"}
        for k in logging_outputs[0]:
            if k not in builtin:
                v = sum(val(l.get(k, 0)) for l in logging_outputs) / len(logging_outputs)
                out[k] = v / sample_size / math.log(2) if k.startswith("loss") else v
        if log_scalar is not None:
            for k, v in out.items():
                log_scalar(k, v)
        return out

    @staticmethod
    def logging_outputs_can_be_summed() -> bool:
        """False, as the reference: reduce_metrics averages the non-builtin keys over len(logging_outputs)"""
        return False
