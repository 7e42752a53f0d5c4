"""CPU oracle of the WavLM / UniSpeech pre-training hot path.

TEST INFRASTRUCTURE ONLY.  This module is a plain fp32 PyTorch-on-CPU restatement of the reference algorithm.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it, and only as the checker / the timed
CPU baseline -- never as part of the product path (unispeech_amd/ imports nothing from here and has no CPU fallback).

Why a restatement and not the reference itself: /root/reference does not exist on the GPU box, so parity tests
there need a self-contained checker.  Pinning: tests/test_oracle_vs_golden.py checks every function here against
fixtures in tests/golden/ that oracle/gen_golden.py produced by running the *reference's own Python*
(WavLM/WavLM.py and src/fairseq WavLMModel + WavLMCriterion) in the build container, and -- when /root/reference is
present -- tests/test_oracle_vs_reference.py re-runs the reference live at WavLM-Base width (12 layers, d=768, padded row).  The reference ships no
golden vectors or tests of its own, so those reference-generated fixtures are the pin.

All functions take a parameter dict `sd` keyed exactly like the reference state_dict and a config object with the
reference's field names.  Shapes follow the reference ([B, C, T] inside the extractor, [T, B, C] inside the encoder).
Arithmetic living in PyTorch itself (conv1d, group_norm, layer_norm, gelu, softmax, cosine_similarity,
cross_entropy) is used through the same torch.nn.functional entry points the reference calls.
"""
import math

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------- feature extractor
def conv_feature_extractor(sd, cfg, source, prefix="feature_extractor."):
    """ConvFeatureExtractionModel.forward, conv_type 'default' (WavLM/WavLM.py:485-504; blocks 391-428):
    x.unsqueeze(1); per block Conv1d(bias=conv_bias) -> Dropout(0) -> [GroupNorm(dim, dim) on block 0 in 'default'
    mode | LayerNorm over channels on every block in 'layer_norm' mode, both evaluated in fp32] -> nn.GELU()."""
    layers = eval(cfg.conv_feature_layers)
    x = source.unsqueeze(1)
    for i, (dim, k, stride) in enumerate(layers):
        w = sd[f"{prefix}conv_layers.{i}.0.weight"]
        b = sd.get(f"{prefix}conv_layers.{i}.0.bias")
        x = F.conv1d(x, w, b, stride=stride)
        if cfg.extractor_mode == "layer_norm":
            g = sd[f"{prefix}conv_layers.{i}.2.1.weight"]
            bt = sd[f"{prefix}conv_layers.{i}.2.1.bias"]
            x = F.layer_norm(x.transpose(-2, -1).float(), (dim,), g.float(), bt.float(), 1e-5).type_as(x)
            x = x.transpose(-2, -1)
        elif i == 0:
            g = sd[f"{prefix}conv_layers.0.2.weight"]
            bt = sd[f"{prefix}conv_layers.0.2.bias"]
            x = F.group_norm(x.float(), dim, g.float(), bt.float(), 1e-5).type_as(x)
        x = F.gelu(x)
    return x  # [B, C, T']


def forward_padding_mask(n_frames, padding_mask):
    """WavLM.forward_padding_mask (WavLM/WavLM.py:311-321)"""
    extra = padding_mask.size(1) % n_frames
    if extra > 0:
        padding_mask = padding_mask[:, :-extra]
    return padding_mask.view(padding_mask.size(0), n_frames, -1).all(-1)


# ---------------------------------------------------------------------------------------------------- attention
def relative_positions_bucket(relative_positions, num_buckets, max_distance):
    """MultiheadAttention._relative_positions_bucket, bidirectional (WavLM/modules.py:417-442)"""
    num_buckets = num_buckets // 2
    buckets = (relative_positions > 0).to(torch.long) * num_buckets
    n = torch.abs(relative_positions)
    max_exact = num_buckets // 2
    is_small = n < max_exact
    if_large = max_exact + (
        torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (num_buckets - max_exact)
    ).to(torch.long)
    if_large = torch.min(if_large, torch.full_like(if_large, num_buckets - 1))
    return buckets + torch.where(is_small, n, if_large)


def compute_bias(emb_weight, q_len, k_len, num_buckets, max_distance):
    """MultiheadAttention.compute_bias (WavLM/modules.py:444-455): [H, q_len, k_len]"""
    ctx = torch.arange(q_len, dtype=torch.long)[:, None]
    mem = torch.arange(k_len, dtype=torch.long)[None, :]
    bucket = relative_positions_bucket(mem - ctx, num_buckets, max_distance)
    return F.embedding(bucket, emb_weight).permute([2, 0, 1])


def self_attention(sd, pre, x, key_padding_mask, position_bias, cfg, first_layer_emb=None):
    """MultiheadAttention.forward fast path (WavLM/modules.py:504-564) with the F.multi_head_attention_forward
    call written out: q/k/v projections, q scaled by head_dim**-0.5, additive float mask = gated position bias,
    key padding -> -inf, softmax in fp32, out_proj.  x: [T, B, D].  Returns (out [T, B, D], position_bias)."""
    T, B, D = x.shape
    H = cfg.encoder_attention_heads
    hd = D // H
    if position_bias is None and first_layer_emb is not None:
        pb = compute_bias(first_layer_emb, T, T, cfg.num_buckets, cfg.max_distance)
        position_bias = pb.unsqueeze(0).repeat(B, 1, 1, 1).view(B * H, T, T)
    mask = None
    if position_bias is not None:
        mask = position_bias
        if cfg.gru_rel_pos:
            ql = x.transpose(0, 1).view(B, T, H, hd).permute(0, 2, 1, 3)  # un-projected input, per head
            lin = F.linear(ql, sd[pre + "grep_linear.weight"], sd[pre + "grep_linear.bias"])
            gate_a, gate_b = torch.sigmoid(lin.view(B, H, T, 2, 4).sum(-1)).chunk(2, dim=-1)
            gate = gate_a * (gate_b * sd[pre + "grep_a"] - 1.0) + 2.0
            mask = gate.view(B * H, -1, 1) * position_bias
        mask = mask.view(-1, T, T)
    q = F.linear(x, sd[pre + "q_proj.weight"], sd[pre + "q_proj.bias"])
    k = F.linear(x, sd[pre + "k_proj.weight"], sd[pre + "k_proj.bias"])
    v = F.linear(x, sd[pre + "v_proj.weight"], sd[pre + "v_proj.bias"])
    q = q.contiguous().view(T, B * H, hd).transpose(0, 1) * (hd ** -0.5)
    k = k.contiguous().view(T, B * H, hd).transpose(0, 1)
    v = v.contiguous().view(T, B * H, hd).transpose(0, 1)
    w = torch.bmm(q, k.transpose(1, 2))
    if mask is not None:
        w = w + mask
    if key_padding_mask is not None:
        w = w.view(B, H, T, T).masked_fill(key_padding_mask.unsqueeze(1).unsqueeze(2), float("-inf")).view(B * H, T, T)
    p = F.softmax(w, dim=-1)
    o = torch.bmm(p, v).transpose(0, 1).contiguous().view(T, B, D)
    o = F.linear(o, sd[pre + "out_proj.weight"], sd[pre + "out_proj.bias"])
    return o, position_bias


def encoder_layer(sd, pre, x, key_padding_mask, position_bias, cfg, first_layer_emb=None):
    """TransformerSentenceEncoderLayer.forward with all dropouts 0 (WavLM/WavLM.py:677-742); gelu upcasts to fp32
    (WavLM/modules.py:140-141)."""
    D = x.shape[-1]

    def ln(name, t):
        return F.layer_norm(t, (D,), sd[pre + name + ".weight"], sd[pre + name + ".bias"], 1e-5)

    def ffn(t):
        # activation_fn (WavLM/modules.py:144-160 get_activation_fn; "glu": fc1 = GLU_Linear(D, F, "swish"),
        # WavLM/modules.py:99-129, WavLM/WavLM.py:668-669, 707-708, followed by the identity)
        act = getattr(cfg, "activation_fn", "gelu")
        if act == "glu":
            h = F.linear(t, sd[pre + "fc1.linear.weight"], sd[pre + "fc1.linear.bias"])
            Fd = h.shape[-1] // 2
            g = h[..., Fd:]
            h = h[..., :Fd] * (g * torch.sigmoid(g))
        else:
            h = F.linear(t, sd[pre + "fc1.weight"], sd[pre + "fc1.bias"])
            if act == "gelu":
                h = F.gelu(h.float()).type_as(h)
            elif act == "relu":
                h = F.relu(h)
            elif act in ("gelu_accurate", "gelu_fast"):  # src/fairseq/modules/gelu.py:14-19
                h = 0.5 * h * (1 + torch.tanh(math.sqrt(2 / math.pi) * (h + 0.044715 * torch.pow(h, 3))))
            elif act == "tanh":
                h = torch.tanh(h)
            elif act != "linear":
                raise RuntimeError("--activation-fn {} not supported".format(act))
        return F.linear(h, sd[pre + "fc2.weight"], sd[pre + "fc2.bias"])

    residual = x
    if cfg.layer_norm_first:
        a, position_bias = self_attention(sd, pre + "self_attn.", ln("self_attn_layer_norm", x), key_padding_mask,
                                          position_bias, cfg, first_layer_emb)
        x = residual + a
        x = x + ffn(ln("final_layer_norm", x))
    else:
        a, position_bias = self_attention(sd, pre + "self_attn.", x, key_padding_mask, position_bias, cfg,
                                          first_layer_emb)
        x = ln("self_attn_layer_norm", residual + a)
        x = ln("final_layer_norm", x + ffn(x))
    return x, position_bias


def pos_conv(sd, cfg, x, prefix="encoder."):
    """weight_norm(dim=2) Conv1d(D, D, k, padding=k//2, groups) -> SamePad -> GELU (WavLM/WavLM.py:514-527;
    SamePad WavLM/modules.py:72-83).  x: [B, T, D] -> [B, T, D]"""
    g = sd[prefix + "pos_conv.0.weight_g"]
    v = sd[prefix + "pos_conv.0.weight_v"]
    w = g * v / v.norm(dim=(0, 1), keepdim=True)  # torch._weight_norm(v, g, dim=2)
    y = F.conv1d(x.transpose(1, 2), w, sd[prefix + "pos_conv.0.bias"], padding=cfg.conv_pos // 2,
                 groups=cfg.conv_pos_groups)
    if cfg.conv_pos % 2 == 0:
        y = y[:, :, :-1]
    return F.gelu(y).transpose(1, 2)


def transformer_encoder(sd, cfg, x, padding_mask=None, tgt_layer=None, prefix="encoder.", extract_layer=None, taps=None,
                        collect=None, collected=None):
    """TransformerEncoder.forward / extract_features, eval-equivalent (dropout, layerdrop = 0)
    (WavLM/WavLM.py:564-612).  x: [B, T, D].  Returns (x [B, T, D], layer_results [(x_l [T,B,D], None)],
    conv_sum [B, T, D])."""
    D = x.shape[-1]
    if padding_mask is not None:
        x = x.masked_fill(padding_mask.unsqueeze(-1), 0.0)
    x = x + pos_conv(sd, cfg, x, prefix)  # out-of-place form of `x += x_conv` (value-identical forward)
    conv_sum = x
    if not cfg.layer_norm_first:
        x = F.layer_norm(x, (D,), sd[prefix + "layer_norm.weight"], sd[prefix + "layer_norm.bias"], 1e-5)
    x = x.transpose(0, 1)
    layer_results = []
    if tgt_layer is not None:
        layer_results.append((x, None))
    pos_bias = None
    emb = sd.get(prefix + "layers.0.self_attn.relative_attention_bias.weight") \
        if getattr(cfg, "relative_position_embedding", False) else None
    r = None
    for i in range(cfg.encoder_layers):
        x, pos_bias = encoder_layer(sd, f"{prefix}layers.{i}.", x, padding_mask, pos_bias, cfg,
                                    first_layer_emb=emb if i == 0 else None)
        if tgt_layer is not None:
            layer_results.append((x, None))
        if extract_layer is not None and i == extract_layer and taps is not None:
            taps.append(x.transpose(0, 1))  # UniSpeech-SAT speaker tap (unispeech_sat.py:1243-1244)
        if collect is not None and (i + 1) in collect:
            collected.append(x.transpose(0, 1))  # ILS: `isinstance(tgt_layer, list) and i+1 in tgt_layer` (wavlm.py:731)
        if i == tgt_layer:
            r = x
            break
    if r is not None:
        x = r
    x = x.transpose(0, 1)
    if cfg.layer_norm_first and tgt_layer is None and collect is None:  # `layer is None` (wavlm.py:699-701): a tapped-layer
        x = F.layer_norm(x, (D,), sd[prefix + "layer_norm.weight"], sd[prefix + "layer_norm.bias"], 1e-5)  # list skips it
        if taps:  # pre-LN UniSpeech-SAT: the speaker tap gets its own final LayerNorm (unispeech_sat.py:1197, 1205-1208)
            taps[0] = F.layer_norm(taps[0], (D,), sd[prefix + "layer_norm_for_extract.weight"],
                                   sd[prefix + "layer_norm_for_extract.bias"], 1e-5)
    return x, layer_results, conv_sum


# ---------------------------------------------------------------------------------------------------- model
def project_features(sd, cfg, source):
    """extractor -> transpose -> LayerNorm(C) -> post_extract_proj (WavLM/WavLM.py:333-348).
    Returns (projected [B, T', D], raw conv features [B, C, T'])"""
    feats = conv_feature_extractor(sd, cfg, source)
    x = feats.transpose(1, 2)
    C = x.shape[-1]
    x = F.layer_norm(x, (C,), sd["layer_norm.weight"], sd["layer_norm.bias"], 1e-5)
    if "post_extract_proj.weight" in sd:
        x = F.linear(x, sd["post_extract_proj.weight"], sd["post_extract_proj.bias"])
    return x, feats


def extract_features(sd, cfg, source, padding_mask=None, mask_indices=None, output_layer=None):
    """WavLM.extract_features (WavLM/WavLM.py:323-375) with the mask given explicitly (bool [B, T'] or None).
    Returns dict(x, padding_mask, features, layer_results) -- `features` carries the reference's three in-place
    updates (mask_emb fill, padding zero-fill, += pos_conv)."""
    x, _ = project_features(sd, cfg, source)
    if padding_mask is not None:
        padding_mask = forward_padding_mask(x.shape[1], padding_mask)
    if mask_indices is not None:
        x = torch.where(mask_indices.unsqueeze(-1), sd["mask_emb"].view(1, 1, -1), x)
    y, layer_results, conv_sum = transformer_encoder(sd, cfg, x, padding_mask,
                                                     None if output_layer is None else output_layer - 1)
    return {"x": y, "padding_mask": padding_mask, "features": conv_sum, "layer_results": layer_results}


def compute_nce(x, pos, negs, logit_temp):
    """WavLMModel.compute_nce (src/fairseq/models/wavlm/wavlm.py:426-438)"""
    neg_is_pos = (pos == negs).all(-1)
    targets = torch.cat([pos.unsqueeze(0), negs], dim=0)
    logits = torch.cosine_similarity(x.float(), targets.float(), dim=-1).type_as(x)
    logits = logits / logit_temp
    if neg_is_pos.any():
        logits = torch.cat([logits[:1], logits[1:].masked_fill(neg_is_pos, float("-inf"))], dim=0)
    return logits.transpose(0, 1)


def pretrain_forward(sd, cfg, source, target_list, padding_mask, mask_indices, num_classes, chan_mask=None):
    """WavLMModel.forward with dropouts 0 (src/fairseq/models/wavlm/wavlm.py:465-576), mask given explicitly.
    chan_mask: bool [B, C] channel mask of apply_mask's second half (wavlm.py:405-422), applied after the time mask."""
    x, feats = project_features(sd, cfg, source)
    T = x.shape[1]
    layers = eval(cfg.conv_feature_layers)
    ds = 1
    for _, _, s in layers:
        ds *= s
    ratio = cfg.label_rate * ds / 16000
    targ_T = min(t.size(1) for t in target_list)
    if ratio * T > targ_T:
        T = int(targ_T / ratio)
        x = x[:, :T]
        feats = feats[..., :T]
    tinds = (torch.arange(T).float() * ratio).long()
    target_list = [t[:, tinds] for t in target_list]
    features_pen = feats.float().pow(2).mean()
    if padding_mask is not None:
        padding_mask = forward_padding_mask(T, padding_mask)
    if mask_indices is not None:
        x = torch.where(mask_indices.unsqueeze(-1), sd["mask_emb"].view(1, 1, -1), x)
    if chan_mask is not None:
        x = torch.where(chan_mask.unsqueeze(1), torch.zeros((), dtype=x.dtype), x)  # x[mask_channel_indices] = 0
    taps = []
    utt = getattr(cfg, "utterance_contrastive_loss", False)
    pl = getattr(cfg, "predict_layers", "")
    pl = eval(pl) if pl else None
    collected = []
    y, _, conv_sum = transformer_encoder(sd, cfg, x, padding_mask, None,
                                         extract_layer=(cfg.utterance_contrastive_layer - 1) if utt else None, taps=taps,
                                         collect=pl, collected=collected)
    pad = padding_mask if padding_mask is not None else torch.zeros(y.shape[:2], dtype=torch.bool)
    lec = sd["label_embs_concat"]
    sources = collected if pl is not None else [y]  # ILS-SSL (ils_hubert.py:180-250): same head on every collected layer
    if pl is not None and cfg.layer_norm_first:  # pre-LN ILS: one LayerNorm per tapped layer (ils_hubert.py:73-76, 186-187)
        D = sources[0].shape[-1]
        sources = [F.layer_norm(t, (D,), sd["post_layer_norm.%d.weight" % i], sd["post_layer_norm.%d.bias" % i], 1e-5)
                   for i, t in enumerate(sources)]
    sep_emb = getattr(cfg, "separate_label_embeds", False)    # per-layer final_proj + label embeddings (ils_hubert.py:78-86)
    sep_tgt = getattr(cfg, "separate_layer_targets", False)   # label set i belongs to predicted layer i (207-236)

    def pred(sel):
        out = []
        for li, src in enumerate(sources):
            fw = sd["final_proj.%d.weight" % li] if sep_emb else sd["final_proj.weight"]
            fb = sd["final_proj.%d.bias" % li] if sep_emb else sd["final_proj.bias"]
            proj = F.linear(src[sel], fw, fb)
            embs = lec[li] if (sep_emb or sep_tgt) else (lec[0] if lec.dim() == 3 else lec)
            if sep_tgt:
                pairs = [(embs[:num_classes[li]], target_list[li])]
            else:
                pairs = list(zip(embs.split(num_classes, 0), target_list))
            for emb, t in pairs:
                if getattr(cfg, "target_glu", False):  # wavlm.py:322-327, 529-531: Linear(F, 2F) + GLU, row by row
                    emb = F.glu(F.linear(emb, sd["target_glu.0.weight"], sd["target_glu.0.bias"]), dim=-1)
                pos = torch.index_select(emb, 0, t[sel].long())
                negs = emb.unsqueeze(1).expand(-1, proj.size(0), -1)
                out.append(compute_nce(proj, pos, negs, cfg.logit_temp))
        return out

    m = mask_indices if mask_indices is not None else torch.zeros_like(pad)
    out = {"x": y, "features": conv_sum, "padding_mask": padding_mask, "features_pen": features_pen,
           "logit_m_list": pred(torch.logical_and(~pad, m)), "logit_u_list": pred(torch.logical_and(~pad, ~m))}
    if utt:
        out.update(utterance_contrastive(sd, cfg, taps[0], torch.logical_and(~pad, m)))
    if getattr(cfg, "weighted_sum", False):
        out["layer_weights"] = sd["weights"]  # hubert_criterion.py:73-76: per-layer losses x softmax(model.weights)
    return out


def sample_instances(bsz, tsz, num, n_instances, cross_sample_instances):
    """UniSpeechSATModel.sample_instances index draws (unispeech_sat.py:487-537): same torch.randint calls, CPU RNG."""
    high, cross_high = tsz, tsz * bsz
    idxs = cross = None
    if n_instances > 0:
        tszs = torch.arange(num).unsqueeze(-1).expand(-1, n_instances).flatten()
        idxs = torch.randint(low=0, high=high - 1, size=(bsz, n_instances * num))
        idxs[idxs >= tszs] += 1
    if cross_sample_instances > 0:
        tszs = torch.arange(num).unsqueeze(-1).expand(-1, cross_sample_instances).flatten()
        cross = torch.randint(low=0, high=cross_high - 1, size=(bsz, cross_sample_instances * num))
        cross[cross >= tszs] += 1
    if n_instances > 0:
        for i in range(1, bsz):
            idxs[i] += i * high
    else:
        idxs = cross
    if cross_sample_instances > 0 and n_instances > 0:
        idxs = torch.cat([idxs, cross], dim=1)
    return idxs


def utterance_contrastive(sd, cfg, spk_x, masked_indices):
    """compute_pred_spk + compute_nce(replace_inf=False) (unispeech_sat.py:545-557, 701-737), no quantiser."""
    B, _, D = spk_x.shape
    N = cfg.num_instances + cfg.cross_sample_instances
    spk_x_m = spk_x[masked_indices].view(B, -1, D)
    proj = F.linear(spk_x_m, sd["spk_proj.weight"], sd["spk_proj.bias"])         # [B, num, C]
    num = proj.size(1)
    b_pos = torch.arange(B).unsqueeze(1).expand(B, num)
    q = None
    ytgt = proj
    if getattr(cfg, "quantize_targets", False):  # unispeech_sat.py:702-705: targets = project_q(quantizer(tapped frames))
        q = gumbel_vq(sd, "quantizer.", spk_x_m, cfg.latent_groups, cfg.latent_vars, cfg.latent_temp[0], True)
        ytgt = F.linear(q["x"], sd["project_q.weight"], sd["project_q.bias"])
    yflat = ytgt.reshape(-1, ytgt.size(-1))
    idx = sample_instances(B, num, num, cfg.num_instances, cfg.cross_sample_instances)
    samples = yflat[idx.view(-1)].view(B, N, num, -1).permute(1, 0, 2, 3)         # [N, B, num, C]
    samples_b = b_pos.reshape(-1)[idx.view(-1)].view(B, N, num).permute(1, 0, 2)
    x_b = b_pos[..., 0].unsqueeze(1).unsqueeze(0).expand_as(samples_b)
    targets = torch.cat([torch.ones(1, B, num, dtype=torch.long), (samples_b == x_b).long()], dim=0)
    px = proj.reshape(-1, proj.size(-1))
    inst = samples.reshape(N, -1, px.size(-1))
    tg = targets.reshape(N + 1, -1).transpose(0, 1)
    cand = torch.cat([yflat.unsqueeze(0), inst], dim=0)
    logits = (torch.cosine_similarity(px.float(), cand.float(), dim=-1) / cfg.logit_temp).transpose(0, 1)
    loss = F.binary_cross_entropy_with_logits(logits, tg.type_as(logits), reduction="none").mean()
    out = {"loss_spk_m": loss, "mean_targets": tg.float().mean(), "contrastive_acc": ((logits >= 0.0) == tg).float().mean(),
           "spk_logits": logits}
    if q is not None:
        out.update(prob_perplexity=q["prob_perplexity"], num_vars=q["num_vars"])
    return out


def sampled_negatives_logits(x, y, neg_idxs, n_neg, logit_temp):
    """Wav2Vec2Model.sample_negatives' gather + compute_preds (models/wav2vec/wav2vec2.py:521-553): x, y [B, T, C];
    neg_idxs int64 [B, T * N] indexing the flattened [B * T] rows of y.  Returns logits [N + 1, B, T]."""
    B, T, C = y.shape
    negs = y.reshape(-1, C)[neg_idxs.view(-1)].view(B, T, n_neg, C).permute(2, 0, 1, 3)
    neg_is_pos = (y == negs).all(-1)
    targets = torch.cat([y.unsqueeze(0), negs], dim=0)
    logits = torch.cosine_similarity(x.float(), targets.float(), dim=-1).type_as(x) / logit_temp
    if neg_is_pos.any():
        logits[1:] = logits[1:].masked_fill(neg_is_pos, float("-inf"))
    return logits


def infonce_loss(logits):
    """Wav2vecCriterion (criterions/wav2vec_criterion.py:44-64) with infonce: logits [N+1, B, T] -> get_logits' [T*B, N+1]
    (wav2vec2.py:738-741), cross_entropy against class 0, sum"""
    l2 = logits.transpose(0, 2).reshape(-1, logits.size(0)).float()
    return F.cross_entropy(l2, l2.new_zeros(l2.size(0), dtype=torch.long), reduction="sum"), l2


def criterion(net_output, pred_masked_weight=1.0, pred_nomask_weight=0.0, loss_weights=None):
    """WavLMCriterion.get_loss (src/fairseq/criterions/wavlm_criterion.py:52-138), sum reduction."""
    loss = 0.0
    sample_size = 0
    log = {}
    lm = [l.float() for l in net_output["logit_m_list"]]
    lu = [l.float() for l in net_output["logit_u_list"]]
    lw = net_output.get("layer_weights")
    nw = F.softmax(lw.float(), dim=-1) if lw is not None else None
    for i, l in enumerate(lm):
        li = F.cross_entropy(l, l.new_zeros(l.size(0), dtype=torch.long), reduction="sum")
        log[f"loss_m_{i}"] = li
        if pred_masked_weight > 0:
            loss = loss + pred_masked_weight * (li if nw is None else nw[i] * li)
    if pred_masked_weight > 0:
        sample_size += lm[0].size(0)
    for i, l in enumerate(lu):
        li = F.cross_entropy(l, l.new_zeros(l.size(0), dtype=torch.long), reduction="sum")
        log[f"loss_u_{i}"] = li
        if pred_nomask_weight > 0:
            loss = loss + pred_nomask_weight * (li if nw is None else nw[i] * li)
    if pred_nomask_weight > 0:
        sample_size += lu[0].size(0)
    if loss_weights is not None:
        p = loss_weights[0] * net_output["features_pen"].float() * sample_size
        loss = loss + p
        log["loss_features_pen"] = p
        if "loss_spk_m" in net_output:  # get_extra_losses order: features_pen, loss_spk_m, loss_spk_u (None)
            w = loss_weights[1] if len(loss_weights) > 1 else loss_weights[0]
            q = w * net_output["loss_spk_m"].float() * sample_size
            loss = loss + q
            log["loss_loss_spk_m"] = q
        if "prob_perplexity" in net_output and len(loss_weights) > 3 and loss_weights[3] != 0:
            # get_extra_losses order: features_pen, loss_spk_m, loss_spk_u (None), prob_perplexity (unispeech_sat.py:803-825)
            d = loss_weights[3] * ((net_output["num_vars"] - net_output["prob_perplexity"]) / net_output["num_vars"]).float() * sample_size
            loss = loss + d
            log["loss_prob_perplexity"] = d
    for name, ls in (("m", lm), ("u", lu)):
        for i, l in enumerate(ls):
            if l.numel() == 0:
                log[f"correct_{name}_{i}"], log[f"count_{name}_{i}"] = 0, 0
                continue
            mx = l.argmax(-1) == 0
            mn = l.argmin(-1) == 0
            log[f"correct_{name}_{i}"] = int(mx.long().sum().item() - (mx & mn).long().sum().item())
            log[f"count_{name}_{i}"] = int(mx.numel())
    return loss, sample_size, log


# ---------------------------------------------------------------------------------------------------- optimizer
def adam_reference_step(p, g, m, v, step, lr, beta1, beta2, eps, weight_decay):
    """fairseq Adam.step for one tensor (src/fairseq/optim/adam.py:203-224), fp32, out-of-place"""
    m = m * beta1 + g * (1 - beta1)
    v = v * beta2 + g * g * (1 - beta2)
    denom = v.sqrt() + eps
    step_size = lr * math.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
    if weight_decay != 0:
        p = p + p * (-weight_decay * lr)
    p = p - step_size * (m / denom)
    return p, m, v


def grad_norm(grads):
    """utils.clip_grad_norm_'s total norm (src/fairseq/utils.py:358-375): L2 norm of the per-tensor fp32 L2 norms"""
    grads = list(grads)
    if len(grads) == 1:
        return float(torch.norm(grads[0], p=2, dtype=torch.float32))
    return float(torch.norm(torch.stack([torch.norm(g, p=2, dtype=torch.float32) for g in grads])))


def clip_coef(total_norm, max_norm):
    """clip coefficient of utils.clip_grad_norm_ (utils.py:380-384) == the factor _FP16OptimizerMixin.clip_grad_norm
    folds into _multiply_factor when there is no loss scaler, i.e. in bf16 mode (optim/fp16_optimizer.py:186-203)"""
    if max_norm <= 0:
        return 1.0
    return min(1.0, float(max_norm) / (float(total_norm) + 1e-6))


def train_steps(sd, cfg, batches, num_classes, lr, betas, eps, weight_decay, max_norm, loss_weights=(10.0,),
                model_dtype=None, return_grads=False):
    """`len(batches)` optimizer updates of the reference training loop in fp32 on the CPU, restated:
    forward + criterion + backward (models/wavlm/wavlm.py:465-576, criterions/wavlm_criterion.py:52-138), GradMultiply on
    the extractor (modules/grad_multiply.py), multiply_grads(1 / sample_size) (trainer.py:796-801, one worker),
    clip_grad_norm (utils.py:338-388), Adam (optim/adam.py:203-224).  batches: [(wav, target, padding_mask, mask)].
    `sd` holds the initial parameters (plain tensors); returns (losses, sample_sizes, grad_norms, final sd).
    model_dtype=torch.bfloat16 restates the reference's --bf16 parameter handling (optim/fp16_optimizer.py:269-289,
    106-133): fp32 master parameters are updated, the model computes with their bf16 rounding (here: rounded values in
    fp32 arithmetic, i.e. only the parameter rounding of that mode, not its activation rounding)."""
    names = [k for k, v in sd.items() if v.is_floating_point()]
    p = {k: sd[k].detach().clone() for k in sd}
    m = {k: torch.zeros_like(p[k]) for k in names}
    v = {k: torch.zeros_like(p[k]) for k in names}
    losses, sizes, norms = [], [], []
    fgm = cfg.feature_grad_mult
    for step, (wav, target, pm, mask) in enumerate(batches, 1):
        def model_copy(t):
            return t.detach().to(model_dtype).float() if model_dtype is not None else t.detach().clone()
        leaf = {k: (model_copy(t).requires_grad_(True) if k in m else t) for k, t in p.items()}
        net = pretrain_forward(leaf, cfg, wav, [target], pm, mask, num_classes)
        loss, ss, _ = criterion(net, 1.0, 0.0, list(loss_weights))
        loss.backward()
        grads = {}
        for k in names:
            g = leaf[k].grad if leaf[k].grad is not None else torch.zeros_like(leaf[k])
            if k.startswith("feature_extractor.") and fgm > 0 and fgm != 1.0:
                g = g * fgm
            grads[k] = g / float(ss)
        gn = grad_norm(grads.values())
        c = clip_coef(gn, max_norm)
        for k in names:
            p[k], m[k], v[k] = adam_reference_step(p[k], grads[k] * c, m[k], v[k], step, lr, betas[0], betas[1], eps,
                                                   weight_decay)
        losses.append(float(loss.detach()))
        sizes.append(int(ss))
        norms.append(gn)
        if return_grads:
            return losses, sizes, norms, p, {k: grads[k] * float(ss) for k in names}  # first update only, unnormalised
    return losses, sizes, norms, p


# ------------------------------------------------------------------------------ wav2vec 2.0
def gumbel_vq(sd, prefix, x, groups, num_vars, tau, training):
    """GumbelVectorQuantizer.forward (src/fairseq/modules/gumbel_vector_quantizer.py:157-213), combine_groups=False,
    time_first=True.  x [B, T, C].  Training draws the Gumbel noise with F.gumbel_softmax from the global torch generator,
    exactly as the reference does."""
    bsz, tsz, fsz = x.shape
    h = x.reshape(-1, fsz)
    i = 0
    while prefix + "weight_proj.%d.0.weight" % i in sd:  # weight_proj_depth > 1: Linear + GELU blocks (lines 54-66)
        h = F.gelu(F.linear(h, sd[prefix + "weight_proj.%d.0.weight" % i], sd[prefix + "weight_proj.%d.0.bias" % i]))
        i += 1
    wk = prefix + ("weight_proj.%d." % i if i else "weight_proj.")
    lg = F.linear(h, sd[wk + "weight"], sd[wk + "bias"])
    lg = lg.view(bsz * tsz * groups, -1)
    _, k = lg.max(-1)
    hard_x = lg.new_zeros(*lg.shape).scatter_(-1, k.view(-1, 1), 1.0).view(bsz * tsz, groups, -1)
    hard_probs = torch.mean(hard_x.float(), dim=0)
    code_ppl = torch.exp(-torch.sum(hard_probs * torch.log(hard_probs + 1e-7), dim=-1)).sum()
    avg_probs = torch.softmax(lg.view(bsz * tsz, groups, -1).float(), dim=-1).mean(dim=0)
    prob_ppl = torch.exp(-torch.sum(avg_probs * torch.log(avg_probs + 1e-7), dim=-1)).sum()
    if training:
        y = F.gumbel_softmax(lg.float(), tau=tau, hard=True).type_as(lg)
    else:
        y = hard_x
    y = y.view(bsz * tsz, -1)
    out = (y.unsqueeze(-1) * sd[prefix + "vars"]).view(bsz * tsz, groups, num_vars, -1).sum(-2).view(bsz, tsz, -1)
    return {"x": out, "prob_perplexity": prob_ppl, "code_perplexity": code_ppl, "num_vars": num_vars * groups}


def wav2vec2_forward(sd, cfg, source, padding_mask, mask_indices, training=True):
    """Wav2Vec2Model.forward (src/fairseq/models/wav2vec/wav2vec2.py:556-718) with dropouts 0, the time mask given
    explicitly; quantize_input, negatives_from_everywhere, codebook_negatives and target_glu as configured (transpose off).  Returns the reference's result dict with `x` = logits [N+1, B, T_m]."""
    feats = conv_feature_extractor(sd, cfg, source)
    features_pen = feats.float().pow(2).mean()
    features = feats.transpose(1, 2)
    C = features.shape[-1]
    features = F.layer_norm(features, (C,), sd["layer_norm.weight"], sd["layer_norm.bias"], 1e-5)
    unmasked = features
    T = features.shape[1]
    if padding_mask is not None:
        padding_mask = forward_padding_mask(T, padding_mask)
    if "post_extract_proj.weight" in sd:
        features = F.linear(features, sd["post_extract_proj.weight"], sd["post_extract_proj.bias"])
    res = {"features_pen": features_pen, "padding_mask": padding_mask}
    if getattr(cfg, "quantize_input", False):  # wav2vec2.py:346-363, 602-609
        pre = "quantizer." if (getattr(cfg, "same_quantizer", False) and cfg.quantize_targets) else "input_quantizer."
        qi = gumbel_vq(sd, pre, features, cfg.latent_groups, cfg.latent_vars, cfg.latent_temp[0], training)
        features = F.linear(qi["x"], sd["project_inp.weight"], sd["project_inp.bias"])
        res.update(prob_perplexity=qi["prob_perplexity"], code_perplexity=qi["code_perplexity"], num_vars=qi["num_vars"])
    x = torch.where(mask_indices.unsqueeze(-1), sd["mask_emb"].view(1, 1, -1), features)
    B = x.shape[0]
    y = unmasked[mask_indices].view(B, -1, C)
    x, _, _ = transformer_encoder(sd, cfg, x, padding_mask, None)
    from unispeech_amd.functional import sample_negatives_indices  # host index draws: pinned bit-exact by sampled_negatives.npz
    N = cfg.num_negatives + cfg.cross_sample_negatives
    nfe = bool(getattr(cfg, "negatives_from_everywhere", False))
    cbn = int(getattr(cfg, "codebook_negatives", 0))
    Tm = y.shape[1]

    def gather(cand, tsz):  # sample_negatives' gather (wav2vec2.py:521-531): cand [B, tsz, F] -> [N, B, Tm, F]
        idx = sample_negatives_indices(B, tsz, Tm, cfg.num_negatives, cfg.cross_sample_negatives)
        return cand.reshape(-1, cand.size(-1))[idx.view(-1)].view(B, Tm, N, -1).permute(2, 0, 1, 3)

    def project_q(t):
        return F.linear(t, sd["project_q.weight"], sd["project_q.bias"])

    if cfg.quantize_targets:
        q = gumbel_vq(sd, "quantizer.", y, cfg.latent_groups, cfg.latent_vars, cfg.latent_temp[0], training)
        y = project_q(q["x"])
        q2 = gumbel_vq(sd, "quantizer.", unmasked, cfg.latent_groups, cfg.latent_vars, cfg.latent_temp[0], training)  # results['q'] (wav2vec2.py:655)
        res.update(prob_perplexity=q["prob_perplexity"], code_perplexity=q["code_perplexity"], num_vars=q["num_vars"])
        negs = gather(project_q(q2["x"]), T) if nfe else gather(y, Tm)   # wav2vec2.py:657-668
        if cbn > 0:  # wav2vec2.py:669-677 + GumbelVectorQuantizer.sample_from_codebook (gumbel_vector_quantizer.py:90-128)
            from itertools import product
            G, V = cfg.latent_groups, cfg.latent_vars
            inds = torch.tensor(list(product(*[range(V)] * G)), dtype=torch.long).view(V ** G, -1)
            for g in range(1, G):
                inds[:, g] += V * g
            assert cbn < inds.size(0)
            sample_idx = torch.randint(low=0, high=inds.size(0), size=(B * Tm * cbn,))
            z = sd["quantizer.vars"].squeeze(0).index_select(0, inds[sample_idx].flatten()).view(B * Tm, cbn, -1)
            negs = torch.cat([negs, project_q(z.view(cbn, B, Tm, -1))], dim=0)
    else:
        y = project_q(y)
        negs = project_q(gather(unmasked, T)) if nfe else gather(y, Tm)      # wav2vec2.py:679-692
    if getattr(cfg, "target_glu", False):  # wav2vec2.py:371-375, 697-699
        def tglu(t):
            return F.glu(F.linear(t, sd["target_glu.0.weight"], sd["target_glu.0.bias"]), dim=-1)
        y, negs = tglu(y), tglu(negs)
    xm = F.linear(x[mask_indices].view(B, -1, x.size(-1)), sd["final_proj.weight"], sd["final_proj.bias"])
    # compute_preds (wav2vec2.py:533-553)
    neg_is_pos = (y == negs).all(-1)
    targets = torch.cat([y.unsqueeze(0), negs], dim=0)
    logits = torch.cosine_similarity(xm.float(), targets.float(), dim=-1).type_as(xm) / cfg.logit_temp
    if neg_is_pos.any():
        logits[1:] = logits[1:].masked_fill(neg_is_pos, float("-inf"))
    res["x"] = logits
    return res


def wav2vec_criterion(res, loss_weights=None):
    """Wav2vecCriterion.get_loss with infonce (criterions/wav2vec_criterion.py:44-123)"""
    loss, l2 = infonce_loss(res["x"])
    sample_size = l2.size(0)
    log = {"loss_0": loss.detach().clone()}
    if loss_weights is not None:
        extra = []
        if "prob_perplexity" in res:
            extra.append((res["num_vars"] - res["prob_perplexity"]) / res["num_vars"])
        extra.append(res["features_pen"])
        w = list(loss_weights)
        if len(w) == 1 and len(extra) != 1:
            w = [w[0]] * len(extra)
        for i, (p, c) in enumerate(zip(extra, w)):
            if c != 0:
                p = c * p.float() * sample_size
                loss = loss + p
                log["loss_%d" % (i + 1)] = p.detach()
    mx, mn = l2.argmax(-1) == 0, l2.argmin(-1) == 0
    log["correct"] = int(mx.long().sum().item() - (mx & mn).long().sum().item())
    log["count"] = int(mx.numel())
    return loss, sample_size, log


# ------------------------------------------------------------------------- utterance mixing
def mix_collated_audios(source, ops, op_begin, noise=None, normalize=False):
    """The arithmetic of UtteranceMixingDataset.mixing_collated_audios (src/fairseq/data/audio/utterance_mixing_dataset.py:
    373-438) for a given list of mixing ops (the random draws, made by unispeech_amd.data.UtteranceMixingCollater.
    draw_mixing_plan from the reference's numpy stream): in place, in row order, numpy float32 powers, the reference's
    expression for the scale, `.clone()` of the added span, per-row F.layer_norm when `normalize`."""
    import numpy as np
    source = source.clone()
    B = source.shape[0]
    for i in range(B):
        for k in range(int(op_begin[i]), int(op_begin[i + 1])):
            row, kind, src, c_start, s_start, c_len, src_len, gbits = [int(v) for v in ops[k]]
            gain = np.array([gbits], dtype=np.int32).view(np.float32)[0]
            assert row == i
            ref_pow = np.mean(source[i].numpy() ** 2)
            if kind == 1:
                nz = noise[src:src + src_len]
                noise_pow = np.mean(nz ** 2)
                scale = 0 if noise_pow == 0 else (ref_pow / (noise_pow * gain)) ** 0.5
                seg = torch.from_numpy(np.asarray(scale * nz, dtype=np.float32))
                source[i, s_start:s_start + c_len] += seg[c_start:c_start + c_len]
            else:
                noise_pow = np.mean(source[src].numpy() ** 2)
                scale = 0 if noise_pow == 0 else (ref_pow / (noise_pow * gain)) ** 0.5
                source[i, s_start:s_start + c_len] += source[src, c_start:c_start + c_len].clone() * scale
        if normalize and op_begin[i + 1] > op_begin[i]:
            with torch.no_grad():
                source[i] = F.layer_norm(source[i], source[i].shape)
    return source
