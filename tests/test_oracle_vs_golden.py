"""Pins the CPU oracle (oracle/wavlm_oracle.py) and the host logic (masking, bucketing, seeded init) against the
fixtures that oracle/gen_golden.py produced by running the reference's own Python.  CPU only."""
import numpy as np
import pytest
import torch

from conftest import Cfg, TINY, golden_state_dict, load_golden
from oracle import wavlm_oracle as O
from unispeech_amd import masking
from unispeech_amd.wavlm import WavLM, WavLMConfig
from unispeech_amd.pretrain import WavLMPretrainConfig, WavLMPretrainModel

RTOL = 1e-4


def rel_err(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


# ------------------------------------------------------------------------------------------------ masks
def test_masks_bit_exact_and_rng_stream():
    z = load_golden("masks.npz")
    for i in range(int(z["ncases"])):
        B, T, L, min_masks, no, space, seed = [int(v) for v in z[f"case{i}/args"]]
        p, other = [float(v) for v in z[f"case{i}/fargs"]]
        kind = str(z[f"case{i}/kind"])
        pad = z[f"case{i}/pad"]
        pm = None
        if pad.size:
            pm = torch.zeros((B, T), dtype=torch.bool)
            for b, n in enumerate(pad):
                pm[b, int(n):] = True
        np.random.seed(seed)
        m = masking.compute_mask_indices((B, T), pm, p, L, kind, other, min_masks=min_masks, no_overlap=bool(no),
                                         min_space=space)
        nxt = np.random.random()
        assert np.array_equal(m, z[f"case{i}/mask"]), f"mask case {i}"
        assert nxt == float(z[f"case{i}/next"]), f"numpy RNG stream diverged after case {i}"
        counts = m.sum(1)
        assert (counts == counts[0]).all()  # every row masked equally (batch minimum)


def test_relative_position_buckets():
    z = load_golden("buckets.npz")
    for key in z.files:
        T, nb, md = [int(s[1:] if s[0] == "T" else s[2:]) for s in key.split("_")]
        mine = masking.relative_position_buckets(T, nb, md).numpy()
        assert np.array_equal(mine, z[key]), key
        # and the oracle's full-grid restatement agrees with the Toeplitz table
        if T <= 749:
            ctx = torch.arange(T)[:, None]
            mem = torch.arange(T)[None, :]
            full = O.relative_positions_bucket(mem - ctx, nb, md)
            idx = (mem - ctx) + T - 1
            assert torch.equal(full, torch.from_numpy(z[key]).long()[idx])


# ------------------------------------------------------------------------------------------ seeded init
def test_seeded_init_matches_reference_constructor():
    z = load_golden("tiny_wavlm.npz")
    ref_sd = golden_state_dict(z)
    torch.manual_seed(0)
    model = WavLM(WavLMConfig(dict(TINY)))
    sd = model.state_dict()
    assert set(sd.keys()) == set(ref_sd.keys())
    for k in ref_sd:
        assert torch.equal(sd[k], ref_sd[k]), k


@pytest.mark.parametrize("golden,overrides", [("tiny_pretrain.npz", {}), ("tiny_targetglu.npz", {"target_glu": True}),
                                              ("tiny_act_glu.npz", {"activation_fn": "glu"})])
def test_seeded_init_pretrain_model(golden, overrides):
    """(tiny_targetglu: the target_glu Linear is created between the encoder and final_proj, as in wavlm.py:322-333;
    tiny_act_glu: fc1 is a GLU_Linear whose parameters are fc1.linear.*, unispeech_sat.py:977-1007, 1065-1066)"""
    z = load_golden(golden)
    ref_sd = golden_state_dict(z)
    d = dict(TINY)
    d.update(overrides)
    cfg = WavLMPretrainConfig(**{k: v for k, v in d.items() if k in WavLMPretrainConfig.__dataclass_fields__})
    torch.manual_seed(0)
    model = WavLMPretrainModel(cfg, None, [range(23)])
    sd = model.state_dict()
    assert set(sd.keys()) == set(ref_sd.keys())
    for k in ref_sd:
        assert torch.equal(sd[k], ref_sd[k]), k
    # released-checkpoint style load
    model.load_state_dict(ref_sd)


# --------------------------------------------------------------------------------------------- oracle
def test_oracle_extract_features(tiny_cfg):
    z = load_golden("tiny_wavlm.npz")
    sd = golden_state_dict(z)
    wav = torch.from_numpy(z["in/source"])
    with torch.no_grad():
        conv = O.conv_feature_extractor(sd, tiny_cfg, wav)
        assert rel_err(conv, z["out/conv_features"]) < RTOL
        r = O.extract_features(sd, tiny_cfg, wav)
        assert rel_err(r["x"], z["out/x"]) < RTOL
        assert rel_err(r["features"], z["out/features_ret_conv"]) < RTOL
        r1 = O.extract_features(sd, tiny_cfg, wav, output_layer=1)
        assert rel_err(r1["x"], z["out/x_layer1"]) < RTOL
        assert len(r1["layer_results"]) == int(z["out/nlayer_results_layer1"])
        assert rel_err(r1["layer_results"][0][0], z["out/layer_results0"]) < RTOL
        assert rel_err(r1["layer_results"][1][0], z["out/layer_results1"]) < RTOL
        pm = torch.from_numpy(z["in/padding_mask"])
        wav_p = wav.clone()
        wav_p[1, 12000:] = 0
        rp = O.extract_features(sd, tiny_cfg, wav_p, padding_mask=pm)
        assert torch.equal(rp["padding_mask"], torch.from_numpy(z["out/padding_mask_frames"]))
        assert rel_err(rp["x"], z["out/x_padded"]) < RTOL
        m = torch.from_numpy(z["out/mask_seed123"])
        rm = O.extract_features(sd, tiny_cfg, wav, mask_indices=m)
        assert rel_err(rm["x"], z["out/x_masked"]) < RTOL


ACT_GOLDENS = ["tiny_act_relu.npz", "tiny_act_glu.npz", "tiny_act_gelu_accurate.npz", "tiny_act_tanh.npz"]


@pytest.mark.parametrize("golden", ["tiny_pretrain.npz", "tiny_chanmask.npz", "tiny_convbias.npz", "tiny_targetglu.npz"] + ACT_GOLDENS)
def test_oracle_pretrain_loss_and_grads(tiny_cfg, golden):
    """tiny_chanmask: the same run with mask_channel_prob 0.25 (apply_mask's channel half, wavlm.py:405-422);
    tiny_convbias: conv_bias=True (Conv1d biases in the extractor); tiny_targetglu: target_glu=True (Linear + GLU on the
    label embeddings, wavlm.py:322-327, 529-531)"""
    tiny_cfg.target_glu = golden == "tiny_targetglu.npz"
    if golden in ACT_GOLDENS:  # tiny_act_*: activation_fn of the feed-forward block (utils.get_activation_fn; glu = GLU_Linear)
        tiny_cfg.activation_fn = golden[len("tiny_act_"):-len(".npz")]
    z = load_golden(golden)
    sd = golden_state_dict(z, as_param=True)
    wav = torch.from_numpy(z["in/source"])
    target = torch.from_numpy(z["in/target"])
    pm = torch.from_numpy(z["in/padding_mask"])
    m = torch.from_numpy(z["out/mask_seed123"])
    cm = torch.from_numpy(z["out/chan_mask_seed123"]) if "out/chan_mask_seed123" in z.files else None
    net = O.pretrain_forward(sd, tiny_cfg, wav, [target], pm, m, [23], chan_mask=cm)
    assert rel_err(net["x"].detach(), z["out/x"]) < RTOL
    lm, lu = net["logit_m_list"][0].detach(), net["logit_u_list"][0].detach()
    gm, gu = torch.from_numpy(z["out/logit_m"]), torch.from_numpy(z["out/logit_u"])
    assert lm.shape == gm.shape and lu.shape == gu.shape
    fin = torch.isfinite(gm)
    assert torch.equal(torch.isfinite(lm), fin)
    assert rel_err(lm[fin], gm[fin]) < RTOL
    assert abs(net["features_pen"].item() - float(z["out/features_pen"])) < RTOL * abs(float(z["out/features_pen"]))
    loss, ss, log = O.criterion(net, 1.0, 0.0, [10.0])
    assert ss == int(z["out/sample_size"])
    assert abs(loss.item() - float(z["out/loss"])) < RTOL * abs(float(z["out/loss"]))
    assert log["correct_m_0"] == int(z["log/correct_m_0"]) and log["count_m_0"] == int(z["log/count_m_0"])
    assert log["correct_u_0"] == int(z["log/correct_u_0"]) and log["count_u_0"] == int(z["log/count_u_0"])
    loss.backward()
    # feature_grad_mult: the reference scales the extractor gradient by 0.1 (GradMultiply); the oracle forward has
    # no such node, so compare extractor grads after applying the factor
    gmax = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith("grad/"))
    for k, p in sd.items():
        if not p.is_floating_point():
            continue
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        if k.startswith("feature_extractor."):
            g = g * tiny_cfg.feature_grad_mult
        ref = torch.from_numpy(z["grad/" + k])
        scale = ref.abs().max().item()
        # k_proj.bias (softmax is invariant to a key bias) and, with conv_bias, block 0's Conv1d bias (cancelled by its
        # GroupNorm) have analytically zero gradients -- rounding noise on both sides: floor relative to the largest gradient
        if k == "feature_extractor.conv_layers.0.0.bias":  # cancelled by GroupNorm: sums of O(0.1) terms that vanish
            assert g.abs().max().item() < 1e-6 * gmax and scale < 1e-6 * gmax, k
            continue
        assert (g - ref).abs().max().item() <= 5e-4 * max(scale, 1e-6 * gmax) + 1e-8, k


@pytest.mark.parametrize("golden", ["tiny_large.npz", "tiny_large_convbias.npz"])
def test_oracle_large_structure(golden):
    """extractor_mode 'layer_norm' + layer_norm_first (WavLM-Large structure, BASELINE.json configs[3]) against the
    reference-generated golden: forward and every parameter gradient of a scalar probe loss.  _convbias: conv_bias=True."""
    from conftest import Cfg, TINY
    z = load_golden(golden)
    d = dict(TINY)
    d.update(extractor_mode="layer_norm", layer_norm_first=True, normalize=True)
    cfg = Cfg(**d)
    sd = golden_state_dict(z, as_param=True)
    wav = torch.from_numpy(z["in/source"])
    with torch.no_grad():
        conv = O.conv_feature_extractor(sd, cfg, wav)
        assert rel_err(conv, z["out/conv_features"]) < RTOL
    r = O.extract_features(sd, cfg, wav)
    assert rel_err(r["x"].detach(), z["out/x"]) < RTOL
    (r["x"] * torch.from_numpy(z["in/probe"])).sum().backward()
    # the standalone reference has feature_grad_mult applied inside extract_features (WavLM.py:339-345)
    gmax = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith("grad/"))
    for k, p in sd.items():
        if not p.is_floating_point() or ("grad/" + k) not in z.files:
            continue
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        ref = torch.from_numpy(z["grad/" + k])
        if k.startswith("feature_extractor."):
            g = g * cfg.feature_grad_mult
        assert (g - ref).abs().max().item() <= 5e-4 * max(ref.abs().max().item(), 1e-6 * gmax) + 1e-8, k


def _sat_cfg(**overrides):
    from conftest import Cfg, TINY
    d = dict(TINY)
    d.update(relative_position_embedding=False, gru_rel_pos=False, utterance_contrastive_loss=True,
             utterance_contrastive_layer=1, num_instances=2, cross_sample_instances=5)
    d.update(overrides)
    return Cfg(**d)


SAT_CASES = [("tiny_sat.npz", {}),
             ("tiny_sat_relpos.npz", {"relative_position_embedding": True, "gru_rel_pos": True}),
             ("tiny_sat_large.npz", {"relative_position_embedding": True, "gru_rel_pos": True,
                                     "extractor_mode": "layer_norm", "layer_norm_first": True}),
             ("tiny_sat_quant.npz", {"quantize_targets": True, "latent_vars": 20, "latent_groups": 2, "latent_dim": 0,
                                     "latent_temp": (2.0, 0.5, 0.999995)})]
SAT_LOSS_WEIGHTS = {"tiny_sat_quant.npz": [10.0, 5.0, 0.0, 0.1]}  # 4th: codebook diversity (prob_perplexity)


@pytest.mark.parametrize("golden,overrides", SAT_CASES)
def test_seeded_init_sat_variants(golden, overrides):
    """parameter set / creation order of the UniSpeech-SAT model incl. encoder.layer_norm_for_extract of the pre-LN
    (Large) structure (unispeech_sat.py:1195-1200): same seed -> bit-identical initial weights, strict state-dict load"""
    z = load_golden(golden)
    ref_sd = golden_state_dict(z)
    d = dict(vars(_sat_cfg(**overrides)))
    cfg = WavLMPretrainConfig(**{k: v for k, v in d.items() if k in WavLMPretrainConfig.__dataclass_fields__})
    torch.manual_seed(0)
    model = WavLMPretrainModel(cfg, None, [range(23)])
    sd = model.state_dict()
    assert set(sd.keys()) == set(ref_sd.keys()), set(sd.keys()) ^ set(ref_sd.keys())
    for k in ref_sd:
        assert torch.equal(sd[k], ref_sd[k]), k


@pytest.mark.parametrize("golden,overrides", SAT_CASES)
def test_oracle_unispeech_sat_head(golden, overrides):
    """UniSpeech-SAT utterance-contrastive head against the reference-generated golden: total loss,
    speaker loss / statistics and every parameter gradient; instance indices come from the same torch.randint stream.
    Variants: gated relative position bias on; the Large structure (layer_norm extractor, pre-LN encoder, speaker tap
    through layer_norm_for_extract; BASELINE.json configs[4])."""
    z = load_golden(golden)
    cfg = _sat_cfg(**overrides)
    sd = golden_state_dict(z, as_param=True)
    wav = torch.from_numpy(z["in/source"])
    target = torch.from_numpy(z["in/target"])
    m = torch.from_numpy(z["out/mask_seed321"])
    torch.manual_seed(77)
    net = O.pretrain_forward(sd, cfg, wav, [target], torch.zeros(3, 16000, dtype=torch.bool), m, [23])
    assert rel_err(net["x"].detach(), z["out/x"]) < RTOL
    assert abs(net["loss_spk_m"].item() - float(z["out/loss_spk_m"])) < RTOL * abs(float(z["out/loss_spk_m"]))
    assert abs(float(net["mean_targets"]) - float(z["out/mean_targets"])) < 1e-6
    assert abs(float(net["contrastive_acc"]) - float(z["out/contrastive_acc"])) < 1e-6
    if "out/prob_perplexity" in z.files:
        assert abs(net["prob_perplexity"].item() - float(z["out/prob_perplexity"])) < RTOL * float(z["out/prob_perplexity"])
    loss, ss, _ = O.criterion(net, 1.0, 0.0, SAT_LOSS_WEIGHTS.get(golden, [10.0, 5.0, 0.0]))
    assert ss == int(z["out/sample_size"])
    assert abs(loss.item() - float(z["out/loss"])) < RTOL * abs(float(z["out/loss"]))
    loss.backward()
    gmax = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith("grad/"))
    for k, p in sd.items():
        if not p.is_floating_point() or ("grad/" + k) not in z.files:
            continue
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        ref = torch.from_numpy(z["grad/" + k])
        if k.startswith("feature_extractor."):
            g = g * cfg.feature_grad_mult
        assert (g - ref).abs().max().item() <= 5e-4 * max(ref.abs().max().item(), 1e-6 * gmax) + 1e-8, k


ILS_CASES = [("tiny_ils.npz", {}, (23,)),
             ("tiny_ils_preln.npz", {"layer_norm_first": True, "extractor_mode": "layer_norm"}, (23,)),
             ("tiny_ils_sep_embeds.npz", {"separate_label_embeds": True}, (23, 17)),
             ("tiny_ils_sep_targets.npz", {"separate_layer_targets": True, "separate_label_embeds": True, "weighted_sum": True}, (23, 17))]

@pytest.mark.parametrize("golden,overrides,vocabs", ILS_CASES)
def test_seeded_init_ils_variants(golden, overrides, vocabs):
    """parameter creation order of the ILS heads (ils_hubert.py:70-107): same seed -> bit-identical initial weights"""
    z = load_golden(golden)
    ref_sd = golden_state_dict(z)
    d = dict(TINY)
    d.update(predict_layers="[1,2]", gru_rel_pos=False)
    d.update(overrides)
    cfg = WavLMPretrainConfig(**{k: v for k, v in d.items() if k in WavLMPretrainConfig.__dataclass_fields__})
    torch.manual_seed(0)
    model = WavLMPretrainModel(cfg, None, [range(V) for V in vocabs])
    sd = model.state_dict()
    assert set(sd.keys()) == set(ref_sd.keys())
    for k in ref_sd:
        assert torch.equal(sd[k], ref_sd[k]), k


@pytest.mark.parametrize("golden,overrides,vocabs", ILS_CASES)
def test_oracle_ils_hubert(golden, overrides, vocabs):
    """ILS-SSL: the masked-prediction head on the outputs of layers [1, 2] against the golden
    generated from the reference's ILSHubertModel + HubertCriterion: logits of both layers, loss, gradients.  Variants:
    per-layer final_proj / label embeddings, one label set per layer with softmax-weighted layer losses."""
    from conftest import Cfg, TINY
    z = load_golden(golden)
    d = dict(TINY)
    d.update(predict_layers="[1,2]", gru_rel_pos=False)  # HuBERT config: relative position bias without the gate
    d.update(overrides)
    cfg = Cfg(**d)
    sd = golden_state_dict(z, as_param=True)
    wav = torch.from_numpy(z["in/source"])
    targets = [torch.from_numpy(z["in/target%d" % i]) for i in range(len(vocabs))] if "in/target0" in z.files \
        else [torch.from_numpy(z["in/target"])]
    m = torch.from_numpy(z["out/mask_seed222"])
    net = O.pretrain_forward(sd, cfg, wav, targets, torch.zeros(2, 16000, dtype=torch.bool), m, list(vocabs))
    assert len(net["logit_m_list"]) == int(z["out/n_logit_m"])
    for i, l in enumerate(net["logit_m_list"]):
        ref = torch.from_numpy(z["out/logit_m%d" % i])
        fin = torch.isfinite(ref)
        assert torch.equal(torch.isfinite(l.detach()), fin)
        assert rel_err(l.detach()[fin], ref[fin]) < RTOL
    loss, ss, _ = O.criterion(net, 1.0, 0.0, [10.0])
    assert ss == int(z["out/sample_size"])
    assert abs(loss.item() - float(z["out/loss"])) < RTOL * abs(float(z["out/loss"]))
    loss.backward()
    gmax = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith("grad/"))
    for k, p in sd.items():
        if not p.is_floating_point() or ("grad/" + k) not in z.files:
            continue
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        ref = torch.from_numpy(z["grad/" + k])
        if k.startswith("feature_extractor."):
            g = g * cfg.feature_grad_mult
        assert (g - ref).abs().max().item() <= 5e-4 * max(ref.abs().max().item(), 1e-6 * gmax) + 1e-8, k


def test_sampled_negatives_oracle_and_index_stream():
    """wav2vec 2.0 sampled negatives: the host index draws are bit-exact with the reference's
    sample_negatives under the same torch seed; the oracle's logits (incl. -inf masking of negatives equal to the
    positive), loss and gradients match the golden produced by the reference functions."""
    from unispeech_amd.functional import sample_negatives_indices
    z = load_golden("sampled_negatives.npz")
    B, T, C = z["in/y"].shape
    torch.manual_seed(31)
    idx = sample_negatives_indices(B, T, T, 7, 3)
    assert torch.equal(idx, torch.from_numpy(z["out/neg_idxs"]))
    x = torch.from_numpy(z["in/x"]).requires_grad_(True)
    y = torch.from_numpy(z["in/y"]).requires_grad_(True)
    logits = O.sampled_negatives_logits(x, y, idx, 10, 0.1)
    ref = torch.from_numpy(z["out/logits"])
    fin = torch.isfinite(ref)
    assert int((~fin).sum()) == int(z["out/n_masked"]) > 0
    assert torch.equal(torch.isfinite(logits.detach()), fin)
    assert rel_err(logits.detach()[fin], ref[fin]) < RTOL
    loss, _ = O.infonce_loss(logits)
    assert abs(loss.item() - float(z["out/loss"])) < RTOL * abs(float(z["out/loss"]))
    loss.backward()
    assert rel_err(x.grad, z["grad/x"]) < 5e-4 and rel_err(y.grad, z["grad/y"]) < 5e-4


def test_boundary_mask_and_target_trim_host_logic_and_oracle(tiny_cfg):
    """boundary_mask=True with segment boundaries on row 0 only, labels shorter than the frame sequence
    (tests/golden/tiny_boundary.npz, generated from the reference's WavLMModel): the host mask logic of the product
    (pretrain._mask_numpy: binomial coin per segment / per-row compute_mask_indices; forward_targets trim 49 -> 45) is
    bit-exact with the reference, and the oracle reproduces loss, logits and every gradient with that mask."""
    z = load_golden("tiny_boundary.npz")
    d = dict(TINY)
    d.update(boundary_mask=True)
    cfg = WavLMPretrainConfig(**{k: v for k, v in d.items() if k in WavLMPretrainConfig.__dataclass_fields__})
    model = WavLMPretrainModel(cfg, None, [range(23)])
    target = torch.from_numpy(z["in/target"])
    T, tinds = model.forward_targets(49, [target])
    assert T == 45 and torch.equal(tinds, torch.arange(45))
    boundary = [list(z["in/boundary0"]), []]
    np.random.seed(909)
    m = model._mask_numpy(2, T, torch.zeros(2, T, dtype=torch.bool), boundary)
    assert np.array_equal(m, z["out/mask_seed909"])
    sd = golden_state_dict(z, as_param=True)
    net = O.pretrain_forward(sd, tiny_cfg, torch.from_numpy(z["in/source"]), [target],
                             torch.zeros(2, 16000, dtype=torch.bool), torch.from_numpy(m), [23])
    assert net["x"].shape[1] == 45
    assert rel_err(net["x"].detach(), z["out/x"]) < RTOL
    ref = torch.from_numpy(z["out/logit_m"])
    fin = torch.isfinite(ref)
    assert rel_err(net["logit_m_list"][0].detach()[fin], ref[fin]) < RTOL
    loss, ss, _ = O.criterion(net, 1.0, 0.0, [10.0])
    assert ss == int(z["out/sample_size"])
    assert abs(loss.item() - float(z["out/loss"])) < RTOL * abs(float(z["out/loss"]))
    loss.backward()
    gmax = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith("grad/"))
    for k, p in sd.items():
        if not p.is_floating_point():
            continue
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        if k.startswith("feature_extractor."):
            g = g * tiny_cfg.feature_grad_mult
        ref = torch.from_numpy(z["grad/" + k])
        assert (g - ref).abs().max().item() <= 5e-4 * max(ref.abs().max().item(), 1e-6 * gmax) + 1e-8, k


def test_oracle_adam_clip_vs_reference_golden():
    """oracle.adam_reference_step + grad_norm + clip_coef against tests/golden/adam_clip.npz: 4 updates of the
    reference's Adam after multiply_grads + clip_grad_norm_ (update 2 is clipped)."""
    z = load_golden("adam_clip.npz")
    lr, b1, b2, eps, wd, max_norm = [float(v) for v in z["in/hyper"]]
    p = torch.from_numpy(z["in/p0"]).clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 5):
        g = torch.from_numpy(z["in/grad%d" % step]) * float(z["in/mult%d" % step])
        gn = O.grad_norm([g])
        assert abs(gn - float(z["out/gnorm%d" % step])) <= 1e-6 * gn
        p, m, v = O.adam_reference_step(p, g * O.clip_coef(gn, max_norm), m, v, step, lr, b1, b2, eps, wd)
        assert rel_err(p, z["out/p%d" % step]) < 1e-6
        assert rel_err(m, z["out/m%d" % step]) < 1e-5 and rel_err(v, z["out/v%d" % step]) < 1e-5


W2V2 = dict(TINY)
W2V2.update(final_dim=32, quantize_targets=True, latent_vars=20, latent_groups=2, latent_dim=0,
            latent_temp=(2.0, 0.5, 0.999995), num_negatives=7, cross_sample_negatives=3, logit_temp=0.1,
            relative_position_embedding=False, gru_rel_pos=False)


# golden -> config overrides: negatives_from_everywhere with the quantiser (+ codebook negatives) and without it
W2V2_GOLDENS = [("tiny_w2v2.npz", {}),
                ("tiny_w2v2_everywhere_cb.npz", {"negatives_from_everywhere": True, "codebook_negatives": 2}),
                ("tiny_w2v2_everywhere.npz", {"negatives_from_everywhere": True, "quantize_targets": False}),
                ("tiny_w2v2_qinput_glu.npz", {"quantize_input": True, "target_glu": True}),
                ("tiny_w2v2_qdepth.npz", {"quantizer_depth": 2, "quantizer_factor": 2})]


def _w2v2_cfg(overrides=None):
    from unispeech_amd.wav2vec2 import Wav2Vec2Config
    d = dict(W2V2)
    d.update(overrides or {})
    return Wav2Vec2Config(**{k: v for k, v in d.items() if k in Wav2Vec2Config.__dataclass_fields__})


@pytest.mark.parametrize("golden,overrides", W2V2_GOLDENS)
def test_seeded_init_wav2vec2_model(golden, overrides):
    """parameter set and creation order of Wav2Vec2Model incl. the Gumbel quantiser (wav2vec2.py:276-395,
    gumbel_vector_quantizer.py:41-74): same seed -> bit-identical initial weights, strict state-dict load"""
    from unispeech_amd.wav2vec2 import Wav2Vec2Model
    z = load_golden(golden)
    ref_sd = golden_state_dict(z)
    torch.manual_seed(0)
    model = Wav2Vec2Model(_w2v2_cfg(overrides))
    sd = model.state_dict()
    assert set(sd.keys()) == set(ref_sd.keys()), set(sd.keys()) ^ set(ref_sd.keys())
    for k in ref_sd:
        assert torch.equal(sd[k], ref_sd[k]), k
    model.load_state_dict(ref_sd)


@pytest.mark.parametrize("golden,overrides", W2V2_GOLDENS)
def test_oracle_wav2vec2_model_and_criterion(golden, overrides):
    """wav2vec 2.0 end to end: extractor, Gumbel quantiser in TRAIN mode (noise from the torch CPU
    generator, as the reference draws it), sampled negatives, InfoNCE, diversity + features penalties, against the golden
    generated from the reference's Wav2Vec2Model + Wav2vecCriterion: logits, perplexities, loss, every gradient."""
    from conftest import Cfg
    z = load_golden(golden)
    d = dict(W2V2)
    d.update(overrides)   # (tiny_w2v2_everywhere*: negatives_from_everywhere / codebook_negatives, wav2vec2.py:653-692)
    cfg = Cfg(**d)
    sd = golden_state_dict(z, as_param=True)
    wav = torch.from_numpy(z["in/source"])
    m = torch.from_numpy(z["out/mask_seed77"])
    torch.manual_seed(31)
    res = O.wav2vec2_forward(sd, cfg, wav, torch.zeros(3, 16000, dtype=torch.bool), m, training=True)
    if cfg.quantize_targets:
        assert abs(res["prob_perplexity"].item() - float(z["out/prob_perplexity"])) < RTOL * float(z["out/prob_perplexity"])
        assert abs(res["code_perplexity"].item() - float(z["out/code_perplexity"])) < RTOL * float(z["out/code_perplexity"])
    lw = [float(v) for v in z["in/loss_weights"]] if "in/loss_weights" in z.files else [0.1, 10.0]
    loss, ss, log = O.wav2vec_criterion(res, lw)
    l2 = res["x"].detach().transpose(0, 2).reshape(-1, res["x"].size(0))
    ref = torch.from_numpy(z["out/logits"])
    fin = torch.isfinite(ref)
    assert torch.equal(torch.isfinite(l2), fin) and rel_err(l2[fin], ref[fin]) < RTOL
    assert ss == int(z["out/sample_size"]) and log["correct"] == int(z["log/correct"])
    assert abs(loss.item() - float(z["out/loss"])) < RTOL * abs(float(z["out/loss"]))
    loss.backward()
    gmax = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith("grad/"))
    for k, p in sd.items():
        if not p.is_floating_point() or ("grad/" + k) not in z.files:
            continue
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        ref = torch.from_numpy(z["grad/" + k])
        if k.startswith("feature_extractor."):
            g = g * cfg.feature_grad_mult
        assert (g - ref).abs().max().item() <= 5e-4 * max(ref.abs().max().item(), 1e-6 * gmax) + 1e-8, k
