"""Generate tests/golden/*.npz by running the REFERENCE's own Python (build container only; needs /root/reference).

    python oracle/gen_golden.py

Fixtures (all seeds fixed, fp32, CPU):
  masks.npz      reference compute_mask_indices outputs + the next numpy draw (pins RNG consumption)
  buckets.npz    reference _relative_positions_bucket tables
  tiny_wavlm.npz standalone WavLM (tiny, relpos + gate): seeded state dict, inputs, extract_features outputs
  tiny_pretrain.npz fairseq WavLMModel + WavLMCriterion (tiny): state dict, inputs, logits, loss, all gradients
The tiny configuration keeps every structural feature of WavLM-Base (7 conv blocks with the same kernels/strides,
GroupNorm on block 0, grouped weight-normed pos_conv, post-LN layers, bucketed gated relative position bias) at
sizes small enough to commit.
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

TINY = dict(
    extractor_mode="default", encoder_layers=2, encoder_embed_dim=64, encoder_ffn_embed_dim=128,
    encoder_attention_heads=2, activation_fn="gelu", layer_norm_first=False,
    conv_feature_layers="[(32,10,5)] + [(32,3,2)] * 4 + [(32,2,2)] * 2", conv_bias=False, feature_grad_mult=0.1,
    dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, encoder_layerdrop=0.0, dropout_input=0.0,
    dropout_features=0.0, mask_length=4, mask_prob=0.65, mask_selection="static", mask_other=0,
    no_mask_overlap=False, mask_min_space=1, mask_channel_prob=0.0, conv_pos=16, conv_pos_groups=4,
    relative_position_embedding=True, num_buckets=32, max_distance=64, gru_rel_pos=True,
)


def sd_to_np(sd):
    return {"sd/" + k: v.detach().cpu().numpy() for k, v in sd.items()}


def gen_masks():
    _, _, _, _, cmi = ref_shim.fairseq_wavlm()
    out = {}
    cases = [
        dict(shape=(4, 49), pad=None, p=0.65, l=4, kind="static", other=0.0, min_masks=2, no=False, space=1, seed=1),
        dict(shape=(3, 749), pad=None, p=0.8, l=10, kind="static", other=0.0, min_masks=2, no=False, space=1, seed=2),
        dict(shape=(3, 749), pad=[749, 600, 375], p=0.8, l=10, kind="static", other=0.0, min_masks=2, no=False,
             space=1, seed=3),
        dict(shape=(2, 200), pad=None, p=0.5, l=10, kind="uniform", other=2.0, min_masks=0, no=False, space=1, seed=4),
        dict(shape=(2, 200), pad=None, p=0.5, l=10, kind="normal", other=3.0, min_masks=0, no=False, space=1, seed=5),
        dict(shape=(2, 200), pad=None, p=0.5, l=10, kind="poisson", other=0.0, min_masks=0, no=False, space=1, seed=6),
        dict(shape=(2, 300), pad=None, p=0.5, l=10, kind="static", other=0.0, min_masks=2, no=True, space=2, seed=7),
        dict(shape=(8, 999), pad=[999, 999, 900, 800, 700, 999, 500, 999], p=0.65, l=10, kind="static", other=0.0,
             min_masks=2, no=False, space=1, seed=8),
    ]
    for i, c in enumerate(cases):
        pm = None
        if c["pad"] is not None:
            pm = torch.zeros(c["shape"], dtype=torch.bool)
            for b, n in enumerate(c["pad"]):
                pm[b, n:] = True
        np.random.seed(c["seed"])
        m = cmi(c["shape"], pm, c["p"], c["l"], c["kind"], c["other"], min_masks=c["min_masks"], no_overlap=c["no"],
                min_space=c["space"])
        nxt = np.random.random()
        out[f"case{i}/mask"] = m
        out[f"case{i}/next"] = np.float64(nxt)
        out[f"case{i}/args"] = np.array([c["shape"][0], c["shape"][1], c["l"], c["min_masks"], int(c["no"]),
                                         c["space"], c["seed"]], dtype=np.int64)
        out[f"case{i}/fargs"] = np.array([c["p"], c["other"]], dtype=np.float64)
        out[f"case{i}/kind"] = np.array(c["kind"])
        out[f"case{i}/pad"] = np.array(c["pad"] if c["pad"] is not None else [], dtype=np.int64)
    out["ncases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(OUT, "masks.npz"), **out)


def gen_buckets():
    _, mods = ref_shim.standalone()
    out = {}
    for (T, nb, md) in [(49, 32, 64), (749, 320, 800), (999, 320, 800), (749, 320, 1280), (1500, 320, 800)]:
        mha = mods.MultiheadAttention(64, 2, has_relative_attention_bias=True, num_buckets=nb, max_distance=md)
        ctx = torch.arange(T, dtype=torch.long)[:, None]
        mem = torch.arange(T, dtype=torch.long)[None, :]
        full = mha._relative_positions_bucket(mem - ctx, bidirectional=True)
        # Toeplitz: first column reversed + first row
        line = torch.cat([full[:, 0].flip(0), full[0, 1:]])
        for i in range(T):  # verify the Toeplitz claim on the full grid before trusting the 1-D table
            assert torch.equal(full[i], line[T - 1 - i: 2 * T - 1 - i])
        out[f"T{T}_nb{nb}_md{md}"] = line.numpy().astype(np.int32)
    np.savez_compressed(os.path.join(OUT, "buckets.npz"), **out)


def gen_tiny_wavlm():
    ref, _ = ref_shim.standalone()
    cfg = ref.WavLMConfig(dict(TINY))
    torch.manual_seed(0)
    model = ref.WavLM(cfg)
    model.eval()
    out = sd_to_np(model.state_dict())
    g = torch.Generator().manual_seed(1234)
    wav = torch.randn(2, 16000, generator=g)
    out["in/source"] = wav.numpy()
    with torch.no_grad():
        x, _ = model.extract_features(wav.clone())
        out["out/x"] = x.numpy()
        feat, _ = model.extract_features(wav.clone(), ret_conv=True)
        out["out/features_ret_conv"] = feat.numpy()
        (x3, lr), _ = model.extract_features(wav.clone(), output_layer=1, ret_layer_results=True)
        out["out/x_layer1"] = x3.numpy()
        out["out/nlayer_results_layer1"] = np.int64(len(lr))
        out["out/layer_results0"] = lr[0][0].numpy()
        out["out/layer_results1"] = lr[1][0].numpy()
        # padded batch
        pm = torch.zeros(2, 16000, dtype=torch.bool)
        pm[1, 12000:] = True
        wav_p = wav.clone()
        wav_p[1, 12000:] = 0
        xp, pmo = model.extract_features(wav_p, padding_mask=pm)
        out["in/padding_mask"] = pm.numpy()
        out["out/x_padded"] = xp.numpy()
        out["out/padding_mask_frames"] = pmo.numpy()
        # masked: the mask consumed is the one compute_mask_indices yields under the same numpy seed
        np.random.seed(123)
        xm, _ = model.extract_features(wav.clone(), mask=True)
        out["out/x_masked"] = xm.numpy()
        np.random.seed(123)
        m = ref.compute_mask_indices((2, 49), None, cfg.mask_prob, cfg.mask_length, cfg.mask_selection, cfg.mask_other,
                                     min_masks=2, no_overlap=cfg.no_mask_overlap, min_space=cfg.mask_min_space)
        out["out/mask_seed123"] = m
        # conv stack output alone
        out["out/conv_features"] = model.feature_extractor(wav).numpy()
    np.savez_compressed(os.path.join(OUT, "tiny_wavlm.npz"), **out)


def gen_tiny_large(overrides=None, fname="tiny_large.npz"):
    """WavLM-Large structure at tiny size: extractor_mode 'layer_norm' (LayerNorm + GELU after every conv block) and
    layer_norm_first=True (pre-LN encoder layers + final encoder LayerNorm).  Forward outputs and, through a scalar
    probe loss, every parameter gradient of the standalone reference (WavLM/WavLM.py with the out-of-place x + x_conv
    patch of ref_shim).  overrides + fname: variants (gen_tiny_large_convbias)."""
    ref, _ = ref_shim.standalone(differentiable=True)
    d = dict(TINY)
    d.update(extractor_mode="layer_norm", layer_norm_first=True, normalize=True)
    d.update(overrides or {})
    cfg = ref.WavLMConfig(d)
    torch.manual_seed(0)
    model = ref.WavLM(cfg)
    model.eval()
    out = sd_to_np(model.state_dict())
    g = torch.Generator().manual_seed(4321)
    wav = torch.randn(2, 16000, generator=g)
    probe = torch.randn(2, 49, 64, generator=g)
    out["in/source"] = wav.numpy()
    out["in/probe"] = probe.numpy()
    with torch.no_grad():
        out["out/conv_features"] = model.feature_extractor(wav).numpy()
        (x, lr), _ = model.extract_features(wav.clone(), ret_layer_results=True)
        out["out/x"] = x.numpy()
        out["out/n_layer_results"] = np.int64(len(lr))
        for i, r in enumerate(lr):
            out["out/layer_results%d" % i] = r[0].numpy()
    model.zero_grad()
    x, _ = model.extract_features(wav.clone())
    (x * probe).sum().backward()
    for n, p in model.named_parameters():
        out["grad/" + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    np.savez_compressed(os.path.join(OUT, fname), **out)


def gen_tiny_large_convbias():
    """tiny_large with conv_bias=True: Conv1d biases in front of the per-frame LayerNorm of every extractor block"""
    gen_tiny_large({"conv_bias": True}, "tiny_large_convbias.npz")


class _Dict:
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


def gen_tiny_pretrain(overrides=None, fname="tiny_pretrain.npz"):
    """overrides + fname: variants of the same run (gen_tiny_chanmask: channel masking on top of the time mask)"""
    WavLMModel, WavLMConfig, WavLMCriterion, _, _ = ref_shim.fairseq_wavlm()
    cfg = WavLMConfig()
    for k, v in TINY.items():
        setattr(cfg, k, v)
    for k, v in (overrides or {}).items():
        setattr(cfg, k, v)
    cfg.label_rate = 50
    cfg.final_dim = 32
    cfg.logit_temp = 0.1
    cfg.skip_masked = False
    cfg.skip_nomask = False
    cfg.untie_final_proj = False
    cfg.target_glu = bool((overrides or {}).get("target_glu", False))
    cfg.boundary_mask = False
    cfg.expand_attention_head_size = -1
    V = 23
    torch.manual_seed(0)
    model = WavLMModel(cfg, SimpleNamespace(sample_rate=16000), [_Dict(V)])
    model.train()  # dropouts are all 0; train() so that the mask/loss path is the training one
    crit = WavLMCriterion(SimpleNamespace(), 1.0, 0.0, loss_weights=[10.0])
    out = sd_to_np(model.state_dict())
    g = torch.Generator().manual_seed(1234)
    wav = torch.randn(2, 16000, generator=g)
    target = torch.randint(4, V, (2, 50), generator=g)
    pm = torch.zeros(2, 16000, dtype=torch.bool)
    sample = {"id": torch.arange(2), "net_input": {"source": wav, "padding_mask": pm}, "target_list": [target]}
    np.random.seed(123)
    loss, sample_size, log = crit(model, sample)
    loss.backward()
    np.random.seed(123)
    net = model(target_list=[target], source=wav, padding_mask=pm)
    out["in/source"] = wav.numpy()
    out["in/target"] = target.numpy()
    out["in/padding_mask"] = pm.numpy()
    out["out/loss"] = np.float64(loss.item())
    out["out/sample_size"] = np.int64(sample_size)
    for k, v in log.items():
        out["log/" + k] = np.float64(v)
    out["out/logit_m"] = net["logit_m_list"][0].detach().float().numpy()
    out["out/logit_u"] = net["logit_u_list"][0].detach().float().numpy()
    out["out/x"] = net["x"].detach().numpy()
    out["out/features_pen"] = np.float64(net["features_pen"].item())
    # the mask the forward consumed under seed 123 (padding mask given -> per-row rand draws)
    np.random.seed(123)
    from fairseq.data.data_utils import compute_mask_indices
    T = net["x"].shape[1]
    m = compute_mask_indices((2, T), torch.zeros(2, T, dtype=torch.bool), cfg.mask_prob, cfg.mask_length,
                             cfg.mask_selection, cfg.mask_other, min_masks=2, no_overlap=False, min_space=1)
    out["out/mask_seed123"] = m
    if cfg.mask_channel_prob > 0:
        # the channel mask is drawn right after the time mask from the same numpy stream (wavlm.py:405-422)
        C = cfg.encoder_embed_dim
        out["out/chan_mask_seed123"] = compute_mask_indices((2, C), None, cfg.mask_channel_prob, cfg.mask_channel_length,
                                                            cfg.mask_channel_selection, cfg.mask_channel_other,
                                                            no_overlap=cfg.no_mask_channel_overlap,
                                                            min_space=cfg.mask_channel_min_space)
    for n, p in model.named_parameters():
        out["grad/" + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    np.savez_compressed(os.path.join(OUT, fname), **out)


def gen_tiny_targetglu():
    """tiny_pretrain with target_glu=True (Linear(F, 2F) + GLU on the label embeddings, wavlm.py:322-327, 529-531)"""
    gen_tiny_pretrain({"target_glu": True}, "tiny_targetglu.npz")


def gen_tiny_activations():
    """tiny_pretrain with the feed-forward activation_fn options no released model uses (utils.get_activation_fn,
    src/fairseq/utils.py:533-555; "glu": fc1 = GLU_Linear(D, F, "swish"), src/fairseq/models/unispeech_sat/unispeech_sat.py:
    977-1007, 1065-1066)"""
    for act in ("relu", "glu", "gelu_accurate", "tanh"):
        gen_tiny_pretrain({"activation_fn": act}, "tiny_act_%s.npz" % act)


def gen_tiny_convbias():
    """tiny_pretrain with conv_bias=True ('default' extractor: block 0's bias cancels in its GroupNorm, blocks 1-6 add it
    before the GELU)"""
    gen_tiny_pretrain({"conv_bias": True}, "tiny_convbias.npz")


def gen_tiny_chanmask():
    """tiny_pretrain with mask_channel_prob 0.25 / mask_channel_length 4 (apply_mask's second half, wavlm.py:405-422)"""
    gen_tiny_pretrain({"mask_channel_prob": 0.25, "mask_channel_length": 4}, "tiny_chanmask.npz")


def gen_tiny_sat(overrides=None, fname="tiny_sat.npz", loss_weights=(10.0, 5.0, 0.0)):
    """UniSpeech-SAT at tiny size (fairseq UniSpeechSATModel, utterance_contrastive_loss with 2 in-utterance and 5
    cross-utterance instances tapped after layer 1 of 2) + HubertCriterion with loss_weights [10, 5, 0]: loss, the
    speaker logits statistics and every parameter gradient."""
    ref_shim.fairseq_wavlm()  # installs stubs + patches nothing SAT-specific
    from fairseq.criterions.hubert_criterion import HubertCriterion
    from fairseq.models.unispeech_sat import unispeech_sat as us
    ref_shim.patch_oop(us)
    cfg = us.UniSpeechSATConfig()
    for k, v in TINY.items():
        if hasattr(cfg, k):
            setattr(cfg, k, v)
    cfg.relative_position_embedding = False
    cfg.gru_rel_pos = False
    for k, v in (overrides or {}).items():
        setattr(cfg, k, v)
    cfg.label_rate = 50
    cfg.final_dim = 32
    cfg.utterance_contrastive_loss = True
    cfg.utterance_contrastive_layer = 1
    cfg.num_instances = 2
    cfg.cross_sample_instances = 5
    V = 23
    torch.manual_seed(0)
    model = us.UniSpeechSATModel(cfg, SimpleNamespace(sample_rate=16000), [_Dict(V)])
    model.train()
    crit = HubertCriterion(SimpleNamespace(), 1.0, 0.0, loss_weights=list(loss_weights))
    out = sd_to_np(model.state_dict())
    g = torch.Generator().manual_seed(99)
    wav = torch.randn(3, 16000, generator=g)
    target = torch.randint(4, V, (3, 50), generator=g)
    pm = torch.zeros(3, 16000, dtype=torch.bool)
    sample = {"id": torch.arange(3), "net_input": {"source": wav, "padding_mask": pm}, "target_list": [target]}
    np.random.seed(321)
    torch.manual_seed(77)
    loss, sample_size, log = crit(model, sample)
    loss.backward()
    np.random.seed(321)
    torch.manual_seed(77)
    net = model(target_list=[target], source=wav, padding_mask=pm)
    out["in/source"] = wav.numpy()
    out["in/target"] = target.numpy()
    out["out/loss"] = np.float64(loss.item())
    out["out/sample_size"] = np.int64(sample_size)
    out["out/loss_spk_m"] = np.float64(net["loss_spk_m"].item())
    if "prob_perplexity" in net:
        out["out/prob_perplexity"] = np.float64(net["prob_perplexity"].item())
    out["out/mean_targets"] = np.float64(float(net["mean_targets"]))
    out["out/contrastive_acc"] = np.float64(float(net["contrastive_acc"]))
    out["out/x"] = net["x"].detach().numpy()
    np.random.seed(321)
    from fairseq.data.data_utils import compute_mask_indices
    T = net["x"].shape[1]
    m = compute_mask_indices((3, T), torch.zeros(3, T, dtype=torch.bool), cfg.mask_prob, cfg.mask_length,
                             cfg.mask_selection, cfg.mask_other, min_masks=2, no_overlap=False, min_space=1)
    out["out/mask_seed321"] = m
    for n, p in model.named_parameters():
        out["grad/" + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    np.savez_compressed(os.path.join(OUT, fname), **out)


def gen_tiny_boundary():
    """tiny_pretrain with boundary_mask=True and a label sequence SHORTER than the frame sequence: exercises the two
    branches of the fairseq model no other fixture reaches -- apply_mask's boundary branch (wavlm.py:363-387: row 0 carries
    segment boundaries -> one np.random.binomial coin per segment; row 1 has none -> per-row compute_mask_indices((1, T)))
    and forward_targets' trim (wavlm.py:440-451: 49 frames vs 45 labels -> 45 frames kept)."""
    WavLMModel, WavLMConfig, WavLMCriterion, _, _ = ref_shim.fairseq_wavlm()
    cfg = WavLMConfig()
    for k, v in TINY.items():
        setattr(cfg, k, v)
    cfg.label_rate = 50
    cfg.final_dim = 32
    cfg.logit_temp = 0.1
    cfg.skip_masked = cfg.skip_nomask = cfg.untie_final_proj = cfg.target_glu = False
    cfg.boundary_mask = True
    cfg.expand_attention_head_size = -1
    V = 23
    torch.manual_seed(0)
    model = WavLMModel(cfg, SimpleNamespace(sample_rate=16000), [_Dict(V)])
    model.train()
    crit = WavLMCriterion(SimpleNamespace(), 1.0, 0.0, loss_weights=[10.0])
    out = sd_to_np(model.state_dict())
    g = torch.Generator().manual_seed(4242)
    wav = torch.randn(2, 16000, generator=g)
    target = torch.randint(4, V, (2, 45), generator=g)
    pm = torch.zeros(2, 16000, dtype=torch.bool)
    boundary = [[0, 4, 9, 13, 20, 22, 30, 37, 41, 45], []]
    sample = {"id": torch.arange(2), "net_input": {"source": wav, "padding_mask": pm, "boundary": boundary},
              "target_list": [target]}
    np.random.seed(909)
    loss, sample_size, log = crit(model, sample)
    loss.backward()
    np.random.seed(909)
    net = model(target_list=[target], source=wav, padding_mask=pm, boundary=boundary)
    out["in/source"] = wav.numpy()
    out["in/target"] = target.numpy()
    out["in/boundary0"] = np.array(boundary[0], dtype=np.int64)
    out["out/loss"] = np.float64(loss.item())
    out["out/sample_size"] = np.int64(sample_size)
    for k, v in log.items():
        out["log/" + k] = np.float64(v)
    out["out/logit_m"] = net["logit_m_list"][0].detach().float().numpy()
    out["out/x"] = net["x"].detach().numpy()
    assert net["x"].shape[1] == 45
    # the mask the forward consumed: recovered from the reference's own output (`features` holds mask_emb on masked frames
    # before pos_conv is added -> not recoverable there); re-draw it with the reference's functions instead
    np.random.seed(909)
    from fairseq.data.data_utils import compute_mask_indices
    m = np.full((2, 45), False)
    coin = np.random.binomial(1, 0.5, size=len(boundary[0]) - 1)
    for j in np.argwhere(coin == 1)[:, 0]:
        m[0][boundary[0][j]:boundary[0][j + 1]] = True
    m[1] = compute_mask_indices((1, 45), None, cfg.mask_prob, cfg.mask_length, cfg.mask_selection, cfg.mask_other,
                                min_masks=2, no_overlap=False, min_space=1)
    assert int(m.sum()) == int(sample_size), (int(m.sum()), int(sample_size))
    out["out/mask_seed909"] = m
    for n, p in model.named_parameters():
        out["grad/" + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    np.savez_compressed(os.path.join(OUT, "tiny_boundary.npz"), **out)


def gen_tiny_sat_variants():
    """tiny_sat with the gated relative position bias ON (post-LN), and UniSpeech-SAT *Large* structure (BASELINE.json
    configs[4]): extractor_mode 'layer_norm' + pre-LN encoder, whose speaker tap goes through `layer_norm_for_extract`
    (unispeech_sat.py:1197,1208)."""
    gen_tiny_sat({"relative_position_embedding": True, "gru_rel_pos": True}, "tiny_sat_relpos.npz")
    gen_tiny_sat({"relative_position_embedding": True, "gru_rel_pos": True, "extractor_mode": "layer_norm",
                  "layer_norm_first": True}, "tiny_sat_large.npz")
    # Gumbel-quantised speaker targets (unispeech_sat.py:391-404, 702-705) + codebook-diversity extra loss (820-825)
    gen_tiny_sat({"quantize_targets": True, "latent_vars": 20, "latent_groups": 2, "latent_dim": 0,
                  "latent_temp": (2.0, 0.5, 0.999995)}, "tiny_sat_quant.npz", loss_weights=(10.0, 5.0, 0.0, 0.1))


def gen_adam_clip():
    """The reference optimizer stack on fixed gradients for 4 updates: fairseq Adam (optim/adam.py:148-228) after
    multiply_grads(c) and utils.clip_grad_norm_ (utils.py:338-388) -- the sequence trainer.py:796-812 runs; in bf16 mode
    _FP16OptimizerMixin folds both factors into _multiply_factor (optim/fp16_optimizer.py:182-218), same arithmetic.
    Update 2 has a large gradient (clipped), the others are not clipped."""
    ref_shim.install()
    import fairseq  # noqa: F401
    from fairseq import utils as futils
    from fairseq.optim.adam import Adam
    g = torch.Generator().manual_seed(77)
    n = 10007
    p = torch.nn.Parameter(torch.randn(n, generator=g))
    lr, betas, eps, wd, max_norm = 5e-4, (0.9, 0.98), 1e-6, 0.01, 1.0
    opt = Adam([p], lr=lr, betas=betas, eps=eps, weight_decay=wd)
    out = {"in/p0": p.detach().numpy().copy(), "in/hyper": np.array([lr, betas[0], betas[1], eps, wd, max_norm])}
    for step in range(1, 5):
        gr = torch.randn(n, generator=g) * (2.0 if step == 2 else 0.01)
        mult = 1.0 / (3.0 + step)
        p.grad = gr.clone() * mult
        gn = futils.clip_grad_norm_([p], max_norm)
        opt.step()
        st = opt.state[p]
        out["in/grad%d" % step] = gr.numpy()
        out["in/mult%d" % step] = np.float64(mult)
        out["out/gnorm%d" % step] = np.float64(float(gn))
        out["out/p%d" % step] = p.detach().numpy().copy()
        out["out/m%d" % step] = st["exp_avg"].numpy().copy()
        out["out/v%d" % step] = st["exp_avg_sq"].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "adam_clip.npz"), **out)


def gen_tiny_ils(overrides=None, fname="tiny_ils.npz", vocabs=(23,)):
    """ILS-HuBERT at tiny size (fairseq ILSHubertModel, predict_layers [1, 2], shared final_proj / label embeddings) +
    HubertCriterion: loss, logits of both layers and every parameter gradient.  overrides / vocabs: the per-layer
    variants (gen_tiny_ils_variants)."""
    ref_shim.fairseq_wavlm()
    from fairseq.criterions.hubert_criterion import HubertCriterion
    from fairseq.models.hubert import ils_hubert as ih
    cfg = ih.ILSHubertConfig()
    for k, v in TINY.items():
        if hasattr(cfg, k):
            setattr(cfg, k, v)
    cfg.label_rate = 50
    cfg.final_dim = 32
    cfg.predict_layers = "[1,2]"
    for k, v in (overrides or {}).items():
        setattr(cfg, k, v)
    torch.manual_seed(0)
    model = ih.ILSHubertModel(cfg, SimpleNamespace(sample_rate=16000), [_Dict(V) for V in vocabs])
    model.train()
    crit = HubertCriterion(SimpleNamespace(), 1.0, 0.0, loss_weights=[10.0])
    out = sd_to_np(model.state_dict())
    g = torch.Generator().manual_seed(555)
    wav = torch.randn(2, 16000, generator=g)
    targets = [torch.randint(4, V, (2, 50), generator=g) for V in vocabs]
    pm = torch.zeros(2, 16000, dtype=torch.bool)
    sample = {"id": torch.arange(2), "net_input": {"source": wav, "padding_mask": pm}, "target_list": targets}
    np.random.seed(222)
    loss, sample_size, log = crit(model, sample)
    loss.backward()
    np.random.seed(222)
    net = model(target_list=targets, source=wav, padding_mask=pm)
    out["in/source"] = wav.numpy()
    out["in/target"] = targets[0].numpy()
    for i, t in enumerate(targets):
        out["in/target%d" % i] = t.numpy()
    out["out/loss"] = np.float64(loss.item())
    out["out/sample_size"] = np.int64(sample_size)
    out["out/n_logit_m"] = np.int64(len(net["logit_m_list"]))
    for i, l in enumerate(net["logit_m_list"]):
        out["out/logit_m%d" % i] = l.detach().float().numpy()
    out["out/x"] = net["x"].detach().numpy()
    np.random.seed(222)
    from fairseq.data.data_utils import compute_mask_indices
    T = net["x"].shape[1]
    m = compute_mask_indices((2, T), torch.zeros(2, T, dtype=torch.bool), cfg.mask_prob, cfg.mask_length,
                             cfg.mask_selection, cfg.mask_other, min_masks=2, no_overlap=False, min_space=1)
    out["out/mask_seed222"] = m
    for n, p in model.named_parameters():
        out["grad/" + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    np.savez_compressed(os.path.join(OUT, fname), **out)


def gen_tiny_ils_variants():
    """ils_hubert.py:78-107, 207-273: per-layer final_proj + label embeddings (separate_label_embeds, two label sets, with
    two label sets), and one label set per predicted layer (separate_layer_targets + weighted_sum, vocabularies of
    different size so that label_embs_concat is padded to the larger one)"""
    gen_tiny_ils({"separate_label_embeds": True}, "tiny_ils_sep_embeds.npz", vocabs=(23, 17))
    # pre-LN ILS: post_layer_norm on every tapped layer (ils_hubert.py:73-76, 186-187), LayerNorm extractor
    gen_tiny_ils({"layer_norm_first": True, "extractor_mode": "layer_norm"}, "tiny_ils_preln.npz")
    # weighted_sum (hubert_criterion.py:73-76: per-layer losses weighted by softmax(model.weights)) needs one loss per layer
    gen_tiny_ils({"separate_layer_targets": True, "separate_label_embeds": True, "weighted_sum": True},
                 "tiny_ils_sep_targets.npz", vocabs=(23, 17))


def gen_tiny_w2v2_variants():
    """negatives_from_everywhere (+ codebook_negatives) with the quantiser, and negatives_from_everywhere without it
    (wav2vec2.py:653-692)"""
    gen_tiny_w2v2({"negatives_from_everywhere": True, "codebook_negatives": 2}, "tiny_w2v2_everywhere_cb.npz")
    gen_tiny_w2v2({"negatives_from_everywhere": True, "quantize_targets": False}, "tiny_w2v2_everywhere.npz")
    # quantised encoder input through its own quantiser + project_inp, and Linear + GLU on targets and negatives
    gen_tiny_w2v2({"quantize_input": True, "target_glu": True}, "tiny_w2v2_qinput_glu.npz")
    # two-layer projection in front of the quantiser's logits (quantizer_depth 2, inner width 2 x)
    gen_tiny_w2v2({"quantizer_depth": 2, "quantizer_factor": 2}, "tiny_w2v2_qdepth.npz")


def gen_tiny_w2v2(overrides=None, fname="tiny_w2v2.npz"):
    """wav2vec 2.0 at tiny size: the reference's Wav2Vec2Model (quantize_targets with a 2 x 20 Gumbel codebook, 7 in-utterance +
    3 cross-utterance negatives) + Wav2vecCriterion(infonce, loss_weights [0.1, 10]) in train mode: loss, logits, perplexities
    and every parameter gradient.  The Gumbel noise (F.gumbel_softmax) and the negative indices (torch.randint) come from the
    torch CPU generator seeded with 31 right before the forward; the time mask from numpy seed 77."""
    ref_shim.fairseq_wavlm()
    from fairseq.criterions.wav2vec_criterion import Wav2vecCriterion
    from fairseq.models.wav2vec import wav2vec2 as w2
    cfg = w2.Wav2Vec2Config()
    for k, v in TINY.items():
        if hasattr(cfg, k):
            setattr(cfg, k, v)
    cfg.final_dim = 32
    cfg.quantize_targets = True
    cfg.latent_vars, cfg.latent_groups, cfg.latent_dim = 20, 2, 0
    cfg.latent_temp = (2.0, 0.5, 0.999995)
    cfg.num_negatives, cfg.cross_sample_negatives = 7, 3
    cfg.logit_temp = 0.1
    cfg.pretrained_path = None
    for k, v in (overrides or {}).items():
        setattr(cfg, k, v)
    torch.manual_seed(0)
    model = w2.Wav2Vec2Model(cfg)
    model.train()
    lw = [0.1, 10.0] if cfg.quantize_targets else [10.0]  # one weight per extra loss (diversity, features_pen)
    crit = Wav2vecCriterion(SimpleNamespace(), infonce=True, loss_weights=list(lw))
    out = sd_to_np(model.state_dict())
    out["in/loss_weights"] = np.array(lw, dtype=np.float64)
    g = torch.Generator().manual_seed(808)
    wav = torch.randn(3, 16000, generator=g)
    pm = torch.zeros(3, 16000, dtype=torch.bool)
    sample = {"id": torch.arange(3), "net_input": {"source": wav, "padding_mask": pm}}
    np.random.seed(77)
    torch.manual_seed(31)
    loss, sample_size, log = crit(model, sample)
    loss.backward()
    np.random.seed(77)
    torch.manual_seed(31)
    with torch.no_grad():
        net = model(source=wav, padding_mask=pm)
    out["in/source"] = wav.numpy()
    out["out/loss"] = np.float64(loss.item())
    out["out/sample_size"] = np.int64(sample_size)
    for k, v in log.items():
        out["log/" + k] = np.float64(v)
    out["out/logits"] = model.get_logits(net).float().numpy()
    if "prob_perplexity" in net:
        out["out/prob_perplexity"] = np.float64(net["prob_perplexity"].item())
        out["out/code_perplexity"] = np.float64(net["code_perplexity"].item())
    out["out/features_pen"] = np.float64(net["features_pen"].item())
    np.random.seed(77)
    from fairseq.data.data_utils import compute_mask_indices
    T = net["features"].shape[1]
    out["out/mask_seed77"] = compute_mask_indices((3, T), torch.zeros(3, T, dtype=torch.bool), cfg.mask_prob, cfg.mask_length,
                                                  cfg.mask_selection, cfg.mask_other, min_masks=2, no_overlap=False, min_space=1)
    for n, p in model.named_parameters():
        out["grad/" + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    np.savez_compressed(os.path.join(OUT, fname), **out)


def gen_mixing():
    """UtteranceMixingDataset.collater of the reference (data/audio/utterance_mixing_dataset.py:323-438) on 6 synthetic
    utterances of different lengths (one all-zero) with frame labels: crop to the shortest (random_crop), utterance mixing
    with mixing_prob 0.7 / mixing_num 2 / normalize, and a second run with pad_audio + noise mixing from an in-memory noise
    list.  The dataset object is created without __init__ (which reads manifests from disk); every attribute collater()
    touches is set by hand.  Stored: inputs, collated source / padding mask / labels, and the next numpy draw (pins the RNG
    consumption of the host-side plan)."""
    ref_shim.fairseq_wavlm()
    from fairseq.data.audio import utterance_mixing_dataset as um
    g = torch.Generator().manual_seed(606)
    lens = [5200, 4800, 6100, 4100, 5000, 4500]
    audios = [torch.randn(n, generator=g) * (0.05 + 0.02 * i) for i, n in enumerate(lens)]
    audios[3] = torch.zeros(lens[3])   # digital silence: the reference then skips the SNR draw for partners == 3
    labels = [torch.randint(4, 23, (n // 320 + 1,), generator=g) for n in lens]
    noise_bank = [(torch.randn(n, generator=g) * 0.1).numpy().astype(np.float32) for n in (3000, 7000, 2500)]
    noise_bank[1][:] = 0.0
    out = {"in/lens": np.array(lens, dtype=np.int64)}
    for i, (a, l) in enumerate(zip(audios, labels)):
        out["in/audio%d" % i] = a.numpy()
        out["in/label%d" % i] = l.numpy()
    for i, nz in enumerate(noise_bank):
        out["in/noise%d" % i] = nz

    def make(**kw):
        ds = object.__new__(um.UtteranceMixingDataset)
        base = dict(sample_rate=16000, label_rates=[50], pad_list=[1], eos_list=[2], num_labels=1, max_sample_size=4600,
                    pad_audio=False, normalize=True, random_crop=True, single_target=False, multitask=False,
                    mixing_max_len=-1, mixing_prob=0.7, mixing_num=2, mixing_noise=False, mixing_noise_prob=0.0,
                    mixing_noise_num=1, noise_list=[], noise_container={})
        base.update(kw)
        for k, v in base.items():
            setattr(ds, k, v)
        return ds

    def run(tag, seed, ds):
        samples = [{"id": i, "source": a.clone(), "label_list": [l.clone()], "boundary": []}
                   for i, (a, l) in enumerate(zip(audios, labels))]
        np.random.seed(seed)
        b = ds.collater(samples)
        out[tag + "/next"] = np.float64(np.random.random())
        out[tag + "/source"] = b["net_input"]["source"].numpy()
        out[tag + "/padding_mask"] = b["net_input"]["padding_mask"].numpy()
        out[tag + "/target"] = b["target_list"][0].numpy()
        out[tag + "/target_lengths"] = b["target_lengths_list"][0].numpy()
        out[tag + "/ntokens"] = np.int64(b["ntokens_list"][0])

    run("utt", 4711, make())
    # noise mixing: the reference reads int16 segments from an h5 file and divides by 32767; here the container is a
    # dict of in-memory int16 arrays addressed by the same "path\tkey\tstart\tend" strings
    i16 = [np.round(nz * 32767).astype(np.int16) for nz in noise_bank]
    cat = np.concatenate(i16)
    offs = np.cumsum([0] + [len(x) for x in i16])
    nlist = [{"loc": "bank\tk%d\t%d\t%d" % (i, offs[i], offs[i + 1])} for i in range(3)]
    out["in/noise_i16"] = cat
    out["in/noise_offs"] = offs.astype(np.int64)
    run("noise", 1213, make(pad_audio=True, max_sample_size=5600, normalize=False, mixing_prob=0.9, mixing_num=1,
                            mixing_noise=True, mixing_noise_prob=0.6, mixing_noise_num=2, noise_list=nlist,
                            noise_container={"bank": cat}))
    np.savez_compressed(os.path.join(OUT, "mixing.npz"), **out)


def gen_sampled_negatives():
    """wav2vec 2.0 head at function level: the reference's Wav2Vec2Model.sample_negatives and
    compute_preds called as unbound functions (they only read n_negatives / cross_sample_negatives / logit_temp), then the
    criterion's cross_entropy(sum).  y contains repeated rows (as quantised targets do) so neg_is_pos masking triggers."""
    ref_shim.fairseq_wavlm()
    from fairseq.models.wav2vec.wav2vec2 import Wav2Vec2Model
    ns = SimpleNamespace(n_negatives=7, cross_sample_negatives=3, logit_temp=0.1)
    g = torch.Generator().manual_seed(2024)
    B, T, C = 3, 40, 32
    code = torch.randn(12, C, generator=g)
    y = code[torch.randint(0, 12, (B, T), generator=g)].clone().requires_grad_(True)     # 12 distinct target vectors
    x = torch.randn(B, T, C, generator=g).requires_grad_(True)
    torch.manual_seed(31)
    negs, idxs = Wav2Vec2Model.sample_negatives(ns, y, T)
    logits = Wav2Vec2Model.compute_preds(ns, x, y, negs)
    l2 = logits.transpose(0, 2).reshape(-1, logits.size(0)).float()
    loss = torch.nn.functional.cross_entropy(l2, l2.new_zeros(l2.size(0), dtype=torch.long), reduction="sum")
    loss.backward()
    out = {"in/x": x.detach().numpy(), "in/y": y.detach().numpy(), "out/neg_idxs": idxs.numpy(),
           "out/logits": logits.detach().numpy(), "out/loss": np.float64(loss.item()),
           "out/n_masked": np.int64(torch.isinf(logits).sum().item()),
           "grad/x": x.grad.numpy(), "grad/y": y.grad.numpy()}
    np.savez_compressed(os.path.join(OUT, "sampled_negatives.npz"), **out)


if __name__ == "__main__":
    if not ref_shim.available():
        raise SystemExit("reference tree not found at %s" % ref_shim.REF_ROOT)
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1:  # python oracle/gen_golden.py gen_tiny_boundary gen_adam_clip ...: only those
        for name in sys.argv[1:]:
            globals()[name]()
        raise SystemExit(0)
    gen_masks()
    gen_buckets()
    gen_tiny_wavlm()
    gen_tiny_pretrain()
    gen_tiny_chanmask()
    gen_tiny_convbias()
    gen_tiny_targetglu()
    gen_tiny_activations()
    gen_tiny_large()
    gen_tiny_large_convbias()
    gen_tiny_sat()
    gen_tiny_sat_variants()
    gen_tiny_boundary()
    gen_adam_clip()
    gen_tiny_ils()
    gen_tiny_ils_variants()
    gen_sampled_negatives()
    gen_tiny_w2v2()
    gen_tiny_w2v2_variants()
    gen_mixing()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
