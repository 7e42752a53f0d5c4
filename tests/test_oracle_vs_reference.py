"""The CPU oracle against the LIVE reference at WavLM-Base width (build container only: needs /root/reference).

tests/golden/ pins the oracle at tiny size (d=64, head_dim 32, 32 buckets).  The GPU parity tests trust it at Base
width (d=768, head_dim 64, 320 buckets, max_distance 800), so this file runs the reference's own fairseq
`WavLMModel` + `WavLMCriterion` (through the inert import stubs of oracle/ref_shim.py; nothing under /root/reference
is modified) at that width with a padded row and compares loss, logits and every parameter gradient with
oracle/wavlm_oracle.py.  It also pins `adam_reference_step` + the clip rule against the reference's `Adam`,
`clip_grad_norm_` and the `_multiply_factor` semantics of its fp16/bf16 optimizer wrapper.

Skipped where /root/reference does not exist (the GPU box): there the committed goldens are the pin.
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import ref_shim  # noqa: E402
from oracle import wavlm_oracle as O  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")


class _Dict:
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


BASE = dict(
    extractor_mode="default", encoder_layers=12, encoder_embed_dim=768, encoder_ffn_embed_dim=3072,
    encoder_attention_heads=12, activation_fn="gelu", layer_norm_first=False,
    conv_feature_layers="[(512,10,5)] + [(512,3,2)] * 4 + [(512,2,2)] * 2", conv_bias=False, feature_grad_mult=0.1,
    dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, encoder_layerdrop=0.0, dropout_input=0.0,
    dropout_features=0.0, mask_length=10, mask_prob=0.8, mask_selection="static", mask_other=0,
    no_mask_overlap=False, mask_min_space=1, mask_channel_prob=0.0, conv_pos=128, conv_pos_groups=16,
    relative_position_embedding=True, num_buckets=320, max_distance=800, gru_rel_pos=True,
    label_rate=50, final_dim=256, logit_temp=0.1, skip_masked=False, skip_nomask=False, untie_final_proj=False,
    target_glu=False, boundary_mask=False, expand_attention_head_size=-1,
)


@pytest.mark.parametrize("n_layers,seconds", [(12, 5.0)])
def test_oracle_matches_live_reference_at_base_width(n_layers, seconds):
    """reference WavLMModel + WavLMCriterion (12 layers, d=768, 2 x 5 s, row 1 padded) vs the oracle: loss within
    1e-6 relative (observed: identical to the last float32 bit), logits 1e-5, every gradient within 2e-3 of its own scale
    with an absolute floor of 1e-6 x the largest gradient for the analytically-zero k_proj.bias gradients."""
    WavLMModel, WavLMConfig, WavLMCriterion, _, cmi = ref_shim.fairseq_wavlm()
    cfg = WavLMConfig()
    for k, v in BASE.items():
        setattr(cfg, k, v)
    cfg.encoder_layers = n_layers
    V = 504
    torch.manual_seed(0)
    model = WavLMModel(cfg, SimpleNamespace(sample_rate=16000), [_Dict(V)])
    model.train()
    crit = WavLMCriterion(SimpleNamespace(), 1.0, 0.0, loss_weights=[10.0])
    B, T = 2, int(16000 * seconds)
    g = torch.Generator().manual_seed(1234)
    wav = torch.randn(B, T, generator=g)
    pm = torch.zeros(B, T, dtype=torch.bool)
    pm[1, int(T * 0.7):] = True
    wav[1, int(T * 0.7):] = 0
    target = torch.randint(4, V, (B, int(50 * seconds)), generator=g)
    sample = {"id": torch.arange(B), "net_input": {"source": wav, "padding_mask": pm}, "target_list": [target]}
    np.random.seed(123)
    loss, ss, log = crit(model, sample)
    loss.backward()
    np.random.seed(123)
    with torch.no_grad():
        net = model(target_list=[target], source=wav, padding_mask=pm)

    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
    ocfg = SimpleNamespace(**BASE)
    ocfg.encoder_layers = n_layers
    Tp = net["x"].shape[1]
    np.random.seed(123)
    m = cmi((B, Tp), O.forward_padding_mask(Tp, pm), cfg.mask_prob, cfg.mask_length, cfg.mask_selection, cfg.mask_other,
            min_masks=2, no_overlap=False, min_space=1)
    onet = O.pretrain_forward(sd, ocfg, wav, [target], pm, torch.from_numpy(m), [V])
    oloss, oss, olog = O.criterion(onet, 1.0, 0.0, [10.0])
    oloss.backward()

    assert ss == oss
    assert abs(loss.item() - oloss.item()) <= 1e-6 * abs(loss.item()), (loss.item(), oloss.item())
    assert torch.equal(net["padding_mask"], onet["padding_mask"])
    for a, b in ((net["logit_m_list"][0], onet["logit_m_list"][0]), (net["logit_u_list"][0], onet["logit_u_list"][0])):
        fin = torch.isfinite(a)
        assert torch.equal(fin, torch.isfinite(b))
        assert ((a[fin] - b[fin]).abs().max() / a[fin].abs().max()).item() < 1e-5
    assert int(log["correct_m_0"]) == olog["correct_m_0"] and int(log["count_m_0"]) == olog["count_m_0"]
    gmax = max(p.grad.abs().max().item() for p in model.parameters() if p.grad is not None)
    bad = []
    for n, p in model.named_parameters():
        ref = p.grad if p.grad is not None else torch.zeros_like(p)
        og = sd[n].grad if sd[n].grad is not None else torch.zeros_like(sd[n])
        if n.startswith("feature_extractor."):
            og = og * cfg.feature_grad_mult  # GradMultiply (wavlm.py:479): the oracle returns the unscaled gradient
        e = (ref - og).abs().max().item()
        if e > 2e-3 * ref.abs().max().item() + 1e-6 * gmax:
            bad.append((n, e, ref.abs().max().item()))
    assert not bad, bad


def test_adam_clip_oracle_matches_reference_optimizer():
    """oracle.adam_reference_step + the oracle's clip rule vs the reference's own optimizer stack on the same
    gradients for 3 steps: fairseq `Adam` (optim/adam.py:148-228) driven the way `_FP16OptimizerMixin` drives it
    (optim/fp16_optimizer.py:186-218: gradients pre-multiplied by `_multiply_factor`, clip_grad_norm on the
    multiplied norm, the clip coefficient folded into the factor) with `utils.clip_grad_norm_` (utils.py:338-388)."""
    ref_shim.install()
    import fairseq  # noqa: F401
    from fairseq import utils as futils
    from fairseq.optim.adam import Adam

    torch.manual_seed(3)
    shapes = [(17, 5), (33,), (4, 3, 2)]
    params = [torch.nn.Parameter(torch.randn(*s)) for s in shapes]
    lr, betas, eps, wd, max_norm = 5e-4, (0.9, 0.98), 1e-6, 0.01, 0.5
    opt = Adam(params, lr=lr, betas=betas, eps=eps, weight_decay=wd)
    p_o = [p.detach().clone() for p in params]
    m_o = [torch.zeros_like(p) for p in params]
    v_o = [torch.zeros_like(p) for p in params]
    for step in range(1, 4):
        grads = [torch.randn(*s) * (3.0 if step == 2 else 0.05) for s in shapes]   # step 2 clips, steps 1 and 3 do not
        mult = 1.0 / (7.0 + step)                                                    # world / sample_size
        # ---- reference: multiply_grads -> clip_grad_norm -> step  (trainer.py:796-812)
        for p, gr in zip(params, grads):
            p.grad = gr.clone() * mult
        gnorm = futils.clip_grad_norm_(params, max_norm)
        opt.step()
        # ---- oracle
        gn = O.grad_norm([gr * mult for gr in grads])
        assert abs(gn - float(gnorm)) <= 1e-6 * float(gnorm)
        coef = O.clip_coef(gn, max_norm)
        for i in range(len(p_o)):
            p_o[i], m_o[i], v_o[i] = O.adam_reference_step(p_o[i], grads[i] * mult * coef, m_o[i], v_o[i], step, lr,
                                                           betas[0], betas[1], eps, wd)
        for p, q in zip(params, p_o):
            assert (p.detach() - q).abs().max().item() <= 1e-6 * p.detach().abs().max().item()
