"""An unmodified `--fp16` recipe under the fp16-as-bf16 switch (unispeech_amd/precision.py), on the HIP kernels: the
Trainer's sequence for an fp16 run -- model.half() (trainer.py:86-89), fp16 batch (trainer.py:1141-1152),
FP16Optimizer.build_optimizer -> front-end with a DynamicLossScaler, zero_grad / backward(loss) / multiply_grads /
clip_grad_norm / step (trainer.py:697-860) -- must (a) produce the same update as the bf16 run of the same step (the loss
scale is a power of two: scaling the loss and unscaling in the update is exact in bf16 / fp32 up to the range), and (b) on
a non-finite gradient raise OverflowError BEFORE anything is updated, with the scale halved (trainer.py:856-862 then
zero_grads and goes on)."""
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from conftest import TINY

pytestmark = pytest.mark.gpu
V = 60


def _cfgs(fp16):
    common = NS(fp16=fp16, bf16=not fp16, fp16_init_scale=128, fp16_scale_window=4, fp16_scale_tolerance=0.0,
                threshold_loss_scale=None, min_loss_scale=1e-4, model_parallel_size=1)
    return NS(common=common, distributed_training=NS(distributed_world_size=1), optimization=NS(update_freq=[1]),
              optimizer=NS(lr=[1e-3], adam_betas="(0.9, 0.98)", adam_eps=1e-6, weight_decay=0.01))


def _one_update(fp16):
    from unispeech_amd import functional as F
    from unispeech_amd import precision
    from unispeech_amd.optim import FairseqFusedAdam
    from unispeech_amd.pretrain import WavLMCriterion, WavLMPretrainConfig, WavLMPretrainModel
    c = dict(TINY)
    c.update(encoder_embed_dim=128, encoder_ffn_embed_dim=256, encoder_attention_heads=2,
             conv_feature_layers="[(64,10,5)] + [(64,3,2)] * 4 + [(64,2,2)] * 2")
    cfg = WavLMPretrainConfig(**{k: v for k, v in c.items() if k in WavLMPretrainConfig.__dataclass_fields__})
    torch.manual_seed(0)
    model = WavLMPretrainModel(cfg, None, [range(V)]).cuda()
    model = model.half() if fp16 else model.to(torch.bfloat16)
    assert next(model.parameters()).dtype == torch.bfloat16
    model.train()
    opt = FairseqFusedAdam.build_optimizer(_cfgs(fp16), [p for p in model.parameters() if p.requires_grad])
    crit = WavLMCriterion(None, 1.0, 0.0, loss_weights=[10.0])
    B, T = 2, 16000
    g = torch.Generator().manual_seed(3)
    wav = torch.randn(B, T, generator=g).cuda()
    wav = wav.half() if fp16 else wav.to(torch.bfloat16)      # trainer.py:1141-1152 casts the batch like the model
    pm = torch.zeros(B, T, dtype=torch.bool)
    sample = {"id": torch.arange(B), "net_input": {"source": wav, "padding_mask": pm.cuda(), "padding_mask_cpu": pm},
              "target_list": [torch.randint(4, V, (B, 50), generator=g).cuda()]}
    np.random.seed(5)
    F._SEED_CTR[0] = 0
    opt.zero_grad()
    loss, ss, _ = crit(model, sample)
    opt.backward(loss)
    opt.multiply_grads(1.0 / ss)
    gn = opt.clip_grad_norm(1.0)
    opt.step()
    torch.cuda.synchronize()
    return model, opt, crit, sample, float(gn), opt.fused.master.detach().float().cpu().clone()


def test_fp16_recipe_update_equals_bf16_update_and_overflow_skips():
    from unispeech_amd import precision
    old = precision.fp16_as_bf16()
    try:
        precision.set_fp16_as_bf16(False)
        with pytest.raises(NotImplementedError, match="WAVLM_FP16_AS_BF16"):
            _one_update(True)
        _, _, _, _, gn_b, w_b = _one_update(False)
        precision.set_fp16_as_bf16(True)
        model, opt, crit, sample, gn_h, w_h = _one_update(True)
        assert opt.scaler is not None and opt.scaler.loss_scale == 128
        assert abs(gn_h - gn_b) <= 2e-2 * gn_b, (gn_h, gn_b)          # the norm the Trainer logs is the UNSCALED one
        # the fp16 waveform differs from the bf16 one in its rounding (11 against 8 significand bits before the cast at the
        # model's door), so "equal" is to bf16 resolution of the update, not bit for bit
        d = (w_h - w_b).abs().max().item()
        moved = (w_b - w_b.mean()).abs().max().item()
        assert d <= 2.5e-3, "update under loss scaling differs from the bf16 update by %.3e (lr 1e-3)" % d
        assert moved > 0
        # three more clean updates: the window of 4 is reached -> the scale doubles (dynamic_loss_scaler.py:30-34)
        for _ in range(3):
            opt.zero_grad()
            loss, ss, _ = crit(model, sample)
            opt.backward(loss)
            opt.multiply_grads(1.0 / ss)
            opt.clip_grad_norm(1.0)
            opt.step()
        assert opt.scaler.loss_scale == 256
        # overflow: a non-finite gradient -> OverflowError out of clip_grad_norm, nothing updated, scale halved
        before = opt.fused.master.clone()
        step_before = opt.fused.step_count
        opt.zero_grad()
        loss, ss, _ = crit(model, sample)
        opt.backward(loss)
        opt.fused.flat_grad[7] = float("inf")
        opt.multiply_grads(1.0 / ss)
        with pytest.raises(OverflowError):
            opt.clip_grad_norm(1.0)
        opt.zero_grad()                                  # what the Trainer does on overflow (trainer.py:856-862)
        torch.cuda.synchronize()
        assert opt.scaler.loss_scale == 128 and opt.fused.step_count == step_before
        assert torch.equal(before, opt.fused.master)
        assert opt.fused.pending_mult == 1.0 / 128
    finally:
        precision.set_fp16_as_bf16(old)
