"""The `--fp16` flag on a bf16 part (unispeech_amd/precision.py): default = raise with the switch's name; with the switch =
bf16 model + the reference's dynamic loss-scaling protocol.  Host logic only (no GPU)."""
import os
import sys
from types import SimpleNamespace

import pytest
import torch

from conftest import TINY

REF = "/root/reference/src"


def test_loss_scaler_restatement_matches_the_reference_class_event_for_event():
    """same overflow / growth / threshold / tolerance / minimum-scale behaviour as optim/dynamic_loss_scaler.py on a random
    event stream (runs where the reference is on disk: the build container)"""
    path = os.path.join(REF, "fairseq", "optim", "dynamic_loss_scaler.py")
    if not os.path.exists(path):
        pytest.skip("reference not on this machine")
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_dls", path)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    from unispeech_amd.optim import DynamicLossScaler
    import random
    for kw in (dict(), dict(init_scale=128.0, scale_window=7, tolerance=0.25), dict(init_scale=4.0, scale_window=3, threshold=1.0),
               dict(init_scale=2.0 ** -10, scale_window=5, min_loss_scale=1e-4)):
        a, b = ref.DynamicLossScaler(**kw), DynamicLossScaler(**kw)
        rnd = random.Random(1)
        for _ in range(400):
            g = rnd.choice([1.0, 3.5, float("inf"), float("nan"), 0.1, 2.0, 9.0])
            ea = eb = None
            try:
                a.check_overflow(g)
                a.update()
            except (OverflowError, FloatingPointError, ZeroDivisionError) as e:  # (after "minimum loss scale reached" the reference divides by zero: training has stopped by then)
                ea = type(e)
            try:
                b.check_overflow(g)
                b.update()
            except (OverflowError, FloatingPointError, ZeroDivisionError) as e:  # (after "minimum loss scale reached" the reference divides by zero: training has stopped by then)
                eb = type(e)
            assert ea == eb and a.loss_scale == b.loss_scale and a._iter == b._iter, (kw, g)


def test_half_raises_by_default_and_converts_under_the_switch():
    from unispeech_amd import precision
    from unispeech_amd.wavlm import WavLM, WavLMConfig
    m = WavLM(WavLMConfig(dict(TINY)))
    old = precision.fp16_as_bf16()
    try:
        precision.set_fp16_as_bf16(False)
        with pytest.raises(NotImplementedError, match="WAVLM_FP16_AS_BF16"):
            m.half()
        precision.set_fp16_as_bf16(True)
        h = m.half()
        assert all(p.dtype == torch.bfloat16 for p in h.parameters())
    finally:
        precision.set_fp16_as_bf16(old)


def test_fp16_optimizer_build_follows_the_switch_and_keeps_the_scaling_protocol():
    """FP16Optimizer.build_optimizer(cfg, params) of an --fp16 run (trainer.py:296-316): raises by default; under the switch
    the front-end carries a DynamicLossScaler built from cfg.common (fp16_optimizer.py:241-268), scales the loss in
    backward(), starts every accumulation with 1 / loss_scale in the deferred factor (fp16_optimizer.py:238-239)"""
    from unispeech_amd import precision
    from unispeech_amd.optim import FairseqFusedAdam
    net = torch.nn.Linear(8, 8)
    cfg = SimpleNamespace(common=SimpleNamespace(fp16=True, bf16=False, fp16_init_scale=128, fp16_scale_window=None,
                                                 fp16_scale_tolerance=0.0, threshold_loss_scale=None, min_loss_scale=1e-4,
                                                 model_parallel_size=1),
                          distributed_training=SimpleNamespace(distributed_world_size=8),
                          optimization=SimpleNamespace(update_freq=[2]),
                          optimizer=SimpleNamespace(lr=[1e-3], adam_betas="(0.9, 0.98)", adam_eps=1e-6, weight_decay=0.0))
    old = precision.fp16_as_bf16()
    try:
        precision.set_fp16_as_bf16(False)
        with pytest.raises(NotImplementedError, match="WAVLM_FP16_AS_BF16"):
            FairseqFusedAdam.build_optimizer(cfg, list(net.parameters()))
        precision.set_fp16_as_bf16(True)
        opt = FairseqFusedAdam.build_optimizer(cfg, list(net.parameters()))
        assert opt.scaler is not None and opt.scaler.loss_scale == 128 and opt.scaler.scale_window == 2 ** 14 // 8 // 2
        opt.zero_grad()
        assert opt.fused.pending_mult == 1.0 / 128
        x = torch.randn(4, 8)
        ref = torch.autograd.grad(net(x).pow(2).sum(), list(net.parameters()))
        opt.backward(net(x).pow(2).sum())
        for p, r in zip(net.parameters(), ref):          # the arena holds loss_scale x gradient, the optimizer sees x 1/128
            assert torch.allclose(p.grad, 128 * r, rtol=1e-5, atol=1e-5)
        opt.multiply_grads(0.5)
        assert opt.fused.pending_mult == 0.5 / 128
        # an overflow: OverflowError for the Trainer (trainer.py:856-862), scale halved, next accumulation uses the new scale
        with pytest.raises(OverflowError):
            opt.scaler.check_overflow(float("inf"))
        assert opt.scaler.loss_scale == 64
        opt.zero_grad()
        assert opt.fused.pending_mult == 1.0 / 64
        # checkpoints carry the loss scale (fp16_optimizer.py:79, 90-91): a resumed --fp16 run continues at 64, not at 128
        sd = opt.state_dict()
        assert sd["loss_scale"] == 64
        opt2 = FairseqFusedAdam.build_optimizer(cfg, list(net.parameters()))
        assert opt2.scaler.loss_scale == 128
        opt2.load_state_dict(sd)
        assert opt2.scaler.loss_scale == 64 and "loss_scale" in sd      # (the caller's dict is not modified)
        # bf16 runs keep scaler None (fp16_optimizer.py:248-250)
        cfg.common.bf16 = True
        net_b = torch.nn.Linear(8, 8)
        opt_b = FairseqFusedAdam.build_optimizer(cfg, list(net_b.parameters()))
        assert opt_b.scaler is None and "loss_scale" not in opt_b.state_dict()
        opt_b.load_state_dict(dict(opt_b.state_dict(), loss_scale=32.0))    # an --fp16 checkpoint resumed with --bf16: key ignored
    finally:
        precision.set_fp16_as_bf16(old)
