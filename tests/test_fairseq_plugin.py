"""The HIP model / criterion register through the reference's own fairseq decorators (build container only: needs
/root/reference; skipped on the GPU box)."""
import pytest
import torch

from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")


def test_register_through_reference_registries():
    ref_shim.fairseq_wavlm()  # installs the import shim and imports fairseq
    from fairseq.criterions import CRITERION_REGISTRY
    from fairseq.models import ARCH_MODEL_REGISTRY, MODEL_REGISTRY, BaseFairseqModel
    from unispeech_amd import fairseq_plugin
    Model, Criterion = fairseq_plugin.register()
    assert MODEL_REGISTRY["wavlm_mi355x"] is Model and ARCH_MODEL_REGISTRY["wavlm_mi355x"] is Model
    assert "wavlm_mi355x" in CRITERION_REGISTRY and "hubert_mi355x" in CRITERION_REGISTRY
    assert issubclass(Model, BaseFairseqModel)
    # the built-ins are still there, untouched
    assert MODEL_REGISTRY["wavlm"].__module__.startswith("fairseq.")
    # surface of the reference model / criterion
    for name in ("forward", "extract_features", "get_logits", "get_targets", "get_extra_losses",
                 "remove_pretraining_modules", "upgrade_state_dict_named", "build_model", "set_num_updates",
                 "max_positions"):
        assert hasattr(Model, name), name
    for name in ("forward", "reduce_metrics", "logging_outputs_can_be_summed"):
        assert hasattr(Criterion, name), name
    # constructing through fairseq's build_model path
    from types import SimpleNamespace
    from unispeech_amd.pretrain import WavLMPretrainConfig
    from conftest import TINY
    cfg = WavLMPretrainConfig(**{k: v for k, v in TINY.items() if k in WavLMPretrainConfig.__dataclass_fields__})
    task = SimpleNamespace(cfg=SimpleNamespace(sample_rate=16000), dictionaries=[range(23)])
    torch.manual_seed(0)
    m = Model.build_model(cfg, task)
    assert m.label_embs_concat.shape == (23, 32)
    assert Criterion.logging_outputs_can_be_summed() is True
    assert "hubert_mi355x" in MODEL_REGISTRY and "ils_hubert_mi355x" in MODEL_REGISTRY
    assert "wav2vec2_mi355x" in MODEL_REGISTRY and "wav2vec_mi355x" in CRITERION_REGISTRY
    w = CRITERION_REGISTRY["wav2vec_mi355x"].build_criterion(SimpleNamespace(infonce=True, loss_weights=[0.1, 10.0], log_keys=[]),
                                                            SimpleNamespace())
    assert w.infonce and w.loss_weights == [0.1, 10.0]


def test_register_override_replaces_builtins_and_model_half_raises():
    """register(override=True): an unmodified recipe (--arch wavlm --criterion wavlm) resolves to the HIP classes through
    fairseq's own registries; the reference's --fp16 path (trainer.py:86-89 model.half()) fails with a clear message."""
    ref_shim.fairseq_wavlm()
    from fairseq.criterions import CRITERION_REGISTRY
    from fairseq.models import ARCH_MODEL_REGISTRY, MODEL_REGISTRY
    from unispeech_amd import fairseq_plugin
    fairseq_plugin.register()                      # the default names first: override must still work afterwards
    saved = (dict(MODEL_REGISTRY), dict(ARCH_MODEL_REGISTRY), dict(CRITERION_REGISTRY))
    try:
        Model, Criterion = fairseq_plugin.register(override=True)
        for name in ("wavlm", "hubert"):
            if name in saved[0]:
                assert MODEL_REGISTRY[name] is Model and ARCH_MODEL_REGISTRY[name] is Model
            assert CRITERION_REGISTRY[name] is Criterion
        from types import SimpleNamespace
        from unispeech_amd.pretrain import WavLMPretrainConfig
        from conftest import TINY
        cfg = WavLMPretrainConfig(**{k: v for k, v in TINY.items() if k in WavLMPretrainConfig.__dataclass_fields__})
        m = MODEL_REGISTRY["wavlm"].build_model(cfg, SimpleNamespace(cfg=SimpleNamespace(sample_rate=16000), dictionaries=[range(23)]))
        with pytest.raises(NotImplementedError, match="bf16"):
            m.half()
        # through FairseqCriterion.build_criterion, the way tasks build it (fairseq_criterion.py:30-59)
        ccfg = SimpleNamespace(pred_masked_weight=1.0, pred_nomask_weight=0.0, loss_weights=[10.0], log_keys=[], defer_logging=True)
        crit = CRITERION_REGISTRY["wavlm"].build_criterion(ccfg, SimpleNamespace())
        assert crit.defer_logging is True and crit.loss_weights == [10.0]
    finally:
        for reg, old in zip((MODEL_REGISTRY, ARCH_MODEL_REGISTRY, CRITERION_REGISTRY), saved):
            reg.clear()
            reg.update(old)
        fairseq_plugin._REGISTERED.discard("override")


def test_reduce_metrics_on_summed_device_style_outputs():
    """logging_outputs_can_be_summed() -> True: the Trainer hands reduce_metrics ONE dict of summed (tensor) scalars
    (trainer.py:1296-1303); the result must equal reducing the per-worker dicts."""
    from unispeech_amd.pretrain import WavLMCriterion
    a = {"loss": 100.0, "ntokens": 40, "nsentences": 2, "sample_size": 40, "loss_m_0": 90.0, "loss_features_pen": 10.0,
         "correct_m_0": 7, "count_m_0": 40, "correct_u_0": 3, "count_u_0": 20}
    b = {"loss": 60.0, "ntokens": 30, "nsentences": 2, "sample_size": 30, "loss_m_0": 55.0, "loss_features_pen": 5.0,
         "correct_m_0": 4, "count_m_0": 30, "correct_u_0": 1, "count_u_0": 25}
    want = WavLMCriterion.reduce_metrics([a, b])
    summed = {k: torch.tensor(float(a[k] + b[k]), dtype=torch.double) for k in a}
    got = WavLMCriterion.reduce_metrics([summed])
    assert want.keys() == got.keys()
    for k in want:
        assert abs(want[k] - got[k]) < 1e-12, k
