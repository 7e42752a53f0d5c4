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
    assert Criterion.logging_outputs_can_be_summed() is False
