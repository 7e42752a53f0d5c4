"""Registration of the MI355X hot path through fairseq's own plugin decorators.

Use from the reference tree:  `python train.py ... --user-dir /path/to/unispeech_amd --arch wavlm_mi355x
--criterion wavlm_mi355x` (fairseq imports this package through utils.import_user_module,
src/fairseq_cli/train.py:53).  The reference refuses duplicate registrations
(src/fairseq/models/__init__.py:124-125, src/fairseq/registry.py:66-67), so the built-in names `wavlm` / `hubert`
stay untouched and the HIP implementations register as `*_mi355x`; `register(override=True)` replaces the
built-ins in fairseq's registries instead, for drop-in use of unmodified recipes.

Registered:
  models     wavlm_mi355x  (WavLMPretrainModel  <- src/fairseq/models/wavlm/wavlm.py:255  @register_model("wavlm"))
             hubert_mi355x, unispeech_sat_mi355x, ils_hubert_mi355x  (same class; structure / heads selected by config fields)
             wav2vec2_mi355x  (Wav2Vec2Model  <- src/fairseq/models/wav2vec/wav2vec2.py:274  @register_model("wav2vec2"))
  criterions wavlm_mi355x, hubert_mi355x  (WavLMCriterion  <- criterions/wavlm_criterion.py:38, hubert_criterion.py:39)
             wav2vec_mi355x  (Wav2vecCriterion  <- criterions/wav2vec_criterion.py:36)
Tasks (`hubert_pretraining`, `utterance_mixing_pretraining`) are the reference's own: they are the *caller* of this
path; the sample dict they collate is consumed unchanged.

fairseq (with omegaconf / hydra) is not installed in the build image; importing this module without it raises
ImportError and nothing else in unispeech_amd depends on it.
"""
from dataclasses import dataclass, field, make_dataclass
from typing import List, Optional

from .pretrain import WavLMCriterion, WavLMPretrainConfig, WavLMPretrainModel

_CLASSES = None          # (Model, Criterion, ModelCfg, CritCfg), built once
_REGISTERED = set()      # {"names", "override"}


def _classes():
    global _CLASSES
    if _CLASSES is not None:
        return _CLASSES
    from fairseq.criterions import FairseqCriterion
    from fairseq.dataclass import FairseqDataclass
    from fairseq.models import BaseFairseqModel

    # dataclasses with fairseq's base so that argparse / hydra generation works (dataclass/utils.py)
    cfg_fields = [(n, f.type, field(default=f.default)) for n, f in WavLMPretrainConfig.__dataclass_fields__.items()]
    ModelCfg = make_dataclass("WavLMMI355XConfig", cfg_fields, bases=(FairseqDataclass,))

    @dataclass
    class CritCfg(FairseqDataclass):
        pred_masked_weight: float = field(default=1.0, metadata={"help": "weight for masked-frame loss"})
        pred_nomask_weight: float = field(default=0.0, metadata={"help": "weight for unmasked-frame loss"})
        loss_weights: Optional[List[float]] = field(default=None, metadata={"help": "weights of extra losses"})
        log_keys: List[str] = field(default_factory=lambda: [], metadata={"help": "output keys to log"})
        defer_logging: bool = field(default=True, metadata={
            "help": "keep logging values on the device (no .item() per micro-batch); they are all-reduced as device "
                    "scalars (logging_outputs_can_be_summed) and read once per update in reduce_metrics"})

    class Model(WavLMPretrainModel, BaseFairseqModel):
        @classmethod
        def build_model(cls, cfg, task):
            return cls(cfg, task.cfg, task.dictionaries)

    class Criterion(WavLMCriterion, FairseqCriterion):
        def __init__(self, task, pred_masked_weight, pred_nomask_weight, loss_weights=None, log_keys=None,
                     defer_logging=True):
            FairseqCriterion.__init__(self, task)  # sets self.task (+ padding_idx when the task has a target dictionary)
            WavLMCriterion.__init__(self, task, pred_masked_weight, pred_nomask_weight, loss_weights, log_keys,
                                    defer_logging=defer_logging)

        @staticmethod
        def reduce_metrics(logging_outputs) -> None:
            from fairseq import metrics
            WavLMCriterion.reduce_metrics(logging_outputs, log_scalar=lambda k, v: metrics.log_scalar(k, v, round=3))

    _CLASSES = (Model, Criterion, ModelCfg, CritCfg)
    return _CLASSES


def _w2v_classes():
    """wav2vec 2.0 / UniSpeech: Wav2Vec2Model <- models/wav2vec/wav2vec2.py:274 @register_model("wav2vec2"),
    Wav2vecCriterion <- criterions/wav2vec_criterion.py:36 @register_criterion("wav2vec")"""
    from fairseq.criterions import FairseqCriterion
    from fairseq.dataclass import FairseqDataclass
    from fairseq.models import BaseFairseqModel
    from .wav2vec2 import Wav2Vec2Config, Wav2Vec2Model, Wav2vecCriterion
    fields_ = [(n, f.type, field(default=f.default)) for n, f in Wav2Vec2Config.__dataclass_fields__.items()]
    W2VCfg = make_dataclass("Wav2Vec2MI355XConfig", fields_, bases=(FairseqDataclass,))

    @dataclass
    class W2VCritCfg(FairseqDataclass):
        infonce: bool = field(default=False, metadata={"help": "cross entropy over (positive, negatives) instead of BCE"})
        loss_weights: Optional[List[float]] = field(default=None, metadata={"help": "weights of extra losses"})
        log_keys: List[str] = field(default_factory=lambda: [], metadata={"help": "output keys to log"})

    class W2VModel(Wav2Vec2Model, BaseFairseqModel):
        @classmethod
        def build_model(cls, cfg, task=None):
            return cls(cfg)

    class W2VCriterion(Wav2vecCriterion, FairseqCriterion):
        def __init__(self, task, infonce=False, loss_weights=None, log_keys=None):
            FairseqCriterion.__init__(self, task)
            Wav2vecCriterion.__init__(self, task, infonce, loss_weights, log_keys)

        @staticmethod
        def reduce_metrics(logging_outputs) -> None:
            from fairseq import metrics
            Wav2vecCriterion.reduce_metrics(logging_outputs, log_scalar=lambda k, v: metrics.log_scalar(k, v, round=3))

    return W2VModel, W2VCriterion, W2VCfg, W2VCritCfg


def register(override: bool = False):
    """Call once per mode (importing the package via --user-dir does the default one).  Returns (model_cls,
    criterion_cls).  override=False registers the `*_mi355x` names; override=True ALSO replaces the built-in
    `wavlm` / `hubert` / `unispeech_sat` / `ils_hubert` model entries and the `wavlm` / `hubert` criteria in fairseq's
    registries, so that an unmodified recipe (`--arch wavlm --criterion wavlm`) runs on the HIP kernels."""
    from fairseq.criterions import register_criterion
    from fairseq.models import register_model
    Model, Criterion, ModelCfg, CritCfg = _classes()
    if override and "override" not in _REGISTERED:
        from fairseq.criterions import CRITERION_DATACLASS_REGISTRY, CRITERION_REGISTRY
        from fairseq.models import (ARCH_CONFIG_REGISTRY, ARCH_MODEL_NAME_REGISTRY, ARCH_MODEL_REGISTRY,
                                    MODEL_DATACLASS_REGISTRY, MODEL_REGISTRY)
        for name in ("wavlm", "hubert", "unispeech_sat", "ils_hubert"):
            if name not in MODEL_REGISTRY:
                continue  # not part of this fairseq fork's build
            MODEL_REGISTRY[name] = Model
            ARCH_MODEL_REGISTRY[name] = Model
            ARCH_MODEL_NAME_REGISTRY[name] = name
            MODEL_DATACLASS_REGISTRY[name] = ModelCfg
            ARCH_CONFIG_REGISTRY.pop(name, None)  # the dataclass defaults are the architecture
        for name in ("wavlm", "hubert"):
            if name in CRITERION_REGISTRY:
                CRITERION_REGISTRY[name] = Criterion
                CRITERION_DATACLASS_REGISTRY[name] = CritCfg
        _REGISTERED.add("override")
    if not override and "names" not in _REGISTERED:
        register_model("wavlm_mi355x", dataclass=ModelCfg)(Model)
        # the same class carries the plain HuBERT structure (no relative position bias; reference "hubert",
        # models/hubert/hubert.py:220), UniSpeech-SAT's utterance-contrastive head (utterance_contrastive_loss=True;
        # "unispeech_sat", models/unispeech_sat/unispeech_sat.py:283) and ILS-SSL (predict_layers="[4,12]"; "ils_hubert",
        # models/hubert/ils_hubert.py:60): registered under their own names for recipes that select by arch
        register_model("hubert_mi355x", dataclass=ModelCfg)(type("HubertMI355X", (Model,), {}))
        register_model("unispeech_sat_mi355x", dataclass=ModelCfg)(type("UniSpeechSATMI355X", (Model,), {}))
        register_model("ils_hubert_mi355x", dataclass=ModelCfg)(type("ILSHubertMI355X", (Model,), {}))
        register_criterion("wavlm_mi355x", dataclass=CritCfg)(Criterion)
        register_criterion("hubert_mi355x", dataclass=CritCfg)(type("HubertCriterionMI355X", (Criterion,), {}))
        W2VModel, W2VCriterion, W2VCfg, W2VCritCfg = _w2v_classes()
        register_model("wav2vec2_mi355x", dataclass=W2VCfg)(W2VModel)
        register_criterion("wav2vec_mi355x", dataclass=W2VCritCfg)(W2VCriterion)
        _REGISTERED.add("names")
    return Model, Criterion
