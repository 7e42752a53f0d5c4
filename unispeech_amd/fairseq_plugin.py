"""Registration of the MI355X hot path through fairseq's own plugin decorators.

Use from the reference tree:  `python train.py ... --user-dir /path/to/unispeech_amd --arch wavlm_mi355x
--criterion wavlm_mi355x` (fairseq imports this package through utils.import_user_module,
src/fairseq_cli/train.py:53).  The reference refuses duplicate registrations
(src/fairseq/models/__init__.py:124-125, src/fairseq/registry.py:66-67), so the built-in names `wavlm` / `hubert`
stay untouched and the HIP implementations register as `*_mi355x`; `register(override=True)` replaces the
built-ins in fairseq's registries instead, for drop-in use of unmodified recipes.

Registered:
  models     wavlm_mi355x  (WavLMPretrainModel  <- src/fairseq/models/wavlm/wavlm.py:255  @register_model("wavlm"))
             hubert_mi355x, unispeech_sat_mi355x, ils_hubert_mi355x  (same class; structure / heads selected by config
             fields; each arch has its OWN dataclass with the field set and defaults of the reference's config for that
             arch -- HubertConfig hubert.py:36-217, UniSpeechSATConfig unispeech_sat.py:44-287, ILSHubertConfig
             ils_hubert.py:44-58 -- see ARCH_CONFIGS)
  optimizer  adam_mi355x  (FairseqFusedAdam: the FairseqOptimizer API over the flat-arena fused Adam, optim.py).  With
             register(override=True) it also stands in for `optim.FP16Optimizer` (what trainer.py:296-316 builds in bf16
             mode), and `DistributedFairseqModel` (models/distributed_fairseq_model.py:32-137) hands models of this
             package to dp.DataParallelWavLM (overlapped bucket all-reduce on a side stream) -- so the step the
             benchmark times (gradient sinks, packed q|k|v, fused Adam, overlapped reducer) is the step train.py runs.
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
_SAVED = {}              # what register(override=True) replaced, for unregister_override()

_SAT_FIELDS = ["utterance_contrastive_loss", "utterance_contrastive_layer", "num_instances", "cross_sample_instances",
               "quantize_targets", "latent_vars", "latent_groups", "latent_dim"]  # (latent_temp exists in all four configs)
_ILS_FIELDS = ["predict_layers", "separate_label_embeds", "separate_layer_targets", "weighted_sum"]
_RELPOS_FIELDS = ["relative_position_embedding", "num_buckets", "max_distance", "gru_rel_pos"]
# arch -> (fields of WavLMPretrainConfig the reference's config of that arch does NOT have, defaults that differ).
# tests/test_fairseq_plugin.py checks field sets and defaults against the reference dataclasses themselves.
ARCH_CONFIGS = {
    "wavlm": (_SAT_FIELDS + _ILS_FIELDS, {}),                                                     # wavlm.py:48-252
    "hubert": (["boundary_mask", "expand_attention_head_size"] + _RELPOS_FIELDS + _SAT_FIELDS + _ILS_FIELDS, {}),  # hubert.py:36-217
    "unispeech_sat": (_ILS_FIELDS, {}),                                                           # unispeech_sat.py:44-287
    "ils_hubert": (["boundary_mask", "gru_rel_pos", "expand_attention_head_size"] + _SAT_FIELDS,  # ils_hubert.py:44-58
                   {"max_distance": 800, "predict_layers": "[12]"}),
}


def complete_config(cfg):
    """an arch-specific config (a subset of the fields) -> the full WavLMPretrainConfig the model reads; fields the arch
    does not have take the value that switches the feature off (the WavLMPretrainConfig default)"""
    full = WavLMPretrainConfig()
    for name in WavLMPretrainConfig.__dataclass_fields__:
        if hasattr(cfg, name):
            setattr(full, name, getattr(cfg, name))
    return full


def _classes():
    global _CLASSES
    if _CLASSES is not None:
        return _CLASSES
    from fairseq.criterions import FairseqCriterion
    from fairseq.dataclass import FairseqDataclass
    from fairseq.models import BaseFairseqModel

    # dataclasses with fairseq's base so that argparse / hydra generation works (dataclass/utils.py); one per arch
    def arch_cfg(arch):
        drop, over = ARCH_CONFIGS[arch]
        fields_ = [(n, f.type, field(default=over.get(n, f.default)))
                   for n, f in WavLMPretrainConfig.__dataclass_fields__.items() if n not in drop]
        return make_dataclass("".join(w.capitalize() for w in arch.split("_")) + "MI355XConfig", fields_, bases=(FairseqDataclass,))

    ModelCfg = {arch: arch_cfg(arch) for arch in ARCH_CONFIGS}

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
            return cls(complete_config(cfg), task.cfg, task.dictionaries)

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


_OPT_CLASSES = None


def _optim_classes():
    """FairseqFusedAdam mixed with the reference's FairseqOptimizer base (so that isinstance checks of the lr schedulers,
    optim/lr_scheduler/fairseq_lr_scheduler.py:15-16, hold) + its config dataclass.  Two classes: `FusedAdamMI355X`
    (registry name adam_mi355x) and a subclass NAMED `FP16Optimizer`, the stand-in for override mode -- the Trainer stores
    and checks the optimizer's class name in checkpoints (trainer.py:391, 521), so checkpoints written by the reference's
    bf16 runs resume here and vice versa."""
    global _OPT_CLASSES
    if _OPT_CLASSES is not None:
        return _OPT_CLASSES
    from typing import Any
    from fairseq.dataclass import FairseqDataclass
    from fairseq.optim import FairseqOptimizer
    from .optim import FairseqFusedAdam

    @dataclass
    class AdamMI355XConfig(FairseqDataclass):   # the fields of FairseqAdamConfig (optim/adam.py:26-46) this path uses
        adam_betas: Any = field(default=(0.9, 0.999), metadata={"help": "betas for Adam optimizer"})
        adam_eps: float = field(default=1e-8, metadata={"help": "epsilon for Adam optimizer"})
        weight_decay: float = field(default=0.0, metadata={"help": "weight decay"})
        use_old_adam: bool = field(default=False, metadata={"help": "ignored (kept for recipe compatibility)"})
        fp16_adam_stats: bool = field(default=False, metadata={"help": "ignored: the moments are fp32 arenas"})
        tpu: bool = False
        lr: Any = field(default_factory=lambda: [1e-3])

    class FusedAdamMI355X(FairseqFusedAdam, FairseqOptimizer):
        def __init__(self, cfg, params):
            FairseqOptimizer.__init__(self, cfg)
            FairseqFusedAdam.__init__(self, cfg, params)

    FP16Standin = type("FP16Optimizer", (FusedAdamMI355X,), {
        "__doc__": "FusedAdamMI355X under the class name the reference's bf16 checkpoints carry (override mode)"})
    _OPT_CLASSES = (FusedAdamMI355X, FP16Standin, AdamMI355XConfig)
    return _OPT_CLASSES


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


def _invalidate_on_step(opt):
    from . import functional
    inner = opt.step

    def step(*a, **k):
        try:
            return inner(*a, **k)
        finally:
            functional.invalidate_derived()

    opt.step = step
    return opt


def register(override: bool = False, fp16_as_bf16=None):
    """Call once per mode (importing the package via --user-dir does the default one).  Returns (model_cls,
    criterion_cls).  override=False registers the `*_mi355x` names; override=True ALSO replaces the built-in
    `wavlm` / `hubert` / `unispeech_sat` / `ils_hubert` model entries and the `wavlm` / `hubert` criteria in fairseq's
    registries, so that an unmodified recipe (`--arch wavlm --criterion wavlm`) runs on the HIP kernels.
    fp16_as_bf16 (None = leave as the environment set it): an unmodified `--fp16` recipe runs on the bf16 kernels with the
    reference's loss-scaling protocol instead of raising (unispeech_amd/precision.py)."""
    if fp16_as_bf16 is not None:
        from . import precision
        precision.set_fp16_as_bf16(fp16_as_bf16)
    # never more host threads than the container may run (hostenv.py: a 256-thread OpenMP pool inside a 16-CPU cgroup quota
    # gets the whole process -- the launch thread included -- throttled ~90 ms of every 100 ms)
    from . import hostenv
    hostenv.cap_threads(None)
    if override:
        # data-parallel runs through the stand-in wrapper: RCCL may take as many channels (= CUs) as the persistent GEMM grids
        # leave free (dp.py); --user-dir is imported before distributed_utils.distributed_init creates the communicator
        from .dp import cap_rccl_channels
        cap_rccl_channels(log=True)
    from fairseq.criterions import register_criterion
    from fairseq.models import register_model
    Model, Criterion, ModelCfg, CritCfg = _classes()
    if override and "override" not in _REGISTERED:
        from fairseq.criterions import CRITERION_DATACLASS_REGISTRY, CRITERION_REGISTRY
        from fairseq.models import (ARCH_CONFIG_REGISTRY, ARCH_MODEL_NAME_REGISTRY, ARCH_MODEL_REGISTRY,
                                    MODEL_DATACLASS_REGISTRY, MODEL_REGISTRY)
        import fairseq.models as fmodels
        import fairseq.models.distributed_fairseq_model as fdfm
        import fairseq.optim as foptim
        _SAVED.update(FP16Optimizer=foptim.FP16Optimizer, DistributedFairseqModel=fmodels.DistributedFairseqModel)
        for name in ("wavlm", "hubert", "unispeech_sat", "ils_hubert"):
            if name not in MODEL_REGISTRY:
                continue  # not part of this fairseq fork's build
            MODEL_REGISTRY[name] = Model
            ARCH_MODEL_REGISTRY[name] = Model
            ARCH_MODEL_NAME_REGISTRY[name] = name
            MODEL_DATACLASS_REGISTRY[name] = ModelCfg[name]
            ARCH_CONFIG_REGISTRY.pop(name, None)  # the dataclass defaults are the architecture
        for name in ("wavlm", "hubert"):
            if name in CRITERION_REGISTRY:
                CRITERION_REGISTRY[name] = Criterion
                CRITERION_DATACLASS_REGISTRY[name] = CritCfg
        # the optimizer the Trainer builds in bf16 mode (trainer.py:296-316 `optim.FP16Optimizer.build_optimizer(cfg, params)`)
        # and the data-parallel wrapper it puts around the model (trainer.py:250-261 `models.DistributedFairseqModel(...)`)
        _FusedAdam, FP16Standin, _ = _optim_classes()
        orig_fp16, orig_dfm = _SAVED["FP16Optimizer"], _SAVED["DistributedFairseqModel"]

        class _FP16Dispatch(orig_fp16):
            """`optim.FP16Optimizer` in override mode: bf16 + an Adam-family optimizer -> the fused arena optimizer;
            anything else (fp16 loss scaling, other optimizers) -> the reference's class, untouched"""

            @classmethod
            def build_optimizer(cls, cfg, params, **kwargs):
                oname = getattr(cfg.optimizer, "_name", "adam")
                from . import precision
                low = getattr(cfg.common, "bf16", False) or (getattr(cfg.common, "fp16", False) and precision.fp16_as_bf16())
                if low and oname in ("adam", "adam_mi355x"):
                    return FP16Standin.build_optimizer(cfg, params, **kwargs)
                # the reference's optimizers write parameters through `p.data` (fp16_optimizer.py:155-165, optim/adam.py:172-226):
                # no version counter moves, so what an opted-in inference cache keeps is dropped after every update
                return _invalidate_on_step(orig_fp16.build_optimizer(cfg, params, **kwargs))

        from . import dp

        def _dfm(args, model, process_group, device):
            return dp.distributed_model(args, model, process_group, device, fallback=orig_dfm)

        foptim.FP16Optimizer = _FP16Dispatch
        fmodels.DistributedFairseqModel = _dfm
        fdfm.DistributedFairseqModel = _dfm
        _REGISTERED.add("override")
    if not override and "names" not in _REGISTERED:
        register_model("wavlm_mi355x", dataclass=ModelCfg["wavlm"])(Model)
        # the same class carries the plain HuBERT structure (no relative position bias; reference "hubert",
        # models/hubert/hubert.py:220), UniSpeech-SAT's utterance-contrastive head (utterance_contrastive_loss=True;
        # "unispeech_sat", models/unispeech_sat/unispeech_sat.py:283) and ILS-SSL (predict_layers="[4,12]"; "ils_hubert",
        # models/hubert/ils_hubert.py:60): registered under their own names for recipes that select by arch
        register_model("hubert_mi355x", dataclass=ModelCfg["hubert"])(type("HubertMI355X", (Model,), {}))
        register_model("unispeech_sat_mi355x", dataclass=ModelCfg["unispeech_sat"])(type("UniSpeechSATMI355X", (Model,), {}))
        register_model("ils_hubert_mi355x", dataclass=ModelCfg["ils_hubert"])(type("ILSHubertMI355X", (Model,), {}))
        register_criterion("wavlm_mi355x", dataclass=CritCfg)(Criterion)
        register_criterion("hubert_mi355x", dataclass=CritCfg)(type("HubertCriterionMI355X", (Criterion,), {}))
        W2VModel, W2VCriterion, W2VCfg, W2VCritCfg = _w2v_classes()
        register_model("wav2vec2_mi355x", dataclass=W2VCfg)(W2VModel)
        register_criterion("wav2vec_mi355x", dataclass=W2VCritCfg)(W2VCriterion)
        from fairseq.optim import register_optimizer
        FusedAdamMI355X, _, AdamCfg = _optim_classes()
        register_optimizer("adam_mi355x", dataclass=AdamCfg)(FusedAdamMI355X)
        _REGISTERED.add("names")
    return Model, Criterion


def unregister_override():
    """put `optim.FP16Optimizer` / `DistributedFairseqModel` back (tests); the model / criterion registry entries are
    restored by the caller from its own snapshot"""
    if "override" in _REGISTERED and _SAVED:
        import fairseq.models as fmodels
        import fairseq.models.distributed_fairseq_model as fdfm
        import fairseq.optim as foptim
        foptim.FP16Optimizer = _SAVED["FP16Optimizer"]
        fmodels.DistributedFairseqModel = _SAVED["DistributedFairseqModel"]
        fdfm.DistributedFairseqModel = _SAVED["DistributedFairseqModel"]
        _SAVED.clear()
    _REGISTERED.discard("override")
