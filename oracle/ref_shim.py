"""Import shim that lets the *unmodified* reference run in the build container (test infrastructure only).

/root/reference/WavLM imports as-is.  /root/reference/src/fairseq needs omegaconf / hydra / soundfile / librosa /
h5py (absent here, no arithmetic of the path lives in them -- configuration and audio I/O only) and three numpy
aliases removed in numpy >= 1.24.  Inert stand-ins are registered for those; nothing under /root/reference is
modified or copied.  Used only by oracle/gen_golden.py and tests/test_oracle_vs_reference.py.
"""
import os
import sys
import types

import numpy as np

REF_ROOT = os.environ.get("WAVLM_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "WavLM")) and os.path.isdir(os.path.join(REF_ROOT, "src", "fairseq"))


class _Inert:
    def __init__(self, name="x"):
        self._n = name

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return _Inert(self._n)

    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Inert(n)

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)


class _InertModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Inert(name)


def install():
    for alias, typ in (("float", float), ("int", int), ("bool", bool)):
        if not hasattr(np, alias):
            setattr(np, alias, typ)
    for name in ["omegaconf", "omegaconf.dictconfig", "omegaconf._utils", "hydra", "hydra.core",
                 "hydra.core.config_store", "hydra.core.global_hydra", "hydra.core.hydra_config",
                 "hydra.core.singleton", "hydra.experimental", "soundfile", "librosa", "h5py"]:
        if name not in sys.modules:
            m = _InertModule(name)
            m.__path__ = []
            sys.modules[name] = m
    om = sys.modules["omegaconf"]
    om.II = lambda s: s
    om.MISSING = "???"
    om.DictConfig = type("DictConfig", (dict,), {})
    for p in (os.path.join(REF_ROOT, "WavLM"), os.path.join(REF_ROOT, "src")):
        if p not in sys.path:
            sys.path.insert(0, p)


def patch_oop(ref):
    """the reference's `x += x_conv` (wavlm.py:713, WavLM.py:579) breaks autograd on torch >= 2; value-identical
    out-of-place form, applied in the harness only"""
    if not getattr(ref.TransformerEncoder, "_oop_patched", False):
        import inspect
        import textwrap
        src = inspect.getsource(ref.TransformerEncoder.extract_features)
        assert "x += x_conv" in src
        ns = {}
        exec(compile(textwrap.dedent(src.replace("x += x_conv", "x = x + x_conv")), "<patched>", "exec"), vars(ref), ns)
        ref.TransformerEncoder.extract_features = ns["extract_features"]
        ref.TransformerEncoder._oop_patched = True


def standalone(differentiable=False):
    """(WavLM module of the reference, its `modules` module); differentiable=True applies the out-of-place patch"""
    install()
    import WavLM as ref_wavlm  # noqa
    import modules as ref_modules  # noqa
    if differentiable:
        patch_oop(ref_wavlm)
    return ref_wavlm, ref_modules


def fairseq_wavlm():
    """(WavLMModel, WavLMConfig, WavLMCriterion, TransformerEncoder, compute_mask_indices) of the reference"""
    install()
    import fairseq  # noqa: F401
    from fairseq.criterions.wavlm_criterion import WavLMCriterion
    from fairseq.data.data_utils import compute_mask_indices
    from fairseq.models.wavlm import wavlm as ref

    patch_oop(ref)
    return ref.WavLMModel, ref.WavLMConfig, WavLMCriterion, ref.TransformerEncoder, compute_mask_indices
