"""The C-ABI library builds for gfx950 without a GPU, loads, and exports every symbol include/wavlm_hip.h declares
(no compute calls here).  Also: the product path refuses CPU tensors instead of falling back."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "wavlm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wavlm_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    from unispeech_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build_library(verbose=False)
    import ctypes
    h = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(h, n), "not exported: " + n
        assert n in _lib.SIGNATURES, "no ctypes signature for " + n
    for n in _lib.SIGNATURES:
        assert n in names, "bound but not declared in the header: " + n
    assert _lib.lib().wavlm_abi_version() == _lib.ABI_VERSION


def test_gemm_desc_layout_matches_header():
    """field order of the ctypes struct == field order of the C struct"""
    from unispeech_amd._lib import GemmDesc
    src = open(os.path.join(ROOT, "include", "wavlm_hip.h")).read()
    body = src[src.index("typedef struct wavlm_gemm_desc {"):src.index("} wavlm_gemm_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for stmt in body.split("{", 1)[1].split(";"):
        stmt = stmt.strip()
        if not stmt:
            continue
        names = stmt.replace("*", " ").split()
        decl = stmt.split(None, 1)[1] if not stmt.startswith("const") else stmt.split(None, 2)[2]
        for nm in decl.split(","):
            fields.append(nm.replace("*", "").strip())
    assert fields == [f[0] for f in GemmDesc._fields_]


def test_no_cpu_fallback():
    from unispeech_amd import _lib, ops
    x = torch.randn(4, 64)
    with pytest.raises(_lib.WavlmHipError):
        ops.layernorm_fwd(x, None, torch.ones(64), torch.zeros(64), 1e-5)
    from unispeech_amd.wavlm import WavLM, WavLMConfig
    from conftest import TINY
    m = WavLM(WavLMConfig(dict(TINY)))
    with pytest.raises(_lib.WavlmHipError):
        m.extract_features(torch.randn(1, 4000))
