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


def _struct_fields(name):
    src = open(os.path.join(ROOT, "include", "wavlm_hip.h")).read()
    body = src[src.index("typedef struct %s {" % name):src.index("} %s;" % name)]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for stmt in body.split("{", 1)[1].split(";"):
        stmt = stmt.strip()
        if not stmt:
            continue
        decl = stmt.split(None, 1)[1] if not stmt.startswith("const") else stmt.split(None, 2)[2]
        for nm in decl.split(","):
            fields.append(nm.replace("*", "").strip())
    return fields


def test_layer_desc_layout_matches_header():
    """wavlm_layer_desc (one encoder block per call): ctypes field order == C field order, and the C side agrees on the
    size (a wrong field type would shift every pointer after it)"""
    from unispeech_amd._lib import LayerDesc
    assert _struct_fields("wavlm_layer_desc") == [f[0] for f in LayerDesc._fields_]
    import ctypes
    d = LayerDesc()
    # sizes only: no GPU needed.  An invalid descriptor reports 0; a valid one the byte counts of its carve-up
    from unispeech_amd import _lib
    L = _lib.lib()
    assert L.wavlm_layer_saved_bytes(ctypes.byref(d)) == 0
    d.B, d.T, d.D, d.H, d.F, d.param_dtype = 2, 100, 128, 2, 256, 1
    for f in ("Wqkv", "bqkv", "Wo", "bo", "W1", "b1", "W2", "b2", "ln1_g", "ln1_b", "ln2_g", "ln2_b"):
        setattr(d, f, 4096)
    n = 2 * 100
    got = L.wavlm_layer_saved_bytes(ctypes.byref(d))
    # lse + 4 row statistics (fp32), qkv, O, s1, x1, s2 (bf16 [n, D] resp. [n, 3D]), u, hact ([n, F]); 256-byte granules
    r = lambda b: (b + 255) // 256 * 256
    want = r(2 * 2 * 100 * 4) + 4 * r(n * 4) + r(n * 3 * 128 * 2) + 4 * r(n * 128 * 2) + 2 * r(n * 256 * 2)
    assert got == want, (got, want)
    assert L.wavlm_layer_fwd_workspace_bytes(ctypes.byref(d)) == r(n * 128 * 2) + want   # saved == NULL: inference
    assert L.wavlm_layer_bwd_workspace_bytes(ctypes.byref(d)) > 10 * n * 128 * 2


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


def test_stored_probability_buffer_size_is_a_host_function():
    """wavlm_attn_fused_pstore_bytes needs no GPU: 4 KiB per (32 query rows x 64 keys) of the padded grid + the running maxima;
    0 beyond T = 1024 (callers then recompute)"""
    from unispeech_amd import _lib
    L = _lib.lib()
    B, H, T = 32, 12, 749
    nq32, nkv, Tq = 6 * 4, 12, 768
    p_bytes = B * H * nq32 * nkv * 4096
    assert L.wavlm_attn_fused_pstore_bytes(B, H, T) == (p_bytes + 255) // 256 * 256 + B * H * nkv * Tq * 4
    assert L.wavlm_attn_fused_pstore_bytes(1, 16, 1024) > 0
    assert L.wavlm_attn_fused_pstore_bytes(1, 16, 1025) == 0
    assert L.wavlm_attn_fused_pstore_bytes(0, 1, 1) == 0


def test_product_library_is_not_the_lab_build():
    """`wavlm_lab_build` exists only under -DWAVLM_EXPERIMENTAL (tools/probe/build_probe.py lab): ops.lab_build() asks the loaded
    library, so Python's slab count for grouped weight gradients always equals csrc/layer.hip's"""
    import ctypes
    from unispeech_amd import _lib
    h = ctypes.CDLL(_lib.LIB_PATH)
    assert not hasattr(h, "wavlm_lab_build")
