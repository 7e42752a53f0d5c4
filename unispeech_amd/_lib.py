"""ctypes binding of libwavlm_hip.so (include/wavlm_hip.h).

The product path has no CPU fallback: if the library is missing, or a kernel entry point returns an error, this
module raises.  Loading the library does not need a GPU (the CPU test-suite checks that every symbol declared
in the header is exported); calling any entry point does.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WAVLM_HIP_LIB") or os.path.join(_HERE, "lib", "libwavlm_hip.so")

_lib = None
ABI_VERSION = 21  # include/wavlm_hip.h WAVLM_HIP_ABI_VERSION this binding was written against

F32, BF16 = 0, 1

c_i32, c_i64, c_u64, c_f32, c_vp = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_void_p


class ConvRelayoutDesc(C.Structure):
    """wavlm_conv_relayout_desc (include/wavlm_hip.h)"""
    _fields_ = [("W", c_vp * 8), ("Wf", c_vp * 8), ("Wb", c_vp * 8),
                ("Cout", c_i32 * 8), ("Cin", c_i32 * 8), ("k", c_i32 * 8), ("s", c_i32 * 8),
                ("n_layers", c_i32), ("dtype", c_i32)]


class GemmDesc(C.Structure):
    _fields_ = [
        ("dtype", c_i32), ("c_dtype", c_i32),
        ("M", c_i32), ("N", c_i32), ("K", c_i32), ("KB", c_i32),
        ("transA", c_i32), ("transB", c_i32),
        ("lda", c_i64), ("ldb", c_i64), ("ldc", c_i64),
        ("sA_kb", c_i64), ("sB_kb", c_i64),
        ("batch_o", c_i32), ("batch_i", c_i32),
        ("sA_o", c_i64), ("sA_i", c_i64), ("sB_o", c_i64), ("sB_i", c_i64), ("sC_o", c_i64), ("sC_i", c_i64),
        ("A", c_vp), ("B", c_vp), ("C", c_vp),
        ("alpha", c_f32), ("epi", c_i32),
        ("bias", c_vp), ("bias_dtype", c_i32), ("sBias_o", c_i64), ("sBias_i", c_i64),
        ("aux", c_vp), ("aux_dtype", c_i32), ("ld_aux", c_i64), ("sAux_o", c_i64), ("sAux_i", c_i64),
        ("res", c_vp), ("res_dtype", c_i32), ("ld_res", c_i64), ("sRes_o", c_i64), ("sRes_i", c_i64),
        ("accumulate", c_i32), ("split_k", c_i32),
        ("workspace", c_vp), ("ws_bytes", c_u64),
        ("colsum", c_vp), ("colsum_dtype", c_i32), ("colsum_accumulate", c_i32),
    ]


class LayerDesc(C.Structure):
    """wavlm_layer_desc (include/wavlm_hip.h): one transformer encoder block per call"""
    _fields_ = [
        ("B", c_i32), ("T", c_i32), ("D", c_i32), ("H", c_i32), ("F", c_i32),
        ("pre_ln", c_i32), ("param_dtype", c_i32), ("dtab_accumulate", c_i32),
        ("eps1", c_f32), ("eps2", c_f32), ("scale", c_f32), ("p_drop", c_f32), ("p_attn", c_f32),
        ("attn_store_p", c_i32),
        ("seed_r1", c_u64), ("seed_r2", c_u64), ("seed_attn", c_u64),
        ("Wqkv", c_vp), ("bqkv", c_vp), ("Wo", c_vp), ("bo", c_vp), ("W1", c_vp), ("b1", c_vp), ("W2", c_vp), ("b2", c_vp),
        ("ln1_g", c_vp), ("ln1_b", c_vp), ("ln2_g", c_vp), ("ln2_b", c_vp),
        ("Wgate", c_vp), ("bgate", c_vp), ("grep_a", c_vp),
        ("dWqkv", c_vp), ("dbqkv", c_vp), ("dWo", c_vp), ("dbo", c_vp), ("dW1", c_vp), ("db1", c_vp), ("dW2", c_vp),
        ("db2", c_vp), ("dln1_g", c_vp), ("dln1_b", c_vp), ("dln2_g", c_vp), ("dln2_b", c_vp),
        ("dWgate", c_vp), ("dbgate", c_vp), ("dgrep_a", c_vp),
        ("db2_prev", c_vp),
        ("tab", c_vp), ("kpm", c_vp),
        ("x", c_vp), ("r_in", c_vp), ("y", c_vp), ("r_out", c_vp),
        ("saved", c_vp), ("saved_bytes", c_u64), ("workspace", c_vp), ("ws_bytes", c_u64),
        ("dy", c_vp), ("dr_out", c_vp), ("dx", c_vp), ("dr_in", c_vp), ("dtab", c_vp),
    ]


# wavlm_grad_listener: void (*)(const void* base, uint64_t bytes, void* stream, void* user)
GRAD_LISTENER = C.CFUNCTYPE(None, c_vp, c_u64, c_vp, c_vp)


# name -> (restype, argtypes); mirrors include/wavlm_hip.h one to one
SIGNATURES = {
    "wavlm_abi_version": (c_i32, []),
    "wavlm_gemm_workspace_bytes": (c_u64, [C.POINTER(GemmDesc)]),
    "wavlm_gemm": (c_i32, [C.POINTER(GemmDesc), c_vp]),
    "wavlm_gemm_grouped": (c_i32, [C.POINTER(GemmDesc), c_i32, c_vp]),
    "wavlm_layernorm_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_f32, c_i32, c_i32,
                                    c_i32, c_f32, c_u64, c_f32, c_u64, c_vp]),
    "wavlm_layernorm_bwd_workspace_bytes": (c_u64, [c_i32]),
    "wavlm_layernorm_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32,
                                    c_i32, c_i32, c_f32, c_u64, c_f32, c_u64, c_f32, c_i32, c_i32, c_vp, c_u64, c_vp]),
    "wavlm_layernorm_bwd_seg": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32,
                                        c_i32, c_i32, c_f32, c_u64, c_f32, c_u64, c_f32, c_i32, c_i32, c_i32, c_i32, c_vp, c_u64,
                                        c_vp]),
    "wavlm_colsum_workspace_bytes": (c_u64, [c_i32]),
    "wavlm_colsum": (c_i32, [c_vp, c_i64, c_i32, c_i64, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_u64, c_vp]),
    "wavlm_select_rows": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp]),
    "wavlm_gather_rows": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp]),
    "wavlm_conv_weights_relayout": (c_i32, [C.POINTER(ConvRelayoutDesc), c_vp]),
    "wavlm_conv_wgrad_scatter": (c_i32, [C.POINTER(ConvRelayoutDesc), c_i32, c_vp]),
    "wavlm_axpby": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_i64, c_f32, c_f32, c_vp]),
    "wavlm_scale_dev": (c_i32, [c_vp, c_i32, c_i64, c_vp, c_f32, c_vp]),
    "wavlm_dropout": (c_i32, [c_vp, c_vp, c_i64, c_f32, c_u64, c_i32, c_vp]),
    "wavlm_dropout_add": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_f32, c_u64, c_i32, c_vp]),
    "wavlm_sumsq_workspace_bytes": (c_u64, []),
    "wavlm_sumsq": (c_i32, [c_vp, c_i32, c_i64, c_f32, c_vp, c_vp, c_u64, c_vp]),
    "wavlm_conv0_gn_workspace_bytes": (c_u64, [c_i32, c_i64, c_i32, c_i32]),
    "wavlm_conv0_gn_gelu_fwd": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_i64, c_i32,
                                        c_i32, c_i32, c_f32, c_vp, c_u64, c_vp]),
    "wavlm_conv0_gn_bwd_workspace_bytes": (c_u64, [c_i32, c_i64, c_i32, c_i32]),
    "wavlm_conv0_gn_gelu_bwd": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp,
                                        c_i32, c_i64, c_i32, c_i32, c_i32, c_f32, c_vp, c_u64, c_vp]),
    "wavlm_conv0_ln_fwd_workspace_bytes": (c_u64, []),
    "wavlm_conv0_ln_gelu_fwd": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_i32, c_i64, c_i32, c_i32,
                                        c_i32, c_f32, c_vp, c_u64, c_vp]),
    "wavlm_conv0_ln_bwd_workspace_bytes": (c_u64, [c_i32, c_i64, c_i32, c_i32]),
    "wavlm_conv0_ln_gelu_bwd": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32,
                                        c_i64, c_i32, c_i32, c_i32, c_f32, c_f32, c_vp, c_u64, c_vp]),
    "wavlm_relpos_gather": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "wavlm_relpos_scatter": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "wavlm_gate_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                               c_vp]),
    "wavlm_gate_bwd_workspace_bytes": (c_u64, [c_i32, c_i32]),
    "wavlm_gate_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32,
                               c_i32, c_i32, c_i32, c_vp, c_u64, c_vp]),
    "wavlm_attn_softmax_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i64, c_i64, c_i32,
                                       c_i32, c_f32, c_u64, c_vp]),
    "wavlm_attn_softmax_bwd_workspace_bytes": (c_u64, [c_i32, c_i32, c_i32]),
    "wavlm_attn_softmax_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32,
                                       c_i32, c_i64, c_i64, c_i32, c_i32, c_f32, c_u64, c_vp, c_u64, c_vp]),
    "wavlm_attn_fused_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_f32, c_u64,
                                     c_vp]),
    "wavlm_attn_fused_bwd_workspace_bytes": (c_u64, [c_i32, c_i32, c_i32]),
    "wavlm_attn_fused_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32,
                                     c_i32, c_i32, c_i32, c_i32, c_f32, c_f32, c_u64, c_vp, c_u64, c_vp]),
    "wavlm_attn_fused_pstore_bytes": (c_u64, [c_i32, c_i32, c_i32]),
    "wavlm_attn_fused_dbits_bytes": (c_u64, [c_i32, c_i32, c_i32]),
    "wavlm_attn_fused_fwd_p": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_u64, c_i32, c_i32, c_i32, c_i32, c_f32, c_f32,
                                       c_u64, c_vp]),
    "wavlm_attn_fused_bwd_p": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_u64, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32,
                                       c_i32, c_i32, c_i32, c_i32, c_f32, c_f32, c_u64, c_vp, c_u64, c_vp]),
    "wavlm_posconv_weight_workspace_bytes": (c_u64, [c_i32, c_i32, c_i32]),
    "wavlm_posconv_weight_fwd": (c_i32, [c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp,
                                         c_u64, c_vp]),
    "wavlm_posconv_dw_direct_splits": (c_i32, [c_i32, c_i32]),
    "wavlm_posconv_dw_direct": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "wavlm_posconv_weight_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_u64,
                                         c_vp]),
    "wavlm_posconv_direct_supported": (c_i32, [c_i32, c_i32, c_i32]),
    "wavlm_posconv_direct": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                     c_vp]),
    "wavlm_posconv_group_major": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                          c_i32, c_vp]),
    "wavlm_l2norm_fwd": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_vp, c_i64, c_i32, c_f32, c_vp]),
    "wavlm_l2norm_bwd": (c_i32, [c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_i64, c_i32, c_vp]),
    "wavlm_ce_rows": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i64, c_i64, c_f32, c_vp]),
    "wavlm_gather_dot": (c_i32, [c_vp, c_vp, c_i32, c_vp, c_vp, c_i64, c_i32, c_i32, c_f32, c_i32, c_vp]),
    "wavlm_rows_wsum": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_vp]),
    "wavlm_bce_workspace_bytes": (c_u64, []),
    "wavlm_bce_logits": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_vp, c_u64, c_vp]),
    "wavlm_glu_fwd": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp]),
    "wavlm_glu_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp]),
    "wavlm_act_fwd": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp]),
    "wavlm_act_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp]),
    "wavlm_sum_workspace_bytes": (c_u64, []),
    "wavlm_sum_f32": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_u64, c_vp]),
    "wavlm_adam_step": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32,
                                c_i64, c_f32, c_vp, c_vp, c_f32, c_vp]),
    "wavlm_mix_workspace_bytes": (c_u64, [c_i32, c_i64]),
    "wavlm_mix_utterances": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i64, c_vp, c_i32, c_vp, c_vp, c_i32, c_f32, c_vp, c_u64, c_vp]),
    "wavlm_gumbel_vq_partial_rows": (c_u64, [c_i64]),
    "wavlm_gumbel_vq_fwd": (c_i32, [c_vp, c_i32, c_vp, c_u64, c_f32, c_i32, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "wavlm_vq_perplexity": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "wavlm_gumbel_vq_bwd": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_f32, c_i64, c_i32, c_i32, c_vp, c_vp]),
    "wavlm_layer_saved_bytes": (c_u64, [C.POINTER(LayerDesc)]),
    "wavlm_layer_fwd_workspace_bytes": (c_u64, [C.POINTER(LayerDesc)]),
    "wavlm_layer_bwd_workspace_bytes": (c_u64, [C.POINTER(LayerDesc)]),
    "wavlm_encoder_layer_fwd": (c_i32, [C.POINTER(LayerDesc), c_vp]),
    "wavlm_encoder_layer_bwd": (c_i32, [C.POINTER(LayerDesc), c_vp]),
    "wavlm_dp_set_listener": (None, [c_vp, c_vp]),
    "wavlm_dp_unique_id": (c_i32, [c_vp]),
    "wavlm_dp_init": (c_i32, [c_i32, c_i32, c_vp, c_i32]),
    "wavlm_dp_bucket_ready": (c_i32, [c_vp, c_u64, c_i32, c_vp]),
    "wavlm_dp_finish": (c_i32, [c_vp]),
    "wavlm_dp_destroy": (c_i32, []),
    "wavlm_prof_enable": (None, [c_i32]),
    "wavlm_gemm_set_variant": (None, [c_i32]),
    "wavlm_set_reserved_cus": (None, [c_i32]),
    "wavlm_get_reserved_cus": (c_i32, []),
    "wavlm_prof_collect": (c_i32, [c_i32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "wavlm_prof_collect_bytes": (C.c_double, [c_i32]),
    "wavlm_prof_collect_class": (c_i32, [c_i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "wavlm_prof_dump": (c_i32, [C.c_char_p]),
}

_ERR = {-1: "invalid argument", -2: "kernel launch failure", -3: "out of memory"}


class WavlmHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle.  Raises if the library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WavlmHipError(
                "libwavlm_hip.so not found at %s -- run `python -m unispeech_amd.build` (there is no CPU fallback)"
                % LIB_PATH)
        # torch ships its own libamdhip64 / libhsa-runtime64; it must be in the process BEFORE this library so that
        # both resolve to ONE HIP runtime (loading the system runtime first leaves torch and the kernels on two
        # different runtimes: "no ROCm-capable device is detected" at the first launch)
        import torch  # noqa: F401
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if h.wavlm_abi_version() != ABI_VERSION:
            raise WavlmHipError("libwavlm_hip.so ABI version mismatch")
        _lib = h
    return _lib


def check(rc, what):
    if rc != 0:
        raise WavlmHipError("%s failed: %s (status %d)" % (what, _ERR.get(rc, "unknown"), rc))
