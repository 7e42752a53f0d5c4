"""Build libwavlm_hip.so (gfx950) in-tree with hipcc.  No torch, no cmake: one object per .hip, one link.

`python -m unispeech_amd.build` or `unispeech_amd.build.build_library()`.  hipcc cross-compiles without a GPU,
so this runs in the CPU-only build container; the resulting .so travels to the GPU box with the tree.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
OBJ_DIR = os.path.join(HERE, "_build")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libwavlm_hip.so")
ARCH = "gfx950"
SOURCES = ["layer.hip", "gemm_bf16.hip", "gemm_pp.hip", "gemm_w4.hip", "gemm_pp3.hip", "gemm_f32.hip", "rowops.hip", "conv0.hip", "conv0_bwd_mfma.hip", "attn.hip", "attn_fused.hip", "attn_fused_dkv.hip", "attn_fused_dkv64.hip", "posconv.hip", "posconv_direct.hip", "loss.hip", "vq.hip", "mixing.hip", "dp_rccl.hip",
           "optim.hip"]
# per-file extra flags (see the headers of those files)
EXTRA_FLAGS = {"attn_fused_dkv.hip": ["-fno-slp-vectorize"], "attn_fused_dkv64.hip": ["-fno-slp-vectorize"], "conv0_bwd_mfma.hip": ["-fno-slp-vectorize"]}
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-inline-asm", "-I", INCLUDE]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libwavlm_hip.so")
    return exe


def _deps_mtime():
    m = 0.0
    for root in (CSRC, INCLUDE):
        for f in os.listdir(root):
            if f.endswith((".hpp", ".h")):
                m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


def _compile(src, force):
    obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
    srcp = os.path.join(CSRC, src)
    newest = max(os.path.getmtime(srcp), _deps_mtime())
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
        return obj, False
    cmd = [_hipcc()] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", srcp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
    return obj, True


def build_library(force=False, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        results = list(ex.map(lambda s: _compile(s, force), SOURCES))
    objs = [o for o, _ in results]
    rebuilt = any(c for _, c in results)
    if rebuilt or not os.path.exists(LIB_PATH):
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB_PATH] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
        if verbose:
            print("[unispeech_amd.build] linked", LIB_PATH)
    elif verbose:
        print("[unispeech_amd.build] up to date:", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
