"""Timing probes of the ping-pong GEMM: rebuild libwavlm_hip with gemm_pp.hip compiled under -DPP_PROBE=<bits> into
tools/probe/lib/libwavlm_hip_probe<bits>.so (built with -DWAVLM_EXPERIMENTAL: the product library carries no probe branch) (bit 0: no LDS fragment reads, bit 1: no operand DMA, bit 2: one MFMA
pair per section instead of four).  Results of probe builds are WRONG by construction; they only answer "what does
the K loop cost without X".  Use: WAVLM_HIP_LIB=.../libwavlm_hip_probe3.so python tools/gemm_ksweep.py 3 2048"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from unispeech_amd import build as B  # noqa: E402

B.build_library(verbose=False)
# `lab`: the whole library with -DWAVLM_EXPERIMENTAL + tools/probe/gemm_h2.hip -> tools/probe/lib/libwavlm_hip_lab.so: the paths the
# product library does not carry (gemm_h2 = variant 6 / WAVLM_GEMM_H2=1, the balanced grouped weight-gradient launch
# WAVLM_WGRAD_STREAMK=1, the lab switches of the attention kernels)
if sys.argv[1:] == ["lab"]:
    objs = []
    for src in B.SOURCES + ["../../tools/probe/gemm_h2.hip"]:
        obj = os.path.join(B.OBJ_DIR, os.path.basename(src).replace(".hip", "_lab.o"))
        subprocess.check_call([B._hipcc()] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + ["-DWAVLM_EXPERIMENTAL", "-c", os.path.normpath(os.path.join(B.CSRC, src)), "-o", obj])
        objs.append(obj)
    os.makedirs(os.path.join(ROOT, "tools", "probe", "lib"), exist_ok=True)
    out = os.path.join(ROOT, "tools", "probe", "lib", "libwavlm_hip_lab.so")
    subprocess.check_call([B._hipcc(), "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", out] + objs)
    print("built", out)
    sys.exit(0)
# arguments: <bits> (PP_PROBE value) or <tag>:-DNAME=V[,-DNAME2=V2] (free-form defines, library suffix <tag>);
# a leading "pp3:" probes gemm_pp3.hip instead (pp3:<tag>:-DP3_PROBE=1), "attn:" attn_fused.hip
for a in sys.argv[1:] or ["1", "2", "3", "4"]:
    src = "gemm_pp.hip"
    if a.startswith("pp3:"):
        src, a = "gemm_pp3.hip", a[4:]
    elif a.startswith("h2:"):  # h2:<tag>:-DH2_PIPE=0,... builds gemm_h2.hip with the defines
        src, a = "gemm_h2.hip", a[3:]
    elif a.startswith("row:"):  # row:<tag>:-DLN_PF2=1 builds rowops.hip with the defines
        src, a = "rowops.hip", a[4:]
    elif a.startswith("attn:"):  # attn:<tag>:-DFA_FWD_LAZY=1,... builds attn_fused.hip (forward + dQ kernels) with the defines
        src, a = "attn_fused.hip", a[5:]
    elif a.startswith("dkv:"):  # dkv:<tag>:-DFA_DKV_RSM=0 builds attn_fused_dkv.hip (the 32-keys-per-wave dK/dV kernels) with the defines
        src, a = "attn_fused_dkv.hip", a[4:]
    elif a.startswith("k64:"):  # k64:<tag>:-DK64_PROBE=3 builds attn_fused_dkv64.hip (64-keys-per-wave dK/dV kernel) with the defines
        src, a = "attn_fused_dkv64.hip", a[4:]
    srcs = [src]
    if a.startswith("attn3:"):  # attn3:<tag>:-DFA_ASM_DMA=0 builds all three attention kernels (both translation units)
        srcs, a = ["attn_fused.hip", "attn_fused_dkv.hip"], a[6:]
    tag, defs = (a, ["-DPP_PROBE=%d" % int(a)]) if a.isdigit() else (a.split(":")[0], [d for d in a.split(":")[1].split(",") if d])
    pobjs = []
    for src in srcs:
        obj = os.path.join(B.OBJ_DIR, "%s_probe%s.o" % (src[:-4], tag))
        subprocess.check_call([B._hipcc()] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + ["-DWAVLM_EXPERIMENTAL"] + defs + ["-c", os.path.join(B.CSRC, src), "-o", obj])
        pobjs.append(obj)
    objs = [os.path.join(B.OBJ_DIR, s.replace(".hip", ".o")) for s in B.SOURCES if s not in srcs] + pobjs
    # probe libraries live outside the package (unispeech_amd/lib/ ships ONLY libwavlm_hip.so): select one with WAVLM_HIP_LIB
    os.makedirs(os.path.join(ROOT, "tools", "probe", "lib"), exist_ok=True)
    out = os.path.join(ROOT, "tools", "probe", "lib", "libwavlm_hip_probe%s.so" % tag)
    subprocess.check_call([B._hipcc(), "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", out] + objs)
    print("built", out)
