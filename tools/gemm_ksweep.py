"""K sweep at fixed M, N: separates the per-K-tile cost from the per-output-tile (prologue + epilogue) cost.
usage: python tools/gemm_ksweep.py <variant> [N]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unispeech_amd import ops  # noqa: E402

dev, bf = "cuda", torch.bfloat16
ops.gemm_set_variant(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2304
M = int(os.environ.get("KS_M", 256 * 94))
for K in [int(k) for k in os.environ.get("KS_K", "64,256,768,1536,3072,6144").split(",")]:
    y = torch.empty(M, N, device=dev, dtype=bf)
    if os.environ.get("KS_TT"):  # both operands K-strided (the weight-gradient form)
        x = torch.randn(K, M, device=dev, dtype=bf)
        W = torch.randn(K, N, device=dev, dtype=bf)
        f = lambda: ops.gemm(x, W, y, M, N, K, lda=M, ldb=N, ldc=N, transA=True, transB=True)
    else:
        x = torch.randn(M, K, device=dev, dtype=bf)
        W = torch.randn(N, K, device=dev, dtype=bf)
        f = lambda: ops.gemm(x, W, y, M, N, K, lda=K, ldb=K, ldc=N)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("M=%d N=%d K=%5d  %8.3f ms  %7.1f TF/s" % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
    if os.environ.get("PROBE_STAMPS"):  # PP_PROBE bit 6: phase stamps of one steady K step, waves 0 (group 0) and 4 (group 1)
        st = y.view(torch.int64).flatten()[2048:2048 + 32].view(2, 16).cpu()
        base = int(st.min())
        for g in range(2):
            print("      group %d: " % g + "  ".join("P%d[%s]" % (q + 1, " ".join("%5d" % (int(st[g, 4 * q + i]) - base) for i in range(4))) for q in range(4)))
    if os.environ.get("PROBE_CLOCK"):  # probe build with PP_PROBE bit 4: per-CU (cycles, 100-MHz ticks) for kernel / K loops
        c = y.view(torch.int64).flatten()[:1024].view(256, 4).double().cpu()
        print("      shader clock: whole kernel %.0f MHz, inside K loops %.0f MHz; K loops are %.0f %% of the kernel; kernel %.1f us"
              % ((c[:, 0] / c[:, 1]).mean() * 100, (c[:, 2] / c[:, 3]).mean() * 100, (c[:, 3] / c[:, 1]).mean() * 100, c[:, 1].mean() / 100))
