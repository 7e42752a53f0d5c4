"""Vendor-library calibration: torch.matmul / F.linear (hipBLASLt / rocBLAS under PyTorch-ROCm) against this library's
hand-written MFMA GEMMs on the shapes the WavLM-Base step launches (B = 32 x 15 s, n = 23 968 rows) and on one large square
GEMM.  HIP events on torch's current stream, 3 warm-up + 10 timed launches per shape, bf16 in / bf16 out, fp32 accumulate.
Measurement only: the product path never calls the vendor library."""
import os
import sys

import torch
import torch.nn.functional as tF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unispeech_amd import ops  # noqa: E402

dev, bf = "cuda", torch.bfloat16


def timeit(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10


def rnd(*shape):
    return torch.randn(*shape, device=dev, dtype=bf)


def row(name, flops, ms_lib, ms_own):
    print("%-44s vendor %8.3f ms %7.1f TF/s | own %8.3f ms %7.1f TF/s | own/vendor time %.2f"
          % (name, ms_lib, flops / ms_lib / 1e9, ms_own, flops / ms_own / 1e9, ms_own / ms_lib), flush=True)


def main():
    n = 32 * 749
    shapes = [(2304, 768, "qkv"), (768, 768, "out_proj"), (3072, 768, "fc1"), (768, 3072, "fc2")]
    for (N, K, nm) in shapes:  # forward: y = x W^T + b
        x, W, b = rnd(n, K), rnd(N, K), rnd(N)
        y = torch.empty(n, N, device=dev, dtype=bf)
        row(f"fwd {nm} [{n}x{N}x{K}] + bias", 2.0 * n * N * K, timeit(lambda: tF.linear(x, W, b)),
            timeit(lambda: ops.gemm(x, W, y, n, N, K, lda=K, ldb=K, ldc=N, bias=b)))
    for (N, K, nm) in shapes:  # dX = dY W
        dy, W = rnd(n, N), rnd(N, K)
        dx = torch.empty(n, K, device=dev, dtype=bf)
        row(f"dX {nm} [{n}x{K}x{N}]", 2.0 * n * N * K, timeit(lambda: torch.matmul(dy, W, out=dx)),
            timeit(lambda: ops.gemm(dy, W, dx, n, K, N, lda=N, ldb=K, ldc=K, transB=True)))
    for (N, K, nm) in shapes:  # dW = dY^T X
        dy, x = rnd(n, N), rnd(n, K)
        dW = torch.empty(N, K, device=dev, dtype=bf)
        sp = ops.pick_split(N, K, (n + 63) // 64)
        row(f"dW {nm} [{N}x{K}x{n}] split={sp}", 2.0 * n * N * K, timeit(lambda: torch.matmul(dy.t(), x, out=dW)),
            timeit(lambda: ops.gemm(dy, x, dW, N, K, n, lda=N, ldb=K, ldc=K, transA=True, transB=True, split_k=sp)))
    # conv1 as the vendor library would see it after an im2col copy (the copy itself is not timed): [B*Tout, 3C] x [C, 3C]^T
    B, Tin, C, Tout = 32, 47999, 512, 23999
    xi, Wf = rnd(B * Tout, 3 * C), rnd(C, 3 * C)
    x = rnd(B, Tin, C)
    y = torch.empty(B, Tout, C, device=dev, dtype=bf)
    row("conv1 fwd [767968x512x1536] (vendor: im2col'd)", 2.0 * B * Tout * C * 3 * C, timeit(lambda: tF.linear(xi, Wf)),
        timeit(lambda: ops.gemm(x, Wf, y, Tout, C, 3 * C, lda=2 * C, ldb=3 * C, ldc=C, batch=(B, 1), sA=(Tin * C, 0),
                                sC=(Tout * C, 0))))
    del xi, x, y
    for S in (4096, 8192):
        a, b2 = rnd(S, S), rnd(S, S)
        c = torch.empty(S, S, device=dev, dtype=bf)
        row(f"square NT [{S}^3]", 2.0 * S ** 3, timeit(lambda: tF.linear(a, b2)),
            timeit(lambda: ops.gemm(a, b2, c, S, S, S, lda=S, ldb=S, ldc=S)))


if __name__ == "__main__":
    print("torch", torch.__version__, "| preferred BLAS:", torch.backends.cuda.preferred_blas_library())
    main()
