"""Times the bf16 MFMA GEMM on the shapes the WavLM-Base step actually launches (B=32 x 15 s) and prints TFLOP/s
per shape.  HIP events on torch's current stream; 3 warm-up + 10 timed launches per shape."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unispeech_amd import ops  # noqa: E402

dev = "cuda"
bf = torch.bfloat16


def timeit(fn, flops, name):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%-46s %8.3f ms  %7.1f TF/s" % (name, ms, flops / ms / 1e9), flush=True)


def rnd(*shape):
    return torch.randn(*shape, device=dev, dtype=bf)


def main():
    n = 32 * 749
    # ---- linears, forward (NN)
    for (N, K, nm) in [(2304, 768, "qkv"), (768, 768, "out_proj"), (3072, 768, "fc1"), (768, 3072, "fc2"),
                       (768, 512, "post_extract_proj")]:
        x, W, y = rnd(n, K), rnd(N, K), torch.empty(n, N, device=dev, dtype=bf)
        b = rnd(N)
        timeit(lambda: ops.gemm(x, W, y, n, N, K, lda=K, ldb=K, ldc=N, bias=b), 2.0 * n * N * K, f"fwd {nm} [{n}x{N}x{K}]")
    x, W, y, u = rnd(n, 768), rnd(3072, 768), torch.empty(n, 3072, device=dev, dtype=bf), torch.empty(n, 3072, device=dev, dtype=bf)
    b = rnd(3072)
    timeit(lambda: ops.gemm(x, W, y, n, 3072, 768, lda=768, ldb=768, ldc=3072, bias=b, epi=1, aux=u, ld_aux=3072),
           2.0 * n * 3072 * 768, "fwd fc1 + bias + gelu + aux")
    # ---- dX (A K-contig, B K-strided)
    for (N, K, nm) in [(3072, 768, "fc1"), (768, 3072, "fc2")]:
        dy, W, dx = rnd(n, N), rnd(N, K), torch.empty(n, K, device=dev, dtype=bf)
        timeit(lambda: ops.gemm(dy, W, dx, n, K, N, lda=N, ldb=K, ldc=K, transB=True), 2.0 * n * N * K, f"dX {nm} [{n}x{K}x{N}]")
    # ---- dW (both K-strided, split-K)
    for (N, K, nm) in [(3072, 768, "fc1"), (768, 3072, "fc2"), (2304, 768, "qkv")]:
        dy, x, dW = rnd(n, N), rnd(n, K), torch.empty(N, K, device=dev, dtype=bf)
        sp = ops.pick_split(N, K, (n + 63) // 64)
        timeit(lambda: ops.gemm(dy, x, dW, N, K, n, lda=N, ldb=K, ldc=K, transA=True, transB=True, split_k=sp),
               2.0 * n * N * K, f"dW {nm} [{N}x{K}x{n}] split={sp}")
    # ---- conv1 forward (overlapping rows) and conv weight grad
    B, Tin, C = 32, 47999, 512
    Tout = 23999
    x, Wf = rnd(B, Tin, C), rnd(C, 3 * C)
    y, u = torch.empty(B, Tout, C, device=dev, dtype=bf), torch.empty(B, Tout, C, device=dev, dtype=bf)
    timeit(lambda: ops.gemm(x, Wf, y, Tout, C, 3 * C, lda=2 * C, ldb=3 * C, ldc=C, batch=(B, 1), sA=(Tin * C, 0),
                            sC=(Tout * C, 0), epi=1, aux=u, ld_aux=C, sAux=(Tout * C, 0)),
           2.0 * B * Tout * C * 3 * C, "conv1 fwd + gelu + aux")
    dWf = torch.empty(C, 3 * C, device=dev, dtype=bf)
    sp = ops.pick_split(C, 3 * C, B * ((Tout + 63) // 64))
    timeit(lambda: ops.gemm(y, x, dWf, C, 3 * C, Tout, lda=C, ldb=2 * C, ldc=3 * C, transA=True, transB=True, KB=B,
                            sA_kb=Tout * C, sB_kb=Tin * C, split_k=sp),
           2.0 * B * Tout * C * 3 * C, f"conv1 dW split={sp}")
    # ---- attention batched products
    T, H, hd, D = 749, 12, 64, 768
    ld = 752
    qkv = rnd(32, T, 3 * D)
    S = torch.empty(32 * H, T, ld, device=dev, dtype=torch.float32)
    timeit(lambda: ops.gemm(qkv, qkv, S, T, T, hd, lda=3 * D, ldb=3 * D, ldc=ld, batch=(32, H), sA=(T * 3 * D, hd),
                            sB=(T * 3 * D, hd), b_off=D, sC=(H * T * ld, T * ld), alpha=0.125),
           2.0 * 32 * H * T * T * hd, "attn S = QK^T (fp32 out)")
    P = rnd(32 * H, T, ld)
    O = torch.empty(32, T, D, device=dev, dtype=bf)
    timeit(lambda: ops.gemm(P, qkv, O, T, hd, T, lda=ld, ldb=3 * D, ldc=D, transB=True, batch=(32, H),
                            sA=(H * T * ld, T * ld), sB=(T * 3 * D, hd), b_off=2 * D, sC=(T * D, hd)),
           2.0 * 32 * H * T * T * hd, "attn O = PV")
    dq = torch.empty(32, T, 3 * D, device=dev, dtype=bf)
    timeit(lambda: ops.gemm(P, qkv, dq, T, hd, T, lda=ld, ldb=3 * D, ldc=3 * D, transA=True, transB=True,
                            batch=(32, H), sA=(H * T * ld, T * ld), sB=(T * 3 * D, hd), sC=(T * 3 * D, hd), c_off=D),
           2.0 * 32 * H * T * T * hd, "attn dK = dS^T Q")


if __name__ == "__main__":
    for v in [int(a) for a in sys.argv[1:]] or [1, 0, 3]:
        print("==== tile variant", {1: "128x128 forced", 0: "auto", 2: "256x128 forced", 3: "256x256 ping-pong",
                                   4: "192x384 ping-pong"}[v & 15], "| pp launch mode", {0: "default", 1: "block per tile", 2: "persistent",
                                                                                         3: "persistent + skew"}[v >> 4])
        ops.gemm_set_variant(v)
        main()
