"""Race detector for the ping-pong GEMMs: the same launch repeated must give bit-identical results (an LDS hazard between
the asynchronous DMA and the fragment reads shows up as run-to-run differences), and must agree with an fp32 reference.
usage: python tools/gemm_determinism.py [repeats]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unispeech_amd import ops  # noqa: E402

dev, bf = "cuda", torch.bfloat16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = 32 * 749
torch.manual_seed(0)
bad = 0
cases = [  # name, M, N, K, tA, tB, split
    ("qkv fwd (192x384)", n, 2304, 768, 0, 0, 1), ("fc2 fwd (192x384, K=3072)", n, 768, 3072, 0, 0, 1),
    ("N=2048 (256x256)", n, 2048, 768, 0, 0, 1), ("fc1 dX (K-strided B)", n, 768, 3072, 0, 1, 1),
    ("fc1 dW (both K-strided, split 7)", 3072, 768, n, 1, 1, 7), ("out_proj dW (split 28)", 768, 768, n, 1, 1, 28),
    ("K tail (K=1000)", 4096, 1024, 1000, 0, 0, 1),
]
for name, M, N, K, tA, tB, split in cases:
    A = torch.randn((K, M) if tA else (M, K), device=dev, dtype=bf)
    B = torch.randn((K, N) if tB else (N, K), device=dev, dtype=bf)
    C0 = None
    diff = 0
    for r in range(reps):
        C = torch.full((M, N), float("nan"), device=dev, dtype=bf)
        ops.gemm(A, B, C, M, N, K, lda=M if tA else K, ldb=N if tB else K, ldc=N, transA=tA, transB=tB, split_k=split)
        if C0 is None:
            C0 = C
        else:
            diff += int((C.view(torch.int16) != C0.view(torch.int16)).sum().item())
    ref = (A.float().t() if tA else A.float()) @ (B.float() if tB else B.float().t())
    err = ((C0.float() - ref).abs().max() / ref.abs().max()).item()
    ok = diff == 0 and err < 1e-2
    bad += not ok
    print("%-36s %d repeats: %d differing elements, rel err vs fp32 %.2e  %s" % (name, reps, diff, err, "ok" if ok else "FAIL"), flush=True)
sys.exit(1 if bad else 0)
