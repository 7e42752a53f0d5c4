"""In-kernel time stamps of the stored-probability dK/dV kernel (probe library built with -DFA_SP_PROBE=32:
`python tools/probe/build_probe.py attn3:sp32:-DFA_SP_PROBE=32`, run with WAVLM_HIP_LIB=tools/probe/lib/libwavlm_hip_probesp32.so).
Wave 0 of every workgroup sums, over its 12 tiles, the shader cycles between consecutive stamps; printed: the mean per tile
and segment over the workgroups, the kernel's duration and the implied clock."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from unispeech_amd import ops  # noqa: E402

B, T, H, hd = 32, 749, 12, 64
D = H * hd
dev = "cuda"
qkv = (0.5 * torch.randn(B, T, 3 * D, device=dev)).to(torch.bfloat16)
gate = 1 + 0.5 * torch.rand(B, H, T, device=dev)
tab = 0.5 * torch.randn(H, 2 * T - 1, device=dev)
dO = torch.randn(B, T, D, device=dev).to(torch.bfloat16)
O, lse, ps = ops.attn_fused_fwd(qkv, gate, tab, None, H, hd ** -0.5, 0.1, 1234, store_p=True)
for _ in range(3):
    ops.attn_fused_bwd(qkv, O, dO, lse, gate, tab, None, H, hd ** -0.5, 0.1, 1234, pstore=ps)
torch.cuda.synchronize()
L = ops._lib.lib()
for name, segs in (("wavlm_probe_read_dkv", ["loop/entry", "copies+ds_write+issue", "dP MFMA + elem f0", "dV/dK MFMA issue f0", "dP MFMA + elem f1",
                                             "dV/dK MFMA issue f1", "vmcnt wait", "barrier"]),):
    if not hasattr(L, name):
        continue
    buf = np.zeros((4096, 12), dtype=np.uint64)
    fn = getattr(L, name)
    fn.argtypes = [C.c_void_p, C.c_uint64]
    fn.restype = C.c_int
    assert fn(buf.ctypes.data, buf.nbytes) == 0
    nwg = 2304
    a = buf[:nwg].astype(np.float64)
    nq = a[:, 10].mean()
    per_tile = a[:, :8].mean(0) / nq
    print("%s: cycles per tile and segment (mean over %d workgroups, wave 0, %d tiles each)" % (name, nwg, int(nq)))
    for s_, v in zip(segs, per_tile):
        print("  %-26s %8.0f" % (s_, v))
    print("  %-26s %8.0f   (+ prologue / epilogue outside)" % ("sum per tile", per_tile.sum()))
    span = (a[:, 9] - a[:, 8])
    print("  tile loop per workgroup: mean %.0f cycles, min %.0f, max %.0f" % (span.mean(), span.min(), span.max()))
    print("  first start .. last end over all workgroups: %.0f cycles" % (a[:, 9].max() - a[:, 8].min()))
