"""The dK/dV kernel variants of round 6 against the kernel of rounds 1-5 on the step's shapes: same formulas and rounding
points, so every backward result must be BIT-IDENTICAL.
  A: default                         every kernel recomputes its dropout decisions (rounds 1-5)
  B: WAVLM_ATTN_DBITS=1              the dQ kernel's bit words select in the dK/dV kernel
  C: + WAVLM_ATTN_DKV64=1            + the 64-keys-per-wave dK/dV kernel
The switches are read once per process: this script re-runs itself.
usage: python tools/attn_dkv64_cmp.py            # compares B and C with A, prints one line per case"""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = [(32, 749, 12, 0.1, False), (4, 999, 16, 0.1, False), (2, 300, 2, 0.25, True), (3, 1000, 2, 0.0, True), (1, 63, 1, 0.1, False),
         (2, 257, 3, 0.1, True)]


def run(path):
    from unispeech_amd import ops
    out = {}
    for ci, (B, T, H, p, pad) in enumerate(CASES):
        g = torch.Generator().manual_seed(100 + ci)
        D = 64 * H
        qkv = (0.5 * torch.randn(B, T, 3 * D, generator=g)).to(torch.bfloat16).cuda()
        gate = (1 + 0.5 * torch.rand(B, H, T, generator=g)).cuda()
        tab = (0.5 * torch.randn(H, 2 * T - 1, generator=g)).cuda()
        dO = torch.randn(B, T, D, generator=g).to(torch.bfloat16).cuda()
        kpm = None
        if pad:
            kpm = torch.zeros(B, T, dtype=torch.uint8)
            kpm[B - 1, T - min(T // 3, 100):] = 1
            kpm = kpm.cuda()
        O, lse, _ = ops.attn_fused_fwd(qkv, gate, tab, kpm, H, 0.125, p, 4242 + ci)
        dbias = torch.zeros(3 * D, dtype=torch.float32, device="cuda")
        res = ops.attn_fused_bwd(qkv, O, dO, lse, gate, tab, kpm, H, 0.125, p, 4242 + ci, dbias=dbias)
        torch.cuda.synchronize()
        out[ci] = [t.detach().cpu() for t in res if torch.is_tensor(t)] + [dbias.cpu()]
    torch.save(out, path)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
        sys.exit(0)
    res = {}
    for tag, env in (("A", {}), ("B", {"WAVLM_ATTN_DBITS": "1"}), ("C", {"WAVLM_ATTN_DBITS": "1", "WAVLM_ATTN_DKV64": "1"})):
        path = "/tmp/attn_dkv_%s.pt" % tag
        e = {k: v for k, v in os.environ.items() if k not in ("WAVLM_ATTN_DBITS", "WAVLM_ATTN_DKV64")}
        e.update(env)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), path], env=e)
        res[tag] = torch.load(path)
    bad = 0
    for tag in ("B", "C"):
        for ci, case in enumerate(CASES):
            worst, same = 0.0, True
            for x, y in zip(res[tag][ci], res["A"][ci]):
                same = same and torch.equal(x, y)
                worst = max(worst, (x.float() - y.float()).abs().max().item() / (y.float().abs().max().item() + 1e-30))
            print("%s vs A  B=%d T=%d H=%d p=%.2f pad=%s: %s (max rel diff %.2e)" % (tag, *case, "bit-identical" if same else "DIFFERENT", worst))
            bad += not same
    sys.exit(1 if bad else 0)
