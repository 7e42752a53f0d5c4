"""In-kernel split-K fix-up of the grouped weight-gradient launch (gemm_w4.hip, WAVLM_WGRAD_FIXUP=1) against slabs + reduction
launch (the default): the fix-up adds the partial sums in split order, so the results must be BIT-IDENTICAL, and
identical from repetition to repetition (the last workgroup to arrive varies).  Shapes: the encoder block of Base (split 2) and
Large (split 4) at their bench row counts, ragged tiles, accumulate into non-zero outputs.  The switch is read once per process:
this script re-runs itself.   usage: python tools/wgrad_fixup_cmp.py"""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = [(32 * 749, [(768, 3072), (3072, 768), (768, 768), (2304, 768)]), (32 * 999, [(1024, 4096), (4096, 1024), (1024, 1024), (3072, 1024)]),
         (2500, [(2304, 768), (768, 768), (3072, 768), (768, 3072)]), (9000, [(2000, 1032), (1288, 1032)]), (1000, [(520, 264), (256, 768)])]


def run(path):
    from unispeech_amd import ops
    out = {}
    for ci, (n, shapes) in enumerate(CASES):
        g = torch.Generator().manual_seed(7 + ci)
        base = [(torch.randn(n, N, generator=g).to(torch.bfloat16).cuda(), torch.randn(n, K, generator=g).to(torch.bfloat16).cuda(),
                 torch.randn(N, K, generator=g).to(torch.bfloat16).cuda()) for N, K in shapes]
        reps = []
        for rep in range(4):
            items = [(a, b, c.clone()) for a, b, c in base]
            ops.gemm_wgrad_grouped(items, torch.bfloat16)
            torch.cuda.synchronize()
            reps.append([it[2].cpu() for it in items])
        out[ci] = reps
    torch.save(out, path)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
        sys.exit(0)
    res = {}
    for tag, env in (("fixup", {"WAVLM_WGRAD_FIXUP": "1"}), ("reduce", {"WAVLM_WGRAD_FIXUP": "0"})):
        path = "/tmp/wgrad_%s.pt" % tag
        e = {k: v for k, v in os.environ.items() if k != "WAVLM_WGRAD_FIXUP"}
        e.update(env)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), path], env=e)
        res[tag] = torch.load(path)
    bad = 0
    for ci, (n, shapes) in enumerate(CASES):
        same = all(torch.equal(x, y) for x, y in zip(res["fixup"][ci][0], res["reduce"][ci][0]))
        stable = all(torch.equal(x, y) for rep in res["fixup"][ci][1:] for x, y in zip(rep, res["fixup"][ci][0]))
        print("rows %6d, %d members: fix-up == reduction launch: %s; 4 repetitions identical: %s" % (n, len(shapes), same, stable))
        bad += (not same) + (not stable)
    sys.exit(1 if bad else 0)
