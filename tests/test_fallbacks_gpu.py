"""The hand-written fast paths added in round 2 each keep the form they replaced behind an environment switch (A/B and
fall-back).  One bf16 training step of a 2-layer Base-width model is run in a fresh process per switch and must reproduce
the default build's loss and gradient norm: the switches select an implementation, never a different result."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import json, sys
import numpy as np, torch
sys.path.insert(0, %r)
sys.path.insert(0, %r)
from test_model_gpu import _base_models
from unispeech_amd.optim import FusedAdam
import unispeech_amd.functional as F
B, T = 2, 48000
g = torch.Generator().manual_seed(3)
wav = torch.randn(B, T, generator=g).cuda().to(torch.bfloat16)
target = torch.randint(4, 504, (B, 150), generator=g).cuda()
sample = {"id": torch.arange(B), "net_input": {"source": wav, "padding_mask": torch.zeros(B, T, dtype=torch.bool).cuda()},
          "target_list": [target]}
import os
if os.environ.get("WAVLM_TEST_DROPOUTS") == "1":   # the recipe's dropouts: the attention / LayerNorm dropout paths of every switch run
    import test_model_gpu as TM
    TM.BASE.update(dropout=0.1, attention_dropout=0.1, dropout_input=0.1, dropout_features=0.1)
model, _sd, _cfg, crit = _base_models(2)
model = model.cuda().to(torch.bfloat16).train()
opt = FusedAdam(model.parameters(), model=model, lr=1e-4)
opt.zero_grad()
np.random.seed(11); torch.manual_seed(5); F._SEED_CTR[0] = 0
loss, _, _ = crit(model, sample)
loss.backward()
gn = torch.sqrt(sum((p.grad.float() ** 2).sum() for p in model.parameters() if p.grad is not None))
print("RESULT " + json.dumps({"loss": float(loss), "gnorm": float(gn)}))
""" % (ROOT, os.path.join(ROOT, "tests"))


def _run(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    out = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


@pytest.mark.gpu
def test_environment_switches_select_an_implementation_not_a_result():
    assert torch.cuda.is_available()
    ref = _run({})
    for name, val in (("WAVLM_POSCONV_DIRECT", "0"), ("WAVLM_CONV0_BWD_MFMA", "0"), ("WAVLM_FUSE_BIAS_COLSUM", "0"),
                      ("WAVLM_CHAIN_CONSUMERS", "0"),
                      ("WAVLM_LAYER_FUSED", "0"),    # round 4: one autograd node per kernel instead of per encoder block
                      ("WAVLM_GEMM_W4", "1"),        # round 4: the four-wave 256 x 256 GEMM wherever the eight-wave one runs
                      ("WAVLM_GEMM_W4", "2")):       # ... only for K-strided x K-strided launches (weight gradients)
        got = _run({name: val})
        # same arithmetic in another order / another kernel: bf16 rounding-level agreement
        assert abs(got["loss"] - ref["loss"]) <= 2e-3 * abs(ref["loss"]), (name, got, ref)
        assert abs(got["gnorm"] - ref["gnorm"]) <= 2e-2 * abs(ref["gnorm"]), (name, got, ref)


@pytest.mark.gpu
def test_round6_opt_in_variants_are_bit_identical_to_the_default():
    """Round 6 built four variants that were measured and left OFF; each keeps the default's arithmetic and rounding points, so
    its results must equal the default's bit for bit (the tools re-run themselves, one process per switch):
      * WAVLM_ATTN_DBITS=1: the dQ kernel hands its dropout decisions to the dK/dV kernel as bit words (attention dropout 0.1
        / 0.25 / 0, padded keys, T = 63 ... 1000), and on top of it WAVLM_ATTN_DKV64=1: the 64-keys-per-wave dK/dV kernel;
      * WAVLM_WGRAD_FIXUP=1: in-kernel split-K fix-up of the grouped weight-gradient launch (Base split 2, Large split 4, ragged
        tiles; also identical from repetition to repetition, whoever arrives last).
    (WAVLM_ATTN_STORE_P=bits, the forward-written words, is covered by gpu_checks.check_dropout_exact / check_attention and
    tests/test_layer_fused_gpu.py.)"""
    for tool in ("attn_dkv64_cmp.py", "wgrad_fixup_cmp.py"):
        env = {k: v for k, v in os.environ.items() if k not in ("WAVLM_ATTN_DBITS", "WAVLM_ATTN_DKV64", "WAVLM_WGRAD_FIXUP")}
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, tool + "\n" + out.stdout[-3000:] + out.stderr[-2000:]
        assert "DIFFERENT" not in out.stdout and ": False" not in out.stdout, out.stdout[-3000:]   # (the tools' own verdict words)


@pytest.mark.gpu
def test_switch_combinations_with_the_recipe_dropouts():
    """every switch is tested alone; here four COMBINATIONS, with the recipe's dropouts on (dropout 0.1,
    attention_dropout 0.1, dropout_input / dropout_features 0.1: same seeds -> same masks whatever the kernels), against the default
    build's loss and gradient norm at bf16 rounding level: the switches compose."""
    base = {"WAVLM_TEST_DROPOUTS": "1"}
    ref = _run(base)
    assert ref["loss"] > 0 and ref["gnorm"] > 0
    combos = [
        {"WAVLM_LAYER_FUSED": "0", "WAVLM_GEMM_W4": "1", "WAVLM_POSCONV_DIRECT": "0", "WAVLM_CONV0_BWD_MFMA": "0"},
        {"WAVLM_ATTN_DBITS": "1", "WAVLM_ATTN_DKV64": "1", "WAVLM_WGRAD_FIXUP": "1", "WAVLM_FUSE_BIAS_COLSUM": "0", "WAVLM_CHAIN_CONSUMERS": "0"},
        {"WAVLM_ATTN_STORE_P": "bits", "WAVLM_LAYER_FUSED": "0", "WAVLM_GEMM_W4": "2"},
        {"WAVLM_ATTN_STORE_P": "1", "WAVLM_WGRAD_FIXUP": "1", "WAVLM_CONV0_FWD_MFMA": "0"},
    ]
    for c in combos:
        got = _run(dict(base, **c))
        assert abs(got["loss"] - ref["loss"]) <= 2e-3 * abs(ref["loss"]), (c, got, ref)
        assert abs(got["gnorm"] - ref["gnorm"]) <= 2e-2 * abs(ref["gnorm"]), (c, got, ref)
