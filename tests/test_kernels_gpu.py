"""GPU parity tests of every kernel / fused autograd function against plain PyTorch fp32 references (pytest -m gpu)."""
import pytest
import torch

import gpu_checks


@pytest.mark.gpu
@pytest.mark.parametrize("group", sorted(gpu_checks.GROUPS))
def test_kernel_group(group):
    assert torch.cuda.is_available()
    results = gpu_checks.GROUPS[group]()
    bad = [(n, e, t) for (n, e, t) in results if not (e <= t)]
    assert not bad, "\n".join(f"{n}: err={e:.3e} tol={t:.1e}" for n, e, t in bad)


@pytest.mark.gpu
def test_conv0_layer_norm_block_valu_forms():
    """WAVLM_CONV0_FWD_MFMA=0 / WAVLM_CONV0_BWD_MFMA=0 keep the VALU kernels of the LayerNorm-mode conv0 block for bf16 at
    C = 512 (the switches are read once per process): the same check group in a fresh process."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, WAVLM_CONV0_FWD_MFMA="0", WAVLM_CONV0_BWD_MFMA="0")
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_checks.py"), "conv0_ln"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
