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
