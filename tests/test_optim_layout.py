"""Arena layout decisions of the flat optimizer that need no GPU (host logic only)."""
import copy

import torch

from conftest import TINY


def _model():
    from unispeech_amd.wavlm import WavLM, WavLMConfig
    torch.manual_seed(0)
    return WavLM(WavLMConfig(dict(TINY)))


def test_deep_copied_model_still_packs_qkv_from_a_bare_parameter_list():
    """fairseq hands its optimizer `model.parameters()` only (trainer.py:275-316); the attention blocks tag their q|k|v
    parameters so that the arena still lays them out back to back.  nn.Parameter.__deepcopy__ drops Python attributes: an
    EMA / copied model must be re-tagged (MultiheadAttention.__setstate__), with ITS OWN modules as owners, or it would
    silently get the unpacked layout (and a native optimizer state dict that no longer matches the original's)."""
    from unispeech_amd.optim import FusedAdam
    # (the whole model cannot be deep-copied: old-style weight_norm keeps a non-leaf `weight` on pos_conv -- the reference's
    # modules have the same limitation; copies are made per sub-module / through state dicts)
    m = _model().encoder.layers
    c = copy.deepcopy(m)
    for orig, cp in zip(m, c):
        assert cp.self_attn.q_proj.weight._wl_pack_owner is cp.self_attn
        assert orig.self_attn.q_proj.weight._wl_pack_owner is orig.self_attn
    a = FusedAdam(list(m.parameters()))
    b = FusedAdam(list(c.parameters()))
    n = len(m)
    assert a.packed_groups == 2 * n and b.packed_groups == 2 * n     # q|k|v weights + biases per block
    assert a._layout() == b._layout()
    for layer in c:                                   # the copy's packed views are views of ITS arena
        at = layer.self_attn
        assert at._packed is not None and at._packed[0].data_ptr() == at.q_proj.weight.data_ptr()
        lo = b.flat_param.data_ptr()
        assert lo <= at._packed[0].data_ptr() < lo + b.flat_param.numel() * b.flat_param.element_size()
    # conversion keeps the tags too
    h = copy.deepcopy(m).to(torch.bfloat16)
    assert FusedAdam(list(h.parameters())).packed_groups == 2 * n


def test_hostenv_reads_the_cgroup_quota_and_caps_the_pools(tmp_path, monkeypatch):
    """hostenv.usable_cpus = min(affinity, cgroup quota); cap_threads never raises the pool size and never exceeds the quota"""
    import builtins
    from unispeech_amd import hostenv
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            p = tmp_path / "cpu.max"
            p.write_text(fake_open.text)
            return real_open(p, *a, **k)
        return real_open(path, *a, **k)

    monkeypatch.setattr(builtins, "open", fake_open)
    fake_open.text = "1600000 100000\n"
    assert hostenv.cpu_quota() == 16.0
    assert 1 <= hostenv.usable_cpus() <= 16
    fake_open.text = "max 100000\n"
    assert hostenv.cpu_quota() is None
    fake_open.text = "50000 100000\n"          # half a CPU: still one thread
    assert hostenv.usable_cpus() == 1
    before = torch.get_num_threads()
    try:
        assert hostenv.cap_threads(4) == 1
        fake_open.text = "1600000 100000\n"
        torch.set_num_threads(2)
        assert hostenv.cap_threads(4) == 2      # never raised
    finally:
        torch.set_num_threads(before)
