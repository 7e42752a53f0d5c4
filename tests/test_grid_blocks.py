"""Host logic around the CU reservation of data-parallel runs (unispeech_amd/dp.py, ops.grid_blocks): every split-K choice
aims at ONE round of the persistent grid that is actually launched (256 CUs minus the reserved ones), and the number of RCCL
channels the launcher allows equals the reservation.  No GPU: the reservation is a host-side setting of the library."""
import pytest

from unispeech_amd import dp, ops


@pytest.fixture
def reserved():
    before = ops.get_reserved_cus()
    yield
    ops.set_reserved_cus(before)


def test_splits_follow_the_reserved_grid(reserved):
    ops.set_reserved_cus(0)
    assert ops.grid_blocks() == 256
    assert ops.pick_split(512, 1536, 32 * 375) == 21            # conv-stack weight gradient: 12 tiles x 21 = 252 of 256
    assert ops.grouped_split(108, 375) == 2                     # Base block: 216 of 256
    assert ops.grouped_split(64, 500) == 4                      # Large: out_proj + q|k|v: 256 of 256
    ops.set_reserved_cus(6)
    assert ops.grid_blocks() == 250
    s = ops.pick_split(512, 1536, 32 * 375)
    assert s == 20 and 12 * s <= 250                            # 240 of 250: one round (21 would run a second round of two blocks)
    assert ops.grouped_split(108, 375) == 2
    assert ops.grouped_split(64, 500) == 3                      # 192 of 250: one round, not 256 work items on 250 blocks
    ops.set_reserved_cus(8)
    assert 12 * ops.pick_split(512, 1536, 32 * 375) <= 248


def test_reserved_channels_env(monkeypatch):
    monkeypatch.delenv("WAVLM_DP_RESERVED_CUS", raising=False)
    assert dp.reserved_channels() == 6
    monkeypatch.setenv("WAVLM_DP_RESERVED_CUS", "0")
    assert dp.reserved_channels() == 0
    monkeypatch.setenv("WAVLM_DP_RESERVED_CUS", "200")
    assert dp.reserved_channels() == 64


def test_rccl_channel_cap_only_for_data_parallel_runs_and_never_over_the_users_value(monkeypatch, capsys):
    """NCCL_MAX_NCHANNELS is process-wide, so it is set only when the run is data-parallel, never over an exported
    value, and the plugin says what it did"""
    import os
    from unispeech_amd import dp
    monkeypatch.delenv("NCCL_MAX_NCHANNELS", raising=False)
    monkeypatch.delenv("WAVLM_DP_RESERVED_CUS", raising=False)
    monkeypatch.setenv("WORLD_SIZE", "1")
    assert dp.cap_rccl_channels(log=True) is None and "NCCL_MAX_NCHANNELS" not in os.environ
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert dp.cap_rccl_channels(log=True) == "6" and os.environ["NCCL_MAX_NCHANNELS"] == "6"
    assert "NCCL_MAX_NCHANNELS=6" in capsys.readouterr().err
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "32")
    assert dp.cap_rccl_channels(world=8) == "32" and os.environ["NCCL_MAX_NCHANNELS"] == "32"
    monkeypatch.delenv("NCCL_MAX_NCHANNELS")
    monkeypatch.setenv("WAVLM_DP_RESERVED_CUS", "0")
    assert dp.cap_rccl_channels(world=8) is None and "NCCL_MAX_NCHANNELS" not in os.environ


def test_padding_mask_word_path_accepts_any_nonzero_byte():
    """forward_padding_mask's 64-bit fast path must not assume that a True byte is exactly 0x01"""
    import numpy as np
    import torch
    from conftest import TINY
    from unispeech_amd.wavlm import WavLM, WavLMConfig
    m = WavLM(WavLMConfig(dict(TINY)))
    B, n_frames, k = 3, 25, 320
    raw = np.zeros((B, n_frames * k), dtype=np.uint8)      # T = 8000, k = 320: the 64-bit word path
    raw[1, 10 * k:] = 1
    raw[2, 17 * k + 5:] = 0xFE          # non-canonical "true" bytes, frame 17 only partly padded
    canon = torch.from_numpy(raw != 0)
    odd = torch.from_numpy(raw.view(np.bool_))
    want = canon[:, :n_frames * k].view(B, n_frames, k).all(-1)
    assert torch.equal(m.forward_padding_mask(n_frames, canon), want)
    assert torch.equal(m.forward_padding_mask(n_frames, odd), want)
    assert want[1, 10:].all() and not want[1, :10].any() and not want[2, 17] and want[2, 18:].all()
