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
