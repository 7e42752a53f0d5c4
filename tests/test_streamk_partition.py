"""Host logic of the balanced grouped weight-gradient launch (csrc/gemm_common.hpp: gemm_sk_plan, decoded by gemm_w4_kernel),
restated in Python: every K step of every tile is computed exactly once, every tile has exactly s + 1 partial sums, main and
tail workgroups finish within a few K steps of each other, and the persistent loop's bound (G * q work ids) reaches every
tile a tail workgroup owns."""
import pytest

from unispeech_amd import ops

SEG_COST = 8


def plan(T, KS, G):
    if T <= 0 or T >= G:
        return None
    s, R = G // T, G - T * (G // T)
    if R == 0 or s < 1:
        return None
    q = (T + R - 1) // R
    L0 = (q * (KS + SEG_COST) - SEG_COST) // (1 + s * q)
    Lm = best = 0
    for L in (L0, L0 + 1):
        if L < 8 or KS - L * s < 1:
            continue
        cost = max(L + SEG_COST, q * (KS - L * s + SEG_COST))
        if Lm == 0 or cost < best:
            Lm, best = L, cost
    if Lm == 0 or best * 100 > 97 * ((KS + s - 1) // s + SEG_COST):
        return None
    return s, R, q, Lm


@pytest.mark.parametrize("T,KS,G", [(108, 375, 256), (108, 375, 248), (108, 40, 256), (32, 65, 255), (192, 500, 248),
                                    (64, 500, 248), (255, 90, 256), (13, 2000, 256), (3, 3000, 256), (107, 79, 193), (27, 375, 256)])
def test_partition_covers_every_k_step_once(T, KS, G):
    pl = plan(T, KS, G)
    if (T, KS, G) in [(108, 375, 256), (108, 375, 248), (64, 500, 248), (192, 500, 248)]:   # Base; Base / Large with reserved CUs
        assert pl is not None
    if pl is None:
        return
    s, R, q, Lm = pl
    cover = {}
    work = [0] * G
    for vid in range(G * q):
        item, seg = vid % G, vid // G       # (the XCD permutation of the physical id is a bijection: left out)
        if item < s * T:
            if seg > 0:
                continue
            split, tile = item // T, item % T
            t0, t1 = split * Lm, split * Lm + Lm
        else:
            e = item - s * T
            first, last = e * T // R, (e + 1) * T // R
            tile = first + seg
            if tile >= last:
                continue
            split, t0, t1 = s, s * Lm, KS
        assert 0 <= tile < T and 0 <= t0 < t1 <= KS
        for k in range(t0, t1):
            assert (tile, k) not in cover
            cover[(tile, k)] = split
        work[item] += t1 - t0 + SEG_COST
    assert len(cover) == T * KS
    for t in range(T):
        assert {cover[(t, k)] for k in range(KS)} == set(range(s + 1))
    assert max(work) <= 0.97 * ((KS + s - 1) // s + SEG_COST)     # a plan exists only where it beats the one-round split


def test_exact_fit_has_no_plan():
    assert plan(64, 500, 256) is None                   # Large on one GPU: 64 tiles x 4 fills the grid


def test_grouped_slabs(monkeypatch):
    monkeypatch.delenv("WAVLM_WGRAD_SPLIT", raising=False)
    monkeypatch.setenv("WAVLM_WGRAD_STREAMK", "1")
    monkeypatch.setattr(ops, "lab_build", lambda: False)
    assert ops.grouped_slabs(108, 375, 256) == 2     # the product library has no balanced launch: the switch alone does nothing
    # pointing WAVLM_HIP_LIB at a PRODUCT build elsewhere must not change that (the capability is asked of the loaded
    # library -- the lab build alone exports `wavlm_lab_build` --, not inferred from the variable)
    monkeypatch.setenv("WAVLM_HIP_LIB", "/somewhere/else/libwavlm_hip.so")
    assert ops.grouped_slabs(108, 375, 256) == 2
    monkeypatch.delenv("WAVLM_HIP_LIB")
    monkeypatch.setattr(ops, "lab_build", lambda: True)   # (the lab library, round 5)
    assert ops.grouped_slabs(108, 375, 256) == 3     # Base: all four weight gradients of a block: 2 main splits + the tail
    assert ops.grouped_slabs(64, 500, 256) == 5      # Large: one more than the one-round split (used once CUs are reserved)
    assert ops.grouped_slabs(9, 16, 256) == 2        # too little work to balance: the one-round split
    monkeypatch.delenv("WAVLM_WGRAD_STREAMK")
    assert ops.grouped_slabs(108, 375, 256) == 2     # default: the one-round split
    assert ops.grouped_split(12, 1000, 250) == 20    # conv-stack weight gradient with 6 CUs left to RCCL: 240 of 250 blocks, one round
