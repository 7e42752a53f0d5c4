"""CPU model of the bf16 MFMA GEMM's data path (unispeech_amd/csrc/gemm_bf16.hip).

There is no GPU in the build container, so the index arithmetic of the kernel -- thread -> global
element maps of both operand loaders, the in-register 4x8 transpose, the XOR-swizzled LDS image and
the v_mfma_f32_32x32x16_bf16 fragment/accumulator lane maps -- is restated here in numpy, formula
by formula, and checked to compute A.B^T for every layout combination.  It also checks the claimed
bank-conflict freedom of the LDS accesses.  The GPU tests (tests/test_gemm_gpu.py) check the
kernel itself.
"""
import numpy as np
import pytest

BK = 64


def lds_off(row, chunk):
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)


def fill_lds_kc(tile, rows):
    """tile: [rows, 64] uint16, K-contiguous loader (load_kc/store_kc)."""
    lds = np.zeros(rows * 128, dtype=np.uint8)
    raw = tile.view(np.uint8).reshape(rows, 128)
    for t in range(256):
        row, ch = t >> 3, t & 7
        for ps in range(rows // 32):
            rr = row + ps * 32
            lds[lds_off(rr, ch): lds_off(rr, ch) + 16] = raw[rr, ch * 16: ch * 16 + 16]
    return lds


def fill_lds_glds(tile, rows):
    """K-contiguous fast path: global_load_lds_dwordx4 writes lane l of wave w at base + 16*l, base =
    (ps*32 + 8*w)*128; the lane fetches the logical chunk (t&7) ^ swizzle(row) (Operand<false>::init/issue)."""
    lds = np.zeros(rows * 128, dtype=np.uint8)
    raw = tile.view(np.uint8).reshape(rows, 128)
    for t in range(256):
        wave, lane = t >> 6, t & 63
        row, pc = t >> 3, t & 7
        for ps in range(rows // 32):
            rr = row + ps * 32
            c = pc ^ ((rr >> 1) & 7)
            dst = (ps * 32 + wave * 8) * 128 + lane * 16
            lds[dst: dst + 16] = raw[rr, c * 16: c * 16 + 16]
    return lds


def test_glds_image_equals_register_path_image():
    rng = np.random.default_rng(0)
    for rows in (128, 64):
        tile = rng.integers(0, 60000, size=(rows, BK)).astype(np.uint16)
        assert np.array_equal(fill_lds_glds(tile, rows), fill_lds_kc(tile, rows))


def fill_lds_ks(tile_t, rows):
    """tile_t: [64 k, rows] uint16 (rows contiguous), K-strided loader (load_ks/store_ks)."""
    lds = np.zeros(rows * 128, dtype=np.uint8)
    for t in range(256):
        kb4 = t & 15
        nb = (t >> 6) * 4 + ((t >> 4) & 3)
        if nb >= rows // 8:
            continue
        # four 16-byte loads: r[i] = 8 consecutive "row" elements of k = kb4*4+i, as 4 dwords
        r = np.zeros((4, 4), dtype=np.uint32)
        for i in range(4):
            e = tile_t[kb4 * 4 + i, nb * 8: nb * 8 + 8].astype(np.uint32)
            r[i] = e[0::2] | (e[1::2] << 16)
        for j in range(8):
            c = j >> 1
            if j & 1 == 0:
                ox = (r[0, c] & 0xFFFF) | ((r[1, c] << 16) & 0xFFFFFFFF)
                oy = (r[2, c] & 0xFFFF) | ((r[3, c] << 16) & 0xFFFFFFFF)
            else:
                ox = (r[0, c] >> 16) | (r[1, c] & 0xFFFF0000)
                oy = (r[2, c] >> 16) | (r[3, c] & 0xFFFF0000)
            row = nb * 8 + j
            off = lds_off(row, kb4 >> 1) + ((kb4 & 1) << 3)
            lds[off: off + 8] = np.array([ox, oy], dtype=np.uint32).view(np.uint8)
    return lds


def read_frag(lds, row0, kk):
    """returns frag[lane, 8] uint16: lane l holds row row0+(l&31), k = 16*kk + 8*(l>>5) .. +7"""
    out = np.zeros((64, 8), dtype=np.uint16)
    for lane in range(64):
        row = row0 + (lane & 31)
        chunk = 2 * kk + (lane >> 5)
        off = lds_off(row, chunk)
        out[lane] = lds[off: off + 16].view(np.uint16)
    return out


def mfma_32x32x16(fa, fb, acc):
    """D[i][j] += sum_k A[i][k] B[k][j]; A lane l: i=l&31, k=8*(l>>5)+e; B lane l: j=l&31, same k.
    acc[lane, reg]: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)"""
    A = np.zeros((32, 16)); B = np.zeros((16, 32))
    for lane in range(64):
        for e in range(8):
            A[lane & 31, 8 * (lane >> 5) + e] = fa[lane, e]
            B[8 * (lane >> 5) + e, lane & 31] = fb[lane, e]
    D = A @ B
    for lane in range(64):
        for reg in range(16):
            acc[lane, reg] += D[(reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), lane & 31]


@pytest.mark.parametrize("ta,tb,bn", [(0, 0, 128), (0, 1, 128), (1, 0, 64), (1, 1, 128), (1, 1, 64)])
def test_tile_data_path(ta, tb, bn):
    rng = np.random.default_rng(ta * 2 + tb)
    BM, BN, WM, WN = 128, bn, 2, 2
    FM, FN = BM // WM // 32, BN // WN // 32
    # small integers are exact as "bf16 bit patterns" stand-ins: we track uint16 payloads
    A = rng.integers(0, 50, size=(BM, BK)).astype(np.uint16)
    B = rng.integers(0, 50, size=(BN, BK)).astype(np.uint16)
    la = fill_lds_ks(np.ascontiguousarray(A.T), BM) if ta else fill_lds_kc(A, BM)
    lb = fill_lds_ks(np.ascontiguousarray(B.T), BN) if tb else fill_lds_kc(B, BN)
    C = np.zeros((BM, BN))
    for wave in range(4):
        wm, wn = wave // WN, wave % WN
        acc = [[np.zeros((64, 16)) for _ in range(FN)] for _ in range(FM)]
        for kk in range(BK // 16):
            fa = [read_frag(la, wm * (BM // WM) + i * 32, kk) for i in range(FM)]
            fb = [read_frag(lb, wn * (BN // WN) + j * 32, kk) for j in range(FN)]
            for i in range(FM):
                for j in range(FN):
                    mfma_32x32x16(fa[i], fb[j], acc[i][j])
        for i in range(FM):
            for j in range(FN):
                for lane in range(64):
                    nn = wn * (BN // WN) + j * 32 + (lane & 31)
                    for r in range(16):
                        mm = wm * (BM // WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
                        C[mm, nn] = acc[i][j][lane, r]
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    np.testing.assert_array_equal(C, ref)


def test_lds_bank_conflicts():
    # ds_read_b128: bank = (addr/4) % 64, lane groups per MI355X_MICROARCH.md LDS table
    groups = [
        [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
        [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
    ]
    groups += [[l + 32 for l in g] for g in groups]
    for kk in range(4):
        for row0 in (0, 32, 64, 96):
            for g in groups:
                slots = set()
                for lane in g:
                    off = lds_off(row0 + (lane & 31), 2 * kk + (lane >> 5))
                    slots.add((off // 16) % 16)
                assert len(slots) == 16, (kk, row0, g)
    # ds_write_b128 of the K-contiguous loader: 8 contiguous lanes per group, 32 banks of 4 B
    for t0 in range(0, 256, 8):
        for ps in range(4):
            slots = {(lds_off((t >> 3) + ps * 32, t & 7) // 16) % 8 for t in range(t0, t0 + 8)}
            assert len(slots) == 8
    # ds_write_b64 of the K-strided loader: 16 contiguous lanes per group
    for t0 in range(0, 256, 16):
        for j in range(8):
            slots = set()
            for t in range(t0, t0 + 16):
                kb4 = t & 15
                nb = (t >> 6) * 4 + ((t >> 4) & 3)
                off = lds_off(nb * 8 + j, kb4 >> 1) + ((kb4 & 1) << 3)
                slots.add((off // 8) % 16)
            assert len(slots) == 16
