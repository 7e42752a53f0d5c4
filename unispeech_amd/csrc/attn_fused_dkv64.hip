// Fused attention backward 2/2, round 6: the dK / dV kernel with 64 KEYS PER WAVE, one wave per SIMD.
//
// Why another shape (profiles/r05/attn_stored_p.txt, pmc_SQ_g.txt).  The 32-keys-per-wave kernel
// (attn_fused_dkv.hip) runs eight waves per CU and is bound by two resources AT ONCE: the CU's LDS pipe (per wave and 64-query
// tile 1152 bytes per lane for 32 MFMAs: operand fragments + per-row scalars + Toeplitz entries = ~4600 LDS cycles per round of
// eight waves) and the VALU (15.7 instructions per score element, ~4000 cycles per SIMD and round) -- against 2048 cycles of
// MFMA.  Round 5 removed VALU work (stored probabilities) and saw no change because the LDS pipe stayed where it was.  This
// kernel removes both:
//   * LDS: a wave owns 64 keys (two 32-key groups g).  Every Q / dO fragment it reads -- the [row][hd] fragments of the score /
//     dP products and the transposed fragments of the dK / dV products -- feeds TWO MFMAs, and the per-row scalars (lse, gate,
//     delta: the lane owns a key, so everything per QUERY row arrives through LDS) are read once for both groups: 1152 bytes
//     per lane for 64 MFMAs instead of 32.  The accumulators (dk, dv: 2 g x 2 f2 x 16 = 128 registers) and the K / V operand
//     fragments (64) live in the AGPR half of a 512-register wave: one wave per SIMD, 4 waves = 256 keys per workgroup.
//   * VALU: the dropout decision costs this layout six instructions per element (row word + column word, two-instruction mix,
//     half select, compare, select: a lane owns ONE key, so it uses one half of every 32-bit word).  The dQ kernel, which runs
//     first and evaluates the same decisions with a lane per query ROW, now leaves them behind as bit words
//     dbits[bh][32-key block][row] (27 MB per Base layer).  Bit k of word (row i, block c) = keep(i, 32 c + k) is exactly the
//     LANE MASK of the select for register "row i" in this kernel (lanes = the 32 keys of the block; the two half-waves hold
//     rows i and i + 4, so the rows of a group of eight are stored in the order 0 4 1 5 2 6 3 7 and a 64-bit mask is one aligned
//     SGPR pair).  The words arrive through the SCALAR cache (s_load_dwordx8 per eight rows and group -- a path these kernels
//     do not use otherwise) and the decision is ONE v_cndmask_b32 with an SGPR-pair condition.
//   * with one wave per SIMD nothing hides a wave's own latencies, so the tile loop is a software pipeline written out in
//     source order and pinned with sched_barrier: per 32-query half f of a tile a PHASE of four chunks (q4 = 8 rows x 2 groups =
//     8 elements per lane); every chunk carries eight MFMAs of OTHER work, one per element --
//         chunk 0: dV / dK products of the previous phase's rows 16..31 (its probabilities were finished by its chunks 2, 3)
//         chunk 1: score + dP products of the NEXT phase, k slices 0, 1       chunk 3: k slices 2, 3
//         chunk 2: dV / dK products of this phase's rows 0..15 (finished by chunks 0, 1)
//     -- fragment reads two MFMAs ahead, per-row scalars / Toeplitz entries / mask words one chunk ahead.
//   * Q / dO tiles arrive by LDS-DMA three tiles ahead through a four-stage ring (a loaded HBM round trip is longer than a
//     tile, profiles/r05); ONE barrier per tile, in the middle of the tile (between its two phases): the next tile has landed
//     (its first score products start in the second phase) and the stage of tile it - 1 is free (its last dV / dK products
//     ran in chunk 0 of this tile's first phase).
//   * the per-row scalars of ALL query tiles are computed once into LDS in the prologue (no vector memory operation besides
//     the DMA inside the loop: vector memory returns in order, any load the compiler has to wait for would drag the DMA along).
// Semantics: modules/multihead_attention.py:259-300 (backward of softmax(scale QK^T + gate * rel + key padding) -> dropout -> PV),
// exactly the arithmetic of attn_bwd_dkv_kernel (same formulas, same rounding points: results are bit-identical).
#include "attn_fused.hpp"

#define FA_K64_STAGES 4
// lab-bench probes (WRONG results by construction; -DWAVLM_EXPERIMENTAL builds only): K64_PROBE bit 0: no element pass (the
// VALU work), bit 1: no MFMAs, bit 2: no mask words (no scalar loads), bit 3: no per-row / Toeplitz reads
#if !defined(WAVLM_EXPERIMENTAL)
#undef K64_PROBE
#endif
#ifndef K64_PROBE
#define K64_PROBE 0
#endif
#if K64_PROBE & 2
#define K64_MFMA(A, B, C) (C)
#else
#define K64_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)
#endif

template <bool DROP>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv64_kernel(FaP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* tabs = reinterpret_cast<float*>(smem + FA_K64_STAGES * 16384);   // [64 zeros | rel table | zeros]
  const int T = p.T, H = p.H;
  const int nq = (T + FA_BQ1 - 1) / FA_BQ1;
  const int Tr = nq * FA_BQ1;
  float* rowl = tabs + p.Ltab + 64;   // [3][Tr]: log2 sc - lse log2 e (-inf past T) | gate log2 e | delta / sc
  int kblk, bh;
  fa_block_map(p.nqb, p.B * H, kblk, bh);
  const int b = bh / H, h = bh % H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int hi = lane >> 5, kl = lane & 31;
  const long D3 = 3L * H * FA_HD, D = (long)H * FA_HD;
  const bf16_t* base = p.qkv + (long)b * T * D3 + h * FA_HD;
  const bf16_t* dobase = p.dO + (long)b * T * D + h * FA_HD;
  const int L = 2 * T - 1;
  const int j0w = kblk * FA_K64 + 64 * wave_u;   // first key of the wave

  // ---- per-key state: K / V operand fragments of both groups, Toeplitz column pointers, key validity
  U4 kf[2][4], vf[2][4];
  const float* tcol[2];
  bool key_ok[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int j = j0w + 32 * g + kl;
    const int jc = j < T ? j : T - 1;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      kf[g][kk].v = *reinterpret_cast<const uint4*>(base + D + (long)jc * D3 + 16 * kk + 8 * hi);
      vf[g][kk].v = *reinterpret_cast<const uint4*>(base + 2 * D + (long)jc * D3 + 16 * kk + 8 * hi);
    }
    key_ok[g] = j < T && !(p.kpm && p.kpm[(long)b * T + jc]);
    tcol[g] = tabs + 64 + (jc + T - 1);   // tcol[-i] = rel[h, j - i]; the 64 zeros in front absorb rows past T
  }
  for (int d = threadIdx.x; d < p.Ltab + 64; d += 256) tabs[d] = (p.tab && d >= 64 && d - 64 < L) ? p.tab[(long)h * L + d - 64] : 0.f;
  for (int ii = threadIdx.x; ii < Tr; ii += 256) {
    const bool ok = ii < T;
    const long o = (long)bh * T + (ok ? ii : T - 1);
    rowl[ii] = ok ? p.log2sc - p.lse[o] * FA_LOG2E : -INFINITY;   // P * sc = 2^(x + this); -inf: rows past T
    rowl[Tr + ii] = p.gate ? p.gate[o] * FA_LOG2E : 0.f;
    rowl[2 * Tr + ii] = p.delta[o] * p.inv_sc;
  }
  // dropout words of the wave's two 32-key blocks: [row] dwords, eight rows = one 32-byte scalar load
  const unsigned* bits[2] = {nullptr, nullptr};
  if constexpr (DROP) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      int c = (j0w >> 5) + g;
      if (c > p.db_nkb - 1) c = p.db_nkb - 1;   // (blocks past the dQ kernel's key tiles hold only keys >= T: discarded)
      bits[g] = p.dbits + ((long)bh * p.db_nkb + c) * p.db_Tq;
    }
  }

  f32x16_t dk[2][2], dv[2][2];   // [g][f2]
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk[g][f2][r] = 0.f; dv[g][f2][r] = 0.f; }
  const unsigned qtr = fa_tr_base(lane);

  FaTileSrc qsrc, dosrc;
  qsrc.init(base, D3, T);
  dosrc.init(dobase, D, T);
  auto qb = [&](int it) __attribute__((always_inline)) { return smem + (it & (FA_K64_STAGES - 1)) * 16384; };
  auto dob = [&](int it) __attribute__((always_inline)) { return smem + (it & (FA_K64_STAGES - 1)) * 16384 + 8192; };
  auto dma_tile = [&](int it) __attribute__((always_inline)) {
    qsrc.issue(it * FA_BQ1, qb(it), wave_u);
    dosrc.issue(it * FA_BQ1, dob(it), wave_u);
  };

  // every load the compiler knows of has landed before the first DMA (which it does not know of) is issued: no `s_waitcnt
  // vmcnt` of its own inside the tile loop (attn_fused_dkv.hip, stored-probability kernel, for the reasoning)
  __builtin_amdgcn_s_waitcnt(0x0f70);
  // the K / V fragments are MFMA operands only: born into AGPRs here, so that the VGPR half of the wave's registers is left to
  // the values the VALU touches (scores / dP: with the fragments in VGPRs the allocator put the MFMA results into AGPRs and
  // paid two v_accvgpr_read per element)
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8_t tk = kf[g][kk].b, tv = vf[g][kk].b;
      asm volatile("" : "+a"(tk));
      asm volatile("" : "+a"(tv));
      kf[g][kk].b = tk; vf[g][kk].b = tv;
    }
  dma_tile(0);
  if (nq > 1) dma_tile(1);
  if (nq > 2) dma_tile(2);
  __builtin_amdgcn_sched_barrier(0);
  if (nq > 2) fa_tile_sync<8>(); else if (nq > 1) fa_tile_sync<4>(); else fa_tile_sync<0>();

  // ---- pipeline state
  f32x16_t s[2][2], dp[2][2];    // [f][g]: scores / dP of the 32 x 32 block, alive from its products to its element pass
  U4 pf[2][2], dsf[2][2];        // [g][s2]: kept probabilities / dS of rows 16 s2 .. +15 of the current half, bf16 B operands
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) { pf[g][s2].v = make_uint4(0, 0, 0, 0); dsf[g][s2].v = make_uint4(0, 0, 0, 0); }

  // score + dP products of half F of the tile in `qt` / `dot`, k slice kk, both groups: two fragment reads, four MFMAs
  using F0 = std::integral_constant<int, 0>;
  using F1 = std::integral_constant<int, 1>;
  // prologue: the products of (tile 0, half 0), nothing to overlap with
  {
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[0][g][r] = 0.f; dp[0][g][r] = 0.f; }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const bf16x8_t aq = frag_plain(qb(0), kl, kk, hi), ad = frag_plain(dob(0), kl, kk, hi);
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        s[0][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, kf[g][kk].b, s[0][g], 0, 0, 0);
        dp[0][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad, vf[g][kk].b, dp[0][g], 0, 0, 0);
      }
    }
  }

  // One phase = the element pass of half F of tile `it` with 32 MFMAs of other work in its shadow:
  //   tail : dV / dK products of rows 16..31 of the PREVIOUS half (tile `it_tail`, half F ^ 1) -- pf / dsf [.][1]
  //   sm   : score / dP products of the NEXT half (tile `it_sm`, half F ^ 1) -> s / dp [F ^ 1]
  //   own  : dV / dK products of rows 0..15 of this half -- pf / dsf [.][0]
  struct Chunk { float4 lse, gat, del; float tq[2][4]; u32x8_t m[2]; };
  // per-chunk inputs, fetched one chunk ahead: three 16-byte row vectors, 2 x 4 Toeplitz entries, 2 x 8 mask words
  auto fetch = [&](int it_, int F_, int q4) __attribute__((always_inline)) {
    Chunk c;
#if K64_PROBE & 8
    c.lse = make_float4(-1.f, -1.f, -1.f, -1.f); c.gat = make_float4(0.f, 0.f, 0.f, 0.f); c.del = make_float4(0.1f, 0.1f, 0.1f, 0.1f);
    for (int g = 0; g < 2; ++g) for (int e = 0; e < 4; ++e) c.tq[g][e] = 0.f;
    if constexpr (DROP) { for (int g = 0; g < 2; ++g) c.m[g] = *(k64_mask_ptr)(bits[g] + (it_ * FA_BQ1 + 32 * F_ + 8 * q4)); }
    return c;
#endif
    const int iq0 = it_ * FA_BQ1;
    const int il0 = 32 * F_ + 8 * q4 + 4 * hi;
    const float* rv = rowl + iq0 + il0;
    c.lse = *reinterpret_cast<const float4*>(rv);
    c.gat = *reinterpret_cast<const float4*>(rv + Tr);
    c.del = *reinterpret_cast<const float4*>(rv + 2 * Tr);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const float* tq = tcol[g] - (iq0 + il0);
#pragma unroll
      for (int e = 0; e < 4; ++e) c.tq[g][e] = tq[-e];
#if K64_PROBE & 4
      if constexpr (DROP) { for (int k = 0; k < 8; ++k) c.m[g][k] = 0xffff0f0fu; }
#else
      if constexpr (DROP) c.m[g] = *(k64_mask_ptr)(bits[g] + (iq0 + 32 * F_ + 8 * q4));
#endif
    }
    return c;
  };
  Chunk cfirst = fetch(0, 0, 0);   // chunk 0 of the next phase travels across the phase boundary
  // ... and so do the fragments of its eight MFMAs.  Before the first tile pf / dsf are zero: any finite fragments do (tile 0's)
  bf16x8_t frfirst[4];
#pragma unroll
  for (int f2 = 0; f2 < 2; ++f2) { frfirst[f2] = frag_tr(dob(0), qtr, f2, 1, 1); frfirst[2 + f2] = frag_tr(qb(0), qtr, f2, 1, 1); }
  // it_next: the tile of the NEXT phase (past the end: the last tile again, inputs unused)
  auto phase = [&](auto fc, int it, int it_tail, int it_sm, int it_next) __attribute__((always_inline)) {
    constexpr int F = decltype(fc)::value, G = F ^ 1;
    (void)it_tail;   // (its fragments travel in frfirst)
    const unsigned char* q_sm = qb(it_sm); const unsigned char* do_sm = dob(it_sm);
    const unsigned char* q_own = qb(it); const unsigned char* do_own = dob(it);
    // the eight MFMAs a chunk carries (index m = 0..7), with their fragment reads
    auto mfma_tail = [&](int m, const bf16x8_t (&fr)[4]) __attribute__((always_inline)) {
      // m: 0..3 dV (f2 = m >> 1, g = m & 1), 4..7 dK; fr[0..1] = dO^T fragments f2 = 0, 1; fr[2..3] = Q^T
      const int g = m & 1, f2 = (m >> 1) & 1;
      if (m < 4) dv[g][f2] = K64_MFMA(fr[f2], pf[g][1].b, dv[g][f2]);
      else dk[g][f2] = K64_MFMA(fr[2 + f2], dsf[g][1].b, dk[g][f2]);
    };
    auto mfma_own = [&](int m, const bf16x8_t (&fr)[4]) __attribute__((always_inline)) {
      const int g = m & 1, f2 = (m >> 1) & 1;
      if (m < 4) dv[g][f2] = K64_MFMA(fr[f2], pf[g][0].b, dv[g][f2]);
      else dk[g][f2] = K64_MFMA(fr[2 + f2], dsf[g][0].b, dk[g][f2]);
    };
    auto mfma_sm = [&](int m, int kk0, const bf16x8_t (&fr)[4]) __attribute__((always_inline)) {
      // m: (kk = kk0 + (m >> 2)), within a k slice: S g0, S g1, dP g0, dP g1; fr[2 (m >> 2)] = Q fragment, fr[2 (m >> 2) + 1] = dO
      const int kk = kk0 + (m >> 2), g = m & 1, which = (m >> 1) & 1;
      if (which == 0) s[G][g] = K64_MFMA(fr[2 * (m >> 2)], kf[g][kk].b, s[G][g]);
      else dp[G][g] = K64_MFMA(fr[2 * (m >> 2) + 1], vf[g][kk].b, dp[G][g]);
    };
    // element (g, e) of chunk q4
    float pv[2][4], dsv[2][4];
    auto element = [&](const Chunk& c, int q4, int g, int e) __attribute__((always_inline)) {
      const int rr = 4 * q4 + e;
#if K64_PROBE & 1
      if (e == 3) {   // (keeps the accumulators' inputs alive without the element arithmetic)
        pf[g][q4 >> 1].u[2 * (q4 & 1)] = __float_as_uint(s[F][g][rr]); pf[g][q4 >> 1].u[2 * (q4 & 1) + 1] = __float_as_uint(c.lse.x);
        dsf[g][q4 >> 1].u[2 * (q4 & 1)] = __float_as_uint(dp[F][g][rr]); dsf[g][q4 >> 1].u[2 * (q4 & 1) + 1] = __float_as_uint(c.tq[g][e]) ^ c.m[g][e];
      }
      return;
#endif
      const float lsev[4] = {c.lse.x, c.lse.y, c.lse.z, c.lse.w};
      const float gatv[4] = {c.gat.x, c.gat.y, c.gat.z, c.gat.w};
      const float delv[4] = {c.del.x, c.del.y, c.del.z, c.del.w};
      const float pe = __builtin_amdgcn_exp2f(fmaf(s[F][g][rr], p.sc2, fmaf(gatv[e], c.tq[g][e], lsev[e])));   // rows past T: -inf -> 0
      float pd = pe;
      if constexpr (DROP) pd = k64_keep(pe, c.m[g][2 * e], c.m[g][2 * e + 1]);
      pv[g][e] = pd;
      dsv[g][e] = fmaf(pd, dp[F][g][rr], -(pe * delv[e]));
      if (e == 3) {
        pf[g][q4 >> 1].u[2 * (q4 & 1)] = pack_bf16(pv[g][0], pv[g][1]);
        pf[g][q4 >> 1].u[2 * (q4 & 1) + 1] = pack_bf16(pv[g][2], pv[g][3]);
        dsf[g][q4 >> 1].u[2 * (q4 & 1)] = pack_bf16(dsv[g][0], dsv[g][1]);
        dsf[g][q4 >> 1].u[2 * (q4 & 1) + 1] = pack_bf16(dsv[g][2], dsv[g][3]);
      }
    };

    // Every chunk starts by issuing the reads of the NEXT chunk -- its per-row inputs and the fragments of the eight MFMAs it
    // carries -- so that a full chunk of work (~350 cycles) covers every LDS / scalar-cache round trip; the first chunk's
    // inputs (cfirst) and fragments (frfirst: this wave's dV / dK fragments of rows 16..31 of the previous half) were issued by
    // the previous phase's last chunk.
    Chunk c0 = cfirst;
    bf16x8_t fr0[4], fr1[4], fr2[4], fr3[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) fr0[k] = frfirst[k];
    // ---- chunk 0 || tail products (previous half, rows 16..31: s2 = 1)
    Chunk c1 = fetch(it, F, 1);
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) { fr1[2 * k2] = frag_plain(q_sm, 32 * G + kl, k2, hi); fr1[2 * k2 + 1] = frag_plain(do_sm, 32 * G + kl, k2, hi); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      mfma_tail(m, fr0);
      element(c0, 0, m >> 2, m & 3);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- chunk 1 || score / dP products of the next half, k slices 0, 1  (s / dp [G] were consumed a phase ago)
    Chunk c2 = fetch(it, F, 2);
#pragma unroll
    for (int f2 = 0; f2 < 2; ++f2) { fr2[f2] = frag_tr(do_own, qtr, f2, F, 0); fr2[2 + f2] = frag_tr(q_own, qtr, f2, F, 0); }
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[G][g][r] = 0.f; dp[G][g][r] = 0.f; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      mfma_sm(m, 0, fr1);
      element(c1, 1, m >> 2, m & 3);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- chunk 2 || own products of rows 0..15 (pf / dsf [.][0] complete after chunk 1)
    Chunk c3 = fetch(it, F, 3);
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) { fr3[2 * k2] = frag_plain(q_sm, 32 * G + kl, 2 + k2, hi); fr3[2 * k2 + 1] = frag_plain(do_sm, 32 * G + kl, 2 + k2, hi); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      mfma_own(m, fr2);
      element(c2, 2, m >> 2, m & 3);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- chunk 3 || score / dP products of the next half, k slices 2, 3
    cfirst = fetch(it_next, G, 0);
#pragma unroll
    for (int f2 = 0; f2 < 2; ++f2) { frfirst[f2] = frag_tr(do_own, qtr, f2, F, 1); frfirst[2 + f2] = frag_tr(q_own, qtr, f2, F, 1); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      mfma_sm(m, 2, fr3);
      element(c3, 3, m >> 2, m & 3);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  for (int it = 0; it < nq; ++it) {
    // first half: tail = (it - 1, half 1) -- before the first tile pf / dsf are zero and the fragments come from tile 0
    // (finite) --, next = (it, half 1)
    phase(F0{}, it, it > 0 ? it - 1 : 0, it, it);
    // middle of the tile: tile it + 1 has landed (its first products run in the second half), the stage of tile it - 1 is free
    if (it + 2 < nq) fa_tile_sync<4>(); else fa_tile_sync<0>();
    if (it + 3 < nq) dma_tile(it + 3);
    __builtin_amdgcn_sched_barrier(0);
    // second half: tail = (it, half 0); next = (it + 1, half 0) -- past the end: tile nq - 1 again, results unused
    phase(F1{}, it, it, it + 1 < nq ? it + 1 : it, it + 1 < nq ? it + 1 : it);
  }
  // epilogue: rows 16..31 of the last half (fragments: frfirst)
#pragma unroll
  for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      dv[g][f2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frfirst[f2], pf[g][1].b, dv[g][f2], 0, 0, 0);
      dk[g][f2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frfirst[2 + f2], dsf[g][1].b, dk[g][f2], 0, 0, 0);
    }
#pragma unroll
  for (int g = 0; g < 2; ++g)
    if (!key_ok[g]) {   // padded keys and keys past T
#pragma unroll
      for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[g][f2][r] = 0.f; dv[g][f2][r] = 0.f; }
    }
  __syncthreads();   // (the column-sum scratch below overlays the tile ring)
  if (p.dbias_part) {
    // one partial row per 32-key group: row (32-key group index) of the [B][nrow] x [3 H 64] partial matrix whose q part the dQ
    // kernel writes (nrow = 4 per 128-row block there).  Groups past nrow hold only keys >= T (zero sums): not written
    const int nrow = 4 * ((T + FA_BQ - 1) / FA_BQ);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int grp = kblk * 8 + 2 * wave_u + g;
      if (grp < nrow) {
        float* drow = p.dbias_part + ((long)b * nrow + grp) * D3 + h * FA_HD;
        fa_wave_colsum(dk[g], p.scale, reinterpret_cast<float*>(smem), drow + D, lane, wave_u);
        fa_wave_colsum(dv[g], 1.f, reinterpret_cast<float*>(smem), drow + 2 * D, lane, wave_u);
      }
    }
  }
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int j = j0w + 32 * g + kl;
    if (j < T) {
      bf16_t* dst = p.dqkv + ((long)b * T + j) * D3 + h * FA_HD;
#pragma unroll
      for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          uint2 w;
          w.x = pack_bf16(dk[g][f2][4 * q4] * p.scale, dk[g][f2][4 * q4 + 1] * p.scale);
          w.y = pack_bf16(dk[g][f2][4 * q4 + 2] * p.scale, dk[g][f2][4 * q4 + 3] * p.scale);
          *reinterpret_cast<uint2*>(dst + D + 32 * f2 + 8 * q4 + 4 * hi) = w;
          w.x = pack_bf16(dv[g][f2][4 * q4], dv[g][f2][4 * q4 + 1]);
          w.y = pack_bf16(dv[g][f2][4 * q4 + 2], dv[g][f2][4 * q4 + 3]);
          *reinterpret_cast<uint2*>(dst + 2 * D + 32 * f2 + 8 * q4 + 4 * hi) = w;
        }
    }
  }
}

size_t fa_dkv64_smem(const FaP& p) {
  const int nq = (p.T + FA_BQ1 - 1) / FA_BQ1;
  size_t b = (size_t)FA_K64_STAGES * 16384 + (size_t)(p.Ltab + 64 + 3 * nq * FA_BQ1) * sizeof(float);
  if (b < FA_CS_FLOATS * sizeof(float)) b = FA_CS_FLOATS * sizeof(float);
  return b;
}

// grid: ceil(T / 256) key blocks per (b, h); p.nqb must be that count.  Returns WL_EINVAL when the per-row scalars of all query
// tiles do not fit the LDS (T > ~6000: the caller then runs the 32-keys-per-wave kernel).
int fa_launch_dkv64(const FaP& p, unsigned grid, hipStream_t st) {
  const size_t smem = fa_dkv64_smem(p);
  if (smem > 160 * 1024) return WL_EINVAL;
  if (p.th) {
    if (!p.dbits) return WL_EINVAL;
    if (fa_set_smem(attn_bwd_dkv64_kernel<true>, smem) != WL_OK) return WL_ELAUNCH;
    WL_LAUNCH(attn_bwd_dkv64_kernel<true>, dim3(grid), dim3(256), smem, st, p);
  } else {
    if (fa_set_smem(attn_bwd_dkv64_kernel<false>, smem) != WL_OK) return WL_ELAUNCH;
    WL_LAUNCH(attn_bwd_dkv64_kernel<false>, dim3(grid), dim3(256), smem, st, p);
  }
  return wl_check_launch();
}
