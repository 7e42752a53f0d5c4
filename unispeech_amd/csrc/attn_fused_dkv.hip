// Fused attention backward 2/2: the dK / dV kernel (design notes and shared pieces: attn_fused.hpp).
// A translation unit of its own: built with -fno-slp-vectorize (build.py).  The SLP vectoriser turns the softmax
// recomputation into v_pk_*_f32, which issue through the same pipe as the MFMAs they are meant to overlap with; scalar
// f32 VALU measured 228 -> 202 us per launch here (the forward is indifferent, the dQ kernel 4 % slower without it).
#include "attn_fused.hpp"

// FA_DKV_BIAS_IN_C: bias - lse as the C operand of the score MFMAs (as in the forward, attn_fused.hip).  No instruction less
// here (the two fmas become fma + multiply) and the bias reads move in front of the MFMAs: measured 196 -> 211 us
// (profiles/r04/ab_attn_bwd_bias_in_c.txt): OFF.
#ifndef FA_DKV_BIAS_IN_C
#define FA_DKV_BIAS_IN_C 0
#endif

// ------------------------------------------------------------------------------------ backward 2/2: dK, dV
// Lane owns a KEY column; scores are in the untransposed layout S[q][kv] so that the query contraction of
// dV^T = dO^T P and dK^T = Q^T dS finds its k-slots in the lane's registers.
// BITS (round 6, with DROP; WAVLM_ATTN_DBITS=1 or the forward's words, WAVLM_ATTN_STORE_P=bits): the keep decisions come from bit
// words (attn_fused.hpp: dbits) instead of being recomputed.  This kernel gains 10-13 us per launch from it (200 -> 187-190), which
// the producer of the words pays back (dQ kernel +10, forward +11.5: profiles/r06): off by default.  This layout pays six instructions per element for a decision (row word + column word, two-instruction mix, half
// select, compare, select: a lane owns ONE key, so it uses one half of every 32-bit word) and a 16-byte LDS read per four rows
// for the row words; bit k of word (row i, 32-key block c) is exactly the LANE MASK of the select for register "row i" here
// (lanes = the 32 keys of the wave).  Two ways to use it were built:
//   * through the scalar cache (s_load_dwordx8 per eight rows, the decision ONE v_cndmask_b32 with an SGPR-pair condition: one
//     instruction per element instead of six).  Measured SLOWER, 214-231 us against 200 (profiles/r06/attn_dbits.txt): eight
//     waves per CU streaming 32-byte pieces miss the small shared scalar cache, a scalar load's first use waits for
//     lgkmcnt(0) -- scalar loads return out of order --, which also drains every LDS read in flight, and requests cannot be
//     issued far ahead for the same reason.  (That form lives on in attn_fused_dkv64.hip.)
//   * through the per-row LDS arrays the kernel has anyway (this form): the words of the wave's key block take the place of
//     the dropout row words (one global load per lane and tile next to lse / delta / gate, the same 16-byte LDS read per four
//     rows), and a decision is v_bfe_i32 (bit `key` of the row's word, sign-extended) + v_and_b32: two instructions.
#define FA_ROWV 448   // floats per stage of the per-row arrays: lse | delta | gate | (row words, or the four waves' bit words)
// FA_DKV_RSM (round 6): the per-row scalars through the MATRIX pipe instead of 16-byte broadcast reads.  The lane owns a key, so
// lse / gate / delta of the query rows reach its registers through LDS: three 16-byte reads per four rows, 384 of the kernel's
// ~1100 bytes of LDS traffic per lane and tile in a kernel that is bound by LDS passes.  An MFMA whose B operand is
// all ones in k slots 0..2 and whose A operand holds a row's scalar as three bf16 pieces (x = h + m + l, 24 significand bits) in
// those slots delivers D[row][key] = x(row) for every key -- the scalar broadcast along the keys, in exactly the register
// layout of the scores.  lse rides as a FIFTH k step of the score product itself (in units of 1 / sc2), gate and delta are one
// product each per 32-row block: +6 MFMAs per tile (the pipe is 35 % busy) for 6 fragment reads instead of 24 broadcast reads.
// Parity green (gpu_checks attention / dropout_exact), 240 VGPRs, no spill -- and measured 3-5 us SLOWER (210 against 206 us, without
// dropout 169 against 163; three alternating same-box runs, profiles/r06/ab_attn_dkv_rsm.txt): a quarter of the kernel's LDS bytes
// gone and six MFMAs added change nothing, i.e. the kernel is not bound by LDS bytes alone either.  Lab switch, OFF.
#if !defined(WAVLM_EXPERIMENTAL)
#undef FA_DKV_RSM
#endif
#ifndef FA_DKV_RSM
#define FA_DKV_RSM 0
#endif
#define FA_RSM_STAGE 4096   // bytes per stage: [3 scalars][64 rows][8 bf16] fragments, then 1 KiB of row words / bit words
template <bool DROP, bool BITS = false>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(FaP p) {
  static_assert(DROP || !BITS, "bit words are dropout decisions");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // per stage: Q [q][hd] | dO [q][hd], 8 KB each; the transposed operands of the query contraction are read from
  // the same tiles with ds_read_b64_tr_b16 (frag_tr)
  auto qbuf = [&](int st) { return smem + st * 16384; };
  auto dobuf = [&](int st) { return smem + st * 16384 + 8192; };
  float* tabs = reinterpret_cast<float*>(smem + 32768);
  float* rowv = tabs + p.Ltab + 64;  // [2 stages][FA_ROWV]: lse * log2e, delta, gate * log2e, dropout row word (BITS: the four waves' bit words) of the query tile
  const int T = p.T, H = p.H;
  int kblk, bh;
  fa_block_map(p.nqb, p.B * H, kblk, bh);
  const int b = bh / H, h = bh % H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int hi = lane >> 5, kl = lane & 31;
  const int j = kblk * FA_BK1 + 32 * wave + kl;
  const int jc = j < T ? j : T - 1;
  const long D3 = 3L * H * FA_HD, D = (long)H * FA_HD;
  const bf16_t* base = p.qkv + (long)b * T * D3 + h * FA_HD;
  const bf16_t* dobase = p.dO + (long)b * T * D + h * FA_HD;
  const int L = 2 * T - 1;

  U4 kf[4], vf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    kf[kk].v = *reinterpret_cast<const uint4*>(base + D + (long)jc * D3 + 16 * kk + 8 * hi);
    vf[kk].v = *reinterpret_cast<const uint4*>(base + 2 * D + (long)jc * D3 + 16 * kk + 8 * hi);
  }
  for (int d = threadIdx.x; d < p.Ltab + 64; d += 256) tabs[d] = (p.tab && d >= 64 && d - 64 < L) ? p.tab[(long)h * L + d - 64] : 0.f;
  const bool key_ok = j < T && !(p.kpm && p.kpm[(long)b * T + jc]);
  const unsigned cw = fa_col_word(p.s1, (unsigned)(jc >> 1));
  const unsigned csh = (jc & 1) << 4;
  const float* tcol = tabs + 64 + (jc + T - 1);  // tcol[-i] = rel[h, j - i]; 64 zero floats in front absorb rows past T
  // BITS: where this lane's key sits in a word: bit `key` in the dQ kernel's words, fa_fbit_of_key in the forward's
  const unsigned kbit = (BITS && p.db_fwd) ? (unsigned)fa_fbit_of_key(kl) : (unsigned)kl;
  const unsigned* bits = nullptr;   // BITS: the words of this wave's 32-key block, [row] (rows of a group of eight: 0 4 1 5 2 6 3 7)
  if constexpr (BITS) {
    int c = kblk * 4 + wave_u;
    if (c > p.db_nkb - 1) c = p.db_nkb - 1;   // (blocks past the dQ kernel's key tiles hold only keys >= T: discarded)
    bits = p.dbits + ((long)bh * p.db_nkb + c) * p.db_Tq;
  }
  // a padded / out-of-range key is NOT masked inside the loop (one add per element): its column only feeds this lane's
  // own dK / dV rows, which are written as zeros at the end

  f32x16_t dk[2], dv[2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[f][r] = 0.f; dv[f][r] = 0.f; }
  const int nq = (T + FA_BQ1 - 1) / FA_BQ1;
  const unsigned qtr = fa_tr_base(lane);

  // per-row scalars of a query tile: fetched into registers BEFORE the tile's DMA is issued and written to LDS at the END of
  // the iteration (rows_store) -- consuming a global load in the middle of the iteration makes the compiler wait for it,
  // and vector memory operations return in order: that wait would also wait for the prefetch DMA issued behind it
  // (raw values only in rows_load: any arithmetic on them would put the wait right behind the loads)
  struct RowRegs { float lse, gate, delta; unsigned o; bool ok; unsigned word; };
  auto rows_load = [&](int it) {
    // every wave loads the tile's 64 rows (wave 0 stores them): no divergent branch around the loads, i.e. no merge of
    // "loaded" and "not loaded" values that would have to wait for the loads
    RowRegs r;
    const int ii = it * FA_BQ1 + lane;
    r.ok = ii < T;
    const long o = (long)bh * T + (r.ok ? ii : T - 1);
    r.o = (unsigned)o;
    r.lse = p.lse[o];
    r.delta = p.delta[o];
    r.gate = (p.gate ? p.gate : p.lse)[o];   // (no gate: any readable address; rows_store writes 0 then)
    r.word = 0;
    if constexpr (BITS) r.word = bits[fa_bitrow(ii)];   // this WAVE's key block, row ii (rows < db_Tq: written by the dQ kernel)
    return r;
  };
  auto rows_store = [&](const RowRegs& r, int st) {
    const int t = threadIdx.x;
#if FA_DKV_RSM
    static_assert(!FA_DKV_BIAS_IN_C, "the matrix-pipe form of the row scalars replaces the C-operand form");
    unsigned char* rs = reinterpret_cast<unsigned char*>(rowv) + st * FA_RSM_STAGE;
    if (t < 64) {
      // x = h + m + l in bf16 pieces (round to nearest each: 24 significand bits together); rows past T: a finite -1e30 (the
      // exponent becomes -inf-like: 2^x = 0) -- an infinity would make m = inf - inf
      auto split3 = [](float x, unsigned char* dst) __attribute__((always_inline)) {
        const bf16_t h = f2bf(x); const float r1 = x - bf2f(h);
        const bf16_t m = f2bf(r1); const float r2 = r1 - bf2f(m);
        const bf16_t l = f2bf(r2);
        *reinterpret_cast<uint4*>(dst) = make_uint4((unsigned)h | ((unsigned)m << 16), (unsigned)l, 0u, 0u);
      };
      split3((r.ok ? p.log2sc - r.lse * FA_LOG2E : -1e30f) / p.sc2, rs + t * 16);            // rides in the score product: units of 1 / sc2
      split3(p.gate ? r.gate * FA_LOG2E : 0.f, rs + 1024 + t * 16);
      split3(r.delta * p.inv_sc, rs + 2048 + t * 16);
      if constexpr (DROP && !BITS) reinterpret_cast<unsigned*>(rs + 3072)[t] = fa_row_word(p.s0, r.o);
    }
    if constexpr (BITS) reinterpret_cast<unsigned*>(rs + 3072)[t] = r.word;   // every wave its own 64 words
    return;
#endif
    if (t < 64) {
#if FA_DKV_BIAS_IN_C
      // bias - lse enters as the C operand of the score MFMAs, in units of 1 / sc2 (P * sc = 2^(sc2 s))
      rowv[st * FA_ROWV + t] = r.ok ? (p.log2sc - r.lse * FA_LOG2E) / p.sc2 : -INFINITY;
      rowv[st * FA_ROWV + 128 + t] = p.gate ? r.gate * FA_LOG2E / p.sc2 : 0.f;
#else
      rowv[st * FA_ROWV + t] = r.ok ? p.log2sc - r.lse * FA_LOG2E : -INFINITY;  // P * sc = 2^(x + this); -inf: rows past T
      rowv[st * FA_ROWV + 128 + t] = p.gate ? r.gate * FA_LOG2E : 0.f;
#endif
      rowv[st * FA_ROWV + 64 + t] = r.delta * p.inv_sc;
      if constexpr (!BITS) rowv[st * FA_ROWV + 192 + t] = __uint_as_float(fa_row_word(p.s0, r.o));
    }
    if constexpr (BITS) rowv[st * FA_ROWV + 192 + t] = __uint_as_float(r.word);   // every wave its own 64 words
  };

#if FA_TILE_SRC
  FaTileSrc qsrc, dosrc;
  qsrc.init(base, D3, T);
  dosrc.init(dobase, D, T);
#define FA_LOAD_QDO(ROW0, QB_, DB_) do { qsrc.issue(ROW0, QB_, wave_u); dosrc.issue(ROW0, DB_, wave_u); } while (0)
#else
#define FA_LOAD_QDO(ROW0, QB_, DB_) do { glds_tile64(base, D3, ROW0, T, QB_, wave_u); glds_tile64(dobase, D, ROW0, T, DB_, wave_u); } while (0)
#endif
  FA_LOAD_QDO(0, qbuf(0), dobuf(0));
  rows_store(rows_load(0), 0);
  fa_tile_sync();

  int cur = 0;
  for (int it = 0; it < nq; ++it) {
    const int iq0 = it * FA_BQ1;
    const bool more = it + 1 < nq;
    const RowRegs nextrows = rows_load(more ? it + 1 : it);   // (last iteration: a harmless reload into the dead stage)
    if (more) {
      FA_LOAD_QDO(iq0 + FA_BQ1, qbuf(cur ^ 1), dobuf(cur ^ 1));
    }
    const float* rv = rowv + cur * FA_ROWV;
#if FA_DKV_RSM
    const unsigned char* rsc = reinterpret_cast<const unsigned char*>(rowv) + cur * FA_RSM_STAGE;
    const unsigned* rwords = reinterpret_cast<const unsigned*>(rsc + 3072) + (BITS ? 64 * wave_u : 0);
    // B operand of the scalar products: ones in k slots 0..2 (held by the hi = 0 half-wave), zeros elsewhere -- both half-waves
    // read the same fragment of their row, the hi = 1 copy meets zeros
    U4 onesb; onesb.v = hi == 0 ? make_uint4(0x3f803f80u, 0x00003f80u, 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);
#endif
    // S = Q K^T, dP = dO V^T  (rows = queries of the tile, col = this lane's key)
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      U4 pf[2], dsf[2];
      f32x16_t s, dp;
#if FA_DKV_BIAS_IN_C
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int il0 = 32 * f + 8 * q4 + 4 * hi;
        const float4 lse4 = *reinterpret_cast<const float4*>(rv + il0);
        const float4 gat4 = *reinterpret_cast<const float4*>(rv + 128 + il0);
        const float* tq = tcol - (iq0 + il0);
        s[4 * q4] = fmaf(gat4.x, tq[0], lse4.x);
        s[4 * q4 + 1] = fmaf(gat4.y, tq[-1], lse4.y);
        s[4 * q4 + 2] = fmaf(gat4.z, tq[-2], lse4.z);
        s[4 * q4 + 3] = fmaf(gat4.w, tq[-3], lse4.w);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[r] = 0.f;
#else
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#endif
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain(qbuf(cur), 32 * f + kl, kk, hi), kf[kk].b, s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain(dobuf(cur), 32 * f + kl, kk, hi), vf[kk].b, dp, 0, 0, 0);
      }
#if FA_DKV_RSM
      // the block's per-row scalars through the matrix pipe: lse / sc2 as a fifth k step of the scores; gate and delta as
      // [32 rows] x [this lane's key] blocks in the scores' register layout
      f32x16_t gblk, dblk;
#pragma unroll
      for (int r = 0; r < 16; ++r) { gblk[r] = 0.f; dblk[r] = 0.f; }
      {
        U4 a0, a1, a2;
        a0.v = *reinterpret_cast<const uint4*>(rsc + (32 * f + kl) * 16);
        a1.v = *reinterpret_cast<const uint4*>(rsc + 1024 + (32 * f + kl) * 16);
        a2.v = *reinterpret_cast<const uint4*>(rsc + 2048 + (32 * f + kl) * 16);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.b, onesb.b, s, 0, 0, 0);
        gblk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.b, onesb.b, gblk, 0, 0, 0);
        dblk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2.b, onesb.b, dblk, 0, 0, 0);
      }
#endif
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        // registers 4 q4 .. 4 q4 + 3 of a block are four consecutive query rows: per-row scalars come as 16-byte
        // LDS vectors; the Toeplitz entries rel[j - i] run downwards in i
        const int il0 = 32 * f + 8 * q4 + 4 * hi;
#if FA_DKV_RSM
        uint4 row4 = make_uint4(0, 0, 0, 0);
        if constexpr (DROP) row4 = *reinterpret_cast<const uint4*>(rwords + il0);
        const float delv[4] = {dblk[4 * q4], dblk[4 * q4 + 1], dblk[4 * q4 + 2], dblk[4 * q4 + 3]};
        const unsigned roww[4] = {row4.x, row4.y, row4.z, row4.w};
        const float gatv[4] = {gblk[4 * q4], gblk[4 * q4 + 1], gblk[4 * q4 + 2], gblk[4 * q4 + 3]};
        const float* tq = tcol - (iq0 + il0);
#else
        const float4 del4 = *reinterpret_cast<const float4*>(rv + 64 + il0);
        uint4 row4 = make_uint4(0, 0, 0, 0);
        if constexpr (DROP) row4 = *reinterpret_cast<const uint4*>(rv + 192 + (BITS ? 64 * wave_u : 0) + il0);
        const float delv[4] = {del4.x, del4.y, del4.z, del4.w};
        const unsigned roww[4] = {row4.x, row4.y, row4.z, row4.w};
#endif
#if !FA_DKV_BIAS_IN_C && !FA_DKV_RSM
        const float4 lse4 = *reinterpret_cast<const float4*>(rv + il0);
        const float4 gat4 = *reinterpret_cast<const float4*>(rv + 128 + il0);
        const float lsev[4] = {lse4.x, lse4.y, lse4.z, lse4.w};
        const float gatv[4] = {gat4.x, gat4.y, gat4.z, gat4.w};
        const float* tq = tcol - (iq0 + il0);
#endif
        float pv[4], dsv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int rr = 4 * q4 + e;
          // x - lse as two fmas (the -lse rides in the bias term)
#if FA_DKV_RSM
          const float pe = __builtin_amdgcn_exp2f(fmaf(s[rr], p.sc2, gatv[e] * tq[-e]));  // (s carries lse / sc2; rows past T: -1e30 -> 0)
#elif FA_DKV_BIAS_IN_C
          const float pe = __builtin_amdgcn_exp2f(s[rr] * p.sc2);  // rows past T: -inf -> 0
#else
          const float pe = __builtin_amdgcn_exp2f(fmaf(s[rr], p.sc2, fmaf(gatv[e], tq[-e], lsev[e])));  // rows past T: -inf -> 0
#endif
          float pd = pe;
          if constexpr (BITS) {
            // bit `key` of the row's word, sign-extended: 0 / ~0 (v_bfe_i32), applied to the bits of the probability
            const int km = __builtin_amdgcn_sbfe(roww[e], kbit, 1);
            pd = __uint_as_float(__float_as_uint(pe) & (unsigned)km);
          } else if constexpr (DROP) {
            const unsigned w = fa_mix(roww[e] + cw);
            const bool kp = (int)(short)((w >> csh) & 0xffffu) >= p.ths;
            pd = kp ? pe : 0.f;
          }
          pv[e] = pd;
          // dS = P (keep dP - delta) = (keep P) dP - P delta: the kept probability is needed anyway (dV), so one select
          // instead of two
          dsv[e] = fmaf(pd, dp[rr], -(pe * delv[e]));
        }
        pf[q4 >> 1].u[2 * (q4 & 1)] = pack_bf16(pv[0], pv[1]);
        pf[q4 >> 1].u[2 * (q4 & 1) + 1] = pack_bf16(pv[2], pv[3]);
        dsf[q4 >> 1].u[2 * (q4 & 1)] = pack_bf16(dsv[0], dsv[1]);
        dsf[q4 >> 1].u[2 * (q4 & 1) + 1] = pack_bf16(dsv[2], dsv[3]);
      }
      // dV^T += dO^T P ; dK^T += Q^T dS   (contraction over this 32-query block)
#pragma unroll
      for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          dv[f2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(dobuf(cur), qtr, f2, f, s2), pf[s2].b, dv[f2], 0, 0, 0);
          dk[f2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(qbuf(cur), qtr, f2, f, s2), dsf[s2].b, dk[f2], 0, 0, 0);
        }
    }
    rows_store(nextrows, cur ^ 1);
    fa_tile_sync();
    cur ^= 1;
  }
  if (!key_ok) {  // padded keys and keys past T
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk[f][r] = 0.f; dv[f][r] = 0.f; }
  }
  if (p.dbias_part) {
    float* drow = p.dbias_part + (((long)b * p.nqb + kblk) * 4 + wave_u) * D3 + h * FA_HD;
    fa_wave_colsum(dk, p.scale, reinterpret_cast<float*>(smem), drow + D, lane, wave_u);
    fa_wave_colsum(dv, 1.f, reinterpret_cast<float*>(smem), drow + 2 * D, lane, wave_u);
  }
  if (j < T) {
    bf16_t* dst = p.dqkv + ((long)b * T + j) * D3 + h * FA_HD;
#pragma unroll
    for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        uint2 w;
        w.x = pack_bf16(dk[f2][4 * q4] * p.scale, dk[f2][4 * q4 + 1] * p.scale);
        w.y = pack_bf16(dk[f2][4 * q4 + 2] * p.scale, dk[f2][4 * q4 + 3] * p.scale);
        *reinterpret_cast<uint2*>(dst + D + 32 * f2 + 8 * q4 + 4 * hi) = w;
        w.x = pack_bf16(dv[f2][4 * q4], dv[f2][4 * q4 + 1]);
        w.y = pack_bf16(dv[f2][4 * q4 + 2], dv[f2][4 * q4 + 3]);
        *reinterpret_cast<uint2*>(dst + 2 * D + 32 * f2 + 8 * q4 + 4 * hi) = w;
      }
  }
}


// ------------------------------------------------------------------- backward 2/2 with stored probabilities
// (attn_fused.hpp, "stored probabilities").  Same decomposition as above -- lane owns a key column, registers walk the
// query rows -- but P comes from the forward's fragment store instead of S = Q K^T + bias -> 2^x -> dropout word:
//   * the forward's lanes own QUERY rows, this kernel's lanes own KEYS: the fragments have to be transposed.  A wave needs,
//     per 64-query tile, the 4 KiB the forward's waves of those rows wrote for ITS 32 keys (2 row blocks x 2 sixteen-key
//     halves x 1 KiB).  They are fetched with four fully coalesced 1 KiB loads into registers, two tiles ahead (two register
//     sets, loop unrolled by two), written to a wave-private 4 KiB LDS image when their tile comes up (four conflict-free
//     ds_write_b128) and read back transposed: ONE ds_read_b64_tr_b16 per lane per (32-query block f, row group q4), 512
//     contiguous bytes per wave;
//     image: chunk = ((((f 4 + q4) 2 + g) 2 + hi) 4 + r) 2 + hi', 16 bytes each: g = 16-key half of the wave's keys, row of the
//     32-block = 8 q4 + 4 hi + r (the MFMA row mapping), hi' = which granule pair of the forward's fragment;
//   * the lane receives P16[i0 .. i0 + 3][own key] = registers 4 q4 .. 4 q4 + 3 of the 32 x 32 block; scale
//     c[i] = 2^(mt[i, key tile] + log2 sc - lse[i] log2 e) (two per row: the block's 128 keys span two forward key tiles), 0 for
//     rows past T;
//   * dS = max(Ps, 0) dP - |Ps| delta / sc; dV takes max(Ps, 0) as bf16; K is not needed at all (no score MFMA).
// The Q / dO tiles travel TWO tiles ahead through a three-stage LDS ring: every instruction-count lever on this kernel and
// on its recompute twin (-40 % VALU, -25 % MFMA, three workgroups per CU instead of two, fragments two tiles ahead) left the
// run time where it was, ~200 us -- PMC: 43 % of the wave cycles in s_waitcnt, 6 % of them for LDS: the wave waits for
// vector memory, i.e. for the tile DMA issued at the top of the SAME tile; a tile's work (~2 us at the 1.45 GHz these
// kernels run at) is shorter than the loaded round trip.  All vector memory operations of a wave complete IN ORDER, which
// dictates the shape: a wait for one of them waits for everything issued before it.  So (1) the per-row scalars of ALL
// query tiles are computed once into LDS in the prologue (a per-tile load + its wait would drag the DMA issued before it
// along), (2) per tile the fragment loads are issued BEFORE the DMA (the compiler's own wait for the next tile's
// fragments then leaves the youngest operations -- the DMA -- in flight), (3) the end-of-tile wait is vmcnt(8): this
// tile's four fragment loads + four DMA instructions stay in flight, everything older (the DMA of tile it + 1) has landed.
// LDS: 3 x (Q 8 K + dO 8 K) + 4 x 4 K (P images) + 3 x 4 K (row scalars, T <= 1024) = 76 KB: two workgroups per CU.
typedef __fp16 f16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) f16x4_t* lds_h4_ptr;
#define FA_DKVP_ROWS FA_PSTORE_MAX_T   // query rows (padded to 64) whose scalars fit the LDS row arrays
#define FA_DKVP_SMEM (3 * 16384 + 16384 + 3 * FA_DKVP_ROWS * 4)
#if FA_SP_PROBE & 32   // s_memtime stamps (shader cycles) of wave 0 of every workgroup, summed over the tiles: where a tile's time goes
__device__ unsigned long long fa_dbg_dkv[4096][12];
extern "C" int wavlm_probe_read_dkv(void* host_dst, uint64_t bytes) {
  return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(fa_dbg_dkv), bytes < sizeof(fa_dbg_dkv) ? bytes : sizeof(fa_dbg_dkv)) == hipSuccess ? 0 : -1;
}
#define FA_STAMP(K_) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tacc[K_] += t_ - tprev; tprev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define FA_STAMP(K_)
#endif
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_p_kernel(FaP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  auto qbuf = [&](int st) { return smem + st * 16384; };
  auto dobuf = [&](int st) { return smem + st * 16384 + 8192; };
  float* rowc = reinterpret_cast<float*>(smem + 4 * 16384);   // [2][FA_DKVP_ROWS] c of key tile 0 / 1 of the block, then [FA_DKVP_ROWS] delta / sc
  const int T = p.T, H = p.H;
  int kblk, bh;
  fa_block_map(p.nqb, p.B * H, kblk, bh);
  const int b = bh / H, h = bh % H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int hi = lane >> 5, kl = lane & 31;
  const int j = kblk * FA_BK1 + 32 * wave + kl;
  const int jc = j < T ? j : T - 1;
  const long D3 = 3L * H * FA_HD, D = (long)H * FA_HD;
  const bf16_t* base = p.qkv + (long)b * T * D3 + h * FA_HD;
  const bf16_t* dobase = p.dO + (long)b * T * D + h * FA_HD;
  unsigned char* pw = smem + 3 * 16384 + wave_u * 4096;   // this wave's P16 image
  const int nq = (T + FA_BQ1 - 1) / FA_BQ1;

  // ---- per-row scalars of every query tile (see the header comment, (1))
  {
    const float* mtb = p.ps.mt + (long)bh * p.ps.nkv * p.ps.Tq;
    int jt0 = 2 * kblk, jt1 = 2 * kblk + 1;
    if (jt0 > p.ps.nkv - 1) jt0 = p.ps.nkv - 1;
    if (jt1 > p.ps.nkv - 1) jt1 = p.ps.nkv - 1;
    for (int ii = threadIdx.x; ii < nq * FA_BQ1; ii += 256) {
      const bool ok = ii < T;
      const long o = (long)bh * T + (ok ? ii : T - 1);
      const float nl = p.log2sc - p.lse[o] * FA_LOG2E;
      const float m0 = mtb[(long)jt0 * p.ps.Tq + ii], m1 = mtb[(long)jt1 * p.ps.Tq + ii];   // (ii < Tq: the row grid is padded to 128)
      rowc[ii] = (ok && m0 > -INFINITY) ? __builtin_amdgcn_exp2f(m0 + nl) : 0.f;
      rowc[FA_DKVP_ROWS + ii] = (ok && m1 > -INFINITY) ? __builtin_amdgcn_exp2f(m1 + nl) : 0.f;
      rowc[2 * FA_DKVP_ROWS + ii] = p.delta[o] * p.inv_sc;
    }
  }

  U4 vf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) vf[kk].v = *reinterpret_cast<const uint4*>(base + 2 * D + (long)jc * D3 + 16 * kk + 8 * hi);
  const bool key_ok = j < T && !(p.kpm && p.kpm[(long)b * T + jc]);

  f32x16_t dk[2], dv[2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[f][r] = 0.f; dv[f][r] = 0.f; }
  const unsigned qtr = fa_tr_base(lane);

  // ---- P16 fetch.  Load (f, g) of a tile: the 1 KiB the forward's wave of row block 2 it + f wrote for fragment
  // (f' = wave & 1, s2' = g) of key tile 2 kblk + (wave >> 1).  Lane l takes the chunk of (row % 32 = 8 (l >> 4) + 4 ((l >> 3) & 1)
  // + (l & 3), half hi' = (l >> 2) & 1) -- a permutation inside the 1 KiB, chosen so that eight consecutive lanes write eight
  // different 16-byte slots of a 128-byte LDS line (r, hi' vary fastest in the image).
  int jt_w = 2 * kblk + (wave_u >> 1); if (jt_w > p.ps.nkv - 1) jt_w = p.ps.nkv - 1;   // (keys past the last tile are discarded)
  const int l_r = lane & 3, l_hp = (lane >> 2) & 1, l_hi = (lane >> 3) & 1, l_q4 = lane >> 4;
  const unsigned char* psrc = p.ps.P16 + (((long)bh * p.ps.nq32) * p.ps.nkv + jt_w) * FA_PTILE_BYTES + (wave_u & 1) * 2048 +
                              ((8 * l_q4 + 4 * l_hi + l_r) + 32 * l_hp) * 16;
  const long q32_stride = (long)p.ps.nkv * FA_PTILE_BYTES;
  unsigned char* pw_st = pw + l_q4 * 512 + l_hi * 128 + l_r * 32 + l_hp * 16;   // + f 2048 + g 256
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t pA[4], pB[4];   // [f 2 + g]
  // asm loads: the compiler does not see them, so it puts no wait of its own in front of the use of a set (its count of the
  // operations "behind" the load cannot include the DMA instructions and would also wait for the set loaded one tile ago);
  // the end-of-tile vmcnt(8) leaves only the current tile's eight operations in flight, so a set is complete one tile
  // before it is used
  auto p_load = [&](int it, u32x4_t (&x)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const unsigned char* src = psrc + (long)(2 * it + f) * q32_stride;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x[f * 2]) : "v"(src) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(x[f * 2 + 1]) : "v"(src) : "memory");
    }
  };
  FaTileSrc qsrc, dosrc;
  qsrc.init(base, D3, T);
  dosrc.init(dobase, D, T);
  auto dma_tile = [&](int it, int st) __attribute__((always_inline)) {
    qsrc.issue(it * FA_BQ1, qbuf(st), wave_u);
    dosrc.issue(it * FA_BQ1, dobuf(st), wave_u);
  };
  // transposing read of (f, q4): lane (g = (lane >> 4) & 1, hi, i = lane & 15: row r = i >> 2, granule c4 = i & 3) addresses
  // chunk (f, q4, g, hi, r, hi' = c4 & 1), half c4 >> 1; it receives rows r = 0..3 at key 16 g + i of the wave
  const unsigned char* pt = pw + ((((lane >> 4) & 1) * 2 + hi) * 4 + ((lane & 15) >> 2)) * 32 + (lane & 1) * 16 + ((lane >> 1) & 1) * 8;

  // every load the compiler knows of (V fragments, key mask, row scalars) has landed before the first operation it does not
  // know of is issued: behind an explicit vmcnt(0) its wait-count pass puts no `s_waitcnt vmcnt` of its own into the tile
  // loop (it would otherwise wait for the V fragments at their first use INSIDE the loop, i.e. drain the prefetch every tile)
  __builtin_amdgcn_s_waitcnt(0x0f70);
  // prologue: fragments of tiles 0 and 1, DMA of tiles 0 and 1; tile 0 has landed when at most tile 1's eight operations are
  // still in flight
  p_load(0, pA);
  dma_tile(0, 0);
  if (nq > 1) { p_load(1, pB); dma_tile(1, 1); }
  __builtin_amdgcn_sched_barrier(0);
  if (nq > 1) fa_tile_sync<8>(); else fa_tile_sync<0>();

  int st = 0;   // LDS stage of the current tile
#if FA_SP_PROBE & 32
  unsigned long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long tstart = __builtin_amdgcn_s_memtime();
  unsigned long long tprev = tstart;
#endif
  // one query tile; x: the register set that holds this tile's fragments and is refilled with tile it + 2
  auto tile_body = [&](int it, u32x4_t (&x)[4]) __attribute__((always_inline)) {
    const int iq0 = it * FA_BQ1;
    FA_STAMP(0);   // (loop overhead / previous barrier release -> here)
    // fragments -> the wave's LDS image (the reads of the previous tile precede these writes in the wave's LDS queue); asm,
    // so that the stores stay in front of the refill of the same registers
    if (!(FA_SP_PROBE & 8) || it == 0) {
      const unsigned pw_a = (unsigned)(unsigned long)(las_ptr)pw_st;
      asm volatile("ds_write_b128 %0, %1" :: "v"(pw_a), "v"(x[0]) : "memory");
      asm volatile("ds_write_b128 %0, %1 offset:256" :: "v"(pw_a), "v"(x[1]) : "memory");
      asm volatile("ds_write_b128 %0, %1 offset:2048" :: "v"(pw_a), "v"(x[2]) : "memory");
      asm volatile("ds_write_b128 %0, %1 offset:2304" :: "v"(pw_a), "v"(x[3]) : "memory");
    }
    // tile it + 2: fragments first, DMA last (header comment, (2)); its stage was read last in tile it - 1
    // (NO load past the end: these are asm loads the compiler does not know of -- a "harmless reload" issued in the last
    // tiles would still be in flight when the loop ends, the register set is dead by then, the compiler hands its registers
    // to the epilogue, and the returning data overwrites dK on its way out: seen as non-deterministic dK for odd tile counts)
    const bool ahead = it + 2 < nq;
    if (!(FA_SP_PROBE & 1) && ahead) p_load(it + 2, x);
    int st2 = st + 2; if (st2 >= 3) st2 -= 3;
    if (ahead) dma_tile(it + 2, st2);
    __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise sinks the loads to the end of the tile)
    FA_STAMP(1);   // fragment copies + ds_write, loads and DMA issued
    const float* rv = rowc + (wave_u >> 1) * FA_DKVP_ROWS + iq0;   // c of this wave's key tile
    const float* rd = rowc + 2 * FA_DKVP_ROWS + iq0;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      U4 pf[2], dsf[2];
      f32x16_t dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain(dobuf(st), 32 * f + kl, kk, hi), vf[kk].b, dp, 0, 0, 0);
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int il0 = 32 * f + 8 * q4 + 4 * hi;
        const f16x4_t ph = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4_ptr)(pt + (f * 4 + q4) * 512));
        const float4 c4 = *reinterpret_cast<const float4*>(rv + il0);
        const float4 del4 = *reinterpret_cast<const float4*>(rd + il0);
        const float cv[4] = {c4.x, c4.y, c4.z, c4.w};
        const float delv[4] = {del4.x, del4.y, del4.z, del4.w};
        float pv[4], dsv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ps = (float)ph[e] * cv[e];          // signed P sc (sign = dropped)
          const float pk = fmaxf(ps, 0.f);                // kept
          pv[e] = pk;
          dsv[e] = fmaf(pk, dp[4 * q4 + e], -(__builtin_fabsf(ps) * delv[e]));
        }
        pf[q4 >> 1].u[2 * (q4 & 1)] = pack_bf16(pv[0], pv[1]);
        pf[q4 >> 1].u[2 * (q4 & 1) + 1] = pack_bf16(pv[2], pv[3]);
        dsf[q4 >> 1].u[2 * (q4 & 1)] = pack_bf16(dsv[0], dsv[1]);
        dsf[q4 >> 1].u[2 * (q4 & 1) + 1] = pack_bf16(dsv[2], dsv[3]);
      }
      if (f == 0) FA_STAMP(2); else FA_STAMP(4);   // dP MFMAs + element pass of the block (waits for the MFMA results)
#pragma unroll
      for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          dv[f2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(dobuf(st), qtr, f2, f, s2), pf[s2].b, dv[f2], 0, 0, 0);
          dk[f2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(qbuf(st), qtr, f2, f, s2), dsf[s2].b, dk[f2], 0, 0, 0);
        }
      if (f == 0) FA_STAMP(3); else FA_STAMP(5);   // the block's eight dV / dK MFMAs issued (fragment reads waited for)
    }
#if FA_SP_PROBE & 32
    if (ahead) __builtin_amdgcn_s_waitcnt(0x0f70 | 8); else __builtin_amdgcn_s_waitcnt(0x0f70);
    FA_STAMP(6);   // vector memory wait
    __syncthreads();
    FA_STAMP(7);   // barrier
    st = st + 1; if (st >= 3) st = 0;
    return;
#endif
    // tile it + 1 has landed when at most this tile's eight operations (four fragment loads, four DMA instructions) are in
    // flight; the last two tiles issue nothing
    if (ahead) fa_tile_sync<8>(); else fa_tile_sync<0>();
    st = st + 1; if (st >= 3) st = 0;
  };
  // whole pairs, then the odd tile: the pair loop's only back edge is B -> A.  (With `if (it + 1 < nq) tile B` inside one loop
  // the compiler's control-flow graph keeps a path from tile A straight back to tile A and its wait-count pass waits for
  // A's refill at the top of A, i.e. for everything issued so far.)
  for (int pi = 0; pi < (nq >> 1); ++pi) {
    tile_body(2 * pi, pA);
    tile_body(2 * pi + 1, pB);
  }
  if (nq & 1) tile_body(nq - 1, pA);
#if FA_SP_PROBE & 32
  if (threadIdx.x == 0 && blockIdx.x < 4096) {
    for (int k = 0; k < 8; ++k) fa_dbg_dkv[blockIdx.x][k] = tacc[k];
    fa_dbg_dkv[blockIdx.x][8] = tstart;
    fa_dbg_dkv[blockIdx.x][9] = tprev;
    fa_dbg_dkv[blockIdx.x][10] = (unsigned long long)nq;
  }
#endif
  if (!key_ok) {  // padded keys and keys past T
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk[f][r] = 0.f; dv[f][r] = 0.f; }
  }
  __syncthreads();   // (the column-sum scratch below overlays the tiles AND the other waves' P images)
  if (p.dbias_part) {
    float* drow = p.dbias_part + (((long)b * p.nqb + kblk) * 4 + wave_u) * D3 + h * FA_HD;
    fa_wave_colsum(dk, p.scale, reinterpret_cast<float*>(smem), drow + D, lane, wave_u);
    fa_wave_colsum(dv, 1.f, reinterpret_cast<float*>(smem), drow + 2 * D, lane, wave_u);
  }
  if (j < T) {
    bf16_t* dst = p.dqkv + ((long)b * T + j) * D3 + h * FA_HD;
#pragma unroll
    for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        uint2 w;
        w.x = pack_bf16(dk[f2][4 * q4] * p.scale, dk[f2][4 * q4 + 1] * p.scale);
        w.y = pack_bf16(dk[f2][4 * q4 + 2] * p.scale, dk[f2][4 * q4 + 3] * p.scale);
        *reinterpret_cast<uint2*>(dst + D + 32 * f2 + 8 * q4 + 4 * hi) = w;
        w.x = pack_bf16(dv[f2][4 * q4], dv[f2][4 * q4 + 1]);
        w.y = pack_bf16(dv[f2][4 * q4 + 2], dv[f2][4 * q4 + 3]);
        *reinterpret_cast<uint2*>(dst + 2 * D + 32 * f2 + 8 * q4 + 4 * hi) = w;
      }
  }
}

int fa_launch_dkv(const FaP& p, unsigned grid, size_t smem, hipStream_t st) {
  if (p.ps.P16) {
    if ((p.T + FA_BQ1 - 1) / FA_BQ1 * FA_BQ1 > FA_DKVP_ROWS) return WL_EINVAL;   // (the caller falls back to recomputation)
    size_t smem_p = FA_DKVP_SMEM;
    if (fa_set_smem(attn_bwd_dkv_p_kernel, smem_p) != WL_OK) return WL_ELAUNCH;
    WL_LAUNCH(attn_bwd_dkv_p_kernel, dim3(grid), dim3(256), smem_p, st, p);
    return wl_check_launch();
  }
  if (p.th && p.dbits) {
    if (fa_set_smem(attn_bwd_dkv_kernel<true, true>, smem) != WL_OK) return WL_ELAUNCH;
    WL_LAUNCH((attn_bwd_dkv_kernel<true, true>), dim3(grid), dim3(256), smem, st, p);
  } else if (p.th) {
    if (fa_set_smem(attn_bwd_dkv_kernel<true>, smem) != WL_OK) return WL_ELAUNCH;
    WL_LAUNCH(attn_bwd_dkv_kernel<true>, dim3(grid), dim3(256), smem, st, p);
  } else {
    if (fa_set_smem(attn_bwd_dkv_kernel<false>, smem) != WL_OK) return WL_ELAUNCH;
    WL_LAUNCH(attn_bwd_dkv_kernel<false>, dim3(grid), dim3(256), smem, st, p);
  }
  return wl_check_launch();
}
