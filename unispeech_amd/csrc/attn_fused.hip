// Fused gated-relative-position attention: forward and dQ kernels, launchers (shared pieces and the design notes:
// attn_fused.hpp; dK/dV kernel: attn_fused_dkv.hip).
#include "attn_fused.hpp"

typedef float f32x2_t __attribute__((ext_vector_type(2)));
// FA_FWD_LAZY: lazy running maximum + packed row sums in the forward (see the tile loop).  Measured -4 % on the forward
// kernel (profiles/r03/ab_attn_lazy_pk.txt) and OFF: with the exact running maximum the dominant key of a row has
// P = 2^0 = 1, which bf16 holds exactly; relative to a stale reference it is 2^frac and rounds like every other term.  The
// 12-layer gradient comparison with the fp32 oracle (tests/test_bf16_e2e_gpu.py) keeps its median (1.52e-2 against
// 1.59e-2 relative L2) but its worst tensor moves from 3.2e-2 to 4.4e-2 of a 4e-2 bound -- not worth 5 us per layer.
#if !defined(WAVLM_EXPERIMENTAL)   // lab-bench switches exist only in -DWAVLM_EXPERIMENTAL builds (tools/probe/build_probe.py)
#undef FA_FWD_LAZY
#undef FA_FWD_THETA
#undef FA_DQ_PK
#endif
#ifndef FA_FWD_LAZY
#define FA_FWD_LAZY 0
#endif
#ifndef FA_FWD_THETA
#define FA_FWD_THETA 8.f
#endif
// FA_DQ_PK: packed-fp32 bias fmas and gate-gradient accumulation in the dQ kernel's element pass
#ifndef FA_DQ_PK
#define FA_DQ_PK 1
#endif
// FA_FWD_MFMA_SUM (round 6, lab switch): the forward's row sums out of four all-ones MFMAs per tile (the matrix pipe is 20 % busy)
// instead of 32 v_add_f32 + a half-wave exchange: l = sum of the bf16-ROUNDED unmasked probabilities.  Parity green, measured
// NEUTRAL (119.8 against 119.8 us, three alternating same-box runs, profiles/r06/ab_attn_fwd_mfma_sum.txt): OFF, the numbers of
// rounds 1-5 stay
#if !defined(WAVLM_EXPERIMENTAL)
#undef FA_FWD_MFMA_SUM
#endif
#ifndef FA_FWD_MFMA_SUM
#define FA_FWD_MFMA_SUM 0
#endif
// FA_FWD_BIAS_IN_C: the forward's Toeplitz bias rides in the C operand of the score MFMAs (see the tile loop)
#ifndef FA_FWD_BIAS_IN_C
#define FA_FWD_BIAS_IN_C 1
#endif
// FA_DQ_BIAS_IN_C: the same in the dQ kernel (bias - lse: no instruction less there, the fma only moves in front of the
// MFMAs).  Measured 215 -> 219 us (profiles/r04/ab_attn_bwd_bias_in_c.txt): OFF.
#ifndef FA_DQ_BIAS_IN_C
#define FA_DQ_BIAS_IN_C 0
#endif

// ------------------------------------------------------------------------------------------------- forward
// STORE: the probabilities (fp16, sign = dropped) and the running maxima go to p.ps for the backward (attn_fused.hpp)
// BSTORE (round 6): the dropout DECISIONS go to p.dbits as bit words (attn_fused.hpp: fa_fbit_of_key) -- the forward evaluates the
// hash anyway; the dQ and dK/dV kernels of the backward then select with stored bits (two instructions per element) instead of
// evaluating it twice more.  One v_lshrrev + one v_and_or per pair here, one 4-byte store per lane and tile.
template <bool DROP, bool STORE = false, bool BSTORE = false>
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(FaP p) {
  static_assert(!STORE || (FA_FWD_BIAS_IN_C && !FA_FWD_LAZY), "the probability store is written for the default forward");
  static_assert(!BSTORE || (DROP && !STORE && FA_FWD_BIAS_IN_C && !FA_FWD_LAZY), "the bit store is written for the default forward with dropout");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  auto kbuf = [&](int st) { return smem + st * 16384; };
  auto vbuf = [&](int st) { return smem + st * 16384 + 8192; };
  float* tabs = reinterpret_cast<float*>(smem + 32768);
  float* kb = tabs + p.Ltab;
  unsigned* colw = reinterpret_cast<unsigned*>(kb + p.Tkb);  // [Tkb / 2] dropout column words
  const int T = p.T, H = p.H;
  int qblk, bh;
  fa_block_map(p.nqb, p.B * H, qblk, bh);
  const int b = bh / H, h = bh % H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int hi = lane >> 5, ql = lane & 31;
  const int i = qblk * FA_BQ + 32 * wave + ql;
  const int ic = i < T ? i : T - 1;
  const long D3 = 3L * H * FA_HD;
  const bf16_t* base = p.qkv + (long)b * T * D3 + h * FA_HD;
  const int L = 2 * T - 1;

  U4 qf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) qf[kk].v = *reinterpret_cast<const uint4*>(base + (long)ic * D3 + 16 * kk + 8 * hi);
  for (int d = threadIdx.x; d < p.Ltab; d += 256) tabs[d] = (p.tab && d < L) ? p.tab[(long)h * L + d] : 0.f;
  for (int j = threadIdx.x; j < p.Tkb; j += 256)
    kb[j] = (j < T && !(p.kpm && p.kpm[(long)b * T + j])) ? 0.f : -INFINITY;
  if constexpr (DROP)
    for (int jp = threadIdx.x; jp < (p.Tkb >> 1); jp += 256) colw[jp] = fa_col_word(p.s1, (unsigned)jp);
  // everything below lives in the log2 domain: x2 = log2(e) * (scale * s + gate * rel), p = 2^(x2 - m2)
  const float g2 = p.gate ? p.gate[(long)bh * T + ic] * FA_LOG2E : 0.f;
#if FA_FWD_BIAS_IN_C
  const float g2s = g2 / p.sc2;
#endif
  const unsigned roww = fa_row_word(p.s0, (unsigned)(bh * T + ic));
  const float* trow = tabs + (T - 1 - ic);  // trow[j] = rel[h, j - i]

  f32x16_t o[2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[f][r] = 0.f;
  float m = -INFINITY, l = 0.f;
#if FA_FWD_LAZY
  float mref = 0.f;
  bool seen = false;
#endif
  const int nkv = (T + FA_BKV - 1) / FA_BKV;
  const unsigned vtr = fa_tr_base(lane);
  // BSTORE: this lane's row in the bit words; lanes 0..31 store the word of key block 2 jt, lanes 32..63 of 2 jt + 1
  unsigned* dbp = nullptr;
  if constexpr (BSTORE) dbp = p.dbits + ((long)bh * p.db_nkb + hi) * p.db_Tq + fa_bitrow(i);
  // STORE: this wave's tile row of the fragment-native probability store and its column of the running maxima
  unsigned char* pst = nullptr;
  float* mtp = nullptr;
  if constexpr (STORE) {
    pst = p.ps.P16 + (((long)bh * p.ps.nq32 + (qblk * 4 + wave_u)) * p.ps.nkv) * FA_PTILE_BYTES + lane * 16;
    mtp = p.ps.mt + (long)bh * p.ps.nkv * p.ps.Tq + (qblk * FA_BQ + 32 * wave_u + ql);
  }

#if FA_TILE_SRC
  FaTileSrc ksrc, vsrc;
  ksrc.init(base + FA_HD * H, D3, T);
  vsrc.init(base + 2 * FA_HD * H, D3, T);
#define FA_LOAD_KV(ROW0, KB_, VB_) do { ksrc.issue(ROW0, KB_, wave_u); vsrc.issue(ROW0, VB_, wave_u); } while (0)
#else
#define FA_LOAD_KV(ROW0, KB_, VB_) do { glds_tile64(base + FA_HD * H, D3, ROW0, T, KB_, wave_u); glds_tile64(base + 2 * FA_HD * H, D3, ROW0, T, VB_, wave_u); } while (0)
#endif
  FA_LOAD_KV(0, kbuf(0), vbuf(0));
  fa_tile_sync();

  int cur = 0;
  for (int jt = 0; jt < nkv; ++jt) {
    const int j0 = jt * FA_BKV;
    const bool more = jt + 1 < nkv;
    if (more) {
      FA_LOAD_KV(j0 + FA_BKV, kbuf(cur ^ 1), vbuf(cur ^ 1));
    }
    const unsigned* cwp = colw + ((j0 + 4 * hi) >> 1);  // j0 + 4 hi is even, r walks in pairs
    // S^T = K Q^T
    f32x16_t s[2];
#if FA_FWD_BIAS_IN_C
    // the Toeplitz bias (and the key mask of edge tiles) enters as the C operand of the score MFMAs, in units of 1 / sc2: the
    // element pass is then max, fma (x sc2 - m), 2^x -- one VALU instruction per score less than bias-fma + subtract
    const bool edge = (p.kpm != nullptr) || (j0 + FA_BKV > T);
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      if (!edge) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[f][r] = g2s * trow[j0 + 32 * f + (r & 3) + 8 * (r >> 2) + 4 * hi];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = j0 + 32 * f + (r & 3) + 8 * (r >> 2) + 4 * hi;
          s[f][r] = fmaf(g2s, trow[j], kb[j]);   // kb is 0 or -inf: scale-free
        }
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        s[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain(kbuf(cur), 32 * f + ql, kk, hi), qf[kk].b, s[f], 0, 0, 0);
    }
#else
#pragma unroll
    for (int f = 0; f < 2; ++f) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[f][r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        s[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain(kbuf(cur), 32 * f + ql, kk, hi), qf[kk].b, s[f], 0, 0, 0);
    }
#endif
#if FA_FWD_LAZY
    // Lazy running maximum: the scores come out of the bias fma already relative to the row's reference `mref` (the
    // subtraction rides in the fma's addend), and the reference moves only when a tile's maximum exceeds it by more than
    // FA_FWD_THETA (log2 units) -- then this tile's 32 values are shifted and o / l rescaled; otherwise 2^x is taken
    // directly (values up to 2^THETA, harmless in fp32 sums and scale-free in bf16).  Removes the per-element subtract.
    float tmax = -INFINITY;
    const bool edge = (p.kpm != nullptr) || (j0 + FA_BKV > T);
    const float nm = -mref;
    if (!edge) {
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = j0 + 32 * f + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const float x = fmaf(s[f][r], p.sc2, fmaf(g2, trow[j], nm));
          s[f][r] = x;
          tmax = fmaxf(tmax, x);
        }
    } else {
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = j0 + 32 * f + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const float x = fmaf(s[f][r], p.sc2, fmaf(g2, trow[j], nm)) + kb[j];
          s[f][r] = x;
          tmax = fmaxf(tmax, x);
        }
    }
    tmax = wl_max_xor32(tmax);
    const bool fin = tmax > -INFINITY;
    // a row without a finite score so far takes the first finite maximum as its reference (either sign, o = l = 0)
    const float delta = !seen ? (fin ? tmax : 0.f) : (tmax > FA_FWD_THETA ? tmax : 0.f);
    if (__any(delta != 0.f)) {
      const float alpha = seen ? __builtin_amdgcn_exp2f(-delta) : 1.f;
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[f][r] -= delta;
      l *= alpha;
#pragma unroll
      for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[f2][r] *= alpha;
      mref += delta;
    }
    seen = seen || fin;
    f32x2_t rs2 = f32x2_t{0.f, 0.f};
    U4 pf[2][2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __builtin_amdgcn_exp2f(s[f][r]);
        const float p1 = __builtin_amdgcn_exp2f(s[f][r + 1]);
        rs2 += f32x2_t{p0, p1};
        unsigned pk = pack_bf16(p0, p1);
        if constexpr (DROP) pk &= fa_keepmask2(fa_mix(roww + cwp[(32 * f + (r & 3) + 8 * (r >> 2)) >> 1]), p.k2);
        pf[f][r >> 3].u[(r & 7) >> 1] = pk;
      }
    l += wl_sum_xor32(rs2[0] + rs2[1]);
#else
#if FA_FWD_BIAS_IN_C
    float tmax = -INFINITY;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[f][r]);
    tmax = wl_max_xor32(tmax) * p.sc2;          // sc2 > 0: the maximum commutes with the scale
    const float m_new = fmaxf(m, tmax);
    const bool dead = (m_new == -INFINITY);
    const float alpha = dead ? 1.f : __builtin_amdgcn_exp2f(m - m_new);
    const float nmsub = dead ? 0.f : -m_new;  // dead rows: every x is -inf -> 2^(-inf) = 0
    float rs = 0.f;
    U4 pf[2][2];
    U4 pu[(FA_FWD_MFMA_SUM && DROP) ? 2 : 1][2];   // the unmasked pairs (DROP: pf is masked)
    U4 pst16[STORE ? 2 : 1][2];
    unsigned bacc[2] = {0u, 0u};
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __builtin_amdgcn_exp2f(fmaf(s[f][r], p.sc2, nmsub));
        const float p1 = __builtin_amdgcn_exp2f(fmaf(s[f][r + 1], p.sc2, nmsub));
#if !FA_FWD_MFMA_SUM
        rs += p0 + p1;
#endif
        unsigned pk = pack_bf16(p0, p1);
        if constexpr (FA_FWD_MFMA_SUM && DROP) pu[f][r >> 3].u[(r & 7) >> 1] = pk;
        if constexpr (STORE) {
          // fp16 pair (round towards zero: one v_cvt_pkrtz_f16_f32); the sign bit takes the drop decision, the bf16 pair of
          // the PV product is masked with the same word
          unsigned h2 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(p0, p1));
          if constexpr (DROP) {
            const unsigned dm = fa_dropmask2(fa_mix(roww + cwp[(32 * f + (r & 3) + 8 * (r >> 2)) >> 1]), p.k3);
            pk &= ~dm;
            h2 |= dm & 0x80008000u;
          }
          pst16[f][r >> 3].u[(r & 7) >> 1] = h2;
        } else {
          if constexpr (DROP) {
            const unsigned km = fa_keepmask2(fa_mix(roww + cwp[(32 * f + (r & 3) + 8 * (r >> 2)) >> 1]), p.k2);
            pk &= km;
            if constexpr (BSTORE) bacc[f] = (bacc[f] >> 1) | (km & 0x80008000u);   // the pair's decisions enter at bits 15 / 31
          }
        }
        pf[f][r >> 3].u[(r & 7) >> 1] = pk;
      }
#if FA_FWD_MFMA_SUM
    {
      // D[m][q] = sum over the k-step's 16 keys of P^T[key][q] for every row m: four MFMAs cover the tile's 64 keys, both
      // half-waves' k slots included -- no exchange; every register of the lane holds its query's sum
      U4 ones; ones.v = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
      f32x16_t lacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) lacc[r] = 0.f;
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
          lacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones.b, ((FA_FWD_MFMA_SUM && DROP) ? pu[f][s2] : pf[f][s2]).b, lacc, 0, 0, 0);
      rs = lacc[0];
    }
#endif
    if constexpr (BSTORE) {
      // eight pairs per block: register r sits at bit (r >> 1) + 16 (r & 1) + 8; the hi = 1 half-wave (keys + 4) moves down by 8,
      // one swap hands lanes 0..31 both halves' word of block 0 and lanes 32..63 of block 1.  The store is issued behind this
      // tile's prefetch DMA: fa_tile_sync<1> lets exactly it stay in flight
      unsigned w0 = bacc[0] >> (8 * hi), w1 = bacc[1] >> (8 * hi);
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(w0), "+v"(w1));
      dbp[(long)jt * 2 * p.db_Tq] = w0 | w1;
    }
    if constexpr (STORE && !(FA_SP_PROBE & 2)) {
      // four coalesced 1 KiB stores per wave (every lane its own 16 bytes) + the running maximum these values are relative
      // to.  Issued AFTER this tile's prefetch DMA: fa_tile_sync<5> lets exactly these five stay in flight.  Both halves
      // of the wave hold the row's maximum and write the same word: one store instruction whatever the lane mask.
      unsigned char* dstp = pst + (long)jt * FA_PTILE_BYTES;
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) *reinterpret_cast<uint4*>(dstp + (f * 2 + s2) * 1024) = pst16[f][s2].v;
      mtp[(long)jt * p.ps.Tq] = m_new;
    }
#else
    // scores -> log2 domain with the Toeplitz bias; key padding / keys past T only on edge tiles
    float tmax = -INFINITY;
    const bool edge = (p.kpm != nullptr) || (j0 + FA_BKV > T);
    if (!edge) {
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = j0 + 32 * f + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const float x = fmaf(s[f][r], p.sc2, g2 * trow[j]);
          s[f][r] = x;
          tmax = fmaxf(tmax, x);
        }
    } else {
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = j0 + 32 * f + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const float x = fmaf(s[f][r], p.sc2, g2 * trow[j]) + kb[j];
          s[f][r] = x;
          tmax = fmaxf(tmax, x);
        }
    }
    tmax = wl_max_xor32(tmax);
    const float m_new = fmaxf(m, tmax);
    const bool dead = (m_new == -INFINITY);
    const float alpha = dead ? 1.f : __builtin_amdgcn_exp2f(m - m_new);
    const float msub = dead ? 0.f : m_new;  // dead rows: every x is -inf -> 2^(-inf) = 0
    float rs = 0.f;
    U4 pf[2][2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __builtin_amdgcn_exp2f(s[f][r] - msub);
        const float p1 = __builtin_amdgcn_exp2f(s[f][r + 1] - msub);
        rs += p0 + p1;
        unsigned pk = pack_bf16(p0, p1);
        if constexpr (DROP) {
          // (lane / tile part of the pair index as a base pointer: the rest is an immediate offset of the LDS read)
          pk &= fa_keepmask2(fa_mix(roww + cwp[(32 * f + (r & 3) + 8 * (r >> 2)) >> 1]), p.k2);
        }
        pf[f][r >> 3].u[(r & 7) >> 1] = pk;
      }
#endif
#if !(FA_FWD_MFMA_SUM && FA_FWD_BIAS_IN_C && !FA_FWD_LAZY)
    rs = wl_sum_xor32(rs);
#endif
    l = l * alpha + rs;
    m = m_new;
    if (__any(alpha != 1.f)) {  // the running maximum settles after the first tiles: skip the rescale then
#pragma unroll
      for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[f2][r] *= alpha;
    }
#endif
    // O^T += V^T P^T
#pragma unroll
    for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
          o[f2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(vbuf(cur), vtr, f2, f, s2), pf[f][s2].b, o[f2], 0, 0, 0);
    if constexpr (STORE && (FA_SP_PROBE & 2)) {   // probe: keep the conversions alive without the stores
      if (pst16[0][0].u[0] == 0x12345678u && pst16[1][1].u[3] == 0x9abcdef0u && pst16[0][1].u[1] == 77u && pst16[1][0].u[2] == 78u) mtp[0] = m;
      fa_tile_sync<0>();
    } else
    fa_tile_sync<STORE ? 5 : (BSTORE ? 1 : 0)>();
    cur ^= 1;
  }
  if (i < T) {
    const float inv = l > 0.f ? p.sc / l : 0.f;  // dropout's 1/(1-p) rides on the normaliser
    bf16_t* dst = p.O + ((long)b * T + i) * (H * FA_HD) + h * FA_HD;
#pragma unroll
    for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        uint2 w;
        w.x = pack_bf16(o[f2][4 * q4] * inv, o[f2][4 * q4 + 1] * inv);
        w.y = pack_bf16(o[f2][4 * q4 + 2] * inv, o[f2][4 * q4 + 3] * inv);
        *reinterpret_cast<uint2*>(dst + 32 * f2 + 8 * q4 + 4 * hi) = w;
      }
#if FA_FWD_LAZY
    m = mref;
#endif
    if (hi == 0) p.lse[(long)bh * T + i] = (m + __log2f(l)) * FA_LN2;
  }
}

// ------------------------------------------------------------------------------- backward 1/2: dQ, dgate, drel
// Same decomposition as the forward (lane owns a query row).  Also writes delta[i] = <dO_i, O_i> for kernel 2/2.
//
// drel[h, d] = sum_i gate_i dS[i, i + d - (T-1)] is a gate-weighted sum along diagonals.  Per key tile a wave holds dS
// for 32 query rows x 64 keys, one row per lane, i.e. the diagonals run ACROSS lanes.  The skew is done by the LDS
// write address, the weighted reduction by the matrix core:
//   * lane (row rho) stores element (rho, dd) as bf16 at Sk[c][rho], c = dd - rho + 31 in [0, 95): every (c, rho) has
//     one writer, the two corner triangles are never written and stay zero from the one-time clear (the stored bf16 is
//     the very dS half that feeds the dQ MFMA -- no second conversion);
//   * gate-weighted column sums of Sk over rho = G[16 x 32] . Sk^T with G[m][rho] = gate_rho (bf16) for every m: six
//     v_mfma_f32_16x16x32_bf16 (one per 16 diagonals) with the 32 rows as the K dimension -- every output row holds the
//     sums, lane l reads diagonal 16 cb + (l & 15).  (Until round 2 the A operand was all ones and every element paid a
//     multiply by the gate plus its own bf16 conversion.);
//   * the window slides 64 diagonals per tile: blocks 0-3 are final after a tile (one coalesced 256-B store to the
//     wave's private partial row), blocks 4-5 are fed back as the C operand of blocks 0-1 of the next tile.
// (History: LDS float atomics cost ~550 cycles per wave-instruction; a register sliding window pulled with
// ds_bpermute cost ~22 VALU ops per element.  This form costs 1.5 LDS ops per element and no VALU.)
// K^T for dQ = dS K is not staged separately: the A operand is read from the K tile itself with the transposing
// ds_read_b64_tr_b16 (lane i of a 16-lane group addresses key (i >> 2), head-dim columns 4 (i & 3) .. +4).
// TAB: relative-position table present (a uniform `if (p.tab)` per element pair inside the score loop cost one basic
// block, its waits and hazard nops per pair: 34 branches and 92 s_nop per tile iteration)
// Structure switches of the dQ kernel's tile loop, both measured on the same box against the plain order (227 us):
// FA_DQ_DEFER (diagonal sums of tile t at the top of tile t + 1) 240 us, FA_DQ_BOTH (both 32-key blocks' score MFMAs
// before the element passes) 230-235 us.  Neither pays: off.
#ifndef FA_DQ_DEFER
#define FA_DQ_DEFER 0
#endif
#ifndef FA_DQ_BOTH
#define FA_DQ_BOTH 0
#endif
// FA_DQ_PREFETCH: bias-table entries and dropout column words of a tile are read into registers before its score MFMAs
// (the compiler cannot move those LDS reads above the skew writes of the block before): 229 -> 220 us.
#ifndef FA_DQ_PREFETCH
#define FA_DQ_PREFETCH 1
#endif
#ifndef FA_DQ_NOSKEW
#define FA_DQ_NOSKEW 0
#endif
// FA_DQ_SKEW_ROWS (round 6): the skew buffer of the diagonal sums ROW-major, Sk[rho][c] (row = query row rho of the wave, 96 diagonals
// + pad = 200 bytes) instead of Sk[c][rho].  A lane's four consecutive keys are then four consecutive diagonals of ITS row: one
// 8-byte store per run at an address that is only 2-byte aligned (31 - rho + 4 hi + dd: gfx950 executes unaligned 8-byte LDS
// stores correctly and at the aligned rate, tools/probe/lds_unaligned.hip) -- 8 ds_write_b64 per tile instead of 32 ds_write_b16;
// the B operand [k = rho][n = c] of the diagonal-sum MFMAs is k-major then and comes through the transposing read (two
// ds_read_b64_tr_b16 per 16-diagonal block instead of one ds_read_b128): 20 LDS instructions per tile instead of 38.  Parity green,
// measured SLOWER: 245 against 205 us (three alternating same-box runs, profiles/r06/ab_attn_dq_skew_rows.txt) -- an unaligned
// 8-byte store touches three banks per lane (16 lanes: 48 bank accesses for 32 banks, at least two passes) where 64 lanes' 2-byte
// stores are one pass: the kernel is bound by LDS PASSES, not by LDS instructions.  Lab switch, OFF.
#if !defined(WAVLM_EXPERIMENTAL)
#undef FA_DQ_SKEW_ROWS
#endif
#ifndef FA_DQ_SKEW_ROWS
#define FA_DQ_SKEW_ROWS 0
#endif
#define FA_SKEW_RB 200                                  // bytes per row of the row-major skew buffer
#define FA_SKEW_WAVE (FA_DQ_SKEW_ROWS ? 32 * FA_SKEW_RB : 6144)   // bytes per wave
#ifndef FA_DQ_OCC
#define FA_DQ_OCC 2
#endif
// SP: stored probabilities (attn_fused.hpp).  The wave reads back the very fragments the forward's wave of the same rows
// wrote (16 bytes per lane, f, s2: four coalesced 1 KiB loads per tile, issued one tile ahead) -- no score MFMA, no bias, no
// exponential, no dropout word; Q is not needed at all.  DROP is irrelevant then (the decision is the stored sign).
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
// BCONS (round 6, with DROP): the keep decisions are the bit words the FORWARD wrote (p.dbits, p.db_fwd; attn_fwd_kernel BSTORE):
// two global loads per lane and tile (its row's words of the tile's two 32-key blocks, requested one tile ahead), v_bfe_i32 +
// v_and_b32 per element -- no row word, no column words in LDS, no mix, no compare, and no bit words to produce for the dK/dV
// kernel (it reads the forward's too).
// EMIT (with DROP): the kernel leaves its dropout decisions behind as bit words for the dK/dV kernel (WAVLM_ATTN_DBITS=1).
template <bool DROP, bool TAB, bool SP = false, bool BCONS = false, bool EMIT = false>
__global__ __launch_bounds__(256, FA_DQ_OCC) void attn_bwd_dq_kernel(FaP p) {
  static_assert(!BCONS || (DROP && !SP), "stored bits are dropout decisions of the recompute form");
  static_assert(!EMIT || (DROP && !SP && !BCONS), "bit words are emitted by the recompute form with dropout");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // per stage: K [kv][hd] 8 KB | V [kv][hd] 8 KB;  then (relative-position table present) the four waves' skew buffers
  // (96 x 64 B each)
  auto kbuf = [&](int st) { return smem + st * 16384; };
  auto vbuf = [&](int st) { return smem + st * 16384 + 8192; };
  constexpr int SKEW_BYTES = TAB ? 4 * FA_SKEW_WAVE : 0;
  float* tabs = reinterpret_cast<float*>(smem + 32768 + SKEW_BYTES);
  float* kb = tabs + p.Ltab;
  unsigned* colw = reinterpret_cast<unsigned*>(kb + p.Tkb);
  // gate fragments of the four waves (64 B each) behind the dropout column words
  unsigned short* gball = reinterpret_cast<unsigned short*>(colw + (p.Tkb >> 1));
  const int T = p.T, H = p.H;
  int qblk, bh;
  fa_block_map(p.nqb, p.B * H, qblk, bh);
  const int b = bh / H, h = bh % H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int hi = lane >> 5, ql = lane & 31;
  const int i = qblk * FA_BQ + 32 * wave + ql;
  const int ic = i < T ? i : T - 1;
  const bool valid_i = i < T;
  unsigned char* skew = smem + 32768 + wave_u * FA_SKEW_WAVE;
  const int ib = qblk * FA_BQ + 32 * wave_u;
  const int dlo0 = -ib - 31 + T - 1;  // diagonal of skew row 0 at tile 0 (negative for rows past the table)
  // d(rel) partial row of THIS WAVE: the finished diagonals leave as plain coalesced 256-byte stores, one per tile.
  // (Until round 2 the four waves added into one LDS row with ds_add_f32: a float LDS atomic occupies the LDS pipe for
  // ~550 cycles per wave-instruction -- with eight waves per CU about half of the pipe's time, and the relative-position
  // part cost 90 of the kernel's 257 us.  Per-wave rows are 4x the partial bytes, 30 MB per launch, and need no memset:
  // the reduction knows which diagonals a wave writes.)
  float* prow = nullptr;
  if constexpr (TAB) prow = p.dtab_part + (((long)bh * p.nqb + qblk) * 4 + wave_u) * (long)((2 * T - 1 + 3) & ~3);
  const long D3 = 3L * H * FA_HD, D = (long)H * FA_HD;
  const bf16_t* base = p.qkv + (long)b * T * D3 + h * FA_HD;
  const int L = 2 * T - 1;

  U4 qf[4], dof[4];
  float dl = 0.f;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    if constexpr (!SP) qf[kk].v = *reinterpret_cast<const uint4*>(base + (long)ic * D3 + 16 * kk + 8 * hi);
    else qf[kk].v = make_uint4(0, 0, 0, 0);
    const long oo = ((long)b * T + ic) * D + h * FA_HD + 16 * kk + 8 * hi;
    dof[kk].v = *reinterpret_cast<const uint4*>(p.dO + oo);
    U4 ov; ov.v = *reinterpret_cast<const uint4*>(p.O + oo);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      dl += __uint_as_float(dof[kk].u[e] << 16) * __uint_as_float(ov.u[e] << 16);
      dl += __uint_as_float(dof[kk].u[e] & 0xffff0000u) * __uint_as_float(ov.u[e] & 0xffff0000u);
    }
  }
  dl = wl_sum_xor32(dl);
  const float dls = dl * p.inv_sc;
  const float ndls = -dls;
  for (int d = threadIdx.x; d < p.Ltab; d += 256) tabs[d] = (p.tab && d < L) ? p.tab[(long)h * L + d] : 0.f;
  if constexpr (!SP) {
    for (int j = threadIdx.x; j < p.Tkb; j += 256)
      kb[j] = (j < T && !(p.kpm && p.kpm[(long)b * T + j])) ? 0.f : -INFINITY;
    if constexpr (DROP && !BCONS)
      for (int jp = threadIdx.x; jp < (p.Tkb >> 1); jp += 256) colw[jp] = fa_col_word(p.s1, (unsigned)jp);
  }
  for (int q = threadIdx.x; q < SKEW_BYTES / 16; q += 256) reinterpret_cast<uint4*>(smem + 32768)[q] = make_uint4(0, 0, 0, 0);
  const float g = p.gate ? p.gate[(long)bh * T + ic] : 0.f;
  const float g2 = g * FA_LOG2E;
  const float nlse2 = valid_i ? p.log2sc - p.lse[(long)bh * T + ic] * FA_LOG2E : -INFINITY;  // P * sc = 2^(x + nlse2)
#if FA_DQ_BIAS_IN_C
  const float g2s = g2 / p.sc2, nlse2s = nlse2 / p.sc2;
#endif
  const unsigned roww = fa_row_word(p.s0, (unsigned)(bh * T + ic));
  const float* trow = tabs + (T - 1 - ic);
  // skew write base of this lane: row (31 - rho + 4 hi), column rho (bf16)
  unsigned short* sk_w = reinterpret_cast<unsigned short*>(skew + (31 - ql + 4 * hi) * 64 + 2 * ql);
  // B-operand read of diagonal block cb: row 16 cb + (l & 15), 16 bytes at (l >> 4) * 16
  const unsigned char* sk_r = skew + (lane & 15) * 64 + (lane >> 4) * 16;
  // row-major form: this lane's row rho = ql starts its runs at diagonal 31 - rho + 4 hi (+ dd); transposing read of a 16-diagonal
  // block: lane i of a 16-lane group addresses row 8 kg + (i >> 2) (second read: + 4), diagonals 4 (i & 3) .. +3 and receives the four
  // rows' values of diagonal i
  const unsigned sk_w3 = (unsigned)(unsigned long)(las_ptr)(skew + ql * FA_SKEW_RB + 2 * (31 - ql + 4 * hi));
  const unsigned char* sk_r3 = skew + (8 * (lane >> 4) + ((lane & 15) >> 2)) * FA_SKEW_RB + 8 * (lane & 3);
  // transposed A-operand read base into a K tile (see the header comment)
  const unsigned ktr = fa_tr_base(lane);

  f32x16_t dq[2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[f][r] = 0.f;
  f32x4_t dacc[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) dacc[c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // A operand of the diagonal sums: A[m][rho] = gate of query row rho for every m (instead of ones), so that
  // sum_rho gate_rho dS[rho][.] needs no per-element multiply and no second bf16 conversion
  U4 gfrag; gfrag.v = make_uint4(0, 0, 0, 0);
  if constexpr (TAB) {
    unsigned short* gb = gball + wave_u * 32;
    if (hi == 0) gb[ql] = f2bf(g);
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): same-wave LDS write -> read
    __builtin_amdgcn_wave_barrier();
    gfrag.v = *reinterpret_cast<const uint4*>(gb + (lane >> 4) * 8);
  }
  float dg = 0.f;
#if FA_DQ_PK
  f32x2_t dg2 = f32x2_t{0.f, 0.f};
#endif
  const int nkv = (T + FA_BKV - 1) / FA_BKV;
  // dropout decisions of this lane's row as bit words for the 64-keys-per-wave dK/dV kernel (attn_fused.hpp: dbits): one
  // v_addc per element while the decision sits in VCC anyway; per tile two words (32-key blocks f = 0, 1), lanes 0..31 end up
  // with word 0 and lanes 32..63 with word 1 of the row: one store per tile, issued at the top of the next tile
  unsigned bacc0 = 0, bacc1 = 0, pend_bits = 0;
  int pend_bits_jt = -1;
  unsigned* dbp = nullptr;
  const unsigned thsv = (unsigned)p.ths;
  if constexpr (EMIT)
    if (p.dbits) dbp = p.dbits + ((long)bh * p.db_nkb + hi) * p.db_Tq + fa_bitrow(i);
  // BCONS: this row's words, block 0 of tile jt at bsrc[jt * 2 * db_Tq], block 1 one db_Tq further; nxt* = the next tile's,
  // requested at the top of the tile before (behind its K / V prefetch: the end-of-tile vmcnt(0) covers them)
  const unsigned* bsrc = nullptr;
  unsigned nxt0 = 0, nxt1 = 0;
  if constexpr (BCONS) {
    bsrc = p.dbits + (long)bh * p.db_nkb * p.db_Tq + fa_bitrow(i);
    nxt0 = bsrc[0]; nxt1 = bsrc[p.db_Tq];
  }

  // SP: fragments and running maxima travel TWO tiles ahead in two register sets (ring A / B, the tile loop is unrolled by
  // two so that no set is ever copied while its loads are in flight).  One tile ahead the kernel was latency-bound at
  // 2.1 TB/s of P traffic: 16 KB in flight per workgroup, a loaded HBM round trip of ~4 us per tile.  (Deeper does not
  // help as long as the K / V tiles arrive by DMA under the same in-order counter: the wait for the DMA of tile t + 1 at
  // the end of tile t also waits for every load issued before it.)
  const unsigned char* pld = nullptr;
  const float* mtp = nullptr;
  U4 pA0[2], pA1[2], pB0[2], pB1[2];
  float mtA = 0.f, mtB = 0.f;
  auto p_load = [&](int jn, U4 (&x0)[2], U4 (&x1)[2], float& mx) __attribute__((always_inline)) {
    const unsigned char* src = pld + (long)jn * FA_PTILE_BYTES;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      x0[s2].v = *reinterpret_cast<const uint4*>(src + s2 * 1024);
      x1[s2].v = *reinterpret_cast<const uint4*>(src + (2 + s2) * 1024);
    }
    mx = mtp[(long)jn * p.ps.Tq];
  };
  if constexpr (SP) {
    pld = p.ps.P16 + (((long)bh * p.ps.nq32 + (qblk * 4 + wave_u)) * p.ps.nkv) * FA_PTILE_BYTES + lane * 16;
    mtp = p.ps.mt + (long)bh * p.ps.nkv * p.ps.Tq + (qblk * FA_BQ + 32 * wave_u + ql);
    p_load(0, pA0, pA1, mtA);
    p_load(p.ps.nkv > 1 ? 1 : 0, pB0, pB1, mtB);
  }
#if FA_TILE_SRC
  FaTileSrc ksrc, vsrc;
  ksrc.init(base + D, D3, T);
  vsrc.init(base + 2 * D, D3, T);
#endif
  FA_LOAD_KV(0, kbuf(0), vbuf(0));   // (D == FA_HD * H: the same macro as the forward's)
  fa_tile_sync();

  float pend_v = 0.f;
  int pend_d = -1;
  // diagonal sums of a finished tile (first key j0p): six B-operand reads of the skew buffer, six small MFMAs; blocks
  // 4, 5 of the tile before carry into blocks 0, 1.  FA_DQ_DEFER: run for tile jt - 1 at the top of tile jt, so that
  // its LDS round trip and MFMA chain sit under this tile's score MFMAs instead of in front of the barrier (the skew
  // buffer is private to the wave and LDS operations of one wave execute in order: the reads precede this tile's
  // skew writes).
  auto skew_sums = [&](int j0p) __attribute__((always_inline)) {
    f32x4_t nacc[6];
#pragma unroll
    for (int cb = 0; cb < 6; ++cb) {
      U4 bfr;
#if FA_DQ_SKEW_ROWS
      {
        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_ptr)(sk_r3 + cb * 32));
        const bf16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_ptr)(sk_r3 + 4 * FA_SKEW_RB + cb * 32));
        bfr.b = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#else
      bfr.v = *reinterpret_cast<const uint4*>(sk_r + cb * 1024);
#endif
      const f32x4_t cin = cb < 2 ? dacc[cb + 4] : f32x4_t{0.f, 0.f, 0.f, 0.f};
      nacc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gfrag.b, bfr.b, cin, 0, 0, 0);
    }
#pragma unroll
    for (int cb = 0; cb < 6; ++cb) dacc[cb] = nacc[cb];
    const int gsel = lane >> 4;
    // (a chain of selects on scalars: written as one nested conditional over dacc[.][0] the compiler builds a
    // dynamically indexed array in scratch)
    const float a0 = dacc[0][0], a1 = dacc[1][0], a2 = dacc[2][0], a3 = dacc[3][0];
    float v = a0;
    v = gsel == 1 ? a1 : v;
    v = gsel == 2 ? a2 : v;
    v = gsel == 3 ? a3 : v;
    // the store is issued at the top of the NEXT tile: gfx950's vmcnt counts stores, so the s_waitcnt vmcnt(0) that
    // guards the K/V prefetch in front of the tile barrier would otherwise wait for this store's round trip
    // (measured neutral: the round trip is not what the relative-position part costs)
    pend_v = v;
    pend_d = dlo0 + j0p + lane;
  };
  auto flush_pending = [&]() __attribute__((always_inline)) {
    if (pend_d >= 0 && pend_d < L) prow[pend_d] = pend_v;
    pend_d = -1;
  };
  int cur = 0;
  // one key tile; (x0, x1, mx): SP, the register set that holds this tile's fragments and is refilled with tile jt + 2
  auto tile_body = [&](int jt, U4 (&x0)[2], U4 (&x1)[2], float& mx) __attribute__((always_inline)) {
    const int j0 = jt * FA_BKV;
    const bool more = jt + 1 < nkv;
    U4 pc0[2], pc1[2];
    float ct = 0.f;
    if constexpr (SP) {
      // (taken out of the set BEFORE the DMA is issued: the compiler's wait for these loads then cannot catch the DMA.  The
      // copies are asm: a plain assignment is coalesced away, the refill then lands in OTHER registers and the loop-carried
      // set is restored by moves at the bottom of the loop -- which wait for the refill that was just issued)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          asm volatile("v_mov_b32 %0, %1" : "=v"(pc0[s2].u[e]) : "v"(x0[s2].u[e]));
          asm volatile("v_mov_b32 %0, %1" : "=v"(pc1[s2].u[e]) : "v"(x1[s2].u[e]));
        }
      // P sc = |P16| 2^(mt + log2 sc - lse log2 e); rows past T: nlse2 = -inf -> 0; rows without a finite score: mt = -inf
      ct = mx > -INFINITY ? __builtin_amdgcn_exp2f(mx + nlse2) : 0.f;
    }
    if (more) {
      FA_LOAD_KV(j0 + FA_BKV, kbuf(cur ^ 1), vbuf(cur ^ 1));
    }
    if constexpr (SP && !(FA_SP_PROBE & 1)) p_load(jt + 2 < nkv ? jt + 2 : jt, x0, x1, mx);   // (past the end: a harmless reload, no branch around the loads)
    if constexpr (TAB) flush_pending();
    unsigned bw0 = 0, bw1 = 0;   // BCONS: this tile's words, the half-wave's own bits moved to positions 0..7 / 16..23
    if constexpr (BCONS) {
      bw0 = nxt0 >> (8 * (1 - hi)); bw1 = nxt1 >> (8 * (1 - hi));
      const int jn = more ? jt + 1 : jt;
      nxt0 = bsrc[(long)jn * 2 * p.db_Tq]; nxt1 = bsrc[((long)jn * 2 + 1) * p.db_Tq];
    }
    if constexpr (EMIT) {
      if (dbp && pend_bits_jt >= 0) dbp[(long)pend_bits_jt * 2 * p.db_Tq] = pend_bits;
      pend_bits_jt = -1;
    }
    if constexpr (TAB && FA_DQ_DEFER) {
      if (jt > 0) skew_sums(j0 - FA_BKV);
    }
    // bias-table entries and dropout column words of the whole tile, read before the score MFMAs are issued.  Only with the
    // relative-position table: without it the kernel needs 43 KB of LDS, and the 48 registers this costs are the
    // difference between two and three workgroups per CU (160 -> 139 us).
    constexpr bool PF = FA_DQ_PREFETCH && TAB;
    float tvv[PF ? 2 : 1][16];
    unsigned cww[PF ? 2 : 1][8];
    if constexpr (PF) {
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = j0 + 32 * f + (r & 3) + 8 * (r >> 2) + 4 * hi;
          tvv[f][r] = trow[j];
          if (DROP && !SP && !BCONS && !(r & 1)) cww[f][r >> 1] = (colw + ((j0 + 4 * hi) >> 1))[(32 * f + (r & 3) + 8 * (r >> 2)) >> 1];
        }
    }
    const bool edge = !SP && ((p.kpm != nullptr) || (j0 + FA_BKV > T));   // (SP: masked keys carry a stored 0)
    U4 dsf0[2], dsf1[2];  // (two arrays, not dsf[2][2]: the 2-D array of unions is not split into registers)
    f32x16_t s[2], dp[2];
    // (block index as a type: a run-time index into s / dp / dsf would put the arrays into scratch)
    using F0 = std::integral_constant<int, 0>;
    using F1 = std::integral_constant<int, 1>;
    auto scores = [&](auto fc) __attribute__((always_inline)) {
      constexpr int f = decltype(fc)::value;
#if FA_DQ_BIAS_IN_C
      // bias - lse (and the key mask of edge tiles) as the C operand of the score MFMAs, in units of 1 / sc2: the fma leaves
      // the dependent chain behind the MFMAs (P = 2^(sc2 s))
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = j0 + 32 * f + (r & 3) + 8 * (r >> 2) + 4 * hi;
        float tv;
        if constexpr (PF) tv = tvv[f][r]; else tv = trow[j];
        float c = fmaf(g2s, tv, nlse2s);
        if (edge) c += kb[j];
        s[f][r] = c; dp[f][r] = 0.f;
      }
#else
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[f][r] = 0.f; dp[f][r] = 0.f; }
#endif
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if constexpr (!SP)
          s[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain(kbuf(cur), 32 * f + ql, kk, hi), qf[kk].b, s[f], 0, 0, 0);
        dp[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain(vbuf(cur), 32 * f + ql, kk, hi), dof[kk].b, dp[f], 0, 0, 0);
      }
    };
    auto elem_pass = [&](auto fc, auto edge_c) __attribute__((always_inline)) {
      constexpr int f = decltype(fc)::value;
      constexpr bool EDGE = decltype(edge_c)::value;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float dv[2];
        unsigned w = 0;
        if constexpr (DROP && !SP && !BCONS) {
          if constexpr (PF) w = fa_mix(roww + cww[f][r >> 1]);
          else w = fa_mix(roww + (colw + ((j0 + 4 * hi) >> 1))[(32 * f + (r & 3) + 8 * (r >> 2)) >> 1]);
        }
        if constexpr (SP) {
          // stored pair -> signed P sc (sign = dropped); dS = keep P sc dP - P delta = max(Ps, 0) dP - |Ps| delta / sc
          const f16x2_t h2 = __builtin_bit_cast(f16x2_t, (f == 0 ? pc0 : pc1)[r >> 3].u[(r & 7) >> 1]);
          const f32x2_t ps2 = f32x2_t{(float)h2[0], (float)h2[1]} * f32x2_t{ct, ct};
          const f32x2_t pk2 = f32x2_t{fmaxf(ps2[0], 0.f), fmaxf(ps2[1], 0.f)};
          const f32x2_t pa2 = f32x2_t{__builtin_fabsf(ps2[0]), __builtin_fabsf(ps2[1])};
          const f32x2_t ds2 = __builtin_elementwise_fma(pk2, f32x2_t{dp[f][r], dp[f][r + 1]}, -(pa2 * f32x2_t{dls, dls}));
          dv[0] = ds2[0]; dv[1] = ds2[1];
          if constexpr (TAB) {
            f32x2_t tv2;
            if constexpr (PF) tv2 = f32x2_t{tvv[f][r], tvv[f][r + 1]};
            else { const int j = j0 + 32 * f + (r & 3) + 8 * (r >> 2) + 4 * hi; tv2 = f32x2_t{trow[j], trow[j + 1]}; }
#if FA_DQ_PK
            dg2 = __builtin_elementwise_fma(ds2, tv2, dg2);
#else
            dg = fmaf(ds2[0], tv2[0], fmaf(ds2[1], tv2[1], dg));
#endif
          }
        } else
#if FA_DQ_PK
        {
          const int j = j0 + 32 * f + (r & 3) + 8 * (r >> 2) + 4 * hi;  // r even: j, j + 1 are this pair's keys
          f32x2_t tv2;
          if constexpr (PF) tv2 = f32x2_t{tvv[f][r], tvv[f][r + 1]}; else tv2 = f32x2_t{trow[j], trow[j + 1]};
#if FA_DQ_BIAS_IN_C
          const f32x2_t x2 = f32x2_t{s[f][r], s[f][r + 1]} * f32x2_t{p.sc2, p.sc2};
#else
          f32x2_t x2 = __builtin_elementwise_fma(f32x2_t{s[f][r], s[f][r + 1]}, f32x2_t{p.sc2, p.sc2},
                                                 __builtin_elementwise_fma(f32x2_t{g2, g2}, tv2, f32x2_t{nlse2, nlse2}));
          if constexpr (EDGE) x2 += f32x2_t{kb[j], kb[j + 1]};
#endif
          const f32x2_t pe2 = f32x2_t{__builtin_amdgcn_exp2f(x2[0]), __builtin_amdgcn_exp2f(x2[1])};
          f32x2_t dp2 = f32x2_t{dp[f][r], dp[f][r + 1]};
          if constexpr (BCONS) {
            // register r = 2 t of the block sits at bit t, r + 1 at bit 16 + t of the shifted word: 0 / ~0 each, applied to dP
            const unsigned bw = f == 0 ? bw0 : bw1;
            const int m0 = __builtin_amdgcn_sbfe(bw, r >> 1, 1), m1 = __builtin_amdgcn_sbfe(bw, 16 + (r >> 1), 1);
            dp2[0] = __uint_as_float(__float_as_uint(dp2[0]) & (unsigned)m0);
            dp2[1] = __uint_as_float(__float_as_uint(dp2[1]) & (unsigned)m1);
            const f32x2_t ds2 = pe2 * (dp2 - f32x2_t{dls, dls});
            dv[0] = ds2[0]; dv[1] = ds2[1];
            dg2 = __builtin_elementwise_fma(ds2, tv2, dg2);
          } else if constexpr (DROP && !EMIT) {
            dp2[0] = fa_keep_lo(w, p.ths) ? dp2[0] : 0.f;
            dp2[1] = fa_keep_hi(w, p.ths) ? dp2[1] : 0.f;
            const f32x2_t ds2 = pe2 * (dp2 - f32x2_t{dls, dls});
            dv[0] = ds2[0]; dv[1] = ds2[1];
            dg2 = __builtin_elementwise_fma(ds2, tv2, dg2);
          } else if constexpr (DROP) {
            // keep decisions of both halves: compare (SDWA, sign-extended half against the threshold), select, and the
            // decision shifted into the block's bit accumulator through the carry.  The select works on dP - delta (kept) against
            // -delta (dropped) -- bit-identical to (kept ? dP : 0) - delta -- so that the first reader of the dP product's
            // registers is an instruction the compiler's hazard recogniser sees (it does not look into inline asm)
            const f32x2_t t2 = dp2 - f32x2_t{dls, dls};
            float d0 = t2[0], d1 = t2[1];
            unsigned acc = f == 0 ? bacc0 : bacc1;
            // (three separate statements per half, the decision in an SGPR pair instead of VCC: one VCC chain through six
            // instructions kept the scheduler from interleaving anything -- measured +10 us per launch)
            unsigned long k0, k1, cdummy;
            asm("v_cmp_le_i32_sdwa %[k], %[ths], sext(%[w]) src0_sel:DWORD src1_sel:WORD_0" : [k] "=s"(k0) : [ths] "v"(thsv), [w] "v"(w));
            asm("v_cmp_le_i32_sdwa %[k], %[ths], sext(%[w]) src0_sel:DWORD src1_sel:WORD_1" : [k] "=s"(k1) : [ths] "v"(thsv), [w] "v"(w));
            asm("v_cndmask_b32_e64 %[d], %[nd], %[d], %[k]" : [d] "+v"(d0) : [nd] "v"(ndls), [k] "s"(k0));
            asm("v_cndmask_b32_e64 %[d], %[nd], %[d], %[k]" : [d] "+v"(d1) : [nd] "v"(ndls), [k] "s"(k1));
            asm("v_addc_co_u32_e64 %[acc], %[c], %[acc], %[acc], %[k]" : [acc] "+v"(acc), [c] "=s"(cdummy) : [k] "s"(k0));
            asm("v_addc_co_u32_e64 %[acc], %[c], %[acc], %[acc], %[k]" : [acc] "+v"(acc), [c] "=s"(cdummy) : [k] "s"(k1));
            if constexpr (f == 0) bacc0 = acc; else bacc1 = acc;
            const f32x2_t ds2m = pe2 * f32x2_t{d0, d1};
            dv[0] = ds2m[0]; dv[1] = ds2m[1];
            dg2 = __builtin_elementwise_fma(ds2m, tv2, dg2);
          } else {
          const f32x2_t ds2 = pe2 * (dp2 - f32x2_t{dls, dls});
          dv[0] = ds2[0]; dv[1] = ds2[1];
          dg2 = __builtin_elementwise_fma(ds2, tv2, dg2);
          }
        }
#else
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int rr = r + e;
          const int j = j0 + 32 * f + (rr & 3) + 8 * (rr >> 2) + 4 * hi;
          float tv;
          if constexpr (PF) tv = tvv[f][rr]; else tv = trow[j];
#if FA_DQ_BIAS_IN_C
          const float x = s[f][rr] * p.sc2;
#else
          float x = fmaf(s[f][rr], p.sc2, fmaf(g2, tv, nlse2));  // x - lse as two fmas (the -lse rides in the bias term)
          if constexpr (EDGE) x += kb[j];
#endif
          const float pe = __builtin_amdgcn_exp2f(x);  // 0 for masked keys (-inf) and rows past T (nlse2 = -inf)
          float dpe = dp[f][rr];
          if constexpr (DROP) dpe = (e ? fa_keep_hi(w, p.ths) : fa_keep_lo(w, p.ths)) ? dpe : 0.f;
          const float ds = pe * (dpe - dls);
          dv[e] = ds;
          dg = fmaf(ds, tv, dg);
        }
#endif
        const unsigned u2 = pack_bf16(dv[0], dv[1]);
        if constexpr (f == 0) dsf0[r >> 3].u[(r & 7) >> 1] = u2; else dsf1[r >> 3].u[(r & 7) >> 1] = u2;
        if constexpr (TAB) {  // the skew buffer takes dS itself; the gate multiplies inside the diagonal-sum MFMA
          const int dd = 32 * f + (r & 3) + 8 * (r >> 2);  // + 4 hi is in sk_w
#if !FA_DQ_NOSKEW && !(FA_SP_PROBE & 4)  // (probe: wrong d(rel), prices the skew writes)
#if FA_DQ_SKEW_ROWS
          if (r & 2) {   // the run's second pair: keys dd - 2 .. dd + 1 = four consecutive diagonals of this lane's row
            const unsigned prev = (f == 0 ? dsf0 : dsf1)[r >> 3].u[((r & 7) >> 1) - 1];
            const unsigned long pair = ((unsigned long)u2 << 32) | prev;
            const unsigned a = sk_w3 + 2 * (dd - 2);
            asm volatile("ds_write_b64 %0, %1" :: "v"(a), "v"(pair) : "memory");
          }
#else
          sk_w[dd * 32] = (unsigned short)u2;
          sk_w[(dd + 1) * 32] = (unsigned short)(u2 >> 16);
#endif
#endif
        }
      }
    };
    // dQ^T += K^T dS^T (one 32-key block); K^T fragments by transposing reads of the K tile
    auto dq_acc = [&](auto fc) __attribute__((always_inline)) {
      constexpr int f = decltype(fc)::value;
#pragma unroll
      for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
          dq[f2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(kbuf(cur), ktr, f2, f, s2), (f == 0 ? dsf0[s2] : dsf1[s2]).b, dq[f2], 0, 0, 0);
    };
#if FA_DQ_BOTH
    // both 32-key blocks' score MFMAs first: 16 independent MFMAs in flight, then 32 elements of VALU, then 8 MFMAs
    scores(F0{}); scores(F1{});
    if (edge) { elem_pass(F0{}, std::true_type{}); elem_pass(F1{}, std::true_type{}); }
    else { elem_pass(F0{}, std::false_type{}); elem_pass(F1{}, std::false_type{}); }
    dq_acc(F0{}); dq_acc(F1{});
#else
    scores(F0{});
    if (edge) elem_pass(F0{}, std::true_type{}); else elem_pass(F0{}, std::false_type{});
    dq_acc(F0{});
    scores(F1{});
    if (edge) elem_pass(F1{}, std::true_type{}); else elem_pass(F1{}, std::false_type{});
    dq_acc(F1{});
#endif
    if constexpr (EMIT) {
      // 16 decisions per block, first element in bit 15: reverse, spread the four 4-key runs to their key positions
      // (r -> (r & 3) + 8 (r >> 2) + 4 hi), exchange the halves' partial words (one swap: lanes 0..31 receive word 0 of both
      // half-waves, lanes 32..63 word 1)
      auto spread = [&](unsigned a) __attribute__((always_inline)) {
        unsigned x = __builtin_bitreverse32(a) >> 16;
        x = (x | (x << 8)) & 0x00FF00FFu;
        x = (x | (x << 4)) & 0x0F0F0F0Fu;
        return x << (4 * hi);
      };
      unsigned w0 = spread(bacc0), w1 = spread(bacc1);
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(w0), "+v"(w1));
      pend_bits = w0 | w1;
      pend_bits_jt = jt;
      bacc0 = 0; bacc1 = 0;
    }
    if constexpr (TAB && !FA_DQ_DEFER) skew_sums(j0);
    // SP: the five loads of p_load were issued behind the DMA and may stay in flight across the barrier
    fa_tile_sync<SP ? 5 : 0>();
    cur ^= 1;
  };
  if constexpr (SP) {
    // whole pairs, then the odd tile: the pair loop's only back edge is B -> A (with a conditional tile B inside one loop the
    // compiler's control-flow graph keeps a path from tile A straight back to tile A, and its wait-count pass then waits
    // for A's refill at the top of A: loads issued one tile ago instead of two)
    for (int pi = 0; pi < (nkv >> 1); ++pi) {
      tile_body(2 * pi, pA0, pA1, mtA);
      tile_body(2 * pi + 1, pB0, pB1, mtB);
    }
    if (nkv & 1) tile_body(nkv - 1, pA0, pA1, mtA);
  } else {
    for (int jt = 0; jt < nkv; ++jt) tile_body(jt, pA0, pA1, mtA);
  }
  if constexpr (TAB && FA_DQ_DEFER) skew_sums((nkv - 1) * FA_BKV);
  if constexpr (TAB) flush_pending();
  if constexpr (EMIT)
    if (dbp && pend_bits_jt >= 0) dbp[(long)pend_bits_jt * 2 * p.db_Tq] = pend_bits;
  if (TAB && lane < 32) {
    // the 31 diagonals past the last tile's first 64 (blocks 4, 5)
    const float v = lane < 16 ? dacc[4][0] : dacc[5][0];
    const int d = dlo0 + 64 * nkv + lane;
    if (d >= 0 && d < L) prow[d] = v;
  }
#if FA_DQ_PK
  dg = dg2[0] + dg2[1];
#endif
  dg = wl_sum_xor32(dg);
  if (p.dbias_part)  // rows past T carry dS = 0, i.e. dq = 0
    fa_wave_colsum(dq, p.scale, reinterpret_cast<float*>(smem),
                   p.dbias_part + (((long)b * p.nqb + qblk) * 4 + wave_u) * D3 + h * FA_HD, lane, wave_u);
  if (valid_i) {
    bf16_t* dst = p.dqkv + ((long)b * T + i) * D3 + h * FA_HD;
#pragma unroll
    for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        uint2 w;
        w.x = pack_bf16(dq[f2][4 * q4] * p.scale, dq[f2][4 * q4 + 1] * p.scale);
        w.y = pack_bf16(dq[f2][4 * q4 + 2] * p.scale, dq[f2][4 * q4 + 3] * p.scale);
        *reinterpret_cast<uint2*>(dst + 32 * f2 + 8 * q4 + 4 * hi) = w;
      }
    if (hi == 0) {
      p.delta[(long)bh * T + i] = dl;
      if (p.dgate) p.dgate[(long)bh * T + i] = dg;
    }
  }
}

// drel[h][d] = sum over (b, q-tile, wave) partial rows (row stride Lp = L rounded up to 4 floats): block = 16 groups of
// 4 diagonals x 64 row slices.  Row (qblk, w) holds the diagonals [lo, lo + 64 nkv + 32) with
// lo = T - 1 - (128 qblk + 32 w) - 31 (attn_bwd_dq_kernel); the rest of it was never written (garbage: select, never
// multiply).  The kernel is pure load latency: four independent 16-byte loads in flight per thread.
__global__ __launch_bounds__(1024) void fa_dtab_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                              int B, int H, int nqt, int L, int Lp, int T, int nkv, int accumulate) {
  __shared__ float4 red[64][17];
  const int col = threadIdx.x & 15, slice = threadIdx.x >> 4;
  const int d = (blockIdx.x * 16 + col) * 4;
  const int h = blockIdx.y;
  const int nrow = nqt * 4, ntot = B * nrow, span = 64 * nkv + 32;
  float4 acc[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (d < Lp) {
    for (int r0 = slice; r0 < ntot; r0 += 256) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = r0 + 64 * u;
        if (r < ntot) {
          const int b = r / nrow, c = r - b * nrow;
          const int k0 = d - (T - 1 - 32 * c - 31);  // position of d in the row's written range
          if (k0 > -4 && k0 < span) {
            const float4 v = *reinterpret_cast<const float4*>(part + (((long)b * H + h) * nrow + c) * Lp + d);
            acc[u].x += (k0 >= 0 && k0 < span) ? v.x : 0.f;
            acc[u].y += (k0 + 1 >= 0 && k0 + 1 < span) ? v.y : 0.f;
            acc[u].z += (k0 + 2 >= 0 && k0 + 2 < span) ? v.z : 0.f;
            acc[u].w += (k0 + 3 >= 0 && k0 + 3 < span) ? v.w : 0.f;
          }
        }
      }
    }
  }
  float4 t;
  t.x = (acc[0].x + acc[1].x) + (acc[2].x + acc[3].x); t.y = (acc[0].y + acc[1].y) + (acc[2].y + acc[3].y);
  t.z = (acc[0].z + acc[1].z) + (acc[2].z + acc[3].z); t.w = (acc[0].w + acc[1].w) + (acc[2].w + acc[3].w);
  red[slice][col] = t;
  __syncthreads();
  if (slice < 4) {  // 64 -> 4 partial sums per diagonal group
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 16; ++k) { const float4 v = red[slice * 16 + k][col]; q.x += v.x; q.y += v.y; q.z += v.z; q.w += v.w; }
    red[slice * 16][col] = q;
  }
  __syncthreads();
  if (slice == 0 && d < L) {
    const float4 a0 = red[0][col], a1 = red[16][col], a2 = red[32][col], a3 = red[48][col];
    const float o[4] = {(a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                        (a0.w + a1.w) + (a2.w + a3.w)};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (d + k < L) out[(long)h * L + d + k] = accumulate ? out[(long)h * L + d + k] + o[k] : o[k];
  }
}

static FaP fa_params(int B, int H, int T, float scale, float p_drop, uint64_t seed) {
  FaP p;
  p.B = B; p.H = H; p.T = T; p.scale = scale;
  p.sc2 = scale * FA_LOG2E;
  double tt = (double)p_drop * 65536.0 + 0.5; if (tt > 65535.0) tt = 65535.0;
  p.th = p_drop > 0.f ? (unsigned)tt : 0u;
  if (p_drop > 0.f && p.th == 0u) p.th = 1u;
  p.sc = p.th ? (float)(1.0 / (1.0 - (double)p.th / 65536.0)) : 1.f;  // unbiased for the quantised probability
  p.log2sc = log2f(p.sc); p.inv_sc = 1.f / p.sc;
  p.ths = (int)p.th - 32768;
  p.k2 = (unsigned)((p.ths - 1) & 0xffff) * 0x10001u;
  p.k3 = (unsigned)(p.ths & 0xffff) * 0x10001u;
  p.ps.P16 = nullptr; p.ps.mt = nullptr;
  p.dbits = nullptr; p.db_fwd = 0; p.db_nkb = ((T + FA_BKV - 1) / FA_BKV) * 2; p.db_Tq = ((T + FA_BQ - 1) / FA_BQ) * FA_BQ;
  p.ps.nq32 = ((T + FA_BQ - 1) / FA_BQ) * 4; p.ps.nkv = (T + FA_BKV - 1) / FA_BKV; p.ps.Tq = ((T + FA_BQ - 1) / FA_BQ) * FA_BQ;
  p.s0 = (unsigned)seed; p.s1 = (unsigned)(seed >> 32);
  const int nkv = (T + FA_BKV - 1) / FA_BKV;
  p.Tkb = nkv * FA_BKV;
  p.Ltab = (T + p.Tkb + 3) & ~3;  // index j - i + T - 1 with j < Tkb, i >= 0
  p.qkv = nullptr; p.O = nullptr; p.lse = nullptr; p.gate = nullptr; p.tab = nullptr; p.kpm = nullptr;
  p.dO = nullptr; p.dqkv = nullptr; p.delta = nullptr; p.dgate = nullptr; p.dtab_part = nullptr; p.dbias_part = nullptr;
  return p;
}

extern "C" {

// O[B,T,H*64] = softmax(scale QK^T + gate*rel + keypad) V from packed qkv [B,T,3*H*64] (bf16); lse[B*H,T] saved
uint64_t wavlm_attn_fused_pstore_bytes(int32_t B, int32_t H, int32_t T) {
  if (B <= 0 || H <= 0 || T <= 0) return 0;
  if (T > FA_PSTORE_MAX_T) return 0;   // longer sequences: recompute only (the dK/dV kernel keeps per-row scalars of all T rows in LDS)
  return ((fa_pstore_p_bytes(B, H, T) + 255) & ~(uint64_t)255) + fa_pstore_mt_bytes(B, H, T);
}

// bytes of the dropout bit words (the `pstore` of the bit mode of wavlm_attn_fused_fwd_p / _bwd_p): one bit per (row, key) of the
// padded grid, 27 MB per layer at B = 32, H = 12, T = 749
uint64_t wavlm_attn_fused_dbits_bytes(int32_t B, int32_t H, int32_t T) {
  if (B <= 0 || H <= 0 || T <= 0) return 0;
  return fa_dbits_bytes(B, H, T);
}

int wavlm_attn_fused_fwd(const void* qkv, void* O, float* lse, const float* gate, const float* tab, const uint8_t* kpm,
                         int32_t B, int32_t H, int32_t T, int32_t head_dim, float scale, float p_drop, uint64_t seed,
                         void* stream) {
  return wavlm_attn_fused_fwd_p(qkv, O, lse, gate, tab, kpm, nullptr, 0, B, H, T, head_dim, scale, p_drop, seed, stream);
}

static void fa_bind_pstore(FaP& p, void* pstore, int B, int H, int T) {
  p.ps.P16 = (unsigned char*)pstore;
  p.ps.mt = (float*)((unsigned char*)pstore + ((fa_pstore_p_bytes(B, H, T) + 255) & ~(uint64_t)255));
}

int wavlm_attn_fused_fwd_p(const void* qkv, void* O, float* lse, const float* gate, const float* tab, const uint8_t* kpm,
                           void* pstore, uint64_t pstore_bytes, int32_t B, int32_t H, int32_t T, int32_t head_dim, float scale,
                           float p_drop, uint64_t seed, void* stream) {
  if (!qkv || !O || !lse || B <= 0 || H <= 0 || T <= 0 || head_dim != FA_HD) return WL_EINVAL;
  if ((gate == nullptr) != (tab == nullptr)) return WL_EINVAL;
  // pstore_bytes == wavlm_attn_fused_dbits_bytes: the BIT mode (the forward stores its dropout decisions only)
  const bool bitmode = pstore && pstore_bytes == fa_dbits_bytes(B, H, T);
  if (bitmode) { if ((uintptr_t)pstore & 255) return WL_EINVAL; }
  else if (pstore && (T > FA_PSTORE_MAX_T || pstore_bytes < wavlm_attn_fused_pstore_bytes(B, H, T) || ((uintptr_t)pstore & 15))) return WL_EINVAL;
  FaP p = fa_params(B, H, T, scale, p_drop, seed);
  p.qkv = (const bf16_t*)qkv; p.O = (bf16_t*)O; p.lse = lse; p.gate = gate; p.tab = tab; p.kpm = kpm;
  if (bitmode) { p.dbits = (unsigned*)pstore; p.db_fwd = 1; pstore = nullptr; }
  if (pstore) fa_bind_pstore(p, pstore, B, H, T);
  const size_t smem = 32768 + (size_t)(p.Ltab + p.Tkb + p.Tkb / 2) * sizeof(float);
  p.nqb = (T + FA_BQ - 1) / FA_BQ;
  const dim3 grid((unsigned)(p.nqb * B * H));
  // algorithmic: QK^T and PV (2 x 2 T^2 hd per head); qkv read once, O written once, lse
  WlProfScope prof(WL_PROF_ATTN_FWD, WL_BF16, 4.0 * B * H * (double)T * T * FA_HD,
                   (double)B * T * H * FA_HD * 2.0 * 4.0 + (double)B * H * T * 4.0, (hipStream_t)stream);
#define FA_FWD(DR, ST) do { if (fa_set_smem(attn_fwd_kernel<DR, ST>, smem) != WL_OK) return WL_ELAUNCH; \
    WL_LAUNCH((attn_fwd_kernel<DR, ST>), grid, dim3(256), smem, (hipStream_t)stream, p); } while (0)
  if (p.th && p.dbits) {
    if (fa_set_smem(attn_fwd_kernel<true, false, true>, smem) != WL_OK) return WL_ELAUNCH;
    WL_LAUNCH((attn_fwd_kernel<true, false, true>), grid, dim3(256), smem, (hipStream_t)stream, p);
  } else
  if (p.th) { if (pstore) FA_FWD(true, true); else FA_FWD(true, false); }
  else { if (pstore) FA_FWD(false, true); else FA_FWD(false, false); }
#undef FA_FWD
  return wl_check_launch();
}

uint64_t wavlm_attn_fused_bwd_workspace_bytes(int32_t B, int32_t H, int32_t T) {
  const uint64_t nqt = (uint64_t)((T + FA_BQ - 1) / FA_BQ);
  const uint64_t Lp = (2 * (uint64_t)T - 1 + 3) & ~(uint64_t)3;
  // per-wave d(rel) rows + delta + per-wave column sums of dq | dk | dv
  // + the dropout bit words the dQ kernel leaves for the 64-keys-per-wave dK/dV kernel
  const uint64_t f = ((uint64_t)B * H * nqt * 4 * Lp + (uint64_t)B * H * T + (uint64_t)B * nqt * 4 * 3 * H * FA_HD) * sizeof(float);
  return ((f + 255) & ~(uint64_t)255) + fa_dbits_bytes(B, H, T);
}

// dqkv[B,T,3*H*64], dgate[B,H,T], dtab[H,2T-1] from dO and the forward's (qkv, O, lse)
int wavlm_attn_fused_bwd(const void* qkv, const void* O, const void* dO, const float* lse, const float* gate,
                         const float* tab, const uint8_t* kpm, void* dqkv, float* dgate, float* dtab, void* dbias,
                         int32_t dbias_dtype, int32_t dbias_accumulate, int32_t B, int32_t H, int32_t T, int32_t head_dim,
                         float scale, float p_drop, uint64_t seed, void* workspace, uint64_t ws_bytes, void* stream) {
  return wl_attn_fused_bwd_ex(qkv, O, dO, lse, gate, tab, kpm, nullptr, 0, dqkv, dgate, dtab, 0, dbias, dbias_dtype, dbias_accumulate,
                              B, H, T, head_dim, scale, p_drop, seed, workspace, ws_bytes, stream);
}

int wavlm_attn_fused_bwd_p(const void* qkv, const void* O, const void* dO, const float* lse, const float* gate,
                           const float* tab, const uint8_t* kpm, const void* pstore, uint64_t pstore_bytes, void* dqkv,
                           float* dgate, float* dtab, void* dbias, int32_t dbias_dtype, int32_t dbias_accumulate, int32_t B,
                           int32_t H, int32_t T, int32_t head_dim, float scale, float p_drop, uint64_t seed, void* workspace,
                           uint64_t ws_bytes, void* stream) {
  if (pstore && pstore_bytes == fa_dbits_bytes(B, H, T)) { if ((uintptr_t)pstore & 255) return WL_EINVAL; }
  else if (pstore && (T > FA_PSTORE_MAX_T || pstore_bytes < wavlm_attn_fused_pstore_bytes(B, H, T) || ((uintptr_t)pstore & 15))) return WL_EINVAL;
  return wl_attn_fused_bwd_ex(qkv, O, dO, lse, gate, tab, kpm, pstore, pstore_bytes, dqkv, dgate, dtab, 0, dbias, dbias_dtype, dbias_accumulate,
                              B, H, T, head_dim, scale, p_drop, seed, workspace, ws_bytes, stream);
}

}  // extern "C"

// the same with dtab (+)= (layer.hip: the table is shared by every block of the encoder, their gradients meet in one buffer)
int wl_attn_fused_bwd_ex(const void* qkv, const void* O, const void* dO, const float* lse, const float* gate,
                         const float* tab, const uint8_t* kpm, const void* pstore, uint64_t pstore_bytes, void* dqkv, float* dgate,
                         float* dtab, int dtab_accumulate,
                         void* dbias, int32_t dbias_dtype, int32_t dbias_accumulate, int32_t B, int32_t H, int32_t T,
                         int32_t head_dim, float scale, float p_drop, uint64_t seed, void* workspace, uint64_t ws_bytes,
                         void* stream) {
  if (!qkv || !O || !dO || !lse || !dqkv || !workspace || B <= 0 || H <= 0 || T <= 0 || head_dim != FA_HD)
    return WL_EINVAL;
  if ((gate == nullptr) != (tab == nullptr)) return WL_EINVAL;
  if (tab && (!dgate || !dtab)) return WL_EINVAL;
  if (ws_bytes < wavlm_attn_fused_bwd_workspace_bytes(B, H, T)) return WL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  FaP p = fa_params(B, H, T, scale, p_drop, seed);
  p.qkv = (const bf16_t*)qkv; p.O = (bf16_t*)const_cast<void*>(O); p.lse = const_cast<float*>(lse);
  p.gate = gate; p.tab = tab; p.kpm = kpm; p.dO = (const bf16_t*)dO; p.dqkv = (bf16_t*)dqkv; p.dgate = dgate;
  // pstore of the BIT mode: the forward's dropout decisions; everything else is recomputed
  const unsigned* fwd_bits = nullptr;
  if (pstore && pstore_bytes == fa_dbits_bytes(B, H, T)) { fwd_bits = (const unsigned*)pstore; pstore = nullptr; }
  if (pstore) fa_bind_pstore(p, const_cast<void*>(pstore), B, H, T);
  const int nqt = (T + FA_BQ - 1) / FA_BQ;
  const int L = 2 * T - 1;
  p.dtab_part = (float*)workspace;
  const int Lp = (L + 3) & ~3;
  p.delta = p.dtab_part + (long)B * H * nqt * 4 * Lp;
  if (dbias) p.dbias_part = p.delta + (long)B * H * T;
  // Round 6, WAVLM_ATTN_DBITS=1: the dQ kernel leaves its dropout decisions behind as bit words and the dK/dV kernel selects with
  // them.  Built, bit-identical, and OFF: the dK/dV kernel gains 10 us per launch and the dQ kernel pays 10 for producing the
  // words (same-box A/B of three alternating runs each, profiles/r06/ab_attn_dq_emit_bits.txt) -- every kernel recomputes.  Not with stored probabilities (the decision is the
  // stored sign there).  WAVLM_ATTN_DKV64=1: the 64-keys-per-wave dK/dV kernel (attn_fused_dkv64.hip: built, bit-identical,
  // measured SLOWER -- one wave per SIMD is issue-bound --, kept for the record).
  static const bool dbits_on = []() { const char* e = getenv("WAVLM_ATTN_DBITS"); return e && e[0] == '1'; }();
  static const bool dkv64_on = []() { const char* e = getenv("WAVLM_ATTN_DKV64"); return e && e[0] == '1'; }();
  bool use64 = dkv64_on && dbits_on && !pstore;
  if (use64) {
    FaP q = p; q.nqb = (T + FA_K64 - 1) / FA_K64;
    if (fa_dkv64_smem(q) > 160 * 1024) use64 = false;
  }
  if (fwd_bits && p.th) {
    p.dbits = const_cast<unsigned*>(fwd_bits); p.db_fwd = 1; use64 = false;   // (the 64-keys-per-wave kernel reads lane masks: dQ-kernel words only)
  } else
  if (dbits_on && !pstore && p.th) {
    const uint64_t fbytes = ((uint64_t)B * H * nqt * 4 * Lp + (uint64_t)B * H * T + (uint64_t)B * nqt * 4 * 3 * H * FA_HD) * sizeof(float);
    p.dbits = (unsigned*)((unsigned char*)workspace + ((fbytes + 255) & ~(uint64_t)255));
  }
  size_t smem1 = 32768 + (tab ? 4 * FA_SKEW_WAVE : 0) + (size_t)(p.Ltab + p.Tkb + p.Tkb / 2) * sizeof(float) + 256;  // + gate fragments
  if (smem1 < FA_CS_FLOATS * sizeof(float)) smem1 = FA_CS_FLOATS * sizeof(float);
  p.nqb = nqt;
  // algorithmic: S, dP, dQ, dK, dV (5 x 2 T^2 hd per head; the kernels recompute S and dP once more); qkv, O, dO read
  // once, dqkv written once
  WlProfScope prof(WL_PROF_ATTN_BWD, WL_BF16, 10.0 * B * H * (double)T * T * FA_HD,
                   (double)B * T * H * FA_HD * 2.0 * 8.0 + (double)B * H * T * 8.0, st);
#define FA_DQ(DR, TB, SP_) do { if (fa_set_smem(attn_bwd_dq_kernel<DR, TB, SP_>, smem1) != WL_OK) return WL_ELAUNCH; \
    WL_LAUNCH((attn_bwd_dq_kernel<DR, TB, SP_>), dim3((unsigned)(nqt * B * H)), dim3(256), smem1, st, p); } while (0)
#define FA_DQB(TB) do { if (fa_set_smem(attn_bwd_dq_kernel<true, TB, false, true>, smem1) != WL_OK) return WL_ELAUNCH; \
    WL_LAUNCH((attn_bwd_dq_kernel<true, TB, false, true>), dim3((unsigned)(nqt * B * H)), dim3(256), smem1, st, p); } while (0)
  if (p.th && p.db_fwd) {   // the forward's bit words: nothing to hash, nothing to produce
    if (tab) FA_DQB(true); else FA_DQB(false);
  } else
  if (pstore) {   // stored probabilities: the dropout decision is the stored sign
    if (tab) FA_DQ(false, true, true); else FA_DQ(false, false, true);
  } else if (p.th && p.dbits) {   // emit the bit words for the dK/dV kernel
    if (tab) { if (fa_set_smem(attn_bwd_dq_kernel<true, true, false, false, true>, smem1) != WL_OK) return WL_ELAUNCH;
               WL_LAUNCH((attn_bwd_dq_kernel<true, true, false, false, true>), dim3((unsigned)(nqt * B * H)), dim3(256), smem1, st, p); }
    else { if (fa_set_smem(attn_bwd_dq_kernel<true, false, false, false, true>, smem1) != WL_OK) return WL_ELAUNCH;
           WL_LAUNCH((attn_bwd_dq_kernel<true, false, false, false, true>), dim3((unsigned)(nqt * B * H)), dim3(256), smem1, st, p); }
  } else if (p.th) {
    if (tab) FA_DQ(true, true, false); else FA_DQ(true, false, false);
  } else {
    if (tab) FA_DQ(false, true, false); else FA_DQ(false, false, false);
  }
#undef FA_DQ
#undef FA_DQB
  size_t smem2 = 32768 + (size_t)(p.Ltab + 64) * sizeof(float) + 2 * 4096;   // (2 x FA_RSM_STAGE of attn_fused_dkv.hip, >= 2 x FA_ROWV floats)
  if (smem2 < FA_CS_FLOATS * sizeof(float)) smem2 = FA_CS_FLOATS * sizeof(float);
  if (use64) {
    p.nqb = (T + FA_K64 - 1) / FA_K64;
    const int rc = fa_launch_dkv64(p, (unsigned)(p.nqb * B * H), st);
    if (rc != WL_OK) return rc;
  } else {
    p.nqb = (T + FA_BK1 - 1) / FA_BK1;
    const dim3 grid2((unsigned)(p.nqb * B * H));
    const int rc = fa_launch_dkv(p, grid2.x, smem2, st);
    if (rc != WL_OK) return rc;
  }
  if (dbias) {  // [B * nqt * 4] partial rows x 3 * H * 64 columns -> the bias gradient ((+)= in its own dtype)
    const int rcb = wl_colsum_finish(p.dbias_part, B * nqt * 4, 3 * H * FA_HD, dbias, dbias_dtype, dbias_accumulate, st);
    if (rcb != WL_OK) return rcb;
    if (dbias_accumulate && !wl_fin_active()) wl_notify_grad(dbias, (uint64_t)3 * H * FA_HD * wl_esize(dbias_dtype), stream);
  }
  if (tab)
    WL_LAUNCH(fa_dtab_reduce_kernel, dim3((unsigned)((Lp / 4 + 15) / 16), (unsigned)H), dim3(1024), 0, st,
              (const float*)p.dtab_part, dtab, (int)B, (int)H, nqt, L, Lp, (int)T, (T + FA_BKV - 1) / FA_BKV, (int)dtab_accumulate);
  return wl_check_launch();
}
