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
template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(FaP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // per stage: Q [q][hd] | dO [q][hd], 8 KB each; the transposed operands of the query contraction are read from
  // the same tiles with ds_read_b64_tr_b16 (frag_tr)
  auto qbuf = [&](int st) { return smem + st * 16384; };
  auto dobuf = [&](int st) { return smem + st * 16384 + 8192; };
  float* tabs = reinterpret_cast<float*>(smem + 32768);
  float* rowv = tabs + p.Ltab + 64;  // [2 stages][4][64]: lse * log2e, delta, gate * log2e, dropout row word of the query tile
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
  // a padded / out-of-range key is NOT masked inside the loop (one add per element): its column only feeds this lane's
  // own dK / dV rows, which are written as zeros at the end

  f32x16_t dk[2], dv[2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[f][r] = 0.f; dv[f][r] = 0.f; }
  const int nq = (T + FA_BQ1 - 1) / FA_BQ1;
  const unsigned qtr = fa_tr_base(lane);

  auto stage_rows = [&](int it, int st) {
    const int t = threadIdx.x;
    if (t < 64) {
      const int ii = it * FA_BQ1 + t;
      const bool ok = ii < T;
      const long o = (long)bh * T + (ok ? ii : T - 1);
#if FA_DKV_BIAS_IN_C
      // bias - lse enters as the C operand of the score MFMAs, in units of 1 / sc2 (P * sc = 2^(sc2 s))
      rowv[st * 256 + t] = ok ? (p.log2sc - p.lse[o] * FA_LOG2E) / p.sc2 : -INFINITY;
      rowv[st * 256 + 128 + t] = p.gate ? p.gate[o] * FA_LOG2E / p.sc2 : 0.f;
#else
      rowv[st * 256 + t] = ok ? p.log2sc - p.lse[o] * FA_LOG2E : -INFINITY;  // P * sc = 2^(x + this); -inf: rows past T
      rowv[st * 256 + 128 + t] = p.gate ? p.gate[o] * FA_LOG2E : 0.f;
#endif
      rowv[st * 256 + 64 + t] = p.delta[o] * p.inv_sc;
      rowv[st * 256 + 192 + t] = __uint_as_float(fa_row_word(p.s0, (unsigned)o));
    }
  };

  glds_tile64(base, D3, 0, T, qbuf(0), wave_u);
  glds_tile64(dobase, D, 0, T, dobuf(0), wave_u);
  stage_rows(0, 0);
  __syncthreads();

  int cur = 0;
  for (int it = 0; it < nq; ++it) {
    const int iq0 = it * FA_BQ1;
    const bool more = it + 1 < nq;
    if (more) {
      glds_tile64(base, D3, iq0 + FA_BQ1, T, qbuf(cur ^ 1), wave_u);
      glds_tile64(dobase, D, iq0 + FA_BQ1, T, dobuf(cur ^ 1), wave_u);
      stage_rows(it + 1, cur ^ 1);
    }
    const float* rv = rowv + cur * 256;
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
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        // registers 4 q4 .. 4 q4 + 3 of a block are four consecutive query rows: per-row scalars come as 16-byte
        // LDS vectors; the Toeplitz entries rel[j - i] run downwards in i
        const int il0 = 32 * f + 8 * q4 + 4 * hi;
        const float4 del4 = *reinterpret_cast<const float4*>(rv + 64 + il0);
        const uint4 row4 = *reinterpret_cast<const uint4*>(rv + 192 + il0);
        const float delv[4] = {del4.x, del4.y, del4.z, del4.w};
        const unsigned roww[4] = {row4.x, row4.y, row4.z, row4.w};
#if !FA_DKV_BIAS_IN_C
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
#if FA_DKV_BIAS_IN_C
          const float pe = __builtin_amdgcn_exp2f(s[rr] * p.sc2);  // rows past T: -inf -> 0
#else
          const float pe = __builtin_amdgcn_exp2f(fmaf(s[rr], p.sc2, fmaf(gatv[e], tq[-e], lsev[e])));  // rows past T: -inf -> 0
#endif
          float pd = pe;
          if constexpr (DROP) {
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
    __syncthreads();
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

int fa_launch_dkv(const FaP& p, unsigned grid, size_t smem, hipStream_t st) {
  if (p.th) {
    if (fa_set_smem(attn_bwd_dkv_kernel<true>, smem) != WL_OK) return WL_ELAUNCH;
    WL_LAUNCH(attn_bwd_dkv_kernel<true>, dim3(grid), dim3(256), smem, st, p);
  } else {
    if (fa_set_smem(attn_bwd_dkv_kernel<false>, smem) != WL_OK) return WL_ELAUNCH;
    WL_LAUNCH(attn_bwd_dkv_kernel<false>, dim3(grid), dim3(256), smem, st, p);
  }
  return wl_check_launch();
}
