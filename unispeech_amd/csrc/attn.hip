// Gated relative-position-bias attention pieces.
//
// The reference materialises bias[B*H, T, T] = gate[b,h,i] * rel[h, j - i] in fp32 every layer
// (WavLM/modules.py:504-535) and hands it to SDPA as an additive mask.  Here the bias is never
// materialised: it is Toeplitz in (i, j), so a [H, 2T-1] table (built once per forward from the
// bucket embedding) plus the per-row gate reproduce it inside the softmax row kernel.
//   scores S = scale * Q.K^T           (MFMA GEMM, fp32 out)
//   P = softmax_j(S + gate_i * rel[j-i] + keypad(-inf))   -> this file, one wave per row
//   O = dropout(P).V                    (MFMA GEMM)
// Backward regenerates P from S and the saved log-sum-exp, and reduces the two bias gradients in the
// same pass: dgate[b,h,i] = sum_j dS*rel  (wave shuffle) and drel[h,d] = sum_{b,i} gate*dS[i,i+d]
// (LDS diagonal accumulators per block, then a deterministic cross-block sum).
#include "common.hpp"
#include <stdlib.h>
#include "../../include/wavlm_hip.h"

#define SM_NCH 4  // chunks of 256 keys: row length up to 1024 (T' = 749 @15 s, 999 @20 s)
#define SM_ROWS_PER_BLOCK 64

// lane l of chunk c owns the 4 consecutive keys j = 256*c + 4*l .. +3: 16-byte score loads, 8/16-byte probability
// stores, and ONE Philox4x32 call per lane per chunk for the dropout mask (counter = row * 256 + j / 4).
__device__ __forceinline__ void sm_load4(const void* p, long off, int dt, float (&v)[4]) {
  if (dt == WL_F32) {
    const float4 a = *reinterpret_cast<const float4*>((const float*)p + off);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  } else {
    const uint2 a = *reinterpret_cast<const uint2*>((const bf16_t*)p + off);
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
  }
}
__device__ __forceinline__ void sm_store4(void* p, long off, int dt, const float (&v)[4]) {
  if (dt == WL_F32) {
    *reinterpret_cast<float4*>((float*)p + off) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    uint2 o;
    o.x = pack_bf16x2(v[0], v[1]);
    o.y = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<uint2*>((bf16_t*)p + off) = o;
  }
}
__device__ __forceinline__ void sm_keep4(unsigned long long seed, long row, int j, unsigned th, bool (&k)[4]) {
  const Philox4 r = philox4x32_10(seed, (unsigned long long)row * 256ull + (unsigned)(j >> 2));
  k[0] = r.x >= th; k[1] = r.y >= th; k[2] = r.z >= th; k[3] = r.w >= th;
}

__global__ __launch_bounds__(256) void attn_softmax_fwd_kernel(const void* __restrict__ S, void* __restrict__ P,
    float* __restrict__ lse, const float* __restrict__ gate, const float* __restrict__ tab,
    const unsigned char* __restrict__ kpm, int B, int H, int T, long ldS, long ldP, int s_dt, int p_dt, unsigned th,
    float sc, unsigned long long seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long nrows = (long)B * H * T;
  const int L = 2 * T - 1;
  for (long row = (long)blockIdx.x * 4 + wave; row < nrows; row += (long)gridDim.x * 4) {
    const int i = (int)(row % T);
    const long bh = row / T;
    const int h = (int)(bh % H), b = (int)(bh / H);
    const float g = gate ? gate[row] : 0.f;
    const float* trow = tab ? tab + (long)h * L + (T - 1 - i) : nullptr;
    float v[SM_NCH][4];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < SM_NCH; ++c) {
      const int j0 = 256 * c + 4 * lane;
      if (j0 < ldS) sm_load4(S, row * ldS + j0, s_dt, v[c]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = j0 + e;
        float x = -INFINITY;
        if (j < T) {
          x = v[c][e];
          if (trow) x += g * trow[j];
          if (kpm && kpm[(long)b * T + j]) x = -INFINITY;
        }
        v[c][e] = x;
        mx = fmaxf(mx, x);
      }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < SM_NCH; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ex = (v[c][e] == -INFINITY) ? 0.f : __expf(v[c][e] - mx);
        v[c][e] = ex; sum += ex;
      }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    if (lane == 0) lse[row] = mx + __logf(sum);
#pragma unroll
    for (int c = 0; c < SM_NCH; ++c) {
      const int j0 = 256 * c + 4 * lane;
      if (j0 < ldP) {
        float o[4];
        bool keep[4] = {true, true, true, true};
        if (th) sm_keep4(seed, row, j0, th, keep);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float pj = v[c][e] * inv;  // exactly 0 for j >= T and for padded keys
          if (th) pj = keep[e] ? pj * sc : 0.f;
          o[e] = pj;
        }
        sm_store4(P, row * ldP + j0, p_dt, o);
      }
    }
  }
}

// grid: (ceil(T / SM_ROWS_PER_BLOCK), B*H); block = 4 waves; each wave walks rows of its chunk.
__global__ __launch_bounds__(256) void attn_softmax_bwd_kernel(const void* __restrict__ S, const void* __restrict__ dP,
    const float* __restrict__ lse, const float* __restrict__ gate, const float* __restrict__ tab,
    const unsigned char* __restrict__ kpm, void* __restrict__ dS, float* __restrict__ dgate,
    float* __restrict__ dtab_part, int B, int H, int T, long ldS, long ldP, int s_dt, int p_dt, unsigned th, float sc,
    unsigned long long seed) {
  extern __shared__ __attribute__((aligned(16))) float diag[];  // [2T-1]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int L = 2 * T - 1;
  const int bh = blockIdx.y;
  const int h = bh % H, b = bh / H;
  const int i0 = blockIdx.x * SM_ROWS_PER_BLOCK;
  if (tab) {
    for (int d = threadIdx.x; d < L; d += 256) diag[d] = 0.f;
    __syncthreads();
  }
  for (int ii = wave; ii < SM_ROWS_PER_BLOCK; ii += 4) {
    const int i = i0 + ii;
    if (i >= T) break;
    const long row = (long)bh * T + i;
    const float g = gate ? gate[row] : 0.f;
    const float* trow = tab ? tab + (long)h * L + (T - 1 - i) : nullptr;
    const float l = lse[row];
    float p[SM_NCH][4], dp[SM_NCH][4];
    float delta = 0.f;
#pragma unroll
    for (int c = 0; c < SM_NCH; ++c) {
      const int j0 = 256 * c + 4 * lane;
      float sv[4] = {0.f, 0.f, 0.f, 0.f}, dv[4] = {0.f, 0.f, 0.f, 0.f};
      bool keep[4] = {true, true, true, true};
      if (j0 < ldS) sm_load4(S, row * ldS + j0, s_dt, sv);
      if (j0 < ldP) sm_load4(dP, row * ldP + j0, p_dt, dv);
      if (th && j0 < T) sm_keep4(seed, row, j0, th, keep);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = j0 + e;
        float pj = 0.f, dpj = 0.f;
        if (j < T) {
          float x = sv[e];
          if (trow) x += g * trow[j];
          const bool masked = kpm && kpm[(long)b * T + j];
          pj = masked ? 0.f : __expf(x - l);
          dpj = dv[e];
          if (th) dpj = keep[e] ? dpj * sc : 0.f;
        }
        p[c][e] = pj; dp[c][e] = dpj;
        delta += pj * dpj;
      }
    }
    delta = wave_sum(delta);
    float dg = 0.f;
#pragma unroll
    for (int c = 0; c < SM_NCH; ++c) {
      const int j0 = 256 * c + 4 * lane;
      if (j0 < ldP) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int j = j0 + e;
          const float ds = p[c][e] * (dp[c][e] - delta);  // 0 for j >= T (p == 0)
          o[e] = ds;
          if (trow && j < T) {
            dg += ds * trow[j];
            atomicAdd(&diag[j - i + T - 1], g * ds);  // LDS float add; waves (different rows) may collide
          }
        }
        sm_store4(dS, row * ldP + j0, p_dt, o);
      }
    }
    if (dgate) {
      dg = wave_sum(dg);
      if (lane == 0) dgate[row] = dg;
    }
  }
  if (tab) {
    __syncthreads();
    float* out = dtab_part + ((long)bh * gridDim.x + blockIdx.x) * L;
    for (int d = threadIdx.x; d < L; d += 256) out[d] = diag[d];
  }
}

// drel[h][d] = sum_{b, chunk} part[((b*H + h) * nchunk + chunk)][d]
__global__ __launch_bounds__(256) void attn_dtab_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                               int B, int H, int nchunk, int L, int accumulate) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  const int h = blockIdx.y;
  if (d >= L) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < nchunk; ++c) s += part[(((long)b * H + h) * nchunk + c) * L + d];
  if (accumulate) s += out[(long)h * L + d];
  out[(long)h * L + d] = s;
}

// rel[h][d] = emb[bucket[d]][h]
__global__ __launch_bounds__(256) void relpos_gather_kernel(const void* __restrict__ emb, int emb_dt,
    const int* __restrict__ bucket, float* __restrict__ tab, int H, int L) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  const int h = blockIdx.y;
  if (d >= L) return;
  tab[(long)h * L + d] = ld_elem(emb, (long)bucket[d] * H + h, emb_dt);
}
// demb[k][h] = sum_{d : bucket[d] == k} drel[h][d]   (one wave per (k, h))
__global__ __launch_bounds__(64) void relpos_scatter_kernel(const float* __restrict__ dtab,
    const int* __restrict__ bucket, void* __restrict__ demb, int emb_dt, int H, int L, int nb) {
  const int k = blockIdx.x, h = blockIdx.y;
  float s = 0.f;
  for (int d = threadIdx.x; d < L; d += 64)
    if (bucket[d] == k) s += dtab[(long)h * L + d];
  s = wave_sum(s);
  if (threadIdx.x == 0) st_elem(demb, (long)k * H + h, emb_dt, s);
}

// ---- gate: g[b,h,t] = ga * (gb * a[h] - 1) + 2, (ga, gb) = sigmoid of the two 4-row sums of grep_linear(x_h) -----
// WavLM/modules.py:523-533.  One wave per (b, t); lane c (+64) owns channel c of every head, so the summed weight
// rows wa = sum_{k<4} W[k], wb = sum_{k>=4} W[k] and (backward) their gradient accumulators stay in registers.
#define GATE_NC 2  // head_dim up to 128
template <typename T, typename TP>
__global__ __launch_bounds__(256) void gate_fwd_kernel(const T* __restrict__ x, const TP* __restrict__ W,
    const TP* __restrict__ bias, const TP* __restrict__ grep_a, float* __restrict__ gate, float* __restrict__ ga_o,
    float* __restrict__ gb_o, int B, int Tn, int H, int hd) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long nbt = (long)B * Tn;
  const int D = H * hd;
  float ba = 0.f, bb = 0.f;
  for (int k = 0; k < 4; ++k) { ba += Elem<TP>::ld(bias + k); bb += Elem<TP>::ld(bias + 4 + k); }
  float wa[GATE_NC], wb[GATE_NC];
#pragma unroll
  for (int q = 0; q < GATE_NC; ++q) {
    const int c = lane + 64 * q;
    wa[q] = 0.f; wb[q] = 0.f;
    if (c < hd)
      for (int k = 0; k < 4; ++k) { wa[q] += Elem<TP>::ld(W + k * hd + c); wb[q] += Elem<TP>::ld(W + (4 + k) * hd + c); }
  }
  for (long bt = (long)blockIdx.x * 4 + wave; bt < nbt; bt += (long)gridDim.x * 4) {
    const long b = bt / Tn; const int t = (int)(bt - b * Tn);
    for (int h = 0; h < H; ++h) {
      float sa = 0.f, sb = 0.f;
#pragma unroll
      for (int q = 0; q < GATE_NC; ++q) {
        const int c = lane + 64 * q;
        if (c < hd) {
          const float xv = Elem<T>::ld(x + bt * D + h * hd + c);
          sa = fmaf(xv, wa[q], sa); sb = fmaf(xv, wb[q], sb);
        }
      }
      sa = wave_sum(sa) + ba; sb = wave_sum(sb) + bb;
      const float ga = 1.f / (1.f + __expf(-sa)), gb = 1.f / (1.f + __expf(-sb));
      if (lane == 0) {
        const long o = (b * H + h) * Tn + t;
        gate[o] = ga * (gb * Elem<TP>::ld(grep_a + h) - 1.f) + 2.f;
        ga_o[o] = ga; gb_o[o] = gb;
      }
    }
  }
}

#define GATE_BLOCKS 1024  // upper bound of the backward grid (sizes the partial-sum workspace); the launch uses 512
// ---- head_dim 64 fast path: 16-byte accesses -------------------------------------------------------------------
// A row of x is H heads x 8 chunks of 8 channels.  A wave takes TWO rows per step = 16 H chunks; lane l owns chunks
// l, l + 64, ... : chunk q -> row q / (8H), head (q % 8H) >> 3, sub-chunk q & 7.  Because 64 is a multiple of 8 a
// lane always sees the same 8 channels of a head (sub = l & 7), so its 2 x 8 summed weights and (backward) its 2 x 8
// weight-gradient accumulators are registers; the 8 lanes of a head are reduced with three xor-shuffles.
// (The generic kernels above move 2 bytes per lane per access: 48 / 130 us per layer at cfg2 against ~10 / ~20 us of
// HBM time.)
__device__ __forceinline__ void gate_ld8(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void gate_ld8(const bf16_t* p, float (&v)[8]) {
  const uint4 a = *reinterpret_cast<const uint4*>(p);
  v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
  v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
  v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
  v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
__device__ __forceinline__ void gate_st8(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void gate_st8(bf16_t* p, const float (&v)[8]) {
  uint4 o;
  o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
  o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = o;
}
#define GATE_MAXI 4  // chunks per lane per step: 16 H / 64 <= 4  (H <= 16)

template <typename T, typename TP>
__global__ __launch_bounds__(256) void gate_fwd64_kernel(const T* __restrict__ x, const TP* __restrict__ W,
    const TP* __restrict__ bias, const TP* __restrict__ grep_a, float* __restrict__ gate, float* __restrict__ ga_o,
    float* __restrict__ gb_o, int B, int Tn, int H) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long nbt = (long)B * Tn;
  const int D = H * 64, CPR = 8 * H, NCH = 2 * CPR;  // chunks per row, per step
  const int sub = lane & 7;
  float ba = 0.f, bb = 0.f;
  for (int k = 0; k < 4; ++k) { ba += Elem<TP>::ld(bias + k); bb += Elem<TP>::ld(bias + 4 + k); }
  float wa[8], wb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    wa[e] = 0.f; wb[e] = 0.f;
    for (int k = 0; k < 4; ++k) { wa[e] += Elem<TP>::ld(W + k * 64 + sub * 8 + e); wb[e] += Elem<TP>::ld(W + (4 + k) * 64 + sub * 8 + e); }
  }
  for (long rp = (long)blockIdx.x * 4 + wave; 2 * rp < nbt; rp += (long)gridDim.x * 4) {
    float xv[GATE_MAXI][8];
    bool ok[GATE_MAXI];
#pragma unroll
    for (int i = 0; i < GATE_MAXI; ++i) {
      const int q = lane + 64 * i;
      const long row = 2 * rp + q / CPR;
      ok[i] = q < NCH && row < nbt;
      if (ok[i]) gate_ld8(x + row * D + (q % CPR) * 8, xv[i]);
    }
#pragma unroll
    for (int i = 0; i < GATE_MAXI; ++i) {
      if (64 * i >= NCH) break;
      float sa = 0.f, sb = 0.f;
      if (ok[i]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { sa = fmaf(xv[i][e], wa[e], sa); sb = fmaf(xv[i][e], wb[e], sb); }
      }
      sa = wl_sum8(sa); sb = wl_sum8(sb);  // the 8 lanes of a head (DPP; was three ds_bpermute round trips each)
      if (ok[i] && sub == 0) {
        const int q = lane + 64 * i;
        const long row = 2 * rp + q / CPR;
        const int h = (q % CPR) >> 3;
        const long b = row / Tn; const int t = (int)(row - b * Tn);
        const float ga = 1.f / (1.f + __expf(-(sa + ba))), gb = 1.f / (1.f + __expf(-(sb + bb)));
        const long o = (b * H + h) * Tn + t;
        gate[o] = ga * (gb * Elem<TP>::ld(grep_a + h) - 1.f) + 2.f;
        ga_o[o] = ga; gb_o[o] = gb;
      }
    }
  }
}

// partial layout per block (same as the generic kernel): [2*64 (dWa, dWb)] [2 (dba, dbb)] [H (da)]
template <typename T, typename TP>
__global__ __launch_bounds__(256) void gate_bwd64_kernel(const float* __restrict__ dgate, const T* __restrict__ x,
    const TP* __restrict__ W, const TP* __restrict__ grep_a, const float* __restrict__ ga_i,
    const float* __restrict__ gb_i, T* __restrict__ dx, float* __restrict__ part, int B, int Tn, int H, int accdx) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [128 + 2 + H], zeroed, accumulated with LDS atomics once per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long nbt = (long)B * Tn;
  const int D = H * 64, CPR = 8 * H, NCH = 2 * CPR;
  const int PW = 128 + 2 + H;
  const int sub = lane & 7;
  for (int i = threadIdx.x; i < 4 * PW; i += 256) sm[i] = 0.f;
  float wa[8], wb[8], dwa[8], dwb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    wa[e] = 0.f; wb[e] = 0.f; dwa[e] = 0.f; dwb[e] = 0.f;
    for (int k = 0; k < 4; ++k) { wa[e] += Elem<TP>::ld(W + k * 64 + sub * 8 + e); wb[e] += Elem<TP>::ld(W + (4 + k) * 64 + sub * 8 + e); }
  }
  float dba = 0.f, dbb = 0.f, da[GATE_MAXI];
  float av[GATE_MAXI];
#pragma unroll
  for (int i = 0; i < GATE_MAXI; ++i) {
    da[i] = 0.f;
    const int q = lane + 64 * i;
    av[i] = q < NCH ? Elem<TP>::ld(grep_a + ((q % CPR) >> 3)) : 0.f;
  }
  __syncthreads();
  for (long rp = (long)blockIdx.x * 4 + wave; 2 * rp < nbt; rp += (long)gridDim.x * 4) {
    float xv[GATE_MAXI][8], dsa[GATE_MAXI], dsb[GATE_MAXI];
    bool ok[GATE_MAXI];
#pragma unroll
    for (int i = 0; i < GATE_MAXI; ++i) {
      const int q = lane + 64 * i;
      const long row = 2 * rp + q / CPR;
      ok[i] = q < NCH && row < nbt;
      dsa[i] = 0.f; dsb[i] = 0.f;
      if (ok[i]) {
        gate_ld8(x + row * D + (q % CPR) * 8, xv[i]);
        const int h = (q % CPR) >> 3;
        const long b = row / Tn; const int t = (int)(row - b * Tn);
        const long o = (b * H + h) * Tn + t;
        const float dg = dgate[o], ga = ga_i[o], gb = gb_i[o];
        dsa[i] = dg * (gb * av[i] - 1.f) * ga * (1.f - ga);
        dsb[i] = dg * ga * av[i] * gb * (1.f - gb);
        if (sub == 0) { da[i] = fmaf(dg * ga, gb, da[i]); dba += dsa[i]; dbb += dsb[i]; }
      }
    }
#pragma unroll
    for (int i = 0; i < GATE_MAXI; ++i) {
      if (ok[i]) {
        const int q = lane + 64 * i;
        const long row = 2 * rp + q / CPR;
        float o8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o8[e] = dsa[i] * wa[e] + dsb[i] * wb[e];
          dwa[e] = fmaf(dsa[i], xv[i][e], dwa[e]);
          dwb[e] = fmaf(dsb[i], xv[i][e], dwb[e]);
        }
        if (accdx) {  // dx already holds another consumer's gradient of the same tensor: add instead of overwrite
          float p8[8];
          gate_ld8(dx + row * D + (q % CPR) * 8, p8);
#pragma unroll
          for (int e = 0; e < 8; ++e) o8[e] += p8[e];
        }
        gate_st8(dx + row * D + (q % CPR) * 8, o8);
      }
    }
  }
  // lanes with equal `sub` hold the same 8 channels: fold lane bits 3..5, then one LDS atomic per value per wave
#pragma unroll
  for (int e = 0; e < 8; ++e) {
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) { dwa[e] += __shfl_xor(dwa[e], o, 64); dwb[e] += __shfl_xor(dwb[e], o, 64); }
  }
  // per-wave slots, summed in wave order below: the block's partial row is the same bits in every run (one shared slot
  // with LDS atomics from four waves summed in arrival order -- run-to-run differences in the last bit of the gate's
  // parameter gradients)
  float* smw = sm + wave * PW;
  if (lane < 8) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { smw[sub * 8 + e] = dwa[e]; smw[64 + sub * 8 + e] = dwb[e]; }
  }
  dba = wave_sum(dba); dbb = wave_sum(dbb);
  if (lane == 0) { smw[128] = dba; smw[129] = dbb; }
  if (sub == 0) {
#pragma unroll
    for (int i = 0; i < GATE_MAXI; ++i) {
      const int q = lane + 64 * i;
      if (q < NCH) atomicAdd(&smw[130 + ((q % CPR) >> 3)], da[i]);   // (collisions only inside the wave: fixed lane order)
    }
  }
  __syncthreads();
  float* out = part + (long)blockIdx.x * PW;
  for (int i = threadIdx.x; i < PW; i += 256) out[i] = (sm[i] + sm[PW + i]) + (sm[2 * PW + i] + sm[3 * PW + i]);
}

// ---- head_dim 64, H = 4 NI (12 or 16 heads): the row pair of a step is exactly NI chunks per lane -----------------
// Round 3: the kernels above ran at 2.2 (forward) / 3.1 TB/s (backward).  Both issued a step's loads, waited, and (backward,
// accumulate form) then issued a second dependent round of loads for dx.  Here every load of step n + 1 (x, dx, dgate / ga /
// gb) is in flight while step n computes: backward 35.0 -> 24.2 us per layer, forward 17.0 -> 14.6 us (profiles/r03/
// envab_gate_piped.txt).  What did NOT matter (same file family): the grid (256 ... 4096 blocks), 16-byte weight loads in the
// prologue, contiguous row ranges per block (whole output lines per XCD), and removing the 64-bit row / Tn per chunk and
// step -- a 37 MB pass of ~13 us is launch ramp + first-fetch latency + tail for half of its duration.  Computing the gate
// inside the LayerNorm kernels that stream this tensor anyway was built and measured too: the extra
// registers cost those kernels an occupancy step and most of what the two gate kernels take.
template <typename T> struct GateRaw;  // one 16-byte (bf16) / 32-byte (fp32) chunk of 8 channels as loaded
template <> struct GateRaw<bf16_t> {
  uint4 a;
  __device__ __forceinline__ void ld(const bf16_t* p) { a = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void get(float (&v)[8]) const {
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
    v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
  }
};
template <> struct GateRaw<float> {
  float4 a, b;
  __device__ __forceinline__ void ld(const float* p) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
  __device__ __forceinline__ void get(float (&v)[8]) const {
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
};
// summed weights of this lane's 8 channels: wa = rows 0-3, wb = rows 4-7 of grep_linear.weight [8, 64]
template <typename TP>
__device__ __forceinline__ void gate_weights(const TP* W, int sub, float (&wa)[8], float (&wb)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) { wa[e] = 0.f; wb[e] = 0.f; }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float ra[8], rb[8];
    gate_ld8(W + k * 64 + sub * 8, ra);
    gate_ld8(W + (4 + k) * 64 + sub * 8, rb);
#pragma unroll
    for (int e = 0; e < 8; ++e) { wa[e] += ra[e]; wb[e] += rb[e]; }
  }
}

template <typename T, typename TP, int NI>
__global__ __launch_bounds__(256) void gate_fwd64p_kernel(const T* __restrict__ x, const TP* __restrict__ W,
    const TP* __restrict__ bias, const TP* __restrict__ grep_a, float* __restrict__ gate, float* __restrict__ ga_o,
    float* __restrict__ gb_o, int B, int Tn) {
  constexpr int H = 4 * NI, D = H * 64, CPR = 8 * H;
  extern __shared__ __attribute__((aligned(16))) float stg[];  // [3][H][R]: the block's outputs, written out at the end
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long nbt = (long)B * Tn, npair = (nbt + 1) >> 1;
  const int sub = lane & 7;
  float ba = 0.f, bb = 0.f;
  for (int k = 0; k < 4; ++k) { ba += Elem<TP>::ld(bias + k); bb += Elem<TP>::ld(bias + 4 + k); }
  float wa[8], wb[8];
  gate_weights(W, sub, wa, wb);
  int rsel[NI], hsel[NI];
  float av[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int q = lane + 64 * i;
    rsel[i] = q / CPR; hsel[i] = (q % CPR) >> 3;
    av[i] = Elem<TP>::ld(grep_a + hsel[i]);
  }
  // A block walks a contiguous range of row pairs and keeps its [B, H, T] outputs in LDS until the end: written from the loop
  // they are 36 scattered 4-byte stores per row, and every wait for the next step's prefetched rows (s_waitcnt vmcnt) also
  // waits for those stores' acknowledgements -- one store round trip per step was the kernel's time (14.6 us for 37 MB,
  // whatever the grid or the instruction count).
  const long per = (npair + gridDim.x - 1) / gridDim.x;
  const long p0 = (long)blockIdx.x * per, pend = min(npair, p0 + per);
  const int R = (int)(2 * per);
  constexpr long pstep = 4;
  long rp = p0 + wave;
  GateRaw<T> px[NI];
  auto fetch = [&](long rp_) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      long row = 2 * rp_ + rsel[i]; if (row > nbt - 1) row = nbt - 1;
      px[i].ld(x + row * D + ((lane + 64 * i) % CPR) * 8);
    }
  };
  if (rp < pend) fetch(rp);
  for (; rp < pend; rp += pstep) {
    float xv[NI][8];
#pragma unroll
    for (int i = 0; i < NI; ++i) px[i].get(xv[i]);
    fetch(rp + pstep < pend ? rp + pstep : rp);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float sa = 0.f, sb = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { sa = fmaf(xv[i][e], wa[e], sa); sb = fmaf(xv[i][e], wb[e], sb); }
      sa = wl_sum8(sa); sb = wl_sum8(sb);
      if (sub == 0) {
        const float ga = __builtin_amdgcn_rcpf(1.f + __expf(-(sa + ba))), gb = __builtin_amdgcn_rcpf(1.f + __expf(-(sb + bb)));
        const int rr = (int)(2 * (rp - p0)) + rsel[i];
        stg[hsel[i] * R + rr] = ga * (gb * av[i] - 1.f) + 2.f;
        stg[(H + hsel[i]) * R + rr] = ga;
        stg[(2 * H + hsel[i]) * R + rr] = gb;
      }
    }
  }
  __syncthreads();
  const long row0 = 2 * p0;
  const int nrow = (int)min((long)R, nbt - row0);
  for (int idx = threadIdx.x; idx < 3 * H * R; idx += 256) {
    const int rr = idx % R, ah = idx / R, h = ah % H, arr = ah / H;
    if (rr < nrow) {
      const long row = row0 + rr;
      const long b = row / Tn; const int t = (int)(row - b * Tn);
      float* dst = arr == 0 ? gate : arr == 1 ? ga_o : gb_o;
      dst[(b * H + h) * Tn + t] = stg[idx];
    }
  }
}

// partial layout per block as gate_bwd64_kernel: [2*64 (dWa, dWb)] [2 (dba, dbb)] [H (da)]
template <typename T, typename TP, int NI, bool ACC>
__global__ __launch_bounds__(256) void gate_bwd64p_kernel(const float* __restrict__ dgate, const T* __restrict__ x,
    const TP* __restrict__ W, const TP* __restrict__ grep_a, const float* __restrict__ ga_i,
    const float* __restrict__ gb_i, T* __restrict__ dx, float* __restrict__ part, int B, int Tn) {
  constexpr int H = 4 * NI, D = H * 64, CPR = 8 * H, PW = 128 + 2 + H;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long nbt = (long)B * Tn, npair = (nbt + 1) >> 1;
  const int sub = lane & 7;
  for (int i = threadIdx.x; i < 4 * PW; i += 256) sm[i] = 0.f;
  float wa[8], wb[8], dwa[8], dwb[8];
  gate_weights(W, sub, wa, wb);
#pragma unroll
  for (int e = 0; e < 8; ++e) { dwa[e] = 0.f; dwb[e] = 0.f; }
  int rsel[NI], hsel[NI];
  float av[NI], da[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int q = lane + 64 * i;
    rsel[i] = q / CPR; hsel[i] = (q % CPR) >> 3;
    av[i] = Elem<TP>::ld(grep_a + hsel[i]);
    da[i] = 0.f;
  }
  float dba = 0.f, dbb = 0.f;
  __syncthreads();
  // contiguous range of row pairs per block (see the forward)
  const long per = (npair + gridDim.x - 1) / gridDim.x;
  const long pend = min(npair, (long)(blockIdx.x + 1) * per);
  constexpr long pstep = 4;
  long rp = (long)blockIdx.x * per + wave;
  GateRaw<T> px[NI], pd[ACC ? NI : 1];
  float pg[NI], pa[NI], pb[NI];
  // [B, H, T] offset of the row being FETCHED, advanced by 8 rows per step (no 64-bit division per chunk and step)
  const long ototal = (long)B * H * Tn;
  long on[NI];
  int tn[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const long row = 2 * rp + rsel[i];
    const long b = row / Tn;
    tn[i] = (int)(row - b * Tn);
    on[i] = (b * H + hsel[i]) * Tn + tn[i];
  }
  auto advance = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      tn[i] += 2 * (int)pstep; on[i] += 2 * pstep;
      while (tn[i] >= Tn) { tn[i] -= Tn; on[i] += (long)(H - 1) * Tn; }
    }
  };
  auto fetch = [&](long rp_) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      long row = 2 * rp_ + rsel[i]; if (row > nbt - 1) row = nbt - 1;
      const long off = row * D + ((lane + 64 * i) % CPR) * 8;
      px[i].ld(x + off);
      if constexpr (ACC) pd[i].ld(dx + off);
      const long o = on[i] < ototal ? on[i] : ototal - 1;  // (the row past an odd row count: any valid address, unused)
      pg[i] = dgate[o]; pa[i] = ga_i[o]; pb[i] = gb_i[o];
    }
  };
  if (rp < pend) fetch(rp);
  for (; rp < pend; rp += pstep) {
    float xv[NI][8], p8[ACC ? NI : 1][8], dsa[NI], dsb[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      px[i].get(xv[i]);
      if constexpr (ACC) pd[i].get(p8[i]);
      const bool ok = 2 * rp + rsel[i] < nbt;
      const float dg = ok ? pg[i] : 0.f, ga = pa[i], gb = pb[i];
      dsa[i] = dg * (gb * av[i] - 1.f) * ga * (1.f - ga);
      dsb[i] = dg * ga * av[i] * gb * (1.f - gb);
      if (sub == 0) { da[i] = fmaf(dg * ga, gb, da[i]); dba += dsa[i]; dbb += dsb[i]; }
    }
    if (rp + pstep < pend) { advance(); fetch(rp + pstep); }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const long row = 2 * rp + rsel[i];
      float o8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o8[e] = dsa[i] * wa[e] + dsb[i] * wb[e];
        if constexpr (ACC) o8[e] += p8[i][e];
        dwa[e] = fmaf(dsa[i], xv[i][e], dwa[e]);
        dwb[e] = fmaf(dsb[i], xv[i][e], dwb[e]);
      }
      if (row < nbt) gate_st8(dx + row * D + ((lane + 64 * i) % CPR) * 8, o8);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) { dwa[e] += __shfl_xor(dwa[e], o, 64); dwb[e] += __shfl_xor(dwb[e], o, 64); }
  }
  float* smw = sm + wave * PW;   // per-wave slots, summed in wave order (deterministic; see gate_bwd64_kernel)
  if (lane < 8) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { smw[sub * 8 + e] = dwa[e]; smw[64 + sub * 8 + e] = dwb[e]; }
  }
  dba = wave_sum(dba); dbb = wave_sum(dbb);
  if (lane == 0) { smw[128] = dba; smw[129] = dbb; }
  if (sub == 0) {
#pragma unroll
    for (int i = 0; i < NI; ++i) atomicAdd(&smw[130 + hsel[i]], da[i]);
  }
  __syncthreads();
  float* out = part + (long)blockIdx.x * PW;
  for (int i = threadIdx.x; i < PW; i += 256) out[i] = (sm[i] + sm[PW + i]) + (sm[2 * PW + i] + sm[3 * PW + i]);
}

// partial layout per block: [2*hd (dWa, dWb)] [2 (dba, dbb)] [H (da)]
template <typename T, typename TP>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ dgate, const T* __restrict__ x,
    const TP* __restrict__ W, const TP* __restrict__ grep_a, const float* __restrict__ ga_i,
    const float* __restrict__ gb_i, T* __restrict__ dx, float* __restrict__ part, int B, int Tn, int H, int hd, int accdx) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [4][2*hd + 2 + H]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long nbt = (long)B * Tn;
  const int D = H * hd;
  const int PW = 2 * hd + 2 + H;
  float wa[GATE_NC], wb[GATE_NC], dwa[GATE_NC], dwb[GATE_NC];
#pragma unroll
  for (int q = 0; q < GATE_NC; ++q) {
    const int c = lane + 64 * q;
    wa[q] = 0.f; wb[q] = 0.f; dwa[q] = 0.f; dwb[q] = 0.f;
    if (c < hd)
      for (int k = 0; k < 4; ++k) { wa[q] += Elem<TP>::ld(W + k * hd + c); wb[q] += Elem<TP>::ld(W + (4 + k) * hd + c); }
  }
  float dba = 0.f, dbb = 0.f, da = 0.f;  // da: lane h accumulates head h (H <= 64)
  for (long bt = (long)blockIdx.x * 4 + wave; bt < nbt; bt += (long)gridDim.x * 4) {
    const long b = bt / Tn; const int t = (int)(bt - b * Tn);
    for (int h = 0; h < H; ++h) {
      const long o = (b * H + h) * Tn + t;
      const float dg = dgate[o], ga = ga_i[o], gb = gb_i[o];
      const float a = Elem<TP>::ld(grep_a + h);
      const float dsa = dg * (gb * a - 1.f) * ga * (1.f - ga);
      const float dsb = dg * ga * a * gb * (1.f - gb);
      if (lane == h) da += dg * ga * gb;
      dba += dsa; dbb += dsb;
#pragma unroll
      for (int q = 0; q < GATE_NC; ++q) {
        const int c = lane + 64 * q;
        if (c < hd) {
          const float xv = Elem<T>::ld(x + bt * D + h * hd + c);
          const float prev = accdx ? Elem<T>::ld(dx + bt * D + h * hd + c) : 0.f;
          Elem<T>::st(dx + bt * D + h * hd + c, prev + dsa * wa[q] + dsb * wb[q]);
          dwa[q] = fmaf(dsa, xv, dwa[q]);
          dwb[q] = fmaf(dsb, xv, dwb[q]);
        }
      }
    }
  }
  float* mine = sm + wave * PW;
#pragma unroll
  for (int q = 0; q < GATE_NC; ++q) {
    const int c = lane + 64 * q;
    if (c < hd) { mine[c] = dwa[q]; mine[hd + c] = dwb[q]; }
  }
  if (lane == 0) { mine[2 * hd] = dba; mine[2 * hd + 1] = dbb; }
  if (lane < H) mine[2 * hd + 2 + lane] = da;
  __syncthreads();
  float* out = part + (long)blockIdx.x * PW;
  for (int i = threadIdx.x; i < PW; i += 256) out[i] = sm[i] + sm[PW + i] + sm[2 * PW + i] + sm[3 * PW + i];
}
// dW[8][hd], dbias[8], dgrep_a[H] from the block partials.  Pure load latency: 16 outputs x 64 row slices per block
// (nblk / 64 loads per thread, four in flight), grid = ceil(PW / 16) -- one block looping over all partials took 33 us.
__global__ __launch_bounds__(1024) void gate_bwd_finish_kernel(const float* __restrict__ part, int nblk, int H, int hd,
    void* dW, void* dbias, void* da, int pdt, int accumulate) {
  __shared__ float red[64][17];
  const int PW = 2 * hd + 2 + H;
  const int col = threadIdx.x & 15, slice = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + col;
  float s = 0.f;
  if (i < PW) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = slice;
    for (; b + 192 < nblk; b += 256) {
      s0 += part[(long)b * PW + i];
      s1 += part[(long)(b + 64) * PW + i];
      s2 += part[(long)(b + 128) * PW + i];
      s3 += part[(long)(b + 192) * PW + i];
    }
    for (; b < nblk; b += 64) s0 += part[(long)b * PW + i];
    s = (s0 + s1) + (s2 + s3);
  }
  red[slice][col] = s;
  __syncthreads();
  if (slice < 4) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[slice * 16 + k][col];
    red[slice * 16][col] = t;
  }
  __syncthreads();
  if (slice == 0 && i < PW) {
    s = (red[0][col] + red[16][col]) + (red[32][col] + red[48][col]);
    // accumulate: the outputs are gradient-sink slices (+=)
    auto put = [&](void* dst, long idx) { st_elem(dst, idx, pdt, accumulate ? s + ld_elem(dst, idx, pdt) : s); };
    if (i < hd) { for (int k = 0; k < 4; ++k) put(dW, (long)k * hd + i); }
    else if (i < 2 * hd) { for (int k = 0; k < 4; ++k) put(dW, (long)(4 + k) * hd + (i - hd)); }
    else if (i == 2 * hd) { for (int k = 0; k < 4; ++k) put(dbias, k); }
    else if (i == 2 * hd + 1) { for (int k = 0; k < 4; ++k) put(dbias, 4 + k); }
    else put(da, i - (2 * hd + 2));
  }
}

extern "C" {

int wavlm_attn_softmax_fwd(const void* S, void* P, float* lse, const float* gate, const float* tab, const uint8_t* kpm,
                           int32_t B, int32_t H, int32_t T, int64_t ldS, int64_t ldP, int32_t s_dtype, int32_t p_dtype,
                           float p_drop, uint64_t seed, void* stream) {
  if (!S || !P || !lse || B <= 0 || H <= 0 || T <= 0 || T > 256 * SM_NCH || ldP > 256 * SM_NCH || (ldS & 3) || (ldP & 3) || ldS < T || ldP < T)
    return WL_EINVAL;
  if ((gate == nullptr) != (tab == nullptr)) return WL_EINVAL;
  double tt = (double)p_drop * 4294967296.0; if (tt > 4294967295.0) tt = 4294967295.0;
  const unsigned th = p_drop > 0.f ? (unsigned)tt : 0u;
  const float sc = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  const long nrows = (long)B * H * T;
  long grid = (nrows + 3) / 4; if (grid > 16384) grid = 16384;
  WL_LAUNCH(attn_softmax_fwd_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, S, P, lse, gate,
                     tab, kpm, (int)B, (int)H, (int)T, (long)ldS, (long)ldP, (int)s_dtype, (int)p_dtype, th, sc,
                     (unsigned long long)seed);
  return wl_check_launch();
}

uint64_t wavlm_attn_softmax_bwd_workspace_bytes(int32_t B, int32_t H, int32_t T) {
  const uint64_t nchunk = (uint64_t)((T + SM_ROWS_PER_BLOCK - 1) / SM_ROWS_PER_BLOCK);
  return (uint64_t)B * H * nchunk * (2 * (uint64_t)T - 1) * sizeof(float);
}

int wavlm_attn_softmax_bwd(const void* S, const void* dP, const float* lse, const float* gate, const float* tab,
                           const uint8_t* kpm, void* dS, float* dgate, float* dtab, int32_t dtab_accumulate, int32_t B,
                           int32_t H, int32_t T, int64_t ldS, int64_t ldP, int32_t s_dtype, int32_t p_dtype,
                           float p_drop, uint64_t seed, void* workspace, uint64_t ws_bytes, void* stream) {
  if (!S || !dP || !lse || !dS || B <= 0 || H <= 0 || T <= 0 || T > 256 * SM_NCH || ldP > 256 * SM_NCH || (ldS & 3) || (ldP & 3) || ldS < T ||
      ldP < T)
    return WL_EINVAL;
  if ((gate == nullptr) != (tab == nullptr)) return WL_EINVAL;
  if (tab && (!dgate || !dtab || !workspace || ws_bytes < wavlm_attn_softmax_bwd_workspace_bytes(B, H, T)))
    return WL_EINVAL;
  double tt = (double)p_drop * 4294967296.0; if (tt > 4294967295.0) tt = 4294967295.0;
  const unsigned th = p_drop > 0.f ? (unsigned)tt : 0u;
  const float sc = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  hipStream_t st = (hipStream_t)stream;
  const int nchunk = (T + SM_ROWS_PER_BLOCK - 1) / SM_ROWS_PER_BLOCK;
  const int L = 2 * T - 1;
  WL_LAUNCH(attn_softmax_bwd_kernel, dim3((unsigned)nchunk, (unsigned)(B * H)), dim3(256),
                     tab ? (size_t)L * sizeof(float) : 0, st, S, dP, lse, gate, tab, kpm, dS, dgate, (float*)workspace,
                     (int)B, (int)H, (int)T, (long)ldS, (long)ldP, (int)s_dtype, (int)p_dtype, th, sc,
                     (unsigned long long)seed);
  int rc = wl_check_launch();
  if (rc != WL_OK || !tab) return rc;
  WL_LAUNCH(attn_dtab_reduce_kernel, dim3((unsigned)((L + 255) / 256), (unsigned)H), dim3(256), 0, st,
                     (const float*)workspace, dtab, (int)B, (int)H, nchunk, L, (int)dtab_accumulate);
  return wl_check_launch();
}

int wavlm_relpos_gather(const void* emb, int32_t emb_dtype, const int32_t* bucket, float* tab, int32_t H, int32_t L,
                        void* stream) {
  if (!emb || !bucket || !tab || H <= 0 || L <= 0) return WL_EINVAL;
  WL_LAUNCH(relpos_gather_kernel, dim3((unsigned)((L + 255) / 256), (unsigned)H), dim3(256), 0,
                     (hipStream_t)stream, emb, (int)emb_dtype, bucket, tab, (int)H, (int)L);
  return wl_check_launch();
}

int wavlm_relpos_scatter(const float* dtab, const int32_t* bucket, void* demb, int32_t emb_dtype, int32_t H, int32_t L,
                         int32_t num_buckets, void* stream) {
  if (!dtab || !bucket || !demb || H <= 0 || L <= 0 || num_buckets <= 0) return WL_EINVAL;
  WL_LAUNCH(relpos_scatter_kernel, dim3((unsigned)num_buckets, (unsigned)H), dim3(64), 0, (hipStream_t)stream,
                     dtab, bucket, demb, (int)emb_dtype, (int)H, (int)L, (int)num_buckets);
  return wl_check_launch();
}

int wavlm_gate_fwd(const void* x, const void* W, const void* bias, const void* grep_a, float* gate, float* ga, float* gb,
                   int32_t B, int32_t T, int32_t H, int32_t hd, int32_t dtype, int32_t param_dtype, void* stream) {
  if (!x || !W || !bias || !grep_a || !gate || !ga || !gb || B <= 0 || T <= 0 || H <= 0 || hd <= 0 || hd > 64 * GATE_NC) return WL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  long grid = ((long)B * T + 3) / 4; if (grid > 8192) grid = 8192;
  const bool fast = hd == 64 && H <= 16 && (((uintptr_t)x) & 15) == 0;
  if (fast) { grid = ((long)B * T + 7) / 8; if (grid > 2048) grid = 2048; }
  static const bool piped = !(getenv("WAVLM_GATE_PIPED") && getenv("WAVLM_GATE_PIPED")[0] == '0');  // A/B switch
  if (fast && piped && (H == 12 || H == 16) && (((uintptr_t)W) & 15) == 0) {
    static const int fwd_blocks = getenv("WAVLM_GATE_FWD_BLOCKS") ? atoi(getenv("WAVLM_GATE_FWD_BLOCKS")) : 512;  // measurement switch
    if (grid > fwd_blocks) grid = fwd_blocks;
    const long npair_ = ((long)B * T + 1) / 2;
    if (grid < (npair_ + 63) / 64) grid = (npair_ + 63) / 64;  // at most 64 row pairs per block: the staging slice stays < 25 KiB
    const long per_ = (npair_ + grid - 1) / grid;
    const size_t smemf = (size_t)3 * H * 2 * per_ * sizeof(float);
#define GFP(TT, TP) do { if (H == 12) WL_LAUNCH((gate_fwd64p_kernel<TT, TP, 3>), dim3((unsigned)grid), dim3(256), smemf, st, (const TT*)x, \
      (const TP*)W, (const TP*)bias, (const TP*)grep_a, gate, ga, gb, (int)B, (int)T); \
    else WL_LAUNCH((gate_fwd64p_kernel<TT, TP, 4>), dim3((unsigned)grid), dim3(256), smemf, st, (const TT*)x, \
      (const TP*)W, (const TP*)bias, (const TP*)grep_a, gate, ga, gb, (int)B, (int)T); } while (0)
    if (dtype == WL_F32 && param_dtype == WL_F32) GFP(float, float);
    else if (dtype == WL_BF16 && param_dtype == WL_BF16) GFP(bf16_t, bf16_t);
    else if (dtype == WL_BF16 && param_dtype == WL_F32) GFP(bf16_t, float);
    else return WL_EINVAL;
#undef GFP
    return wl_check_launch();
  }
#define GF(TT, TP) do { if (fast) WL_LAUNCH((gate_fwd64_kernel<TT, TP>), dim3((unsigned)grid), dim3(256), 0, st, (const TT*)x, \
    (const TP*)W, (const TP*)bias, (const TP*)grep_a, gate, ga, gb, (int)B, (int)T, (int)H); \
  else WL_LAUNCH((gate_fwd_kernel<TT, TP>), dim3((unsigned)grid), dim3(256), 0, st, (const TT*)x, \
    (const TP*)W, (const TP*)bias, (const TP*)grep_a, gate, ga, gb, (int)B, (int)T, (int)H, (int)hd); } while (0)
  if (dtype == WL_F32 && param_dtype == WL_F32) GF(float, float);
  else if (dtype == WL_BF16 && param_dtype == WL_BF16) GF(bf16_t, bf16_t);
  else if (dtype == WL_BF16 && param_dtype == WL_F32) GF(bf16_t, float);
  else return WL_EINVAL;
#undef GF
  return wl_check_launch();
}

uint64_t wavlm_gate_bwd_workspace_bytes(int32_t H, int32_t hd) {
  return (uint64_t)GATE_BLOCKS * (2 * (uint64_t)hd + 2 + H) * sizeof(float);
}

int wavlm_gate_bwd(const float* dgate, const void* x, const void* W, const void* grep_a, const float* ga,
                   const float* gb, void* dx, void* dW, void* dbias, void* dgrep_a, int32_t B, int32_t T, int32_t H,
                   int32_t hd, int32_t dtype, int32_t param_dtype, int32_t accumulate_params, void* workspace,
                   uint64_t ws_bytes, void* stream) {
  if (!dgate || !x || !W || !grep_a || !ga || !gb || !dx || !dW || !dbias || !dgrep_a || !workspace) return WL_EINVAL;
  if (B <= 0 || T <= 0 || H <= 0 || H > 64 || hd <= 0 || hd > 64 * GATE_NC || ws_bytes < wavlm_gate_bwd_workspace_bytes(H, hd)) return WL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  long grid = ((long)B * T + 3) / 4; if (grid > GATE_BLOCKS) grid = GATE_BLOCKS;
  const size_t smem = 4 * (2 * (size_t)hd + 2 + H) * sizeof(float);
  const bool fast = hd == 64 && H <= 16 && (((uintptr_t)x) & 15) == 0 && (((uintptr_t)dx) & 15) == 0;
  static const bool piped = !(getenv("WAVLM_GATE_PIPED") && getenv("WAVLM_GATE_PIPED")[0] == '0');  // A/B switch
  const bool fastp = fast && piped && (H == 12 || H == 16) && (((uintptr_t)W) & 15) == 0;
  static const int bwd_blocks = getenv("WAVLM_GATE_BWD_BLOCKS") ? atoi(getenv("WAVLM_GATE_BWD_BLOCKS")) : 512;  // measurement switch
  if (fastp && bwd_blocks > 0 && bwd_blocks < GATE_BLOCKS && grid > bwd_blocks) grid = bwd_blocks;
  if (!fastp && grid > 512) grid = 512;
#define GBP4(TT, TP, NI_, ACC_) WL_LAUNCH((gate_bwd64p_kernel<TT, TP, NI_, ACC_>), dim3((unsigned)grid), dim3(256), smem, st, dgate, \
    (const TT*)x, (const TP*)W, (const TP*)grep_a, ga, gb, (TT*)dx, (float*)workspace, (int)B, (int)T)
#define GBP(TT, TP) do { if (H == 12) { if (accumulate_params & 2) GBP4(TT, TP, 3, true); else GBP4(TT, TP, 3, false); } \
    else { if (accumulate_params & 2) GBP4(TT, TP, 4, true); else GBP4(TT, TP, 4, false); } } while (0)
  if (fastp) {
    if (dtype == WL_F32 && param_dtype == WL_F32) GBP(float, float);
    else if (dtype == WL_BF16 && param_dtype == WL_BF16) GBP(bf16_t, bf16_t);
    else if (dtype == WL_BF16 && param_dtype == WL_F32) GBP(bf16_t, float);
    else return WL_EINVAL;
  } else {
#define GB(TT, TP) do { if (fast) WL_LAUNCH((gate_bwd64_kernel<TT, TP>), dim3((unsigned)grid), dim3(256), smem, st, dgate, \
    (const TT*)x, (const TP*)W, (const TP*)grep_a, ga, gb, (TT*)dx, (float*)workspace, (int)B, (int)T, (int)H, (int)(accumulate_params & 2)); \
  else WL_LAUNCH((gate_bwd_kernel<TT, TP>), dim3((unsigned)grid), dim3(256), smem, st, dgate, \
    (const TT*)x, (const TP*)W, (const TP*)grep_a, ga, gb, (TT*)dx, (float*)workspace, (int)B, (int)T, (int)H, (int)hd, (int)(accumulate_params & 2)); } while (0)
  if (dtype == WL_F32 && param_dtype == WL_F32) GB(float, float);
  else if (dtype == WL_BF16 && param_dtype == WL_BF16) GB(bf16_t, bf16_t);
  else if (dtype == WL_BF16 && param_dtype == WL_F32) GB(bf16_t, float);
  else return WL_EINVAL;
#undef GB
  }
#undef GBP
#undef GBP4
  int rc = wl_check_launch();
  if (rc != WL_OK) return rc;
  {
    // deferred finishing (one encoder block's backward collects them, common.hpp): the partial row is [dWa hd][dWb hd][dba]
    // [dbb][da H]; grep_linear's eight rows are four copies of dWa then four of dWb (gate_bwd_finish_kernel), its bias alike
    const float* part = (const float*)workspace;
    const int PW = 2 * hd + 2 + H, acc = accumulate_params & 1;
    if (wl_fin_add(part, (int)grid, PW, hd, dW, param_dtype, acc, 4, hd)) {
      const uint64_t es = wl_esize(param_dtype);
      bool ok = wl_fin_add(part + hd, (int)grid, PW, hd, (char*)dW + 4ull * hd * es, param_dtype, acc, 4, hd);
      ok = ok && wl_fin_add(part + 2 * hd, (int)grid, PW, 1, dbias, param_dtype, acc, 4, 1);
      ok = ok && wl_fin_add(part + 2 * hd + 1, (int)grid, PW, 1, (char*)dbias + 4ull * es, param_dtype, acc, 4, 1);
      ok = ok && wl_fin_add(part + 2 * hd + 2, (int)grid, PW, H, dgrep_a, param_dtype, acc);
      return ok ? WL_OK : WL_EINVAL;
    }
  }
  WL_LAUNCH(gate_bwd_finish_kernel, dim3((unsigned)((2 * hd + 2 + H + 15) / 16)), dim3(1024), 0, st, (const float*)workspace,
                     (int)grid, (int)H, (int)hd, dW, dbias, dgrep_a, (int)param_dtype, (int)(accumulate_params & 1));
  rc = wl_check_launch();
  if (rc == WL_OK && (accumulate_params & 1)) {   // grep_linear.weight [8, hd], grep_linear.bias [8], grep_a [H]
    const uint64_t es = wl_esize(param_dtype);
    wl_notify_grad(dW, 8ull * hd * es, stream); wl_notify_grad(dbias, 8ull * es, stream); wl_notify_grad(dgrep_a, (uint64_t)H * es, stream);
  }
  return rc;
}

}  // extern "C"
