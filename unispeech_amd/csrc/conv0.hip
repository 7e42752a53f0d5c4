// First feature-extractor block: Conv1d(1 -> C, k = 10, stride 5, no bias) + GroupNorm(C groups == per-(b, c)
// statistics over time) + exact GELU, written channel-last [B, T0, C] so that conv1..6 become overlapping-row
// GEMMs (WavLM/WavLM.py:391-428,485-500; Fp32GroupNorm WavLM/modules.py:45-57).
//
// This is the HBM-bound stage of the extractor: 4 B/sample in, C/5 outputs per sample out.  The raw conv
// output is never written: the statistics pass and the apply pass both recompute the 10-tap conv from an LDS
// copy of the waveform segment (10 FMA per output on the VALU, far below the write-bandwidth bound), so HBM
// traffic is: waveform read twice, activation written once.  Lane l of a wave owns channels 8l..8l+7 for all
// time steps of the wave, so every store is one 16-byte (bf16) / 2 x 16-byte (f32) vector and a wave writes a
// full contiguous 1 KiB / 2 KiB row.  Backward recomputes the same way (conv0 has no input gradient).
#include <atomic>
#include "common.hpp"
#include "conv0_shared.hpp"
#include "../../include/wavlm_hip.h"


template <typename T> struct V8 {
  static __device__ __forceinline__ void ld(const T* p, float (&v)[8]);
  static __device__ __forceinline__ void st(T* p, const float (&v)[8]);
};
template <> __device__ __forceinline__ void V8<float>::ld(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void V8<float>::st(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void V8<bf16_t>::ld(const bf16_t* p, float (&v)[8]) {
  const uint4 a = *reinterpret_cast<const uint4*>(p);
  v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
  v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
  v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
  v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
template <> __device__ __forceinline__ void V8<bf16_t>::st(bf16_t* p, const float (&v)[8]) {
  uint4 o;
  o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
  o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = o;
}

// ---- GELU / GELU' by table (bf16 activations only) ------------------------------------------------------------------
// The erf evaluation (one v_rcp, one v_exp, six FMAs, sign handling: ~18 VALU slots) was more than half of the VALU work
// per output of the apply pass, which is what kept this HBM-bound stage at 2.2 TB/s.  With bf16 output (relative
// resolution 2^-9) a piecewise-linear table is indistinguishable: 2048 cells over [-8, 8), cell i holds (slope,
// intercept) of the chord through the exact values at its ends -> |error| <= h^2/8 * max|f''| = 8e-6 absolute, and the
// evaluation is fma + med3 + cvt + one 8-byte LDS read + fma.  Outside the range the end cells extrapolate (slope 1 /
// 0 for gelu, 0 for gelu'), NaN and inf propagate through the final fma.  The fp32 (parity) instantiations keep the
// exact erf path.
__device__ float2 g_gelu_tab[2][GT_N];  // [0] gelu, [1] gelu'
__global__ void gelu_tab_init_kernel() {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= GT_N) return;
  const double h = 16.0 / GT_N, x0 = -8.0 + i * h, x1 = x0 + h;
  auto f = [](double x) { return 0.5 * x * (1.0 + erf(x * 0.70710678118654752440)); };
  auto d = [](double x) { return 0.5 * (1.0 + erf(x * 0.70710678118654752440)) + x * 0.39894228040143267794 * exp(-0.5 * x * x); };
  const double s0 = (f(x1) - f(x0)) / h, s1 = (d(x1) - d(x0)) / h;
  g_gelu_tab[0][i] = make_float2((float)s0, (float)(f(x0) - s0 * x0));
  g_gelu_tab[1][i] = make_float2((float)s1, (float)(d(x0) - s1 * x0));
}
// per device, thread-safe, and complete before the first user on ANY stream (same scheme as gelu_tab4_get, gemm_bf16.hip)
static const float2* gelu_tab_get(hipStream_t st) {
  static std::mutex mu;
  static std::atomic<const float2*> ptr[WL_MAX_DEVICES];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= WL_MAX_DEVICES) return nullptr;
  const float2* p = ptr[dev].load(std::memory_order_acquire);
  if (p) return p;   // steady state: no lock
  std::lock_guard<std::mutex> lk(mu);
  p = ptr[dev].load(std::memory_order_acquire);
  if (!p) {
    void* a = nullptr;
    if (hipGetSymbolAddress(&a, HIP_SYMBOL(g_gelu_tab)) != hipSuccess) return nullptr;
    hipLaunchKernelGGL(gelu_tab_init_kernel, dim3(GT_N / 256), dim3(256), 0, st);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return nullptr;
    p = (const float2*)a;
    ptr[dev].store(p, std::memory_order_release);
  }
  return p;
}
static int gelu_tab_ensure(hipStream_t st) { return gelu_tab_get(st) ? WL_OK : WL_ELAUNCH; }
// copy one table (16 KiB) into LDS; the caller synchronises
__device__ __forceinline__ void gelu_tab_to_lds(int which, float2* tab) {
  const float4* src = reinterpret_cast<const float4*>(g_gelu_tab[which]);
  float4* dst = reinterpret_cast<float4*>(tab);
  for (int i = threadIdx.x; i < GT_N / 2; i += blockDim.x) dst[i] = src[i];
}
__device__ __forceinline__ float tab_eval(const float2* tab, float x) {
  float u = fmaf(x, GT_INV_H, -GT_LO * GT_INV_H);
  u = __builtin_amdgcn_fmed3f(u, 0.f, (float)(GT_N - 1));
  const float2 sc = tab[(int)u];
  return fmaf(sc.x, x, sc.y);
}

// stage the waveform samples feeding time steps [t0, t0 + nt) of batch row b into LDS (as f32)
template <typename TW>
__device__ __forceinline__ void stage_wave(const TW* __restrict__ wav, long T, int b, int t0, int nt, int stride,
                                           float* seg) {
  const long s0 = (long)t0 * stride;
  const int ns = (nt - 1) * stride + C0_KW;
  for (int i = threadIdx.x; i < ns; i += blockDim.x) seg[i] = Elem<TW>::ld(wav + (long)b * T + s0 + i);
}

// lane's 8 x KW weights (channels 8*lane .. +7)
template <typename TP>
__device__ __forceinline__ void load_w(const TP* __restrict__ W, int lane, int C, float (&w)[8][C0_KW]) {
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int k = 0; k < C0_KW; ++k) {
      const int c = lane * 8 + e;
      w[e][k] = c < C ? Elem<TP>::ld(W + (long)c * C0_KW + k) : 0.f;
    }
}
__device__ __forceinline__ void conv_at(const float* seg, int tt, int stride, const float (&w)[8][C0_KW], float (&y)[8]) {
  float xw[C0_KW];
#pragma unroll
  for (int k = 0; k < C0_KW; ++k) xw[k] = seg[tt * stride + k];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < C0_KW; ++k) a = fmaf(xw[k], w[e][k], a);
    y[e] = a;
  }
}

// pass 1: GroupNorm statistics WITHOUT evaluating the convolution.  mean_c = sum_k w[c][k] Q[k] / T0 and
// E[y_c^2] = sum_{j,k} w[c][j] w[c][k] XX[j][k] / T0 with the waveform-only sums Q[k] = sum_t x[s t + k] and
// XX[j][k] = sum_t x[s t + j] x[s t + k] (a 10 x 10 Gram matrix per batch row): O(110) instead of O(10 C) work per
// frame -- the pass that used to recompute the conv for all C channels (0.19 ms at cfg2) is a few microseconds.
// partx[(b * nchunk + chunk)][112]: Q[10], XX[10][10] of the chunk (fp32 partials over <= 512 frames).
template <typename TW>
__global__ __launch_bounds__(256) void conv0_gram_kernel(const TW* __restrict__ wav, float* __restrict__ partx, long T,
                                                         int T0, int stride) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* seg = sm;
  __shared__ float half[2][C0_NX];
  const int b = blockIdx.y, t0 = blockIdx.x * C0_TCH;
  const int nt = min(C0_TCH, T0 - t0);
  stage_wave(wav, T, b, t0, nt, stride, seg);
  __syncthreads();
  const int idx = threadIdx.x % 110, hf = threadIdx.x / 110;  // two halves of the chunk's frames, 110 sums each
  if (hf < 2) {
    const int ta = hf ? nt / 2 : 0, tb = hf ? nt : nt / 2;
    float acc = 0.f;
    if (idx < C0_KW) {
      for (int tt = ta; tt < tb; ++tt) acc += seg[tt * stride + idx];
    } else {
      const int j = (idx - C0_KW) / C0_KW, k = (idx - C0_KW) % C0_KW;
      for (int tt = ta; tt < tb; ++tt) acc = fmaf(seg[tt * stride + j], seg[tt * stride + k], acc);
    }
    half[hf][idx] = acc;
  }
  __syncthreads();
  if (threadIdx.x < 110) partx[((long)b * gridDim.x + blockIdx.x) * C0_NX + threadIdx.x] = half[0][threadIdx.x] + half[1][threadIdx.x];
}

// stats[b][c] = (mean, rstd) from the chunk Gram partials (double); gram[b][112] = the reduced sums (backward reuses them)
template <typename TP>
__global__ __launch_bounds__(256) void conv0_stats_from_gram_kernel(const float* __restrict__ partx, const TP* __restrict__ W,
    float* __restrict__ stats, int nchunk, int C, int T0, float eps) {
  __shared__ double xs[C0_NX];
  const int b = blockIdx.y;
  if (threadIdx.x < 110) {
    double s = 0.0;
    for (int k = 0; k < nchunk; ++k) s += partx[((long)b * nchunk + k) * C0_NX + threadIdx.x];
    xs[threadIdx.x] = s;
  }
  __syncthreads();
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double wv[C0_KW];
#pragma unroll
  for (int j = 0; j < C0_KW; ++j) wv[j] = Elem<TP>::ld(W + (long)c * C0_KW + j);
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int j = 0; j < C0_KW; ++j) {
    s1 += wv[j] * xs[j];
    double r = 0.0;
#pragma unroll
    for (int k = 0; k < C0_KW; ++k) r += wv[k] * xs[C0_KW + j * C0_KW + k];
    s2 += wv[j] * r;
  }
  const double mean = s1 / T0;
  double var = s2 / T0 - mean * mean;
  if (var < 0.0) var = 0.0;
  stats[((long)b * C + c) * 2] = (float)mean;
  stats[((long)b * C + c) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// pass 2: y0 = gelu((conv - mean) * rstd * gamma + beta), channel-last
template <typename TW, typename TP, typename TO>
__global__ __launch_bounds__(256) void conv0_apply_kernel(const TW* __restrict__ wav, const TP* __restrict__ W,
    const TP* __restrict__ gamma, const TP* __restrict__ beta, const float* __restrict__ stats, TO* __restrict__ out,
    long T, int T0, int C, int stride) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* seg = sm;
  constexpr bool TAB = sizeof(TO) == 2;  // bf16 activations: table GELU
  float2* tab = reinterpret_cast<float2*>(sm + ((C0_TCH - 1) * stride + C0_KW + 3) / 4 * 4);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, t0 = blockIdx.x * C0_TCH;
  const int nt = min(C0_TCH, T0 - t0);
  stage_wave(wav, T, b, t0, nt, stride, seg);
  if constexpr (TAB) gelu_tab_to_lds(0, tab);
  float w[8][C0_KW];
  load_w(W, lane, C, w);
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = lane * 8 + e;
    if (c < C) {
      const float mean = stats[((long)b * C + c) * 2], rstd = stats[((long)b * C + c) * 2 + 1];
      const float g = Elem<TP>::ld(gamma + c), bt = Elem<TP>::ld(beta + c);
      sc[e] = rstd * g; sh[e] = bt - mean * rstd * g;
    } else { sc[e] = 0.f; sh[e] = 0.f; }
  }
  __syncthreads();
  if (lane * 8 >= C) return;
  for (int tt = wave; tt < nt; tt += 4) {
    float y[8];
    conv_at(seg, tt, stride, w, y);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float zz = fmaf(y[e], sc[e], sh[e]);
      if constexpr (TAB) y[e] = tab_eval(tab, zz); else y[e] = gelu_f(zz);
    }
    V8<TO>::st(out + ((long)b * T0 + t0 + tt) * C + lane * 8, y);
  }
}

// Backward in ONE pass over the incoming gradient (1.57 GB at cfg2 -- this stage is HBM-bound on reading it):
//   dz   = g * gscale * gelu'(z),  xhat = (conv - mean) * rstd,  z = xhat * gamma + beta
//   dconv = rstd * gamma * (dz - A/T0 - xhat * Bq/T0),   A = sum_t dz,  Bq = sum_t dz * xhat      (GroupNorm backward)
//   dW[c][k] = sum_{b,t} dconv[b,t,c] * x[b, 5t+k]
//            = sum_b rstd * gamma * ( P[b,c,k] - A/T0 * Q[b,k] - Bq/T0 * R[b,c,k] )
//   with P = sum_t dz * x_k            (the only term that needs g: accumulated in the same pass as A and Bq)
//        Q = sum_t x_k                 (waveform only)
//        R = sum_t xhat * x_k = rstd * ( sum_j w[c][j] * XX[b][j][k] - mean * Q[b,k] ),  XX = sum_t x_j x_k  (waveform only)
// so the second sweep over g (and the second recomputation of conv + gelu') of a two-pass GroupNorm backward is
// replaced by a 10 x 10 waveform Gram matrix per batch row.
// Bq needs no accumulation of its own either: sum_t dz * y = sum_k w[c][k] P[c][k], so
//   Bq = sum_t dz * xhat = rstd * (sum_k w[c][k] P[b,c,k] - mean * A).
// part[(b * nchunk + chunk)][11][C]: A, P[0..9];  partx[(b * nchunk + chunk)][112]: Q[10], XX[10][10]
#define C0_LN_NQ (3 + C0_KW)  // layer_norm mode: dbeta, dgamma, dW[.][0..9], dbias (conv_bias=True; zero work otherwise)
template <typename TW, typename TP, typename TO>
__global__ __launch_bounds__(256) void conv0_bwd_fused_kernel(const TW* __restrict__ wav, const TP* __restrict__ W,
    const TP* __restrict__ gamma, const TP* __restrict__ beta, const float* __restrict__ stats,
    const TO* __restrict__ g, float* __restrict__ part, float* __restrict__ partx, long T, int T0, int C, int stride,
    float gscale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* seg = sm;
  float* red = sm + ((C0_TCH_BWD - 1) * stride + C0_KW + 3) / 4 * 4;  // [4 waves][512]
  constexpr bool TAB = sizeof(TO) == 2;  // bf16 gradients: table gelu'
  float2* tab = reinterpret_cast<float2*>(red + 4 * 512);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, t0 = blockIdx.x * C0_TCH_BWD;
  const int nt = min(C0_TCH_BWD, T0 - t0);
  stage_wave(wav, T, b, t0, nt, stride, seg);
  if constexpr (TAB) gelu_tab_to_lds(1, tab);
  float w[8][C0_KW];
  load_w(W, lane, C, w);
  float mean[8], rstd[8], gm[8], bt[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = min(lane * 8 + e, C - 1);
    mean[e] = stats[((long)b * C + c) * 2]; rstd[e] = stats[((long)b * C + c) * 2 + 1];
    gm[e] = Elem<TP>::ld(gamma + c); bt[e] = Elem<TP>::ld(beta + c);
  }
  __syncthreads();
  // waveform-only sums of this chunk: Q[k] and XX[j][k]
  if (threadIdx.x < C0_KW + C0_KW * C0_KW) {
    const int idx = threadIdx.x;
    float acc = 0.f;
    if (idx < C0_KW) {
      for (int tt = 0; tt < nt; ++tt) acc += seg[tt * stride + idx];
    } else {
      const int j = (idx - C0_KW) / C0_KW, k = (idx - C0_KW) % C0_KW;
      for (int tt = 0; tt < nt; ++tt) acc = fmaf(seg[tt * stride + j], seg[tt * stride + k], acc);
    }
    partx[((long)b * gridDim.x + blockIdx.x) * C0_NX + idx] = acc;
  }
  float a1[8], pw[8][C0_KW];
  float zs[8], zb[8];  // z = y * zs + zb  (GroupNorm affine folded: one fma per output)
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    a1[e] = 0.f;
    zs[e] = rstd[e] * gm[e]; zb[e] = bt[e] - mean[e] * rstd[e] * gm[e];
#pragma unroll
    for (int k = 0; k < C0_KW; ++k) pw[e][k] = 0.f;
  }
  if (lane * 8 < C) {
    for (int tt = wave; tt < nt; tt += 4) {
      float xw[C0_KW], y[8], gv[8];
#pragma unroll
      for (int k = 0; k < C0_KW; ++k) xw[k] = seg[tt * stride + k];
      V8<TO>::ld(g + ((long)b * T0 + t0 + tt) * C + lane * 8, gv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < C0_KW; ++k) a = fmaf(xw[k], w[e][k], a);
        y[e] = a;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float zz = fmaf(y[e], zs[e], zb[e]);
        float gp;
        if constexpr (TAB) gp = tab_eval(tab, zz); else gp = gelu_grad_f(zz);
        const float dz = gv[e] * gscale * gp;
        a1[e] += dz;
#pragma unroll
        for (int k = 0; k < C0_KW; ++k) pw[e][k] = fmaf(dz, xw[k], pw[e][k]);
      }
    }
  }
  float* out = part + ((long)b * gridDim.x + blockIdx.x) * (long)C0_NQ * C;
#pragma unroll
  for (int q = 0; q < C0_NQ; ++q) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave * 512 + lane * 8 + e] = q == 0 ? a1[e] : pw[e][q >= 1 ? q - 1 : 0];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) out[(long)q * C + c] = red[c] + red[512 + c] + red[1024 + c] + red[1536 + c];
  }
}

// Per (b, c): reduce the chunk partials (double), form this batch row's dW contribution, and (A, Bq) for dgamma/dbeta.
// grid (ceil(C / 64), B), 256 threads = 64 channels x 4 chunk slices.
template <typename TP>
__global__ __launch_bounds__(256) void conv0_bwd_combine_kernel(const float* __restrict__ part, const float* __restrict__ partx,
    const float* __restrict__ stats, const TP* __restrict__ W, const TP* __restrict__ gamma, float* __restrict__ ab,
    float* __restrict__ dwb, int nchunk, int C, int T0) {
  __shared__ double red[4][C0_NQ][64];
  __shared__ double xs[C0_NX];
  const int col = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + col, b = blockIdx.y;
  if (threadIdx.x < C0_KW + C0_KW * C0_KW) {
    double s = 0.0;
    for (int k = 0; k < nchunk; ++k) s += partx[((long)b * nchunk + k) * C0_NX + threadIdx.x];
    xs[threadIdx.x] = s;
  }
  double acc[C0_NQ];
#pragma unroll
  for (int q = 0; q < C0_NQ; ++q) acc[q] = 0.0;
  if (c < C)
    for (int k = slice; k < nchunk; k += 4) {
      const float* pp = part + ((long)b * nchunk + k) * (long)C0_NQ * C + c;
#pragma unroll
      for (int q = 0; q < C0_NQ; ++q) acc[q] += pp[(long)q * C];
    }
#pragma unroll
  for (int q = 0; q < C0_NQ; ++q) red[slice][q][col] = acc[q];
  __syncthreads();
  if (slice == 0 && c < C) {
#pragma unroll
    for (int q = 0; q < C0_NQ; ++q) acc[q] = red[0][q][col] + red[1][q][col] + red[2][q][col] + red[3][q][col];
    const double mean = stats[((long)b * C + c) * 2], rstd = stats[((long)b * C + c) * 2 + 1];
    const double gm = Elem<TP>::ld(gamma + c);
    double wv[C0_KW];
#pragma unroll
    for (int j = 0; j < C0_KW; ++j) wv[j] = Elem<TP>::ld(W + (long)c * C0_KW + j);
    double dzy = 0.0;  // sum_t dz * y = sum_k w[k] P[k]
#pragma unroll
    for (int j = 0; j < C0_KW; ++j) dzy += wv[j] * acc[1 + j];
    const double bq = rstd * (dzy - mean * acc[0]);
    const double am = acc[0] / T0, bm = bq / T0;
    ab[((long)b * C + c) * 2] = (float)acc[0];
    ab[((long)b * C + c) * 2 + 1] = (float)bq;
#pragma unroll
    for (int k = 0; k < C0_KW; ++k) {
      double r = -mean * xs[k];
#pragma unroll
      for (int j = 0; j < C0_KW; ++j) r += wv[j] * xs[C0_KW + j * C0_KW + k];
      r *= rstd;
      dwb[((long)b * C + c) * C0_KW + k] = (float)(rstd * gm * (acc[1 + k] - am * xs[k] - bm * r));
    }
  }
}

// dgamma[c] = sum_b Bq[b][c], dbeta[c] = sum_b A[b][c]
__global__ __launch_bounds__(256) void conv0_bwd_affine_kernel(const float* __restrict__ ab, void* dgamma, void* dbeta,
                                                               int pdt, int B, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double ta = 0.0, tb = 0.0;
  for (int b = 0; b < B; ++b) { ta += ab[((long)b * C + c) * 2]; tb += ab[((long)b * C + c) * 2 + 1]; }
  st_elem(dbeta, c, pdt, (float)ta);
  st_elem(dgamma, c, pdt, (float)tb);
}

// dW[i] = sum over nblk partial rows (double)
__global__ __launch_bounds__(256) void conv0_bwd_w_finish_kernel(const float* __restrict__ part, int nblk, int n,
                                                                 void* dW, int pdt) {
  __shared__ double red[4][64];
  const int col = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + col;
  double s = 0.0;
  if (i < n)
    for (int b = slice; b < nblk; b += 4) s += part[(long)b * n + i];
  red[slice][col] = s;
  __syncthreads();
  if (slice == 0 && i < n) st_elem(dW, i, pdt, (float)(red[0][col] + red[1][col] + red[2][col] + red[3][col]));
}

// ---------------------------------------------------------------------------------------------------------------
// extractor_mode = "layer_norm" (WavLM-Large: WavLM/WavLM.py:403-418): conv0 -> LayerNorm over the C channels of
// every frame -> GELU.  A frame's 512 channels are one wave (lane owns 8), so the LayerNorm statistics are two wave
// reductions per frame and everything fuses into one pass; backward recomputes conv + statistics and needs two more
// reductions per frame, again a single pass over the incoming gradient.
template <typename TW, typename TP, typename TO>
__global__ __launch_bounds__(256) void conv0_ln_fwd_kernel(const TW* __restrict__ wav, const TP* __restrict__ W,
    const TP* __restrict__ cbias, const TP* __restrict__ gamma, const TP* __restrict__ beta, TO* __restrict__ out, long T,
    int T0, int C, int stride, float eps) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* seg = sm;
  constexpr bool TAB = sizeof(TO) == 2;  // bf16 activations: table GELU (see gelu_tab_to_lds)
  float2* tab = reinterpret_cast<float2*>(sm + ((C0_TCH - 1) * stride + C0_KW + 3) / 4 * 4);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, t0 = blockIdx.x * C0_TCH;
  const int nt = min(C0_TCH, T0 - t0);
  stage_wave(wav, T, b, t0, nt, stride, seg);
  if constexpr (TAB) gelu_tab_to_lds(0, tab);
  float w[8][C0_KW];
  load_w(W, lane, C, w);
  float gm[8], bt[8], cb[8];
  const bool act = lane * 8 < C;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = min(lane * 8 + e, C - 1);
    gm[e] = Elem<TP>::ld(gamma + c); bt[e] = Elem<TP>::ld(beta + c);
    cb[e] = cbias ? Elem<TP>::ld(cbias + c) : 0.f;  // Conv1d bias (conv_bias=True), added before the LayerNorm
  }
  __syncthreads();
  const float invC = 1.f / (float)C;
  for (int tt = wave; tt < nt; tt += 4) {
    float y[8];
    conv_at(seg, tt, stride, w, y);
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] += cb[e];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += act ? y[e] : 0.f;
    const float mean = wave_sum(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = y[e] - mean; q += act ? d * d : 0.f; }
    const float rstd = rsqrtf(wave_sum(q) * invC + eps);
    if (act) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float zz = fmaf((y[e] - mean) * rstd, gm[e], bt[e]);
        if constexpr (TAB) y[e] = tab_eval(tab, zz); else y[e] = gelu_f(zz);
      }
      V8<TO>::st(out + ((long)b * T0 + t0 + tt) * C + lane * 8, y);
    }
  }
}

// part[(b * nchunk + chunk)][12][C]: dbeta, dgamma, dW[.][0..9]
// OCC: workgroups per CU the register allocation aims at (2: 256 VGPRs with a dozen spilled words reloaded per frame;
// 1: 258-260 VGPRs, no spill, one wave per SIMD) -- WAVLM_CONV0_LN_OCC selects, default from the A/B in profiles/r03
template <typename TW, typename TP, typename TO, int OCC>
__global__ __launch_bounds__(256, OCC) void conv0_ln_bwd_kernel(const TW* __restrict__ wav, const TP* __restrict__ W,
    const TP* __restrict__ cbias, const TP* __restrict__ gamma, const TP* __restrict__ beta, const TO* __restrict__ g,
    float* __restrict__ part, long T, int T0, int C, int stride, float eps, float gscale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* seg = sm;
  float* red = sm + ((C0_TCH_BWD - 1) * stride + C0_KW + 3) / 4 * 4;  // [4 waves][512]
  constexpr bool TAB = sizeof(TO) == 2;  // bf16 gradients: table gelu'
  float2* tab = reinterpret_cast<float2*>(red + 4 * 512);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, t0 = blockIdx.x * C0_TCH_BWD;
  const int nt = min(C0_TCH_BWD, T0 - t0);
  stage_wave(wav, T, b, t0, nt, stride, seg);
  if constexpr (TAB) gelu_tab_to_lds(1, tab);
  float w[8][C0_KW];
  load_w(W, lane, C, w);
  float gm[8], bt[8], cb[8];
  const bool act = lane * 8 < C;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = min(lane * 8 + e, C - 1);
    gm[e] = Elem<TP>::ld(gamma + c); bt[e] = Elem<TP>::ld(beta + c);
    cb[e] = cbias ? Elem<TP>::ld(cbias + c) : 0.f;
  }
  __syncthreads();
  float a1[8], a2[8], pw[8][C0_KW], pb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    a1[e] = 0.f; a2[e] = 0.f; pb[e] = 0.f;
#pragma unroll
    for (int k = 0; k < C0_KW; ++k) pw[e][k] = 0.f;
  }
  const float invC = 1.f / (float)C;
  // A wave owns a frame: per frame four DEPENDENT wave reductions sit between the gradient load and the accumulation.  With
  // the reductions as LDS butterflies (six ds_bpermute each) and the kernel's ~260 VGPRs (one wave per SIMD) the frame took
  // 4.2 us: 4.3 ms at Large, 0.06 of the HBM roofline; common.hpp's DPP reductions are what this kernel needed.  (Issuing the
  // next frame's gradient load early was measured and changes nothing: 4.34 -> 4.24 ms; it is not the load that is waited for.)
  for (int tt = wave; tt < nt; tt += 4) {
    float xw[C0_KW], y[8], gv[8];
#pragma unroll
    for (int k = 0; k < C0_KW; ++k) xw[k] = seg[tt * stride + k];
#pragma unroll
    for (int e = 0; e < 8; ++e) gv[e] = 0.f;
    if (act) V8<TO>::ld(g + ((long)b * T0 + t0 + tt) * C + lane * 8, gv);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = cb[e];
#pragma unroll
      for (int k = 0; k < C0_KW; ++k) a = fmaf(xw[k], w[e][k], a);
      y[e] = a; s += act ? a : 0.f;
    }
    const float mean = wave_sum(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = y[e] - mean; q += act ? d * d : 0.f; }
    const float rstd = rsqrtf(wave_sum(q) * invC + eps);
    float h[8], xh[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      xh[e] = (y[e] - mean) * rstd;
      const float zz = fmaf(xh[e], gm[e], bt[e]);
      float gp;
      if constexpr (TAB) gp = tab_eval(tab, zz); else gp = gelu_grad_f(zz);
      const float dz = act ? gv[e] * gscale * gp : 0.f;
      a1[e] += dz; a2[e] = fmaf(dz, xh[e], a2[e]);
      h[e] = dz * gm[e];
      s1 += h[e]; s2 = fmaf(h[e], xh[e], s2);
    }
    s1 = wave_sum(s1) * invC; s2 = wave_sum(s2) * invC;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float dc = act ? rstd * (h[e] - s1 - xh[e] * s2) : 0.f;
      pb[e] += dc;  // d(conv bias) = sum over frames of the conv-output gradient
#pragma unroll
      for (int k = 0; k < C0_KW; ++k) pw[e][k] = fmaf(dc, xw[k], pw[e][k]);
    }
  }
  float* out = part + ((long)b * gridDim.x + blockIdx.x) * (long)C0_LN_NQ * C;
#pragma unroll
  for (int q = 0; q < C0_LN_NQ; ++q) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e)
      red[wave * 512 + lane * 8 + e] = q == 0 ? a1[e] : q == 1 ? a2[e] : q == C0_LN_NQ - 1 ? pb[e] : pw[e][(q >= 2 && q < C0_LN_NQ - 1) ? q - 2 : 0];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) out[(long)q * C + c] = red[c] + red[512 + c] + red[1024 + c] + red[1536 + c];
  }
}

// sums the [nblk][13][C] partials: row 0 -> dbeta, row 1 -> dgamma, rows 2..11 -> dW[c][k], row 12 -> dbias (optional)
__global__ __launch_bounds__(256) void conv0_ln_bwd_finish_kernel(const float* __restrict__ part, int nblk, int C, void* dW,
                                                                  void* dgamma, void* dbeta, void* dcbias, int pdt) {
  __shared__ double red[4][64];
  const int col = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + col, n = C0_LN_NQ * C;
  double s = 0.0;
  if (i < n)
    for (int b = slice; b < nblk; b += 4) s += part[(long)b * n + i];
  red[slice][col] = s;
  __syncthreads();
  if (slice == 0 && i < n) {
    const float v = (float)(red[0][col] + red[1][col] + red[2][col] + red[3][col]);
    const int q = i / C, c = i - q * C;
    if (q == 0) st_elem(dbeta, c, pdt, v);
    else if (q == 1) st_elem(dgamma, c, pdt, v);
    else if (q == C0_LN_NQ - 1) { if (dcbias) st_elem(dcbias, c, pdt, v); }
    else st_elem(dW, (long)c * C0_KW + (q - 2), pdt, v);
  }
}

// WAVLM_CONV0_BWD_MFMA=0 keeps the VALU form of the GroupNorm-mode backward (A/B and fallback)
static const bool g_conv0_bwd_mfma = [] { const char* e = getenv("WAVLM_CONV0_BWD_MFMA"); return !(e && e[0] == '0'); }();
static const bool g_conv0_fwd_mfma = [] { const char* e = getenv("WAVLM_CONV0_FWD_MFMA"); return !(e && e[0] == '0'); }();

static inline size_t seg_floats(int tch, int stride) { return (size_t)(((tch - 1) * stride + C0_KW + 3) / 4 * 4); }

extern "C" {

uint64_t wavlm_conv0_gn_workspace_bytes(int32_t B, int64_t T, int32_t C, int32_t stride) {
  const long T0 = (T - C0_KW) / stride + 1;
  const uint64_t nchunk = (uint64_t)((T0 + C0_TCH - 1) / C0_TCH);
  (void)C;
  return (uint64_t)B * nchunk * C0_NX * sizeof(float);
}

// forward: y0[B, T0, C] and stats[B, C, 2] (mean, rstd) which backward needs
int wavlm_conv0_gn_gelu_fwd(const void* wav, int32_t wav_dtype, const void* W, const void* gamma, const void* beta,
                            int32_t param_dtype, void* out, int32_t out_dtype, float* stats, int32_t B, int64_t T,
                            int32_t C, int32_t kw, int32_t stride, float eps, void* workspace, uint64_t ws_bytes,
                            void* stream) {
  if (!wav || !W || !gamma || !beta || !out || !stats || !workspace) return WL_EINVAL;
  if (kw != C0_KW || stride < 1 || stride > 8 || C <= 0 || C > 512 || (C & 7) || B <= 0 || T < kw) return WL_EINVAL;
  if (ws_bytes < wavlm_conv0_gn_workspace_bytes(B, T, C, stride)) return WL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int T0 = (int)((T - kw) / stride + 1);
  const int nchunk = (T0 + C0_TCH - 1) / C0_TCH;
  const dim3 grid((unsigned)nchunk, (unsigned)B);
  const size_t sm_gram = seg_floats(C0_TCH, stride) * sizeof(float);
  const size_t sm_apply = seg_floats(C0_TCH, stride) * sizeof(float) + (out_dtype == WL_BF16 ? GT_N * sizeof(float2) : 0);
  float* partx = (float*)workspace;
  if (out_dtype == WL_BF16 && gelu_tab_ensure(st) != WL_OK) return WL_ELAUNCH;
  // algorithmic: 2 * kw flops per output; the waveform read once, the output written once
  WlProfScope prof(WL_PROF_CONV0_FWD, out_dtype, 2.0 * kw * B * (double)T0 * C,
                   (double)B * T * (wav_dtype == WL_BF16 ? 2 : 4) + (double)B * T0 * C * (out_dtype == WL_BF16 ? 2 : 4), st);
  int nrec = nchunk;  // partial Gram records per batch row
  int rc;
  if (wav_dtype == WL_F32) {
    WL_LAUNCH((conv0_gram_kernel<float>), grid, dim3(256), sm_gram, st, (const float*)wav, partx, (long)T, T0, (int)stride);
    rc = wl_check_launch();
  } else if (g_conv0_fwd_mfma) {  // the matrix-core form (conv0_bwd_mfma.hip): 4096 frames per record
    rc = conv0_gram_mfma_launch(wav, partx, (long)T, T0, (int)stride, (int)B, st, &nrec);
  } else {
    WL_LAUNCH((conv0_gram_kernel<bf16_t>), grid, dim3(256), sm_gram, st, (const bf16_t*)wav, partx, (long)T, T0, (int)stride);
    rc = wl_check_launch();
  }
  if (rc != WL_OK) return rc;
  const dim3 gs((unsigned)((C + 255) / 256), (unsigned)B);
  if (param_dtype == WL_F32)
    WL_LAUNCH((conv0_stats_from_gram_kernel<float>), gs, dim3(256), 0, st, partx, (const float*)W, stats, nrec, (int)C, T0, eps);
  else
    WL_LAUNCH((conv0_stats_from_gram_kernel<bf16_t>), gs, dim3(256), 0, st, partx, (const bf16_t*)W, stats, nrec, (int)C, T0, eps);
#define AP(TW, TP, TO) WL_LAUNCH((conv0_apply_kernel<TW, TP, TO>), grid, dim3(256), sm_apply, st, (const TW*)wav, \
    (const TP*)W, (const TP*)gamma, (const TP*)beta, stats, (TO*)out, (long)T, T0, (int)C, (int)stride)
  const int key = wav_dtype * 100 + param_dtype * 10 + out_dtype;
  if (key == 111 && C == 512 && g_conv0_fwd_mfma) {  // the matrix-core form of the apply pass (conv0_bwd_mfma.hip)
    const float2* tab0 = gelu_tab_get(st);
    if (!tab0) return WL_ELAUNCH;
    return conv0_gn_fwd_mfma_launch(wav, W, gamma, beta, stats, out, (long)T, T0, (int)stride, (int)B, tab0, st);
  }
  if (key == 0) AP(float, float, float);
  else if (key == 111) AP(bf16_t, bf16_t, bf16_t);
  else if (key == 11) AP(float, bf16_t, bf16_t);
  else if (key == 1) AP(float, float, bf16_t);
  else return WL_EINVAL;
#undef AP
  return wl_check_launch();
}

uint64_t wavlm_conv0_gn_bwd_workspace_bytes(int32_t B, int64_t T, int32_t C, int32_t stride) {
  const long T0 = (T - C0_KW) / stride + 1;
  const uint64_t nchunk = (uint64_t)((T0 + C0_TCH_BWD - 1) / C0_TCH_BWD);
  // chunk partials [12][C] + waveform partials [112] + per-batch-row dW [C][KW] + ab[B][C][2]
  return ((uint64_t)B * nchunk * ((uint64_t)C0_NQ * C + C0_NX) + (uint64_t)B * C * C0_KW + (uint64_t)B * C * 2) * sizeof(float);
}

// backward: dW[C, kw], dgamma[C], dbeta[C] (param dtype); g = dL/dy0 [B, T0, C]; gscale = feature_grad_mult
int wavlm_conv0_gn_gelu_bwd(const void* wav, int32_t wav_dtype, const void* W, const void* gamma, const void* beta,
                            int32_t param_dtype, const void* g, int32_t g_dtype, const float* stats, void* dW,
                            void* dgamma, void* dbeta, int32_t B, int64_t T, int32_t C, int32_t kw, int32_t stride,
                            float gscale, void* workspace, uint64_t ws_bytes, void* stream) {
  if (!wav || !W || !gamma || !beta || !g || !stats || !dW || !dgamma || !dbeta || !workspace) return WL_EINVAL;
  if (kw != C0_KW || stride < 1 || stride > 8 || C <= 0 || C > 512 || (C & 7) || B <= 0 || T < kw) return WL_EINVAL;
  if (ws_bytes < wavlm_conv0_gn_bwd_workspace_bytes(B, T, C, stride)) return WL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int T0 = (int)((T - kw) / stride + 1);
  const int nchunk = (T0 + C0_TCH_BWD - 1) / C0_TCH_BWD;
  const dim3 grid((unsigned)nchunk, (unsigned)B);
  float* part = (float*)workspace;
  float* partx = part + (long)B * nchunk * C0_NQ * C;
  float* dwb = partx + (long)B * nchunk * C0_NX;
  float* ab = dwb + (long)B * C * C0_KW;
  const size_t sm1 = (seg_floats(C0_TCH_BWD, stride) + 4 * 512) * sizeof(float) + (g_dtype == WL_BF16 ? GT_N * sizeof(float2) : 0);
  if (g_dtype == WL_BF16 && gelu_tab_ensure(st) != WL_OK) return WL_ELAUNCH;
  WlProfScope prof(WL_PROF_CONV0_BWD, g_dtype, 4.0 * C0_KW * B * (double)T0 * C,
                   (double)B * T * (wav_dtype == WL_BF16 ? 2 : 4) + (double)B * T0 * C * (g_dtype == WL_BF16 ? 2 : 4), st);
#define B1(TW, TP, TO) WL_LAUNCH((conv0_bwd_fused_kernel<TW, TP, TO>), grid, dim3(256), sm1, st, (const TW*)wav, \
    (const TP*)W, (const TP*)gamma, (const TP*)beta, stats, (const TO*)g, part, partx, (long)T, T0, (int)C, (int)stride, gscale)
  const int key = wav_dtype * 100 + param_dtype * 10 + g_dtype;
  if (key == 111 && C == 512 && g_conv0_bwd_mfma) {
    const float2* tab0 = gelu_tab_get(st);
    if (!tab0) return WL_ELAUNCH;
    const float2* tab1 = tab0 + GT_N;  // [1] = gelu'
    const int rc1 = conv0_bwd_mfma_launch(wav, W, gamma, beta, stats, g, part, partx, (long)T, T0, (int)stride, gscale, nchunk,
                                          (int)B, tab1, st);
    if (rc1 != WL_OK) return rc1;
  }
  else if (key == 0) B1(float, float, float);
  else if (key == 111) B1(bf16_t, bf16_t, bf16_t);
  else if (key == 11) B1(float, bf16_t, bf16_t);
  else if (key == 1) B1(float, float, bf16_t);
  else return WL_EINVAL;
#undef B1
  int rc = wl_check_launch();
  if (rc != WL_OK) return rc;
  const dim3 g2((unsigned)((C + 63) / 64), (unsigned)B);
  if (param_dtype == WL_F32)
    WL_LAUNCH((conv0_bwd_combine_kernel<float>), g2, dim3(256), 0, st, part, partx, stats, (const float*)W,
              (const float*)gamma, ab, dwb, nchunk, (int)C, T0);
  else
    WL_LAUNCH((conv0_bwd_combine_kernel<bf16_t>), g2, dim3(256), 0, st, part, partx, stats, (const bf16_t*)W,
              (const bf16_t*)gamma, ab, dwb, nchunk, (int)C, T0);
  WL_LAUNCH(conv0_bwd_affine_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, st, ab, dgamma, dbeta,
            (int)param_dtype, (int)B, (int)C);
  const int n = C * C0_KW;
  WL_LAUNCH(conv0_bwd_w_finish_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, dwb, (int)B, n, dW,
            (int)param_dtype);
  return wl_check_launch();
}

uint64_t wavlm_conv0_ln_bwd_workspace_bytes(int32_t B, int64_t T, int32_t C, int32_t stride) {
  const long T0 = (T - C0_KW) / stride + 1;
  const uint64_t nchunk = (uint64_t)((T0 + C0_TCH_BWD - 1) / C0_TCH_BWD);
  const uint64_t valu = (uint64_t)B * nchunk * C0_LN_NQ * C * sizeof(float);
  const uint64_t mfma = conv0_ln_bwd_mfma_workspace_bytes((int)B, (int)T0);  // whichever form runs (dtypes are not known here)
  return valu > mfma ? valu : mfma;
}

// extractor_mode "layer_norm", block 0: out[B, T0, C] = gelu(LayerNorm_C(conv0(wav)))  (WavLM/WavLM.py:403-418)
uint64_t wavlm_conv0_ln_fwd_workspace_bytes(void) { return 128 * sizeof(float); }

int wavlm_conv0_ln_gelu_fwd(const void* wav, int32_t wav_dtype, const void* W, const void* conv_bias, const void* gamma,
                            const void* beta, int32_t param_dtype, void* out, int32_t out_dtype, int32_t B, int64_t T,
                            int32_t C, int32_t kw, int32_t stride, float eps, void* workspace, uint64_t ws_bytes,
                            void* stream) {
  if (!wav || !W || !gamma || !beta || !out || !workspace) return WL_EINVAL;
  if (kw != C0_KW || stride < 1 || stride > 8 || C <= 0 || C > 512 || (C & 7) || B <= 0 || T < kw) return WL_EINVAL;
  if (ws_bytes < wavlm_conv0_ln_fwd_workspace_bytes()) return WL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int T0 = (int)((T - kw) / stride + 1);
  const dim3 grid((unsigned)((T0 + C0_TCH - 1) / C0_TCH), (unsigned)B);
  const size_t smem = seg_floats(C0_TCH, stride) * sizeof(float) + (out_dtype == WL_BF16 ? GT_N * sizeof(float2) : 0);
  if (out_dtype == WL_BF16 && gelu_tab_ensure(st) != WL_OK) return WL_ELAUNCH;
  WlProfScope prof(WL_PROF_CONV0_FWD, out_dtype, 2.0 * kw * B * (double)T0 * C,
                   (double)B * T * (wav_dtype == WL_BF16 ? 2 : 4) + (double)B * T0 * C * (out_dtype == WL_BF16 ? 2 : 4), st);
#define FW(TW, TP, TO) WL_LAUNCH((conv0_ln_fwd_kernel<TW, TP, TO>), grid, dim3(256), smem, st, (const TW*)wav, \
    (const TP*)W, (const TP*)conv_bias, (const TP*)gamma, (const TP*)beta, (TO*)out, (long)T, T0, (int)C, (int)stride, eps)
  const int key = wav_dtype * 100 + param_dtype * 10 + out_dtype;
  if (key == 111 && C == 512 && g_conv0_fwd_mfma) {  // the matrix-core form (conv0_bwd_mfma.hip); WAVLM_CONV0_FWD_MFMA=0: this file's
    const float2* tab0 = gelu_tab_get(st);
    if (!tab0) return WL_ELAUNCH;
    return conv0_ln_fwd_mfma_launch(wav, W, conv_bias, gamma, beta, out, (float*)workspace, (long)T, T0, (int)stride, (int)B, eps,
                                    tab0, st);
  }
  if (key == 0) FW(float, float, float);
  else if (key == 111) FW(bf16_t, bf16_t, bf16_t);
  else if (key == 11) FW(float, bf16_t, bf16_t);
  else if (key == 1) FW(float, float, bf16_t);
  else return WL_EINVAL;
#undef FW
  return wl_check_launch();
}

// backward of the above: dW[C, kw], dgamma[C], dbeta[C] (param dtype) from g = dL/dout; conv0 has no input gradient
int wavlm_conv0_ln_gelu_bwd(const void* wav, int32_t wav_dtype, const void* W, const void* conv_bias, const void* gamma,
                            const void* beta, int32_t param_dtype, const void* g, int32_t g_dtype, void* dW,
                            void* dconv_bias, void* dgamma, void* dbeta, int32_t B, int64_t T, int32_t C, int32_t kw,
                            int32_t stride, float eps, float gscale, void* workspace, uint64_t ws_bytes, void* stream) {
  if (!wav || !W || !gamma || !beta || !g || !dW || !dgamma || !dbeta || !workspace) return WL_EINVAL;
  if (kw != C0_KW || stride < 1 || stride > 8 || C <= 0 || C > 512 || (C & 7) || B <= 0 || T < kw) return WL_EINVAL;
  if (ws_bytes < wavlm_conv0_ln_bwd_workspace_bytes(B, T, C, stride)) return WL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int T0 = (int)((T - kw) / stride + 1);
  const int nchunk = (T0 + C0_TCH_BWD - 1) / C0_TCH_BWD;
  const dim3 grid((unsigned)nchunk, (unsigned)B);
  float* part = (float*)workspace;
  const size_t smem = (seg_floats(C0_TCH_BWD, stride) + 4 * 512) * sizeof(float) + (g_dtype == WL_BF16 ? GT_N * sizeof(float2) : 0);
  if (g_dtype == WL_BF16 && gelu_tab_ensure(st) != WL_OK) return WL_ELAUNCH;
  // algorithmic: the incoming gradient read once + the waveform; conv recompute + weight gradient = 4 * kw flops per output
  WlProfScope prof(WL_PROF_CONV0_BWD, g_dtype, 4.0 * kw * B * (double)T0 * C,
                   (double)B * T * (wav_dtype == WL_BF16 ? 2 : 4) + (double)B * T0 * C * (g_dtype == WL_BF16 ? 2 : 4), st);
  const int key = wav_dtype * 100 + param_dtype * 10 + g_dtype;
  if (key == 111 && C == 512 && g_conv0_bwd_mfma) {  // the matrix-core form (conv0_bwd_mfma.hip); WAVLM_CONV0_BWD_MFMA=0: the VALU form
    const float2* tab0 = gelu_tab_get(st);
    if (!tab0) return WL_ELAUNCH;
    return conv0_ln_bwd_mfma_launch(wav, W, conv_bias, gamma, beta, g, dW, dconv_bias, dgamma, dbeta, workspace, (long)T, T0,
                                    (int)stride, (int)B, eps, gscale, tab0 + GT_N, st);
  }
  static const int occ = (getenv("WAVLM_CONV0_LN_OCC") && getenv("WAVLM_CONV0_LN_OCC")[0] == '1') ? 1 : 2;
#define BW_O(TW, TP, TO, O) WL_LAUNCH((conv0_ln_bwd_kernel<TW, TP, TO, O>), grid, dim3(256), smem, st, (const TW*)wav, \
    (const TP*)W, (const TP*)conv_bias, (const TP*)gamma, (const TP*)beta, (const TO*)g, part, (long)T, T0, (int)C, (int)stride, eps, gscale)
#define BW(TW, TP, TO) do { if (occ == 1) BW_O(TW, TP, TO, 1); else BW_O(TW, TP, TO, 2); } while (0)
  if (key == 0) BW(float, float, float);
  else if (key == 111) BW(bf16_t, bf16_t, bf16_t);
  else if (key == 11) BW(float, bf16_t, bf16_t);
  else if (key == 1) BW(float, float, bf16_t);
  else return WL_EINVAL;
#undef BW
#undef BW_O
  int rc = wl_check_launch();
  if (rc != WL_OK) return rc;
  const int n = C0_LN_NQ * C;
  WL_LAUNCH(conv0_ln_bwd_finish_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, part, (int)(B * nchunk), (int)C, dW,
            dgamma, dbeta, dconv_bias, (int)param_dtype);
  return wl_check_launch();
}

}  // extern "C"
