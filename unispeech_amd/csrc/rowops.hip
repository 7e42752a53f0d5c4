// Row kernels of the WavLM hot path: LayerNorm (+residual, +dropout, +GELU) forward/backward, column sums
// (bias / affine gradients), row select / gather (mask embedding, padding zero-fill, masked-frame gather),
// element-wise helpers.  All are HBM-bound: one 64-lane wave owns a row, 16-byte vector loads, row
// statistics through wave shuffles, column gradients accumulated in registers per lane and reduced
// across waves/blocks deterministically (per-block partials + a finishing kernel, no atomics).
#include "common.hpp"
#include <stdlib.h>
#include "../../include/wavlm_hip.h"

// ---- 8-wide vector access ------------------------------------------------------------------------
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
  const uint4 a = *reinterpret_cast<const uint4*>(p);
  v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
  v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
  v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
  v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
  uint4 o;
  o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
  o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = o;
}
// 8 keep-bits for elements idx .. idx+7 (idx % 8 == 0): two Philox calls
__device__ __forceinline__ unsigned keep8(unsigned long long seed, unsigned long long idx, unsigned thresh) {
  const Philox4 a = philox4x32_10(seed, idx >> 2), b = philox4x32_10(seed, (idx >> 2) + 1);
  return (a.x >= thresh ? 1u : 0u) | (a.y >= thresh ? 2u : 0u) | (a.z >= thresh ? 4u : 0u) | (a.w >= thresh ? 8u : 0u) |
         (b.x >= thresh ? 16u : 0u) | (b.y >= thresh ? 32u : 0u) | (b.z >= thresh ? 64u : 0u) | (b.w >= thresh ? 128u : 0u);
}
static inline unsigned drop_thresh(float p) {
  if (p <= 0.f) return 0u;
  double t = (double)p * 4294967296.0;
  if (t > 4294967295.0) t = 4294967295.0;
  return (unsigned)t;
}

#define LN_MAXC 4  // 64 lanes x 4 chunks x 8 = rows up to 2048 channels

// LayerNorm dropout masks (input-side and output-side): row word per (seed, row), four column words per 8-channel
// chunk (shared by both masks; they depend on both seeds), see common.hpp.  Forward and backward evaluate the same
// function, nothing is stored.
template <int NC>
__device__ __forceinline__ void ln_drop_cols(unsigned long long seed_in, unsigned long long seed_out, int lane, int nch,
                                             unsigned (&cw)[NC][4]) {
  const unsigned k = (unsigned)seed_in * 0x2545F491u + (unsigned)seed_out + 0x7F4A7C15u;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int ch = lane + 64 * c;
#pragma unroll
    for (int j = 0; j < 4; ++j) cw[c][j] = ch < nch ? hash32(k ^ (unsigned)(ch * 4 + j)) : 0u;
  }
}
__device__ __forceinline__ unsigned ln_drop_row(unsigned long long seed, long row) {
  return hash32((unsigned)(seed >> 32) + (unsigned)row * 0x27D4EB2Fu + ((unsigned)seed ^ 0x165667B1u));
}
// v[0..7] -> dropout(v) for one chunk: element 2j / 2j+1 <- low / high half of word j
__device__ __forceinline__ void ln_drop_apply(float (&v)[8], unsigned rw, const unsigned (&cw)[4], unsigned th, float sc) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned w = drop_mix(rw + cw[j]);
    v[2 * j] = (w & 0xffffu) >= th ? v[2 * j] * sc : 0.f;
    v[2 * j + 1] = (w >> 16) >= th ? v[2 * j + 1] * sc : 0.f;
  }
}

// y = dropout_out( act( LN( x + dropout_in(r) ) ) ); optionally stores s = x + dropout_in(r), mean, rstd
template <typename T, typename TP, int NC>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const T* __restrict__ r, T* __restrict__ y,
    T* __restrict__ s, float* __restrict__ mean_o, float* __restrict__ rstd_o, const TP* __restrict__ gamma,
    const TP* __restrict__ beta, long rows, int D, float eps, int act, unsigned th_in, float sc_in,
    unsigned long long seed_in, unsigned th_out, float sc_out, unsigned long long seed_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = D >> 3;
  unsigned cw[NC][4];
  if (th_in | th_out) ln_drop_cols<NC>(seed_in, seed_out, lane, nch, cw);
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    float v[NC][8];
    float sum = 0.f;
    const unsigned rw_in = th_in ? ln_drop_row(seed_in, row) : 0u, rw_out = th_out ? ln_drop_row(seed_out, row) : 0u;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch) {
        const long off = row * D + ch * 8;
        load8(x + off, v[c]);
        if (r) {
          float rv[8];
          load8(r + off, rv);
          if (th_in) ln_drop_apply(rv, rw_in, cw[c], th_in, sc_in);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[c][e] += rv[e];
          // statistics are taken on the sum as it is stored (bf16-rounded in bf16 mode): that is the
          // tensor the reference normalises, and what backward re-reads from s
          if (sizeof(T) == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[c][e] = bf2f(f2bf(v[c][e]));
          }
        }
        if (s) store8(s + off, v[c]);
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += v[c][e];
      }
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; sq += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
    if (lane == 0) { if (mean_o) mean_o[row] = mean; if (rstd_o) rstd_o[row] = rstd; }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch) {
        const long off = row * D + ch * 8;
        float g[8], b[8], o[8];
        load8(gamma + ch * 8, g);
        load8(beta + ch * 8, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float z = (v[c][e] - mean) * rstd * g[e] + b[e];
          if (act) z = gelu_f(z);
          o[e] = z;
        }
        if (th_out) ln_drop_apply(o, rw_out, cw[c], th_out, sc_out);
        store8(y + off, o);
      }
    }
  }
}

// backward of the above.  dx -> gradient of x (and of the un-dropped residual path); dr (optional) ->
// gradient of r (= dx with the input-dropout mask).  dgamma/dbeta partials: part[block][2 or 3][D]; with CS the third
// array is the column sum of the gradient of r -- the bias gradient of the nn.Linear that produced r (out_proj / fc2
// in a post-LN block), which otherwise costs a separate pass over dr.
template <typename T, typename TP, int NC, bool CS>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ s,
    const float* __restrict__ mean_i, const float* __restrict__ rstd_i, const TP* __restrict__ gamma,
    const TP* __restrict__ beta, T* __restrict__ dx, T* __restrict__ dr, const T* __restrict__ dx_add,
    float* __restrict__ part, long rows, int D, int act, unsigned th_in, float sc_in, unsigned long long seed_in, unsigned th_out, float sc_out,
    unsigned long long seed_out, float grad_scale, int dr_incl_add) {
  constexpr int NA = CS ? 3 : 2;
  __shared__ float red[4][NA][512];  // cross-wave reduce of one chunk slot (64 lanes x 8 columns) at a time
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = D >> 3;
  float ag[NC][8], ab[NC][8], gm[NC][8], bt[NC][8], ac[CS ? NC : 1][8];
#pragma unroll
  for (int c = 0; c < (CS ? NC : 1); ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) ac[c][e] = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int ch = lane + 64 * c;
#pragma unroll
    for (int e = 0; e < 8; ++e) { ag[c][e] = 0.f; ab[c][e] = 0.f; gm[c][e] = 0.f; bt[c][e] = 0.f; }
    if (ch < nch) { load8(gamma + ch * 8, gm[c]); if (act) load8(beta + ch * 8, bt[c]); }
  }
  unsigned cw[NC][4];
  if (th_in | th_out) ln_drop_cols<NC>(seed_in, seed_out, lane, nch, cw);
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const float mean = mean_i[row], rstd = rstd_i[row];
    const unsigned rw_in = th_in ? ln_drop_row(seed_in, row) : 0u, rw_out = th_out ? ln_drop_row(seed_out, row) : 0u;
    float h[NC][8], xh[NC][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch) {
        const long off = row * D + ch * 8;
        float g[8], sv[8];
        load8(dy + off, g);
        load8(s + off, sv);
        if (th_out) ln_drop_apply(g, rw_out, cw[c], th_out, sc_out);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xhat = (sv[e] - mean) * rstd;
          float ge = g[e];
          if (act) ge *= gelu_grad_f(xhat * gm[c][e] + bt[c][e]);
          ag[c][e] += ge * xhat;
          ab[c][e] += ge;
          const float hh = ge * gm[c][e];
          h[c][e] = hh; xh[c][e] = xhat;
          s1 += hh; s2 += hh * xhat;
        }
      }
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch) {
        const long off = row * D + ch * 8;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = grad_scale * rstd * (h[c][e] - s1 - xh[c][e] * s2);  // grad_scale: input gradient only, not dgamma/dbeta
        if (dx_add) {  // gradient that reaches x past the LayerNorm (the residual stream of a pre-LN block)
          float a[8], t[8];
          load8(dx_add + off, a);
#pragma unroll
          for (int e = 0; e < 8; ++e) t[e] = o[e] + a[e];
          store8(dx + off, t);
          if (dr_incl_add) {  // the sum x + dropout(r) itself continues as the residual stream: r sees the total too
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = t[e];
          }
        } else store8(dx + off, o);
        if (dr || CS) {
          if (th_in) ln_drop_apply(o, rw_in, cw[c], th_in, sc_in);
          if (dr) store8(dr + off, o);
          if constexpr (CS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) ac[c][e] += o[e];
          }
        }
      }
    }
  }
  // cross-wave reduction of the column accumulators, one chunk-slot at a time
  float* pg = part + (long)blockIdx.x * NA * D;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[wave][0][lane * 8 + e] = ag[c][e]; red[wave][1][lane * 8 + e] = ab[c][e];
      if constexpr (CS) red[wave][2][lane * 8 + e] = ac[c][e];
    }
    __syncthreads();
    // 512 columns of this slot, 256 threads -> 2 each, for both arrays
    for (int i = threadIdx.x; i < 512; i += 256) {
      const int ch = (i >> 3) + 64 * c;
      if (ch < nch) {
        const int col = ch * 8 + (i & 7);
        pg[col] = red[0][0][i] + red[1][0][i] + red[2][0][i] + red[3][0][i];
        pg[D + col] = red[0][1][i] + red[1][1][i] + red[2][1][i] + red[3][1][i];
        if constexpr (CS) pg[2 * D + col] = red[0][2][i] + red[1][2][i] + red[2][2][i] + red[3][2][i];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Full-width LayerNorm kernels: D == VEC * 64 * NC exactly (512 = 8 x 64 x 1, 768 = 4 x 64 x 3, 1024 = 8 x 64 x 2 -- every
// width of the path), so every lane owns NC full vectors and the row loop has NO branch around a memory instruction:
// the compiler can then count its s_waitcnt vmcnt (gfx950 counts stores too; behind a maybe-skipped store every wait
// degrades to vmcnt(0), i.e. "wait for the stores of the row before").  Rows are software-pipelined: the loads of the
// wave's next row are issued right after the current row has been unpacked, before its reductions and stores.  gamma /
// beta live in registers for the whole kernel.  The dropout masks are the same function of (seed, row, column) as in the
// general kernels above: word(column >> 1) half (column & 1).
template <typename T, int VEC> struct RawV;
template <> struct RawV<bf16_t, 8> {
  uint4 a;
  __device__ __forceinline__ void ld(const bf16_t* p) { a = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void get(float (&v)[8]) const {
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
    v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
  }
};
template <> struct RawV<bf16_t, 4> {
  uint2 a;
  __device__ __forceinline__ void ld(const bf16_t* p) { a = *reinterpret_cast<const uint2*>(p); }
  __device__ __forceinline__ void get(float (&v)[4]) const {
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
  }
};
template <> struct RawV<float, 8> {
  float4 a, b;
  __device__ __forceinline__ void ld(const float* p) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
  __device__ __forceinline__ void get(float (&v)[8]) const {
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
};
template <> struct RawV<float, 4> {
  float4 a;
  __device__ __forceinline__ void ld(const float* p) { a = *reinterpret_cast<const float4*>(p); }
  __device__ __forceinline__ void get(float (&v)[4]) const { v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; }
};
template <int VEC> __device__ __forceinline__ void stv(float* p, const float (&v)[VEC]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  if constexpr (VEC == 8) *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <int VEC> __device__ __forceinline__ void stv(bf16_t* p, const float (&v)[VEC]) {
  if constexpr (VEC == 8) {
    uint4 o;
    o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = o;
  } else {
    uint2 o;
    o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = o;
  }
}
template <typename TP, int VEC> __device__ __forceinline__ void ldp(const TP* p, float (&v)[VEC]) {
  RawV<TP, VEC> q; q.ld(p); q.get(v);
}
// column words of the lane's vectors: word w covers columns 2 w, 2 w + 1 (same function as ln_drop_cols)
template <int VEC, int NC>
__device__ __forceinline__ void lnf_drop_cols(unsigned long long seed_in, unsigned long long seed_out, int lane,
                                              unsigned (&cw)[NC][VEC / 2]) {
  const unsigned k = (unsigned)seed_in * 0x2545F491u + (unsigned)seed_out + 0x7F4A7C15u;
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int j = 0; j < VEC / 2; ++j) cw[c][j] = hash32(k ^ (unsigned)(((lane + 64 * c) * VEC >> 1) + j));
}
template <int VEC>
__device__ __forceinline__ void lnf_drop_apply(float (&v)[VEC], unsigned rw, const unsigned (&cw)[VEC / 2], unsigned th, float sc) {
#pragma unroll
  for (int j = 0; j < VEC / 2; ++j) {
    const unsigned w = drop_mix(rw + cw[j]);
    v[2 * j] = (w & 0xffffu) >= th ? v[2 * j] * sc : 0.f;
    v[2 * j + 1] = (w >> 16) >= th ? v[2 * j + 1] * sc : 0.f;
  }
}

// the GEMM epilogues' chord table of the normal CDF (gemm_bf16.hip / gemm_common.hpp: 2048 float4 = 4096 chords (slope,
// intercept) over [-8, 8); gelu(x) = x Phi(x), |error| 1e-6): device address, filled on first use
const float4* wl_gelu_tab4(hipStream_t st);
#define LN_GT_CELLS 4096
// HR: residual operand r present; HS: the pre-norm sum s is stored (training); GT: the activation is GELU by the chord table
// staged into LDS (bf16 output only).  The erf evaluation (~18 VALU slots per element on top of LayerNorm's ~8) made the
// LayerNorm + GELU rows of the layer_norm-mode extractor VALU-bound: 3.6 TB/s where the same kernel without it streams 4.2.
template <typename T, typename TP, int VEC, int NC, bool HR, bool HS, bool GT = false>
__global__ __launch_bounds__(256) void layernorm_fwd_full_kernel(const T* __restrict__ x, const T* __restrict__ r, T* __restrict__ y,
    T* __restrict__ s, float* __restrict__ mean_o, float* __restrict__ rstd_o, const TP* __restrict__ gamma,
    const TP* __restrict__ beta, long rows, float eps, int act, unsigned th_in, float sc_in,
    unsigned long long seed_in, unsigned th_out, float sc_out, unsigned long long seed_out, const float4* __restrict__ gtab = nullptr) {
  constexpr int D = VEC * 64 * NC;
  __shared__ float4 gt4[GT ? LN_GT_CELLS / 2 : 1];
  if constexpr (GT) {
    for (int i = threadIdx.x; i < LN_GT_CELLS / 2; i += 256) gt4[i] = gtab[i];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float gm[NC][VEC], bt[NC][VEC];
#pragma unroll
  for (int c = 0; c < NC; ++c) { ldp<TP, VEC>(gamma + (lane + 64 * c) * VEC, gm[c]); ldp<TP, VEC>(beta + (lane + 64 * c) * VEC, bt[c]); }
  unsigned cw[NC][VEC / 2];
  if (th_in | th_out) lnf_drop_cols<VEC, NC>(seed_in, seed_out, lane, cw);
  RawV<T, VEC> px[NC], pr[NC];
  const long rstep = (long)gridDim.x * 4;
  long row = (long)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  auto fetch = [&](long row_) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const long off = row_ * D + (lane + 64 * c) * VEC;
      px[c].ld(x + off);
      if constexpr (HR) pr[c].ld(r + off);
    }
  };
  fetch(row);
  for (; row < rows; row += rstep) {
    float v[NC][VEC];
    float sum = 0.f;
    const unsigned rw_in = th_in ? ln_drop_row(seed_in, row) : 0u, rw_out = th_out ? ln_drop_row(seed_out, row) : 0u;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      px[c].get(v[c]);
      if constexpr (HR) {
        float rv[VEC];
        pr[c].get(rv);
        if (th_in) lnf_drop_apply<VEC>(rv, rw_in, cw[c], th_in, sc_in);
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[c][e] += rv[e];
        // statistics are taken on the sum as it is stored (bf16-rounded in bf16 mode): that is the tensor the
        // reference normalises, and what backward re-reads from s
        if (sizeof(T) == 2) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) v[c][e] = bf2f(f2bf(v[c][e]));
        }
      }
    }
    {  // next row of this wave (clamped: the last iteration re-reads its own row instead of branching around the loads)
      const long nrow = row + rstep < rows ? row + rstep : row;
      fetch(nrow);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if constexpr (HS) { if (s) stv<VEC>(s + row * D + (lane + 64 * c) * VEC, v[c]); }  // (no residual: s IS x, not stored again)
#pragma unroll
      for (int e = 0; e < VEC; ++e) sum += v[c][e];
    }
    const float mean = wave_sum(sum) * (1.f / (float)D);
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int e = 0; e < VEC; ++e) { const float d = v[c][e] - mean; sq += d * d; }
    const float rstd = rsqrtf(wave_sum(sq) * (1.f / (float)D) + eps);
    if (HS && lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      float o[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float z = (v[c][e] - mean) * rstd * gm[c][e] + bt[c][e];
        if constexpr (GT) {
          float u = fmaf(z, LN_GT_CELLS / 16.0f, 8.0f * (LN_GT_CELLS / 16.0f));
          u = __builtin_amdgcn_fmed3f(u, 0.f, (float)(LN_GT_CELLS - 1));
          const float2 t = reinterpret_cast<const float2*>(gt4)[(int)u];
          z = z * fmaf(t.x, z, t.y);
        } else {
          if (act) z = gelu_f(z);
        }
        o[e] = z;
      }
      if (th_out) lnf_drop_apply<VEC>(o, rw_out, cw[c], th_out, sc_out);
      stv<VEC>(y + row * D + (lane + 64 * c) * VEC, o);
    }
  }
}

// CS: column sums of the residual-branch gradient (third partial array); HA: dx_add present; HD: dr is written
template <typename T, typename TP, int VEC, int NC, bool CS, bool HA, bool HD, bool GT = false>
__global__ __launch_bounds__(256) void layernorm_bwd_full_kernel(const T* __restrict__ dy, const T* __restrict__ s,
    const float* __restrict__ mean_i, const float* __restrict__ rstd_i, const TP* __restrict__ gamma,
    const TP* __restrict__ beta, T* __restrict__ dx, T* __restrict__ dr, const T* __restrict__ dx_add,
    float* __restrict__ part, long rows, int act, unsigned th_in, float sc_in, unsigned long long seed_in, unsigned th_out,
    float sc_out, unsigned long long seed_out, float grad_scale, int dr_incl_add, int dx_tn, int dx_gap,
    const float4* __restrict__ gtab = nullptr) {
  constexpr int D = VEC * 64 * NC;
  constexpr int NA = CS ? 3 : 2;
  __shared__ float red[4][NA][64 * VEC];
  // GT: gelu'(z) = Phi(z) + z phi(z) from the chord (slope ~ phi, intercept) of the normal CDF, as the GEMM epilogues do
  // (gemm_common.hpp: |error| <= 5.7e-4 absolute, a seventh of bf16's half ulp at 1) instead of the erf evaluation
  __shared__ float4 gt4[GT ? LN_GT_CELLS / 2 : 1];
  if constexpr (GT) {
    for (int i = threadIdx.x; i < LN_GT_CELLS / 2; i += 256) gt4[i] = gtab[i];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float ag[NC][VEC], ab[NC][VEC], gm[NC][VEC], bt[NC][VEC], ac[CS ? NC : 1][VEC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) { ag[c][e] = 0.f; ab[c][e] = 0.f; bt[c][e] = 0.f; if (CS) ac[c][e] = 0.f; }
    ldp<TP, VEC>(gamma + (lane + 64 * c) * VEC, gm[c]);
    if (act) ldp<TP, VEC>(beta + (lane + 64 * c) * VEC, bt[c]);
  }
  unsigned cw[NC][VEC / 2];
  if (th_in | th_out) lnf_drop_cols<VEC, NC>(seed_in, seed_out, lane, cw);
  RawV<T, VEC> pdy[NC], psv[NC], padd[NC];
  const long rstep = (long)gridDim.x * 4;
  long row = (long)blockIdx.x * 4 + wave;
  float nmean = 0.f, nrstd = 0.f;
  auto fetch = [&](long row_) __attribute__((always_inline)) {
    nmean = mean_i[row_]; nrstd = rstd_i[row_];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const long off = row_ * D + (lane + 64 * c) * VEC;
      pdy[c].ld(dy + off);
      psv[c].ld(s + off);
      if constexpr (HA) padd[c].ld(dx_add + off);
    }
  };
  if (row < rows) fetch(row);
  // dx_tn > 0: dx is written in segments of dx_tn rows with dx_gap rows between them (the zero-padded layout the conv
  // layer's data-gradient GEMMs read: row r of segment b lands at r + b * dx_gap); scalar bookkeeping, no division per row
  long ob = dx_tn ? row / dx_tn : 0, ot = dx_tn ? row - ob * dx_tn : 0;
  for (; row < rows; row += rstep) {
    const float mean = nmean, rstd = nrstd;
    const long orow = row + ob * dx_gap;
    if (dx_tn) { ot += rstep; while (ot >= dx_tn) { ot -= dx_tn; ++ob; } }
    const unsigned rw_in = th_in ? ln_drop_row(seed_in, row) : 0u, rw_out = th_out ? ln_drop_row(seed_out, row) : 0u;
    float h[NC][VEC], xh[NC][VEC], addv[HA ? NC : 1][VEC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      float g[VEC], sv[VEC];
      pdy[c].get(g); psv[c].get(sv);
      if constexpr (HA) padd[c].get(addv[c]);
      if (th_out) lnf_drop_apply<VEC>(g, rw_out, cw[c], th_out, sc_out);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float xhat = (sv[e] - mean) * rstd;
        float ge = g[e];
        if constexpr (GT) {
          const float z = fmaf(xhat, gm[c][e], bt[c][e]);
          float u = fmaf(z, LN_GT_CELLS / 16.0f, 8.0f * (LN_GT_CELLS / 16.0f));
          u = __builtin_amdgcn_fmed3f(u, 0.f, (float)(LN_GT_CELLS - 1));
          const float2 t = reinterpret_cast<const float2*>(gt4)[(int)u];
          ge *= fmaf(z, t.x, fmaf(t.x, z, t.y));
        } else {
          if (act) ge *= gelu_grad_f(xhat * gm[c][e] + bt[c][e]);
        }
        ag[c][e] += ge * xhat;
        ab[c][e] += ge;
        const float hh = ge * gm[c][e];
        h[c][e] = hh; xh[c][e] = xhat;
        s1 += hh; s2 += hh * xhat;
      }
    }
    {
      const long nrow = row + rstep < rows ? row + rstep : row;
      fetch(nrow);
    }
    s1 = wave_sum(s1) * (1.f / (float)D);
    s2 = wave_sum(s2) * (1.f / (float)D);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const long off = row * D + (lane + 64 * c) * VEC;
      const long ooff = orow * D + (lane + 64 * c) * VEC;
      float o[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) o[e] = grad_scale * rstd * (h[c][e] - s1 - xh[c][e] * s2);  // grad_scale: input gradient only
      if constexpr (HA) {  // gradient that reaches x past the LayerNorm (the residual stream of a pre-LN block)
        float t[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) t[e] = o[e] + addv[c][e];
        stv<VEC>(dx + ooff, t);
        if (dr_incl_add) {  // the sum x + dropout(r) itself continues as the residual stream: r sees the total too
#pragma unroll
          for (int e = 0; e < VEC; ++e) o[e] = t[e];
        }
      } else stv<VEC>(dx + ooff, o);
      if constexpr (HD || CS) {
        if (th_in) lnf_drop_apply<VEC>(o, rw_in, cw[c], th_in, sc_in);
        if constexpr (HD) stv<VEC>(dr + off, o);
        if constexpr (CS) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) ac[c][e] += o[e];
        }
      }
    }
  }
  // cross-wave reduction of the column accumulators, one slot (64 lanes x VEC columns) at a time
  float* pg = part + (long)blockIdx.x * NA * D;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      red[wave][0][lane * VEC + e] = ag[c][e]; red[wave][1][lane * VEC + e] = ab[c][e];
      if constexpr (CS) red[wave][2][lane * VEC + e] = ac[c][e];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * VEC; i += 256) {
      const int col = c * 64 * VEC + i;
      pg[col] = red[0][0][i] + red[1][0][i] + red[2][0][i] + red[3][0][i];
      pg[D + col] = red[0][1][i] + red[1][1][i] + red[2][1][i] + red[3][1][i];
      if constexpr (CS) pg[2 * D + col] = red[0][2][i] + red[1][2][i] + red[2][2][i] + red[3][2][i];
    }
  }
}

// out[c] (+)= sum_b part[b * stride + c].  Block = 16 columns x 64 row slices: the kernel is pure load latency (a few MB
// read through short dependent chains), so the chains are kept short (nblk / 64 loads per thread) and the grid wide
// (n / 16 blocks); slices are combined through LDS.
// blockIdx.y selects one of up to three interleaved partial arrays / outputs (LayerNorm's dgamma, dbeta and the
// residual branch's bias gradient in one launch)
__global__ __launch_bounds__(1024) void colsum_finish_kernel(const float* __restrict__ part, int nblk, long stride, int n,
                                                              void* out, int out_dtype, int accumulate, long part_y, void* out_y,
                                                              void* out_z) {
  __shared__ float red[64][17];
  const int col = threadIdx.x & 15, slice = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + col;
  if (blockIdx.y) { part += part_y * blockIdx.y; out = blockIdx.y == 1 ? out_y : out_z; }
  float s = 0.f;
  if (c < n) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = slice;
    for (; b + 192 < nblk; b += 256) {  // four independent loads in flight per thread
      s0 += part[(long)b * stride + c];
      s1 += part[(long)(b + 64) * stride + c];
      s2 += part[(long)(b + 128) * stride + c];
      s3 += part[(long)(b + 192) * stride + c];
    }
    for (; b < nblk; b += 64) s0 += part[(long)b * stride + c];
    s = (s0 + s1) + (s2 + s3);
  }
  red[slice][col] = s;
  __syncthreads();
  if (slice < 4) {  // 64 -> 4 partial sums per column
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[slice * 16 + k][col];
    red[slice * 16][col] = t;
  }
  __syncthreads();
  if (slice == 0 && c < n) {
    s = (red[0][col] + red[16][col]) + (red[32][col] + red[48][col]);
    if (accumulate) s += ld_elem(out, c, out_dtype);
    st_elem(out, c, out_dtype, s);
  }
}

// the same reduction for a LIST of segments in one launch (deferred finishing launches, common.hpp): block -> segment by
// its first block, then exactly colsum_finish_kernel's arithmetic
__global__ __launch_bounds__(1024) void colsum_finish_multi_kernel(WlFinList L) {
  __shared__ float red[64][17];
  int si = 0;
#pragma unroll 1
  for (int k = 1; k < L.n; ++k) if ((int)blockIdx.x >= L.s[k].blk0) si = k;
  const WlFinSeg& g = L.s[si];
  const float* __restrict__ part = g.part;
  const long stride = g.stride;
  const int nblk = g.nblk, n = g.n;
  const int col = threadIdx.x & 15, slice = threadIdx.x >> 4;
  const int c = ((int)blockIdx.x - g.blk0) * 16 + col;
  float s = 0.f;
  if (c < n) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = slice;
    for (; b + 192 < nblk; b += 256) {
      s0 += part[(long)b * stride + c];
      s1 += part[(long)(b + 64) * stride + c];
      s2 += part[(long)(b + 128) * stride + c];
      s3 += part[(long)(b + 192) * stride + c];
    }
    for (; b < nblk; b += 64) s0 += part[(long)b * stride + c];
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
  if (slice == 0 && c < n) {
    s = (red[0][col] + red[16][col]) + (red[32][col] + red[48][col]);
    for (int r = 0; r < g.rep; ++r) {
      const long idx = (long)r * g.rep_stride + c;
      st_elem(g.out, idx, g.out_dtype, g.accumulate ? s + ld_elem(g.out, idx, g.out_dtype) : s);
    }
  }
}

static thread_local WlFinList* g_fin_list = nullptr;
void wl_fin_defer(WlFinList* l) { g_fin_list = l; if (l) { l->n = 0; l->blocks = 0; } }
bool wl_fin_active() { return g_fin_list != nullptr; }
bool wl_fin_add(const float* part, int nblk, long stride, int n, void* out, int out_dtype, int accumulate, int rep, long rep_stride) {
  WlFinList* l = g_fin_list;
  if (!l || l->n >= WL_FIN_MAX || n <= 0) return false;
  WlFinSeg& g = l->s[l->n++];
  g.part = part; g.out = out; g.stride = stride; g.rep_stride = rep_stride; g.nblk = nblk; g.n = n; g.out_dtype = out_dtype;
  g.accumulate = accumulate; g.rep = rep < 1 ? 1 : rep; g.blk0 = l->blocks;
  l->blocks += (n + 15) / 16;
  return true;
}
int wl_fin_flush(WlFinList& l, void* stream) {
  if (l.n == 0) return WL_OK;
  WL_LAUNCH(colsum_finish_multi_kernel, dim3((unsigned)l.blocks), dim3(1024), 0, (hipStream_t)stream, l);
  const int rc = wl_check_launch();
  if (rc == WL_OK)
    for (int k = 0; k < l.n; ++k) {
      const WlFinSeg& g = l.s[k];
      if (g.accumulate) wl_notify_grad(g.out, (uint64_t)((long)(g.rep - 1) * g.rep_stride + g.n) * wl_esize(g.out_dtype), stream);
    }
  l.n = 0; l.blocks = 0;
  return rc;
}

#define CS_MAXC 8  // up to 4096 columns
// column sums of x[rows, N] (row stride ld) with optional row masks: a row is counted iff
// (!inc || inc[row]) && (!exc || !exc[row]).  part[block][N]
// (a single-launch form -- device-scope fp32 atomics into N running sums, last block converts -- measured 2x SLOWER
// than partials + finish: 53 / 78 us against 26 / 44 us at N = 768 / 3072)
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, long rows, int N, long ld,
    const unsigned char* __restrict__ inc, const unsigned char* __restrict__ exc, float* __restrict__ part) {
  __shared__ float red[4][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = N >> 3;
  float acc[CS_MAXC][8];
#pragma unroll
  for (int c = 0; c < CS_MAXC; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[c][e] = 0.f;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    if (inc && !inc[row]) continue;
    if (exc && exc[row]) continue;
#pragma unroll
    for (int c = 0; c < CS_MAXC; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch) {
        float v[8];
        load8(x + row * ld + ch * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[c][e] += v[e];
      }
    }
  }
  float* pg = part + (long)blockIdx.x * N;
#pragma unroll
  for (int c = 0; c < CS_MAXC; ++c) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave][lane * 8 + e] = acc[c][e];
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256) {
      const int ch = (i >> 3) + 64 * c;
      if (ch < nch) pg[ch * 8 + (i & 7)] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    }
  }
}

// y[row] = zero[row] ? 0 : (sel[row] ? emb (or 0 if emb == NULL) : x[row])
template <typename T, typename TP>
__global__ __launch_bounds__(256) void select_rows_kernel(const T* __restrict__ x, T* __restrict__ y,
    const unsigned char* __restrict__ sel, const TP* __restrict__ emb, const unsigned char* __restrict__ zero,
    long rows, int D) {
  const int nch = D >> 3;
  const long total = rows * nch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / nch; const int ch = (int)(i - row * nch);
    float v[8];
    if (zero && zero[row]) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    } else if (sel && sel[row]) {
      if (emb) load8(emb + ch * 8, v);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
      }
    } else {
      load8(x + row * D + ch * 8, v);
    }
    store8(y + row * D + ch * 8, v);
  }
}

// dst[i] = idx[i] >= 0 ? src[idx[i]] : 0   (rows of D elements)
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ src, const int* __restrict__ idx,
                                                           T* __restrict__ dst, long n_out, int D) {
  const int nch = D >> 3;
  const long total = n_out * nch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / nch; const int ch = (int)(i - row * nch);
    const int srow = idx[row];
    float v[8];
    if (srow >= 0) load8(src + (long)srow * D + ch * 8, v);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    store8(dst + row * D + ch * 8, v);
  }
}

// y = a * x + b * y (element-wise; mixed dtypes)
__global__ __launch_bounds__(256) void axpby_kernel(const void* x, int xdt, void* y, int ydt, long n, float a, float b) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = a * ld_elem(x, i, xdt);
    if (b != 0.f) v += b * ld_elem(y, i, ydt);
    st_elem(y, i, ydt, v);
  }
}


// y *= s[0] (device scalar): lets a backward pass apply the upstream scalar gradient without a host sync
__global__ __launch_bounds__(256) void scale_dev_kernel(void* y, int ydt, long n, const float* __restrict__ s, float extra) {
  const float a = s[0] * extra;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    st_elem(y, i, ydt, a * ld_elem(y, i, ydt));
}

// y = dropout(x) with the counter-based mask (same call regenerates the mask for the gradient)
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, long n8, unsigned th,
                                                       float sc, unsigned long long seed) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float v[8];
    load8(x + i * 8, v);
    const unsigned k = keep8(seed, (unsigned long long)i * 8, th);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = ((k >> e) & 1u) ? v[e] * sc : 0.f;
    store8(y + i * 8, v);
  }
}

// y = x + dropout(r): the residual add of a pre-LN block (no LayerNorm follows the add to fuse it into), one pass
// instead of clone + dropout + axpby (three kernels, 7 tensor passes).  Same mask as dropout_kernel for the same seed.
template <typename T>
__global__ __launch_bounds__(256) void dropout_add_kernel(const T* __restrict__ x, const T* __restrict__ r, T* __restrict__ y,
                                                           long n8, unsigned th, float sc, unsigned long long seed) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float v[8], a[8];
    load8(r + i * 8, v);
    load8(x + i * 8, a);
    if (th) {
      const unsigned k = keep8(seed, (unsigned long long)i * 8, th);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = ((k >> e) & 1u) ? v[e] * sc : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += a[e];
    store8(y + i * 8, v);
  }
}

// block partial sums of x^2 (features_pen, global gradient norm) -> part[block] (double).  16-byte loads; eight fp32
// running sums per thread (each over <= n / (8 * threads) terms) folded into a double per thread, doubles from there on.
template <typename T>
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const T* __restrict__ x, long n, double* part) {
  __shared__ double red[4];
  float a[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = 0.f;
  const long nv = n >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    float v[8];
    load8(x + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = fmaf(v[e], v[e], a[e]);
  }
  double s = ((double)a[0] + a[1]) + ((double)a[2] + a[3]) + ((double)a[4] + a[5]) + ((double)a[6] + a[7]);
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {  // tail elements
    const float v = Elem<T>::ld(x + nv * 8 + threadIdx.x);
    s += (double)v * v;
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(64) void sum_finish_d_kernel(const double* part, int n, float* out, float scale) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += part[i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) out[0] = (float)(s * scale);
}

static inline unsigned grid_for(long work_items, int per_block, unsigned cap) {
  long g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (unsigned)g;
}

// Grid caps of the LayerNorm kernels, measured at 23 968 x 768 bf16 (tools/ln_bench.py, same-box A/B of builds): backward
// (+ finish) 256 blocks 52 us, 512: 36 us, 768: 43 us, 1024: 41 us, 2048: 42 us, 4096: 55 us -- two blocks per CU halve the
// partial-sum traffic of 1024 and still fill the memory pipeline; forward with dropout 512: 54 us, 1024: 42 us.
#ifndef LN_BWD_BLOCKS
#define LN_BWD_BLOCKS 512
#endif
#ifndef LN_FWD_DROP_BLOCKS
#define LN_FWD_DROP_BLOCKS 1024
#endif
#define CS_BLOCKS 512

// Operand images of the extractor's convolution weights W[Cout][Cin][k] for ALL layers of the stack in one launch
// (blockIdx.y = layer): the forward GEMM's Wf[co][kk * Cin + ci] and, per stride phase r, the data-gradient GEMM's
// Wb_r[ci][j * Cout + co] with the phase's taps newest first (tap kk = r + s * (J_r - 1 - j)), the s images back to back.
// Replaces six permute copies and twelve flip + copy pairs of torch per step (41 launches of ~5 us).
__global__ __launch_bounds__(256) void conv_weights_relayout_kernel(wavlm_conv_relayout_desc d) {
  const int l = blockIdx.y;
  const int Cout = d.Cout[l], Cin = d.Cin[l], k = d.k[l], st = d.s[l];
  const long n = (long)Cout * Cin * k;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const int kk = (int)(e % k);
    const long t = e / k;
    const int ci = (int)(t % Cin), co = (int)(t / Cin);
    const float v = ld_elem(d.W[l], e, d.dtype);
    st_elem(d.Wf[l], (long)co * (k * Cin) + (long)kk * Cin + ci, d.dtype, v);
    if (d.Wb[l]) {
      const int r = kk % st, jj = kk / st;
      const int Jr = (k - r + st - 1) / st;
      long base = 0;  // images of the phases before r: Cin * J_r' * Cout each
      for (int q = 0; q < r; ++q) base += (long)Cin * ((k - q + st - 1) / st) * Cout;
      st_elem(d.Wb[l], base + (long)ci * (Jr * Cout) + (long)(Jr - 1 - jj) * Cout + co, d.dtype, v);
    }
  }
}
// the inverse for the weight gradients: G[co][ci][kk] (+)= dWf[co][kk * Cin + ci] (d.W = G, d.Wf = dWf), all layers at once
__global__ __launch_bounds__(256) void conv_wgrad_scatter_kernel(wavlm_conv_relayout_desc d, int accumulate) {
  const int l = blockIdx.y;
  const int Cout = d.Cout[l], Cin = d.Cin[l], k = d.k[l];
  const long n = (long)Cout * Cin * k;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const int kk = (int)(e % k);
    const long t = e / k;
    const int ci = (int)(t % Cin), co = (int)(t / Cin);
    float v = ld_elem(d.Wf[l], (long)co * (k * Cin) + (long)kk * Cin + ci, d.dtype);
    if (accumulate) v += ld_elem(d.W[l], e, d.dtype);
    st_elem(const_cast<void*>(d.W[l]), e, d.dtype, v);
  }
}

extern "C" {

int wavlm_abi_version(void) { return WAVLM_HIP_ABI_VERSION; }
static double g_ln_prof_bytes = 0.0;
// WAVLM_LN_FULL=0: the general kernels for every width (same-build A/B, tools/ln_bench.py)
static bool ln_full_enabled() { static const bool on = !(getenv("WAVLM_LN_FULL") && getenv("WAVLM_LN_FULL")[0] == '0'); return on; }

int wavlm_layernorm_fwd(const void* x, const void* r, void* y, void* s, float* mean, float* rstd, const void* gamma,
                        const void* beta, int64_t rows, int32_t D, float eps, int32_t dtype, int32_t param_dtype,
                        int32_t act, float p_in, uint64_t seed_in, float p_out, uint64_t seed_out, void* stream) {
  if (!x || !y || !gamma || !beta || rows < 0 || D <= 0 || (D & 7) || D > LN_MAXC * 512) return WL_EINVAL;
  if (rows == 0) return WL_OK;
  hipStream_t st = (hipStream_t)stream;
  const unsigned ti = drop_thresh16(p_in), to = drop_thresh16(p_out);
  const float si = drop_scale16(ti), so = drop_scale16(to);
  {
    const double es = dtype == WL_BF16 ? 2.0 : 4.0;  // x (+ r) read, y (+ s) written
    g_ln_prof_bytes = (double)rows * D * es * (2.0 + (r ? 1.0 : 0.0) + (s ? 1.0 : 0.0));
  }
  WlProfScope prof(WL_PROF_LN_FWD, dtype, 8.0 * rows * D, g_ln_prof_bytes, st);
  const unsigned grid = grid_for(rows, 4, (ti | to) ? LN_FWD_DROP_BLOCKS : 8192);  // with dropout: several rows per wave amortise the column words
  // the path's own widths run the branch-free, software-pipelined kernels (a few rows per wave: the pipeline needs them)
  // (HS = the row statistics are kept; without a residual the pre-norm sum is x itself and s stays NULL -- until round 3 that
  // case fell through to the general kernels: the extractor's LayerNorms of layer_norm-mode models, the encoder's first one)
  if (ln_full_enabled() && (D == 512 || D == 768 || D == 1024) && (mean == nullptr) == (rstd == nullptr) && (!s || mean)) {
    // grid cap: 1024 blocks up to 131 k rows (the encoder: measured at 23 968 rows), then one block per 128 rows up to 8192 --
    // the extractor LayerNorms of layer_norm-mode models run over up to 1.02 M rows: 618 us at 1024 blocks, 558 at 8192
    // (tools/ln_conv_bench.py, profiles/r05/ln_conv_sweep.txt)
    static const int cap_env = getenv("WAVLM_LN_FWD_BLOCKS") ? atoi(getenv("WAVLM_LN_FWD_BLOCKS")) : 0;
    long cap = cap_env > 0 ? cap_env : rows / 128;
    if (cap_env <= 0) { if (cap < 1024) cap = 1024; if (cap > 8192) cap = 8192; }
    const unsigned gridf = grid_for(rows, 4, (unsigned)cap);
    // LayerNorm + GELU rows of the layer_norm-mode extractor (bf16, D = 512, no residual, no dropout): GELU by table
    static const bool gt_on = !(getenv("WAVLM_LN_GELU_TAB") && getenv("WAVLM_LN_GELU_TAB")[0] == '0');
    if (gt_on && act == 1 && dtype == WL_BF16 && param_dtype == WL_BF16 && D == 512 && !r && !(ti | to)) {
      const float4* gtab = wl_gelu_tab4(st);
      if (!gtab) return WL_ELAUNCH;
      if (mean)
        WL_LAUNCH((layernorm_fwd_full_kernel<bf16_t, bf16_t, 8, 1, false, true, true>), dim3(gridf), dim3(256), 0, st, (const bf16_t*)x,
                  (const bf16_t*)r, (bf16_t*)y, (bf16_t*)s, mean, rstd, (const bf16_t*)gamma, (const bf16_t*)beta, (long)rows, eps, (int)act,
                  ti, si, (unsigned long long)seed_in, to, so, (unsigned long long)seed_out, gtab);
      else
        WL_LAUNCH((layernorm_fwd_full_kernel<bf16_t, bf16_t, 8, 1, false, false, true>), dim3(gridf), dim3(256), 0, st, (const bf16_t*)x,
                  (const bf16_t*)r, (bf16_t*)y, (bf16_t*)s, mean, rstd, (const bf16_t*)gamma, (const bf16_t*)beta, (long)rows, eps, (int)act,
                  ti, si, (unsigned long long)seed_in, to, so, (unsigned long long)seed_out, gtab);
      return wl_check_launch();
    }
#define LNF_K(T, TP, VEC, NCS, HR, HS) WL_LAUNCH((layernorm_fwd_full_kernel<T, TP, VEC, NCS, HR, HS>), dim3(gridf), dim3(256), 0, st, \
    (const T*)x, (const T*)r, (T*)y, (T*)s, mean, rstd, (const TP*)gamma, (const TP*)beta, (long)rows, eps, (int)act, ti, si, \
    (unsigned long long)seed_in, to, so, (unsigned long long)seed_out)
#define LNF_W(T, TP, HR, HS) do { if (D == 512) LNF_K(T, TP, 8, 1, HR, HS); else if (D == 768) LNF_K(T, TP, 4, 3, HR, HS); else LNF_K(T, TP, 8, 2, HR, HS); } while (0)
#define LNF(T, TP) do { if (r) { if (mean) LNF_W(T, TP, true, true); else LNF_W(T, TP, true, false); } \
                        else { if (mean) LNF_W(T, TP, false, true); else LNF_W(T, TP, false, false); } } while (0)
    if (dtype == WL_F32 && param_dtype == WL_F32) LNF(float, float);
    else if (dtype == WL_BF16 && param_dtype == WL_BF16) LNF(bf16_t, bf16_t);
    else if (dtype == WL_BF16 && param_dtype == WL_F32) LNF(bf16_t, float);
    else return WL_EINVAL;
#undef LNF
#undef LNF_W
#undef LNF_K
    return wl_check_launch();
  }
  // chunk slots per lane are a template parameter: registers (and occupancy) follow the actual row width
#define LN_FWD_N(T, TP, NCS) WL_LAUNCH((layernorm_fwd_kernel<T, TP, NCS>), dim3(grid), dim3(256), 0, st, (const T*)x, \
    (const T*)r, (T*)y, (T*)s, mean, rstd, (const TP*)gamma, (const TP*)beta, (long)rows, (int)D, eps, (int)act, ti, si, \
    (unsigned long long)seed_in, to, so, (unsigned long long)seed_out)
#define LN_FWD(T, TP) do { if (D <= 512) LN_FWD_N(T, TP, 1); else if (D <= 1024) LN_FWD_N(T, TP, 2); else LN_FWD_N(T, TP, 4); } while (0)
  if (dtype == WL_F32 && param_dtype == WL_F32) LN_FWD(float, float);
  else if (dtype == WL_BF16 && param_dtype == WL_BF16) LN_FWD(bf16_t, bf16_t);
  else if (dtype == WL_BF16 && param_dtype == WL_F32) LN_FWD(bf16_t, float);
  else return WL_EINVAL;
#undef LN_FWD
#undef LN_FWD_N
  return wl_check_launch();
}

// backward grid: LN_BWD_BLOCKS (512) up to 131 k rows, then one block per 256 rows up to 4096 (1.02 M rows x 512: 926 us at
// 512 blocks, 737 at 4096 incl. the finish over 4096 partial rows; profiles/r05/ln_conv_sweep.txt); WAVLM_LN_BWD_BLOCKS fixes it
#define LN_BWD_BLOCKS_MAX 4096
static int ln_bwd_blocks_env() { static const int n = getenv("WAVLM_LN_BWD_BLOCKS") ? atoi(getenv("WAVLM_LN_BWD_BLOCKS")) : 0; return n > 0 && n <= LN_BWD_BLOCKS_MAX ? n : 0; }
static int ln_bwd_blocks(long rows) {
  if (ln_bwd_blocks_env()) return ln_bwd_blocks_env();
  long n = rows / 256;
  if (n < LN_BWD_BLOCKS) n = LN_BWD_BLOCKS;
  if (n > LN_BWD_BLOCKS_MAX) n = LN_BWD_BLOCKS_MAX;
  return (int)n;
}
// (sized for the largest grid: the row count is not known here)
uint64_t wavlm_layernorm_bwd_workspace_bytes(int32_t D) { return (uint64_t)LN_BWD_BLOCKS_MAX * 3 * D * sizeof(float); }

int wavlm_layernorm_bwd(const void* dy, const void* s, const float* mean, const float* rstd, const void* gamma,
                        const void* beta, void* dx, void* dr, const void* dx_add, void* dgamma, void* dbeta,
                        void* dr_colsum, int64_t rows, int32_t D, int32_t dtype, int32_t param_dtype, int32_t act,
                        float p_in, uint64_t seed_in, float p_out, uint64_t seed_out, float grad_scale,
                        int32_t accumulate_params, int32_t dr_incl_add, void* workspace, uint64_t ws_bytes,
                        void* stream) {
  return wavlm_layernorm_bwd_seg(dy, s, mean, rstd, gamma, beta, dx, dr, dx_add, dgamma, dbeta, dr_colsum, rows, D, dtype,
                                 param_dtype, act, p_in, seed_in, p_out, seed_out, grad_scale, accumulate_params, dr_incl_add,
                                 0, 0, workspace, ws_bytes, stream);
}

int wavlm_layernorm_bwd_seg(const void* dy, const void* s, const float* mean, const float* rstd, const void* gamma,
                            const void* beta, void* dx, void* dr, const void* dx_add, void* dgamma, void* dbeta,
                            void* dr_colsum, int64_t rows, int32_t D, int32_t dtype, int32_t param_dtype, int32_t act,
                            float p_in, uint64_t seed_in, float p_out, uint64_t seed_out, float grad_scale,
                            int32_t accumulate_params, int32_t dr_incl_add, int32_t dx_seg_rows, int32_t dx_seg_gap,
                            void* workspace, uint64_t ws_bytes, void* stream) {
  if (!dy || !s || !mean || !rstd || !gamma || !dx || !dgamma || !dbeta || !workspace) return WL_EINVAL;
  if (dx_seg_rows < 0 || dx_seg_gap < 0 || (dx_seg_rows == 0 && dx_seg_gap != 0)) return WL_EINVAL;
  // the segmented dx layout exists in the kernels of the path's own widths only
  if (dx_seg_rows && !(ln_full_enabled() && (D == 512 || D == 768 || D == 1024))) return WL_EINVAL;
  if (rows <= 0 || D <= 0 || (D & 7) || D > LN_MAXC * 512) return WL_EINVAL;
  if (act && !beta) return WL_EINVAL;
  if (ws_bytes < wavlm_layernorm_bwd_workspace_bytes(D)) return WL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const unsigned ti = drop_thresh16(p_in), to = drop_thresh16(p_out);
  const float si = drop_scale16(ti), so = drop_scale16(to);
  const double es_p = dtype == WL_BF16 ? 2.0 : 4.0;  // dy, s (+ dx_add) read, dx (+ dr) written
  WlProfScope prof(WL_PROF_LN_BWD, dtype, 12.0 * rows * D,
                   (double)rows * D * es_p * (3.0 + (dx_add ? 1.0 : 0.0) + ((dr && dr != dx) ? 1.0 : 0.0)), st);
  const unsigned grid = grid_for(rows, 4, (unsigned)ln_bwd_blocks(rows));
  float* part = (float*)workspace;
  static const bool gt_on = !(getenv("WAVLM_LN_GELU_TAB") && getenv("WAVLM_LN_GELU_TAB")[0] == '0');
  if (ln_full_enabled() && gt_on && act == 1 && dtype == WL_BF16 && param_dtype == WL_BF16 && D == 512 && !dr && !dx_add && !dr_colsum &&
      !(ti | to)) {
    // LayerNorm + GELU rows of the layer_norm-mode extractor: gelu' by table
    const float4* gtab = wl_gelu_tab4(st);
    if (!gtab) return WL_ELAUNCH;
    WL_LAUNCH((layernorm_bwd_full_kernel<bf16_t, bf16_t, 8, 1, false, false, false, true>), dim3(grid), dim3(256), 0, st,
              (const bf16_t*)dy, (const bf16_t*)s, mean, rstd, (const bf16_t*)gamma, (const bf16_t*)beta, (bf16_t*)dx, (bf16_t*)dr,
              (const bf16_t*)dx_add, part, (long)rows, (int)act, ti, si, (unsigned long long)seed_in, to, so,
              (unsigned long long)seed_out, grad_scale, (int)dr_incl_add, (int)dx_seg_rows, (int)dx_seg_gap, gtab);
  } else if (ln_full_enabled() && (D == 512 || D == 768 || D == 1024)) {
#define LNB_K(T, TP, VEC, NCS, CSF, HA, HD) WL_LAUNCH((layernorm_bwd_full_kernel<T, TP, VEC, NCS, CSF, HA, HD>), dim3(grid), dim3(256), 0, st, \
    (const T*)dy, (const T*)s, mean, rstd, (const TP*)gamma, (const TP*)beta, (T*)dx, (T*)dr, (const T*)dx_add, part, (long)rows, (int)act, \
    ti, si, (unsigned long long)seed_in, to, so, (unsigned long long)seed_out, grad_scale, (int)dr_incl_add, (int)dx_seg_rows, (int)dx_seg_gap)
#define LNB_W(T, TP, CSF, HA, HD) do { if (D == 512) LNB_K(T, TP, 8, 1, CSF, HA, HD); else if (D == 768) LNB_K(T, TP, 4, 3, CSF, HA, HD); \
                                       else LNB_K(T, TP, 8, 2, CSF, HA, HD); } while (0)
#define LNB_D(T, TP, CSF, HA) do { if (dr) LNB_W(T, TP, CSF, HA, true); else LNB_W(T, TP, CSF, HA, false); } while (0)
#define LNB_A(T, TP, CSF) do { if (dx_add) LNB_D(T, TP, CSF, true); else LNB_D(T, TP, CSF, false); } while (0)
#define LNB(T, TP) do { if (dr_colsum) LNB_A(T, TP, true); else LNB_A(T, TP, false); } while (0)
    if (dtype == WL_F32 && param_dtype == WL_F32) LNB(float, float);
    else if (dtype == WL_BF16 && param_dtype == WL_BF16) LNB(bf16_t, bf16_t);
    else if (dtype == WL_BF16 && param_dtype == WL_F32) LNB(bf16_t, float);
    else return WL_EINVAL;
#undef LNB
#undef LNB_A
#undef LNB_D
#undef LNB_W
#undef LNB_K
  } else {
#define LN_BWD_C(T, TP, NCS, CSF) WL_LAUNCH((layernorm_bwd_kernel<T, TP, NCS, CSF>), dim3(grid), dim3(256), 0, st, (const T*)dy, \
    (const T*)s, mean, rstd, (const TP*)gamma, (const TP*)beta, (T*)dx, (T*)dr, (const T*)dx_add, part, (long)rows, (int)D, (int)act, ti, si, \
    (unsigned long long)seed_in, to, so, (unsigned long long)seed_out, grad_scale, (int)dr_incl_add)
#define LN_BWD_N(T, TP, NCS) do { if (dr_colsum) LN_BWD_C(T, TP, NCS, true); else LN_BWD_C(T, TP, NCS, false); } while (0)
#define LN_BWD(T, TP) do { if (D <= 512) LN_BWD_N(T, TP, 1); else if (D <= 1024) LN_BWD_N(T, TP, 2); else LN_BWD_N(T, TP, 4); } while (0)
  if (dtype == WL_F32 && param_dtype == WL_F32) LN_BWD(float, float);
  else if (dtype == WL_BF16 && param_dtype == WL_BF16) LN_BWD(bf16_t, bf16_t);
  else if (dtype == WL_BF16 && param_dtype == WL_F32) LN_BWD(bf16_t, float);
  else return WL_EINVAL;
#undef LN_BWD
#undef LN_BWD_N
#undef LN_BWD_C
  }
  int rc = wl_check_launch();
  if (rc != WL_OK) return rc;
  const unsigned g2 = (unsigned)((D + 15) / 16);
  const int na = dr_colsum ? 3 : 2;
  // a caller that collects finishing launches (wl_fin_defer: one encoder block's backward) gets the three sums as
  // segments of its one launch; the flush also reports them to the gradient listener
  if (wl_fin_add(part, (int)grid, (long)(na * D), D, dgamma, param_dtype, accumulate_params)) {
    bool ok = wl_fin_add(part + D, (int)grid, (long)(na * D), D, dbeta, param_dtype, accumulate_params);
    if (ok && dr_colsum) ok = wl_fin_add(part + 2 * D, (int)grid, (long)(na * D), D, dr_colsum, param_dtype, accumulate_params);
    if (ok) return WL_OK;
    return WL_EINVAL;   // (a full list: the caller sized WL_FIN_MAX for its sequence)
  }
  WL_LAUNCH(colsum_finish_kernel, dim3(g2, na), dim3(1024), 0, st, part, (int)grid, (long)(na * D), (int)D, dgamma,
                     (int)param_dtype, (int)accumulate_params, (long)D, dbeta, dr_colsum);
  rc = wl_check_launch();
  if (rc == WL_OK && accumulate_params) {
    const uint64_t nb = (uint64_t)D * wl_esize(param_dtype);
    wl_notify_grad(dgamma, nb, stream); wl_notify_grad(dbeta, nb, stream);
    if (dr_colsum) wl_notify_grad(dr_colsum, nb, stream);
  }
  return rc;
}

}  // extern "C"
// out[c] (+)= sum over nblk partial rows of n floats (row stride n): the finishing launch on its own, for kernels that
// produce the partial rows themselves (attention backward: q|k|v bias gradient)
int wl_colsum_finish(const float* part, int nblk, int n, void* out, int out_dtype, int accumulate, hipStream_t st) {
  if (wl_fin_add(part, nblk, (long)n, n, out, out_dtype, accumulate)) return WL_OK;   // deferred to the caller's one launch
  WL_LAUNCH(colsum_finish_kernel, dim3((unsigned)((n + 15) / 16)), dim3(1024), 0, st, part, nblk, (long)n, n, out, out_dtype,
            accumulate, 0L, (void*)nullptr, (void*)nullptr);
  return wl_check_launch();
}
extern "C" {

uint64_t wavlm_colsum_workspace_bytes(int32_t N) { return (uint64_t)CS_BLOCKS * N * sizeof(float); }

int wavlm_colsum(const void* x, int64_t rows, int32_t N, int64_t ld, int32_t dtype, const uint8_t* include_mask,
                 const uint8_t* exclude_mask, void* out, int32_t out_dtype, int32_t accumulate, void* workspace,
                 uint64_t ws_bytes, void* stream) {
  if (!x || !out || !workspace || rows < 0 || N <= 0 || (N & 7) || N > CS_MAXC * 512 || (ld & 7)) return WL_EINVAL;
  if (ws_bytes < wavlm_colsum_workspace_bytes(N)) return WL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = grid_for(rows > 0 ? rows : 1, 4, CS_BLOCKS);
  float* part = (float*)workspace;
  if (dtype == WL_F32)
    WL_LAUNCH((colsum_partial_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)x, (long)rows, (int)N,
                       (long)ld, include_mask, exclude_mask, part);
  else if (dtype == WL_BF16)
    WL_LAUNCH((colsum_partial_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (long)rows, (int)N,
                       (long)ld, include_mask, exclude_mask, part);
  else return WL_EINVAL;
  int rc = wl_check_launch();
  if (rc != WL_OK) return rc;
  if (wl_fin_add(part, (int)grid, (long)N, N, out, out_dtype, accumulate)) return WL_OK;   // deferred (the flush notifies)
  WL_LAUNCH(colsum_finish_kernel, dim3((unsigned)((N + 15) / 16)), dim3(1024), 0, st, part, (int)grid, (long)N,
                     (int)N, out, (int)out_dtype, (int)accumulate, 0L, (void*)nullptr, (void*)nullptr);
  rc = wl_check_launch();
  if (rc == WL_OK && accumulate) wl_notify_grad(out, (uint64_t)N * wl_esize(out_dtype), stream);
  return rc;
}

int wavlm_select_rows(const void* x, void* y, const uint8_t* sel, const void* emb, const uint8_t* zero, int64_t rows,
                      int32_t D, int32_t dtype, int32_t emb_dtype, void* stream) {
  if (!x || !y || rows < 0 || D <= 0 || (D & 7)) return WL_EINVAL;
  if (rows == 0) return WL_OK;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = grid_for(rows * (D >> 3), 256, 4096);
#define SEL(T, TP) WL_LAUNCH((select_rows_kernel<T, TP>), dim3(grid), dim3(256), 0, st, (const T*)x, (T*)y, sel, \
    (const TP*)emb, zero, (long)rows, (int)D)
  if (dtype == WL_F32 && emb_dtype == WL_F32) SEL(float, float);
  else if (dtype == WL_BF16 && emb_dtype == WL_BF16) SEL(bf16_t, bf16_t);
  else if (dtype == WL_BF16 && emb_dtype == WL_F32) SEL(bf16_t, float);
  else return WL_EINVAL;
#undef SEL
  return wl_check_launch();
}

int wavlm_gather_rows(const void* src, const int32_t* idx, void* dst, int64_t n_out, int32_t D, int32_t dtype,
                      void* stream) {
  if (!src || !idx || !dst || n_out < 0 || D <= 0 || (D & 7)) return WL_EINVAL;
  if (n_out == 0) return WL_OK;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = grid_for(n_out * (D >> 3), 256, 4096);
  if (dtype == WL_F32)
    WL_LAUNCH((gather_rows_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)src, idx, (float*)dst,
                       (long)n_out, (int)D);
  else if (dtype == WL_BF16)
    WL_LAUNCH((gather_rows_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)src, idx, (bf16_t*)dst,
                       (long)n_out, (int)D);
  else return WL_EINVAL;
  return wl_check_launch();
}

int wavlm_conv_weights_relayout(const wavlm_conv_relayout_desc* d, void* stream) {
  if (!d || d->n_layers < 1 || d->n_layers > WL_CONV_RELAYOUT_MAX || (d->dtype != WL_F32 && d->dtype != WL_BF16)) return WL_EINVAL;
  long mx = 0;
  for (int l = 0; l < d->n_layers; ++l) {
    if (!d->W[l] || !d->Wf[l] || d->Cout[l] <= 0 || d->Cin[l] <= 0 || d->k[l] <= 0 || d->s[l] <= 0 || d->s[l] > d->k[l]) return WL_EINVAL;
    const long n = (long)d->Cout[l] * d->Cin[l] * d->k[l];
    if (n > mx) mx = n;
  }
  const unsigned gx = grid_for(mx, 256, 8192);
  WL_LAUNCH(conv_weights_relayout_kernel, dim3(gx, (unsigned)d->n_layers), dim3(256), 0, (hipStream_t)stream, *d);
  return wl_check_launch();
}

int wavlm_conv_wgrad_scatter(const wavlm_conv_relayout_desc* d, int32_t accumulate, void* stream) {
  if (!d || d->n_layers < 1 || d->n_layers > WL_CONV_RELAYOUT_MAX || (d->dtype != WL_F32 && d->dtype != WL_BF16)) return WL_EINVAL;
  long mx = 0;
  for (int l = 0; l < d->n_layers; ++l) {
    if (!d->W[l] || !d->Wf[l] || d->Cout[l] <= 0 || d->Cin[l] <= 0 || d->k[l] <= 0) return WL_EINVAL;
    const long n = (long)d->Cout[l] * d->Cin[l] * d->k[l];
    if (n > mx) mx = n;
  }
  const unsigned gx = grid_for(mx, 256, 8192);
  WL_LAUNCH(conv_wgrad_scatter_kernel, dim3(gx, (unsigned)d->n_layers), dim3(256), 0, (hipStream_t)stream, *d, (int)accumulate);
  const int rc = wl_check_launch();
  if (rc == WL_OK && accumulate)
    for (int l = 0; l < d->n_layers; ++l)
      wl_notify_grad(d->W[l], (uint64_t)d->Cout[l] * d->Cin[l] * d->k[l] * wl_esize(d->dtype), stream);
  return rc;
}

int wavlm_axpby(const void* x, int32_t x_dtype, void* y, int32_t y_dtype, int64_t n, float a, float b, void* stream) {
  if (!x || !y || n < 0) return WL_EINVAL;
  if (n == 0) return WL_OK;
  WL_LAUNCH(axpby_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, (int)x_dtype, y,
                     (int)y_dtype, (long)n, a, b);
  return wl_check_launch();
}

int wavlm_scale_dev(void* y, int32_t dtype, int64_t n, const float* scalar, float extra, void* stream) {
  if (!y || !scalar || n < 0) return WL_EINVAL;
  if (n == 0) return WL_OK;
  WL_LAUNCH(scale_dev_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, y, (int)dtype,
                     (long)n, scalar, extra);
  return wl_check_launch();
}

int wavlm_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, int32_t dtype, void* stream) {
  if (!x || !y || n < 0 || (n & 7) || p < 0.f || p >= 1.f) return WL_EINVAL;
  if (n == 0) return WL_OK;
  hipStream_t st = (hipStream_t)stream;
  const unsigned th = drop_thresh(p);
  const float sc = 1.f / (1.f - p);
  const unsigned grid = grid_for(n >> 3, 256, 8192);
  if (dtype == WL_F32)
    WL_LAUNCH((dropout_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)x, (float*)y, (long)(n >> 3), th,
                       sc, (unsigned long long)seed);
  else if (dtype == WL_BF16)
    WL_LAUNCH((dropout_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y,
                       (long)(n >> 3), th, sc, (unsigned long long)seed);
  else return WL_EINVAL;
  return wl_check_launch();
}

// y = x + dropout(r, p, seed)  (p = 0: plain add); n % 8 == 0, 16-byte aligned pointers
int wavlm_dropout_add(const void* x, const void* r, void* y, int64_t n, float p, uint64_t seed, int32_t dtype, void* stream) {
  if (!x || !r || !y || n < 0 || (n & 7) || p < 0.f || p >= 1.f) return WL_EINVAL;
  if (n == 0) return WL_OK;
  hipStream_t st = (hipStream_t)stream;
  const unsigned th = p > 0.f ? drop_thresh(p) : 0u;
  const float sc = 1.f / (1.f - p);
  const unsigned grid = grid_for(n >> 3, 256, 8192);
  if (dtype == WL_F32)
    WL_LAUNCH((dropout_add_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)x, (const float*)r, (float*)y,
              (long)(n >> 3), th, sc, (unsigned long long)seed);
  else if (dtype == WL_BF16)
    WL_LAUNCH((dropout_add_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)r, (bf16_t*)y,
              (long)(n >> 3), th, sc, (unsigned long long)seed);
  else return WL_EINVAL;
  return wl_check_launch();
}

uint64_t wavlm_sumsq_workspace_bytes(void) { return 1024 * sizeof(double); }

// out[0] = scale * sum(x^2)
int wavlm_sumsq(const void* x, int32_t dtype, int64_t n, float scale, float* out, void* workspace, uint64_t ws_bytes,
                void* stream) {
  if (!x || !out || !workspace || n <= 0 || ws_bytes < wavlm_sumsq_workspace_bytes()) return WL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = grid_for(n, 256 * 16, 1024);
  if ((((uintptr_t)x) & 15) != 0) return WL_EINVAL;
  if (dtype == WL_F32) WL_LAUNCH((sumsq_partial_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)x, (long)n, (double*)workspace);
  else if (dtype == WL_BF16) WL_LAUNCH((sumsq_partial_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (long)n, (double*)workspace);
  else return WL_EINVAL;
  WL_LAUNCH(sum_finish_d_kernel, dim3(1), dim3(64), 0, st, (const double*)workspace, (int)grid, out, scale);
  return wl_check_launch();
}

}  // extern "C"
