// Convolutional position embedding, weight side (WavLM/WavLM.py:514-527).
//
// pos_conv = weight_norm(Conv1d(D, D, k=K, padding=K/2, groups=G), dim=2): w[co,ci,k] = g[k] * v[co,ci,k] / ||v[:,:,k]||.
// The convolution itself runs on the MFMA GEMM as G*B overlapping-row GEMMs over a group-major activation
// copy; this file produces the two GEMM-ready weight images from (g, v) and the (g, v) gradients from the
// GEMM-layout weight gradient:
//   Wf[g][col][tap*Cg + ci]  = w[g*Cg + col, ci, tap]            (forward,   B operand, N = Cg, K = K*Cg)
//   Wb[g][ci][tap*Cg + col]  = w[g*Cg + col, ci, K-1-tap]        (backward-data: correlation with flipped taps)
#include "common.hpp"
#include "../../include/wavlm_hip.h"

#define PC_BLOCKS 512

// part[blk][k] = sum over the block's rows r=(co,ci) of a[r][k]*b[r][k]   (a == b gives squared norms).
// 512 blocks x four independent accumulators: with 64 blocks and one dependent fma chain over 576 rows per thread this
// took 240 us for 4.7 M elements.
template <typename TA, typename TB>
__global__ void pc_rowdot_partial_kernel(const TA* __restrict__ a, const TB* __restrict__ b, long rows, int K,
                                         float* __restrict__ part) {
  const int k = threadIdx.x;
  if (k >= K) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const long G = gridDim.x;
  long r = blockIdx.x;
  for (; r + 3 * G < rows; r += 4 * G) {
    s0 = fmaf(Elem<TA>::ld(a + r * K + k), Elem<TB>::ld(b + r * K + k), s0);
    s1 = fmaf(Elem<TA>::ld(a + (r + G) * K + k), Elem<TB>::ld(b + (r + G) * K + k), s1);
    s2 = fmaf(Elem<TA>::ld(a + (r + 2 * G) * K + k), Elem<TB>::ld(b + (r + 2 * G) * K + k), s2);
    s3 = fmaf(Elem<TA>::ld(a + (r + 3 * G) * K + k), Elem<TB>::ld(b + (r + 3 * G) * K + k), s3);
  }
  for (; r < rows; r += G) s0 = fmaf(Elem<TA>::ld(a + r * K + k), Elem<TB>::ld(b + r * K + k), s0);
  part[(long)blockIdx.x * K + k] = (s0 + s1) + (s2 + s3);
}

// tot[k] = sum_b part[b][k] (optionally sqrt): block = 16 k x 64 block slices, grid = ceil(K / 16)
__global__ __launch_bounds__(1024) void pc_finish_kernel(const float* __restrict__ part, int nblk, int K, float* __restrict__ tot, int do_sqrt) {
  __shared__ double red[64][17];
  const int col = threadIdx.x & 15, slice = threadIdx.x >> 4;
  const int k = blockIdx.x * 16 + col;
  double s = 0.0;
  if (k < K)
    for (int b = slice; b < nblk; b += 64) s += part[(long)b * K + k];
  red[slice][col] = s;
  __syncthreads();
  if (slice == 0 && k < K) {
    s = 0.0;
    for (int j = 0; j < 64; ++j) s += red[j][col];
    tot[k] = do_sqrt ? (float)sqrt(s) : (float)s;
  }
}

// one thread per (co, ci, k): writes both GEMM images and norm[k]
template <typename TP, typename TO>
__global__ __launch_bounds__(256) void pc_weight_kernel(const TP* __restrict__ v, const TP* __restrict__ g,
    const float* __restrict__ norm, TO* __restrict__ Wf, TO* __restrict__ Wb, int D, int Cg, int K, int layout) {
  const long total = (long)D * Cg * K;
  const long KC = (long)K * Cg;
  // element (column n, tap, channel c) of group grp: layout 0 = [n][tap][c] (B operand of the GEMM form); layout 1 = the
  // direct convolution's image [c / 8][(tap % 16) / 4][tap / 16][tap % 4][n][c % 8] (posconv_direct.hip)
  auto off = [&](int grp, int n, int tap, int c) -> long {
    if (layout == 0) return ((long)grp * Cg + n) * KC + (long)tap * Cg + c;
    const int J = K >> 4;
    return (long)grp * Cg * KC + (((((long)(c >> 3) * 4 + ((tap & 15) >> 2)) * J + (tap >> 4)) * 4 + (tap & 3)) * Cg + n) * 8 + (c & 7);
  };
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % K);
    const long r = i / K;
    const int ci = (int)(r % Cg), co = (int)(r / Cg);
    const int grp = co / Cg, col = co % Cg;
    const float nrm = norm[k];
    const float w = Elem<TP>::ld(g + k) * Elem<TP>::ld(v + i) / nrm;
    Elem<TO>::st(Wf + off(grp, col, k, ci), w);
    Elem<TO>::st(Wb + off(grp, ci, K - 1 - k, col), w);
  }
}

// dWf (GEMM layout, f32: [co][k][ci]) -> dw in (co,ci,k) order (f32 scratch), so the row-dot kernel can be reused.
// nsplit slabs of D*Cg*K elements are summed on the way (the direct weight-gradient kernel splits the batch).  One block
// per output channel co: its [K][Cg] matrix is read row-wise (coalesced, all slabs), transposed through LDS and written
// row-wise -- the one-thread-per-element gather this replaces read with a stride of Cg floats: 28 us for one slab, 117 us
// for four.
__global__ __launch_bounds__(256) void pc_unpack_dw_kernel(const float* __restrict__ dWf, float* __restrict__ dw, int D,
                                                           int Cg, int K, int nsplit) {
  extern __shared__ float pc_tile[];  // [K][Cg + 1]
  const int co = blockIdx.x;
  const long n = (long)K * Cg, total = (long)D * n;
  const float* src = dWf + (long)co * n;
  for (int i = threadIdx.x; i < (int)n; i += 256) {
    float a = src[i];
    for (int sidx = 1; sidx < nsplit; ++sidx) a += src[sidx * total + i];
    const int k = i / Cg, ci = i - k * Cg;
    pc_tile[k * (Cg + 1) + ci] = a;
  }
  __syncthreads();
  float* dst = dw + (long)co * n;
  for (int i = threadIdx.x; i < (int)n; i += 256) {
    const int ci = i / K, k = i - ci * K;
    dst[i] = pc_tile[k * (Cg + 1) + ci];
  }
}
// dv = g/n * (dw - v * S/n^2), dg[k] = S/n with S[k] = sum dw*v
template <typename TP>
__global__ __launch_bounds__(256) void pc_weight_bwd_kernel(const float* __restrict__ dw, const TP* __restrict__ v,
    const TP* __restrict__ g, const float* __restrict__ norm, const float* __restrict__ Stot,
    TP* __restrict__ dv, TP* __restrict__ dg, int D, int Cg, int K) {
  const long total = (long)D * Cg * K;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % K);
    const float S = Stot[k];
    const float n = norm[k];
    const float gk = Elem<TP>::ld(g + k);
    Elem<TP>::st(dv + i, gk / n * (dw[i] - Elem<TP>::ld(v + i) * S / (n * n)));
    if (i < K) Elem<TP>::st(dg + k, S / n);
  }
}

// activation relayout for the grouped conv: x[B,T,D] (optionally times gelu'(aux)) -> out[B,G,Tp,Cg], zero rows
// outside [left_pad, left_pad + T).  nat (optional) receives the same values in the natural [B,T,D] layout.
template <typename T>
__global__ __launch_bounds__(256) void pc_group_major_kernel(const T* __restrict__ x, const T* __restrict__ aux,
    T* __restrict__ out, T* __restrict__ nat, int B, int Tn, int D, int G, int left_pad, int Tp, int aux_is_grad) {
  const int Cg = D / G, c8n = Cg >> 3;
  const long total = (long)B * G * Tp * c8n;  // (< 2^31, checked by the launcher: the index arithmetic below is 32-bit --
  // three 64-bit divisions per 16-byte chunk were most of this copy kernel's instruction stream)
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {
    const int c8 = (int)(i % (unsigned)c8n);
    unsigned r = i / (unsigned)c8n;
    const int tp = (int)(r % (unsigned)Tp); r /= (unsigned)Tp;
    const int g = (int)(r % (unsigned)G);
    const int b = (int)(r / (unsigned)G);
    const int t = tp - left_pad;
    unsigned short o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (t >= 0 && t < Tn) {
      const long src = ((long)b * Tn + t) * D + g * Cg + c8 * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = Elem<T>::ld(x + src + e);
      if (aux) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a = Elem<T>::ld(aux + src + e);
          v[e] *= aux_is_grad ? a : gelu_grad_f(a);  // aux holds gelu'(u) (GEMM epi 3) or the pre-activation u (epi 1)
        }
      }
      if (nat) {
#pragma unroll
        for (int e = 0; e < 8; ++e) Elem<T>::st(nat + src + e, v[e]);
      }
    }
    const long dst = (((long)b * G + g) * Tp + tp) * Cg + c8 * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) Elem<T>::st(out + dst + e, v[e]);
  }
}

extern "C" {

uint64_t wavlm_posconv_weight_workspace_bytes(int32_t D, int32_t Cg, int32_t K) {
  return ((uint64_t)(PC_BLOCKS + 1) * K + (uint64_t)D * Cg * K) * sizeof(float);
}

int wavlm_posconv_weight_fwd(const void* v, const void* g, int32_t param_dtype, void* Wf, void* Wb, int32_t out_dtype,
                             float* norm, int32_t D, int32_t Cg, int32_t K, int32_t layout, void* workspace,
                             uint64_t ws_bytes, void* stream) {
  if (!v || !g || !Wf || !Wb || !norm || !workspace || D <= 0 || Cg <= 0 || K <= 0 || K > 1024 || D % Cg) return WL_EINVAL;
  if (layout != 0 && (layout != 1 || (K & 15) || (Cg & 7))) return WL_EINVAL;
  if (ws_bytes < wavlm_posconv_weight_workspace_bytes(D, Cg, K)) return WL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  const long rows = (long)D * Cg;
  const unsigned th = (unsigned)((K + 63) / 64 * 64);
  const long total = rows * K;
  long grid = (total + 255) / 256; if (grid > 4096) grid = 4096;
  if (param_dtype == WL_F32) {
    WL_LAUNCH((pc_rowdot_partial_kernel<float, float>), dim3(PC_BLOCKS), dim3(th), 0, st, (const float*)v,
                       (const float*)v, rows, (int)K, part);
    WL_LAUNCH(pc_finish_kernel, dim3((unsigned)((K + 15) / 16)), dim3(1024), 0, st, part, PC_BLOCKS, (int)K, norm, 1);
    if (out_dtype == WL_F32)
      WL_LAUNCH((pc_weight_kernel<float, float>), dim3((unsigned)grid), dim3(256), 0, st, (const float*)v,
                         (const float*)g, norm, (float*)Wf, (float*)Wb, (int)D, (int)Cg, (int)K, (int)layout);
    else
      WL_LAUNCH((pc_weight_kernel<float, bf16_t>), dim3((unsigned)grid), dim3(256), 0, st, (const float*)v,
                         (const float*)g, norm, (bf16_t*)Wf, (bf16_t*)Wb, (int)D, (int)Cg, (int)K, (int)layout);
  } else if (param_dtype == WL_BF16 && out_dtype == WL_BF16) {
    WL_LAUNCH((pc_rowdot_partial_kernel<bf16_t, bf16_t>), dim3(PC_BLOCKS), dim3(th), 0, st, (const bf16_t*)v,
                       (const bf16_t*)v, rows, (int)K, part);
    WL_LAUNCH(pc_finish_kernel, dim3((unsigned)((K + 15) / 16)), dim3(1024), 0, st, part, PC_BLOCKS, (int)K, norm, 1);
    WL_LAUNCH((pc_weight_kernel<bf16_t, bf16_t>), dim3((unsigned)grid), dim3(256), 0, st, (const bf16_t*)v,
                       (const bf16_t*)g, norm, (bf16_t*)Wf, (bf16_t*)Wb, (int)D, (int)Cg, (int)K, (int)layout);
  } else return WL_EINVAL;
  return wl_check_launch();
}

// dWf: f32 [G, Cg, K*Cg] (the weight-gradient GEMM's output); outputs dv [D,Cg,K], dg [K] in param dtype
int wavlm_posconv_weight_bwd(const float* dWf, const void* v, const void* g, const float* norm, int32_t param_dtype,
                             void* dv, void* dg, int32_t D, int32_t Cg, int32_t K, int32_t nsplit, void* workspace,
                             uint64_t ws_bytes, void* stream) {
  if (!dWf || !v || !g || !norm || !dv || !dg || !workspace || D <= 0 || Cg <= 0 || K <= 0 || K > 1024 || D % Cg || nsplit < 1)
    return WL_EINVAL;
  if ((size_t)K * (Cg + 1) * sizeof(float) > 65536) return WL_EINVAL;  // one output channel's [K][Cg] matrix sits in LDS
  if (ws_bytes < wavlm_posconv_weight_workspace_bytes(D, Cg, K)) return WL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  float* Stot = part + (long)PC_BLOCKS * K;
  float* dw = Stot + K;
  const long rows = (long)D * Cg;
  const long total = rows * K;
  long grid = (total + 255) / 256; if (grid > 4096) grid = 4096;
  const unsigned th = (unsigned)((K + 63) / 64 * 64);
  WL_LAUNCH(pc_unpack_dw_kernel, dim3((unsigned)D), dim3(256), (size_t)K * (Cg + 1) * sizeof(float), st, dWf, dw, (int)D, (int)Cg, (int)K, (int)nsplit);
  if (param_dtype == WL_F32) {
    WL_LAUNCH((pc_rowdot_partial_kernel<float, float>), dim3(PC_BLOCKS), dim3(th), 0, st, (const float*)dw,
                       (const float*)v, rows, (int)K, part);
    WL_LAUNCH(pc_finish_kernel, dim3((unsigned)((K + 15) / 16)), dim3(1024), 0, st, part, PC_BLOCKS, (int)K, Stot, 0);
    WL_LAUNCH((pc_weight_bwd_kernel<float>), dim3((unsigned)grid), dim3(256), 0, st, (const float*)dw,
                       (const float*)v, (const float*)g, norm, Stot, (float*)dv, (float*)dg, (int)D, (int)Cg, (int)K);
  } else if (param_dtype == WL_BF16) {
    WL_LAUNCH((pc_rowdot_partial_kernel<float, bf16_t>), dim3(PC_BLOCKS), dim3(th), 0, st, (const float*)dw,
                       (const bf16_t*)v, rows, (int)K, part);
    WL_LAUNCH(pc_finish_kernel, dim3((unsigned)((K + 15) / 16)), dim3(1024), 0, st, part, PC_BLOCKS, (int)K, Stot, 0);
    WL_LAUNCH((pc_weight_bwd_kernel<bf16_t>), dim3((unsigned)grid), dim3(256), 0, st, (const float*)dw,
                       (const bf16_t*)v, (const bf16_t*)g, norm, Stot, (bf16_t*)dv, (bf16_t*)dg, (int)D, (int)Cg, (int)K);
  } else return WL_EINVAL;
  return wl_check_launch();
}

int wavlm_posconv_group_major(const void* x, const void* aux, void* out, void* nat_out, int32_t B, int32_t T, int32_t D,
                              int32_t G, int32_t left_pad, int32_t Tp, int32_t dtype, int32_t aux_is_grad, void* stream) {
  if (!x || !out || B <= 0 || T <= 0 || D <= 0 || G <= 0 || D % G || ((D / G) & 7) || left_pad < 0 || Tp < left_pad + T)
    return WL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long total = (long)B * G * Tp * ((D / G) >> 3);
  if (total >= (1L << 31) - 8192L * 256) return WL_EINVAL;  // 32-bit chunk indices in the kernel (16 bytes per chunk: 32 GiB)
  long grid = (total + 255) / 256; if (grid > 8192) grid = 8192;
  if (dtype == WL_F32)
    WL_LAUNCH((pc_group_major_kernel<float>), dim3((unsigned)grid), dim3(256), 0, st, (const float*)x,
                       (const float*)aux, (float*)out, (float*)nat_out, (int)B, (int)T, (int)D, (int)G, (int)left_pad, (int)Tp, (int)aux_is_grad);
  else if (dtype == WL_BF16)
    WL_LAUNCH((pc_group_major_kernel<bf16_t>), dim3((unsigned)grid), dim3(256), 0, st, (const bf16_t*)x,
                       (const bf16_t*)aux, (bf16_t*)out, (bf16_t*)nat_out, (int)B, (int)T, (int)D, (int)G, (int)left_pad,
                       (int)Tp, (int)aux_is_grad);
  else return WL_EINVAL;
  return wl_check_launch();
}

}  // extern "C"
