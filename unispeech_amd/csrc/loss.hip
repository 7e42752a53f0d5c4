// HuBERT-style masked-prediction loss pieces.
//
// The reference builds targets[V+1, S, 256] in fp32 (7.2 GB at B=32) to take cosine similarities against
// every codebook row (src/fairseq/models/wavlm/wavlm.py:426-438, 525-535) and then a (V+1)-way cross entropy
// with class 0 (src/fairseq/criterions/wavlm_criterion.py:68-71).  Because the duplicated positive among the
// negatives is masked to -inf, that is exactly a V-way cross entropy over cos(x, E_v) / temp with class =
// label.  So: L2-normalise the projected frames and the codebook rows (this file), one [S,256]x[V,256]^T
// MFMA GEMM with alpha = 1/temp, and a fused log-softmax / NLL / accuracy / gradient row kernel (this file).
#include "common.hpp"
#include "../../include/wavlm_hip.h"

// y = x / max(||x||, eps); inv[row] = 1 / max(||x||, eps)     (torch.cosine_similarity clamps each norm)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const TI* __restrict__ x, TO* __restrict__ y,
                                                          float* __restrict__ inv, long rows, int D, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    float s = 0.f;
    for (int c = lane; c < D; c += 64) { const float v = Elem<TI>::ld(x + row * D + c); s = fmaf(v, v, s); }
    s = wave_sum(s);
    const float iv = 1.f / fmaxf(sqrtf(s), eps);
    if (lane == 0) inv[row] = iv;
    for (int c = lane; c < D; c += 64) Elem<TO>::st(y + row * D + c, Elem<TI>::ld(x + row * D + c) * iv);
  }
}
// dx = inv * (dy - y * <y, dy>)   (y = normalised row as stored)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const TO* __restrict__ dy, const TO* __restrict__ y,
    const float* __restrict__ inv, TI* __restrict__ dx, long rows, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s = fmaf(Elem<TO>::ld(y + row * D + c), Elem<TO>::ld(dy + row * D + c), s);
    s = wave_sum(s);
    const float iv = inv[row];
    for (int c = lane; c < D; c += 64)
      Elem<TI>::st(dx + row * D + c, iv * (Elem<TO>::ld(dy + row * D + c) - Elem<TO>::ld(y + row * D + c) * s));
  }
}

// per row: loss = lse(logits) - logits[target]; correct = logits[target] >= max; dlogits = w * (softmax - onehot)
template <typename TO>
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, const int* __restrict__ target,
    float* __restrict__ loss_rows, float* __restrict__ correct_rows, TO* __restrict__ dlogits, long S, int V, long ldl,
    long ldd, float weight) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long row = (long)blockIdx.x * 4 + wave; row < S; row += (long)gridDim.x * 4) {
    const float* lr = logits + row * ldl;
    float mx = -INFINITY;
    for (int c = lane; c < V; c += 64) mx = fmaxf(mx, lr[c]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = lane; c < V; c += 64) sum += expf(lr[c] - mx);
    sum = wave_sum(sum);
    // a label outside [0, V) (pad / special id leaking through, dictionary that does not match the label set) must not
    // become an out-of-bounds read: the row's loss is NaN -- loud (fairseq's NaN detector, trainer.py:840-855) where the
    // reference's index_select raises -- and its gradient row is zero
    const int t = target[row];
    const bool valid = (unsigned)t < (unsigned)V;
    const float lt = valid ? lr[t] : __builtin_nanf("");
    if (lane == 0) {
      loss_rows[row] = (mx + logf(sum)) - lt;
      correct_rows[row] = (lt >= mx) ? 1.f : 0.f;
    }
    if (dlogits) {
      const float inv = 1.f / sum;
      for (int c = lane; c < ldd; c += 64) {
        float g = 0.f;
        if (c < V && valid) g = weight * (expf(lr[c] - mx) * inv - (c == t ? 1.f : 0.f));
        Elem<TO>::st(dlogits + row * ldd + c, g);
      }
    }
  }
}

__global__ __launch_bounds__(256) void sum_partial_f32_kernel(const float* __restrict__ x, long n, double* part) {
  __shared__ double red[4];
  double s = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += x[i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void sum_finish_kernel(const double* part, int n, float* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += part[i];
    out[0] = (float)s;
  }
}


// ---- sampled-instance cosine logits (UniSpeech-SAT utterance-contrastive head, models/unispeech_sat/unispeech_sat.py:
// 487-557 + 701-737; the same shape as wav2vec 2.0's sampled negatives, models/wav2vec/wav2vec2.py:474-553) ----------
// The reference gathers the sampled rows into [N, S, C] (1.4 GB fp32 at cfg2-like sizes) and calls
// cosine_similarity; here rows are L2-normalised once and every logit is one gathered dot product.
// out[s, n] = scale * <X[s], Y[idx[s, n]]>     (one wave per s; lane l owns elements 4l .. 4l+3 of up to 4 chunks)
// mask_equal: columns n >= 1 whose gathered row equals the row of column 0 (the positive) get -inf -- wav2vec 2.0's
// `neg_is_pos` rule (models/wav2vec/wav2vec2.py:535-551) for quantised targets that repeat
template <typename T>
__global__ __launch_bounds__(256) void gather_dot_kernel(const T* __restrict__ X, const T* __restrict__ Y,
    const int* __restrict__ idx, float* __restrict__ out, long S, int N, int D, float scale, int mask_equal) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = D >> 2;  // float4 groups per row
  for (long s = (long)blockIdx.x * 4 + wave; s < S; s += (long)gridDim.x * 4) {
    float x[4][4], pz[4][4];
    const long j0 = idx[s * N];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool in = lane + 64 * c < nv;
        x[c][e] = in ? Elem<T>::ld(X + s * D + (lane + 64 * c) * 4 + e) : 0.f;
        pz[c][e] = (in && mask_equal) ? Elem<T>::ld(Y + j0 * D + (lane + 64 * c) * 4 + e) : 0.f;
      }
    for (int n = 0; n < N; ++n) {
      const long j = idx[s * N + n];
      float a = 0.f;
      bool same = true;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (lane + 64 * c < nv) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float yv = Elem<T>::ld(Y + j * D + (lane + 64 * c) * 4 + e);
            a = fmaf(x[c][e], yv, a);
            same = same && (yv == pz[c][e]);
          }
        }
      a = wave_sum(a);
      const bool all_same = mask_equal && n > 0 && __all(same);
      if (lane == 0) out[s * N + n] = all_same ? -INFINITY : a * scale;
    }
  }
}

// out[j] (+)= sum_{e in [off[j], off[j+1])} w[e] * Y[src[e]]   -- both halves of the gathered dot product's backward
// (direct: entries of row j in order; transposed: entries sorted by their gathered index)
template <typename T, typename TO>
__global__ __launch_bounds__(256) void rows_wsum_kernel(const T* __restrict__ Y, const int* __restrict__ src,
    const float* __restrict__ w, const int* __restrict__ off, TO* __restrict__ out, long rows, int D, int accumulate) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = D >> 2;
  for (long j = (long)blockIdx.x * 4 + wave; j < rows; j += (long)gridDim.x * 4) {
    float acc[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[c][e] = 0.f;
    const int e0 = off[j], e1 = off[j + 1];
    for (int q = e0; q < e1; ++q) {
      const long r = src[q];
      const float ww = w[q];
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (lane + 64 * c < nv) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[c][e] = fmaf(ww, Elem<T>::ld(Y + r * D + (lane + 64 * c) * 4 + e), acc[c][e]);
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (lane + 64 * c < nv) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          TO* o = out + j * D + (lane + 64 * c) * 4 + e;
          Elem<TO>::st(o, acc[c][e] + (accumulate ? Elem<TO>::ld(o) : 0.f));
        }
      }
  }
}

// binary cross-entropy with logits (F.binary_cross_entropy_with_logits, reduction 'none' -> caller takes the mean):
// part[block] = (sum loss, count of (logit >= 0) == target); dlogits = gscale * (sigmoid(l) - t)
__global__ __launch_bounds__(256) void bce_logits_kernel(const float* __restrict__ logits, const unsigned char* __restrict__ tgt,
                                                          float* __restrict__ dlogits, double* __restrict__ part, long n,
                                                          float gscale) {
  __shared__ double red[4][2];
  double ls = 0.0, cs = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float l = logits[i], t = tgt[i] ? 1.f : 0.f;
    const float sp = __logf(1.f + __expf(-fabsf(l)));
    ls += (double)(fmaxf(l, 0.f) - l * t + sp);
    cs += ((l >= 0.f) == (t > 0.5f)) ? 1.0 : 0.0;
    if (dlogits) dlogits[i] = gscale * (1.f / (1.f + __expf(-l)) - t);
  }
  ls = wave_sum_d(ls); cs = wave_sum_d(cs);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = ls; red[threadIdx.x >> 6][1] = cs; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    part[2 * blockIdx.x + 1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
  }
}
__global__ void bce_finish_kernel(const double* part, int nblk, float* out, double inv_n) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int i = 0; i < nblk; ++i) { a += part[2 * i]; b += part[2 * i + 1]; }
    out[0] = (float)(a * inv_n);  // mean loss
    out[1] = (float)(b * inv_n);  // accuracy
  }
}

// Gated linear unit over the last dimension: y[r][c] = a * g(b) with (a, b) = x[r][c], x[r][F + c]; backward
// da = dy * g(b), db = dy * a * g'(b).  gate: 0 sigmoid (nn.GLU = target_glu, wavlm.py:322-327, applied to the [V, F]
// label-embedding table), 1 swish b * sigmoid(b) (GLU_Linear(.., "swish"), the feed-forward block's fc1 under
// activation_fn = "glu": WavLM/modules.py:99-129, WavLM/WavLM.py:668-669), 2 relu, 3 gelu (erf), 4 bilinear (g = b).
__device__ __forceinline__ void glu_gate(int gate, float b, float& g, float& dg) {
  if (gate == 0) { const float sg = 1.f / (1.f + __expf(-b)); g = sg; dg = sg * (1.f - sg); }
  else if (gate == 1) { const float sg = 1.f / (1.f + __expf(-b)); g = b * sg; dg = sg * (1.f + b * (1.f - sg)); }
  else if (gate == 2) { g = b > 0.f ? b : 0.f; dg = b > 0.f ? 1.f : 0.f; }
  else if (gate == 3) { const float cdf = 0.5f * (1.f + erff(b * 0.70710678118654752f)); g = b * cdf; dg = cdf + b * 0.3989422804014327f * __expf(-0.5f * b * b); }
  else { g = b; dg = 1.f; }
}
__global__ __launch_bounds__(256) void glu_fwd_kernel(const void* __restrict__ x, void* __restrict__ y, long rows, int F, int dt, int gate) {
  // rows over blocks, columns over threads: no per-element 64-bit division (the feed-forward form runs on [24 k, 2 x 3072])
  for (long r = blockIdx.x; r < rows; r += gridDim.x)
    for (int c = threadIdx.x; c < F; c += 256) {
      const float a = ld_elem(x, r * 2 * F + c, dt), b = ld_elem(x, r * 2 * F + F + c, dt);
      float g, dg;
      glu_gate(gate, b, g, dg);
      st_elem(y, r * F + c, dt, a * g);
    }
}
__global__ __launch_bounds__(256) void glu_bwd_kernel(const void* __restrict__ x, const void* __restrict__ dy, void* __restrict__ dx,
                                                      long rows, int F, int dt, int gate) {
  for (long r = blockIdx.x; r < rows; r += gridDim.x)
    for (int c = threadIdx.x; c < F; c += 256) {
      const float a = ld_elem(x, r * 2 * F + c, dt), b = ld_elem(x, r * 2 * F + F + c, dt), gy = ld_elem(dy, r * F + c, dt);
      float g, dg;
      glu_gate(gate, b, g, dg);
      st_elem(dx, r * 2 * F + c, dt, gy * g);
      st_elem(dx, r * 2 * F + F + c, dt, gy * a * dg);
    }
}

// Elementwise activations of the feed-forward block other than the erf GELU (which lives in the GEMM epilogues):
// utils.get_activation_fn (src/fairseq/utils.py:533-555; WavLM/modules.py:144-160).  kind: 1 relu, 2 gelu_accurate
// 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))) (src/fairseq/modules/gelu.py:14-19), 3 tanh, 4 erf gelu.
// Backward takes the PRE-activation.
__device__ __forceinline__ void act_eval(int kind, float x, float& y, float& dy) {
  if (kind == 1) { y = x > 0.f ? x : 0.f; dy = x > 0.f ? 1.f : 0.f; }
  else if (kind == 2) {
    const float c0 = 0.7978845608028654f, c1 = 0.044715f;
    const float u = c0 * (x + c1 * x * x * x), t = tanhf(u);
    y = 0.5f * x * (1.f + t);
    dy = 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * c0 * (1.f + 3.f * c1 * x * x);
  } else if (kind == 3) { const float t = tanhf(x); y = t; dy = 1.f - t * t; }
  else { const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f)); y = x * cdf; dy = cdf + x * 0.3989422804014327f * __expf(-0.5f * x * x); }
}
__global__ __launch_bounds__(256) void act_fwd_kernel(const void* __restrict__ x, void* __restrict__ y, long n, int dt, int kind) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float v, d;
    act_eval(kind, ld_elem(x, i, dt), v, d);
    st_elem(y, i, dt, v);
  }
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const void* __restrict__ x, const void* __restrict__ dy, void* __restrict__ dx,
                                                      long n, int dt, int kind) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float v, d;
    act_eval(kind, ld_elem(x, i, dt), v, d);
    st_elem(dx, i, dt, ld_elem(dy, i, dt) * d);
  }
}

extern "C" {

int wavlm_l2norm_fwd(const void* x, int32_t x_dtype, void* y, int32_t y_dtype, float* inv_norm, int64_t rows,
                     int32_t D, float eps, void* stream) {
  if (!x || !y || !inv_norm || rows < 0 || D <= 0) return WL_EINVAL;
  if (rows == 0) return WL_OK;
  hipStream_t st = (hipStream_t)stream;
  long grid = (rows + 3) / 4; if (grid > 8192) grid = 8192;
#define L2F(TI, TO) WL_LAUNCH((l2norm_fwd_kernel<TI, TO>), dim3((unsigned)grid), dim3(256), 0, st, (const TI*)x, \
    (TO*)y, inv_norm, (long)rows, (int)D, eps)
  if (x_dtype == WL_F32 && y_dtype == WL_F32) L2F(float, float);
  else if (x_dtype == WL_BF16 && y_dtype == WL_BF16) L2F(bf16_t, bf16_t);
  else if (x_dtype == WL_F32 && y_dtype == WL_BF16) L2F(float, bf16_t);
  else return WL_EINVAL;
#undef L2F
  return wl_check_launch();
}

int wavlm_l2norm_bwd(const void* dy, const void* y, int32_t y_dtype, const float* inv_norm, void* dx, int32_t x_dtype,
                     int64_t rows, int32_t D, void* stream) {
  if (!dy || !y || !inv_norm || !dx || rows < 0 || D <= 0) return WL_EINVAL;
  if (rows == 0) return WL_OK;
  hipStream_t st = (hipStream_t)stream;
  long grid = (rows + 3) / 4; if (grid > 8192) grid = 8192;
#define L2B(TI, TO) WL_LAUNCH((l2norm_bwd_kernel<TI, TO>), dim3((unsigned)grid), dim3(256), 0, st, (const TO*)dy, \
    (const TO*)y, inv_norm, (TI*)dx, (long)rows, (int)D)
  if (x_dtype == WL_F32 && y_dtype == WL_F32) L2B(float, float);
  else if (x_dtype == WL_BF16 && y_dtype == WL_BF16) L2B(bf16_t, bf16_t);
  else if (x_dtype == WL_F32 && y_dtype == WL_BF16) L2B(float, bf16_t);
  else return WL_EINVAL;
#undef L2B
  return wl_check_launch();
}

int wavlm_ce_rows(const float* logits, const int32_t* target, float* loss_rows, float* correct_rows, void* dlogits,
                  int32_t d_dtype, int64_t S, int32_t V, int64_t ld_logits, int64_t ld_dlogits, float weight,
                  void* stream) {
  if (!logits || !target || !loss_rows || !correct_rows || S < 0 || V <= 0 || ld_logits < V) return WL_EINVAL;
  if (dlogits && ld_dlogits < V) return WL_EINVAL;
  if (S == 0) return WL_OK;
  hipStream_t st = (hipStream_t)stream;
  long grid = (S + 3) / 4; if (grid > 8192) grid = 8192;
  if (!dlogits || d_dtype == WL_F32)
    WL_LAUNCH((ce_rows_kernel<float>), dim3((unsigned)grid), dim3(256), 0, st, logits, target, loss_rows,
                       correct_rows, (float*)dlogits, (long)S, (int)V, (long)ld_logits, (long)ld_dlogits, weight);
  else if (d_dtype == WL_BF16)
    WL_LAUNCH((ce_rows_kernel<bf16_t>), dim3((unsigned)grid), dim3(256), 0, st, logits, target, loss_rows,
                       correct_rows, (bf16_t*)dlogits, (long)S, (int)V, (long)ld_logits, (long)ld_dlogits, weight);
  else return WL_EINVAL;
  return wl_check_launch();
}

uint64_t wavlm_sum_workspace_bytes(void) { return 1024 * sizeof(double); }

int wavlm_sum_f32(const float* x, int64_t n, float* out, void* workspace, uint64_t ws_bytes, void* stream) {
  if (!x || !out || !workspace || n < 0 || ws_bytes < wavlm_sum_workspace_bytes()) return WL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  long grid = (n + 256 * 8 - 1) / (256 * 8); if (grid < 1) grid = 1; if (grid > 1024) grid = 1024;
  WL_LAUNCH(sum_partial_f32_kernel, dim3((unsigned)grid), dim3(256), 0, st, x, (long)n, (double*)workspace);
  WL_LAUNCH(sum_finish_kernel, dim3(1), dim3(64), 0, st, (const double*)workspace, (int)grid, out);
  return wl_check_launch();
}

int wavlm_gather_dot(const void* X, const void* Y, int32_t dtype, const int32_t* idx, float* out, int64_t S, int32_t N,
                     int32_t D, float scale, int32_t mask_equal, void* stream) {
  if (!X || !Y || !idx || !out || S < 0 || N <= 0 || D <= 0 || (D & 3) || D > 1024) return WL_EINVAL;
  if (S == 0) return WL_OK;
  hipStream_t st = (hipStream_t)stream;
  long grid = (S + 3) / 4; if (grid > 8192) grid = 8192;
  if (dtype == WL_F32) WL_LAUNCH((gather_dot_kernel<float>), dim3((unsigned)grid), dim3(256), 0, st, (const float*)X, (const float*)Y, idx, out, (long)S, (int)N, (int)D, scale, (int)mask_equal);
  else if (dtype == WL_BF16) WL_LAUNCH((gather_dot_kernel<bf16_t>), dim3((unsigned)grid), dim3(256), 0, st, (const bf16_t*)X, (const bf16_t*)Y, idx, out, (long)S, (int)N, (int)D, scale, (int)mask_equal);
  else return WL_EINVAL;
  return wl_check_launch();
}

int wavlm_rows_wsum(const void* Y, int32_t dtype, const int32_t* src, const float* w, const int32_t* off, void* out,
                    int32_t out_dtype, int64_t rows, int32_t D, int32_t accumulate, void* stream) {
  if (!Y || !src || !w || !off || !out || rows < 0 || D <= 0 || (D & 3) || D > 1024) return WL_EINVAL;
  if (rows == 0) return WL_OK;
  hipStream_t st = (hipStream_t)stream;
  long grid = (rows + 3) / 4; if (grid > 8192) grid = 8192;
  const int key = dtype * 10 + out_dtype;
#define RW(T, TO) WL_LAUNCH((rows_wsum_kernel<T, TO>), dim3((unsigned)grid), dim3(256), 0, st, (const T*)Y, src, w, off, \
                            (TO*)out, (long)rows, (int)D, (int)accumulate)
  if (key == 0) RW(float, float);
  else if (key == 11) RW(bf16_t, bf16_t);
  else if (key == 10) RW(bf16_t, float);
  else if (key == 1) RW(float, bf16_t);
  else return WL_EINVAL;
#undef RW
  return wl_check_launch();
}

uint64_t wavlm_bce_workspace_bytes(void) { return 2 * 1024 * sizeof(double); }

/* out[0] = mean BCE-with-logits, out[1] = fraction of (logit >= 0) == target; dlogits (optional) = gscale * dloss_sum/dlogit */
int wavlm_bce_logits(const float* logits, const uint8_t* targets, float* dlogits, float* out, int64_t n, float gscale,
                     void* workspace, uint64_t ws_bytes, void* stream) {
  if (!logits || !targets || !out || !workspace || n <= 0 || ws_bytes < wavlm_bce_workspace_bytes()) return WL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  long grid = (n + 256 * 8 - 1) / (256 * 8); if (grid > 1024) grid = 1024;
  WL_LAUNCH(bce_logits_kernel, dim3((unsigned)grid), dim3(256), 0, st, logits, targets, dlogits, (double*)workspace, (long)n, gscale);
  WL_LAUNCH(bce_finish_kernel, dim3(1), dim3(64), 0, st, (const double*)workspace, (int)grid, out, 1.0 / (double)n);
  return wl_check_launch();
}

int wavlm_glu_fwd(const void* x, void* y, int64_t rows, int32_t F, int32_t dtype, int32_t gate, void* stream) {
  if (!x || !y || rows < 0 || F <= 0 || gate < 0 || gate > 4 || (dtype != WL_F32 && dtype != WL_BF16)) return WL_EINVAL;
  if (rows == 0) return WL_OK;
  long grid = rows; if (grid > 8192) grid = 8192;
  WL_LAUNCH(glu_fwd_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, y, (long)rows, (int)F, (int)dtype, (int)gate);
  return wl_check_launch();
}

int wavlm_glu_bwd(const void* x, const void* dy, void* dx, int64_t rows, int32_t F, int32_t dtype, int32_t gate, void* stream) {
  if (!x || !dy || !dx || rows < 0 || F <= 0 || gate < 0 || gate > 4 || (dtype != WL_F32 && dtype != WL_BF16)) return WL_EINVAL;
  if (rows == 0) return WL_OK;
  long grid = rows; if (grid > 8192) grid = 8192;
  WL_LAUNCH(glu_bwd_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, dy, dx, (long)rows, (int)F, (int)dtype, (int)gate);
  return wl_check_launch();
}

int wavlm_act_fwd(const void* x, void* y, int64_t n, int32_t dtype, int32_t kind, void* stream) {
  if (!x || !y || n < 0 || kind < 1 || kind > 4 || (dtype != WL_F32 && dtype != WL_BF16)) return WL_EINVAL;
  if (n == 0) return WL_OK;
  long grid = (n + 255) / 256; if (grid > 8192) grid = 8192;
  WL_LAUNCH(act_fwd_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, y, (long)n, (int)dtype, (int)kind);
  return wl_check_launch();
}

int wavlm_act_bwd(const void* x, const void* dy, void* dx, int64_t n, int32_t dtype, int32_t kind, void* stream) {
  if (!x || !dy || !dx || n < 0 || kind < 1 || kind > 4 || (dtype != WL_F32 && dtype != WL_BF16)) return WL_EINVAL;
  if (n == 0) return WL_OK;
  long grid = (n + 255) / 256; if (grid > 8192) grid = 8192;
  WL_LAUNCH(act_bwd_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, dy, dx, (long)n, (int)dtype, (int)kind);
  return wl_check_launch();
}

}  // extern "C"
