// Gumbel-softmax vector quantiser (wav2vec 2.0 / UniSpeech / UniSpeech-SAT quantised targets):
// src/fairseq/modules/gumbel_vector_quantizer.py:157-213.  The weight projection x -> logits[n, G*V] is a plain
// GEMM (wavlm_gemm); these kernels do everything per (row, group) on those logits:
//   forward : argmax of the raw logits (hard one-hot -> code_perplexity, and the eval output), softmax of the raw
//             logits (-> prob_perplexity), training: y_soft = softmax((logits + gumbel) / tau), index = argmax(y_soft)
//             (the straight-through one-hot's forward value).  The reference then multiplies the [n, G, V] one-hot with
//             the codebook and sums over V: a gather of G code vectors per row (wavlm_gather_rows).
//   backward: d logits = y_soft * (d_ret - <y_soft, d_ret>) / tau  (straight-through: d y_hard := d y_soft)
//                      + softmax(raw) * (dA - <softmax(raw), dA>) / n  (codebook-diversity term through prob_perplexity)
// Reductions over rows are deterministic: every wave keeps its column sums in registers over a fixed row sequence and
// writes one partial row; the partial rows are summed by wavlm_colsum.
#include "common.hpp"
#include "../../include/wavlm_hip.h"

#define VQ_MAXC 5   // V <= 320: lane l owns codes l + 64 c
#define VQ_MAXG 4

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(v, o, 64); v = t < v ? t : v; }
  return v;
}

// uniform (0, 1) from a counter hash; gumbel = -log(-log(u)) (F.gumbel_softmax draws -log(Exp(1)), the same law)
__device__ __forceinline__ float vq_gumbel(unsigned long long seed, unsigned long long ctr) {
  unsigned lo = (unsigned)ctr ^ (unsigned)seed, hi = (unsigned)(ctr >> 32) ^ (unsigned)(seed >> 32);
  unsigned h = hash32(lo + 0x9E3779B9u * hash32(hi + 0x7F4A7C15u));
  const float u = ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);
  return -__logf(-__logf(u));
}

template <typename TL>
__global__ __launch_bounds__(256) void gumbel_vq_fwd_kernel(const TL* __restrict__ logits, const float* __restrict__ noise,
    unsigned long long seed, float inv_tau, int training, long n, int G, int V, float* __restrict__ ysoft,
    int* __restrict__ idx, float* __restrict__ part) {
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
  float accp[VQ_MAXG][VQ_MAXC], acch[VQ_MAXG][VQ_MAXC];
#pragma unroll
  for (int g = 0; g < VQ_MAXG; ++g)
#pragma unroll
    for (int c = 0; c < VQ_MAXC; ++c) { accp[g][c] = 0.f; acch[g][c] = 0.f; }
  for (long row = wid; row < n; row += nw) {
#pragma unroll
    for (int g = 0; g < VQ_MAXG; ++g) {
      if (g >= G) break;
      const TL* lr = logits + (row * G + g) * V;
      float l[VQ_MAXC];
      float mx = -INFINITY;
#pragma unroll
      for (int c = 0; c < VQ_MAXC; ++c) {
        const int v = lane + 64 * c;
        l[c] = v < V ? Elem<TL>::ld(lr + v) : -INFINITY;
        mx = fmaxf(mx, l[c]);
      }
      mx = wave_max(mx);
      int kr = 0x7fffffff;
      float sum = 0.f, e[VQ_MAXC];
#pragma unroll
      for (int c = 0; c < VQ_MAXC; ++c) {
        const int v = lane + 64 * c;
        if (v < V && l[c] == mx) kr = v < kr ? v : kr;
        e[c] = __expf(l[c] - mx);  // 0 for the padding codes
        sum += e[c];
      }
      kr = wave_min_i(kr);
      sum = wave_sum(sum);
      const float inv = 1.f / sum;
#pragma unroll
      for (int c = 0; c < VQ_MAXC; ++c) {
        accp[g][c] = fmaf(e[c], inv, accp[g][c]);
        acch[g][c] += (lane + 64 * c == kr) ? 1.f : 0.f;
      }
      int k = kr;
      if (training) {
        float z[VQ_MAXC], mz = -INFINITY;
#pragma unroll
        for (int c = 0; c < VQ_MAXC; ++c) {
          const int v = lane + 64 * c;
          float gn = 0.f;
          if (v < V) gn = noise ? noise[(row * G + g) * V + v] : vq_gumbel(seed, (unsigned long long)((row * G + g) * V + v));
          z[c] = v < V ? (l[c] + gn) * inv_tau : -INFINITY;
          mz = fmaxf(mz, z[c]);
        }
        mz = wave_max(mz);
        float sz = 0.f;
        k = 0x7fffffff;
#pragma unroll
        for (int c = 0; c < VQ_MAXC; ++c) {
          const int v = lane + 64 * c;
          if (v < V && z[c] == mz) k = v < k ? v : k;
          z[c] = __expf(z[c] - mz);
          sz += z[c];
        }
        k = wave_min_i(k);
        sz = wave_sum(sz);
        const float iz = 1.f / sz;
        float* yr = ysoft + (row * G + g) * V;
#pragma unroll
        for (int c = 0; c < VQ_MAXC; ++c) {
          const int v = lane + 64 * c;
          if (v < V) yr[v] = z[c] * iz;
        }
      }
      if (lane == 0) idx[row * G + g] = g * V + k;
    }
  }
  if (wid < nw) {
    float* pp = part + wid * (2L * G * V);
#pragma unroll
    for (int g = 0; g < VQ_MAXG; ++g) {
      if (g >= G) break;
#pragma unroll
      for (int c = 0; c < VQ_MAXC; ++c) {
        const int v = lane + 64 * c;
        if (v < V) { pp[g * V + v] = accp[g][c]; pp[G * V + g * V + v] = acch[g][c]; }
      }
    }
  }
}

// sums[2][G*V] (softmax sums, hard counts over n rows) -> out[0] = prob_perplexity, out[1] = code_perplexity,
// dA[g*V + v] = d prob_perplexity / d avg_probs[g, v] = -ppl_g * (log(a + 1e-7) + a / (a + 1e-7))
__global__ __launch_bounds__(64) void vq_perplexity_kernel(const float* __restrict__ sums, float inv_n, int G, int V,
                                                          float* __restrict__ out, float* __restrict__ dA) {
  const int lane = threadIdx.x;
  float tot_p = 0.f, tot_h = 0.f;
  for (int g = 0; g < G; ++g) {
    float hp = 0.f, hh = 0.f;
    for (int v = lane; v < V; v += 64) {
      const float a = sums[g * V + v] * inv_n, b = sums[G * V + g * V + v] * inv_n;
      hp += a * __logf(a + 1e-7f);
      hh += b * __logf(b + 1e-7f);
    }
    hp = wave_sum(hp); hh = wave_sum(hh);
    const float ppl = __expf(-hp);
    tot_p += ppl; tot_h += __expf(-hh);
    for (int v = lane; v < V; v += 64) {
      const float a = sums[g * V + v] * inv_n;
      dA[g * V + v] = -ppl * (__logf(a + 1e-7f) + a / (a + 1e-7f));
    }
  }
  if (lane == 0) { out[0] = tot_p; out[1] = tot_h; }
}

template <typename TL>
__global__ __launch_bounds__(256) void gumbel_vq_bwd_kernel(const TL* __restrict__ logits, const float* __restrict__ ysoft,
    const float* __restrict__ dret, const float* __restrict__ dA, const float* __restrict__ dppl, float inv_tau,
    float inv_n, long n, int G, int V, TL* __restrict__ dlogits) {
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
  const float gp = dppl ? dppl[0] * inv_n : 0.f;
  for (long rg = wid; rg < n * G; rg += nw) {
    const int g = (int)(rg % G);
    const long o = rg * V;
    float out[VQ_MAXC];
#pragma unroll
    for (int c = 0; c < VQ_MAXC; ++c) out[c] = 0.f;
    if (ysoft && dret) {
      float y[VQ_MAXC], d[VQ_MAXC], dot = 0.f;
#pragma unroll
      for (int c = 0; c < VQ_MAXC; ++c) {
        const int v = lane + 64 * c;
        y[c] = v < V ? ysoft[o + v] : 0.f;
        d[c] = v < V ? dret[o + v] : 0.f;
        dot = fmaf(y[c], d[c], dot);
      }
      dot = wave_sum(dot);
#pragma unroll
      for (int c = 0; c < VQ_MAXC; ++c) out[c] = y[c] * (d[c] - dot) * inv_tau;
    }
    if (dppl) {
      float l[VQ_MAXC], mx = -INFINITY;
#pragma unroll
      for (int c = 0; c < VQ_MAXC; ++c) {
        const int v = lane + 64 * c;
        l[c] = v < V ? Elem<TL>::ld(logits + o + v) : -INFINITY;
        mx = fmaxf(mx, l[c]);
      }
      mx = wave_max(mx);
      float sum = 0.f, dot = 0.f, a[VQ_MAXC];
#pragma unroll
      for (int c = 0; c < VQ_MAXC; ++c) {
        const int v = lane + 64 * c;
        l[c] = __expf(l[c] - mx);
        sum += l[c];
        a[c] = v < V ? dA[g * V + v] : 0.f;
        dot = fmaf(l[c], a[c], dot);
      }
      sum = wave_sum(sum); dot = wave_sum(dot);
      const float inv = 1.f / sum;
      dot *= inv;
#pragma unroll
      for (int c = 0; c < VQ_MAXC; ++c) out[c] = fmaf(l[c] * inv * gp, a[c] - dot, out[c]);
    }
#pragma unroll
    for (int c = 0; c < VQ_MAXC; ++c) {
      const int v = lane + 64 * c;
      if (v < V) Elem<TL>::st(dlogits + o + v, out[c]);
    }
  }
}

static int vq_grid(long n) { long g = (n + 3) / 4; if (g > 2048) g = 2048; if (g < 1) g = 1; return (int)g; }

extern "C" {

uint64_t wavlm_gumbel_vq_partial_rows(int64_t n) { return (uint64_t)vq_grid(n) * 4; }

int wavlm_gumbel_vq_fwd(const void* logits, int32_t dtype, const float* noise, uint64_t seed, float tau, int32_t training,
                        int64_t n, int32_t G, int32_t V, float* ysoft, int32_t* idx, float* part, void* stream) {
  if (!logits || !idx || !part || n <= 0 || G < 1 || G > VQ_MAXG || V < 1 || V > 64 * VQ_MAXC || !(tau > 0.f)) return WL_EINVAL;
  if (training && !ysoft) return WL_EINVAL;
  const int grid = vq_grid(n);
  if (dtype == WL_F32)
    WL_LAUNCH(gumbel_vq_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)logits, noise,
              (unsigned long long)seed, 1.f / tau, (int)training, (long)n, (int)G, (int)V, ysoft, idx, part);
  else
    WL_LAUNCH(gumbel_vq_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)logits, noise,
              (unsigned long long)seed, 1.f / tau, (int)training, (long)n, (int)G, (int)V, ysoft, idx, part);
  return wl_check_launch();
}

int wavlm_vq_perplexity(const float* sums, int64_t n, int32_t G, int32_t V, float* out, float* dA, void* stream) {
  if (!sums || !out || !dA || n <= 0 || G < 1 || V < 1) return WL_EINVAL;
  WL_LAUNCH(vq_perplexity_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, 1.f / (float)n, (int)G, (int)V, out, dA);
  return wl_check_launch();
}

int wavlm_gumbel_vq_bwd(const void* logits, int32_t dtype, const float* ysoft, const float* dret, const float* dA,
                        const float* dppl, float tau, int64_t n, int32_t G, int32_t V, void* dlogits, void* stream) {
  if (!logits || !dlogits || n <= 0 || G < 1 || G > VQ_MAXG || V < 1 || V > 64 * VQ_MAXC || !(tau > 0.f)) return WL_EINVAL;
  if ((ysoft == nullptr) != (dret == nullptr)) return WL_EINVAL;
  if (dppl && !dA) return WL_EINVAL;
  const int grid = vq_grid(n * G);
  if (dtype == WL_F32)
    WL_LAUNCH(gumbel_vq_bwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)logits, ysoft, dret,
              dA, dppl, 1.f / tau, 1.f / (float)n, (long)n, (int)G, (int)V, (float*)dlogits);
  else
    WL_LAUNCH(gumbel_vq_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)logits, ysoft, dret,
              dA, dppl, 1.f / tau, 1.f / (float)n, (long)n, (int)G, (int)V, (bf16_t*)dlogits);
  return wl_check_launch();
}

}  // extern "C"
