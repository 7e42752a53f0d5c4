// Utterance / noise mixing of a collated waveform batch on the device (replaces the per-sample
// numpy / torch loop of src/fairseq/data/audio/utterance_mixing_dataset.py:373-438 `mixing_collated_audios`).
//
// The reference mixes IN PLACE and IN ROW ORDER: row i adds a scaled span of row c, where row c is already mixed if
// c < i, still original if c > i, and the current state of row i itself if c == i; the scale is
// sqrt(mean(row_i^2) / (mean(row_c^2) * 10^(snr/10))) with both powers taken at that moment.  All random numbers (which
// rows, which partner, span positions, SNR) are drawn on the HOST from the same numpy stream as the reference and arrive
// as a list of ops; only the arithmetic runs here.
//
// One workgroup per row, all rows in flight at once: `src` stays immutable (the "still original" state of higher
// rows), `dst` is the working copy, and a row that needs the finished state of a lower row spins on that row's flag
// (dependencies only point to lower-numbered workgroups, which the dispatcher starts first; the grid never exceeds the
// CU count, so every workgroup is resident).  A self-mix (c == i) stages the scaled span in scratch first, which is
// the reference's `.clone()`.
#include "common.hpp"
#include "../../include/wavlm_hip.h"

#define MIX_THREADS 1024
#define MIX_MAX_ROWS 256

struct MixOp {  // 8 x int32, built on the host (unispeech_amd/data.py)
  int row, kind, src, c_start, s_start, c_len, src_len;
  float gain;  // float32(10 ** (snr / 10))
};

__device__ __forceinline__ double block_sum_d(double v, double* red) {
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < MIX_THREADS / 64; ++w) s += red[w];
  return s;
}

__device__ __forceinline__ double row_sumsq(const float* __restrict__ p, long n, double* red) {
  double s = 0.0;
  for (long t = threadIdx.x; t < n; t += MIX_THREADS) { const float v = p[t]; s += (double)v * (double)v; }
  return block_sum_d(s, red);
}

__global__ __launch_bounds__(MIX_THREADS) void mix_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
    bf16_t* __restrict__ dst_lowp, int row0, int B, long T, const MixOp* __restrict__ ops, const int* __restrict__ op_begin,
    const float* __restrict__ noise, float* __restrict__ scratch, int normalize, float eps, int* __restrict__ flags) {
  __shared__ double red[MIX_THREADS / 64];
  const int i = row0 + blockIdx.x;
  float* di = dst + (long)i * T;
  const float* si = src + (long)i * T;
  for (long t = threadIdx.x; t < T; t += MIX_THREADS) di[t] = si[t];
  __syncthreads();
  const int ob = op_begin[i], oe = op_begin[i + 1];
  for (int k = ob; k < oe; ++k) {
    const MixOp op = ops[k];
    const float* p;
    long plen;
    if (op.kind == 1) { p = noise + op.src; plen = op.src_len; }  // external noise segment (offset, length)
    else {
      const int c = op.src;
      plen = T;
      if (c < i) {
        if (threadIdx.x == 0) {
          while (__hip_atomic_load(flags + c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(32);
        }
        __syncthreads();
        p = dst + (long)c * T;
      } else if (c == i) p = di;
      else p = src + (long)c * T;
    }
    // powers as the reference takes them: np.mean(x ** 2) in float32 (here: double accumulation, rounded once)
    const float ref_pow = (float)(row_sumsq(di, T, red) / (double)T);
    const float noise_pow = (p == di) ? ref_pow : (float)(row_sumsq(p, plen, red) / (double)plen);
    float scale = 0.f;
    if (noise_pow != 0.f) scale = sqrtf(ref_pow / (noise_pow * op.gain));
    const long n = op.c_len;
    if (p == di) {  // self-mix: read everything before the first write
      float* sc = scratch + (long)blockIdx.x * T;
      for (long t = threadIdx.x; t < n; t += MIX_THREADS) sc[t] = di[op.c_start + t] * scale;
      __syncthreads();
      for (long t = threadIdx.x; t < n; t += MIX_THREADS) di[op.s_start + t] += sc[t];
    } else {
      for (long t = threadIdx.x; t < n; t += MIX_THREADS) di[op.s_start + t] += p[op.c_start + t] * scale;
    }
    __syncthreads();
  }
  if (normalize && oe > ob) {  // F.layer_norm(source[i], source[i].shape): whole-row mean / biased variance, eps 1e-5
    double s = 0.0;
    for (long t = threadIdx.x; t < T; t += MIX_THREADS) s += (double)di[t];
    const double mean = block_sum_d(s, red) / (double)T;
    double q = 0.0;
    for (long t = threadIdx.x; t < T; t += MIX_THREADS) { const double d = (double)di[t] - mean; q += d * d; }
    const double var = block_sum_d(q, red) / (double)T;
    const float rstd = 1.0f / sqrtf((float)var + eps), mf = (float)mean;
    for (long t = threadIdx.x; t < T; t += MIX_THREADS) di[t] = (di[t] - mf) * rstd;
  }
  if (dst_lowp) {
    bf16_t* dl = dst_lowp + (long)i * T;
    for (long t = threadIdx.x; t < T; t += MIX_THREADS) dl[t] = f2bf(di[t]);
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flags + i, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void mix_clear_flags_kernel(int* flags, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flags[i] = 0;
}

extern "C" {

// scratch for self-mixes (one row per resident workgroup) + one completion flag per row
uint64_t wavlm_mix_workspace_bytes(int32_t B, int64_t T) {
  const uint64_t rows = (uint64_t)(B < MIX_MAX_ROWS ? B : MIX_MAX_ROWS);
  return rows * (uint64_t)T * sizeof(float) + (((uint64_t)B * sizeof(int) + 255) / 256) * 256;
}

// dst[B, T] (fp32) = src[B, T] with the ops applied in order (ops sorted by row; op_begin[B + 1] = first op of each row).
// dst_lowp (optional, bf16 [B, T]): the Trainer's cast of the waveform (trainer.py:1141-1152) in the same pass.
int wavlm_mix_utterances(const float* src, float* dst, void* dst_lowp, int32_t B, int64_t T, const int32_t* ops,
                         int32_t n_ops, const int32_t* op_begin, const float* noise, int32_t normalize, float eps,
                         void* workspace, uint64_t ws_bytes, void* stream) {
  if (!src || !dst || src == dst || B <= 0 || T <= 0 || n_ops < 0 || !op_begin || (n_ops > 0 && !ops) || !workspace)
    return WL_EINVAL;
  if (ws_bytes < wavlm_mix_workspace_bytes(B, T)) return WL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const uint64_t rows = (uint64_t)(B < MIX_MAX_ROWS ? B : MIX_MAX_ROWS);
  float* scratch = (float*)workspace;
  int* flags = (int*)((char*)workspace + rows * (uint64_t)T * sizeof(float));
  WL_LAUNCH(mix_clear_flags_kernel, dim3((B + 255) / 256), dim3(256), 0, st, flags, (int)B);
  // chunks of <= 256 rows, one workgroup per row: a chunk only waits on rows of earlier chunks (complete) or lower rows
  // of its own chunk (resident)
  for (int r0 = 0; r0 < B; r0 += MIX_MAX_ROWS) {
    const int nb = B - r0 < MIX_MAX_ROWS ? B - r0 : MIX_MAX_ROWS;
    WL_LAUNCH(mix_rows_kernel, dim3(nb), dim3(MIX_THREADS), 0, st, src, dst, (bf16_t*)dst_lowp, r0, (int)B, (long)T,
              (const MixOp*)ops, op_begin, noise, scratch, (int)normalize, eps, flags);
  }
  return wl_check_launch();
}

}  // extern "C"
