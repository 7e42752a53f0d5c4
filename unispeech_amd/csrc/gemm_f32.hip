// Exact-fp32 GEMM for the parity mode (BASELINE.json: "fp32 encoder activations and loss within 1e-4").
// Same descriptor, batching, K-batch reduction, split-K and epilogue as the bf16 MFMA kernel, but every
// product is an fp32 FMA accumulated in k order (what the CPU oracle computes up to summation order).
// 64x64 tile, BK = 16, 256 threads, 4x4 outputs per thread; operands addressed through (row, k) strides so
// all four layout combinations share one body.  This kernel is the checker-grade path, not the fast path.
#include "gemm_common.hpp"

#define F32_BM 64
#define F32_BN 64
#define F32_BK 16

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmP p, int transA, int transB) {
  __shared__ float As[F32_BK][F32_BM + 4];
  __shared__ float Bs[F32_BK][F32_BN + 4];
  const int tile = blockIdx.x;
  const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
  const int z = blockIdx.y, split = blockIdx.z;
  const int zo = z / p.batch_i, zi = z % p.batch_i;
  const int m0 = tm * F32_BM, n0 = tn * F32_BN;
  const float* Ab = (const float*)p.A + (long)zo * p.sA_o + (long)zi * p.sA_i;
  const float* Bb = (const float*)p.B + (long)zo * p.sB_o + (long)zi * p.sB_i;
  const long a_rs = transA ? 1 : p.lda, a_ks = transA ? p.lda : 1;
  const long b_rs = transB ? 1 : p.ldb, b_ks = transB ? p.ldb : 1;
  const int kt_per = (p.K + F32_BK - 1) / F32_BK;
  int t0, t1;
  gemm_split_range(p.KB * kt_per, p.split_k, split, t0, t1);

  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // thread computes rows ty*4.., cols tx*4..
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  for (int t = t0; t < t1; ++t) {
    const int kb = t / kt_per, k0 = (t % kt_per) * F32_BK;
    const float* a = Ab + (long)kb * p.sA_kb;
    const float* b = Bb + (long)kb * p.sB_kb;
    // 64x16 elements per operand, 256 threads -> 4 each.  Pick the thread->element map so that the
    // contiguous global dimension runs across adjacent lanes.
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = threadIdx.x + e * 256;
      int r, k;
      if (transA) { r = idx & 63; k = idx >> 6; } else { k = idx & 15; r = idx >> 4; }
      float v = 0.f;
      if (m0 + r < p.M && k0 + k < p.K) v = a[(long)(m0 + r) * a_rs + (long)(k0 + k) * a_ks];
      As[k][r] = v;
      if (transB) { r = idx & 63; k = idx >> 6; } else { k = idx & 15; r = idx >> 4; }
      v = 0.f;
      if (n0 + r < p.N && k0 + k < p.K) v = b[(long)(n0 + r) * b_rs + (long)(k0 + k) * b_ks];
      Bs[k][r] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < F32_BK; ++k) {
      float av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = As[k][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = Bs[k][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int mm = m0 + ty * 4 + i, nn = n0 + tx * 4 + j;
      if (mm < p.M && nn < p.N) gemm_store(p, z, split, mm, nn, acc[i][j]);
    }
}

__global__ __launch_bounds__(256) void gemm_f32_splitk_reduce_kernel(GemmP p, int nbatch) {
  const long mn = (long)p.M * p.N;
  const long total = mn * nbatch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int z = (int)(i / mn);
    const long r = i - (long)z * mn;
    float s = 0.f;
    for (int k = 0; k < p.split_k; ++k) s += p.ws[((long)z * p.split_k + k) * mn + r];
    gemm_epi_final(p, z / p.batch_i, z % p.batch_i, (int)(r / p.N), (int)(r % p.N), s);
  }
}

int gemm_f32_launch(const wavlm_gemm_desc* d, hipStream_t st) {
  GemmP p = make_gemm_params(d);
  const int nbatch = (d->batch_o < 1 ? 1 : d->batch_o) * p.batch_i;
  p.tiles_m = (p.M + F32_BM - 1) / F32_BM;
  p.tiles_n = (p.N + F32_BN - 1) / F32_BN;
  dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)nbatch, (unsigned)p.split_k);
  WL_LAUNCH(gemm_f32_kernel, grid, dim3(256), 0, st, p, d->transA ? 1 : 0, d->transB ? 1 : 0);
  int rc = wl_check_launch();
  if (rc != WL_OK) return rc;
  if (p.split_k > 1) {
    const long total = (long)p.M * p.N * nbatch;
    long blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
    WL_LAUNCH(gemm_f32_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, nbatch);
    rc = wl_check_launch();
  }
  return rc;
}
