// bf16 MFMA GEMM for gfx950: C = epi(alpha * A.B^T + bias), fp32 accumulate.
//
// One kernel family carries every dense contraction of the WavLM hot path (SURVEY.md 8(a) rows
// A,E,G,J,K,L): linears, the strided Conv1d stack (overlapping-row A, lda < K, no im2col), the
// grouped pos_conv, the attention batched products and the cosine-logit product, plus all their
// backward contractions.  The only thing that changes between them is how an operand is laid out
// in HBM, so the two operand loaders are the design:
//   * K-contiguous operand  ([rows][K]): 16-B global loads -> ds_write_b128, 8 lanes per 128-B row
//   * K-strided operand     ([K][rows], weight-/activation-gradient forms): each lane loads a
//     4(k) x 8(row) block with four 16-B loads, transposes it in registers and stores eight
//     ds_write_b64 -- lanes of a 16-lane group walk k, so the stores are bank-conflict-free.
// Both fill the same LDS image: [rows][64 k] bf16, 128-B rows, 16-B chunks XOR-swizzled with
// ((row >> 1) & 7) so that the ds_read_b128 fragment reads of v_mfma_f32_32x32x16_bf16 (lane l ->
// row l&31, chunk 2*kk + (l>>5)) hit 16 distinct 16-B slots per lane group.
// Tile 128 x {128,64} x 64, 4 waves (2x2), each wave FM x FN 32x32 accumulators, register-staged
// double buffering (global loads for tile t+1 are in flight while tile t is multiplied), one
// barrier per K tile.
#include "gemm_common.hpp"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define GEMM_BK 64

union U4 { uint4 v; bf16x8_t b; unsigned u[4]; };

__device__ __forceinline__ unsigned lds_off(int row, int chunk) {
  return (unsigned)row * 128u + (unsigned)((chunk ^ ((row >> 1) & 7)) << 4);
}

// ---- K-contiguous operand: tile [ROWS][64], global row stride ld --------------------------------
template <int ROWS>
__device__ __forceinline__ void load_kc(const bf16_t* __restrict__ base, long ld, int rows_valid, int k_valid,
                                        uint4 (&r)[ROWS / 32]) {
  const int t = threadIdx.x;
  const int row = t >> 3, kk = (t & 7) * 8;
#pragma unroll
  for (int ps = 0; ps < ROWS / 32; ++ps) {
    const int rr = row + ps * 32;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (rr < rows_valid) {
      const bf16_t* src = base + (long)rr * ld + kk;
      if (kk + 8 <= k_valid) {
        v = *reinterpret_cast<const uint4*>(src);
      } else if (kk < k_valid) {
        unsigned short e[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = (kk + i < k_valid) ? src[i] : (unsigned short)0;
        v.x = e[0] | ((unsigned)e[1] << 16); v.y = e[2] | ((unsigned)e[3] << 16);
        v.z = e[4] | ((unsigned)e[5] << 16); v.w = e[6] | ((unsigned)e[7] << 16);
      }
    }
    r[ps] = v;
  }
}
template <int ROWS>
__device__ __forceinline__ void store_kc(unsigned char* lds, const uint4 (&r)[ROWS / 32]) {
  const int t = threadIdx.x;
  const int row = t >> 3, ch = t & 7;
#pragma unroll
  for (int ps = 0; ps < ROWS / 32; ++ps) {
    const int rr = row + ps * 32;
    *reinterpret_cast<uint4*>(lds + lds_off(rr, ch)) = r[ps];
  }
}

// ---- K-strided operand: global tile [64 k][ROWS] (rows contiguous), stride ld between k ---------
template <int ROWS>
__device__ __forceinline__ void load_ks(const bf16_t* __restrict__ base, long ld, int rows_valid, int k_valid,
                                        uint4 (&r)[4]) {
  static_assert(ROWS <= 128, "one pass covers at most 128 rows");
  const int t = threadIdx.x;
  const int kb4 = t & 15, nb = (t >> 6) * 4 + ((t >> 4) & 3);
  const int n = nb * 8;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = kb4 * 4 + i;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (nb < ROWS / 8 && k < k_valid && n < rows_valid) {
      const bf16_t* src = base + (long)k * ld + n;
      if (n + 8 <= rows_valid) {
        v = *reinterpret_cast<const uint4*>(src);
      } else {
        unsigned short e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = (n + j < rows_valid) ? src[j] : (unsigned short)0;
        v.x = e[0] | ((unsigned)e[1] << 16); v.y = e[2] | ((unsigned)e[3] << 16);
        v.z = e[4] | ((unsigned)e[5] << 16); v.w = e[6] | ((unsigned)e[7] << 16);
      }
    }
    r[i] = v;
  }
}
template <int ROWS>
__device__ __forceinline__ void store_ks(unsigned char* lds, const uint4 (&r)[4]) {
  const int t = threadIdx.x;
  const int kb4 = t & 15, nb = (t >> 6) * 4 + ((t >> 4) & 3);
  if (nb >= ROWS / 8) return;
  const unsigned a0[4] = {r[0].x, r[0].y, r[0].z, r[0].w};
  const unsigned a1[4] = {r[1].x, r[1].y, r[1].z, r[1].w};
  const unsigned a2[4] = {r[2].x, r[2].y, r[2].z, r[2].w};
  const unsigned a3[4] = {r[3].x, r[3].y, r[3].z, r[3].w};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = j >> 1;
    uint2 o;
    if ((j & 1) == 0) {
      o.x = (a0[c] & 0xffffu) | (a1[c] << 16);
      o.y = (a2[c] & 0xffffu) | (a3[c] << 16);
    } else {
      o.x = (a0[c] >> 16) | (a1[c] & 0xffff0000u);
      o.y = (a2[c] >> 16) | (a3[c] & 0xffff0000u);
    }
    const int row = nb * 8 + j;
    *reinterpret_cast<uint2*>(lds + lds_off(row, kb4 >> 1) + ((kb4 & 1) << 3)) = o;
  }
}

template <bool TR, int ROWS> struct Stage {
  uint4 r[TR ? 4 : ROWS / 32];
  __device__ __forceinline__ void load(const bf16_t* base, long ld, int rows_valid, int k_valid) {
    if constexpr (TR) load_ks<ROWS>(base, ld, rows_valid, k_valid, r);
    else load_kc<ROWS>(base, ld, rows_valid, k_valid, r);
  }
  __device__ __forceinline__ void store(unsigned char* lds) {
    if constexpr (TR) store_ks<ROWS>(lds, r);
    else store_kc<ROWS>(lds, r);
  }
};

template <bool TA, bool TB, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmP p) {
  constexpr int FM = BM / WM / 32, FN = BN / WN / 32;
  static_assert(WM * WN == 4 && FM >= 1 && FN >= 1, "4 waves");
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (A_BYTES + B_BYTES)];

  const int tile = blockIdx.x;
  const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
  const int z = blockIdx.y, split = blockIdx.z;
  const int zo = z / p.batch_i, zi = z % p.batch_i;
  const int m0 = tm * BM, n0 = tn * BN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;

  const bf16_t* Ab = (const bf16_t*)p.A + (long)zo * p.sA_o + (long)zi * p.sA_i;
  const bf16_t* Bb = (const bf16_t*)p.B + (long)zo * p.sB_o + (long)zi * p.sB_i;
  const int kt_per = (p.K + GEMM_BK - 1) / GEMM_BK;
  int t0, t1;
  gemm_split_range(p.KB * kt_per, p.split_k, split, t0, t1);

  f32x16_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  Stage<TA, BM> sa;
  Stage<TB, BN> sb;
  auto issue = [&](int t) {
    const int kb = t / kt_per, k0 = (t % kt_per) * GEMM_BK;
    const bf16_t* a = Ab + (long)kb * p.sA_kb + (TA ? ((long)k0 * p.lda + m0) : ((long)m0 * p.lda + k0));
    const bf16_t* b = Bb + (long)kb * p.sB_kb + (TB ? ((long)k0 * p.ldb + n0) : ((long)n0 * p.ldb + k0));
    sa.load(a, p.lda, p.M - m0, p.K - k0);
    sb.load(b, p.ldb, p.N - n0, p.K - k0);
  };

  int cur = 0;
  if (t0 < t1) {
    issue(t0);
    sa.store(smem);
    sb.store(smem + A_BYTES);
  }
  __syncthreads();

  for (int t = t0; t < t1; ++t) {
    const bool more = (t + 1 < t1);
    if (more) issue(t + 1);
    const unsigned char* la = smem + cur * (A_BYTES + B_BYTES);
    const unsigned char* lb = la + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < GEMM_BK / 16; ++kk) {
      U4 fa[FM], fb[FN];
      const int chunk = 2 * kk + (lane >> 5);
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = wm * (BM / WM) + i * 32 + (lane & 31);
        fa[i].v = *reinterpret_cast<const uint4*>(la + lds_off(row, chunk));
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int row = wn * (BN / WN) + j * 32 + (lane & 31);
        fb[j].v = *reinterpret_cast<const uint4*>(lb + lds_off(row, chunk));
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i].b, fb[j].b, acc[i][j], 0, 0, 0);
    }
    if (more) {
      unsigned char* na = smem + (cur ^ 1) * (A_BYTES + B_BYTES);
      sa.store(na);
      sb.store(na + A_BYTES);
    }
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int nn = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mm = m0 + wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (mm < p.M && nn < p.N) gemm_store(p, z, split, mm, nn, acc[i][j][r]);
      }
    }
}

// sums the split-K slabs and applies the final epilogue
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmP p, int nbatch) {
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

template <bool TA, bool TB, int BM, int BN, int WM, int WN>
static int launch_cfg(GemmP& p, int nbatch, hipStream_t st) {
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)nbatch, (unsigned)p.split_k);
  WL_LAUNCH((gemm_bf16_kernel<TA, TB, BM, BN, WM, WN>), grid, dim3(256), 0, st, p);
  return wl_check_launch();
}

template <bool TA, bool TB>
static int launch_t(GemmP& p, int nbatch, hipStream_t st) {
  if (p.N <= 64) return launch_cfg<TA, TB, 128, 64, 2, 2>(p, nbatch, st);
  return launch_cfg<TA, TB, 128, 128, 2, 2>(p, nbatch, st);
}

int gemm_f32_launch(const wavlm_gemm_desc* d, hipStream_t st);  // gemm_f32.hip

static bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---- optional per-launch timing (bench.py roofline leg): HIP events recorded on the launch stream around every
// wavlm_gemm call while enabled.  Single-threaded use only (one stream, one host thread).
#define PROF_MAX 16384
static struct {
  int enabled, n;
  hipEvent_t ev0[PROF_MAX], ev1[PROF_MAX];
  double flops[PROF_MAX];
  int dtype[PROF_MAX];
  int created;
} g_prof;

static int prof_begin(const wavlm_gemm_desc* d, hipStream_t st) {
  if (!g_prof.enabled || g_prof.n >= PROF_MAX) return -1;
  const int i = g_prof.n;
  if (i >= g_prof.created) {
    if (hipEventCreate(&g_prof.ev0[i]) != hipSuccess || hipEventCreate(&g_prof.ev1[i]) != hipSuccess) return -1;
    g_prof.created = i + 1;
  }
  const double nb = (double)(d->batch_o < 1 ? 1 : d->batch_o) * (d->batch_i < 1 ? 1 : d->batch_i);
  g_prof.flops[i] = 2.0 * d->M * d->N * (double)d->K * (d->KB < 1 ? 1 : d->KB) * nb;
  g_prof.dtype[i] = d->dtype;
  hipEventRecord(g_prof.ev0[i], st);
  return i;
}
static void prof_end(int i, hipStream_t st) {
  if (i < 0) return;
  hipEventRecord(g_prof.ev1[i], st);
  g_prof.n = i + 1;
}

extern "C" void wavlm_prof_enable(int on) { g_prof.enabled = on; if (on) g_prof.n = 0; }
// Sums over the launches recorded since wavlm_prof_enable(1) with element type `dtype` (-1: all).
// Blocks until those launches have finished.  Returns the number of launches.
extern "C" int wavlm_prof_collect(int dtype, double* total_ms, double* total_flops) {
  double ms = 0.0, fl = 0.0;
  int cnt = 0;
  for (int i = 0; i < g_prof.n; ++i) {
    if (dtype >= 0 && g_prof.dtype[i] != dtype) continue;
    if (hipEventSynchronize(g_prof.ev1[i]) != hipSuccess) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_prof.ev0[i], g_prof.ev1[i]) != hipSuccess) continue;
    ms += t; fl += g_prof.flops[i]; ++cnt;
  }
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  return cnt;
}

extern "C" uint64_t wavlm_gemm_workspace_bytes(const wavlm_gemm_desc* d) {
  if (!d || d->split_k <= 1) return 0;
  const uint64_t nb = (uint64_t)(d->batch_o < 1 ? 1 : d->batch_o) * (d->batch_i < 1 ? 1 : d->batch_i);
  return nb * (uint64_t)d->split_k * (uint64_t)d->M * (uint64_t)d->N * sizeof(float);
}

extern "C" int wavlm_gemm(const wavlm_gemm_desc* d, void* stream) {
  if (!d || !d->A || !d->B || !d->C) return WL_EINVAL;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) return WL_EINVAL;
  if (d->epi == 2 && !d->aux) return WL_EINVAL;
  if (d->split_k > 1 && (!d->workspace || d->ws_bytes < wavlm_gemm_workspace_bytes(d))) return WL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (d->dtype == WL_F32) {
    const int pi = prof_begin(d, st);
    const int r = gemm_f32_launch(d, st);
    prof_end(pi, st);
    return r;
  }
  if (d->dtype != WL_BF16) return WL_EINVAL;
  // 16-byte vector loads: base pointers and every stride must be multiples of 8 elements
  if (!aligned16(d->A) || !aligned16(d->B)) return WL_EINVAL;
  const int64_t strides[] = {d->lda, d->ldb, d->sA_kb, d->sB_kb, d->sA_o, d->sA_i, d->sB_o, d->sB_i};
  for (int64_t s : strides) if (s % 8 != 0) return WL_EINVAL;
  GemmP p = make_gemm_params(d);
  const int nbatch = (d->batch_o < 1 ? 1 : d->batch_o) * p.batch_i;
  const int pi = prof_begin(d, st);
  int rc;
  if (!d->transA && !d->transB) rc = launch_t<false, false>(p, nbatch, st);
  else if (!d->transA && d->transB) rc = launch_t<false, true>(p, nbatch, st);
  else if (d->transA && !d->transB) rc = launch_t<true, false>(p, nbatch, st);
  else rc = launch_t<true, true>(p, nbatch, st);
  if (rc == WL_OK && p.split_k > 1) {
    const long total = (long)p.M * p.N * nbatch;
    long blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
    WL_LAUNCH(gemm_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, nbatch);
    rc = wl_check_launch();
  }
  prof_end(pi, st);
  return rc;
}
