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
#include <atomic>

#include "gemm_common.hpp"

#include "tile_loaders.hpp"

template <bool TA, bool TB, int BM, int BN, int WM, int WN, bool VEC>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_kernel(GemmP p) {
  constexpr int FM = BM / WM / 32, FN = BN / WN / 32;
  constexpr int NT = WM * WN * 64;
  static_assert((WM * WN == 4 || WM * WN == 8) && FM >= 1 && FN >= 1, "4 or 8 waves");
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 * (A_BYTES + B_BYTES)

  // XCD-aware tile order: the dispatcher places block b on XCD b % 8 (private L2 per XCD); remap so each XCD walks
  // a contiguous range of tiles (neighbours share the A row panel and all of B).  Pure speed, any placement is valid.
  int tile;
  {
    const int nt = p.tiles_m * p.tiles_n, bid = blockIdx.x;
    const int q = nt >> 3, rem = nt & 7, xcd = bid & 7, idx = bid >> 3;
    tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
  }
  const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
  const int z = blockIdx.y, split = blockIdx.z;
  const int zo = z / p.batch_i, zi = z % p.batch_i;
  const int m0 = tm * BM, n0 = tn * BN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
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

  Operand<TA, BM, NT> oa;
  Operand<TB, BN, NT> ob;
  oa.init(Ab + (TA ? (long)m0 : (long)m0 * p.lda), p.lda, p.M - m0);
  ob.init(Bb + (TB ? (long)n0 : (long)n0 * p.ldb), p.ldb, p.N - n0);
  auto issue = [&](int t, unsigned char* buf) {
    const int kb = t / kt_per, k0 = (t - kb * kt_per) * GEMM_BK;
    oa.issue((long)kb * p.sA_kb + (TA ? (long)k0 * p.lda : (long)k0), p.K - k0, buf, wave_u);
    ob.issue((long)kb * p.sB_kb + (TB ? (long)k0 * p.ldb : (long)k0), p.K - k0, buf + A_BYTES, wave_u);
    return p.K - k0;
  };

  int cur = 0;
  if (t0 < t1) {
    const int kv = issue(t0, smem);
    oa.commit(kv, smem);
    ob.commit(kv, smem + A_BYTES);
  }
  __syncthreads();

  for (int t = t0; t < t1; ++t) {
    const bool more = (t + 1 < t1);
    unsigned char* nbuf = smem + (cur ^ 1) * (A_BYTES + B_BYTES);
    int kv = GEMM_BK;
    if (more) kv = issue(t + 1, nbuf);
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
      oa.commit(kv, nbuf);
      ob.commit(kv, nbuf + A_BYTES);
    }
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue ------------------------------------------------------------------------------------------
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
  if constexpr (!VEC) {
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
  } else {
    // Row-vector epilogue: each wave stages one 32-row block of its accumulators through its private LDS
    // slice as fp32 (ds_write_b32: 32 lanes -> 32 consecutive dwords, conflict-free), then every lane owns
    // 8 consecutive columns of a row: bias / aux / residual are 16-byte loads and C is written as full
    // 16-byte (bf16) or 2 x 16-byte (f32) vectors -- 8 rows x 128 B per wave-instruction instead of 2-byte
    // scatters.  The main loop's last barrier has already retired every LDS read of the operand tiles.
    constexpr int WCOLS = BN / WN;          // columns of the wave tile (FN * 32)
    constexpr int EP_LD = WCOLS + 4;        // padded fp32 row: keeps 16-B alignment, rotates banks by 4
    constexpr int CH = WCOLS / 8;           // 8-column chunks per row
    float* ep = reinterpret_cast<float*>(smem) + wave * (32 * EP_LD);
    const int zo2 = zo, zi2 = zi;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          ep[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * EP_LD + j * 32 + (lane & 31)] = acc[i][j][r];
      __syncthreads();
#pragma unroll
      for (int q = 0; q < (32 * CH) / 64; ++q) {
        const int id = lane + 64 * q;
        const int rl = id / CH, ch = id % CH;
        const int mm = m0 + wm * (BM / WM) + i * 32 + rl;
        const int nn = n0 + wn * WCOLS + ch * 8;
        if (mm < p.M && nn < p.N) {
          const float4 lo = *reinterpret_cast<const float4*>(ep + rl * EP_LD + ch * 8);
          const float4 hi = *reinterpret_cast<const float4*>(ep + rl * EP_LD + ch * 8 + 4);
          float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
          gemm_store8(p, zo2, zi2, z, split, mm, nn, v);
        }
      }
      __syncthreads();
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

// row-vector form (N % 8 == 0, 16-byte aligned rows: vec_epilogue_ok): 8 outputs per thread, two float4 loads per
// slab with the loads of up to four slabs in flight, 16-byte epilogue accesses -- the scalar form above spends 22 us
// on 66 MB of slabs (one float and a 64-bit div/mod per thread and iteration)
__global__ __launch_bounds__(256) void gemm_splitk_reduce8_kernel(GemmP p, int nbatch) {
  const long mn = (long)p.M * p.N;
  const long total8 = (mn * nbatch) >> 3;
  const int S = p.split_k;
  GemmP q = p;
  q.split_k = 1;  // gemm_store8 then takes the final-epilogue path
  for (long i8 = (long)blockIdx.x * blockDim.x + threadIdx.x; i8 < total8; i8 += (long)gridDim.x * blockDim.x) {
    const long i = i8 << 3;
    const int z = (int)(i / mn);
    const long r = i - (long)z * mn;
    const float* src = p.ws + (long)z * S * mn + r;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    int k = 0;
    for (; k + 4 <= S; k += 4) {
      float4 a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = *reinterpret_cast<const float4*>(src + (long)(k + u) * mn);
        b[u] = *reinterpret_cast<const float4*>(src + (long)(k + u) * mn + 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v[0] += a[u].x; v[1] += a[u].y; v[2] += a[u].z; v[3] += a[u].w;
        v[4] += b[u].x; v[5] += b[u].y; v[6] += b[u].z; v[7] += b[u].w;
      }
    }
    for (; k < S; ++k) {
      const float4 a = *reinterpret_cast<const float4*>(src + (long)k * mn);
      const float4 b = *reinterpret_cast<const float4*>(src + (long)k * mn + 4);
      v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
    const int m = (int)(r / p.N), n = (int)(r - (long)m * p.N);
    gemm_store8(q, z / p.batch_i, z % p.batch_i, z, 0, m, n, v);
  }
}

// the slab reductions of a grouped weight-gradient launch as ONE launch: problem g owns the 8-element vectors
// [base[g], base[g + 1]); plain epilogue only (alpha, optional accumulate into C), which is what wavlm_gemm_grouped accepts
struct RedGrpP {
  float* ws[4]; void* C[4]; long ldc[4]; int M[4], N[4], c_dtype[4], accumulate[4]; float alpha[4];
  long base[5]; int n, S;
};
__global__ __launch_bounds__(256) void gemm_splitk_reduce8_grouped_kernel(RedGrpP p) {
  const long total8 = p.base[p.n];
  for (long i8 = (long)blockIdx.x * blockDim.x + threadIdx.x; i8 < total8; i8 += (long)gridDim.x * blockDim.x) {
    const int g = (i8 >= p.base[1]) + (i8 >= p.base[2]) + (i8 >= p.base[3]);
    const long i = (i8 - p.base[g]) << 3;
    const int N = p.N[g];
    const long mn = (long)p.M[g] * N;
    const float* src = p.ws[g] + i;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    for (int k = 0; k < p.S; ++k) {
      const float4 a = *reinterpret_cast<const float4*>(src + (long)k * mn);
      const float4 b = *reinterpret_cast<const float4*>(src + (long)k * mn + 4);
      v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
    const long m = i / N, n = i - m * N;
    const long off = m * p.ldc[g] + n;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= p.alpha[g];
    if (p.accumulate[g]) {
      float c[8];
      ld8_dt(p.C[g], off, p.c_dtype[g], c);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += c[e];
    }
    st8_dt(p.C[g], off, p.c_dtype[g], v);
  }
}

template <bool TA, bool TB, int BM, int BN, int WM, int WN>
static int launch_cfg(GemmP& p, int nbatch, bool vec, hipStream_t st) {
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)nbatch, (unsigned)p.split_k);
  constexpr size_t smem = 2 * (size_t)(BM + BN) * 128;
  constexpr int nt = WM * WN * 64;
  if (smem > 65536) {  // opt in to > 64 KiB of dynamic LDS once per instantiation
    static bool done_v = false, done_s = false;
    bool& done = vec ? done_v : done_s;
    if (!done) {
      const void* fn = vec ? (const void*)gemm_bf16_kernel<TA, TB, BM, BN, WM, WN, true>
                           : (const void*)gemm_bf16_kernel<TA, TB, BM, BN, WM, WN, false>;
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return WL_ELAUNCH;
      done = true;
    }
  }
  if (vec) WL_LAUNCH((gemm_bf16_kernel<TA, TB, BM, BN, WM, WN, true>), grid, dim3(nt), smem, st, p);
  else WL_LAUNCH((gemm_bf16_kernel<TA, TB, BM, BN, WM, WN, false>), grid, dim3(nt), smem, st, p);
  return wl_check_launch();
}

static int g_gemm_variant = 0;  // 0: auto, 1: force 128x128, 2: force 256x128, 3 / 4: force the 256x256 / 192x384 ping-pong kernel where it applies
extern int g_pp_mode;  // gemm_pp.hip
// low 4 bits: tile variant; bits 4-5 (+1): launch mode of the 256x256 ping-pong kernel (tools A/B); 0 keeps the default
extern "C" void wavlm_gemm_set_variant(int v) { g_gemm_variant = v & 15; if (v >> 4) g_pp_mode = (v >> 4) - 1; }

template <bool TA, bool TB>
static int launch_t(GemmP& p, int nbatch, bool vec, hipStream_t st) {
  if (p.N <= 64) return launch_cfg<TA, TB, 128, 64, 2, 2>(p, nbatch, vec, st);
  // large problems: 256 x 128 tile, 8 waves (4 x 2), 96 KiB of LDS, one block per CU -- 2/3 of the operand bytes per
  // flop of the 128 x 128 tile and twice the MFMA work between barriers
  if (g_gemm_variant == 2 ||
      (g_gemm_variant == 0 && p.M >= 2048 && (long)((p.M + 255) / 256) * ((p.N + 127) / 128) * nbatch * p.split_k >= 192))
    return launch_cfg<TA, TB, 256, 128, 4, 2>(p, nbatch, vec, st);
  return launch_cfg<TA, TB, 128, 128, 2, 2>(p, nbatch, vec, st);
}

// the row-vector epilogue needs every row start of C / aux / res / bias 16-byte aligned for 8-element vectors
// final_stage: the checks of the final epilogue even for a split-K launch (the slab reduction applies it)
static bool vec_epilogue_ok(const wavlm_gemm_desc* d, bool final_stage = false) {
  auto ok = [](const void* ptr, int64_t a, int64_t b, int64_t c) {
    return (((uintptr_t)ptr) & 15) == 0 && a % 8 == 0 && b % 8 == 0 && c % 8 == 0;
  };
  if (d->split_k > 1 && !final_stage) return (((uintptr_t)d->workspace) & 15) == 0 && d->N % 8 == 0;
  if (final_stage && d->N % 8) return false;  // the slab reduction walks 8 outputs at a time; a plain launch handles a ragged last chunk itself
  if (!ok(d->C, d->ldc, d->sC_o, d->sC_i)) return false;
  if (d->bias && !ok(d->bias, 0, d->sBias_o, d->sBias_i)) return false;
  if (d->aux && !ok(d->aux, d->ld_aux, d->sAux_o, d->sAux_i)) return false;
  if (d->res && !ok(d->res, d->ld_res, d->sRes_o, d->sRes_i)) return false;
  return true;
}

int gemm_f32_launch(const wavlm_gemm_desc* d, hipStream_t st);  // gemm_f32.hip
bool gemm_pp_ok(const wavlm_gemm_desc* d);                       // gemm_pp.hip
int gemm_pp_launch(GemmP& p, int nbatch, bool transA, bool transB, int ep, hipStream_t st);
bool gemm_pp3_ok(const wavlm_gemm_desc* d);                      // gemm_pp3.hip
int gemm_pp3_launch(GemmP& p, int nbatch, bool transA, bool transB, int ep, hipStream_t st);
// Lab paths (measured slower / neutral, kept as the record of the experiments): compiled only into the lab library
// (-DWAVLM_EXPERIMENTAL, tools/probe/build_probe.py lab); libwavlm_hip.so carries neither.
//   * gemm_h2 (tools/probe/gemm_h2.hip): 192 x 192 x 32 tiles, two workgroups per CU; variant 6 / WAVLM_GEMM_H2=1
//   * the balanced grouped weight-gradient launch (gemm_common.hpp: gemm_sk_plan); WAVLM_WGRAD_STREAMK=1
#if defined(WAVLM_EXPERIMENTAL)
#define GEMM_LAB 1
bool gemm_h2_ok(const wavlm_gemm_desc* d, int ec);                // gemm_h2.hip: 192 x 192 x 32, two workgroups per CU (short K)
int gemm_h2_launch(GemmP& p, bool transB, int ep, hipStream_t st);
// WAVLM_GEMM_H2: 0 never | 1 wherever the shape fits (gemm_h2_ok) | unset: by measurement (h2_takes)
static int h2_mode() { static const int m = getenv("WAVLM_GEMM_H2") ? atoi(getenv("WAVLM_GEMM_H2")) : -1; return m; }
static bool h2_takes(const wavlm_gemm_desc* d, int ec) {
  const int m = h2_mode();
  if (m == 0 || !gemm_h2_ok(d, ec)) return false;
  if (m == 1) return true;
  return false;
}
#else
#define GEMM_LAB 0
#endif
int gemm_w4_launch(GemmP& p, int nbatch, bool transA, bool transB, int ep, hipStream_t st);  // gemm_w4.hip (same shapes as gemm_pp)
int gemm_w4_launch_grouped(GemmP& p, hipStream_t st);
// which 256 x 256 kernel takes a launch: the eight-wave ping-pong (gemm_pp.hip) or the four-wave one (gemm_w4.hip).
// WAVLM_GEMM_W4: unset = the grouped weight-gradient launch only | 0 never | 1 every launch the 256 x 256 kernel takes |
// 2 every launch with BOTH operands K-strided.  Measured (profiles/r04/gemm_w4_ab.txt): the grouped launch of a block's four
// weight gradients (one K batch, 188 steady K steps per work item) runs 306 us against 335 us; the conv stack's weight
// gradients (32 K batches with a K tail each: no steady steps) 343 against 307 us; K-contiguous operands lose 3-8 %.
static int w4_mode() { static const int m = getenv("WAVLM_GEMM_W4") ? atoi(getenv("WAVLM_GEMM_W4")) : 3; return m; }
static bool w4_takes(const wavlm_gemm_desc* d, bool grouped) {
  const int m = w4_mode();
  if (m == 1) return true;
  if (m == 2) return grouped || (d->transA && d->transB);
  if (m == 3) return grouped;
  return false;
}

// fraction of a round's MFMA work that is useful for a BM x BN tiling on the persistent grid (256 CUs minus those left to the
// RCCL kernels): edge waste x round quantisation
static double tile_efficiency(const wavlm_gemm_desc* d, int nbatch, int BM, int BN) {
  const long tm = (d->M + BM - 1) / BM, tn = (d->N + BN - 1) / BN;
  const long tiles = tm * tn * nbatch * (d->split_k < 1 ? 1 : d->split_k);
  const double useful = ((double)d->M * d->N) / ((double)tm * BM * tn * BN);
  const long G = 256 - wavlm_get_reserved_cus();
  return useful * (double)tiles / ((double)G * (double)((tiles + G - 1) / G));
}

static bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// chord table of the normal CDF for the fast GELU epilogue (gemm_common.hpp gelu_both_tab); filled once per process and device
__device__ float4 g_gelu_tab4[GT4_N];
__global__ void gelu_tab4_init_kernel() {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= GT2_N) return;
  const double h = 16.0 / GT2_N, x0 = -8.0 + i * h, x1 = x0 + h;
  auto cdf = [](double x) { return 0.5 * erfc(-x * 0.70710678118654752440); };   // (erfc: no cancellation in the left tail)
  const double a = (cdf(x1) - cdf(x0)) / h;
  reinterpret_cast<float2*>(g_gelu_tab4)[i] = make_float2((float)a, (float)(cdf(x0) - a * x0));
}
// Per device (hipGetSymbolAddress resolves against the CURRENT device's copy of the module), under a mutex, and the host
// waits for the init kernel once: the table is then visible to every stream and thread that asks for it afterwards (a
// pointer cached before the fill had run could be read unfilled by a first GELU-epilogue GEMM on another stream).
static const float4* gelu_tab4_get(hipStream_t st) {
  static std::mutex mu;
  static std::atomic<const float4*> ptr[WL_MAX_DEVICES];   // zero-initialised; published with release once filled
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= WL_MAX_DEVICES) return nullptr;
  const float4* p = ptr[dev].load(std::memory_order_acquire);
  if (p) return p;   // every launch after the first per device: one atomic load, no lock, no runtime call besides hipGetDevice
  // first GELU-epilogue launch on this device: fill the table and WAIT for it (the one place the library synchronises: a
  // pointer published before the fill had run could be read unfilled by a GEMM on another stream)
  std::lock_guard<std::mutex> lk(mu);
  p = ptr[dev].load(std::memory_order_acquire);
  if (!p) {
    void* a = nullptr;
    if (hipGetSymbolAddress(&a, HIP_SYMBOL(g_gelu_tab4)) != hipSuccess) return nullptr;
    hipLaunchKernelGGL(gelu_tab4_init_kernel, dim3(GT2_N / 256), dim3(256), 0, st);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return nullptr;
    p = (const float4*)a;
    ptr[dev].store(p, std::memory_order_release);
  }
  return p;
}

// the same table for the other translation units of the library (posconv_direct.hip)
const float4* wl_gelu_tab4(hipStream_t st) { return gelu_tab4_get(st); }

// ---- optional per-launch timing (bench.py roofline leg): HIP events recorded on the launch stream around every
// wavlm_gemm call while enabled.  Single-threaded use only (one stream, one host thread).
#define PROF_MAX 16384
static struct {
  int enabled, n;
  hipEvent_t ev0[PROF_MAX], ev1[PROF_MAX];
  double flops[PROF_MAX];
  double bytes[PROF_MAX];  // algorithmic HBM bytes: each operand / output / epilogue tensor touched once
  int dtype[PROF_MAX];
  int cls[PROF_MAX];       // WL_PROF_* kernel class (0 = wavlm_gemm)
  int shape[PROF_MAX][8];  // wavlm_gemm launches: M, N, K, KB, transA | transB << 1, epi, split_k, batches
  int created;
} g_prof;

// shared with the other translation units (common.hpp): a launcher brackets its kernels with wl_prof_begin / wl_prof_end
int wl_prof_begin(int cls, int dtype, double flops, double bytes, hipStream_t st) {
  if (!g_prof.enabled || g_prof.n >= PROF_MAX) return -1;
  const int i = g_prof.n;
  if (i >= g_prof.created) {
    if (hipEventCreate(&g_prof.ev0[i]) != hipSuccess || hipEventCreate(&g_prof.ev1[i]) != hipSuccess) return -1;
    g_prof.created = i + 1;
  }
  g_prof.flops[i] = flops; g_prof.bytes[i] = bytes; g_prof.dtype[i] = dtype; g_prof.cls[i] = cls;
  hipEventRecord(g_prof.ev0[i], st);
  g_prof.n = i + 1;            // claimed now: a nested launcher (a GEMM inside a profiled composite) takes the next slot
  return i;
}
void wl_prof_end(int i, hipStream_t st) {
  if (i < 0) return;
  hipEventRecord(g_prof.ev1[i], st);
}

static int prof_begin(const wavlm_gemm_desc* d, hipStream_t st) {
  if (!g_prof.enabled) return -1;
  const double nb = (double)(d->batch_o < 1 ? 1 : d->batch_o) * (d->batch_i < 1 ? 1 : d->batch_i);
  const double flops = 2.0 * d->M * d->N * (double)d->K * (d->KB < 1 ? 1 : d->KB) * nb;
  const double es = d->dtype == WL_BF16 ? 2.0 : 4.0, cs = d->c_dtype == WL_BF16 ? 2.0 : 4.0;
  const double kb = d->KB < 1 ? 1 : d->KB, mn = (double)d->M * d->N;
  // overlapping-row operands (lda < K: strided conv) are counted once per stored element, not once per use
  const double a_el = (!d->transA && d->lda < d->K) ? (double)d->M * d->lda : (double)d->M * d->K * kb;
  const double b_el = (double)d->N * d->K * kb;
  const double bytes = nb * (a_el * es + mn * cs * (d->accumulate ? 2.0 : 1.0) + (d->aux ? mn * 2.0 : 0.0) + (d->res ? mn * 2.0 : 0.0)) +
                       ((d->sB_o || d->sB_i) ? nb : 1.0) * b_el * es;
  const int pi = wl_prof_begin(WL_PROF_GEMM, d->dtype, flops, bytes, st);
  if (pi >= 0) {
    int* sh = g_prof.shape[pi];
    sh[0] = d->M; sh[1] = d->N; sh[2] = d->K; sh[3] = d->KB < 1 ? 1 : d->KB; sh[4] = (d->transA ? 1 : 0) | (d->transB ? 2 : 0);
    sh[5] = d->epi | (d->aux ? 8 : 0) | (d->res ? 16 : 0) | (d->colsum ? 32 : 0); sh[6] = d->split_k < 1 ? 1 : d->split_k; sh[7] = (int)nb;
  }
  return pi;
}
static void prof_end(int i, hipStream_t st) { wl_prof_end(i, st); }

extern "C" void wavlm_prof_enable(int on) { g_prof.enabled = on; if (on) g_prof.n = 0; }
// Sums over the wavlm_gemm launches recorded since wavlm_prof_enable(1) with element type `dtype` (-1: all).
// Blocks until those launches have finished.  Returns the number of launches.
extern "C" int wavlm_prof_collect(int dtype, double* total_ms, double* total_flops) {
  double ms = 0.0, fl = 0.0;
  int cnt = 0;
  for (int i = 0; i < g_prof.n; ++i) {
    if (g_prof.cls[i] != WL_PROF_GEMM || (dtype >= 0 && g_prof.dtype[i] != dtype)) continue;
    if (hipEventSynchronize(g_prof.ev1[i]) != hipSuccess) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_prof.ev0[i], g_prof.ev1[i]) != hipSuccess) continue;
    ms += t; fl += g_prof.flops[i]; ++cnt;
  }
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  return cnt;
}
// one line per recorded wavlm_gemm launch: "M N K KB trans epi split batches ms gflop" (trans: bit 0 A, bit 1 B K-strided; epi:
// low 3 bits the epilogue, +8 aux, +16 residual, +32 fused column sums).  Blocks until the launches have finished.
extern "C" int wavlm_prof_dump(const char* path) {
  FILE* f = fopen(path, "w");
  if (!f) return WL_EINVAL;
  int cnt = 0;
  for (int i = 0; i < g_prof.n; ++i) {
    if (g_prof.cls[i] != WL_PROF_GEMM) continue;
    if (hipEventSynchronize(g_prof.ev1[i]) != hipSuccess) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_prof.ev0[i], g_prof.ev1[i]) != hipSuccess) continue;
    const int* sh = g_prof.shape[i];
    fprintf(f, "%d %d %d %d %d %d %d %d %.4f %.3f\n", sh[0], sh[1], sh[2], sh[3], sh[4], sh[5], sh[6], sh[7], t, g_prof.flops[i] * 1e-9);
    ++cnt;
  }
  fclose(f);
  return cnt;
}
// algorithmic HBM bytes of the same launches (no synchronisation)
extern "C" double wavlm_prof_collect_bytes(int dtype) {
  double by = 0.0;
  for (int i = 0; i < g_prof.n; ++i)
    if (g_prof.cls[i] == WL_PROF_GEMM && (dtype < 0 || g_prof.dtype[i] == dtype)) by += g_prof.bytes[i];
  return by;
}
// the same per kernel class (WL_PROF_*): number of recorded calls; total duration, algorithmic flops and bytes
extern "C" int wavlm_prof_collect_class(int cls, double* total_ms, double* total_flops, double* total_bytes) {
  double ms = 0.0, fl = 0.0, by = 0.0;
  int cnt = 0;
  for (int i = 0; i < g_prof.n; ++i) {
    if (g_prof.cls[i] != cls) continue;
    if (hipEventSynchronize(g_prof.ev1[i]) != hipSuccess) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_prof.ev0[i], g_prof.ev1[i]) != hipSuccess) continue;
    ms += t; fl += g_prof.flops[i]; by += g_prof.bytes[i]; ++cnt;
  }
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  if (total_bytes) *total_bytes = by;
  return cnt;
}

extern "C" uint64_t wavlm_gemm_workspace_bytes(const wavlm_gemm_desc* d) {
  if (!d) return 0;
  uint64_t need = 0;
  if (d->split_k > 1) {
    const uint64_t nb = (uint64_t)(d->batch_o < 1 ? 1 : d->batch_o) * (d->batch_i < 1 ? 1 : d->batch_i);
    need = nb * (uint64_t)d->split_k * (uint64_t)d->M * (uint64_t)d->N * sizeof(float);
  }
  if (d->colsum) {  // per-tile partial rows of the fused form (at most ceil(M / 192) of them) or the stand-alone pass
    const uint64_t fused = (uint64_t)((d->M + 191) / 192) * (uint64_t)d->N * sizeof(float);
    const uint64_t pass = wavlm_colsum_workspace_bytes(d->N);
    need += fused > pass ? fused : pass;
  }
  return need;
}

// gradient listener (wavlm_dp_set_listener): an accumulating GEMM is a weight gradient landing in the caller's arena
static void gemm_notify(const wavlm_gemm_desc* d, bool colsum_fused, void* stream) {
  if (d->accumulate && d->batch_o <= 1 && d->batch_i <= 1)
    wl_notify_grad(d->C, ((uint64_t)(d->M - 1) * d->ldc + d->N) * wl_esize(d->c_dtype), stream);
  if (d->colsum && d->colsum_accumulate && colsum_fused && !wl_fin_active())   // (the stand-alone pass reports through wavlm_colsum; a deferred finish through its flush)
    wl_notify_grad(d->colsum, (uint64_t)d->N * wl_esize(d->colsum_dtype), stream);
}

extern "C" int wavlm_gemm(const wavlm_gemm_desc* d, void* stream) {
  if (!d || !d->A || !d->B || !d->C) return WL_EINVAL;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) return WL_EINVAL;
  if ((d->epi == 2 || d->epi == 4) && !d->aux) return WL_EINVAL;
  if (d->epi < 0 || d->epi > 4) return WL_EINVAL;
  if ((d->split_k > 1 || d->colsum) && (!d->workspace || d->ws_bytes < wavlm_gemm_workspace_bytes(d))) return WL_EINVAL;
  if (d->colsum && (d->split_k > 1 || d->batch_o > 1 || d->batch_i > 1 || d->sC_o || d->sC_i)) return WL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (d->dtype == WL_F32) {
    const int pi = prof_begin(d, st);
    int r = gemm_f32_launch(d, st);
    prof_end(pi, st);
    if (r == WL_OK && d->colsum)
      r = wavlm_colsum(d->C, d->M, d->N, d->ldc, d->c_dtype, nullptr, nullptr, d->colsum, d->colsum_dtype,
                       d->colsum_accumulate, d->workspace, d->ws_bytes, stream);
    if (r == WL_OK) gemm_notify(d, false, stream);
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
  int csum_rows = 0;  // > 0: the kernel left that many partial rows of column sums in the workspace
  const bool vec = vec_epilogue_ok(d);
#if GEMM_LAB
  if (g_gemm_variant == 6 ? gemm_h2_ok(d, gemm_epilogue_class(d, vec)) : (g_gemm_variant == 0 && h2_takes(d, gemm_epilogue_class(d, vec)))) {
    const int ec = gemm_epilogue_class(d, vec);
    if (ec == 3) p.gtab = gelu_tab4_get(st);
    if (d->colsum && (ec == 2 || ec == 4)) { p.colsum_part = (float*)d->workspace; csum_rows = (d->M + 191) / 192; }
    rc = gemm_h2_launch(p, d->transB != 0, ec, st);
  }
  else
#endif
  if ((g_gemm_variant == 4 && gemm_pp3_ok(d)) ||
      (g_gemm_variant == 0 && d->N >= 384 && gemm_pp3_ok(d) && gemm_pp_ok(d) &&
       tile_efficiency(d, nbatch, 192, 384) > 1.06 * tile_efficiency(d, nbatch, 256, 256)))  // measured at 24 k rows: N = 768 (+25 %), 2304 (+10 % at K = 768), 3072 (+6 %) go to 192 x 384; N = 2048 and the conv stack (N = 512) stay
    { const int ec = gemm_epilogue_class(d, vec); if (ec == 3) p.gtab = gelu_tab4_get(st);
      if (d->colsum && (ec == 2 || ec == 4)) { p.colsum_part = (float*)d->workspace; csum_rows = (d->M + 191) / 192; }
      rc = gemm_pp3_launch(p, nbatch, d->transA != 0, d->transB != 0, ec, st); }
  else if ((g_gemm_variant == 3 || g_gemm_variant == 5 || (g_gemm_variant == 0 && d->N >= 256)) && gemm_pp_ok(d)) {
    const int ec = gemm_epilogue_class(d, vec); if (ec == 3) p.gtab = gelu_tab4_get(st);
    if (d->colsum && (ec == 2 || ec == 4)) { p.colsum_part = (float*)d->workspace; csum_rows = (d->M + 255) / 256; }
    rc = (g_gemm_variant == 5 || (g_gemm_variant == 0 && w4_takes(d, false)))
             ? gemm_w4_launch(p, nbatch, d->transA != 0, d->transB != 0, ec, st)
             : gemm_pp_launch(p, nbatch, d->transA != 0, d->transB != 0, ec, st);
  }
  else if (!d->transA && !d->transB) rc = launch_t<false, false>(p, nbatch, vec, st);
  else if (!d->transA && d->transB) rc = launch_t<false, true>(p, nbatch, vec, st);
  else if (d->transA && !d->transB) rc = launch_t<true, false>(p, nbatch, vec, st);
  else rc = launch_t<true, true>(p, nbatch, vec, st);
  if (rc == WL_OK && p.split_k > 1) {
    const long total = (long)p.M * p.N * nbatch;
    if (vec && vec_epilogue_ok(d, true)) {
      long blocks = ((total >> 3) + 255) / 256; if (blocks > 2048) blocks = 2048;
      WL_LAUNCH(gemm_splitk_reduce8_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, nbatch);
    } else {
      long blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
      WL_LAUNCH(gemm_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, nbatch);
    }
    rc = wl_check_launch();
  }
  if (rc == WL_OK && d->colsum) {
    if (csum_rows > 0) rc = wl_colsum_finish((const float*)d->workspace, csum_rows, d->N, d->colsum, d->colsum_dtype, d->colsum_accumulate, st);
    else rc = wavlm_colsum(d->C, d->M, d->N, d->ldc, d->c_dtype, nullptr, nullptr, d->colsum, d->colsum_dtype,
                           d->colsum_accumulate, d->workspace, d->ws_bytes, stream);
  }
  prof_end(pi, st);
  if (rc == WL_OK) gemm_notify(d, csum_rows > 0, stream);
  return rc;
}

int gemm_pp_launch_grouped(GemmP& p, hipStream_t st);  // gemm_pp.hip
bool gemm_w4_fixup_ok(const GemmP& p);                   // gemm_w4.hip

// Several weight-gradient-shaped problems (both operands K-strided, same K and split_k >= 2, plain epilogue) as ONE
// split-K launch of the 256 x 256 kernel plus one slab reduction per problem.  Anything else runs as n wavlm_gemm calls.
extern "C" int wavlm_gemm_grouped(const wavlm_gemm_desc* d, int32_t n, void* stream) {
  if (!d || n <= 0) return WL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  bool ok = n >= 2 && n <= 4;
  for (int i = 0; ok && i < n; ++i) {
    const wavlm_gemm_desc* e = d + i;
    ok = e->A && e->B && e->C && e->workspace && e->dtype == WL_BF16 && e->transA && e->transB && e->KB <= 1 &&
         e->batch_o <= 1 && e->batch_i <= 1 && e->K == d->K && e->split_k == d->split_k && e->split_k >= 2 &&
         e->epi == 0 && !e->bias && !e->aux && !e->res && e->M > 0 && e->N > 0 && gemm_pp_ok(e) &&
         aligned16(e->A) && aligned16(e->B) && e->lda % 8 == 0 && e->ldb % 8 == 0 &&
         vec_epilogue_ok(e) && vec_epilogue_ok(e, true) && e->ws_bytes >= wavlm_gemm_workspace_bytes(e);
  }
  if (!ok || g_gemm_variant != 0) {
    for (int i = 0; i < n; ++i) {
      const int rc = wavlm_gemm(d + i, stream);
      if (rc != WL_OK) return rc;
    }
    return WL_OK;
  }
  GemmP p = make_gemm_params(d);
  p.ngrp = n;
  // `split_k` = slabs the workspaces hold.  The one-round split (the largest with tiles x split <= grid) is used when it fills
  // the grid exactly; otherwise, with one more slab, the balanced launch (gemm_common.hpp: gemm_sk_plan): the CUs the one-round
  // split leaves idle take the K tail of every tile (Base: 108 tiles x 2 = 216 of 256 CUs busy -> 256).  Opt-in (WAVLM_WGRAD_STREAMK=1):
  // measured neutral at the launch and the step level (profiles/r04/ab_wgrad_balanced_*.txt).
  long tiles_all = 0;
  for (int i = 0; i < n; ++i) tiles_all += (long)((d[i].M + 255) / 256) * ((d[i].N + 255) / 256);
  const int G = 256 - wavlm_get_reserved_cus();
#if GEMM_LAB
  static const bool sk_on = [] { const char* e = getenv("WAVLM_WGRAD_STREAMK"); return e && *e == '1'; }();
#else
  constexpr bool sk_on = false;   // (lab library only)
#endif
  const int ksteps = (d->K + 63) / 64;
  GemmSk plan;
  const bool sk = sk_on && w4_takes(d, true) && tiles_all * (long)ksteps >= 8l * G && gemm_sk_plan(tiles_all, ksteps, G, plan) &&
                  d->split_k >= plan.s + 1;
  if (sk) p.split_k = plan.s + 1;
  else if (sk_on && !getenv("WAVLM_WGRAD_SPLIT") && G / tiles_all >= 2 && d->split_k == G / tiles_all + 1)
    p.split_k = (int)(G / tiles_all);   // the spare slab stays unused: the one-round split fills the grid or nothing is to be won
  int vb = 0;
  double flops = 0.0;
  for (int i = 0; i < 4; ++i) {
    const wavlm_gemm_desc* e = d + (i < n ? i : n - 1);
    GemmGrp& g = p.grp[i];
    g.A = e->A; g.B = e->B; g.ws = (float*)e->workspace; g.lda = e->lda; g.ldb = e->ldb; g.M = e->M; g.N = e->N;
    g.tiles_m = (e->M + 255) / 256; g.tiles_n = (e->N + 255) / 256;
    g.C = e->C; g.ldc = e->ldc; g.c_dtype = e->c_dtype; g.accumulate = e->accumulate; g.alpha = e->alpha;
    g.vbase = i < n ? vb : 0x7fffffff;   // first work item of the member (balanced launch: its first tile)
    if (i < n) { vb += g.tiles_m * g.tiles_n * (sk ? 1 : p.split_k); flops += 2.0 * e->M * e->N * (double)e->K; }
  }
  p.vtotal = vb;
  if (sk) {
    p.sk_ks = ksteps; p.sk_wgs = G; p.sk_tiles = (int)tiles_all; p.sk_s = plan.s; p.sk_lm = plan.Lm;
    p.vtotal = G * plan.q;
    static const bool spread = [] { const char* e = getenv("WAVLM_SK_SPREAD"); return !(e && *e == '0'); }();
    p.sk_spread = spread && (plan.s * tiles_all) % 8 == 0 && plan.R % 8 == 0 && G % 8 == 0;
  }
  const int pi = prof_begin(d, st);
  if (pi >= 0) {
    g_prof.flops[pi] = flops;
    double by = 0.0;
    for (int i = 0; i < n; ++i) by += 2.0 * ((double)d[i].M * d[i].K + (double)d[i].N * d[i].K) + 2.0 * 2.0 * d[i].M * d[i].N;
    g_prof.bytes[pi] = by;
    int* sh = g_prof.shape[pi];
    sh[0] = -n; sh[1] = 0; sh[2] = d->K; sh[3] = 1; sh[4] = 3; sh[5] = 0; sh[6] = sk ? 0 : p.split_k; sh[7] = n;   // M = -members: a grouped launch; split 0 = balanced
  }
  // in-kernel fix-up (gemm_w4.hip; WAVLM_WGRAD_FIXUP=1, measured slower and off): the last workgroup of a tile to arrive adds the
  // other splits' slabs and writes the final result -- no reduction launch, one slab round trip less per tile
  const bool fixup = w4_takes(d, true) && gemm_w4_fixup_ok(p);
  p.fix_epoch = fixup ? 1u : 0u; p.fix_cnt = nullptr; p.fix_flag = nullptr;
  int rc = w4_takes(d, true) ? gemm_w4_launch_grouped(p, st) : gemm_pp_launch_grouped(p, st);
  if (rc == WL_OK && !fixup) {
    RedGrpP r;
    r.n = n; r.S = p.split_k; r.base[0] = 0;
    for (int i = 0; i < 4; ++i) {
      const wavlm_gemm_desc* e = d + (i < n ? i : n - 1);
      r.ws[i] = (float*)e->workspace; r.C[i] = e->C; r.ldc[i] = e->ldc; r.M[i] = e->M; r.N[i] = e->N;
      r.c_dtype[i] = e->c_dtype; r.accumulate[i] = e->accumulate; r.alpha[i] = e->alpha;
      r.base[i + 1] = i < n ? r.base[i] + (((long)e->M * e->N) >> 3) : 0x7fffffffffffffffL;
    }
    const long total8 = r.base[n];  // (base[n + 1 ..] are "never reached": the kernel's problem search stops at n)
    long blocks = (total8 + 255) / 256; if (blocks > 4096) blocks = 4096;
    WL_LAUNCH(gemm_splitk_reduce8_grouped_kernel, dim3((unsigned)blocks), dim3(256), 0, st, r);
    rc = wl_check_launch();
  }
  prof_end(pi, st);
  if (rc == WL_OK)
    for (int i = 0; i < n; ++i) gemm_notify(d + i, false, stream);
  return rc;
}
