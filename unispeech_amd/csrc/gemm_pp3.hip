// bf16 MFMA GEMM for gfx950, second large-problem tile: 192 x 384 x 64, 8 waves (2 x 4), 96 x 96 accumulators per
// wave (3 x 3 v_mfma_f32_32x32x16_bf16 blocks), same ping-pong / LDS-DMA / counted-vmcnt machinery as gemm_pp.hip.
//
// Why a second shape: the transformer's token count (B = 32 x 749 = 23 968 rows) against N = 768 gives 94 x 3 = 282
// tiles of 256 x 256 -- two rounds on 256 CUs with the second round 10 % full (55 % efficiency).  192 x 384 tiles
// have the same area per flop-byte ratio (0.0078 B/flop) and give 125 x 2 = 250 tiles: one round, 98 % full.  For
// N = 2304 / 3072 they give 750 / 1000 tiles (2.93 / 3.9 rounds) against 846 / 1128 (3.3 / 4.4 rounds).
//
// Staging units per K step: A_0, A_1, A_2 (the i-th 32-row block of both wave rows: 64 rows, 8 KiB) and B (384 rows,
// 48 KiB); two stages = 144 KiB.  Nine accumulator blocks (144 VGPRs) leave room for the fragments of only two of the
// four k-slices at a time, so a K step is SIX phases q = 3 kh + i (k half kh, accumulator row block i), 6 MFMA each:
//     q = 0, 3 read B (k half) + A_0;   q = 1, 4 read A_1;   q = 2, 5 read A_2
// A unit is free after its second read (B and A_0 after q = 3, A_1 after 4, A_2 after 5) and may be refilled two
// phases later (WAR with the two wave groups one barrier apart, see gemm_pp.hip).  DMA pieces per wave (1 KiB each):
//     q=0: B pieces 2,3 (t+1)   q=1: B pieces 4,5 + A_0 (t+1)   q=2: A_1, A_2 (t+1)   q=3, 4: -   q=5: B pieces 0,1 (t+2)
// s_waitcnt vmcnt is counted and placed one phase before the FIRST read of a unit (q=5: B + A_0 of t+1, q=0: A_1,
// q=1: A_2), so every request has >= 4 phases to land (RAW: the read follows the barrier after both groups' waits).
// K-strided operands: A units are [64 k][64 rows] (128-B k rows, 32-B granule ^ 2*((k>>1)&1)), B is
// [64 k][384 rows] (768-B k rows, granule ^ 2*(k&3) inside aligned groups of 8), both read with ds_read_b64_tr_b16.
#include <type_traits>
#include "gemm_common.hpp"
#include <stdlib.h>

#include "tile_loaders.hpp"

// Lab-bench switches exist only in builds made with -DWAVLM_EXPERIMENTAL (tools/probe/build_probe.py); the product build
// has P3_PROBE == 0.  Timing probes (results are wrong by construction): bit 0 skips the K loop and the prologue DMA.
#if !defined(WAVLM_EXPERIMENTAL)
#undef P3_PROBE
#endif
#ifndef P3_PROBE
#define P3_PROBE 0
#endif
#define P3_AU 8192
#define P3_BB 49152
#define P3_STAGE (3 * P3_AU + P3_BB)  // 73728
#ifndef P3_AUX_PRELOAD
#define P3_AUX_PRELOAD 1
#endif
#ifndef P3_AUX_UNROLL
#define P3_AUX_UNROLL 0
#endif
#ifndef P3_BIAS_PRELOAD
#define P3_BIAS_PRELOAD 1   // the lane's three bias chunks of a tile in registers (one dependent load per CHUNK before)
#endif

__device__ __attribute__((aligned(256))) unsigned char g_pp3_zero[256];  // zero page for K positions past the end

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((address_space(3))) bf16x4_t* lds_b4_ptr;

struct P3Cursor { long off; int kt; };

template <bool TA, bool TB, int EP>
__global__ __launch_bounds__(512) void gemm_pp3_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 * P3_STAGE

  // Persistent launch (as gemm_pp.hip): gridDim.x workgroups (at most one per CU) walk the virtual ids vid = block,
  // block + grid, ...; vid -> (tile, batch z, split).  A launch of several rounds (N = 2304 / 3072 at 24 k rows: 2.9 / 3.9)
  // no longer pays a workgroup launch per tile, and a tile's epilogue stores drain under the next tile's first DMA.
  if (p.skew) {
    // de-phase the CUs of a multi-round launch: in lockstep every CU reaches its HBM-bound epilogue (and the next tile's
    // cold prologue) at the same time and the memory system alternates between idle and saturated
    for (int i = (int)((blockIdx.x >> 3) & 3) * p.skew; i > 0; --i) __builtin_amdgcn_s_sleep(127);
  }
  for (int vid = blockIdx.x; vid < p.vtotal; vid += gridDim.x) {
  int tile;
  const int ntile_ = p.tiles_m * p.tiles_n;
  {
    const int nt = ntile_, bid = vid % ntile_;
    const int q = nt >> 3, rem = nt & 7, xcd = bid & 7, idx = bid >> 3;
    tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
  }
  int tm, tn;
  gemm_tile_rc(tile, p.tiles_m, p.tiles_n, p.swz_r, tm, tn);
  const int z = (vid / ntile_) % p.nbatch, split = vid / (ntile_ * p.nbatch);
  const int zo = z / p.batch_i, zi = z % p.batch_i;
  const int m0 = tm * 192, n0 = tn * 384;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // panel bases and K range pinned to scalar registers: the DMA asm below takes "SGPR base + 32-bit VGPR offset"
  auto uni_ptr = [](const char* q_) __attribute__((always_inline)) {
    const unsigned long v = (unsigned long)q_;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long)hi << 32) | lo);
  };
  const char* Ab = uni_ptr((const char*)((const bf16_t*)p.A + (long)zo * p.sA_o + (long)zi * p.sA_i + (TA ? (long)m0 : (long)m0 * p.lda)));
  const char* Bb = uni_ptr((const char*)((const bf16_t*)p.B + (long)zo * p.sB_o + (long)zi * p.sB_i + (TB ? (long)n0 : (long)n0 * p.ldb)));
  const int kt_per = (p.K + 63) >> 6;
  const int kv_last = p.K - (kt_per - 1) * 64;
  int t0, t1;
  gemm_split_range(p.KB * kt_per, p.split_k, split, t0, t1);
  t0 = __builtin_amdgcn_readfirstlane(t0); t1 = __builtin_amdgcn_readfirstlane(t1);
  // LDS-DMA of one 1 KiB piece of a full tile (gemm_pp.hip): M0 = LDS destination, one wait state, then the load with
  // its address formed by the instruction itself -- no 64-bit address VALU between MFMAs
  auto dma16 = [&](const char* sbase, unsigned voff32, unsigned char* ldst) __attribute__((always_inline)) {
    const unsigned lds_dst = (unsigned)(unsigned long)(las_ptr)ldst;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(lds_dst), "v"(voff32), "s"(sbase) : "memory", "m0");
  };
  const int nt = (P3_PROBE & 1) ? 0 : t1 - t0;

  // ---- DMA side -------------------------------------------------------------------------------------------------
  // A unit i, piece = wave: K-contiguous -> buffer rows 8 w .. +8 (lane: row + (l >> 3), physical chunk l & 7);
  //                         K-strided   -> k rows 8 w .. +8 (lane: k row + (l >> 3), physical 16-B chunk l & 7 of 8)
  // B piece p = 6 w + j:    K-contiguous -> buffer rows 8 p .. +8;  K-strided -> bytes 1024 p .. of the [64][768 B] image
  // buffer row R of A unit i <-> tile row (R >> 5) * 96 + 32 i + (R & 31);  B buffer row = tile column
  unsigned voffA[3], voffB[6];
  {
    const int rows_valid = p.M - m0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (!TA) {
        const int R = wave * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((R >> 1) & 7);
        int r = (R >> 5) * 96 + i * 32 + (R & 31);
        if (r >= rows_valid) r = rows_valid - 1;
        voffA[i] = (unsigned)(((long)r * p.lda + c * 8) * 2);
      } else {
        const int kr = wave * 8 + (lane >> 3);
        const int pc = lane & 7;
        const int R = (((pc >> 1) ^ (2 * ((kr >> 1) & 1))) << 4) + ((pc & 1) << 3);
        int r = (R >> 5) * 96 + i * 32 + (R & 31);
        if (r + 8 > rows_valid) r = rows_valid - 8;
        voffA[i] = (unsigned)(((long)kr * p.lda + r) * 2);
      }
    }
    const int cols_valid = p.N - n0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int pi = wave * 6 + j;
      if (!TB) {
        const int R = pi * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((R >> 1) & 7);
        int r = R;
        if (r >= cols_valid) r = cols_valid - 1;
        voffB[j] = (unsigned)(((long)r * p.ldb + c * 8) * 2);
      } else {
        const int o = pi * 1024 + lane * 16;
        const int kr = o / 768, pc = (o % 768) >> 4;
        const int pg = pc >> 1;
        const int g = (pg & ~7) | ((pg & 7) ^ (2 * (kr & 3)));
        int r = g * 16 + ((pc & 1) << 3);
        if (r + 8 > cols_valid) r = cols_valid - 8;
        voffB[j] = (unsigned)(((long)kr * p.ldb + r) * 2);
      }
    }
  }
  const long stepA = TA ? 64 * p.lda : 64, stepB = TB ? 64 * p.ldb : 64;
  const long jumpA = p.sA_kb - (long)kt_per * stepA, jumpB = p.sB_kb - (long)kt_per * stepB;
  // streams: 0..2 = A_0..A_2, 3..5 = B pieces {0,1}, {2,3}, {4,5}
  P3Cursor cur[6];
  {
    const int kb0 = t0 / kt_per, kt0 = t0 - kb0 * kt_per;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      cur[q].off = q < 3 ? (long)kb0 * p.sA_kb + (long)kt0 * stepA : (long)kb0 * p.sB_kb + (long)kt0 * stepB;
      cur[q].kt = kt0;
    }
  }
  // first k position (K-contiguous) / k row (K-strided) this lane's 16 bytes of a piece cover -- only needed on K tails
  auto kidx_a = [&]() __attribute__((always_inline)) -> int {
    if (!TA) { const int R = wave * 8 + (lane >> 3); return ((lane & 7) ^ ((R >> 1) & 7)) * 8; }
    return wave * 8 + (lane >> 3);
  };
  auto kidx_b = [&](int j) __attribute__((always_inline)) -> int {
    const int pi = wave * 6 + j;
    if (!TB) { const int R = pi * 8 + (lane >> 3); return ((lane & 7) ^ ((R >> 1) & 7)) * 8; }
    return (pi * 1024 + lane * 16) / 768;
  };
  auto issue_gen = [&](auto which_c, int stage) __attribute__((always_inline)) {
    constexpr int W = decltype(which_c)::value;
    P3Cursor& c = cur[W];
    const int kv = (c.kt == kt_per - 1) ? kv_last : 64;
    if constexpr (W < 3) {
      const char* src = Ab + c.off * 2 + voffA[W];
      if (kv < 64 && !(kidx_a() < kv)) src = (const char*)g_pp3_zero;
      __builtin_amdgcn_global_load_lds((gas_ptr)src, (las_ptr)(smem + stage * P3_STAGE + W * P3_AU + wave * 1024), 16, 0, 0);
      c.off += stepA;
      if (++c.kt == kt_per) { c.kt = 0; c.off += jumpA; }
    } else {
      constexpr int J0 = 2 * (W - 3);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const char* src = Bb + c.off * 2 + voffB[J0 + j];
        if (kv < 64 && !(kidx_b(J0 + j) < kv)) src = (const char*)g_pp3_zero;
        __builtin_amdgcn_global_load_lds((gas_ptr)src, (las_ptr)(smem + stage * P3_STAGE + 3 * P3_AU + (wave * 6 + J0 + j) * 1024),
                                         16, 0, 0);
      }
      c.off += stepB;
      if (++c.kt == kt_per) { c.kt = 0; c.off += jumpB; }
    }
  };
  // Steady-state form (gemm_pp.hip): full tile inside one K batch, one instruction per call so that it can be placed
  // between the MFMAs of the issuing wave's own section.
  auto issue_one = [&](auto which_c, int stage, int j) __attribute__((always_inline)) {
    constexpr int W = decltype(which_c)::value;
    P3Cursor& c = cur[W];
    if constexpr (W < 3) {
      dma16(Ab + c.off * 2, voffA[W], smem + stage * P3_STAGE + W * P3_AU + wave * 1024);
      c.off += stepA;
      if (++c.kt == kt_per) { c.kt = 0; c.off += jumpA; }
    } else {
      constexpr int J0 = 2 * (W - 3);
      dma16(Bb + c.off * 2, voffB[J0 + j], smem + stage * P3_STAGE + 3 * P3_AU + (wave * 6 + J0 + j) * 1024);
      if (j == 1) {
        c.off += stepB;
        if (++c.kt == kt_per) { c.kt = 0; c.off += jumpB; }
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>;
  using I5 = std::integral_constant<int, 5>;

  // ---- fragment side ---------------------------------------------------------------------------------------------
  // one copy of every per-lane fragment address per LDS stage (gemm_pp.hip: a stage-1 read otherwise needs an offset
  // beyond the 16-bit field of ds_read and costs a v_add_u32 per fragment)
  unsigned kc_a[2][4], kc_b[2][4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const unsigned lo = (unsigned)((lane & 31) * 128 + (((2 * s + (lane >> 5)) ^ ((lane >> 1) & 7)) << 4));
    kc_a[0][s] = lo + wm * 4096;
    kc_b[0][s] = lo + wn * 12288;
    kc_a[1][s] = kc_a[0][s] + P3_STAGE;
    kc_b[1][s] = kc_b[0][s] + P3_STAGE;
    if constexpr (!TA) asm volatile("" : "+v"(kc_a[1][s]));
    if constexpr (!TB) asm volatile("" : "+v"(kc_b[1][s]));
  }
  unsigned tr_a[2], tr_b[2][3];
  {
    const int i = lane & 15, g1 = (lane >> 4) & 1, o = lane >> 5;
    tr_a[0] = (unsigned)((8 * o + (i >> 2)) * 128 + (((wm * 2 + g1) ^ (2 * ((i >> 3) & 1))) << 5) + (i & 3) * 8);
    tr_a[1] = tr_a[0] + P3_STAGE;
    if constexpr (TA) asm volatile("" : "+v"(tr_a[1]));
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int gB = wn * 6 + j * 2 + g1;
      tr_b[0][j] = (unsigned)((8 * o + (i >> 2)) * 768 + (((gB & ~7) | ((gB & 7) ^ (2 * (i >> 2)))) << 5) + (i & 3) * 8);
      tr_b[1][j] = tr_b[0][j] + P3_STAGE;
      if constexpr (TB) asm volatile("" : "+v"(tr_b[1][j]));
    }
  }
  auto rd_a = [&](int base, int s) __attribute__((always_inline)) -> bf16x8_t {  // base: byte offset of the A unit
    const int st = base >= P3_STAGE ? 1 : 0, ib = base - st * P3_STAGE;  // `base` is a constant at every call site
    if constexpr (!TA) {
      return *reinterpret_cast<const bf16x8_t*>(smem + kc_a[st][s] + ib);
    } else {
      const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_ptr)(smem + tr_a[st] + (ib + s * 2048)));
      const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_ptr)(smem + tr_a[st] + (ib + s * 2048 + 512)));
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };
  auto rd_b = [&](int base, int j, int s) __attribute__((always_inline)) -> bf16x8_t {  // base: byte offset of B
    const int st = base >= P3_STAGE ? 1 : 0, ib = base - st * P3_STAGE;
    if constexpr (!TB) {
      return *reinterpret_cast<const bf16x8_t*>(smem + kc_b[st][s] + (ib + j * 4096));
    } else {
      const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_ptr)(smem + tr_b[st][j] + (ib + s * 12288)));
      const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_ptr)(smem + tr_b[st][j] + (ib + s * 12288 + 3072)));
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };

  f32x16_t acc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  bf16x8_t bf[3][2], af[2];

#define P3_MFMA_SECTION(I_) P3_MFMA_SECTION_H(I_, (void)0, (void)0, (void)0)
#define P3_MFMA_SECTION_H(I_, H0_, H1_, H2_)                                                          \
  __builtin_amdgcn_sched_barrier(0);                                                                \
  __builtin_amdgcn_s_barrier();                                                                     \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
  __builtin_amdgcn_sched_barrier(0);                                                                \
  __builtin_amdgcn_s_setprio(1);                                                                    \
  _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                   \
    _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                   \
    {                                                                                               \
      acc[I_][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], bf[j][s], acc[I_][j], 0, 0, 0);    \
      if (s == 0 && j == 1) { H0_; }                                                                \
      if (s == 1 && j == 0) { H1_; }                                                                \
      if (s == 1 && j == 1) { H2_; }                                                                \
    }                                                                                               \
  }                                                                                                 \
  __builtin_amdgcn_s_setprio(0);                                                                    \
  __builtin_amdgcn_sched_barrier(0);                                                                \
  __builtin_amdgcn_s_barrier();                                                                     \
  __builtin_amdgcn_sched_barrier(0);

  auto k_step = [&](auto stage_c, auto steady_c, int t) __attribute__((always_inline)) {
    constexpr int ST = decltype(stage_c)::value;
    constexpr int SB = ST * P3_STAGE;
    constexpr bool SD = decltype(steady_c)::value;  // steady: tiles t+1, t+2 exist, are full, and share the K batch
    auto issue = [&](auto which_c, int stage) __attribute__((always_inline)) { issue_gen(which_c, stage); };
    if constexpr (SD) {
      // same units and order as below, each DMA instruction issued inside the wave's own MFMA section; the waits precede
      // the phase's own issue (q=0: A_1(t), younger A_2(t) + B 0-1 (t+1) = 3; q=1: A_2(t), younger B 0-3 = 4;
      // q=5: B + A_0 of t+1, younger A_1, A_2 (t+1) = 2)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int s = 0; s < 2; ++s) bf[j][s] = rd_b(SB + 3 * P3_AU, j, 2 * kh + s);
#pragma unroll
        for (int s = 0; s < 2; ++s) af[s] = rd_a(SB, 2 * kh + s);
        if (kh == 0) {
          asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
          P3_MFMA_SECTION_H(0, issue_one(I4{}, ST ^ 1, 0), issue_one(I4{}, ST ^ 1, 1), (void)0)
        } else {
          P3_MFMA_SECTION(0)
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) af[s] = rd_a(SB + P3_AU, 2 * kh + s);
        if (kh == 0) {
          asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          P3_MFMA_SECTION_H(1, issue_one(I5{}, ST ^ 1, 0), issue_one(I5{}, ST ^ 1, 1), issue_one(I0{}, ST ^ 1, 0))
        } else {
          P3_MFMA_SECTION(1)
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) af[s] = rd_a(SB + 2 * P3_AU, 2 * kh + s);
        if (kh == 0) {
          P3_MFMA_SECTION_H(2, issue_one(I1{}, ST ^ 1, 0), issue_one(I2{}, ST ^ 1, 0), (void)0)
        } else {
          asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
          P3_MFMA_SECTION_H(2, issue_one(I3{}, ST, 0), issue_one(I3{}, ST, 1), (void)0)
        }
      }
      return;
    }
    const bool more1 = t + 1 < nt, more2 = t + 2 < nt;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      // ---- q = 3 kh: B (this k half), A_0
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s) bf[j][s] = rd_b(SB + 3 * P3_AU, j, 2 * kh + s);
#pragma unroll
      for (int s = 0; s < 2; ++s) af[s] = rd_a(SB, 2 * kh + s);
      if (kh == 0) {  // wait: A_1(t) landed; younger: A_2(t) [, B 0-3 of t+1]
        if (more1) {
          issue(I4{}, ST ^ 1);
          asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        }
      }
      P3_MFMA_SECTION(0)
      // ---- q = 3 kh + 1: A_1
#pragma unroll
      for (int s = 0; s < 2; ++s) af[s] = rd_a(SB + P3_AU, 2 * kh + s);
      if (kh == 0) {  // wait: A_2(t) landed; younger: B (6 pieces) and A_0 of t+1
        if (more1) {
          issue(I5{}, ST ^ 1);
          issue(I0{}, ST ^ 1);
          asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      P3_MFMA_SECTION(1)
      // ---- q = 3 kh + 2: A_2
#pragma unroll
      for (int s = 0; s < 2; ++s) af[s] = rd_a(SB + 2 * P3_AU, 2 * kh + s);
      if (kh == 0) {
        if (more1) { issue(I1{}, ST ^ 1); issue(I2{}, ST ^ 1); }
      } else {  // wait: B and A_0 of t+1 landed (read by q = 0 of step t+1); younger: A_1, A_2 (t+1) [, B 0-1 of t+2]
        if (more2) {
          issue(I3{}, ST);
          asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else if (more1) {
          asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
      }
      P3_MFMA_SECTION(2)
    }
  };

  if (nt > 0) {
    issue_gen(I3{}, 0); issue_gen(I4{}, 0); issue_gen(I5{}, 0); issue_gen(I0{}, 0); issue_gen(I1{}, 0); issue_gen(I2{}, 0);
    if (nt > 1) {
      issue_gen(I3{}, 1);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // B(0), A_0(0) landed; younger: A_1, A_2 (0), B 0-1 (1)
    } else {
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind group 0
    const int has_tail = (kv_last < 64 && t1 == p.KB * kt_per) ? 1 : 0;
    // (K batches without a K tail are steady too: the in-section issue carries the cursor wrap)
    const int n_steady = (kv_last == 64 || p.KB == 1) ? max(0, nt - 2 - has_tail) & ~1 : 0;
    int t = 0;
    for (; t < n_steady; t += 2) {
      k_step(I0{}, std::true_type{}, t);
      k_step(I1{}, std::true_type{}, t + 1);
    }
    for (; t < nt; t += 2) {
      k_step(I0{}, std::false_type{}, t);
      if (t + 1 < nt) k_step(I1{}, std::false_type{}, t + 1);
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
  }
#undef P3_MFMA_SECTION
#undef P3_MFMA_SECTION_H
  __syncthreads();

  // ---- epilogue ----------------------------------------------------------------------------------------------------
  const int mw = m0 + wm * 96, nw = n0 + wn * 96;  // wave tile origin; block i -> rows 32 i, j -> cols 32 j
  constexpr int EPX = EP;
  if constexpr (EP == 0) {
    auto store_block = [&](const f32x16_t (&a)[3], int i) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int nn = nw + j * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int mm = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (mm < p.M && nn < p.N) gemm_store(p, z, split, mm, nn, a[j][r]);
        }
      }
    };
    store_block(acc[0], 0); store_block(acc[1], 1); store_block(acc[2], 2);
  } else {
    constexpr int EP_LD = 96 + 4;
    float* ep = reinterpret_cast<float*>(smem) + wave * (32 * EP_LD);
    // GELU epilogue: chord table behind the staging slices (8 x 12800 B = 100 KiB) at 104 KiB of the 144 KiB
    const float4* tab = nullptr;
    if constexpr (EPX == 3) {
      if (p.gtab) {
        float4* tl = reinterpret_cast<float4*>(smem + 106496);
        gelu_tab_stage(p.gtab, tl);
        __syncthreads();
        tab = tl;
      }
    }
    auto stage_block = [&](const f32x16_t (&a)[3]) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          ep[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * EP_LD + j * 32 + (lane & 31)] = a[j][r];
    };
    constexpr bool CSUM = EPX == 2 || EPX == 4;  // fused column sums of C: plain and aux-multiplying fast epilogues
    const bool csum = CSUM && p.colsum_part != nullptr;
    float cs[2] = {0.f, 0.f};
    // one 32-row block of the wave's tile out of the staging slice: six 8-wide chunks per lane.  axp (EP 4 with
    // P3_AUX_PRELOAD): the block's aux chunks, loaded by aux_load() ahead of time
    // chunk q of block i: ax = its preloaded aux values (EP 4 with P3_AUX_PRELOAD)
    // The tile's bias values: the wave's 96 columns once into a wave-private LDS slice (the 4 KiB between the staging slices and
    // the table) instead of one dependent global load inside each of the tile's 18 chunks (5.5 us of a forward launch,
    // tools/gemm_aux_bound.py); a chunk reads its eight floats back with two LDS reads.
    float* bl = reinterpret_cast<float*>(smem + 102400) + wave * 96;
    const bool bias_pre = P3_BIAS_PRELOAD && (EPX == 2 || EPX == 3) && p.bias != nullptr;  // (the x-aux epilogue belongs to data gradients: no bias)
    if (bias_pre) {
      if (lane < 12) {
        const int n0 = nw + lane * 8;
        float b8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (n0 < p.N) {
          const uint4 raw = *reinterpret_cast<const uint4*>((const bf16_t*)p.bias + (long)zo * p.sBias_o + (long)zi * p.sBias_i + n0);
          gemm_unpack8(raw, b8);
        }
        *reinterpret_cast<float4*>(bl + lane * 8) = make_float4(b8[0], b8[1], b8[2], b8[3]);
        *reinterpret_cast<float4*>(bl + lane * 8 + 4) = make_float4(b8[4], b8[5], b8[6], b8[7]);
      }
    }
    // ax: the chunk's preloaded aux values (EP 4: has_pre)
    auto process_chunk = [&](int i, int q, const uint4& ax, auto has_pre) __attribute__((always_inline)) {
      const int id = lane + 64 * q;   // 32 rows x 12 chunks
      const int rl = id / 12, ch = id - rl * 12;
      const int mm = mw + i * 32 + rl;
      const int nn = nw + ch * 8;
      const float* bqp = bias_pre ? bl + ch * 8 : nullptr;
      float vo[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (mm < p.M && nn < p.N) {
        const float4 lo = *reinterpret_cast<const float4*>(ep + rl * EP_LD + ch * 8);
        const float4 hi = *reinterpret_cast<const float4*>(ep + rl * EP_LD + ch * 8 + 4);
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        if constexpr (EPX == 1) gemm_store8(p, zo, zi, z, split, mm, nn, v);
        else if constexpr (P3_AUX_PRELOAD && EPX == 4) gemm_store8_fast<4>(p, zo, zi, mm, nn, v, tab, CSUM ? vo : nullptr, &ax, bqp);
        else gemm_store8_fast<(EPX == 2 ? 0 : EPX)>(p, zo, zi, mm, nn, v, tab, CSUM ? vo : nullptr, nullptr, bqp);
      }
      if constexpr (CSUM) {
        if (csum) {
          *reinterpret_cast<float4*>(ep + rl * EP_LD + ch * 8) = make_float4(vo[0], vo[1], vo[2], vo[3]);
          *reinterpret_cast<float4*>(ep + rl * EP_LD + ch * 8 + 4) = make_float4(vo[4], vo[5], vo[6], vo[7]);
        }
      }
    };
    // one 32-row block of the wave's tile out of the staging slice: six 8-wide chunks per lane
    auto process_block = [&](int i, const uint4 (&axp)[6], auto has_pre) __attribute__((always_inline)) {
      if constexpr (P3_AUX_PRELOAD && P3_AUX_UNROLL && EPX == 4) {
#pragma unroll
        for (int q = 0; q < 6; ++q) process_chunk(i, q, axp[q], has_pre);
      } else {
#pragma unroll 1
        for (int q = 0; q < 6; ++q) {  // not unrolled: with 144 accumulator registers live the store code must stay small
          uint4 ax = axp[0];
          if constexpr (decltype(has_pre)::value) {   // (a chain of selects: a run-time index would put the array into scratch)
#pragma unroll
            for (int k = 1; k < 6; ++k) {
              ax.x = q == k ? axp[k].x : ax.x; ax.y = q == k ? axp[k].y : ax.y;
              ax.z = q == k ? axp[k].z : ax.z; ax.w = q == k ? axp[k].w : ax.w;
            }
          }
          process_chunk(i, q, ax, has_pre);
        }
      }
      if constexpr (CSUM) {
        if (csum) gemm_colsum_block<96, EP_LD>(ep, lane, cs);
      }
    };
    if constexpr (P3_AUX_PRELOAD && EPX == 4) {
      // The aux chunks (GELU' of fc1's / a conv layer's pre-activation, written a forward pass ago) travel one block ahead in two
      // register sets: with the load inside the chunk loop every chunk waited for its own round trip -- 18 dependent round trips
      // per wave and tile, 47 of the 193 us of fc2's dX launch (tools/gemm_aux_bound.py, profiles/r05/ab_gemm_aux_preload.txt).
      // Block i + 1's six loads are issued when block i has been staged (its 48 accumulator registers are free by then).
      auto aux_load = [&](int i, uint4 (&a)[6]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          const int id = lane + 64 * q;
          const int rl = id / 12, ch = id - rl * 12;
          const int mm = mw + i * 32 + rl, nn = nw + ch * 8;
          a[q] = make_uint4(0, 0, 0, 0);
          if (mm < p.M && nn < p.N)
            a[q] = *reinterpret_cast<const uint4*>((const bf16_t*)p.aux + (long)zo * p.sAux_o + (long)zi * p.sAux_i + (long)mm * p.ld_aux + nn);
        }
      };
      uint4 a0[6], a1[6];
      aux_load(0, a0);
      stage_block(acc[0]);
      aux_load(1, a1);
      process_block(0, a0, std::true_type{});
      stage_block(acc[1]);
      aux_load(2, a0);
      process_block(1, a1, std::true_type{});
      stage_block(acc[2]);
      process_block(2, a0, std::true_type{});
    } else {
      // three 32-row blocks, written out: as a rolled loop over the block index (a switch selects the accumulator block) the
      // same code ran fc1's forward launch at 131 us, written out at 127 us (profiles/r05/ab_gemm_bias_preload.txt)
      const uint4 none[6] = {};
      stage_block(acc[0]); process_block(0, none, std::false_type{});
      stage_block(acc[1]); process_block(1, none, std::false_type{});
      stage_block(acc[2]); process_block(2, none, std::false_type{});
    }
    if constexpr (CSUM) {
      if (csum) gemm_colsum_finish<96>(p, reinterpret_cast<float*>(smem + 106496), wave, wm, lane, tm, nw, cs);
    }
  }
  __syncthreads();  // every wave has read its staging slice (and the table) before the next tile's DMA lands there
  }
}

extern int g_pp_reserved_cus;  // gemm_pp.hip: CUs left to the RCCL kernels in data-parallel runs

template <bool TA, bool TB>
static int pp3_launch_t(GemmP& p, int nbatch, int ep, hipStream_t st) {
  p.tiles_m = (p.M + 191) / 192;
  p.tiles_n = (p.N + 383) / 384;
  p.nbatch = nbatch;
  p.vtotal = p.tiles_m * p.tiles_n * nbatch * p.split_k;
  p.swz_r = (nbatch == 1 && p.split_k == 1) ? gemm_pick_swizzle(192, 384, p.tiles_m, p.tiles_n, (long)p.N * p.K * 2) : 0;
  const int pgrid = 256 - g_pp_reserved_cus;
  dim3 grid((unsigned)(p.vtotal < pgrid ? p.vtotal : pgrid), 1, 1);   // persistent: a tile's epilogue stores drain under the next tile's DMA
#if defined(WAVLM_EXPERIMENTAL)
  // WAVLM_PP3_SKEW=n: start phases 0..3 x n x ~3.9 us (s_sleep 127) for launches of >= 2 rounds (measured: no gain, profiles/HISTORY.md)
  static const int skew = getenv("WAVLM_PP3_SKEW") ? atoi(getenv("WAVLM_PP3_SKEW")) : 0;
  p.skew = (skew > 0 && p.vtotal >= 2 * pgrid) ? skew : 0;
#endif
  constexpr int smem = 2 * P3_STAGE;
  static bool done[5] = {false, false, false, false, false};
#define PP_CASE(E) case E: { \
    if (!done[E]) { \
      if (hipFuncSetAttribute((const void*)gemm_pp3_kernel<TA, TB, E>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return WL_ELAUNCH; \
      done[E] = true; \
    } \
    WL_LAUNCH((gemm_pp3_kernel<TA, TB, E>), grid, dim3(512), smem, st, p); } break;
  switch (ep) { PP_CASE(0) PP_CASE(1) PP_CASE(2) PP_CASE(3) default: PP_CASE(4) }
#undef PP_CASE
  return wl_check_launch();
}

bool gemm_pp3_ok(const wavlm_gemm_desc* d) {
  if (d->M < 192 || d->N < 192 || d->K < 64) return false;
  if (!d->transA && d->K % 8) return false;
  if (!d->transB && d->K % 8) return false;
  if (d->transA && d->M % 8) return false;
  if (d->transB && d->N % 8) return false;
  const int64_t ra = d->transA ? 64 : 192, rb = d->transB ? 64 : 384;
  if (ra * d->lda * 2 >= (1ll << 31) || rb * d->ldb * 2 >= (1ll << 31)) return false;
  return true;
}

int gemm_pp3_launch(GemmP& p, int nbatch, bool transA, bool transB, int ep, hipStream_t st) {
  if (!transA && !transB) return pp3_launch_t<false, false>(p, nbatch, ep, st);
  if (!transA && transB) return pp3_launch_t<false, true>(p, nbatch, ep, st);
  if (transA && !transB) return pp3_launch_t<true, false>(p, nbatch, ep, st);
  return pp3_launch_t<true, true>(p, nbatch, ep, st);
}
