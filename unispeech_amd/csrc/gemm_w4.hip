// bf16 MFMA GEMM for gfx950, long-K path: the 256 x 256 x 64 block tile of gemm_pp.hip computed by FOUR waves (one per
// SIMD, 128 x 128 accumulators each) instead of eight.
//
// Why: gemm_pp.hip's K loop is power-limited (shader clock 1.93 GHz without the operand DMA, 1.53 GHz with it,
// profiles/HISTORY.md section 4.1); what is left to save is energy per flop.  A 128 x 128 wave tile reads (128 + 128) x 64
// operand elements from LDS per 64 MFMAs -- 0.5 ds_read_b128 per MFMA against 0.75 for the 128 x 64 tiles of the eight-wave
// kernel, i.e. 128 KiB instead of 192 KiB of LDS reads per K step and CU.  The vendor library's kernel for the same block
// tile reaches 1.54 PF/s at 8192^3 where the eight-wave ping-pong reaches 1.28-1.37.
//
// What is the same as gemm_pp.hip (its header has the details): LDS-DMA for both operand layouts with the swizzles on the
// source address, the half-tile staging [A-top][A-bot][B-left][B-right] x 2 stages = 128 KiB, counted vmcnt waits, the
// persistent XCD-aware work order, the grouped split-K launch, the K-tail / edge handling, the row-vector epilogues.
// The LDS image is IDENTICAL (a wave here owns two of the eight-wave kernel's wave columns), so the DMA side only
// distributes the eight 2 KiB slices of a half-tile over four waves.
//
// What is different: with one wave per SIMD there is no partner wave to own the MFMA pipe while this one reads LDS, so the
// fragment reads are software-pipelined inside the wave -- the reads of phase q + 1 are issued BETWEEN the MFMAs of phase
// q (one ds_read per MFMA), as are the four DMA instructions of the phase:
//     P4(t-1): 16 MFMA A-bot(t-1) x B-left(t-1) | reads A-top(t), B-left(t)        | DMA B-left(t+1)
//     P1(t)  : 16 MFMA A-top x B-left            | reads B-right(t)                 | DMA B-right(t+1)
//     P2(t)  : 16 MFMA A-top x B-right           | reads A-bot(t)                   | DMA A-bot(t+1)
//     P3(t)  : 16 MFMA A-bot x B-right           | -                                | DMA A-top(t+2)
// Each phase starts with [s_waitcnt vmcnt(12)] s_barrier, s_waitcnt lgkmcnt(0) (the fragments read a phase ago).
// Hazards: RAW -- the half-tile read in phase q was waited for (own pieces) before the barrier that opens phase q, by every
// wave; WAR -- reads of a buffer issued in phase q are complete at each wave's lgkmcnt(0) of phase q + 1, i.e. before the
// barrier that opens phase q + 2, and no buffer is refilled earlier than two phases after its reads were issued.
// B-left fragments are double-buffered by K-step parity (P4 uses B-left(t) while B-left(t+1) is being read).
#include "gemm_common.hpp"
// the balanced grouped launch (gemm_common.hpp: gemm_sk_plan) is a lab path: decoded only in -DWAVLM_EXPERIMENTAL builds
#if defined(WAVLM_EXPERIMENTAL)
#define GEMM_SK 1
#else
#define GEMM_SK 0
#endif

#include "tile_loaders.hpp"

#define W4_HB 16384     // bytes per half-tile buffer
#define W4_STAGE 65536  // bytes per stage: [A-top][A-bot][B-left][B-right]

__device__ __attribute__((aligned(256))) unsigned char g_w4_zero[256];  // K positions past the end are fetched from here
extern int g_pp_reserved_cus;

typedef __attribute__((ext_vector_type(4))) __bf16 w4_bf16x4_t;
typedef __attribute__((address_space(3))) w4_bf16x4_t* w4_lds_b4_ptr;

struct W4Cursor { long off; int kt; };

// TAILS: as in gemm_pp.hip -- K batches that each end in a K tail run the steady schedule (zero-page form of a tail piece)
template <bool TA, bool TB, int EP, bool GRP = false, bool TAILS = false>
__global__ __launch_bounds__(256) void gemm_w4_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 * W4_STAGE

  const int ntile = p.tiles_m * p.tiles_n;
  for (int vid = blockIdx.x; vid < p.vtotal; vid += gridDim.x) {
  int tm, tn, z, split;
  int fix_uid = 0;
  int sk_t0 = 0, sk_t1 = 0;
  GemmP q = p;
  if constexpr (GRP) {
    int item, local, g;
    if ((GEMM_SK && p.sk_ks > 0)) {
      // balanced launch (gemm_common.hpp: gemm_sk_plan): vid = physical workgroup + G * (which of a tail workgroup's tiles)
      const int G = p.sk_wgs, T = p.sk_tiles, S = p.sk_s;
      const int wp = vid % G, seg = vid / G;
      const int q8 = G >> 3, rem = G & 7, xcd = wp & 7, idx = wp >> 3;
      if (p.sk_spread) {   // every XCD hosts S T / 8 main and R / 8 tail workgroups (both divisible: checked on the host)
        const int mx = S * T / 8, tx = (G - S * T) / 8;
        item = idx < mx ? xcd * mx + idx : S * T + xcd * tx + (idx - mx);
      } else
      item = (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + idx;   // as the one-round split: neighbouring tiles share an XCD
      int tile_all;
      if (item < S * T) {
        if (seg > 0) continue;
        split = item / T; tile_all = item - split * T;
        sk_t0 = split * p.sk_lm; sk_t1 = sk_t0 + p.sk_lm;
      } else {
        const int e = item - S * T, R = G - S * T;
        const int first = (int)((long)e * T / R), last = (int)((long)(e + 1) * T / R);
        tile_all = first + seg;
        if (tile_all >= last) continue;
        split = S; sk_t0 = S * p.sk_lm; sk_t1 = p.sk_ks;
      }
      g = (tile_all >= p.grp[1].vbase) + (tile_all >= p.grp[2].vbase) + (tile_all >= p.grp[3].vbase);
      local = tile_all;
    } else {
      const int nv = p.vtotal;
      const int q8 = nv >> 3, rem = nv & 7, xcd = vid & 7, idx = vid >> 3;
      item = (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + idx;
      g = (item >= p.grp[1].vbase) + (item >= p.grp[2].vbase) + (item >= p.grp[3].vbase);
      local = item;
    }
#define W4_SEL(F) (g == 0 ? p.grp[0].F : g == 1 ? p.grp[1].F : g == 2 ? p.grp[2].F : p.grp[3].F)
    q.A = W4_SEL(A); q.B = W4_SEL(B); q.ws = W4_SEL(ws); q.lda = W4_SEL(lda); q.ldb = W4_SEL(ldb);
    q.M = W4_SEL(M); q.N = W4_SEL(N); q.tiles_m = W4_SEL(tiles_m); q.tiles_n = W4_SEL(tiles_n);
    q.C = W4_SEL(C); q.ldc = W4_SEL(ldc); q.c_dtype = W4_SEL(c_dtype); q.accumulate = W4_SEL(accumulate); q.alpha = W4_SEL(alpha);
    local -= W4_SEL(vbase);
#undef W4_SEL
    const int nt_g = q.tiles_m * q.tiles_n;
    int tile;
    if ((GEMM_SK && p.sk_ks > 0)) tile = local;
    else { split = local / nt_g; tile = local - split * nt_g; }
    tm = tile / q.tiles_n; tn = tile - tm * q.tiles_n;
    z = 0;
    fix_uid = item - split * nt_g;   // one id per (member, tile): members' item ranges are disjoint
  } else
  if (p.patch_m == 0) {
    const int nt = ntile, bid = vid % ntile;
    const int q_ = nt >> 3, rem = nt & 7, xcd = bid & 7, idx = bid >> 3;
    const int tile = (xcd < rem ? xcd * (q_ + 1) : rem * (q_ + 1) + (xcd - rem) * q_) + idx;
    tn = tile % p.tiles_n; tm = tile / p.tiles_n;
    z = (vid / ntile) % p.nbatch; split = vid / (ntile * p.nbatch);
  } else {
    const int nv = p.vtotal;
    const int q_ = nv >> 3, rem = nv & 7, xcd = vid & 7, idx = vid >> 3;
    const int item = (xcd < rem ? xcd * (q_ + 1) : rem * (q_ + 1) + (xcd - rem) * q_) + idx;
    const int full = p.patch_m * p.tiles_n * p.split_k;
    const int npatch = (p.tiles_m + p.patch_m - 1) / p.patch_m;
    int pi = item / full; if (pi > npatch - 1) pi = npatch - 1;
    const int r = item - pi * full;
    const int h = min(p.patch_m, p.tiles_m - pi * p.patch_m);
    split = r / (h * p.tiles_n);
    const int qq = r - split * h * p.tiles_n;
    tm = pi * p.patch_m + qq / p.tiles_n; tn = qq % p.tiles_n;
    z = 0;
  }
  const GemmP& P = q;
  const int zo = z / P.batch_i, zi = z % P.batch_i;
  const int m0 = tm * 256, n0 = tn * 256;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn2 = wave & 1;   // wave tile: rows 128 wm .. +128, columns 128 wn2 .. +128

  auto uni_ptr = [](const char* q_) __attribute__((always_inline)) {
    const unsigned long v = (unsigned long)q_;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long)hi << 32) | lo);
  };
  const char* Ab = uni_ptr((const char*)((const bf16_t*)P.A + (long)zo * P.sA_o + (long)zi * P.sA_i + (TA ? (long)m0 : (long)m0 * P.lda)));
  const char* Bb = uni_ptr((const char*)((const bf16_t*)P.B + (long)zo * P.sB_o + (long)zi * P.sB_i + (TB ? (long)n0 : (long)n0 * P.ldb)));
  const int kt_per = (P.K + 63) >> 6;
  const int kv_last = P.K - (kt_per - 1) * 64;
  int t0, t1;
  if (GRP && (GEMM_SK && p.sk_ks > 0)) { t0 = sk_t0; t1 = sk_t1; }
  else gemm_split_range(P.KB * kt_per, P.split_k, split, t0, t1);
  t0 = __builtin_amdgcn_readfirstlane(t0); t1 = __builtin_amdgcn_readfirstlane(t1);
  const int nt = t1 - t0;

  // ---- DMA side: the eight 2 KiB slices (w8) of a half-tile buffer are fetched by waves w8 & 3; slice w8, piece j as in
  // gemm_pp.hip: K-contiguous -> buffer rows 16 w8 + 8 j .. +8; K-strided -> k rows 8 w8 + 4 j .. +4
  unsigned voff[4][4];   // [half-tile][piece jj = 2 * (w8 >> 2) + j]
  int kidx[2][4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int w8 = wave + 4 * (jj >> 1), j = jj & 1;
#pragma unroll
    for (int op = 0; op < 2; ++op) {
      const bool TR = op ? TB : TA;
      const long ld = op ? P.ldb : P.lda;
      const int rows_valid = op ? P.N - n0 : P.M - m0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        unsigned v;
        if (!TR) {
          const int R = w8 * 16 + j * 8 + (lane >> 3);
          const int c = (lane & 7) ^ ((R >> 1) & 7);
          int r = op ? ((R >> 5) * 64 + h * 32 + (R & 31)) : ((R >> 6) * 128 + h * 64 + (R & 63));
          if (r >= rows_valid) r = rows_valid - 1;
          v = (unsigned)(((long)r * ld + c * 8) * 2);
          kidx[op][jj] = c * 8;
        } else {
          const int kr = w8 * 8 + j * 4 + (lane >> 4);
          const int pc = lane & 15;
          const int R = (((pc >> 1) ^ (2 * (kr & 3))) << 4) + ((pc & 1) << 3);
          int r = op ? ((R >> 5) * 64 + h * 32 + (R & 31)) : ((R >> 6) * 128 + h * 64 + (R & 63));
          if (r + 8 > rows_valid) r = rows_valid - 8;
          v = (unsigned)(((long)kr * ld + r) * 2);
          kidx[op][jj] = kr;
        }
        voff[op * 2 + h][jj] = v;
      }
    }
  }
  const long stepA = TA ? 64 * P.lda : 64, stepB = TB ? 64 * P.ldb : 64;
  const long jumpA = P.sA_kb - (long)kt_per * stepA, jumpB = P.sB_kb - (long)kt_per * stepB;
  W4Cursor cur[4];
  {
    const int kb0 = t0 / kt_per, kt0 = t0 - kb0 * kt_per;
    cur[0].off = cur[1].off = (long)kb0 * P.sA_kb + (long)kt0 * stepA;
    cur[2].off = cur[3].off = (long)kb0 * P.sB_kb + (long)kt0 * stepB;
    cur[0].kt = cur[1].kt = cur[2].kt = cur[3].kt = kt0;
  }
  auto dma16 = [&](const char* sbase, unsigned voff32, unsigned char* ldst) __attribute__((always_inline)) {
    const unsigned lds_dst = (unsigned)(unsigned long)(las_ptr)ldst;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(lds_dst), "v"(voff32), "s"(sbase) : "memory", "m0");
  };
  // piece jj of half-tile W lands at slice w8 = wave + 4 (jj >> 1), 1 KiB piece j = jj & 1
  auto piece_dst = [&](int W, int stage, int jj) __attribute__((always_inline)) -> unsigned char* {
    return smem + stage * W4_STAGE + W * W4_HB + (wave + 4 * (jj >> 1)) * 2048 + (jj & 1) * 1024;
  };
  auto advance = [&](auto which_c) __attribute__((always_inline)) {
    constexpr int W = decltype(which_c)::value;
    constexpr int OP = W >> 1;
    W4Cursor& c = cur[W];
    c.off += OP ? stepB : stepA;
    if (++c.kt == kt_per) { c.kt = 0; c.off += OP ? jumpB : jumpA; }
  };
  // general form: a whole half-tile (4 instructions), K tail through the zero page
  auto issue_gen = [&](auto which_c, int stage) __attribute__((always_inline)) {
    constexpr int W = decltype(which_c)::value;
    constexpr int OP = W >> 1;
    W4Cursor& c = cur[W];
    const char* src = (OP ? Bb : Ab) + c.off * 2;
    const int kv = (c.kt == kt_per - 1) ? kv_last : 64;
    if (kv >= 64) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) dma16(src, voff[W][jj], piece_dst(W, stage, jj));
    } else {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const bool in = kidx[OP][jj] < kv;
        const char* s = in ? src + voff[W][jj] : (const char*)g_w4_zero;
        __builtin_amdgcn_global_load_lds((gas_ptr)s, (las_ptr)piece_dst(W, stage, jj), 16, 0, 0);
      }
    }
    advance(which_c);
  };
  // steady form: ONE instruction of a full tile; the cursor moves with the last piece
  auto issue_one = [&](auto which_c, int stage, int jj) __attribute__((always_inline)) {
    constexpr int W = decltype(which_c)::value;
    constexpr int OP = W >> 1;
    const char* src = (OP ? Bb : Ab) + cur[W].off * 2;
    if (TAILS && cur[W].kt == kt_per - 1) {
      // asm with a 64-bit per-lane address, not the builtin: a builtin LDS-DMA anywhere in the K loop makes the compiler
      // drain vmcnt(0) before every later LDS read
      const bool in = kidx[OP][jj] < kv_last;
      const char* sp = in ? src + voff[W][jj] : (const char*)g_w4_zero;
      const unsigned lds_dst = (unsigned)(unsigned long)(las_ptr)piece_dst(W, stage, jj);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(lds_dst), "v"(sp) : "memory", "m0");
    } else {
      dma16(src, voff[W][jj], piece_dst(W, stage, jj));
    }
    if (jj == 3) advance(which_c);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;

  // ---- fragment side (addresses as gemm_pp.hip; the wave's two B wave-columns are 2 wn2 and 2 wn2 + 1) ------------
  unsigned kc_a[2][4], kc_b[2][4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const unsigned lo = (unsigned)((lane & 31) * 128 + (((2 * s + (lane >> 5)) ^ ((lane >> 1) & 7)) << 4));
    kc_a[0][s] = lo + wm * 8192;
    kc_b[0][s] = lo + wn2 * 8192;
    kc_a[1][s] = kc_a[0][s] + W4_STAGE;
    kc_b[1][s] = kc_b[0][s] + W4_STAGE;
    if constexpr (!TA) asm volatile("" : "+v"(kc_a[1][s]));
    if constexpr (!TB) asm volatile("" : "+v"(kc_b[1][s]));
  }
  unsigned tr_a[2][2], tr_b[2][2];
  {
    const int i = lane & 15, g1 = (lane >> 4) & 1;
    const unsigned kpart = (unsigned)((8 * (lane >> 5) + (i >> 2)) * 256 + (i & 3) * 8);
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      tr_a[0][f] = kpart + (unsigned)(((wm * 4 + f * 2 + g1) ^ (2 * (i >> 2))) << 5);
      tr_a[1][f] = tr_a[0][f] + W4_STAGE;
      if constexpr (TA) asm volatile("" : "+v"(tr_a[1][f]));
      tr_b[0][f] = kpart + (unsigned)((((2 * wn2 + f) * 2 + g1) ^ (2 * (i >> 2))) << 5);
      tr_b[1][f] = tr_b[0][f] + W4_STAGE;
      if constexpr (TB) asm volatile("" : "+v"(tr_b[1][f]));
    }
  }
  // base: byte offset of the half-tile buffer (a constant at every call site); f: 32-row block of the half-tile's 64 rows
  // (A) / wave column of the pair (B); s: k slice
  auto rd_a = [&](int base, int f, int s) __attribute__((always_inline)) -> bf16x8_t {
    const int st = base >= W4_STAGE ? 1 : 0, ib = base - st * W4_STAGE;
    if constexpr (!TA) {
      return *reinterpret_cast<const bf16x8_t*>(smem + kc_a[st][s] + (ib + f * 4096));
    } else {
      const w4_bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((w4_lds_b4_ptr)(smem + tr_a[st][f] + (ib + s * 4096)));
      const w4_bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((w4_lds_b4_ptr)(smem + tr_a[st][f] + (ib + s * 4096 + 1024)));
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };
  auto rd_b = [&](int base, int g, int s) __attribute__((always_inline)) -> bf16x8_t {
    const int st = base >= W4_STAGE ? 1 : 0, ib = base - st * W4_STAGE;
    if constexpr (!TB) {
      return *reinterpret_cast<const bf16x8_t*>(smem + kc_b[st][s] + (ib + g * 4096));
    } else {
      const w4_bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((w4_lds_b4_ptr)(smem + tr_b[st][g] + (ib + s * 4096)));
      const w4_bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((w4_lds_b4_ptr)(smem + tr_b[st][g] + (ib + s * 4096 + 1024)));
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };

  // acc[i][c]: rows 32 i (i = 0, 1: A-top; 2, 3: A-bot), columns 32 c with c = 2 g + h (g: wave column of the pair, h = 0
  // B-left / 1 B-right) -- the 128 columns of the wave in memory order
  f32x16_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  bf16x8_t at[2][4], ab[2][4], bl[2][2][4], br[2][4];   // bl[parity of the K step][g][s]

#define W4_VMWAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
  // 16 MFMAs: rows FA[f] (f = 0, 1) x columns FB[g] (g = 0, 1) over the four k slices, into acc[I0 + f][2 g + H].
  // The MFMAs only need their own fragment registers (read a phase ago: lgkmcnt(0)), so the first W4_PRE of them run BEFORE
  // the phase's barrier -- the matrix pipe works through them while the wave waits for its peers; what the barrier orders
  // (this phase's fragment reads and DMA pieces against the other waves' DMA / reads) is issued by hook(k), k = 0 .. 13,
  // after MFMA W4_PRE + k.  vm_wait(): the counted s_waitcnt vmcnt of the phase (own DMA pieces of the half-tile read next).
#define W4_PRE 2
  auto section = [&](const bf16x8_t (&FA)[2][4], int I0_, const bf16x8_t (&FB)[2][4], int H_, auto vm_wait, auto hook) __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int m = 4 * s + 2 * f + g;
          if (m == W4_PRE) {
            vm_wait();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
          }
          acc[I0_ + f][2 * g + H_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[f][s], FB[g][s], acc[I0_ + f][2 * g + H_], 0, 0, 0);
          if (m >= W4_PRE) hook(m - W4_PRE);
          __builtin_amdgcn_sched_barrier(0);
        }
  };

  auto k_step = [&](auto stage_c, auto steady_c, int t) __attribute__((always_inline)) {
    constexpr int ST = decltype(stage_c)::value;
    constexpr int SB = ST * W4_STAGE, NB = (ST ^ 1) * W4_STAGE;
    constexpr bool SD = decltype(steady_c)::value;   // steady: tiles t+1 and t+2 exist, are full, and share the K batch
    const bool more1 = SD || t + 1 < nt, more2 = SD || t + 2 < nt;
    auto dma_slot = [](int k) { return k == 3 || k == 6 || k == 9 || k == 12; };   // the phase's four DMA pieces
    auto dma_idx = [](int k) { return (k - 3) / 3; };
    // ---- P1: A-top x B-left; reads B-right(t); DMA B-right(t+1)
    section(at, 0, bl[ST], 0,
            [&]() __attribute__((always_inline)) { if (more1) W4_VMWAIT(12); else W4_VMWAIT(4); },   // B-right(t) landed; younger: A-bot(t) [, A-top / B-left(t+1)]
            [&](int k) __attribute__((always_inline)) {
      if (k < 8) br[k >> 2][k & 3] = rd_b(SB + 3 * W4_HB, k >> 2, k & 3);
      if constexpr (SD) { if (dma_slot(k)) issue_one(I3{}, ST ^ 1, dma_idx(k)); }
      else { if (k == 8 && more1) issue_gen(I3{}, ST ^ 1); }
    });
    // ---- P2: A-top x B-right; reads A-bot(t); DMA A-bot(t+1)
    section(at, 0, br, 1,
            [&]() __attribute__((always_inline)) { if (more1) W4_VMWAIT(12); else W4_VMWAIT(0); },   // A-bot(t) landed; younger: A-top / B-left / B-right(t+1)
            [&](int k) __attribute__((always_inline)) {
      if (k < 8) ab[k >> 2][k & 3] = rd_a(SB + W4_HB, k >> 2, k & 3);
      if constexpr (SD) { if (dma_slot(k)) issue_one(I1{}, ST ^ 1, dma_idx(k)); }
      else { if (k == 8 && more1) issue_gen(I1{}, ST ^ 1); }
    });
    // ---- P3: A-bot x B-right; no reads; DMA A-top(t+2)
    section(ab, 2, br, 1, [&]() __attribute__((always_inline)) {},
            [&](int k) __attribute__((always_inline)) {
      if constexpr (SD) { if (dma_slot(k)) issue_one(I0{}, ST, dma_idx(k)); }
      else { if (k == 0 && more2) issue_gen(I0{}, ST); }
    });
    // ---- P4: A-bot x B-left; reads A-top(t+1), B-left(t+1) (16 reads in 14 slots); DMA B-left(t+2)
    section(ab, 2, bl[ST], 0,
            [&]() __attribute__((always_inline)) { if (more1) { if (more2) W4_VMWAIT(12); else W4_VMWAIT(8); } },   // A-top / B-left(t+1) landed; younger: B-right / A-bot(t+1) [, A-top(t+2)]
            [&](int k) __attribute__((always_inline)) {
      auto rd = [&](int i) __attribute__((always_inline)) {
        if (i < 8) at[i >> 2][i & 3] = rd_a(NB, i >> 2, i & 3);
        else bl[ST ^ 1][(i - 8) >> 2][i & 3] = rd_b(NB + 2 * W4_HB, (i - 8) >> 2, i & 3);
      };
      if (more1) {
        if (k < 2) { rd(2 * k); rd(2 * k + 1); } else rd(k + 2);
      }
      if constexpr (SD) { if (dma_slot(k)) issue_one(I2{}, ST, dma_idx(k)); }
      else { if (k == 13 && more2) issue_gen(I2{}, ST); }
    });
  };

  if (nt > 0) {
    issue_gen(I0{}, 0); issue_gen(I2{}, 0); issue_gen(I3{}, 0); issue_gen(I1{}, 0);
    if (nt > 1) {
      issue_gen(I0{}, 1); issue_gen(I2{}, 1);
      W4_VMWAIT(16);  // A-top(0), B-left(0) landed
    } else {
      W4_VMWAIT(8);
    }
    __builtin_amdgcn_s_barrier();
    // the "P4(-1)" reads: fragments of tile 0's first phase
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int s = 0; s < 4; ++s) { at[f][s] = rd_a(0, f, s); bl[0][f][s] = rd_b(2 * W4_HB, f, s); }
    // steady: tiles t+1 and t+2 exist, are full, and share a tail-free K range (K batches without a tail are steady too: the
    // in-section issue carries the cursor wrap)
    const int has_tail = (kv_last < 64 && t1 == P.KB * kt_per) ? 1 : 0;
    const int n_steady = TAILS ? (max(0, nt - 2) & ~1)
                               : ((kv_last == 64 || P.KB == 1) ? max(0, nt - 2 - has_tail) & ~1 : 0);
    int t = 0;
    for (; t < n_steady; t += 2) {
      k_step(I0{}, std::true_type{}, t);
      k_step(I1{}, std::true_type{}, t + 1);
    }
    for (; t < nt; t += 2) {
      k_step(I0{}, std::false_type{}, t);
      if (t + 1 < nt) k_step(I1{}, std::false_type{}, t + 1);
    }
  }
#undef W4_VMWAIT
  __syncthreads();

  // ---- in-kernel split-K fix-up (grouped launch, p.fix_epoch != 0): arrival order decides who finishes the tile.  A workgroup
  // that is NOT the last to arrive writes its slab as ever and raises its flag; the LAST one waits for the others' flags --
  // they all arrived before it, i.e. are past their K loops and wait for nobody: no cycle, whatever the grid and whoever is
  // resident -- and adds their slabs to its own partial sum IN SPLIT ORDER (bit-identical to the reduction kernel, and the same
  // from run to run whoever arrives last), scales, accumulates into C and stores: one slab write + read per tile less, and no
  // reduction launch.
  bool fix_last = false;
  if constexpr (GRP && EP == 1) {
    if (p.fix_epoch != 0 && !(GEMM_SK && p.sk_ks > 0)) {
      unsigned* xch = reinterpret_cast<unsigned*>(smem);
      if (threadIdx.x == 0) xch[0] = __hip_atomic_fetch_add(p.fix_cnt + fix_uid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      fix_last = xch[0] == (unsigned)(P.split_k - 1);
      __syncthreads();   // (the staging below reuses smem)
      if (fix_last) {
        if (threadIdx.x == 0) {
          __hip_atomic_store(p.fix_cnt + fix_uid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every split has arrived: ready for the next launch
          for (int s_ = 0; s_ < P.split_k; ++s_) {
            if (s_ == split) continue;
            while (__hip_atomic_load(p.fix_flag + fix_uid * 8 + s_, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != p.fix_epoch) __builtin_amdgcn_s_sleep(2);
          }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // every thread: the slabs are read from L2, not from a stale L1 line
      }
    }
  }

  // ---- epilogue: per-wave LDS staging of a 32 x 128 row block -> 16-byte row vectors --------------------------------
  const int mw = m0 + wm * 128, nw = n0 + wn2 * 128;
  if constexpr (EP == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int nn = nw + c * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int mm = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (mm < P.M && nn < P.N) gemm_store(P, z, split, mm, nn, acc[i][c][r]);
        }
      }
  } else {
    constexpr int EP_LD = 128 + 4;
    float* ep = reinterpret_cast<float*>(smem) + wave * (32 * EP_LD);   // 4 x 16.5 KiB = 66 KiB
    const float4* tab = nullptr;
    if constexpr (EP == 3) {
      if (P.gtab) {
        float4* tl = reinterpret_cast<float4*>(smem + 98304);
        gelu_tab_stage(P.gtab, tl);
        __syncthreads();
        tab = tl;
      }
    }
    auto stage_block = [&](const f32x16_t (&a)[4]) __attribute__((always_inline)) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          ep[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * EP_LD + c * 32 + (lane & 31)] = a[c][r];
    };
    constexpr bool CSUM = (EP == 2 || EP == 4) && !GRP;
    const bool csum = CSUM && P.colsum_part != nullptr;
    float cs[2] = {0.f, 0.f};
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
      switch (i) {
        case 0: stage_block(acc[0]); break;
        case 1: stage_block(acc[1]); break;
        case 2: stage_block(acc[2]); break;
        default: stage_block(acc[3]); break;
      }
#pragma unroll
      for (int qq = 0; qq < 8; ++qq) {
        const int id = lane + 64 * qq;
        const int rl = id >> 4, ch = id & 15;
        const int mm = mw + i * 32 + rl;
        const int nn = nw + ch * 8;
        float vo[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (mm < P.M && nn < P.N) {
          const float4 lo = *reinterpret_cast<const float4*>(ep + rl * EP_LD + ch * 8);
          const float4 hi = *reinterpret_cast<const float4*>(ep + rl * EP_LD + ch * 8 + 4);
          float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
          if constexpr (EP == 1) {
            if (GRP && fix_last) {
              const long mn = (long)P.M * P.N, at = (long)mm * P.N + nn;
              float tot[8];
#pragma unroll 1
              for (int s_ = 0; s_ < P.split_k; ++s_) {
                float src[8];
                if (s_ == split) {
#pragma unroll
                  for (int e = 0; e < 8; ++e) src[e] = v[e];
                } else {
                  const float4 a = *reinterpret_cast<const float4*>(P.ws + (long)s_ * mn + at);
                  const float4 b = *reinterpret_cast<const float4*>(P.ws + (long)s_ * mn + at + 4);
                  src[0] = a.x; src[1] = a.y; src[2] = a.z; src[3] = a.w; src[4] = b.x; src[5] = b.y; src[6] = b.z; src[7] = b.w;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) tot[e] = s_ == 0 ? src[e] : tot[e] + src[e];
              }
#pragma unroll
              for (int e = 0; e < 8; ++e) tot[e] *= P.alpha;
              const long off = (long)mm * P.ldc + nn;
              if (P.accumulate) {
                float c[8];
                ld8_dt(P.C, off, P.c_dtype, c);
#pragma unroll
                for (int e = 0; e < 8; ++e) tot[e] += c[e];
              }
              st8_dt(P.C, off, P.c_dtype, tot);
            } else gemm_store8(P, zo, zi, z, split, mm, nn, v);
          }
          else gemm_store8_fast<(EP == 2 ? 0 : EP)>(P, zo, zi, mm, nn, v, tab, CSUM ? vo : nullptr);
        }
        if constexpr (CSUM) {
          if (csum) {
            *reinterpret_cast<float4*>(ep + rl * EP_LD + ch * 8) = make_float4(vo[0], vo[1], vo[2], vo[3]);
            *reinterpret_cast<float4*>(ep + rl * EP_LD + ch * 8 + 4) = make_float4(vo[4], vo[5], vo[6], vo[7]);
          }
        }
      }
      if constexpr (CSUM) {
        if (csum) gemm_colsum_block<128, EP_LD>(ep, lane, cs);
      }
    }
    if constexpr (CSUM) {
      if (csum) {  // the two waves that share a column range (wm = 0, 1) meet in LDS; wm = 0 writes the tile's partial row
        float* xch = reinterpret_cast<float*>(smem + 98304);
        xch[wave * 128 + lane] = cs[0];
        xch[wave * 128 + 64 + lane] = cs[1];
        __syncthreads();
        if (wm == 0) {
          const float* other = xch + (wave + 2) * 128;
          float* dst = P.colsum_part + (long)tm * P.N;
          if (nw + lane < P.N) dst[nw + lane] = cs[0] + other[lane];
          if (nw + 64 + lane < P.N) dst[nw + 64 + lane] = cs[1] + other[64 + lane];
        }
      }
    }
  }
  if constexpr (GRP && EP == 1) {
    if (p.fix_epoch != 0 && !(GEMM_SK && p.sk_ks > 0) && !fix_last) {
      __threadfence();   // this thread's slab stores are visible device-wide ...
      __syncthreads();   // ... for every thread of the workgroup, before the flag goes up
      if (threadIdx.x == 0)
        __hip_atomic_store(p.fix_flag + fix_uid * 8 + split, p.fix_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  }
}

template <bool TA, bool TB>
static int w4_launch_t(GemmP& p, int nbatch, int ep, hipStream_t st) {
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  p.nbatch = nbatch;
  p.vtotal = p.tiles_m * p.tiles_n * nbatch * p.split_k;
  p.patch_m = 0;
  if (p.split_k > 1 && nbatch == 1) {
    p.patch_m = 16 / p.tiles_n; if (p.patch_m < 1) p.patch_m = 1; if (p.patch_m > p.tiles_m) p.patch_m = p.tiles_m;
  }
  p.skew = 0;
  const int pgrid = 256 - g_pp_reserved_cus;
  dim3 grid((unsigned)(p.vtotal < pgrid || p.vtotal > 2048 ? p.vtotal : pgrid), 1, 1);
  constexpr int smem = 2 * W4_STAGE;
  static bool done[5] = {false, false, false, false, false};
#define W4_CASE(E) case E: { \
    if (!done[E]) { \
      if (hipFuncSetAttribute((const void*)gemm_w4_kernel<TA, TB, E>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return WL_ELAUNCH; \
      done[E] = true; \
    } \
    WL_LAUNCH((gemm_w4_kernel<TA, TB, E>), grid, dim3(256), smem, st, p); } break;
  if constexpr (TA && TB) {
    if (ep == 1 && p.KB > 1 && (p.K & 63) != 0) {   // K batches with a K tail each (conv stack weight gradients)
      static bool done_t = false;
      if (!done_t) {
        if (hipFuncSetAttribute((const void*)gemm_w4_kernel<true, true, 1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return WL_ELAUNCH;
        done_t = true;
      }
      WL_LAUNCH((gemm_w4_kernel<true, true, 1, false, true>), grid, dim3(256), smem, st, p);
      return wl_check_launch();
    }
  }
  switch (ep) { W4_CASE(0) W4_CASE(1) W4_CASE(2) W4_CASE(3) default: W4_CASE(4) }
#undef W4_CASE
  return wl_check_launch();
}

// in-kernel split-K fix-up: arrival counters and flags of up to 4096 (member, tile) ids x 8 splits; zero at load, counters are
// returned to zero by the last arriver, flags carry the launch's epoch (never 0).  One grouped launch at a time per process
// (the path has one compute stream).
__device__ unsigned g_w4_fix_cnt[4096];
__device__ unsigned g_w4_fix_flag[4096 * 8];
static unsigned g_w4_fix_epoch = 0;
bool gemm_w4_fixup_ok(const GemmP& p) {
  // OFF by default: built, bit-identical to the reduction launch and deterministic (tools/wgrad_fixup_cmp.py), and measured
  // SLOWER -- 402 against 338 us per Base block launch + reduction, the step +1.0 ms (profiles/r06/ab_wgrad_fixup.txt): the
  // tiles' last arrivers all finish at the end of the launch and each reads its 256 KB of slabs through the epilogue's
  // dependent 32-byte loads, alone on its CU, where the reduction kernel streams the same bytes with every CU in 17 us.
  static const bool on = [] { const char* e = getenv("WAVLM_WGRAD_FIXUP"); return e && *e == '1'; }();
  return on && p.split_k >= 2 && p.split_k <= 8 && p.vtotal <= 4096 && !(GEMM_SK && p.sk_ks > 0);
}
int gemm_w4_launch_grouped(GemmP& p, hipStream_t st) {
  constexpr int smem = 2 * W4_STAGE;
  if (p.fix_epoch != 0) {   // the caller asked for the fix-up (gemm_w4_fixup_ok): bind the buffers, draw the epoch
    static unsigned* cnt = nullptr; static unsigned* flag = nullptr;
    if (!cnt) {
      if (hipGetSymbolAddress((void**)&cnt, HIP_SYMBOL(g_w4_fix_cnt)) != hipSuccess || hipGetSymbolAddress((void**)&flag, HIP_SYMBOL(g_w4_fix_flag)) != hipSuccess) return WL_ELAUNCH;
    }
    if (++g_w4_fix_epoch == 0) g_w4_fix_epoch = 1;
    p.fix_epoch = g_w4_fix_epoch; p.fix_cnt = cnt; p.fix_flag = flag;
  }
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute((const void*)gemm_w4_kernel<true, true, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return WL_ELAUNCH;
    done = true;
  }
  p.patch_m = 0; p.skew = 0; p.nbatch = 1;
  const int pgrid = 256 - g_pp_reserved_cus;
  dim3 grid((unsigned)((GEMM_SK && p.sk_ks > 0) ? p.sk_wgs : p.vtotal < pgrid ? p.vtotal : pgrid), 1, 1);
  WL_LAUNCH((gemm_w4_kernel<true, true, 1, true>), grid, dim3(256), smem, st, p);
  return wl_check_launch();
}

int gemm_w4_launch(GemmP& p, int nbatch, bool transA, bool transB, int ep, hipStream_t st) {
  if (!transA && !transB) return w4_launch_t<false, false>(p, nbatch, ep, st);
  if (!transA && transB) return w4_launch_t<false, true>(p, nbatch, ep, st);
  if (transA && !transB) return w4_launch_t<true, false>(p, nbatch, ep, st);
  return w4_launch_t<true, true>(p, nbatch, ep, st);
}
