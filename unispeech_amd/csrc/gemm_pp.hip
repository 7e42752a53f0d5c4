// bf16 MFMA GEMM for gfx950, large-problem path: 256 x 256 x 64 block tile, 8 waves, "ping-pong" schedule.
//
// Why a second kernel: at 128 x 128 the operand traffic L2 -> LDS (32 KiB per 2.1 MFLOP) saturates the per-CU load
// path near ~40 % of the MFMA peak; 256 x 256 halves the bytes per flop, and each wave's 128 x 64 accumulator
// tile needs 0.75 ds_read_b128 per MFMA instead of 1.0.  What makes the big tile pay is the schedule:
//
//   * Operands move HBM/L2 -> LDS only by LDS-DMA (global_load_lds_dwordx4), never through VGPRs, for BOTH operand
//     layouts.  K-contiguous operands ([rows][K]) land as [row][64 k] (128-B rows, 16-B chunks XOR-swizzled by
//     (row >> 1) & 7, fragments by ds_read_b128).  K-strided operands ([K][rows]: weight-gradient and
//     activation-gradient forms) land untransposed as [k][128 rows] (256-B rows, 32-B granules XOR-swizzled by
//     2 * (k & 3)) and are read with the transposing ds_read_b64_tr_b16, which hands each lane the four k of its
//     row -- the transposition costs no instruction.  The swizzles are applied on the SOURCE address (the DMA writes
//     lane l at base + 16 l).
//   * A block tile is staged as four half-tiles per K step: A-top / A-bottom (each wave-row's first / second 64
//     rows), B-left / B-right (each wave-column's first / second 32 columns), two stages = 128 KiB.  A K step is
//     four phases, one 64 x 32 accumulator quadrant each (8 MFMA 32x32x16):
//         P1 reads A-top + B-left, P2 reads B-right, P3 reads A-bottom, P4 reads nothing
//     so a half-tile buffer is free again after its single read phase and is refilled 5 phases before its next
//     use; every phase issues one half-tile of DMA (2 instructions per lane).  s_waitcnt vmcnt is counted (8 = four
//     half-tiles stay in flight) and sits one phase before the first read of the half-tile it covers (P4: A-top /
//     B-left of t+1, P1: B-right, P2: A-bottom), so every request has a full K step to land.
//   * Waves 0-3 (rows 0-127) and waves 4-7 (rows 128-255) run the same phase sequence one barrier apart: while one
//     group issues its ds_reads, the other owns the MFMA pipe (s_setprio 1).  Each SIMD hosts one wave of each
//     group, so its MFMA pipe always has a wave in an MFMA section.
//   * Steady state (all K steps but the last two, no K tail in flight): the two DMA instructions of a phase are
//     issued by each wave INSIDE its own MFMA section (after MFMA pairs 1 and 3) and carry no tail select.  Measured
//     with the probe builds (tools/probe): the K step costs 2160 cycles without DMA and cost 2900-3500 with the DMA
//     in the read interval -- the vector-memory path is ~55 % busy (64 KiB per K step against 64 B/clk), so a wave
//     that issues DMA behind its ds_reads blocks on a full queue for longer than the other group's MFMA window,
//     and every barrier interval then takes max(read wave, MFMA wave).  In-section issue: ~2500 cycles, at which
//     point the kernel is power-limited (shader clock 1.5-1.7 GHz under load; register-only MFMA peaks at 1.95
//     PFLOP/s with random operands, 2.45 with zeros).
//
// Hazards (LDS-DMA is ordered for a ds_read only by the issuing wave's vmcnt followed by a barrier):
//   RAW  a wait in phase q covers what phase q+1 reads; group 0 waits before barrier b, group 1 before b+1, the
//        read (group 0, phase q+1) follows barrier b+1.
//   WAR  buffer X is read in phase q by group 0 (interval J) and group 1 (J+1, complete before barrier J+2 via the
//        lgkmcnt(0) that precedes its MFMA section); it is refilled from phase q+2 (group 0: interval J+4).
// K tails and M/N edges stay on the DMA path: rows past the edge are clamped (they feed outputs never stored), K
// positions past the end are fetched from a zero page.
#include "gemm_common.hpp"

#include "tile_loaders.hpp"
#ifndef PP_BIAS_LDS
#define PP_BIAS_LDS 1
#endif
#ifndef PP_AUX_PRELOAD
#define PP_AUX_PRELOAD 1
#endif

// Lab-bench switches exist only in builds made with -DWAVLM_EXPERIMENTAL (tools/probe/build_probe.py).
#if !defined(WAVLM_EXPERIMENTAL)
#undef PP_PROBE
#undef PP_DMA_RD
#endif
#ifndef PP_PROBE
// 0 in the product build.  Timing probes (tools/probe/build_probe.py; results are wrong by construction): bit 0 drops the
// LDS fragment reads, bit 1 the DMA, bit 2 three quarters of the MFMAs, bit 3 makes the DMA source hot and contiguous,
// bit 4 reports clocks into C, bit 5 drops the steady-state DMA waits, bit 6 stamps the phases of one K step.
#define PP_PROBE 0
#endif
// PP_DMA_RD: steady-state DMA issue in the idle read intervals (1: P2 and P4 read intervals, 2: A-bot stays in P2's MFMA
// section) instead of inside the MFMA sections (0)
#ifndef PP_DMA_RD
#define PP_DMA_RD 1
#endif
#define PP_HB 16384     // bytes per half-tile buffer
#define PP_STAGE 65536  // bytes per stage: [A-top][A-bot][B-left][B-right]

__device__ __attribute__((aligned(256))) unsigned char g_pp_zero[256];

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((address_space(3))) bf16x4_t* lds_b4_ptr;

struct PPCursor {  // position of one half-tile stream in the flattened (K-batch, K-tile) sequence; all uniform
  long off;        // element offset of the current K tile from the operand tile base
  int kt;          // K-tile index inside the current K batch
};

// TAILS: the launch has several K batches that each end in a K tail (K % 64 != 0, KB > 1: the conv stack's weight
// gradients, 32 batches of 375 K tiles).  Its steady-state DMA issue then carries the zero-page form of a tail piece behind a
// uniform branch, so the whole K range runs the steady schedule; without TAILS such launches fall back to the general form
// for every step (the branch costs the tail-free launches ~2 % when it is compiled in, hence a separate instantiation).
template <bool TA, bool TB, int EP, bool GRP = false, bool TAILS = false>
__global__ __launch_bounds__(512) void gemm_pp_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 * PP_STAGE

  // Persistent launch: gridDim.x blocks (one per CU) walk the virtual block ids vid = block, block + grid, ...;
  // vid -> (tile, batch z, split).  The epilogue stores of one tile are in flight while the next tile's DMA starts.
  const int ntile = p.tiles_m * p.tiles_n;
  if (p.skew && blockIdx.x < (unsigned)p.vtotal) {
    // de-phase the CUs: without it every CU reaches its HBM-bound epilogue at the same time in every round
    for (int i = (int)(blockIdx.x & 3) * p.skew; i > 0; --i) __builtin_amdgcn_s_sleep(127);
  }
#if PP_PROBE & 16
  const unsigned long long pc0 = __builtin_readcyclecounter(), pr0 = wall_clock64();
  unsigned long long pck = 0, prk = 0;
#endif
  for (int vid = blockIdx.x; vid < p.vtotal; vid += gridDim.x) {
  int tm, tn, z, split;
  GemmP q = p;  // grouped launch: the work item's own problem; otherwise an alias of p
  if constexpr (GRP) {
    // Grouped split-K launch (the weight gradients of one encoder layer in ONE launch: 108 tiles x split 2 instead of
    // four launches with splits of 7-28 -- a quarter of the slab traffic and 3-4x longer K loops per tile).  Work items
    // are ordered (problem, split, tile); each XCD takes a contiguous run, i.e. tiles of one split of one problem.
    const int nv = p.vtotal;
    const int q8 = nv >> 3, rem = nv & 7, xcd = vid & 7, idx = vid >> 3;
    const int item = (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + idx;
    const int g = (item >= p.grp[1].vbase) + (item >= p.grp[2].vbase) + (item >= p.grp[3].vbase);
#define PP_SEL(F) (g == 0 ? p.grp[0].F : g == 1 ? p.grp[1].F : g == 2 ? p.grp[2].F : p.grp[3].F)
    q.A = PP_SEL(A); q.B = PP_SEL(B); q.ws = PP_SEL(ws); q.lda = PP_SEL(lda); q.ldb = PP_SEL(ldb);
    q.M = PP_SEL(M); q.N = PP_SEL(N); q.tiles_m = PP_SEL(tiles_m); q.tiles_n = PP_SEL(tiles_n);
    const int local = item - PP_SEL(vbase);
#undef PP_SEL
    const int nt_g = q.tiles_m * q.tiles_n;
    split = local / nt_g;
    const int tile = local - split * nt_g;
    tm = tile / q.tiles_n; tn = tile - tm * q.tiles_n;
    z = 0;
  } else
  if (p.patch_m == 0) {
    // XCD-aware tile order: the dispatcher places block b on XCD b % 8 (private L2 per XCD); each XCD walks a contiguous
    // run of tiles (neighbours share the A row panel and all of B).  Batch and split are the outer dimensions.
    const int nt = ntile, bid = vid % ntile;
    const int q = nt >> 3, rem = nt & 7, xcd = bid & 7, idx = bid >> 3;
    const int tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
    gemm_tile_rc(tile, p.tiles_m, p.tiles_n, p.swz_r, tm, tn);
    z = (vid / ntile) % p.nbatch; split = vid / (ntile * p.nbatch);
  } else {
    // Split-K (weight-gradient) order.  Blocks that share operand panels are the tiles of ONE split: they must sit on
    // one XCD at the same time or every XCD re-reads the panels from HBM (measured 630 MB per fc1 dW launch against
    // 184 MB algorithmic -- the launch was HBM-bound).  Work items are ordered (patch of patch_m tile rows x all tile
    // columns, split, tile in patch) and each XCD takes a contiguous run of that order.
    const int nv = p.vtotal;
    const int q = nv >> 3, rem = nv & 7, xcd = vid & 7, idx = vid >> 3;
    const int item = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
    const int full = p.patch_m * p.tiles_n * p.split_k;           // items of a full patch
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
  const int wm = wave >> 2, wn = wave & 3;

  // The DMA asm below takes its base address in an SGPR pair: pin the (wave-uniform) panel bases and the K range to
  // scalar registers here, once per tile -- in some instantiations the work-item decode above runs on the VALU and the
  // compiler would otherwise hand the asm a VGPR pair (an assembler error, not a slow path).
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
  gemm_split_range(P.KB * kt_per, P.split_k, split, t0, t1);
  t0 = __builtin_amdgcn_readfirstlane(t0); t1 = __builtin_amdgcn_readfirstlane(t1);
  const int nt = t1 - t0;

  // ---- DMA side: per-lane byte offsets of the two pieces (j) this lane fetches of each half-tile ------------------
  // piece (wave, j) of a half-tile buffer: K-contiguous -> buffer rows 16 w + 8 j .. +8 (lane: row + (l >> 3), physical
  // chunk l & 7); K-strided -> k rows 8 w + 4 j .. +4 (lane: k row + (l >> 4), physical 16-B chunk l & 15)
  unsigned voff[4][2];
  int kidx[2][2];  // [operand][j]: first k position this lane's 16 bytes cover (K-contiguous) / its k row (K-strided)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int op = 0; op < 2; ++op) {
      const bool TR = op ? TB : TA;
      const long ld = op ? P.ldb : P.lda;
      const int rows_valid = op ? P.N - n0 : P.M - m0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        unsigned v;
        if (!TR) {
          const int R = wave * 16 + j * 8 + (lane >> 3);
          const int c = (lane & 7) ^ ((R >> 1) & 7);
          int r = op ? ((R >> 5) * 64 + h * 32 + (R & 31)) : ((R >> 6) * 128 + h * 64 + (R & 63));
          if (r >= rows_valid) r = rows_valid - 1;
          v = (unsigned)(((long)r * ld + c * 8) * 2);
          kidx[op][j] = c * 8;
        } else {
          const int kr = wave * 8 + j * 4 + (lane >> 4);
          const int pc = lane & 15;
          const int R = (((pc >> 1) ^ (2 * (kr & 3))) << 4) + ((pc & 1) << 3);  // buffer row of the first of 8
          int r = op ? ((R >> 5) * 64 + h * 32 + (R & 31)) : ((R >> 6) * 128 + h * 64 + (R & 63));
          if (r + 8 > rows_valid) r = rows_valid - 8;
          v = (unsigned)(((long)kr * ld + r) * 2);
          kidx[op][j] = kr;
        }
        voff[op * 2 + h][j] = v;
      }
    }
  }
  const long stepA = TA ? 64 * P.lda : 64, stepB = TB ? 64 * P.ldb : 64;
  const long jumpA = P.sA_kb - (long)kt_per * stepA, jumpB = P.sB_kb - (long)kt_per * stepB;
  PPCursor cur[4];
  {
    const int kb0 = t0 / kt_per, kt0 = t0 - kb0 * kt_per;
    cur[0].off = cur[1].off = (long)kb0 * P.sA_kb + (long)kt0 * stepA;
    cur[2].off = cur[3].off = (long)kb0 * P.sB_kb + (long)kt0 * stepB;
    cur[0].kt = cur[1].kt = cur[2].kt = cur[3].kt = kt0;
  }
  // issue half-tile `which` (0 A-top, 1 A-bot, 2 B-left, 3 B-right) of the stream's current K tile into `stage`,
  // then advance the stream by one K tile
  // LDS-DMA of one 1 KiB piece: M0 = LDS destination, address = SGPR base + 32-bit VGPR offset.  EVERY DMA of this kernel
  // goes through here, so the compiler never tracks M0 itself (it would not see these writes).
  auto dma16 = [&](const char* sbase, unsigned voff32, unsigned char* ldst) __attribute__((always_inline)) {
    const unsigned lds_dst = (unsigned)(unsigned long)(las_ptr)ldst;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(lds_dst), "v"(voff32), "s"(sbase) : "memory", "m0");  // m0: so that the K-tail's builtin DMA re-initialises it
  };
  // the same with a full 64-bit per-lane address (K-tail pieces: lanes past the end of K fetch the zero page).  asm, not the
  // builtin: with a builtin LDS-DMA anywhere in the K loop the compiler drains vmcnt(0) before every later LDS read
  auto dma16_v = [&](const char* lane_ptr, unsigned char* ldst) __attribute__((always_inline)) {
    const unsigned lds_dst = (unsigned)(unsigned long)(las_ptr)ldst;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                 :: "s"(lds_dst), "v"(lane_ptr) : "memory", "m0");
  };
  auto issue_gen = [&](auto which_c, int stage) __attribute__((always_inline)) {
    constexpr int W = decltype(which_c)::value;
    constexpr int OP = W >> 1;
    PPCursor& c = cur[W];
    const char* src = (OP ? Bb : Ab) + c.off * 2;
    const int kv = (c.kt == kt_per - 1) ? kv_last : 64;
    unsigned char* dst = smem + stage * PP_STAGE + W * PP_HB + wave * 2048;
#if PP_PROBE & 2
    if (false) {
#else
    if (kv >= 64) {
#endif
#pragma unroll
      for (int j = 0; j < 2; ++j) dma16(src, voff[W][j], dst + j * 1024);
    } else if (!(PP_PROBE & 2)) {
      // K tail: lanes whose k position lies past the end read the zero page instead (per-lane choice of the OFFSET from
      // the uniform base: the zero page's distance from `src` does not fit 32 bits in general, so those lanes use the
      // zero page as base through a second, fully predicated instruction)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bool in = kidx[OP][j] < kv;
        const char* s = in ? src + voff[W][j] : (const char*)g_pp_zero;
        __builtin_amdgcn_global_load_lds((gas_ptr)s, (las_ptr)(dst + j * 1024), 16, 0, 0);
      }
    }
    c.off += OP ? stepB : stepA;
    if (++c.kt == kt_per) { c.kt = 0; c.off += OP ? jumpB : jumpA; }
  };
  // Steady-state form: one DMA instruction of a full tile (no K-tail select).  In the general form the ~40 scalar /
  // branch instructions per half-tile run in the phase in which the wave must finish inside the other group's 8-MFMA
  // window (256 cycles at one instruction per 4 cycles per wave) -- the K loop cannot afford them; here the
  // instruction and its scalar cursor work sit between the wave's own MFMAs.
  auto issue_one = [&](auto which_c, int stage, int j) __attribute__((always_inline)) {
    constexpr int W = decltype(which_c)::value;
    constexpr int OP = W >> 1;
    PPCursor& c = cur[W];
    const char* src = (OP ? Bb : Ab) + c.off * 2;
    unsigned char* dst = smem + stage * PP_STAGE + W * PP_HB + wave * 2048;
#if PP_PROBE & 8
    src = Ab + W * 16384 + wave * 2048 + j * 1024 - voff[W][j] + lane * 16;  // hot, contiguous window
#endif
#if !(PP_PROBE & 2)
    // the per-lane offset stays a 32-bit register and the address is formed as "SGPR base + VGPR offset" by the
    // instruction itself (global_load_lds ... v, s[base]): without the barrier the zero-extension is hoisted out of
    // the loop as eight 64-bit register pairs and every DMA costs a v_lshl_add_u64 between two MFMAs
    // (written as asm: through the builtin the zero-extension of the offset is hoisted out of the loop as eight 64-bit
    // register pairs and every DMA costs a v_lshl_add_u64 -- or, with the offset hidden from the optimiser, a v_mov_b32
    // -- between two MFMAs.  M0 = LDS destination of the wave's 1 KiB piece; one wait state after writing it.)
    // TAILS instantiation: a K-tail tile takes the zero-page form of this one piece (asm, 64-bit per-lane address).
    if (TAILS && c.kt == kt_per - 1) {
      const bool in = kidx[OP][j] < kv_last;
      const char* sp = in ? src + voff[W][j] : (const char*)g_pp_zero;
      dma16_v(sp, dst + j * 1024);
    } else {
      dma16(src, voff[W][j], dst + j * 1024);
    }
#endif
    if (j == 1) {  // scalar cursor work hides behind the MFMAs around it
      c.off += OP ? stepB : stepA;
      if (++c.kt == kt_per) { c.kt = 0; c.off += OP ? jumpB : jumpA; }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;

  // ---- fragment side ---------------------------------------------------------------------------------------------
  // K-contiguous image: lane -> row (l & 31), chunk 2 s + (l >> 5), swizzle ((l >> 1) & 7)
  // Two copies of every per-lane fragment address, one per LDS stage: with ONE copy the stage-1 reads need
  // "register + 65536 + ..." and the 16-bit offset field of ds_read cannot hold that, so the compiler spent a v_add_u32 on
  // every fragment read (48 per two K steps, all in the read interval that decides when the wave reaches its barrier).
  // The asm barrier keeps the second copy a register instead of a rematerialised add.
  unsigned kc_a[2][4], kc_b[2][4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const unsigned lo = (unsigned)((lane & 31) * 128 + (((2 * s + (lane >> 5)) ^ ((lane >> 1) & 7)) << 4));
    kc_a[0][s] = lo + wm * 8192;
    kc_b[0][s] = lo + wn * 4096;
    kc_a[1][s] = kc_a[0][s] + PP_STAGE;
    kc_b[1][s] = kc_b[0][s] + PP_STAGE;
    if constexpr (!TA) asm volatile("" : "+v"(kc_a[1][s]));
    if constexpr (!TB) asm volatile("" : "+v"(kc_b[1][s]));
  }
  // K-strided image: 16-lane group g reads the [4 k][16 rows] block of k octet (l >> 5), rows 16 (g & 1) + ..;
  // lane i of the group supplies the address of k row (i >> 2), rows 4 (i & 3) .. +4
  unsigned tr_a[2][2], tr_b[2];
  {
    const int i = lane & 15, g1 = (lane >> 4) & 1;
    const unsigned kpart = (unsigned)((8 * (lane >> 5) + (i >> 2)) * 256 + (i & 3) * 8);
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      tr_a[0][f] = kpart + (unsigned)(((wm * 4 + f * 2 + g1) ^ (2 * (i >> 2))) << 5);
      tr_a[1][f] = tr_a[0][f] + PP_STAGE;
      if constexpr (TA) asm volatile("" : "+v"(tr_a[1][f]));
    }
    tr_b[0] = kpart + (unsigned)(((wn * 2 + g1) ^ (2 * (i >> 2))) << 5);
    tr_b[1] = tr_b[0] + PP_STAGE;
    if constexpr (TB) asm volatile("" : "+v"(tr_b[1]));
  }
  auto rd_a = [&](int base, int f, int s) __attribute__((always_inline)) -> bf16x8_t {  // base: byte offset of the half-tile buffer
#if PP_PROBE & 1
    bf16x8_t z; for (int i = 0; i < 8; ++i) z[i] = (__bf16)(float)(base + f + s); return z;
#endif
    const int st = base >= PP_STAGE ? 1 : 0, ib = base - st * PP_STAGE;  // `base` is a constant at every call site
    if constexpr (!TA) {
      return *reinterpret_cast<const bf16x8_t*>(smem + kc_a[st][s] + (ib + f * 4096));
    } else {
      const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_ptr)(smem + tr_a[st][f] + (ib + s * 4096)));
      const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_ptr)(smem + tr_a[st][f] + (ib + s * 4096 + 1024)));
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };
  auto rd_b = [&](int base, int s) __attribute__((always_inline)) -> bf16x8_t {
#if PP_PROBE & 1
    bf16x8_t z; for (int i = 0; i < 8; ++i) z[i] = (__bf16)(float)(base + s); return z;
#endif
    const int st = base >= PP_STAGE ? 1 : 0, ib = base - st * PP_STAGE;
    if constexpr (!TB) {
      return *reinterpret_cast<const bf16x8_t*>(smem + kc_b[st][s] + ib);
    } else {
      const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_ptr)(smem + tr_b[st] + (ib + s * 4096)));
      const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_ptr)(smem + tr_b[st] + (ib + s * 4096 + 1024)));
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  bf16x8_t at[2][4], ab[2][4], bl[4], br[4];

#if PP_PROBE & 32
#define PP_VMWAIT(N) if (!SD) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#else
#define PP_VMWAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#endif
#if PP_PROBE & 64  // s_memtime stamps of the last steady stage-0 K step (block 0, waves 0 and 4)
  unsigned long long ts[16];
#define PP_STAMP(K_) if constexpr (SD && ST == 0) asm volatile("s_memtime %0" : "=s"(ts[K_]))
#else
#define PP_STAMP(K_)
#endif
#define PP_MFMA_SECTION(FA, I0_, BF, J_) PP_MFMA_SECTION_H4(FA, I0_, BF, J_, (void)0, (void)0, (void)0, (void)0, (void)0)
#define PP_MFMA_SECTION_H(FA, I0_, BF, J_, H0_, H1_) PP_MFMA_SECTION_H4(FA, I0_, BF, J_, H0_, (void)0, H1_, (void)0, (void)0)
#define PP_MFMA_SECTION_H4(FA, I0_, BF, J_, H0_, H1_, H2_, H3_, TAIL_)                              \
  __builtin_amdgcn_sched_barrier(0);                                                                \
  PP_STAMP(4 * (I0_ + (I0_ ? 1 - J_ : J_)) + 0);                                                     \
  __builtin_amdgcn_s_barrier();                                                                     \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
  PP_STAMP(4 * (I0_ + (I0_ ? 1 - J_ : J_)) + 1);                                                     \
  __builtin_amdgcn_sched_barrier(0);                                                                \
  __builtin_amdgcn_s_setprio(1);                                                                    \
  _Pragma("unroll") for (int s = 0; s < ((PP_PROBE & 4) ? 1 : 4); ++s) {                            \
    acc[I0_][J_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0][s], BF[s], acc[I0_][J_], 0, 0, 0);  \
    acc[I0_ + 1][J_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[1][s], BF[s], acc[I0_ + 1][J_], 0, 0, 0); \
    if (s == 0) { H0_; }                                                                            \
    if (s == 1) { H1_; }                                                                            \
    if (s == 2) { H2_; }                                                                            \
    if (s == 3) { H3_; }                                                                            \
  }                                                                                                 \
  TAIL_;                                                                                            \
  PP_STAMP(4 * (I0_ + (I0_ ? 1 - J_ : J_)) + 2);                                                     \
  __builtin_amdgcn_s_setprio(0);                                                                    \
  __builtin_amdgcn_sched_barrier(0);                                                                \
  __builtin_amdgcn_s_barrier();                                                                     \
  PP_STAMP(4 * (I0_ + (I0_ ? 1 - J_ : J_)) + 3);                                                     \
  __builtin_amdgcn_sched_barrier(0);

  constexpr bool EARLY = (TA && TB) && !(PP_PROBE & 128);
  auto k_step = [&](auto stage_c, auto steady_c, int t, auto first_c) __attribute__((always_inline)) {
    constexpr int ST = decltype(stage_c)::value;
    constexpr int SB = ST * PP_STAGE;
    constexpr bool SD = decltype(steady_c)::value;  // steady: tiles t+1 and t+2 exist, are full, and share the K batch
    constexpr bool FIRST = decltype(first_c)::value;
    const bool more1 = SD || t + 1 < nt, more2 = SD || t + 2 < nt;
    auto issue = [&](auto which_c, int stage) __attribute__((always_inline)) { issue_gen(which_c, stage); };
    // (the grouped weight-gradient launch keeps the in-section issue: read-interval issue measured +6 % there, 340 -> 361 us
    // per layer, against -1 ... -2 % on the conv stack's launches: profiles/r03/ab_pp_dma_rd_step.txt)
    if constexpr (SD && PP_DMA_RD != 0 && !GRP) {
      // Steady state, DMA in the IDLE read intervals: the four phases read 12 / 4 / 8 / 0 fragments, so the waves have
      // nothing to do in the P4 read interval and little in P2's, while a DMA instruction costs its issuer 40-60 cycles
      // wherever it sits -- inside an MFMA section that is 40 cycles of an idle matrix pipe per instruction (K step 2160
      // cycles without DMA, ~2500 with).  Same issue ORDER as the in-section form (B-right, A-bot of t+1; A-top, B-left
      // of t+2), so steady and general steps mix freely; only the waits move.
#pragma unroll
      for (int s = 0; s < 4; ++s) bl[s] = rd_b(SB + 2 * PP_HB, s);
#pragma unroll
      for (int f = (EARLY && !FIRST) ? 1 : 0; f < 2; ++f)
#pragma unroll
        for (int s = 0; s < 4; ++s) at[f][s] = rd_a(SB, f, s);
      PP_VMWAIT(6);  // B-right(t) landed; younger: A-bot(t), A-top / B-left(t+1)
      PP_MFMA_SECTION(at, 0, bl, 0)
#pragma unroll
      for (int s = 0; s < 4; ++s) br[s] = rd_b(SB + 3 * PP_HB, s);
      issue_one(I3{}, ST ^ 1, 0); issue_one(I3{}, ST ^ 1, 1);
#if PP_DMA_RD == 1
      issue_one(I1{}, ST ^ 1, 0); issue_one(I1{}, ST ^ 1, 1);
      PP_VMWAIT(8);  // A-bot(t) landed; younger: A-top / B-left(t+1), B-right / A-bot(t+1)
      PP_MFMA_SECTION(at, 0, br, 1)
#else
      PP_VMWAIT(6);
      PP_MFMA_SECTION_H(at, 0, br, 1, issue_one(I1{}, ST ^ 1, 0), issue_one(I1{}, ST ^ 1, 1))
#endif
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int s = 0; s < 4; ++s) ab[f][s] = rd_a(SB + PP_HB, f, s);
      if constexpr (EARLY) {
        PP_VMWAIT(6);  // A-top(t+1) landed (its first fragment is read inside P4's section); younger: B-left, B-right, A-bot(t+1)
        PP_MFMA_SECTION(ab, 2, br, 1)
        issue_one(I0{}, ST, 0); issue_one(I0{}, ST, 1); issue_one(I2{}, ST, 0); issue_one(I2{}, ST, 1);
        PP_VMWAIT(8);  // B-left(t+1) landed
        constexpr int NB = (ST ^ 1) * PP_STAGE;
#define PP_RD_AT(S_) { at[0][S_] = rd_a(NB, 0, S_); }
        PP_MFMA_SECTION_H4(ab, 2, bl, 0, PP_RD_AT(0), PP_RD_AT(1), PP_RD_AT(2), PP_RD_AT(3), (void)0)
#undef PP_RD_AT
      } else {
        PP_MFMA_SECTION(ab, 2, br, 1)
        issue_one(I0{}, ST, 0); issue_one(I0{}, ST, 1); issue_one(I2{}, ST, 0); issue_one(I2{}, ST, 1);
        PP_VMWAIT(8);  // A-top / B-left(t+1) landed; younger: B-right / A-bot(t+1), A-top / B-left(t+2)
        PP_MFMA_SECTION(ab, 2, bl, 0)
      }
      return;
    }

    if constexpr (SD) {
      // same phases; the half-tile of a phase is issued by each wave inside its own MFMA section (after MFMA pairs 1 and
      // 3), so the read interval is ds_reads only and a full vector-memory queue stalls behind running MFMAs.  The
      // waits come before the phase's own issue: three younger half-tiles (6) instead of four.
#pragma unroll
      for (int s = 0; s < 4; ++s) bl[s] = rd_b(SB + 2 * PP_HB, s);
#pragma unroll
      for (int f = (EARLY && !FIRST) ? 1 : 0; f < 2; ++f)
#pragma unroll
        for (int s = 0; s < 4; ++s) at[f][s] = rd_a(SB, f, s);
      PP_VMWAIT(6);
      PP_MFMA_SECTION_H(at, 0, bl, 0, issue_one(I3{}, ST ^ 1, 0), issue_one(I3{}, ST ^ 1, 1))
#pragma unroll
      for (int s = 0; s < 4; ++s) br[s] = rd_b(SB + 3 * PP_HB, s);
      PP_VMWAIT(6);
      PP_MFMA_SECTION_H(at, 0, br, 1, issue_one(I1{}, ST ^ 1, 0), issue_one(I1{}, ST ^ 1, 1))
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int s = 0; s < 4; ++s) ab[f][s] = rd_a(SB + PP_HB, f, s);
      if constexpr (EARLY) {
        PP_MFMA_SECTION_H4(ab, 2, br, 1, issue_one(I0{}, ST, 0), (void)0, issue_one(I0{}, ST, 1), (void)0, PP_VMWAIT(8))
        PP_VMWAIT(6);
        constexpr int NB = (ST ^ 1) * PP_STAGE;
#define PP_RD_AT(S_) { at[0][S_] = rd_a(NB, 0, S_); }
        PP_MFMA_SECTION_H4(ab, 2, bl, 0, { issue_one(I2{}, ST, 0); PP_RD_AT(0) }, PP_RD_AT(1),
                           { issue_one(I2{}, ST, 1); PP_RD_AT(2) }, PP_RD_AT(3), (void)0)
#undef PP_RD_AT
      } else {
        PP_MFMA_SECTION_H(ab, 2, br, 1, issue_one(I0{}, ST, 0), issue_one(I0{}, ST, 1))
        PP_VMWAIT(6);
        PP_MFMA_SECTION_H(ab, 2, bl, 0, issue_one(I2{}, ST, 0), issue_one(I2{}, ST, 1))
      }
      return;
    }

    // ---- P1: A-top, B-left -> quadrant (0,0)
#pragma unroll
    for (int s = 0; s < 4; ++s) bl[s] = rd_b(SB + 2 * PP_HB, s);
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int s = 0; s < 4; ++s) at[f][s] = rd_a(SB, f, s);
    if (more1) {
      issue(I3{}, ST ^ 1);
      PP_VMWAIT(8);  // B-right(t) landed; younger: A-bot(t), A-top/B-left/B-right(t+1)
    } else {
      PP_VMWAIT(2);  // younger: A-bot(t)
    }
    PP_MFMA_SECTION(at, 0, bl, 0)
    // ---- P2: B-right -> quadrant (0,1)
#pragma unroll
    for (int s = 0; s < 4; ++s) br[s] = rd_b(SB + 3 * PP_HB, s);
    if (more1) {
      issue(I1{}, ST ^ 1);
      PP_VMWAIT(8);  // A-bot(t) landed; younger: the four half-tiles of t+1
    } else {
      PP_VMWAIT(0);
    }
    PP_MFMA_SECTION(at, 0, br, 1)
    // ---- P3: A-bottom -> quadrant (1,1); P4 reads nothing, so nothing to wait for
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int s = 0; s < 4; ++s) ab[f][s] = rd_a(SB + PP_HB, f, s);
    if (more2) issue(I0{}, ST);
    PP_MFMA_SECTION(ab, 2, br, 1)
    // ---- P4: quadrant (1,0); A-top / B-left of step t+1 must have landed before P1(t+1)
    if (more2) {
      issue(I2{}, ST);
      PP_VMWAIT(8);  // younger: B-right/A-bot(t+1), A-top/B-left(t+2)
    } else if (more1) {
      PP_VMWAIT(4);  // younger: B-right/A-bot(t+1)
    }
    PP_MFMA_SECTION(ab, 2, bl, 0)
  };

#if PP_PROBE & 16
  const unsigned long long kc0 = __builtin_readcyclecounter(), kr0 = wall_clock64();
#endif
  if (nt > 0) {
    issue_gen(I0{}, 0); issue_gen(I2{}, 0); issue_gen(I3{}, 0); issue_gen(I1{}, 0);
    if (nt > 1) {
      issue_gen(I0{}, 1); issue_gen(I2{}, 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // A-top(0), B-left(0) landed
    } else {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind group 0
    // steady steps: everything prefetched (up to tile t+2) is a full tile of the same K batch
    // (K batches without a K tail are steady too: the in-section issue carries the cursor wrap; TAILS: the zero-page piece too)
    const int has_tail = (kv_last < 64 && t1 == P.KB * kt_per) ? 1 : 0;
    const int n_steady = TAILS ? (max(0, nt - 2) & ~1)
                               : ((kv_last == 64 || P.KB == 1) ? max(0, nt - 2 - has_tail) & ~1 : 0);
    int t = 0;
    if (n_steady > 0) {
      k_step(I0{}, std::true_type{}, 0, std::true_type{});
      k_step(I1{}, std::true_type{}, 1, std::false_type{});
      t = 2;
    }
    for (; t < n_steady; t += 2) {
      k_step(I0{}, std::true_type{}, t, std::false_type{});
      k_step(I1{}, std::true_type{}, t + 1, std::false_type{});
    }
    for (; t < nt; t += 2) {
      k_step(I0{}, std::false_type{}, t, std::false_type{});
      if (t + 1 < nt) k_step(I1{}, std::false_type{}, t + 1, std::false_type{});
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
  }
#undef PP_MFMA_SECTION
#undef PP_MFMA_SECTION_H
#undef PP_MFMA_SECTION_H4
#undef PP_VMWAIT
#undef PP_STAMP
#if PP_PROBE & 16
  pck += __builtin_readcyclecounter() - kc0; prk += wall_clock64() - kr0;
#endif
  __syncthreads();

  // ---- epilogue (same as the 128-wide kernel: per-wave LDS staging -> 16-byte row vectors) -------------------------
  const int mw = m0 + wm * 128, nw = n0 + wn * 64;  // wave tile origin; fragment i -> rows 32 i, j -> cols 32 j
  if constexpr (EP == 0) {
    auto store_block = [&](const f32x16_t (&a)[2], int i) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int nn = nw + j * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int mm = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (mm < P.M && nn < P.N) gemm_store(P, z, split, mm, nn, a[j][r]);
        }
      }
    };
    store_block(acc[0], 0); store_block(acc[1], 1); store_block(acc[2], 2); store_block(acc[3], 3);
  } else {
    constexpr int EP_LD = 64 + 4;
    float* ep = reinterpret_cast<float*>(smem) + wave * (32 * EP_LD);
    // GELU epilogue: the chord table sits behind the staging slices (8 x 8704 B = 68 KiB) at 96 KiB; the K loop is over
    // (barrier above), the next tile's DMA only starts after the barrier that ends this epilogue.  (Keeping the table
    // RESIDENT behind the two stages -- staged once per workgroup instead of once per tile -- was built and measured in
    // round 3: no change, fc1 + GELU 1.50 x the plain launch against 1.48 x, conv1 1.33 x against 1.30 x.)
    const float4* tab = nullptr;
    if constexpr (EP == 3) {
      if (P.gtab) {
        float4* tl = reinterpret_cast<float4*>(smem + 98304);
        gelu_tab_stage(P.gtab, tl);
        __syncthreads();
        tab = tl;
      }
    }
    auto stage_block = [&](const f32x16_t (&a)[2]) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          ep[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * EP_LD + j * 32 + (lane & 31)] = a[j][r];
    };
    constexpr bool CSUM = (EP == 2 || EP == 4) && !GRP;  // fused column sums of C (gemm_common.hpp)
    const bool csum = CSUM && P.colsum_part != nullptr;
    float cs[2] = {0.f, 0.f};
    // one 32-row block out of the staging slice: four 8-wide chunks per lane; axp (EP 4 with PP_AUX_PRELOAD): the block's aux
    // chunks, loaded one block ahead (see gemm_pp3.hip: P3_AUX_PRELOAD)
#if PP_BIAS_LDS
    // the wave's 64 bias values once per tile into a wave-private LDS slice behind the staging slices (see gemm_pp3.hip)
    float* bl = reinterpret_cast<float*>(smem + 69632) + wave * 64;
    const bool bias_pre = (EP == 2 || EP == 3) && P.bias != nullptr;
    if (bias_pre) {
      if (lane < 8) {
        const int n0 = nw + lane * 8;
        float b8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (n0 < P.N) {
          const uint4 raw = *reinterpret_cast<const uint4*>((const bf16_t*)P.bias + (long)zo * P.sBias_o + (long)zi * P.sBias_i + n0);
          gemm_unpack8(raw, b8);
        }
        *reinterpret_cast<float4*>(bl + lane * 8) = make_float4(b8[0], b8[1], b8[2], b8[3]);
        *reinterpret_cast<float4*>(bl + lane * 8 + 4) = make_float4(b8[4], b8[5], b8[6], b8[7]);
      }
    }
    const float* bqp = bias_pre ? bl + (lane & 7) * 8 : nullptr;
#else
    const float* bqp = nullptr;
#endif
    auto process_block = [&](int i, const uint4 (&axp)[4]) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int id = lane + 64 * q;
        const int rl = id >> 3, ch = id & 7;
        const int mm = mw + i * 32 + rl;
        const int nn = nw + ch * 8;
        float vo[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (mm < P.M && nn < P.N) {
          const float4 lo = *reinterpret_cast<const float4*>(ep + rl * EP_LD + ch * 8);
          const float4 hi = *reinterpret_cast<const float4*>(ep + rl * EP_LD + ch * 8 + 4);
          float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
          if constexpr (EP == 1) gemm_store8(P, zo, zi, z, split, mm, nn, v);
          else if constexpr (PP_AUX_PRELOAD && EP == 4 && !GRP) gemm_store8_fast<4>(P, zo, zi, mm, nn, v, tab, CSUM ? vo : nullptr, &axp[q]);
          else gemm_store8_fast<(EP == 2 ? 0 : EP)>(P, zo, zi, mm, nn, v, tab, CSUM ? vo : nullptr, nullptr, bqp);
        }
        if constexpr (CSUM) {
          if (csum) {
            *reinterpret_cast<float4*>(ep + rl * EP_LD + ch * 8) = make_float4(vo[0], vo[1], vo[2], vo[3]);
            *reinterpret_cast<float4*>(ep + rl * EP_LD + ch * 8 + 4) = make_float4(vo[4], vo[5], vo[6], vo[7]);
          }
        }
      }
      if constexpr (CSUM) {
        if (csum) gemm_colsum_block<64, EP_LD>(ep, lane, cs);
      }
    };
    if constexpr (PP_AUX_PRELOAD && EP == 4 && !GRP) {
      // the GELU' factor of the conv stack's / fc2's dX travels one 32-row block ahead in two register sets: a block's loads
      // are issued when the block before it has been staged (profiles/r05/ab_gemm_aux_preload.txt)
      auto aux_load = [&](int i, uint4 (&a)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int id = lane + 64 * q;
          const int rl = id >> 3, ch = id & 7;
          const int mm = mw + i * 32 + rl, nn = nw + ch * 8;
          a[q] = make_uint4(0, 0, 0, 0);
          if (mm < P.M && nn < P.N)
            a[q] = *reinterpret_cast<const uint4*>((const bf16_t*)P.aux + (long)zo * P.sAux_o + (long)zi * P.sAux_i + (long)mm * P.ld_aux + nn);
        }
      };
      uint4 a0[4], a1[4];
      aux_load(0, a0);
      stage_block(acc[0]);
      aux_load(1, a1);
      process_block(0, a0);
      stage_block(acc[1]);
      aux_load(2, a0);
      process_block(1, a1);
      stage_block(acc[2]);
      aux_load(3, a1);
      process_block(2, a0);
      stage_block(acc[3]);
      process_block(3, a1);
    } else {
      const uint4 none[4] = {};
#pragma unroll 1
      for (int i = 0; i < 4; ++i) {
        // constant accumulator indices in every arm: the (large) store code is emitted once, the accumulators stay in registers
        switch (i) {
          case 0: stage_block(acc[0]); break;
          case 1: stage_block(acc[1]); break;
          case 2: stage_block(acc[2]); break;
          default: stage_block(acc[3]); break;
        }
        process_block(i, none);
      }
    }
    if constexpr (CSUM) {
      if (csum) gemm_colsum_finish<64>(P, reinterpret_cast<float*>(smem + 98304), wave, wm, lane, tm, nw, cs);
    }
  }
  __syncthreads();  // every wave has read its staging slice before the next tile's DMA lands in it
#if PP_PROBE & 64
  if (blockIdx.x == 0 && (threadIdx.x & 255) == 0) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long* o = (unsigned long long*)P.C + 2048 + (threadIdx.x >> 8) * 16;
    for (int i = 0; i < 16; ++i) o[i] = ts[i];
  }
#endif
  }
#if PP_PROBE & 16
  if (threadIdx.x == 0) {
    unsigned long long* o = (unsigned long long*)p.C + blockIdx.x * 4;
    o[0] = __builtin_readcyclecounter() - pc0; o[1] = wall_clock64() - pr0; o[2] = pck; o[3] = prk;
  }
#endif
}

int g_pp_mode = 1;  // 0: one block per tile, 1: persistent (256 blocks), 2: persistent with start skew
int g_pp_reserved_cus = 0;  // data-parallel runs: persistent grids leave this many CUs to the RCCL kernels
extern "C" void wavlm_set_reserved_cus(int n) { g_pp_reserved_cus = n < 0 ? 0 : (n > 64 ? 64 : n); }
extern "C" int wavlm_get_reserved_cus(void) { return g_pp_reserved_cus; }

template <bool TA, bool TB>
static int pp_launch_t(GemmP& p, int nbatch, int ep, hipStream_t st) {
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  p.nbatch = nbatch;
  p.vtotal = p.tiles_m * p.tiles_n * nbatch * p.split_k;
  p.patch_m = 0;
  p.swz_r = (nbatch == 1 && p.split_k == 1 && g_pp_mode != 0) ? gemm_pick_swizzle(256, 256, p.tiles_m, p.tiles_n, (long)p.N * p.K * 2) : 0;
  if (p.split_k > 1 && nbatch == 1 && g_pp_mode != 0) {
    p.patch_m = 16 / p.tiles_n; if (p.patch_m < 1) p.patch_m = 1; if (p.patch_m > p.tiles_m) p.patch_m = p.tiles_m;
  }
  const int kt = ((p.K + 63) / 64) * p.KB / p.split_k;
  p.skew = (g_pp_mode == 2 && p.vtotal >= 768) ? (kt * 11 / 100 > 0 ? kt * 11 / 100 : 1) : 0;  // s_sleep(127) ~ 3.9 us; one step ~ a quarter of a tile's main loop (1.72 us per K step)
  // persistent launch pays on the transformer GEMMs (4-5 rounds: +5-10 %); the long conv GEMMs (24 rounds) measured
  // 1 % better with one block per tile
  const int pgrid = 256 - g_pp_reserved_cus;
  dim3 grid((unsigned)(g_pp_mode == 0 || p.vtotal < pgrid || p.vtotal > 2048 ? p.vtotal : pgrid), 1, 1);
  constexpr int smem = 2 * PP_STAGE;
  static bool done[5] = {false, false, false, false, false};
#define PP_CASE(E) case E: { \
    if (!done[E]) { \
      if (hipFuncSetAttribute((const void*)gemm_pp_kernel<TA, TB, E>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return WL_ELAUNCH; \
      done[E] = true; \
    } \
    WL_LAUNCH((gemm_pp_kernel<TA, TB, E>), grid, dim3(512), smem, st, p); } break;
  if constexpr (TA && TB) {
    // weight gradients over K batches with a K tail each (conv stack): the steady-schedule instantiation
    if (ep == 1 && p.KB > 1 && (p.K & 63) != 0) {
      static bool done_t = false;
      if (!done_t) {
        if (hipFuncSetAttribute((const void*)gemm_pp_kernel<true, true, 1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return WL_ELAUNCH;
        done_t = true;
      }
      WL_LAUNCH((gemm_pp_kernel<true, true, 1, false, true>), grid, dim3(512), smem, st, p);
      return wl_check_launch();
    }
  }
  switch (ep) { PP_CASE(0) PP_CASE(1) PP_CASE(2) PP_CASE(3) default: PP_CASE(4) }
#undef PP_CASE
  return wl_check_launch();
}

// Grouped split-K launch of K-strided x K-strided problems (p.grp / p.ngrp / p.vtotal / p.split_k filled by the caller):
// every tile writes its fp32 slab, the caller reduces per problem.
int gemm_pp_launch_grouped(GemmP& p, hipStream_t st) {
  constexpr int smem = 2 * PP_STAGE;
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute((const void*)gemm_pp_kernel<true, true, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return WL_ELAUNCH;
    done = true;
  }
  p.patch_m = 0; p.skew = 0; p.nbatch = 1;
  const int pgrid = 256 - g_pp_reserved_cus;
  dim3 grid((unsigned)(p.vtotal < pgrid ? p.vtotal : pgrid), 1, 1);
  WL_LAUNCH((gemm_pp_kernel<true, true, 1, true>), grid, dim3(512), smem, st, p);
  return wl_check_launch();
}

// Shapes the ping-pong kernel takes: everything stays on 16-byte DMA pieces (K % 8 for K-contiguous operands, M / N
// % 8 for K-strided ones), offsets fit 32 bits, and the tile is worth filling.
bool gemm_pp_ok(const wavlm_gemm_desc* d) {
  if (d->M < 256 || d->N < 128 || d->K < 64) return false;
  if (!d->transA && d->K % 8) return false;
  if (!d->transB && d->K % 8) return false;
  if (d->transA && d->M % 8) return false;
  if (d->transB && d->N % 8) return false;
  const int64_t ra = d->transA ? 64 : 256, rb = d->transB ? 64 : 256;
  if (ra * d->lda * 2 >= (1ll << 31) || rb * d->ldb * 2 >= (1ll << 31)) return false;
  return true;
}

int gemm_pp_launch(GemmP& p, int nbatch, bool transA, bool transB, int ep, hipStream_t st) {
  if (!transA && !transB) return pp_launch_t<false, false>(p, nbatch, ep, st);
  if (!transA && transB) return pp_launch_t<false, true>(p, nbatch, ep, st);
  if (transA && !transB) return pp_launch_t<true, false>(p, nbatch, ep, st);
  return pp_launch_t<true, true>(p, nbatch, ep, st);
}
