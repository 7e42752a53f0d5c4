// LAB RECORD, not part of libwavlm_hip.so (round 5: moved out of unispeech_amd/csrc/).  Built only into the lab library:
// `python tools/probe/build_probe.py lab` -> tools/probe/lib/libwavlm_hip_lab.so (-DWAVLM_EXPERIMENTAL), selected with
// WAVLM_HIP_LIB; parity + timing: tools/h2_ab.py.  Measured 1-22 % slower than the one-workgroup-per-CU kernels on all eight
// transformer shapes (profiles/r04/gemm_h2_ab.txt, gemm_h2_variants.txt).
// bf16 MFMA GEMM for gfx950, short-K path: 192 x 192 x 32 block tile, FOUR waves (96 x 96 accumulators each), three LDS
// stages of 24 KiB -- 72 KiB per workgroup, at most 256 registers per lane -- so that TWO workgroups are resident per CU.
//
// Why: the transformer's activation x weight GEMMs at 24 k rows have K = 768 ... 3072, i.e. 12-48 K steps of 64 per tile, and
// the one-workgroup-per-CU kernels (gemm_pp.hip, gemm_pp3.hip: 128-144 KiB of LDS, all registers) pay every tile's prologue
// (first DMA round trip, pipeline fill) and epilogue (LDS staging, conversion, 150-300 KiB of stores) with an idle matrix
// pipe: measured 16-24 us of fixed cost around a 21 us K loop at K = 768 (tools/gemm_step_table.py: fc1 + GELU 0.25, fc2's
// GELU'-multiplying dX 0.23, out_proj 0.27 of the MFMA peak).  That cost is serial work of the one workgroup a CU holds
// (profiles/HISTORY.md section 4.1: de-phasing, prefetching and pipelined epilogues inside ONE workgroup did not move it).  With
// two resident workgroups the SIMD's other wave runs its K loop while this one stores its tile.
//
// Shape of the kernel (deliberately simpler than the ping-pong kernels: the overlap comes from the second workgroup, not
// from a hand-phased schedule):
//   * operands by LDS-DMA (global_load_lds_dwordx4, SGPR base + VGPR offset), swizzles on the source address:
//       K-contiguous [rows][32 k]: 64-byte rows, 16-byte chunk c stored at slot c ^ ((row >> 2) & 3)  (ds_read_b128),
//       K-strided    [32 k][192 rows]: 384-byte k rows, 32-byte granule g stored at g ^ (2 (k & 3)) within each group of 8
//                                      granules (ds_read_b64_tr_b16, as gemm_pp.hip)
//   * per K step of 32: s_waitcnt vmcnt (own pieces of this step's stage) -> s_barrier -> issue the DMA of step t + 2 into the
//     stage read in step t - 1 -> 12 fragment reads -> 18 MFMA 32x32x16.  Prefetch distance two steps.
//   * epilogue through the workgroup's own LDS in 16-row half blocks (25.6 KiB + the 32 KiB GELU table fit the 72 KiB).
// Handles: no batches, no split-K, K % 32 == 0, the three fast epilogues (bias / GELU + GELU' store / x aux, + bf16
// residual, + fused column sums); everything else stays on the other kernels.
#include "../../unispeech_amd/csrc/gemm_common.hpp"

#include "../../unispeech_amd/csrc/tile_loaders.hpp"

#define H2_BM 192
#define H2_BN 192
#define H2_BK 32
#define H2_OPB (192 * 64)           // bytes of one operand image per stage (either layout)
#define H2_STAGE (2 * H2_OPB)       // 24576
#define H2_NSTAGE 3
#ifndef H2_PIPE
#define H2_PIPE 1   // lab switches: software-pipelined K loop; s_setprio around the MFMA groups
#endif
#ifndef H2_PRIO
#define H2_PRIO 0
#endif
#define H2_SMEM (H2_NSTAGE * H2_STAGE)   // 73728

typedef __attribute__((ext_vector_type(4))) __bf16 h2_bf16x4_t;
typedef __attribute__((address_space(3))) h2_bf16x4_t* h2_lds_b4_ptr;

template <bool TB, int EP>
__global__ __launch_bounds__(256, 2) void gemm_h2_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // XCD-aware tile order (block b runs on XCD b % 8): each XCD walks a contiguous run of tiles
  int tile;
  {
    const int nt = p.tiles_m * p.tiles_n, bid = blockIdx.x;
    const int q = nt >> 3, rem = nt & 7, xcd = bid & 7, idx = bid >> 3;
    tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
  }
  const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
  const int m0 = tm * H2_BM, n0 = tn * H2_BN;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  auto uni_ptr = [](const char* q_) __attribute__((always_inline)) {
    const unsigned long v = (unsigned long)q_;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long)hi << 32) | lo);
  };
  const char* Ab = uni_ptr((const char*)((const bf16_t*)p.A + (long)m0 * p.lda));
  const char* Bb = uni_ptr((const char*)((const bf16_t*)p.B + (TB ? (long)n0 : (long)n0 * p.ldb)));
  const int nt = p.K / H2_BK;

  // ---- DMA side: 24 pieces of 1 KiB per stage (12 per operand), six per wave: piece id = wave + 4 j
  //   K-contiguous piece q: buffer rows 16 q .. +16; lane l -> row 16 q + (l >> 2), slot l & 3 (source chunk slot ^ ((row >> 2) & 3))
  //   K-strided piece q (B only): bytes 1024 q .. of the [32 k][384 B] image; lane l -> byte 1024 q + 16 l: k = byte / 384,
  //     slot (32-byte granule g' = (byte % 384) / 32, half h = (byte / 16) & 1); source granule: see the swizzle below
  unsigned voff[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int q = wave + 4 * j;          // 0 .. 23; q < 12: operand A, else B
    const bool isB = q >= 12;
    const int qq = isB ? q - 12 : q;
    if (!isB || !TB) {
      const long ld = isB ? p.ldb : p.lda;
      const int rows_valid = isB ? p.N - n0 : p.M - m0;
      const int R = qq * 16 + (lane >> 2);
      const int c = (lane & 3) ^ ((R >> 2) & 3);
      int r = R; if (r >= rows_valid) r = rows_valid - 1;
      voff[j] = (unsigned)(((long)r * ld + c * 8) * 2);
    } else {
      const int byte = qq * 1024 + lane * 16;
      const int k = byte / 384, rem = byte - k * 384;
      const int gp = rem >> 5, h = (rem >> 4) & 1;
      // source granule: XOR swizzle inside the group of 8 granules (0-7) and the group of 4 (8-11).  k rows are 384 B apart,
      // i.e. at bank offsets 0 / 128 / 0 / 128 B (mod 256) for k & 3 = 0 .. 3: the four k rows a transposing read touches land
      // in four different 32-byte bank slots (g, g ^ 6, g ^ 4, g ^ 2 resp. x, 4 + x, x ^ 1, 4 + (x ^ 1))
      const int g = gp < 8 ? (gp ^ (2 * (k & 3))) : (8 + ((gp - 8) ^ ((k >> 1) & 1)));
      int col = g * 16 + h * 8;
      if (col + 8 > p.N - n0) col = p.N - n0 - 8;
      voff[j] = (unsigned)(((long)k * p.ldb + col) * 2);
    }
  }
  const long stepA = H2_BK, stepB = TB ? (long)H2_BK * p.ldb : H2_BK;
  long offA = 0, offB = 0;   // element offsets of the K tile being issued next
  auto dma16 = [&](const char* sbase, unsigned voff32, unsigned char* ldst) __attribute__((always_inline)) {
    const unsigned lds_dst = (unsigned)(unsigned long)(las_ptr)ldst;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(lds_dst), "v"(voff32), "s"(sbase) : "memory", "m0");
  };
  auto issue_tile = [&](int stage) __attribute__((always_inline)) {
    const char* sa = Ab + offA * 2;
    const char* sb = Bb + offB * 2;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int q = wave + 4 * j;
      unsigned char* dst = smem + stage * H2_STAGE + (q >= 12 ? H2_OPB + (q - 12) * 1024 : q * 1024);
      dma16(q >= 12 ? sb : sa, voff[j], dst);
    }
    offA += stepA; offB += stepB;
  };

  // ---- fragment side
  // K-contiguous: lane -> row (l & 31) of a 32-row block, chunk 2 s + (l >> 5), swizzle ((l & 31) >> 2) & 3
  unsigned kc[2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
    kc[s] = (unsigned)((lane & 31) * 64 + ((((2 * s + (lane >> 5)) ^ (((lane & 31) >> 2) & 3))) << 4));
  const unsigned a_base = (unsigned)(wm * 96 * 64);
  const unsigned b_base = (unsigned)(H2_OPB + (TB ? 0 : wn * 96 * 64));
  // K-strided B: 16-lane group g1 reads the [4 k][16 cols] block of k octet (l >> 5) of the 16-k slice; lane i of the group
  // supplies the address of k row (i >> 2), columns 4 (i & 3) .. +4 (ds_read_b64_tr_b16); two reads (k 0-3 | 4-7 of the octet)
  unsigned trb[3];
  if constexpr (TB) {
    const int i = lane & 15, g1 = (lane >> 4) & 1;
#pragma unroll
    for (int blk = 0; blk < 3; ++blk) {
      const int gran = (wn * 96 + blk * 32) / 16 + g1;          // logical 32-byte granule (16 columns) of this half block
      const int kq = i >> 2;                                     // k & 3 of the row this lane addresses
      const int gp = gran < 8 ? (gran ^ (2 * kq)) : (8 + ((gran - 8) ^ ((kq >> 1) & 1)));
      trb[blk] = (unsigned)((8 * (lane >> 5) + kq) * 384 + gp * 32 + (i & 3) * 8);
    }
  }
  auto rd_a = [&](int sb, int blk, int s) __attribute__((always_inline)) -> bf16x8_t {
    return *reinterpret_cast<const bf16x8_t*>(smem + sb + a_base + kc[s] + blk * 2048);
  };
  auto rd_b = [&](int sb, int blk, int s) __attribute__((always_inline)) -> bf16x8_t {
    if constexpr (!TB) {
      return *reinterpret_cast<const bf16x8_t*>(smem + sb + b_base + kc[s] + blk * 2048);
    } else {
      const unsigned char* q_ = smem + sb + b_base + trb[blk] + s * (16 * 384);
      const h2_bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((h2_lds_b4_ptr)q_);
      const h2_bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((h2_lds_b4_ptr)(q_ + 4 * 384));
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

  // ---- K loop
#if H2_PRIO
#define H2_PRIO_HI() __builtin_amdgcn_s_setprio(1)
#define H2_PRIO_LO() __builtin_amdgcn_s_setprio(0)
#else
#define H2_PRIO_HI()
#define H2_PRIO_LO()
#endif
  // the second workgroup of a CU (blocks 256 .. 511 of the launch: the dispatcher fills every CU once before it doubles up)
  // starts late, so that the pair is out of phase -- one in its K loop while the other stores (lab switch WAVLM_H2_SKEW)
  if (p.skew > 0 && blockIdx.x >= 256 && blockIdx.x < 512)
    for (int i = 0; i < p.skew; ++i) __builtin_amdgcn_s_sleep(127);
#if H2_PIPE
  // Software-pipelined: the fragments of K half s = 0 of tile t + 1 are read under the MFMAs of half 1 of tile t, those of
  // half 1 under the MFMAs of half 0; the wave meets the others once per tile, in the middle of it: by then it has read all
  // of tile t (stage t % 3 is free for tile t + 3) and tile t + 1 must have landed.
  issue_tile(0);
  if (nt > 1) issue_tile(1);
  if (nt > 2) issue_tile(2);
  if (nt > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (nt > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  bf16x8_t f0a[3], f0b[3], f1a[3], f1b[3];
#pragma unroll
  for (int b = 0; b < 3; ++b) { f0a[b] = rd_a(0, b, 0); f0b[b] = rd_b(0, b, 0); }
  // (each MFMA group issues its first MFMA BEFORE the reads of the next fragments: the compiler's wait for the group's own
  //  fragments then sees nothing younger in the LDS queue)
#define H2_MFMA_GROUP(FA, FB, READS)                                                                   \
  __builtin_amdgcn_sched_barrier(0);                                                                   \
  H2_PRIO_HI();                                                                                        \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0], FB[0], acc[0][0], 0, 0, 0);                \
  __builtin_amdgcn_sched_barrier(0);                                                                   \
  READS                                                                                                \
  __builtin_amdgcn_sched_barrier(0);                                                                   \
  _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                        \
    _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                      \
      if (i + j > 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[i], FB[j], acc[i][j], 0, 0, 0); \
  H2_PRIO_LO();                                                                                        \
  __builtin_amdgcn_sched_barrier(0);
  for (int t = 0; t < nt; ++t) {
    const int sb = (t % H2_NSTAGE) * H2_STAGE;
    H2_MFMA_GROUP(f0a, f0b, _Pragma("unroll") for (int b = 0; b < 3; ++b) { f1a[b] = rd_a(sb, b, 1); f1b[b] = rd_b(sb, b, 1); })
    const bool more = t + 1 < nt;
    const int sn = ((t + 1) % H2_NSTAGE) * H2_STAGE;
    if (more) {
      if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + 3 < nt) issue_tile(t % H2_NSTAGE);
    }
    // (unconditional: behind the last tile these read a stale stage and nobody uses them -- a branch around the reads would make
    //  the compiler's wait counts those of the path without them)
    H2_MFMA_GROUP(f1a, f1b, _Pragma("unroll") for (int b = 0; b < 3; ++b) { f0a[b] = rd_a(sn, b, 0); f0b[b] = rd_b(sn, b, 0); })
  }
#undef H2_MFMA_GROUP
#else
  issue_tile(0);
  if (nt > 1) issue_tile(1);
  for (int t = 0; t < nt; ++t) {
    const int st = t % H2_NSTAGE, sb = st * H2_STAGE;
    // own pieces of tile t landed (younger: tile t + 1, if it exists)
    if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // every wave has its fragments of step t - 1 in registers: the stage read then is free for tile t + 2
    if (t + 2 < nt) issue_tile((t + 2) % H2_NSTAGE);
    bf16x8_t fa[3][2], fb[3][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int b = 0; b < 3; ++b) { fa[b][s] = rd_a(sb, b, s); fb[b][s] = rd_b(sb, b, s); }
    H2_PRIO_HI();
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
    H2_PRIO_LO();
  }
#endif
  __syncthreads();

  // ---- epilogue: 16-row half blocks through the wave's LDS slice -> 16-byte row vectors
  const int mw = m0 + wm * 96, nw = n0 + wn * 96;
  constexpr int EP_LD = 96 + 4;
  float* ep = reinterpret_cast<float*>(smem) + wave * (16 * EP_LD);   // 4 x 6400 B
  const float4* tab = nullptr;
  if constexpr (EP == 3) {
    if (p.gtab) {
      float4* tl = reinterpret_cast<float4*>(smem + 32768);
      gelu_tab_stage(p.gtab, tl);
      __syncthreads();
      tab = tl;
    }
  }
  constexpr bool CSUM = (EP == 2 || EP == 4);
  const bool csum = CSUM && p.colsum_part != nullptr;
  float cs[2] = {0.f, 0.f};
#pragma unroll 1
  for (int ih = 0; ih < 6; ++ih) {   // (32-row block i = ih >> 1, half hh = ih & 1: rows 16 hh .. +16 <-> registers 8 hh .. +8)
    const int hh = ih & 1;
    auto stage_half = [&](const f32x16_t (&a)[3]) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int rr = 8 * hh + r;   // register index: row (rr & 3) + 8 (rr >> 2) + 4 (lane >> 5) of the 32-row block
          const float v = hh ? a[j][8 + r] : a[j][r];
          ep[((rr & 3) + 8 * ((rr >> 2) & 1) + 4 * (lane >> 5)) * EP_LD + j * 32 + (lane & 31)] = v;
        }
    };
    switch (ih >> 1) {
      case 0: stage_half(acc[0]); break;
      case 1: stage_half(acc[1]); break;
      default: stage_half(acc[2]); break;
    }
#pragma unroll
    for (int qv = 0; qv < 3; ++qv) {
      const int id = lane + 64 * qv;        // 192 row vectors of 8 columns: 16 rows x 12 chunks
      const int rl = id / 12, ch = id - rl * 12;
      const int mm = mw + (ih >> 1) * 32 + 16 * hh + rl;
      const int nn = nw + ch * 8;
      float vo[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (mm < p.M && nn < p.N) {
        const float4 lo = *reinterpret_cast<const float4*>(ep + rl * EP_LD + ch * 8);
        const float4 hi = *reinterpret_cast<const float4*>(ep + rl * EP_LD + ch * 8 + 4);
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        gemm_store8_fast<(EP == 2 ? 0 : EP)>(p, 0, 0, mm, nn, v, tab, CSUM ? vo : nullptr);
      }
      if constexpr (CSUM) {
        if (csum) {
          *reinterpret_cast<float4*>(ep + rl * EP_LD + ch * 8) = make_float4(vo[0], vo[1], vo[2], vo[3]);
          *reinterpret_cast<float4*>(ep + rl * EP_LD + ch * 8 + 4) = make_float4(vo[4], vo[5], vo[6], vo[7]);
        }
      }
    }
    if constexpr (CSUM) {
      if (csum) {  // lane c adds up column c (and 64 + c) of the 16 staged rows
        asm volatile("" ::: "memory");
        float a0 = 0.f, b0 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { a0 += ep[r * EP_LD + lane]; if (lane < 32) b0 += ep[r * EP_LD + 64 + lane]; }
        cs[0] += a0; cs[1] += b0;
        asm volatile("" ::: "memory");
      }
    }
  }
  if constexpr (CSUM) {
    if (csum) {   // the two waves of a column range (wm = 0, 1) meet in LDS; wm = 0 writes the tile's partial row segment
      __syncthreads();
      float* xch = reinterpret_cast<float*>(smem + 32768);
      xch[wave * 96 + lane] = cs[0];
      if (lane < 32) xch[wave * 96 + 64 + lane] = cs[1];
      __syncthreads();
      if (wm == 0) {
        const float* other = xch + (wave + 2) * 96;
        float* dst = p.colsum_part + (long)tm * p.N;
        if (nw + lane < p.N) dst[nw + lane] = cs[0] + other[lane];
        if (lane < 32 && nw + 64 + lane < p.N) dst[nw + 64 + lane] = cs[1] + other[64 + lane];
      }
    }
  }
}

// Shapes this kernel takes (see the header); `ec` = gemm_epilogue_class
bool gemm_h2_ok(const wavlm_gemm_desc* d, int ec) {
  if (d->transA) return false;
  if (ec != 2 && ec != 3 && ec != 4) return false;
  if (d->batch_o > 1 || d->batch_i > 1 || d->KB > 1 || d->split_k > 1) return false;
  if (d->M < 192 || d->N < 192 || d->K < 64 || (d->K % H2_BK) != 0) return false;
  if (d->lda < d->K) return false;                       // (overlapping-row operands stay on the conv kernels)
  if (d->transB && (d->N % 8)) return false;
  if (192 * d->lda * 2 >= (1ll << 31) || (d->transB ? 32 : 192) * d->ldb * 2 >= (1ll << 31)) return false;
  return true;
}
int gemm_h2_colsum_rows(const wavlm_gemm_desc* d) { return (d->M + H2_BM - 1) / H2_BM; }

int gemm_h2_launch(GemmP& p, bool transB, int ep, hipStream_t st) {
  p.tiles_m = (p.M + H2_BM - 1) / H2_BM;
  p.tiles_n = (p.N + H2_BN - 1) / H2_BN;
  const dim3 grid((unsigned)(p.tiles_m * p.tiles_n));
  static const int skew = [] { const char* e = getenv("WAVLM_H2_SKEW"); return e ? atoi(e) : 0; }();
  p.skew = skew;
  static bool done[2][5] = {};
#define H2_CASE(TB_, E) { \
    if (!done[TB_][E]) { \
      if (hipFuncSetAttribute((const void*)gemm_h2_kernel<TB_, E>, hipFuncAttributeMaxDynamicSharedMemorySize, H2_SMEM) != hipSuccess) return WL_ELAUNCH; \
      done[TB_][E] = true; \
    } \
    WL_LAUNCH((gemm_h2_kernel<TB_, E>), grid, dim3(256), H2_SMEM, st, p); }
  if (!transB) { if (ep == 2) H2_CASE(false, 2) else if (ep == 3) H2_CASE(false, 3) else H2_CASE(false, 4) }
  else { if (ep == 2) H2_CASE(true, 2) else if (ep == 3) H2_CASE(true, 3) else H2_CASE(true, 4) }
#undef H2_CASE
  return wl_check_launch();
}
