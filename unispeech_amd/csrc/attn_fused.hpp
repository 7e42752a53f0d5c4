// Shared pieces of the fused attention kernels (attn_fused.hip: forward, dQ kernel, launchers; attn_fused_dkv.hip: the
// dK/dV kernel, a translation unit of its own because it is compiled with -fno-slp-vectorize, see build.py).
#pragma once
// Fused gated-relative-position attention for gfx950 (bf16, head_dim 64): QK^T + Toeplitz bias + key padding +
// online softmax + dropout + PV in one kernel, and a two-kernel backward that recomputes the probabilities from
// the saved log-sum-exp.  Nothing of size [B*H, T, T] ever reaches HBM (the reference writes the bias, the scores
// and the probabilities, 862 MB each at B=32: WavLM/modules.py:504-563 + SDPA with a float mask).
//
// Layout choice (all three kernels): scores are produced TRANSPOSED, S^T = K.Q^T, with v_mfma_f32_32x32x16_bf16.
// In the C/D layout (col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)) a lane then owns ONE query row and 16
// keys per 32x32 block, so
//   * row max / row sum are 15 in-register ops + one exchange with lane ^ 32 (no LDS, no 32-lane shuffles),
//   * the per-row scalars (gate, running max, normaliser, lse, delta) are plain per-lane registers,
//   * the probabilities feed the next MFMA directly as its B operand: registers r = 8s .. 8s+7 of a block ARE the
//     eight k-slots of k-step s.  The A operand (V^T or K^T rows from LDS) is read with the matching key
//     permutation k-slot (hi, e) <-> key 16s + 4hi + (e&3) + 8(e>>2): two 8-byte LDS reads instead of one 16-byte.
// The key-contraction products (O = P V, dQ = dS K) therefore need no cross-lane data movement at all.  The
// query-contraction products (dV = P^T dO, dK = dS^T Q) use the untransposed layout in their own kernel.
//
// Bias: bias[i, j] = gate[b,h,i] * rel[h, j - i] is Toeplitz; rel[h, :] (2T-1 floats) is staged in LDS once per
// block.  Dropout: stateless hash of the element index (b,h,i,j) -> identical mask in all three kernels
// regardless of which lane holds the element.
#include "tile_loaders.hpp"
#include "../../include/wavlm_hip.h"

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((address_space(3))) bf16x4_t* lds_b4_ptr;

#define FA_HD 64
#define FA_BQ 128   // query rows per block (forward, dQ kernel): 4 waves x 32
#define FA_BKV 64   // keys per iteration
#define FA_BK1 128  // key rows per block (dK/dV kernel): 4 waves x 32
#define FA_BQ1 64   // query rows per iteration (dK/dV kernel)
#define FA_K64 256  // key rows per block of the 64-keys-per-wave dK/dV kernel (attn_fused_dkv64.hip): 4 waves x 64

// Dropout mask: stateless, identical in all three kernels whichever lane holds element (i, j).
//   word(i, j >> 1) = mix(row_word(b, h, i) + col_word(j >> 1));  keep(i, j) = 16-bit half (j & 1) of it >= th16
// row_word / col_word are strong multiplicative hashes (3 x v_mul_lo_u32 each, quarter rate) evaluated once per
// query row / key pair; the per-element work is only the multiply-free mix (xor-shift / shift-add, full rate) --
// the previous per-element multiplicative hash was ~1/3 of the forward kernel's VALU time.  The 1/(1-p) factor is
// never applied per element: forward folds it into the final 1/l, backward into the exponent (lse - log2 sc) and
// into delta / sc.
__device__ __forceinline__ unsigned fa_row_word(unsigned s0, unsigned grow) { return hash32(grow ^ s0); }
__device__ __forceinline__ unsigned fa_col_word(unsigned s1, unsigned jpair) { return hash32((jpair ^ s1) + 0x68E31DA4u); }
__device__ __forceinline__ unsigned fa_mix(unsigned x) { return drop_mix(x); }
// Keep decisions of the fused kernels: the two 16-bit halves of a word as SIGNED numbers >= ths = th - 32768 (the same
// probability as the unsigned form).  Signed, because the forward applies the mask to the PACKED bf16 pair with three
// packed-integer instructions and no compare / VCC / select:  d = sat(ths - 1 - half) is negative iff the half is kept,
// d >> 15 (arithmetic) is the 0xffff / 0 keep mask of each half, one v_and_b32 applies both.
typedef short s16x2_t __attribute__((ext_vector_type(2)));
// the complement for the stored-probability forward: 0xffff in the halves that are DROPPED (half < ths): d = sat(half - ths) is
// negative iff the half is dropped; k3 = ths in both halves
__device__ __forceinline__ unsigned fa_dropmask2(unsigned w, unsigned k3) {
  s16x2_t d = __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2_t, w), __builtin_bit_cast(s16x2_t, k3));
  d = d >> (short)15;
  return __builtin_bit_cast(unsigned, d);
}
__device__ __forceinline__ unsigned fa_keepmask2(unsigned w, unsigned k2) {
  s16x2_t d = __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2_t, k2), __builtin_bit_cast(s16x2_t, w));
  d = d >> (short)15;
  return __builtin_bit_cast(unsigned, d);
}
__device__ __forceinline__ bool fa_keep_lo(unsigned w, int ths) { return (int)(short)(w & 0xffffu) >= ths; }
__device__ __forceinline__ bool fa_keep_hi(unsigned w, int ths) { return ((int)w >> 16) >= ths; }
#define FA_LOG2E 1.4426950408889634f
#define FA_LN2 0.6931471805599453f

__device__ __forceinline__ unsigned pack_bf16(float a, float b) { return pack_bf16x2(a, b); }

// A-operand fragment for a contraction over the tile's 64 "k" positions stored along LDS rows ([rows][64 k]):
// k-slot (hi, e) of k-step (f, s) <-> position 32f + 16s + 4hi + (e&3) + 8(e>>2)
__device__ __forceinline__ bf16x8_t frag_perm(const unsigned char* lds, int row, int f, int s, int hi) {
  const int c0 = 4 * f + 2 * s;
  const uint2 lo = *reinterpret_cast<const uint2*>(lds + lds_off(row, c0) + 8 * hi);
  const uint2 hi2 = *reinterpret_cast<const uint2*>(lds + lds_off(row, c0 + 1) + 8 * hi);
  U4 u; u.v = make_uint4(lo.x, lo.y, hi2.x, hi2.y);
  return u.b;
}
// plain fragment (row, 8 consecutive k at 16kk + 8hi)
__device__ __forceinline__ bf16x8_t frag_plain(const unsigned char* lds, int row, int kk, int hi) {
  U4 u; u.v = *reinterpret_cast<const uint4*>(lds + lds_off(row, 2 * kk + hi));
  return u.b;
}

// Transposed A-operand fragment straight from a K-contiguous [64 keys][64 hd] tile (no separately staged transpose):
// rows = head-dim 32 f2 + (l & 31), k-slot (hi, e) of k-step (f, s) <-> key 32f + 16s + 4hi + (e&3) + 8(e>>2), the
// order in which a lane holds P / dS.  ds_read_b64_tr_b16: lane i of a 16-lane group supplies the address of key
// (i >> 2), head-dim columns 4 (i & 3) .. +4 and receives column i of the 4 x 16 block.  `tr_base` is the per-lane
// part of the (swizzled) address, fa_tr_base(); everything else is an immediate.
__device__ __forceinline__ unsigned fa_tr_base(int lane) {
  const int hi = lane >> 5, li = lane & 15, g1 = (lane >> 4) & 1;
  return (unsigned)((4 * hi + (li >> 2)) * 128 + ((((g1 ^ hi) << 1) | (((li >> 1) & 1) ^ (li >> 3))) << 4) + (li & 1) * 8);
}
__device__ __forceinline__ bf16x8_t frag_tr(const unsigned char* tile, unsigned tr_base, int f2, int f, int s) {
  const unsigned char* kt = tile + tr_base + (32 * f + 16 * s) * 128;
  const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_ptr)(kt + (f2 << 6)));
  const bf16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_ptr)(kt + 8 * 128 + ((f2 ^ 1) << 6)));
  return __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
}

// LDS-DMA of one 1 KiB piece (16 bytes per lane, lane l lands at ldst + 16 l; ldst wave-uniform).
// FA_ASM_DMA (round 5): inline asm with its own M0 write instead of __builtin_amdgcn_global_load_lds.  With the builtin
// anywhere in a loop the compiler drains `s_waitcnt vmcnt(0)` in front of EVERY later LDS read (it cannot tell which LDS
// bytes the DMA writes): the "prefetch" of the next K / V tile was waited for right after it was issued, in front of the
// first fragment read of the CURRENT tile -- every wave paid one full L2 / HBM round trip per tile and only the other
// resident waves covered it (two or three per SIMD).  Through asm the compiler does not see a memory operation it has to
// order; the kernels wait themselves, once, in front of the barrier that ends the tile (fa_tile_sync).  Safe against the
// compiler's own vmcnt bookkeeping: vector memory operations return in order on gfx9, so operations it does not know of
// can only make its counted waits wait longer, never shorter.
#ifndef FA_ASM_DMA
#define FA_ASM_DMA 1
#endif
// lab-bench probes of the stored-probability kernels (WRONG results by construction; -DWAVLM_EXPERIMENTAL builds only):
// FA_SP_PROBE bit 0: the backward kernels do not load P16 (registers keep their first tile) -- what the HBM stream costs;
// bit 1: the forward converts but does not store; bit 2: dQ kernel without the skew writes; bit 3: dK/dV kernel without the
// LDS round trip of the fragments; bit 4: no wait for the tile DMA at the end of a tile (races) -- what that wait costs
#if !defined(WAVLM_EXPERIMENTAL)
#undef FA_SP_PROBE
#endif
#ifndef FA_SP_PROBE
#define FA_SP_PROBE 0
#endif
__device__ __forceinline__ void fa_dma16(const void* lane_ptr, unsigned char* ldst) {
#if FA_ASM_DMA
  const unsigned lds_dst = (unsigned)(unsigned long)(las_ptr)ldst;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(lds_dst), "v"(lane_ptr) : "memory", "m0");
#else
  __builtin_amdgcn_global_load_lds((gas_ptr)lane_ptr, (las_ptr)ldst, 16, 0, 0);
#endif
}
// end of a tile iteration: this wave's DMA pieces have landed, then every wave's (N: vector memory operations issued AFTER
// the DMA that may stay in flight -- stores of the iteration)
template <int N = 0> __device__ __forceinline__ void fa_tile_sync() {
#if FA_ASM_DMA && !(FA_SP_PROBE & 16)
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  // the builtin, not asm: the compiler's wait-count pass reads an explicit s_waitcnt and learns from it that ITS OWN loads
  // (q / dO fragments fetched in the prologue) have landed -- behind an opaque asm wait it kept `s_waitcnt vmcnt(3..0)` in
  // front of the first MFMAs of every iteration, which drained the prefetch again
  __builtin_amdgcn_s_waitcnt(0x0f70 | (N & 15) | ((N >> 4) << 14));  // vmcnt(N); expcnt / lgkmcnt fields at their maxima
#endif
  __syncthreads();
}

// K-contiguous [64 rows][64] tile -> LDS through LDS-DMA; rows past `nrows` are clamped (results unused/masked)
__device__ __forceinline__ void glds_tile64(const bf16_t* base, long ld, int row0, int nrows, unsigned char* lds,
                                            int wave_u) {
  const int t = threadIdx.x;
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int rr = (t >> 3) + 32 * ps;
    int row = row0 + rr; if (row > nrows - 1) row = nrows - 1;
    const bf16_t* src = base + (long)row * ld + (((t & 7) ^ ((rr >> 1) & 7)) << 3);
    fa_dma16(src, lds + (ps * 32 + wave_u * 8) * 128);
  }
}

// block -> (tile along the sequence, batch*head).  The grid is one-dimensional: the dispatcher hands consecutive
// workgroups to consecutive XCDs (private 4 MiB L2 each), so with the natural order the nqb tiles of one (b, h) -- the
// only blocks that share K / V (Q / dO in the dK/dV kernel) -- landed on nqb DIFFERENT XCDs and every one of them pulled
// its own copy from HBM: 483 MB fetched per forward launch against 150 MB algorithmic (PMC, profiles/r01).  Here XCD x
// takes the heads bh = x (mod 8) and walks their tiles back to back, so a head's tiles run on one XCD at the same time
// and share its L2.
__device__ __forceinline__ void fa_block_map(int nqb, int BH, int& qb, int& bh) {
  const int L = blockIdx.x;
  if ((BH & 7) == 0) {
    const int xcd = L & 7, idx = L >> 3;
    bh = (idx / nqb) * 8 + xcd;
    qb = idx - (idx / nqb) * nqb;
  } else {
    bh = L / nqb;
    qb = L - bh * nqb;
  }
}

// ---- stored probabilities (round 5) ---------------------------------------------------------------------------------
// The backward kernels above recompute P = softmax(...) and the dropout decisions from (q, k, bias, lse, seed): the whole
// element pass (bias fma, 2^x, dropout word, select) runs THREE times per layer (forward, dQ kernel, dK/dV kernel) on a
// part whose VALU is the saturated resource of these kernels (12-21 VALU instructions per MFMA) while < 1 TB/s of its
// 8 TB/s HBM and a few GB of its 288 GB are in use.  With a `pstore` buffer the forward writes what it holds anyway:
//   P16[bh][q32][jt][f][s2][lane][8]   fp16, FRAGMENT-NATIVE: the 16 bytes of a lane are the forward's own B-operand
//       fragment pf[f][s2] of key tile jt (lane = 32 hi + query row % 32 of the 32-row block q32; the eight values are keys
//       64 jt + 32 f + 16 s2 + 4 hi + {0..3, 8..11}) -- one fully coalesced 1 KiB store per wave, f, s2.  Value = the
//       forward's p = 2^(x - m_jt) relative to the RUNNING maximum after tile jt (<= 1: fp16 carries it with 11 bits,
//       4x finer than the bf16 the PV product uses; values below 6e-8 flush to zero), SIGN BIT = the element was dropped.
//   mt[bh][jt][Tq]   fp32: that running maximum (log2 domain), Tq = rows padded to 128.
// Backward: P * sc = |P16| * 2^(mt + log2 sc - lse log2 e), keep = sign clear.  The dQ kernel reads its fragments straight
// back into registers (same decomposition as the forward); the dK/dV kernel gathers them into LDS by LDS-DMA and reads them
// transposed with ds_read_b64_tr_b16.  Per element: convert, scale, max(., 0), multiply, fma -- no score MFMA, no bias, no
// exponential, no hash.  Cost: 4 KiB per (32 rows x 64 keys) = 453 MB per Base layer at 32 x 15 s (5.4 GB per step), read
// once by each backward kernel.  Rows / keys past T: the forward writes every tile of the padded grid (clamped rows give
// finite values), the backward zeroes them through the scale (rows) or discards them (keys).
#define FA_PTILE_BYTES 4096
#define FA_PSTORE_MAX_T 1024   // wavlm_attn_fused_pstore_bytes returns 0 beyond (callers then recompute)
struct FaPstore { unsigned char* P16; float* mt; int nq32, nkv, Tq; };
__host__ __device__ inline uint64_t fa_pstore_p_bytes(int B, int H, int T) {
  const uint64_t nq32 = (uint64_t)((T + FA_BQ - 1) / FA_BQ) * 4, nkv = (uint64_t)((T + FA_BKV - 1) / FA_BKV);
  return (uint64_t)B * H * nq32 * nkv * FA_PTILE_BYTES;
}
__host__ __device__ inline uint64_t fa_pstore_mt_bytes(int B, int H, int T) {
  const uint64_t Tq = (uint64_t)((T + FA_BQ - 1) / FA_BQ) * FA_BQ, nkv = (uint64_t)((T + FA_BKV - 1) / FA_BKV);
  return (uint64_t)B * H * nkv * Tq * sizeof(float);
}

struct FaP {
  const bf16_t* qkv; bf16_t* O; float* lse;
  const float* gate; const float* tab; const unsigned char* kpm;
  const bf16_t* dO; bf16_t* dqkv; float* delta; float* dgate; float* dtab_part;
  float* dbias_part;  // optional [B * nqb * 4][3 * H * 64]: per-wave column sums of dq | dk | dv (the q|k|v bias gradient)
  int B, H, T; float scale; float sc2; unsigned th; float sc, log2sc, inv_sc; unsigned s0, s1;  // th: 16-bit keep threshold (0 = no dropout)
  int ths; unsigned k2;  // signed threshold th - 32768; (ths - 1) in both halves
  unsigned k3;           // ths in both halves (fa_dropmask2)
  FaPstore ps;           // stored probabilities (P16 == nullptr: recompute)
  // dropout decisions as bit words, written by the dQ kernel for the 64-keys-per-wave dK/dV kernel (attn_fused_dkv64.hip):
  // dbits[(bh * db_nkb + c) * db_Tq + fa_bitrow(i)], bit k = keep(row i, key 32 c + k); db_nkb = 2 x key tiles of the dQ kernel,
  // db_Tq = rows padded to its 128-row blocks.  nullptr: not written
  unsigned* dbits; int db_nkb, db_Tq;
  // db_fwd = 1: the words were written by the FORWARD (attn_fwd_kernel<true, false, true>) in ITS bit order (fa_fbit) and the dQ
  // kernel consumes them as well; 0: written by the dQ kernel of this backward, bit k = key k of the block
  int db_fwd;
  int Ltab, Tkb;  // LDS extents: rel table (zero padded) and key bias
  int nqb;        // tiles along the sequence per (b, h) of the kernel being launched
};


// Column sums of a wave's gradient rows (the packed q|k|v projection's bias gradient falls out of the kernels that
// produce dq / dk / dv instead of a separate pass over dqkv).  The wave's 32 x 64 fp32 block goes through its own LDS
// slice (rows padded to 68 floats: 16-byte row-vector writes, conflict-free column reads), lane c adds up column c and
// writes it to the wave's own partial row -- no barrier, every (row, column) of the partial matrix has one writer.
// `red`: 4 * 32 * 68 floats of LDS, free after the tile loop's last barrier.
// (History: a butterfly of 160 ds_bpermute per wave cost 28-58 us per launch; a block-level sum with two barriers per call
// 20 us per launch -- 2300 blocks pay the barrier latency four and a half deep per CU.)
#define FA_CS_LD 68
#define FA_CS_FLOATS (4 * 32 * FA_CS_LD)
__device__ __forceinline__ void fa_wave_colsum(const f32x16_t (&a)[2], float scale, float* red, float* dst, int lane,
                                               int wave) {
  const int hi = lane >> 5, ql = lane & 31;
  float* buf = red + wave * (32 * FA_CS_LD);
#pragma unroll
  for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const float4 v = make_float4(a[f2][4 * q4] * scale, a[f2][4 * q4 + 1] * scale, a[f2][4 * q4 + 2] * scale,
                                   a[f2][4 * q4 + 3] * scale);
      *reinterpret_cast<float4*>(buf + ql * FA_CS_LD + 32 * f2 + 8 * q4 + 4 * hi) = v;
    }
  asm volatile("" ::: "memory");  // same wave, LDS executes in order; the compiler must keep the order too
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
  for (int r = 0; r < 32; r += 4) {
    s0 += buf[r * FA_CS_LD + lane]; s1 += buf[(r + 1) * FA_CS_LD + lane];
    s2 += buf[(r + 2) * FA_CS_LD + lane]; s3 += buf[(r + 3) * FA_CS_LD + lane];
  }
  dst[lane] = (s0 + s1) + (s2 + s3);
  asm volatile("" ::: "memory");  // the slice is reused by the wave's next call
}

// The same tile loader with the address arithmetic taken out of the tile loop: per-lane row pointers once (two 16-byte pieces
// per lane and tile), per tile one uniform offset.  (glds_tile64 re-derives row, clamp and a 64-bit row * ld product per
// piece: 8 quarter-rate v_mul_lo_u32 + 6 v_mad_u64_u32 per pair of tiles and iteration, ~250 cycles of every tile's ~4500.)
// Tiles whose rows all exist (`row0 + 64 <= nrows`) take the fast path; the last, partial tile the clamping one.
// FA_TILE_SRC: the forward / dQ / recompute dK-dV kernels through FaTileSrc as well.  Measured neutral on all of them (same-box
// A/B, profiles/r05/ab_tilesrc.txt: they are LDS-bound, not VALU-bound) at +8 VGPRs: OFF; the stored-probability dK/dV kernel
// uses it unconditionally.
#ifndef FA_TILE_SRC
#define FA_TILE_SRC 0
#endif
struct FaTileSrc {
  const bf16_t* lane_ptr[2];   // row (t >> 3) + 32 ps of tile 0, this lane's (swizzled) 16-byte chunk
  const bf16_t* base; long ld; int nrows;
  __device__ __forceinline__ void init(const bf16_t* base_, long ld_, int nrows_) {
    base = base_; ld = ld_; nrows = nrows_;
    const int t = threadIdx.x;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int rr = (t >> 3) + 32 * ps;
      lane_ptr[ps] = base_ + (long)rr * ld_ + (((t & 7) ^ ((rr >> 1) & 7)) << 3);
    }
  }
  __device__ __forceinline__ void issue(int row0, unsigned char* lds, int wave_u) const {
    if (row0 + 64 <= nrows) {
      const long off = (long)row0 * ld;   // uniform
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) fa_dma16(lane_ptr[ps] + off, lds + (ps * 32 + wave_u * 8) * 128);
    } else {
      glds_tile64(base, ld, row0, nrows, lds, wave_u);
    }
  }
};

// position of row i inside its group of eight in the dropout bit words: 0 4 1 5 2 6 3 7 -- rows i and i + 4 (the two half-waves
// of the dK/dV kernel's 32 x 32 blocks) are neighbours, so that their two words form one aligned SGPR pair of a scalar load
__host__ __device__ inline int fa_bitrow(int i) { return (i & ~7) + 2 * (i & 3) + ((i >> 2) & 1); }
// Bit order of the words the forward writes.  A forward lane (row i, half-wave hi) holds the 16 keys (r & 3) + 8 (r >> 2) + 4 hi,
// r = 0..15, of a 32-key block, walks them in pairs and shifts every pair's two decisions in from the top
// (acc = acc >> 1 | mask & 0x80008000): register r ends at bit (r >> 1) + 16 (r & 1) + 8 of the half-wave's own word, and the
// halves combine as hi = 0 | (hi = 1) >> 8.  fa_fbit_of_key: the bit of key k (0..31) of the block.
__host__ __device__ inline int fa_fbit_of_key(int k) {
  const int hi = (k >> 2) & 1, r = (k & 3) + 4 * (k >> 3);
  return (r >> 1) + 16 * (r & 1) + 8 * (1 - hi);
}
__host__ __device__ inline uint64_t fa_dbits_bytes(int B, int H, int T) {
  const uint64_t nkb = (uint64_t)((T + FA_BKV - 1) / FA_BKV) * 2, Tq = (uint64_t)((T + FA_BQ - 1) / FA_BQ) * FA_BQ;
  return (uint64_t)B * H * nkb * Tq * sizeof(unsigned) + 256;   // + one scalar-load line of slack behind the last word
}
// eight rows' words of one 32-key block = one scalar load (constant address space: the address is wave-uniform, the words were
// written by the kernel before); words 2 e, 2 e + 1 = rows e, e + 4 of the group = the 64-bit lane mask of register "row e"
typedef unsigned u32x8_t __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(4))) u32x8_t* fa_mask_ptr;
typedef fa_mask_ptr k64_mask_ptr;
// select by an SGPR-pair lane mask: lanes whose bit is set keep `v`, the others get 0.  `v` comes straight out of v_exp_f32: on
// gfx940+ a non-transcendental VALU instruction that reads the result of a transcendental one needs one wait state in between,
// and the compiler's hazard recogniser does not look into inline asm (seen: the select read the register's OLD content) --
// the s_nop is that wait state
__device__ __forceinline__ float k64_keep(float v, unsigned lo, unsigned hi) {
  const unsigned long m = ((unsigned long)hi << 32) | lo;
  float r;
  asm("s_nop 0\n\tv_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(m));
  return r;
}
// the 64-keys-per-wave dK/dV kernel (attn_fused_dkv64.hip); p.nqb = ceil(T / FA_K64).  WL_EINVAL: not applicable (the caller
// launches the 32-keys-per-wave kernel)
int fa_launch_dkv64(const FaP& p, unsigned grid, hipStream_t st);
size_t fa_dkv64_smem(const FaP& p);

// launcher of the dK/dV kernel (attn_fused_dkv.hip); returns a WL_* code.  smem: of the recompute form; the stored-P form
// sizes its own
int fa_launch_dkv(const FaP& p, unsigned grid, size_t smem, hipStream_t st);

template <typename K> static int fa_set_smem(K kernel, size_t bytes) {
  if (bytes <= 65536) return WL_OK;
  return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess
             ? WL_OK : WL_ELAUNCH;
}
