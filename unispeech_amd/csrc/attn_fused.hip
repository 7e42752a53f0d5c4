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

// K-contiguous [64 rows][64] tile -> LDS through LDS-DMA; rows past `nrows` are clamped (results unused/masked)
__device__ __forceinline__ void glds_tile64(const bf16_t* base, long ld, int row0, int nrows, unsigned char* lds,
                                            int wave_u) {
  const int t = threadIdx.x;
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int rr = (t >> 3) + 32 * ps;
    int row = row0 + rr; if (row > nrows - 1) row = nrows - 1;
    const bf16_t* src = base + (long)row * ld + (((t & 7) ^ ((rr >> 1) & 7)) << 3);
    __builtin_amdgcn_global_load_lds((gas_ptr)src, (las_ptr)(lds + (ps * 32 + wave_u * 8) * 128), 16, 0, 0);
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

struct FaP {
  const bf16_t* qkv; bf16_t* O; float* lse;
  const float* gate; const float* tab; const unsigned char* kpm;
  const bf16_t* dO; bf16_t* dqkv; float* delta; float* dgate; float* dtab_part;
  int B, H, T; float scale; float sc2; unsigned th; float sc, log2sc, inv_sc; unsigned s0, s1;  // th: 16-bit keep threshold (0 = no dropout)
  int ths; unsigned k2;  // signed threshold th - 32768; (ths - 1) in both halves
  int Ltab, Tkb;  // LDS extents: rel table (zero padded) and key bias
  int nqb;        // tiles along the sequence per (b, h) of the kernel being launched
};

// ------------------------------------------------------------------------------------------------- forward
template <bool DROP>
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(FaP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  auto kbuf = [&](int st) { return smem + st * 16384; };
  auto vbuf = [&](int st) { return smem + st * 16384 + 8192; };
  float* tabs = reinterpret_cast<float*>(smem + 32768);
  float* kb = tabs + p.Ltab;
  unsigned* colw = reinterpret_cast<unsigned*>(kb + p.Tkb);  // [Tkb / 2] dropout column words
  const int T = p.T, H = p.H;
  int qblk, bh;
  fa_block_map(p.nqb, p.B * H, qblk, bh);
  const int b = bh / H, h = bh % H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int hi = lane >> 5, ql = lane & 31;
  const int i = qblk * FA_BQ + 32 * wave + ql;
  const int ic = i < T ? i : T - 1;
  const long D3 = 3L * H * FA_HD;
  const bf16_t* base = p.qkv + (long)b * T * D3 + h * FA_HD;
  const int L = 2 * T - 1;

  U4 qf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) qf[kk].v = *reinterpret_cast<const uint4*>(base + (long)ic * D3 + 16 * kk + 8 * hi);
  for (int d = threadIdx.x; d < p.Ltab; d += 256) tabs[d] = (p.tab && d < L) ? p.tab[(long)h * L + d] : 0.f;
  for (int j = threadIdx.x; j < p.Tkb; j += 256)
    kb[j] = (j < T && !(p.kpm && p.kpm[(long)b * T + j])) ? 0.f : -INFINITY;
  if constexpr (DROP)
    for (int jp = threadIdx.x; jp < (p.Tkb >> 1); jp += 256) colw[jp] = fa_col_word(p.s1, (unsigned)jp);
  // everything below lives in the log2 domain: x2 = log2(e) * (scale * s + gate * rel), p = 2^(x2 - m2)
  const float g2 = p.gate ? p.gate[(long)bh * T + ic] * FA_LOG2E : 0.f;
  const unsigned roww = fa_row_word(p.s0, (unsigned)(bh * T + ic));
  const float* trow = tabs + (T - 1 - ic);  // trow[j] = rel[h, j - i]

  f32x16_t o[2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[f][r] = 0.f;
  float m = -INFINITY, l = 0.f;
  const int nkv = (T + FA_BKV - 1) / FA_BKV;
  const unsigned vtr = fa_tr_base(lane);

  glds_tile64(base + FA_HD * H, D3, 0, T, kbuf(0), wave_u);
  glds_tile64(base + 2 * FA_HD * H, D3, 0, T, vbuf(0), wave_u);
  __syncthreads();

  int cur = 0;
  for (int jt = 0; jt < nkv; ++jt) {
    const int j0 = jt * FA_BKV;
    const bool more = jt + 1 < nkv;
    if (more) {
      glds_tile64(base + FA_HD * H, D3, j0 + FA_BKV, T, kbuf(cur ^ 1), wave_u);
      glds_tile64(base + 2 * FA_HD * H, D3, j0 + FA_BKV, T, vbuf(cur ^ 1), wave_u);
    }
    // S^T = K Q^T
    f32x16_t s[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[f][r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        s[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain(kbuf(cur), 32 * f + ql, kk, hi), qf[kk].b, s[f], 0, 0, 0);
    }
    // scores -> log2 domain with the Toeplitz bias; key padding / keys past T only on edge tiles
    float tmax = -INFINITY;
    const bool edge = (p.kpm != nullptr) || (j0 + FA_BKV > T);
    if (!edge) {
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = j0 + 32 * f + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const float x = fmaf(s[f][r], p.sc2, g2 * trow[j]);
          s[f][r] = x;
          tmax = fmaxf(tmax, x);
        }
    } else {
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = j0 + 32 * f + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const float x = fmaf(s[f][r], p.sc2, g2 * trow[j]) + kb[j];
          s[f][r] = x;
          tmax = fmaxf(tmax, x);
        }
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m, tmax);
    const bool dead = (m_new == -INFINITY);
    const float alpha = dead ? 1.f : __builtin_amdgcn_exp2f(m - m_new);
    const float msub = dead ? 0.f : m_new;  // dead rows: every x is -inf -> 2^(-inf) = 0
    float rs = 0.f;
    U4 pf[2][2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __builtin_amdgcn_exp2f(s[f][r] - msub);
        const float p1 = __builtin_amdgcn_exp2f(s[f][r + 1] - msub);
        rs += p0 + p1;
        unsigned pk = pack_bf16(p0, p1);
        if constexpr (DROP) {
          const int jp = (j0 + 32 * f + (r & 3) + 8 * (r >> 2) + 4 * hi) >> 1;
          pk &= fa_keepmask2(fa_mix(roww + colw[jp]), p.k2);
        }
        pf[f][r >> 3].u[(r & 7) >> 1] = pk;
      }
    rs += __shfl_xor(rs, 32, 64);
    l = l * alpha + rs;
    m = m_new;
    if (__any(alpha != 1.f)) {  // the running maximum settles after the first tiles: skip the rescale then
#pragma unroll
      for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[f2][r] *= alpha;
    }
    // O^T += V^T P^T
#pragma unroll
    for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
          o[f2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(vbuf(cur), vtr, f2, f, s2), pf[f][s2].b, o[f2], 0, 0, 0);
    __syncthreads();
    cur ^= 1;
  }
  if (i < T) {
    const float inv = l > 0.f ? p.sc / l : 0.f;  // dropout's 1/(1-p) rides on the normaliser
    bf16_t* dst = p.O + ((long)b * T + i) * (H * FA_HD) + h * FA_HD;
#pragma unroll
    for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        uint2 w;
        w.x = pack_bf16(o[f2][4 * q4] * inv, o[f2][4 * q4 + 1] * inv);
        w.y = pack_bf16(o[f2][4 * q4 + 2] * inv, o[f2][4 * q4 + 3] * inv);
        *reinterpret_cast<uint2*>(dst + 32 * f2 + 8 * q4 + 4 * hi) = w;
      }
    if (hi == 0) p.lse[(long)bh * T + i] = (m + __log2f(l)) * FA_LN2;
  }
}

// ------------------------------------------------------------------------------- backward 1/2: dQ, dgate, drel
// Same decomposition as the forward (lane owns a query row).  Also writes delta[i] = <dO_i, O_i> for kernel 2/2.
//
// drel[h, d] = sum_i gate_i dS[i, i + d - (T-1)] is a sum along diagonals.  Per key tile a wave holds U = gate * dS
// for 32 query rows x 64 keys, one row per lane, i.e. the diagonals run ACROSS lanes.  The skew is done by the LDS
// write address, the reduction by the matrix core:
//   * lane (row rho) stores element (rho, dd) as bf16 at Sk[c][rho], c = dd - rho + 31 in [0, 95): every (c, rho) has
//     one writer, the two corner triangles are never written and stay zero from the one-time clear;
//   * column sums of Sk over rho = ones[16 x 32] . Sk^T: six v_mfma_f32_16x16x32_bf16 (one per 16 diagonals) with the
//     32 rows as the K dimension -- every output row holds the sums, lane l reads diagonal 16 cb + (l & 15);
//   * the window slides 64 diagonals per tile: blocks 0-3 are final after a tile (one coalesced 256-B store to the
//     wave's private partial row), blocks 4-5 are fed back as the C operand of blocks 0-1 of the next tile.
// (History: LDS float atomics cost ~550 cycles per wave-instruction; a register sliding window pulled with
// ds_bpermute cost ~22 VALU ops per element.  This form costs 1.5 LDS ops per element and no VALU.)
// K^T for dQ = dS K is not staged separately: the A operand is read from the K tile itself with the transposing
// ds_read_b64_tr_b16 (lane i of a 16-lane group addresses key (i >> 2), head-dim columns 4 (i & 3) .. +4).
// TAB: relative-position table present (a uniform `if (p.tab)` per element pair inside the score loop cost one basic
// block, its waits and hazard nops per pair: 34 branches and 92 s_nop per tile iteration)
template <bool DROP, bool TAB>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(FaP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // per stage: K [kv][hd] 8 KB | V [kv][hd] 8 KB;  then the four waves' skew buffers (96 x 64 B each)
  auto kbuf = [&](int st) { return smem + st * 16384; };
  auto vbuf = [&](int st) { return smem + st * 16384 + 8192; };
  float* tabs = reinterpret_cast<float*>(smem + 32768 + 4 * 6144);
  float* kb = tabs + p.Ltab;
  unsigned* colw = reinterpret_cast<unsigned*>(kb + p.Tkb);
  // d(rel) row of this block: the four waves' diagonal sums are added here (LDS float adds) and leave as ONE row of
  // 2T - 1 floats -- per-wave rows in HBM cost 4x the partial traffic, a memset, and a 4x longer reduction
  float* drow = reinterpret_cast<float*>(colw + (p.Tkb >> 1));
  const int T = p.T, H = p.H;
  int qblk, bh;
  fa_block_map(p.nqb, p.B * H, qblk, bh);
  const int b = bh / H, h = bh % H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int hi = lane >> 5, ql = lane & 31;
  const int i = qblk * FA_BQ + 32 * wave + ql;
  const int ic = i < T ? i : T - 1;
  const bool valid_i = i < T;
  unsigned char* skew = smem + 32768 + wave_u * 6144;
  const int ib = qblk * FA_BQ + 32 * wave_u;
  const int dlo0 = -ib - 31 + T - 1;  // diagonal of skew row 0 at tile 0 (negative for rows past the table)
  if constexpr (TAB)
    for (int d = threadIdx.x; d < 2 * T - 1; d += 256) drow[d] = 0.f;  // ordered before the first add by the tile loop's barriers
  const long D3 = 3L * H * FA_HD, D = (long)H * FA_HD;
  const bf16_t* base = p.qkv + (long)b * T * D3 + h * FA_HD;
  const int L = 2 * T - 1;

  U4 qf[4], dof[4];
  float dl = 0.f;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    qf[kk].v = *reinterpret_cast<const uint4*>(base + (long)ic * D3 + 16 * kk + 8 * hi);
    const long oo = ((long)b * T + ic) * D + h * FA_HD + 16 * kk + 8 * hi;
    dof[kk].v = *reinterpret_cast<const uint4*>(p.dO + oo);
    U4 ov; ov.v = *reinterpret_cast<const uint4*>(p.O + oo);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      dl += __uint_as_float(dof[kk].u[e] << 16) * __uint_as_float(ov.u[e] << 16);
      dl += __uint_as_float(dof[kk].u[e] & 0xffff0000u) * __uint_as_float(ov.u[e] & 0xffff0000u);
    }
  }
  dl += __shfl_xor(dl, 32, 64);
  const float dls = dl * p.inv_sc;
  for (int d = threadIdx.x; d < p.Ltab; d += 256) tabs[d] = (p.tab && d < L) ? p.tab[(long)h * L + d] : 0.f;
  for (int j = threadIdx.x; j < p.Tkb; j += 256)
    kb[j] = (j < T && !(p.kpm && p.kpm[(long)b * T + j])) ? 0.f : -INFINITY;
  if constexpr (DROP)
    for (int jp = threadIdx.x; jp < (p.Tkb >> 1); jp += 256) colw[jp] = fa_col_word(p.s1, (unsigned)jp);
  for (int q = threadIdx.x; q < 4 * 6144 / 16; q += 256) reinterpret_cast<uint4*>(smem + 32768)[q] = make_uint4(0, 0, 0, 0);
  const float g = p.gate ? p.gate[(long)bh * T + ic] : 0.f;
  const float g2 = g * FA_LOG2E;
  const float lse2 = valid_i ? p.lse[(long)bh * T + ic] * FA_LOG2E - p.log2sc : INFINITY;  // P * sc = 2^(x - lse2)
  const unsigned roww = fa_row_word(p.s0, (unsigned)(bh * T + ic));
  const float* trow = tabs + (T - 1 - ic);
  // skew write base of this lane: row (31 - rho + 4 hi), column rho (bf16)
  unsigned short* sk_w = reinterpret_cast<unsigned short*>(skew + (31 - ql + 4 * hi) * 64 + 2 * ql);
  // B-operand read of diagonal block cb: row 16 cb + (l & 15), 16 bytes at (l >> 4) * 16
  const unsigned char* sk_r = skew + (lane & 15) * 64 + (lane >> 4) * 16;
  // transposed A-operand read base into a K tile (see the header comment)
  const unsigned ktr = fa_tr_base(lane);

  f32x16_t dq[2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[f][r] = 0.f;
  f32x4_t dacc[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) dacc[c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  U4 ones; ones.v = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
  float dg = 0.f;
  const int nkv = (T + FA_BKV - 1) / FA_BKV;

  glds_tile64(base + D, D3, 0, T, kbuf(0), wave_u);
  glds_tile64(base + 2 * D, D3, 0, T, vbuf(0), wave_u);
  __syncthreads();

  int cur = 0;
  for (int jt = 0; jt < nkv; ++jt) {
    const int j0 = jt * FA_BKV;
    const bool more = jt + 1 < nkv;
    if (more) {
      glds_tile64(base + D, D3, j0 + FA_BKV, T, kbuf(cur ^ 1), wave_u);
      glds_tile64(base + 2 * D, D3, j0 + FA_BKV, T, vbuf(cur ^ 1), wave_u);
    }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      // one 32-key block at a time keeps only one (S, dP) accumulator pair live (computing both pairs up front, so that
      // block 0's element pass runs under block 1's MFMAs, measured no gain: the pass is VALU-bound either way)
      U4 dsf[2];
      f32x16_t s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain(kbuf(cur), 32 * f + ql, kk, hi), qf[kk].b, s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain(vbuf(cur), 32 * f + ql, kk, hi), dof[kk].b, dp, 0, 0, 0);
      }
      auto elem_pass = [&](auto edge_c) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_c)::value;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          float dv[2];
          unsigned w = 0;
          if constexpr (DROP) w = fa_mix(roww + colw[(j0 + 32 * f + (r & 3) + 8 * (r >> 2) + 4 * hi) >> 1]);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int rr = r + e;
            const int j = j0 + 32 * f + (rr & 3) + 8 * (rr >> 2) + 4 * hi;
            const float tv = trow[j];
            float x = fmaf(s[rr], p.sc2, g2 * tv);
            if constexpr (EDGE) x += kb[j];
            const float pe = __builtin_amdgcn_exp2f(x - lse2);  // 0 for masked keys (-inf) and rows past T (lse = +inf)
            float dpe = dp[rr];
            if constexpr (DROP) dpe = (e ? fa_keep_hi(w, p.ths) : fa_keep_lo(w, p.ths)) ? dpe : 0.f;
            const float ds = pe * (dpe - dls);
            dv[e] = ds;
            dg = fmaf(ds, tv, dg);
          }
          dsf[r >> 3].u[(r & 7) >> 1] = pack_bf16(dv[0], dv[1]);
          if constexpr (TAB) {
            const unsigned u2 = pack_bf16(g * dv[0], g * dv[1]);
            const int dd = 32 * f + (r & 3) + 8 * (r >> 2);  // + 4 hi is in sk_w
            sk_w[dd * 32] = (unsigned short)u2;
            sk_w[(dd + 1) * 32] = (unsigned short)(u2 >> 16);
          }
        }
      };
      if ((p.kpm != nullptr) || (j0 + FA_BKV > T)) elem_pass(std::true_type{}); else elem_pass(std::false_type{});
      // dQ^T += K^T dS^T (this 32-key block); K^T fragments by transposing reads of the K tile
#pragma unroll
      for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          dq[f2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(kbuf(cur), ktr, f2, f, s2), dsf[s2].b, dq[f2], 0, 0, 0);
        }
    }
    if constexpr (TAB) {
      // diagonal sums of this tile; blocks 4, 5 of the previous tile carry into blocks 0, 1
      f32x4_t nacc[6];
#pragma unroll
      for (int cb = 0; cb < 6; ++cb) {
        U4 bfr; bfr.v = *reinterpret_cast<const uint4*>(sk_r + cb * 1024);
        const f32x4_t cin = cb < 2 ? dacc[cb + 4] : f32x4_t{0.f, 0.f, 0.f, 0.f};
        nacc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones.b, bfr.b, cin, 0, 0, 0);
      }
#pragma unroll
      for (int cb = 0; cb < 6; ++cb) dacc[cb] = nacc[cb];
      const int gsel = lane >> 4;
      const float v = gsel == 0 ? dacc[0][0] : gsel == 1 ? dacc[1][0] : gsel == 2 ? dacc[2][0] : dacc[3][0];
      const int d = dlo0 + j0 + lane;
      if (d >= 0 && d < L) unsafeAtomicAdd(drow + d, v);
    }
    __syncthreads();
    cur ^= 1;
  }
  if (TAB && lane < 32) {
    // the 31 diagonals past the last tile's first 64 (blocks 4, 5)
    const float v = lane < 16 ? dacc[4][0] : dacc[5][0];
    const int d = dlo0 + 64 * nkv + lane;
    if (d >= 0 && d < L) unsafeAtomicAdd(drow + d, v);
  }
  if constexpr (TAB) {
    __syncthreads();
    float* prow = p.dtab_part + ((long)bh * p.nqb + qblk) * L;
    for (int d = threadIdx.x; d < L; d += 256) prow[d] = drow[d];
  }
  dg += __shfl_xor(dg, 32, 64);
  if (valid_i) {
    bf16_t* dst = p.dqkv + ((long)b * T + i) * D3 + h * FA_HD;
#pragma unroll
    for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        uint2 w;
        w.x = pack_bf16(dq[f2][4 * q4] * p.scale, dq[f2][4 * q4 + 1] * p.scale);
        w.y = pack_bf16(dq[f2][4 * q4 + 2] * p.scale, dq[f2][4 * q4 + 3] * p.scale);
        *reinterpret_cast<uint2*>(dst + 32 * f2 + 8 * q4 + 4 * hi) = w;
      }
    if (hi == 0) {
      p.delta[(long)bh * T + i] = dl;
      if (p.dgate) p.dgate[(long)bh * T + i] = dg;
    }
  }
}

// ------------------------------------------------------------------------------------ backward 2/2: dK, dV
// Lane owns a KEY column; scores are in the untransposed layout S[q][kv] so that the query contraction of
// dV^T = dO^T P and dK^T = Q^T dS finds its k-slots in the lane's registers.
template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(FaP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // per stage: Q [q][hd] | dO [q][hd], 8 KB each; the transposed operands of the query contraction are read from
  // the same tiles with ds_read_b64_tr_b16 (frag_tr)
  auto qbuf = [&](int st) { return smem + st * 16384; };
  auto dobuf = [&](int st) { return smem + st * 16384 + 8192; };
  float* tabs = reinterpret_cast<float*>(smem + 32768);
  float* rowv = tabs + p.Ltab + 64;  // [2 stages][4][64]: lse * log2e, delta, gate * log2e, dropout row word of the query tile
  const int T = p.T, H = p.H;
  int kblk, bh;
  fa_block_map(p.nqb, p.B * H, kblk, bh);
  const int b = bh / H, h = bh % H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int hi = lane >> 5, kl = lane & 31;
  const int j = kblk * FA_BK1 + 32 * wave + kl;
  const int jc = j < T ? j : T - 1;
  const long D3 = 3L * H * FA_HD, D = (long)H * FA_HD;
  const bf16_t* base = p.qkv + (long)b * T * D3 + h * FA_HD;
  const bf16_t* dobase = p.dO + (long)b * T * D + h * FA_HD;
  const int L = 2 * T - 1;

  U4 kf[4], vf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    kf[kk].v = *reinterpret_cast<const uint4*>(base + D + (long)jc * D3 + 16 * kk + 8 * hi);
    vf[kk].v = *reinterpret_cast<const uint4*>(base + 2 * D + (long)jc * D3 + 16 * kk + 8 * hi);
  }
  for (int d = threadIdx.x; d < p.Ltab + 64; d += 256) tabs[d] = (p.tab && d >= 64 && d - 64 < L) ? p.tab[(long)h * L + d - 64] : 0.f;
  const bool key_ok = j < T && !(p.kpm && p.kpm[(long)b * T + jc]);
  const unsigned cw = fa_col_word(p.s1, (unsigned)(jc >> 1));
  const unsigned csh = (jc & 1) << 4;
  const float* tcol = tabs + 64 + (jc + T - 1);  // tcol[-i] = rel[h, j - i]; 64 zero floats in front absorb rows past T
  const float kadd = key_ok ? 0.f : -INFINITY;

  f32x16_t dk[2], dv[2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[f][r] = 0.f; dv[f][r] = 0.f; }
  const int nq = (T + FA_BQ1 - 1) / FA_BQ1;
  const unsigned qtr = fa_tr_base(lane);

  auto stage_rows = [&](int it, int st) {
    const int t = threadIdx.x;
    if (t < 64) {
      const int ii = it * FA_BQ1 + t;
      const bool ok = ii < T;
      const long o = (long)bh * T + (ok ? ii : T - 1);
      rowv[st * 256 + t] = ok ? p.lse[o] * FA_LOG2E - p.log2sc : INFINITY;  // P * sc = 2^(x - this)
      rowv[st * 256 + 64 + t] = p.delta[o] * p.inv_sc;
      rowv[st * 256 + 128 + t] = p.gate ? p.gate[o] * FA_LOG2E : 0.f;
      rowv[st * 256 + 192 + t] = __uint_as_float(fa_row_word(p.s0, (unsigned)o));
    }
  };

  glds_tile64(base, D3, 0, T, qbuf(0), wave_u);
  glds_tile64(dobase, D, 0, T, dobuf(0), wave_u);
  stage_rows(0, 0);
  __syncthreads();

  int cur = 0;
  for (int it = 0; it < nq; ++it) {
    const int iq0 = it * FA_BQ1;
    const bool more = it + 1 < nq;
    if (more) {
      glds_tile64(base, D3, iq0 + FA_BQ1, T, qbuf(cur ^ 1), wave_u);
      glds_tile64(dobase, D, iq0 + FA_BQ1, T, dobuf(cur ^ 1), wave_u);
      stage_rows(it + 1, cur ^ 1);
    }
    const float* rv = rowv + cur * 256;
    // S = Q K^T, dP = dO V^T  (rows = queries of the tile, col = this lane's key)
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      U4 pf[2], dsf[2];
      f32x16_t s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain(qbuf(cur), 32 * f + kl, kk, hi), kf[kk].b, s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain(dobuf(cur), 32 * f + kl, kk, hi), vf[kk].b, dp, 0, 0, 0);
      }
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        // registers 4 q4 .. 4 q4 + 3 of a block are four consecutive query rows: per-row scalars come as 16-byte
        // LDS vectors; the Toeplitz entries rel[j - i] run downwards in i
        const int il0 = 32 * f + 8 * q4 + 4 * hi;
        const float4 lse4 = *reinterpret_cast<const float4*>(rv + il0);
        const float4 del4 = *reinterpret_cast<const float4*>(rv + 64 + il0);
        const float4 gat4 = *reinterpret_cast<const float4*>(rv + 128 + il0);
        const uint4 row4 = *reinterpret_cast<const uint4*>(rv + 192 + il0);
        const float lsev[4] = {lse4.x, lse4.y, lse4.z, lse4.w}, delv[4] = {del4.x, del4.y, del4.z, del4.w};
        const float gatv[4] = {gat4.x, gat4.y, gat4.z, gat4.w};
        const unsigned roww[4] = {row4.x, row4.y, row4.z, row4.w};
        const float* tq = tcol - (iq0 + il0);
        float pv[4], dsv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int rr = 4 * q4 + e;
          const float x = fmaf(s[rr], p.sc2, gatv[e] * tq[-e]) + kadd;   // kadd = -inf for a padded / out-of-range key
          const float pe = __builtin_amdgcn_exp2f(x - lsev[e]);           // rows past T: lse = +inf -> 0
          float dpe = dp[rr];
          float pd = pe;
          if constexpr (DROP) {
            const unsigned w = fa_mix(roww[e] + cw);
            const bool kp = (int)(short)((w >> csh) & 0xffffu) >= p.ths;
            pd = kp ? pe : 0.f;
            dpe = kp ? dpe : 0.f;
          }
          pv[e] = pd;
          dsv[e] = pe * (dpe - delv[e]);
        }
        pf[q4 >> 1].u[2 * (q4 & 1)] = pack_bf16(pv[0], pv[1]);
        pf[q4 >> 1].u[2 * (q4 & 1) + 1] = pack_bf16(pv[2], pv[3]);
        dsf[q4 >> 1].u[2 * (q4 & 1)] = pack_bf16(dsv[0], dsv[1]);
        dsf[q4 >> 1].u[2 * (q4 & 1) + 1] = pack_bf16(dsv[2], dsv[3]);
      }
      // dV^T += dO^T P ; dK^T += Q^T dS   (contraction over this 32-query block)
#pragma unroll
      for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          dv[f2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(dobuf(cur), qtr, f2, f, s2), pf[s2].b, dv[f2], 0, 0, 0);
          dk[f2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(qbuf(cur), qtr, f2, f, s2), dsf[s2].b, dk[f2], 0, 0, 0);
        }
    }
    __syncthreads();
    cur ^= 1;
  }
  if (j < T) {
    bf16_t* dst = p.dqkv + ((long)b * T + j) * D3 + h * FA_HD;
#pragma unroll
    for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        uint2 w;
        w.x = pack_bf16(dk[f2][4 * q4] * p.scale, dk[f2][4 * q4 + 1] * p.scale);
        w.y = pack_bf16(dk[f2][4 * q4 + 2] * p.scale, dk[f2][4 * q4 + 3] * p.scale);
        *reinterpret_cast<uint2*>(dst + D + 32 * f2 + 8 * q4 + 4 * hi) = w;
        w.x = pack_bf16(dv[f2][4 * q4], dv[f2][4 * q4 + 1]);
        w.y = pack_bf16(dv[f2][4 * q4 + 2], dv[f2][4 * q4 + 3]);
        *reinterpret_cast<uint2*>(dst + 2 * D + 32 * f2 + 8 * q4 + 4 * hi) = w;
      }
  }
}

// drel[h][d] = sum over (b, q-tile) partial rows: block = 64 d x 16 row slices
__global__ __launch_bounds__(1024) void fa_dtab_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                              int B, int H, int nchunk, int L) {
  __shared__ float red[16][64];
  const int col = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int d = blockIdx.x * 64 + col;
  const int h = blockIdx.y;
  float s = 0.f;
  if (d < L)
    for (int r = slice; r < B * nchunk; r += 16) {
      const int b = r / nchunk, c = r - b * nchunk;
      s += part[(((long)b * H + h) * nchunk + c) * L + d];
    }
  red[slice][col] = s;
  __syncthreads();
  if (slice == 0 && d < L) {
    s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][col];
    out[(long)h * L + d] = s;
  }
}

static FaP fa_params(int B, int H, int T, float scale, float p_drop, uint64_t seed) {
  FaP p;
  p.B = B; p.H = H; p.T = T; p.scale = scale;
  p.sc2 = scale * FA_LOG2E;
  double tt = (double)p_drop * 65536.0 + 0.5; if (tt > 65535.0) tt = 65535.0;
  p.th = p_drop > 0.f ? (unsigned)tt : 0u;
  if (p_drop > 0.f && p.th == 0u) p.th = 1u;
  p.sc = p.th ? (float)(1.0 / (1.0 - (double)p.th / 65536.0)) : 1.f;  // unbiased for the quantised probability
  p.log2sc = log2f(p.sc); p.inv_sc = 1.f / p.sc;
  p.ths = (int)p.th - 32768;
  p.k2 = (unsigned)((p.ths - 1) & 0xffff) * 0x10001u;
  p.s0 = (unsigned)seed; p.s1 = (unsigned)(seed >> 32);
  const int nkv = (T + FA_BKV - 1) / FA_BKV;
  p.Tkb = nkv * FA_BKV;
  p.Ltab = (T + p.Tkb + 3) & ~3;  // index j - i + T - 1 with j < Tkb, i >= 0
  p.qkv = nullptr; p.O = nullptr; p.lse = nullptr; p.gate = nullptr; p.tab = nullptr; p.kpm = nullptr;
  p.dO = nullptr; p.dqkv = nullptr; p.delta = nullptr; p.dgate = nullptr; p.dtab_part = nullptr;
  return p;
}

template <typename K> static int fa_set_smem(K kernel, size_t bytes) {
  if (bytes <= 65536) return WL_OK;
  return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess
             ? WL_OK : WL_ELAUNCH;
}

extern "C" {

// O[B,T,H*64] = softmax(scale QK^T + gate*rel + keypad) V from packed qkv [B,T,3*H*64] (bf16); lse[B*H,T] saved
int wavlm_attn_fused_fwd(const void* qkv, void* O, float* lse, const float* gate, const float* tab, const uint8_t* kpm,
                         int32_t B, int32_t H, int32_t T, int32_t head_dim, float scale, float p_drop, uint64_t seed,
                         void* stream) {
  if (!qkv || !O || !lse || B <= 0 || H <= 0 || T <= 0 || head_dim != FA_HD) return WL_EINVAL;
  if ((gate == nullptr) != (tab == nullptr)) return WL_EINVAL;
  FaP p = fa_params(B, H, T, scale, p_drop, seed);
  p.qkv = (const bf16_t*)qkv; p.O = (bf16_t*)O; p.lse = lse; p.gate = gate; p.tab = tab; p.kpm = kpm;
  const size_t smem = 32768 + (size_t)(p.Ltab + p.Tkb + p.Tkb / 2) * sizeof(float);
  p.nqb = (T + FA_BQ - 1) / FA_BQ;
  const dim3 grid((unsigned)(p.nqb * B * H));
  if (p.th) {
    if (fa_set_smem(attn_fwd_kernel<true>, smem) != WL_OK) return WL_ELAUNCH;
    WL_LAUNCH(attn_fwd_kernel<true>, grid, dim3(256), smem, (hipStream_t)stream, p);
  } else {
    if (fa_set_smem(attn_fwd_kernel<false>, smem) != WL_OK) return WL_ELAUNCH;
    WL_LAUNCH(attn_fwd_kernel<false>, grid, dim3(256), smem, (hipStream_t)stream, p);
  }
  return wl_check_launch();
}

uint64_t wavlm_attn_fused_bwd_workspace_bytes(int32_t B, int32_t H, int32_t T) {
  const uint64_t nqt = (uint64_t)((T + FA_BQ - 1) / FA_BQ);
  return ((uint64_t)B * H * nqt * (2 * (uint64_t)T - 1) + (uint64_t)B * H * T) * sizeof(float);
}

// dqkv[B,T,3*H*64], dgate[B,H,T], dtab[H,2T-1] from dO and the forward's (qkv, O, lse)
int wavlm_attn_fused_bwd(const void* qkv, const void* O, const void* dO, const float* lse, const float* gate,
                         const float* tab, const uint8_t* kpm, void* dqkv, float* dgate, float* dtab, int32_t B,
                         int32_t H, int32_t T, int32_t head_dim, float scale, float p_drop, uint64_t seed,
                         void* workspace, uint64_t ws_bytes, void* stream) {
  if (!qkv || !O || !dO || !lse || !dqkv || !workspace || B <= 0 || H <= 0 || T <= 0 || head_dim != FA_HD)
    return WL_EINVAL;
  if ((gate == nullptr) != (tab == nullptr)) return WL_EINVAL;
  if (tab && (!dgate || !dtab)) return WL_EINVAL;
  if (ws_bytes < wavlm_attn_fused_bwd_workspace_bytes(B, H, T)) return WL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  FaP p = fa_params(B, H, T, scale, p_drop, seed);
  p.qkv = (const bf16_t*)qkv; p.O = (bf16_t*)const_cast<void*>(O); p.lse = const_cast<float*>(lse);
  p.gate = gate; p.tab = tab; p.kpm = kpm; p.dO = (const bf16_t*)dO; p.dqkv = (bf16_t*)dqkv; p.dgate = dgate;
  const int nqt = (T + FA_BQ - 1) / FA_BQ;
  const int L = 2 * T - 1;
  p.dtab_part = (float*)workspace;
  p.delta = p.dtab_part + (long)B * H * nqt * L;
  const size_t smem1 = 32768 + 4 * 6144 + (size_t)(2 * p.Ltab + p.Tkb + p.Tkb / 2) * sizeof(float);
  p.nqb = nqt;
  if (p.th) {
#define FA_DQ(DR, TB) do { if (fa_set_smem(attn_bwd_dq_kernel<DR, TB>, smem1) != WL_OK) return WL_ELAUNCH; \
    WL_LAUNCH((attn_bwd_dq_kernel<DR, TB>), dim3((unsigned)(nqt * B * H)), dim3(256), smem1, st, p); } while (0)
    if (tab) FA_DQ(true, true); else FA_DQ(true, false);
  } else {
    if (tab) FA_DQ(false, true); else FA_DQ(false, false);
  }
#undef FA_DQ
  const size_t smem2 = 32768 + (size_t)(p.Ltab + 64 + 2 * 256) * sizeof(float);
  p.nqb = (T + FA_BK1 - 1) / FA_BK1;
  const dim3 grid2((unsigned)(p.nqb * B * H));
  if (p.th) {
    if (fa_set_smem(attn_bwd_dkv_kernel<true>, smem2) != WL_OK) return WL_ELAUNCH;
    WL_LAUNCH(attn_bwd_dkv_kernel<true>, grid2, dim3(256), smem2, st, p);
  } else {
    if (fa_set_smem(attn_bwd_dkv_kernel<false>, smem2) != WL_OK) return WL_ELAUNCH;
    WL_LAUNCH(attn_bwd_dkv_kernel<false>, grid2, dim3(256), smem2, st, p);
  }
  if (tab)
    WL_LAUNCH(fa_dtab_reduce_kernel, dim3((unsigned)((L + 63) / 64), (unsigned)H), dim3(1024), 0, st,
              (const float*)p.dtab_part, dtab, (int)B, (int)H, nqt, L);
  return wl_check_launch();
}

}  // extern "C"
