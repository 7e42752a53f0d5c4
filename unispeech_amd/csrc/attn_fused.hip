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

#define FA_HD 64
#define FA_BQ 128   // query rows per block (forward, dQ kernel): 4 waves x 32
#define FA_BKV 64   // keys per iteration
#define FA_BK1 128  // key rows per block (dK/dV kernel): 4 waves x 32
#define FA_BQ1 64   // query rows per iteration (dK/dV kernel)

__device__ __forceinline__ bool fa_keep(unsigned s0, unsigned s1, unsigned idx, unsigned thresh) {
  unsigned x = (idx ^ s0) * 0x9E3779B1u;
  x ^= x >> 15; x = (x + s1) * 0x85EBCA77u;
  x ^= x >> 13; x *= 0xC2B2AE3Du;
  x ^= x >> 16;
  return x >= thresh;
}

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

struct FaP {
  const bf16_t* qkv; bf16_t* O; float* lse;
  const float* gate; const float* tab; const unsigned char* kpm;
  const bf16_t* dO; bf16_t* dqkv; float* delta; float* dgate; float* dtab_part;
  int B, H, T; float scale; unsigned th; float sc; unsigned s0, s1;
  int Ltab, Tkb;  // LDS extents: rel table (zero padded) and key bias
};

// ------------------------------------------------------------------------------------------------- forward
__global__ __launch_bounds__(256) void attn_fwd_kernel(FaP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  auto kbuf = [&](int st) { return smem + st * 16384; };
  auto vbuf = [&](int st) { return smem + st * 16384 + 8192; };
  float* tabs = reinterpret_cast<float*>(smem + 32768);
  float* kb = tabs + p.Ltab;
  const int T = p.T, H = p.H;
  const int bh = blockIdx.y, b = bh / H, h = bh % H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int hi = lane >> 5, ql = lane & 31;
  const int i = blockIdx.x * FA_BQ + 32 * wave + ql;
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
  const float g = p.gate ? p.gate[(long)bh * T + ic] : 0.f;

  f32x16_t o[2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[f][r] = 0.f;
  float m = -INFINITY, l = 0.f;
  const int nkv = (T + FA_BKV - 1) / FA_BKV;
  uint4 vr[4];

  glds_tile64(base + FA_HD * H, D3, 0, T, kbuf(0), wave_u);
  load_ks<64>(base + 2 * FA_HD * H, D3, 64, T, vr);
  store_ks<64>(vbuf(0), vr);
  __syncthreads();

  int cur = 0;
  for (int jt = 0; jt < nkv; ++jt) {
    const int j0 = jt * FA_BKV;
    const bool more = jt + 1 < nkv;
    if (more) {
      glds_tile64(base + FA_HD * H, D3, j0 + FA_BKV, T, kbuf(cur ^ 1), wave_u);
      load_ks<64>(base + 2 * FA_HD * H + (long)(j0 + FA_BKV) * D3, D3, 64, T - j0 - FA_BKV, vr);
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
    float tmax = -INFINITY;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = j0 + 32 * f + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float x = s[f][r] * p.scale + g * tabs[j - ic + T - 1] + kb[j];
        s[f][r] = x;
        tmax = fmaxf(tmax, x);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m, tmax);
    const bool dead = (m_new == -INFINITY);
    const float alpha = dead ? 1.f : __expf(m - m_new);
    float rs = 0.f;
    U4 pf[2][2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float pv[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int rr = r + e;
          float pe = dead ? 0.f : __expf(s[f][rr] - m_new);
          rs += pe;
          if (p.th) {
            const int j = j0 + 32 * f + (rr & 3) + 8 * (rr >> 2) + 4 * hi;
            pe = fa_keep(p.s0, p.s1, (unsigned)(((long)bh * T + ic) * T + j), p.th) ? pe * p.sc : 0.f;
          }
          pv[e] = pe;
        }
        pf[f][r >> 3].u[(r & 7) >> 1] = pack_bf16(pv[0], pv[1]);
      }
    rs += __shfl_xor(rs, 32, 64);
    l = l * alpha + rs;
    m = m_new;
#pragma unroll
    for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[f2][r] *= alpha;
    // O^T += V^T P^T
#pragma unroll
    for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
          o[f2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_perm(vbuf(cur), 32 * f2 + ql, f, s2, hi), pf[f][s2].b,
                                                          o[f2], 0, 0, 0);
    if (more) store_ks<64>(vbuf(cur ^ 1), vr);
    __syncthreads();
    cur ^= 1;
  }
  if (i < T) {
    const float inv = l > 0.f ? 1.f / l : 0.f;
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
    if (hi == 0) p.lse[(long)bh * T + i] = m + __logf(l);
  }
}

// ------------------------------------------------------------------------------- backward 1/2: dQ, dgate, drel
// Same decomposition as the forward (lane owns a query row).  Also writes delta[i] = <dO_i, O_i> for kernel 2/2.
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(FaP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // per stage: K [kv][hd] 8 KB | V [kv][hd] 8 KB | K^T [hd][kv] 8 KB
  auto kbuf = [&](int st) { return smem + st * 24576; };
  auto vbuf = [&](int st) { return smem + st * 24576 + 8192; };
  auto ktbuf = [&](int st) { return smem + st * 24576 + 16384; };
  float* tabs = reinterpret_cast<float*>(smem + 49152);
  float* kb = tabs + p.Ltab;
  const int T = p.T, H = p.H;
  const int bh = blockIdx.y, b = bh / H, h = bh % H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int hi = lane >> 5, ql = lane & 31;
  const int i = blockIdx.x * FA_BQ + 32 * wave + ql;
  const int ic = i < T ? i : T - 1;
  const bool valid_i = i < T;
  // drel accumulation without atomics (LDS float atomics measured ~550 cycles per wave-instruction here):
  // d = j - i + T - 1 is Toeplitz, so while the wave walks the key tiles its 32 rows touch a window of 95
  // consecutive d that slides by 64 per tile.  The window lives in registers, one circular 128-entry buffer
  // per wave: entry e = d & 127 is owned by lane e & 63, register e >> 6.  Each owner PULLS its contributions
  // with ds_bpermute (for a fixed element slot the 32 source rows map to 32 consecutive entries, so a lane has
  // exactly one source or none); after a tile the 64 entries that can no longer be touched go to the wave's
  // private row of the partial buffer with a plain store.
  const int ib = blockIdx.x * FA_BQ + 32 * wave_u;
  const int dlo0 = -ib - 31 + T - 1;               // first d of tile 0 (may be negative for rows past the table)
  const int r0 = (lane - dlo0) & 127;              // window-relative position of entry `lane` at tile 0
  float W0 = 0.f, W1 = 0.f;
  float* wpart = p.tab ? p.dtab_part + (((long)bh * gridDim.x + blockIdx.x) * 4 + wave_u) * (2 * T - 1) : nullptr;
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
  for (int d = threadIdx.x; d < p.Ltab; d += 256) tabs[d] = (p.tab && d < L) ? p.tab[(long)h * L + d] : 0.f;
  for (int j = threadIdx.x; j < p.Tkb; j += 256)
    kb[j] = (j < T && !(p.kpm && p.kpm[(long)b * T + j])) ? 0.f : -INFINITY;
  const float g = p.gate ? p.gate[(long)bh * T + ic] : 0.f;
  const float lse_i = valid_i ? p.lse[(long)bh * T + ic] : INFINITY;

  f32x16_t dq[2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[f][r] = 0.f;
  float dg = 0.f;
  const int nkv = (T + FA_BKV - 1) / FA_BKV;
  uint4 kr[4];

  glds_tile64(base + D, D3, 0, T, kbuf(0), wave_u);
  glds_tile64(base + 2 * D, D3, 0, T, vbuf(0), wave_u);
  load_ks<64>(base + D, D3, 64, T, kr);
  store_ks<64>(ktbuf(0), kr);
  __syncthreads();

  int cur = 0;
  for (int jt = 0; jt < nkv; ++jt) {
    const int j0 = jt * FA_BKV;
    const bool more = jt + 1 < nkv;
    if (more) {
      glds_tile64(base + D, D3, j0 + FA_BKV, T, kbuf(cur ^ 1), wave_u);
      glds_tile64(base + 2 * D, D3, j0 + FA_BKV, T, vbuf(cur ^ 1), wave_u);
      load_ks<64>(base + D + (long)(j0 + FA_BKV) * D3, D3, 64, T - j0 - FA_BKV, kr);
    }
    const int rel0 = r0 ^ ((jt & 1) << 6);         // window position of entry `lane` (W0); W1 is rel0 ^ 64
    const int A0 = 31 - rel0, A1 = 31 - (rel0 ^ 64);
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      // one 32-key block at a time keeps only one (S, dP) accumulator pair live
      U4 dsf[2];
      f32x16_t s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain(kbuf(cur), 32 * f + ql, kk, hi), qf[kk].b, s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain(vbuf(cur), 32 * f + ql, kk, hi), dof[kk].b, dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float dv[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int rr = r + e;
          const int j = j0 + 32 * f + (rr & 3) + 8 * (rr >> 2) + 4 * hi;
          const float tv = tabs[j - ic + T - 1];
          const float x = s[rr] * p.scale + g * tv + kb[j];
          const float pe = __expf(x - lse_i);  // 0 for masked keys (x = -inf) and for rows past T (lse = +inf)
          float dpe = dp[rr];
          if (p.th) dpe = fa_keep(p.s0, p.s1, (unsigned)(((long)bh * T + ic) * T + j), p.th) ? dpe * p.sc : 0.f;
          const float ds = pe * (dpe - dl);
          dv[e] = ds;
          if (p.tab) {
            dg = fmaf(ds, tv, dg);
            const float u = g * ds;
            // element slot offset within the tile; the source row for window position rel is 31 + dd - rel
#pragma unroll
            for (int hs = 0; hs < 2; ++hs) {
              const int dd = 32 * f + (rr & 3) + 8 * (rr >> 2) + 4 * hs;
              const int q0 = A0 + dd, q1 = A1 + dd;
              const bool v0 = (unsigned)q0 < 32u, v1 = (unsigned)q1 < 32u;
              const int src = (v0 ? q0 : q1) + 32 * hs;
              const float got = __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(u)));
              W0 += v0 ? got : 0.f;
              W1 += v1 ? got : 0.f;
            }
          }
        }
        dsf[r >> 3].u[(r & 7) >> 1] = pack_bf16(dv[0], dv[1]);
      }
      // dQ^T += K^T dS^T (this 32-key block)
#pragma unroll
      for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
          dq[f2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_perm(ktbuf(cur), 32 * f2 + ql, f, s2, hi), dsf[s2].b,
                                                           dq[f2], 0, 0, 0);
    }
    if (p.tab) {
      // entries at window positions [0, 64) are final: d = dlo + rel
      const bool first = rel0 < 64;
      const int d = dlo0 + 64 * jt + (first ? rel0 : (rel0 ^ 64));
      if (d >= 0 && d < L) wpart[d] = first ? W0 : W1;
      if (first) W0 = 0.f; else W1 = 0.f;
    }
    if (more) store_ks<64>(ktbuf(cur ^ 1), kr);
    __syncthreads();
    cur ^= 1;
  }
  if (p.tab) {
    // what is left sits at window positions [0, 64) of the tile after the last one
    const int relN = r0 ^ ((nkv & 1) << 6);
    const bool first = relN < 64;
    const int d = dlo0 + 64 * nkv + (first ? relN : (relN ^ 64));
    if (d >= 0 && d < L) wpart[d] = first ? W0 : W1;
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
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(FaP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // per stage: Q [q][hd] | dO [q][hd] | Q^T [hd][q] | dO^T [hd][q], 8 KB each
  auto qbuf = [&](int st) { return smem + st * 32768; };
  auto dobuf = [&](int st) { return smem + st * 32768 + 8192; };
  auto qtbuf = [&](int st) { return smem + st * 32768 + 16384; };
  auto dotbuf = [&](int st) { return smem + st * 32768 + 24576; };
  float* tabs = reinterpret_cast<float*>(smem + 65536);
  float* rowv = tabs + p.Ltab;  // [2 stages][3][64]: lse, delta, gate of the query tile
  const int T = p.T, H = p.H;
  const int bh = blockIdx.y, b = bh / H, h = bh % H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int hi = lane >> 5, kl = lane & 31;
  const int j = blockIdx.x * FA_BK1 + 32 * wave + kl;
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
  for (int d = threadIdx.x; d < p.Ltab; d += 256) tabs[d] = (p.tab && d < L) ? p.tab[(long)h * L + d] : 0.f;
  const bool key_ok = j < T && !(p.kpm && p.kpm[(long)b * T + jc]);

  f32x16_t dk[2], dv[2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[f][r] = 0.f; dv[f][r] = 0.f; }
  const int nq = (T + FA_BQ1 - 1) / FA_BQ1;
  uint4 qr[4], dor[4];

  auto stage_rows = [&](int it, int st) {
    const int t = threadIdx.x;
    if (t < 64) {
      const int ii = it * FA_BQ1 + t;
      const bool ok = ii < T;
      const long o = (long)bh * T + (ok ? ii : T - 1);
      rowv[st * 192 + t] = ok ? p.lse[o] : INFINITY;
      rowv[st * 192 + 64 + t] = p.delta[o];
      rowv[st * 192 + 128 + t] = p.gate ? p.gate[o] : 0.f;
    }
  };

  glds_tile64(base, D3, 0, T, qbuf(0), wave_u);
  glds_tile64(dobase, D, 0, T, dobuf(0), wave_u);
  load_ks<64>(base, D3, 64, T, qr);
  load_ks<64>(dobase, D, 64, T, dor);
  store_ks<64>(qtbuf(0), qr);
  store_ks<64>(dotbuf(0), dor);
  stage_rows(0, 0);
  __syncthreads();

  int cur = 0;
  for (int it = 0; it < nq; ++it) {
    const int iq0 = it * FA_BQ1;
    const bool more = it + 1 < nq;
    if (more) {
      glds_tile64(base, D3, iq0 + FA_BQ1, T, qbuf(cur ^ 1), wave_u);
      glds_tile64(dobase, D, iq0 + FA_BQ1, T, dobuf(cur ^ 1), wave_u);
      load_ks<64>(base + (long)(iq0 + FA_BQ1) * D3, D3, 64, T - iq0 - FA_BQ1, qr);
      load_ks<64>(dobase + (long)(iq0 + FA_BQ1) * D, D, 64, T - iq0 - FA_BQ1, dor);
      stage_rows(it + 1, cur ^ 1);
    }
    const float* rv = rowv + cur * 192;
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
      for (int r = 0; r < 16; r += 2) {
        float pv[2], dsv[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int rr = r + e;
          const int il = 32 * f + (rr & 3) + 8 * (rr >> 2) + 4 * hi;  // query row within the tile
          int ii = iq0 + il; if (ii > T - 1) ii = T - 1;
          const float x = s[rr] * p.scale + rv[128 + il] * tabs[jc - ii + T - 1];
          float pe = key_ok ? __expf(x - rv[il]) : 0.f;  // rows past T: lse = +inf -> 0
          float dpe = dp[rr];
          float pd = pe;
          if (p.th) {
            const bool kp = fa_keep(p.s0, p.s1, (unsigned)(((long)bh * T + ii) * T + jc), p.th);
            pd = kp ? pe * p.sc : 0.f;
            dpe = kp ? dpe * p.sc : 0.f;
          }
          pv[e] = pd;
          dsv[e] = pe * (dpe - rv[64 + il]);
        }
        pf[r >> 3].u[(r & 7) >> 1] = pack_bf16(pv[0], pv[1]);
        dsf[r >> 3].u[(r & 7) >> 1] = pack_bf16(dsv[0], dsv[1]);
      }
      // dV^T += dO^T P ; dK^T += Q^T dS   (contraction over this 32-query block)
#pragma unroll
      for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          dv[f2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_perm(dotbuf(cur), 32 * f2 + kl, f, s2, hi), pf[s2].b,
                                                           dv[f2], 0, 0, 0);
          dk[f2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_perm(qtbuf(cur), 32 * f2 + kl, f, s2, hi), dsf[s2].b,
                                                           dk[f2], 0, 0, 0);
        }
    }
    if (more) {
      store_ks<64>(qtbuf(cur ^ 1), qr);
      store_ks<64>(dotbuf(cur ^ 1), dor);
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

// drel[h][d] = sum over (b, q-tile, wave) partial rows: block = 64 d x 16 row slices
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
  double tt = (double)p_drop * 4294967296.0; if (tt > 4294967295.0) tt = 4294967295.0;
  p.th = p_drop > 0.f ? (unsigned)tt : 0u;
  p.sc = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
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
  const size_t smem = 32768 + (size_t)(p.Ltab + p.Tkb) * sizeof(float);
  if (fa_set_smem(attn_fwd_kernel, smem) != WL_OK) return WL_ELAUNCH;
  WL_LAUNCH(attn_fwd_kernel, dim3((unsigned)((T + FA_BQ - 1) / FA_BQ), (unsigned)(B * H)), dim3(256), smem,
            (hipStream_t)stream, p);
  return wl_check_launch();
}

uint64_t wavlm_attn_fused_bwd_workspace_bytes(int32_t B, int32_t H, int32_t T) {
  const uint64_t nqt = (uint64_t)((T + FA_BQ - 1) / FA_BQ);
  return ((uint64_t)B * H * nqt * 4 * (2 * (uint64_t)T - 1) + (uint64_t)B * H * T) * sizeof(float);
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
  p.delta = p.dtab_part + (long)B * H * nqt * 4 * L;
  // each wave stores every d of its own window range exactly once; everything else must read as zero
  if (tab && hipMemsetAsync(p.dtab_part, 0, (size_t)B * H * nqt * 4 * L * sizeof(float), st) != hipSuccess) return WL_ELAUNCH;
  const size_t smem1 = 49152 + (size_t)(p.Ltab + p.Tkb) * sizeof(float);
  if (fa_set_smem(attn_bwd_dq_kernel, smem1) != WL_OK) return WL_ELAUNCH;
  WL_LAUNCH(attn_bwd_dq_kernel, dim3((unsigned)nqt, (unsigned)(B * H)), dim3(256), smem1, st, p);
  const size_t smem2 = 65536 + (size_t)(p.Ltab + 2 * 192) * sizeof(float);
  if (fa_set_smem(attn_bwd_dkv_kernel, smem2) != WL_OK) return WL_ELAUNCH;
  WL_LAUNCH(attn_bwd_dkv_kernel, dim3((unsigned)((T + FA_BK1 - 1) / FA_BK1), (unsigned)(B * H)), dim3(256), smem2, st, p);
  if (tab)
    WL_LAUNCH(fa_dtab_reduce_kernel, dim3((unsigned)((L + 63) / 64), (unsigned)H), dim3(1024), 0, st,
              (const float*)p.dtab_part, dtab, (int)B, (int)H, nqt * 4, L);
  return wl_check_launch();
}

}  // extern "C"
