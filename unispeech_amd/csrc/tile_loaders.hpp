// Operand tile loaders shared by the MFMA kernels (GEMM, fused attention): HBM -> swizzled LDS image of a
// [rows][64 k] bf16 tile, and the fragment addressing of v_mfma_f32_32x32x16_bf16.  See gemm_bf16.hip for the
// design notes (K-contiguous: global_load_lds; K-strided: in-register 4x8 transpose).
#pragma once
#include "common.hpp"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define GEMM_BK 64

union U4 { uint4 v; bf16x8_t b; unsigned u[4]; };

__device__ __forceinline__ unsigned lds_off(int row, int chunk) {
  return (unsigned)row * 128u + (unsigned)((chunk ^ ((row >> 1) & 7)) << 4);
}

// ---- K-contiguous operand: tile [ROWS][64], global row stride ld --------------------------------
template <int ROWS, int NT = 256>
__device__ __forceinline__ void load_kc(const bf16_t* __restrict__ base, long ld, int rows_valid, int k_valid,
                                        uint4 (&r)[ROWS / (NT / 8)]) {
  const int t = threadIdx.x;
  const int row = t >> 3, kk = (t & 7) * 8;
#pragma unroll
  for (int ps = 0; ps < ROWS / (NT / 8); ++ps) {
    const int rr = row + ps * (NT / 8);
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
template <int ROWS, int NT = 256>
__device__ __forceinline__ void store_kc(unsigned char* lds, const uint4 (&r)[ROWS / (NT / 8)]) {
  const int t = threadIdx.x;
  const int row = t >> 3, ch = t & 7;
#pragma unroll
  for (int ps = 0; ps < ROWS / (NT / 8); ++ps) {
    const int rr = row + ps * (NT / 8);
    *reinterpret_cast<uint4*>(lds + lds_off(rr, ch)) = r[ps];
  }
}

// ---- K-strided operand: global tile [64 k][ROWS] (rows contiguous), stride ld between k ---------
template <int ROWS, int NT = 256>
__device__ __forceinline__ void load_ks(const bf16_t* __restrict__ base, long ld, int rows_valid, int k_valid,
                                        uint4 (&r)[4]) {
  static_assert(ROWS <= NT / 2, "one pass covers at most NT/2 rows");
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
template <int ROWS, int NT = 256>
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

typedef const __attribute__((address_space(1))) void* gas_ptr;
typedef __attribute__((address_space(3))) void* las_ptr;

// One operand of the block tile.  Loop-invariant per-thread state (pointers, predicates) is set up once; per K
// tile only a uniform element offset is added.
//   K-contiguous, full 64-wide K tile : global_load_lds_dwordx4 straight into the swizzled LDS image (no VGPR
//       staging, no ds_write).  LDS-DMA writes lane l at base + 16*l, so thread t owns physical chunk t&7 of row
//       t>>3 and fetches the *logical* chunk (t&7) ^ swizzle(row) from HBM (the swizzle lives on the source
//       address); rows past the edge are clamped to the last valid row (they only feed outputs never stored).
//   K-contiguous, K-tail tile         : register path with zero fill beyond K.
//   K-strided                         : 4 x 16-B loads -> in-register 4x8 transpose -> ds_write_b64.
template <bool TR, int ROWS, int NT = 256> struct Operand {
  static constexpr int RP = NT / 8;  // rows per K-contiguous pass
  static constexpr int NP = TR ? 4 : ROWS / RP;
  const bf16_t* gp[NP];
  const bf16_t* base;
  long ld;
  int rows_valid;
  int mode;  // K-strided: 0 inactive, 1 whole 8-row chunk valid, 2 partial chunk (slow loads)
  uint4 r[NP];

  __device__ __forceinline__ void init(const bf16_t* tile_base, long ld_, int rows_valid_) {
    base = tile_base; ld = ld_; rows_valid = rows_valid_;
    const int t = threadIdx.x;
    if constexpr (!TR) {
      const int row = t >> 3, pc = t & 7;
#pragma unroll
      for (int ps = 0; ps < NP; ++ps) {
        const int rr = row + ps * RP;
        const int rc = rr < rows_valid ? rr : rows_valid - 1;
        gp[ps] = tile_base + (long)rc * ld + ((pc ^ ((rr >> 1) & 7)) << 3);
      }
      mode = 1;
    } else {
      const int kb4 = t & 15, nb = (t >> 6) * 4 + ((t >> 4) & 3);
      const int n = nb * 8;
      mode = (nb < ROWS / 8 && n < rows_valid) ? ((n + 8 <= rows_valid) ? 1 : 2) : 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        gp[i] = tile_base + (long)(kb4 * 4 + i) * ld + n;
        r[i] = make_uint4(0, 0, 0, 0);
      }
    }
  }
  // koff: element offset of this K tile from the tile base (kb * s_kb + k0 [* ld if K-strided])
  __device__ __forceinline__ void issue(long koff, int k_valid, unsigned char* lds_tile, int wave_u) {
    if constexpr (!TR) {
      if (k_valid >= GEMM_BK) {
#pragma unroll
        for (int ps = 0; ps < NP; ++ps)
          __builtin_amdgcn_global_load_lds((gas_ptr)(gp[ps] + koff), (las_ptr)(lds_tile + (ps * RP + wave_u * 8) * 128),
                                           16, 0, 0);
      } else {
        load_kc<ROWS, NT>(base + koff, ld, rows_valid, k_valid, r);
      }
    } else {
      if (k_valid >= GEMM_BK && mode == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const uint4*>(gp[i] + koff);
      } else if (mode != 0) {
        load_ks<ROWS, NT>(base + koff, ld, rows_valid, k_valid, r);
      }
    }
  }
  __device__ __forceinline__ void commit(int k_valid, unsigned char* lds_tile) {
    if constexpr (!TR) {
      if (k_valid < GEMM_BK) store_kc<ROWS, NT>(lds_tile, r);
    } else {
      store_ks<ROWS, NT>(lds_tile, r);
    }
  }
};

