// conv0 + GroupNorm + GELU, backward on the matrix cores (see conv0.hip for the pass it replaces and the algebra of the
// partial sums).  Own translation unit: compiled with -fno-slp-vectorize (build.py) -- the SLP vectoriser pairs the eight
// table lookups of a channel tile two by two (v_pk_fma_f32 on each pair), which turns one LDS round trip into four.
#include "common.hpp"
#include "conv0_shared.hpp"
#include "../../include/wavlm_hip.h"

static_assert(C0_TCH_BWD == 1024, "the chunk partials are laid out per 1024 frames");

// ---- the same pass on the matrix cores (all-bf16 instantiation, C = 512) ---------------------------------------------
// The VALU form above spends ~29 slots per output (10 FMAs for the conv, 10 for P, the table GELU', the affine) and runs
// at 1.8 TB/s of the 1.57 GB gradient.  Here both 10-tap contractions are v_mfma_f32_16x16x32_bf16:
//   conv:  D[t][c] = X[t][k] W^T[k][c]        A = im2col rows of the waveform (16 frames x (10 taps + 6 zeros), the upper 16
//                                             k slots are zero registers), B = the wave's weights (registers)
//   P, A:  D[c][k'] += dz^T[c][t] X'[t][k']   A = dz in exactly the register layout the first MFMA leaves it in (lane: channel
//                                             l & 15, frames 4 q + i of two 16-frame tiles = k slots (q, e) <-> frame
//                                             (e < 4 ? 4 q + e : 16 + 4 q + e - 4)), B = the transposed im2col image with
//                                             a row of ones appended (k' = 10 accumulates A = sum dz)
//   Q, XX: D[j][k] += X'^T X'                 one wave, 32 MFMAs per chunk (the 110 serial LDS loops of the VALU form ran
//                                             12 us on two of the four SIMDs)
// What stays on the VALU per output: the affine (1 fma), table GELU' (3 + LDS read + 1), dz = g * gelu' (1), unpacking g (1),
// packing dz (0.5).  feature_grad_mult scales the partial sums once at the end.
// Workgroup = 8 waves x 64 channels over 1024 frames.  The gradient tile of a wave (32 frames x 64 channels) arrives by
// LDS-DMA into a wave-private double buffer (rows 128 B, 16-byte granules XOR-swizzled by 2 * (row & 3) on the source
// address) and is read with ds_read_b64_tr_b16, which hands lane (channel l & 15, q) its four frames 4 q .. 4 q + 3 -- the
// layout of D[t][c].  No barrier in the frame loop: a wave waits for its own DMA with a counted vmcnt.
typedef __attribute__((ext_vector_type(8))) __bf16 c0_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 c0_bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float c0_f32x4_t;
typedef __attribute__((address_space(3))) c0_bf16x4_t* c0_lds_b4_ptr;
typedef const __attribute__((address_space(1))) void* c0_gas_ptr;
typedef __attribute__((address_space(3))) void* c0_las_ptr;
union C0U4 { uint4 v; c0_bf16x8_t b; unsigned u[4]; };
#define C0M_FR 1024                      // frames per workgroup
#define C0M_XTB (C0M_FR * 2 + 16)        // bytes per row of the transposed image (padded: 11 rows, one bank group apart)
#define C0M_OFF_XC 0                     // [1024][16] bf16 im2col rows
#define C0M_OFF_XT (C0M_FR * 32)         // [11][1024 + 8] bf16
#define C0M_OFF_TAB (C0M_OFF_XT + 11 * C0M_XTB + 112)  // 16-byte aligned
#define C0M_OFF_G (C0M_OFF_TAB + GT_N * 8)
#define C0M_SMEM (C0M_OFF_G + 8 * 2 * 4096)

__global__ __launch_bounds__(512, 1) void conv0_bwd_mfma_kernel(const bf16_t* __restrict__ wav, const bf16_t* __restrict__ W,
    const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta, const float* __restrict__ stats,
    const bf16_t* __restrict__ g, float* __restrict__ part, float* __restrict__ partx, long T, int T0, int stride,
    float gscale, const float2* __restrict__ gtab) {
  constexpr int C = 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char c0sm[];
  unsigned char* xc = c0sm + C0M_OFF_XC;
  unsigned char* xt = c0sm + C0M_OFF_XT;
  float2* tab = reinterpret_cast<float2*>(c0sm + C0M_OFF_TAB);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, q = lane >> 4;
  const int b = blockIdx.y, t0 = blockIdx.x * C0M_FR;
  const int nt = min(C0M_FR, T0 - t0);
  unsigned char* gbuf = c0sm + C0M_OFF_G + wave * 8192;  // 2 stages x [32 frames][64 channels] bf16

  // ---- gradient tiles by LDS-DMA: piece j (1 KiB) = rows 8 j .. 8 j + 7 of the tile, lane -> (row 8 j + (l >> 3), granule l & 7)
  const bf16_t* gsrc;
  {  // wave-uniform base in scalar registers
    const unsigned long a = (unsigned long)(g + ((long)b * T0 + t0) * C + wave * 64);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    gsrc = (const bf16_t*)(((unsigned long)hi << 32) | lo);
  }
  const int grow = lane >> 3;
  const int gcol = ((lane & 7) ^ (2 * (grow & 3))) * 8;  // source granule of LDS granule l & 7 (elements)
  // (inline asm, SGPR base + 32-bit VGPR offset, M0 = LDS destination: through the builtin the compiler orders every
  // later ds_read behind ALL outstanding DMA with s_waitcnt vmcnt(0) -- no prefetch distance -- and spends 64-bit address
  // arithmetic per piece; the waits are counted by hand below)
  const unsigned gcolb = (unsigned)(gcol * 2);
  auto gdma = [&](int ft, int st) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int fr = 32 * ft + 8 * j + grow;
      if (fr > nt - 1) fr = nt - 1;  // clamped rows hold finite data; their frames carry zeros in the im2col images
      const unsigned voff = (unsigned)fr * (unsigned)(C * 2) + gcolb;
      const unsigned ldst = (unsigned)(unsigned long)(c0_las_ptr)(gbuf + st * 4096 + j * 1024);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                   :: "s"(ldst), "v"(voff), "s"(gsrc) : "memory", "m0");
    }
  };
  gdma(0, 0);

  // ---- im2col images of the chunk's waveform, built straight from global memory (every sample is used by two frames)
  const bf16_t* wsrc = wav + (long)b * T + (long)t0 * stride;
  for (int i = threadIdx.x; i < C0M_FR * 8; i += 512) {  // xc[t][2 pr], xc[t][2 pr + 1]
    const int t = i >> 3, pr = i & 7;
    unsigned v = 0;
    if (t < nt && pr < 5) v = (unsigned)wsrc[(long)t * stride + 2 * pr] | ((unsigned)wsrc[(long)t * stride + 2 * pr + 1] << 16);
    *reinterpret_cast<unsigned*>(xc + t * 32 + pr * 4) = v;
  }
  for (int i = threadIdx.x; i < 11 * (C0M_FR / 2); i += 512) {  // xt[k'][t], xt[k'][t + 1]
    const int kk = i / (C0M_FR / 2), t = 2 * (i - kk * (C0M_FR / 2));
    unsigned lo = 0, hi = 0;
    if (kk < C0_KW) {
      if (t < nt) lo = wsrc[(long)t * stride + kk];
      if (t + 1 < nt) hi = wsrc[(long)(t + 1) * stride + kk];
    } else {
      if (t < nt) lo = 0x3f80u;
      if (t + 1 < nt) hi = 0x3f80u;
    }
    *reinterpret_cast<unsigned*>(xt + kk * C0M_XTB + t * 2) = lo | (hi << 16);
  }
  for (int i = threadIdx.x; i < GT_N / 2; i += 512) reinterpret_cast<float4*>(tab)[i] = reinterpret_cast<const float4*>(gtab)[i];

  // ---- per-wave constants: weights as B fragments, affine of the wave's 4 channel tiles
  C0U4 wf[4];
  float zs[4], zb[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int c = wave * 64 + 16 * ct + li;
    wf[ct].v = make_uint4(0, 0, 0, 0);
    if (q == 0) wf[ct].v = *reinterpret_cast<const uint4*>(W + (long)c * C0_KW);  // taps 0..7 (rows are 20 B: 4-byte aligned)
    if (q == 1) wf[ct].u[0] = *reinterpret_cast<const unsigned*>(W + (long)c * C0_KW + 8);  // taps 8, 9
    const float mean = stats[((long)b * C + c) * 2], rstd = stats[((long)b * C + c) * 2 + 1];
    const float gm = bf2f(gamma[c]), bt = bf2f(beta[c]);
    zs[ct] = rstd * gm; zb[ct] = bt - mean * rstd * gm;
  }
  c0_f32x4_t pacc[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) pacc[ct] = c0_f32x4_t{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  // ---- waveform-only sums: XX (10 x 10) and Q (row 10: the ones row) as X'^T X', one wave
  if (wave == 7) {
    c0_f32x4_t xx = c0_f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < C0M_FR / 32; ++s) {
      C0U4 f; f.v = make_uint4(0, 0, 0, 0);
      if (li <= C0_KW) f.v = *reinterpret_cast<const uint4*>(xt + li * C0M_XTB + (32 * s + 8 * q) * 2);
      xx = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.b, f.b, xx, 0, 0, 0);
    }
    // D[j][k]: lane holds rows j = 4 q + i, column k = li
    float* px = partx + ((long)b * gridDim.x + blockIdx.x) * C0_NX;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = 4 * q + i;
      if (li < C0_KW) {
        if (j < C0_KW) px[C0_KW + j * C0_KW + li] = xx[i];
        else if (j == C0_KW) px[li] = xx[i];
      }
    }
  }

  const unsigned trb = (unsigned)((4 * q + (li >> 2)) * 128 + 8 * (li & 1));  // + ((granule ^ swizzle) * 16), + 16-row half
  const int gsw = 2 * (li >> 2);
  const int nft = (nt + 31) >> 5;
  for (int ft = 0; ft < nft; ++ft) {
    if (ft + 1 < nft) {
      gdma(ft + 1, (ft + 1) & 1);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // this tile's four pieces have landed; the next tile's are in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned char* gt = gbuf + (ft & 1) * 4096 + trb;
    C0U4 a0, a1, xf;
    a0.v = make_uint4(0, 0, 0, 0); a1.v = a0.v; xf.v = a0.v;
    if (q < 2) {
      a0.v = *reinterpret_cast<const uint4*>(xc + (32 * ft + li) * 32 + q * 16);
      a1.v = *reinterpret_cast<const uint4*>(xc + (32 * ft + 16 + li) * 32 + q * 16);
    }
    if (li <= C0_KW) {
      const uint2 x0 = *reinterpret_cast<const uint2*>(xt + li * C0M_XTB + (32 * ft + 4 * q) * 2);
      const uint2 x1 = *reinterpret_cast<const uint2*>(xt + li * C0M_XTB + (32 * ft + 16 + 4 * q) * 2);
      xf.v = make_uint4(x0.x, x0.y, x1.x, x1.y);
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const unsigned goff = (unsigned)((((2 * ct + ((li & 3) >> 1)) ^ gsw) & 7) * 16);
      const c0_bf16x4_t g0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((c0_lds_b4_ptr)(gt + goff));
      const c0_bf16x4_t g1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((c0_lds_b4_ptr)(gt + goff + 16 * 128));
      const c0_f32x4_t zero = c0_f32x4_t{0.f, 0.f, 0.f, 0.f};
      const c0_f32x4_t y0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.b, wf[ct].b, zero, 0, 0, 0);
      const c0_f32x4_t y1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.b, wf[ct].b, zero, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise sinks the two gradient reads to their first use)
      // all eight table cells are requested before the first is used (source order is LDS issue order)
      float z[8], u[8], d[8];
      float2 cell[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) { z[i] = fmaf(y0[i], zs[ct], zb[ct]); z[4 + i] = fmaf(y1[i], zs[ct], zb[ct]); }
#pragma unroll
      for (int i = 0; i < 8; ++i) u[i] = __builtin_amdgcn_fmed3f(fmaf(z[i], GT_INV_H, -GT_LO * GT_INV_H), 0.f, (float)(GT_N - 1));
#pragma unroll
      for (int i = 0; i < 8; ++i) cell[i] = tab[(int)u[i]];
      __builtin_amdgcn_sched_barrier(0);  // (... and pairs every table read with its use: eight round trips instead of one)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        d[i] = (float)g0[i] * fmaf(cell[i].x, z[i], cell[i].y);
        d[4 + i] = (float)g1[i] * fmaf(cell[4 + i].x, z[4 + i], cell[4 + i].y);
      }
      C0U4 dz;
      dz.u[0] = pack_bf16x2(d[0], d[1]); dz.u[1] = pack_bf16x2(d[2], d[3]);
      dz.u[2] = pack_bf16x2(d[4], d[5]); dz.u[3] = pack_bf16x2(d[6], d[7]);
      pacc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dz.b, xf.b, pacc[ct], 0, 0, 0);
    }
  }
  // D[c][k']: lane holds channels 16 ct + 4 q + i, column k' = li (0..9: P, 10: A)
  float* out = part + ((long)b * gridDim.x + blockIdx.x) * (long)C0_NQ * C;
  if (li <= C0_KW) {
    const int qi = li == C0_KW ? 0 : li + 1;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i) out[(long)qi * C + wave * 64 + 16 * ct + 4 * q + i] = pacc[ct][i] * gscale;
  }
}

int conv0_bwd_mfma_launch(const void* wav, const void* W, const void* gamma, const void* beta, const float* stats, const void* g,
                          float* part, float* partx, long T, int T0, int stride, float gscale, int nchunk, int B,
                          const float2* tab1, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)conv0_bwd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, C0M_SMEM) != hipSuccess)
      return WL_ELAUNCH;
    attr = true;
  }
  WL_LAUNCH(conv0_bwd_mfma_kernel, dim3((unsigned)nchunk, (unsigned)B), dim3(512), C0M_SMEM, st, (const bf16_t*)wav,
            (const bf16_t*)W, (const bf16_t*)gamma, (const bf16_t*)beta, stats, (const bf16_t*)g, part, partx, T, T0, stride,
            gscale, tab1);
  return wl_check_launch();
}
