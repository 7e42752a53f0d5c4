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

// =================================================================================================================
// extractor_mode = "layer_norm" (WavLM-Large / UniSpeech-SAT Large: WavLM/WavLM.py:403-418), all-bf16 instantiation, C = 512.
// The VALU form (conv0.hip: a wave owns a frame, four dependent wave reductions per frame, ~260 VGPRs) ran 2.7 ms on the
// 1 GB gradient of the Large step.  Here the frame statistics come from the waveform alone and the second pass over the
// channels disappears by algebra, so the pass has the shape of the GroupNorm-mode kernel above:
//   mean_t = wbar . x_t + cbar,  var_t = x_t^T G x_t + 2 u . x_t + s        (G, u, s: centred second moments of (W, bias) over
//                                                                            the channels: conv0_ln_gram_kernel, 122 numbers)
//   y = conv (MFMA, the conv bias as an eleventh tap against a column of ones), xh = (y - mean_t) rstd_t, z = xh gamma + beta,
//   dz = g gelu'(z), h = dz gamma, s1_t = mean_c h, s2_t = mean_c h xh,  dconv[t][c] = rstd_t (h - s1_t - xh s2_t)
//   dW[c][k] = sum_t dconv[t][c] x[t][k]
//            = gamma_c T[c][k] - v[k] - (sum_j W[c][j] M[j][k] + cb_c m1[k] - m0[k])
//     T[c][k] = sum_t dz[t][c] rstd_t x[t][k]     second MFMA, dz in the layout the first leaves it in, against the image
//                                                  rstd_t x[t][k] (+ columns rstd_t: the bias gradient, 1: dbeta)
//     v, M, m1, m0: moments of the waveform weighted by b_t = rstd_t s1_t and a_t = rstd_t^2 s2_t -- a 16 x 16 MFMA
//                   X''^T diag(.) X'' over the workgroup's frames once s1 / s2 are complete (weights split hi + lo in bf16)
//   dgamma[c] = sum_t dz xh (lane-local: the lane owns the channel), dbeta[c] = sum_t dz (the ones column).
// s1_t / s2_t: the lane sums its four channel tiles, a 16-lane DPP sum per frame and tile, the eight waves' partials meet
// in LDS at the end -- deterministic, no atomics.  512 frames per workgroup (the per-wave frame partials are 32 KB of LDS).
#define C0L_FR 512
#define C0L_XTB (C0L_FR * 2 + 16)                 // bytes per row of a transposed image
#define C0L_OFF_XC 0                              // [512][16] bf16: taps 0..9, 1.0, zeros (A operand of the conv)
#define C0L_OFF_XT (C0L_FR * 32)                  // [12][512 + 8] bf16: rstd_t x[t][k], rstd_t, 1
#define C0L_OFF_XR (C0L_OFF_XT + 12 * C0L_XTB)    // [13][512 + 8] bf16: x[t][k], 1, mean_t (hi), mean_t (lo)
#define C0L_OFF_ST (C0L_OFF_XR + 13 * C0L_XTB)    // [512] float2 (rstd_t, -mean_t rstd_t); after the frame loop (a_t, b_t)
#define C0L_OFF_TAB (C0L_OFF_ST + C0L_FR * 8)
#define C0L_OFF_S12 (C0L_OFF_TAB + GT_N * 8)      // [8 waves][512] float2 (s1, s2) partials; the prologue's waveform segment
#define C0L_OFF_G (C0L_OFF_S12 + 8 * C0L_FR * 8)  // gradient tiles, 2 stages x 4 KiB per wave
#define C0L_SMEM (C0L_OFF_G + 8 * 2 * 4096)
static_assert(C0L_SMEM <= 160 * 1024, "LDS budget");
static_assert(C0L_OFF_XT % 16 == 0 && C0L_OFF_XR % 16 == 0 && C0L_OFF_ST % 16 == 0 && C0L_OFF_TAB % 16 == 0, "alignment");

// gc[128]: [0..9] wbar, [10] cbar, [11] s, [12..21] 2 u, [22..121] G[j][k]
// One workgroup of 1024 threads, the parameters in LDS, every output summed in double by eight threads over 64 channels each
// (a single thread per output over 512 dependent global loads took 226 us).
__global__ __launch_bounds__(1024) void conv0_ln_gram_kernel(const bf16_t* __restrict__ W, const bf16_t* __restrict__ cb,
                                                             float* __restrict__ gc, int C) {
  __shared__ float sw[512][C0_KW + 1];   // [c][0..9] taps, [c][10] conv bias
  __shared__ double pp[111][8];
  __shared__ double wb[11];
  const int tid = threadIdx.x;
  for (int i = tid; i < C * C0_KW; i += 1024) sw[i / C0_KW][i % C0_KW] = bf2f(W[i]);
  for (int c = tid; c < C; c += 1024) sw[c][C0_KW] = cb ? bf2f(cb[c]) : 0.f;
  __syncthreads();
  const int o = tid >> 3, p = tid & 7, per = C / 8;
  if (o < 11) {
    double s = 0.0;
    for (int e = 0; e < per; ++e) s += (double)sw[e * 8 + p][o];  // (channels interleaved over the eight threads: LDS banks)
    pp[o][p] = s;
  }
  __syncthreads();
  if (tid < 11) {
    double s = 0.0;
    for (int e = 0; e < 8; ++e) s += pp[tid][e];
    wb[tid] = s / C;
    gc[tid] = (float)wb[tid];
  }
  __syncthreads();
  if (o < 111) {  // o < 100: G[j][k]; 100..109: u[k]; 110: s  (u, s: column 10 = the bias)
    const int j = o < 100 ? o / C0_KW : C0_KW, k = o < 100 ? o % C0_KW : (o < 110 ? o - 100 : C0_KW);
    const double mj = wb[j], mk = wb[k];
    double s = 0.0;
    for (int e = 0; e < per; ++e) { const int c = e * 8 + p; s += ((double)sw[c][j] - mj) * ((double)sw[c][k] - mk); }
    pp[o][p] = s;
  }
  __syncthreads();
  if (tid < 111) {
    double s = 0.0;
    for (int e = 0; e < 8; ++e) s += pp[tid][e];
    s /= C;
    if (tid < 100) gc[22 + tid] = (float)s;
    else if (tid < 110) gc[12 + (tid - 100)] = (float)(2.0 * s);
    else gc[11] = (float)s;
  }
}

__device__ __forceinline__ float c0_row_sum(float v) {  // sum over the 16 lanes of a DPP row, in every lane
  v += wl_dpp_f32<0xB1>(v);
  v += wl_dpp_f32<0x4E>(v);
  v += wl_dpp_f32<0x141>(v);
  v += wl_dpp_f32<0x140>(v);
  return v;
}

#define C0L_NROW 13                          // per-channel partial rows: T[.][0..9], sum dz rstd, dbeta, dgamma
#define C0L_NV (C0L_NROW * 512 + 256)        // ... + the 16 x 16 moment matrix: one workgroup's partial record (floats)
#define C0L_SLICES 32

__global__ __launch_bounds__(512, 1) void conv0_ln_bwd_mfma_kernel(const bf16_t* __restrict__ wav, const bf16_t* __restrict__ W,
    const bf16_t* __restrict__ cbias, const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta,
    const bf16_t* __restrict__ g, const float* __restrict__ gc, float* __restrict__ part, long T, int T0, int stride, float eps,
    const float2* __restrict__ gtab, int nchunk, int cpb) {
  constexpr int C = 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char c0sm[];
  unsigned char* xc = c0sm + C0L_OFF_XC;
  unsigned char* xt = c0sm + C0L_OFF_XT;
  unsigned char* xr = c0sm + C0L_OFF_XR;
  float2* stt = reinterpret_cast<float2*>(c0sm + C0L_OFF_ST);
  float2* tab = reinterpret_cast<float2*>(c0sm + C0L_OFF_TAB);
  float2* s12 = reinterpret_cast<float2*>(c0sm + C0L_OFF_S12);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, q = lane >> 4;
  const int b = blockIdx.y;
  unsigned char* gbuf = c0sm + C0L_OFF_G + wave * 8192;
  int nt = 0;
  const bf16_t* gsrc = g;
  const int grow = lane >> 3;
  const int gcol = ((lane & 7) ^ (2 * (grow & 3))) * 8;
  const unsigned gcolb = (unsigned)(gcol * 2);
  auto gdma = [&](int ft, int st) __attribute__((always_inline)) {  // as in the GroupNorm-mode kernel
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int fr = 32 * ft + 8 * j + grow;
      if (fr > nt - 1) fr = nt - 1;
      const unsigned voff = (unsigned)fr * (unsigned)(C * 2) + gcolb;
      const unsigned ldst = (unsigned)(unsigned long)(c0_las_ptr)(gbuf + st * 4096 + j * 1024);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                   :: "s"(ldst), "v"(voff), "s"(gsrc) : "memory", "m0");
    }
  };
  // ---- per-wave constants: weights (+ conv bias as tap 10) as B fragments, affine of the wave's 4 channel tiles
  C0U4 wf[4];
  float gm[4], bt[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int c = wave * 64 + 16 * ct + li;
    wf[ct].v = make_uint4(0, 0, 0, 0);
    if (q == 0) wf[ct].v = *reinterpret_cast<const uint4*>(W + (long)c * C0_KW);
    if (q == 1) {
      wf[ct].u[0] = *reinterpret_cast<const unsigned*>(W + (long)c * C0_KW + 8);
      wf[ct].u[1] = cbias ? (unsigned)cbias[c] : 0u;
    }
    gm[ct] = bf2f(gamma[c]); bt[ct] = bf2f(beta[c]);
  }
  c0_f32x4_t pacc[4];
  float a2[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) { pacc[ct] = c0_f32x4_t{0.f, 0.f, 0.f, 0.f}; a2[ct] = 0.f; }
  c0_f32x4_t macc = c0_f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < GT_N / 2; i += 512) reinterpret_cast<float4*>(tab)[i] = reinterpret_cast<const float4*>(gtab)[i];

  // A workgroup walks cpb consecutive 512-frame chunks of one batch row and keeps the channel sums, dgamma and the moment
  // matrix in registers across them: one partial record per workgroup instead of one per chunk.
  for (int cc = 0; cc < cpb; ++cc) {
  const int chunk = blockIdx.x * cpb + cc;
  if (chunk >= nchunk) break;
  const int t0 = chunk * C0L_FR;
  nt = min(C0L_FR, T0 - t0);
  {
    const unsigned long a = (unsigned long)(g + ((long)b * T0 + t0) * C + wave * 64);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    gsrc = (const bf16_t*)(((unsigned long)hi << 32) | lo);
  }
  gdma(0, 0);

  // ---- the chunk's waveform segment (into the frame-partial region, which the frame loop only writes later) + GELU' table
  bf16_t* seg = reinterpret_cast<bf16_t*>(c0sm + C0L_OFF_S12);
  const bf16_t* wsrc = wav + (long)b * T + (long)t0 * stride;
  const int nseg = (nt - 1) * stride + C0_KW;
  for (int i = threadIdx.x; i < nseg; i += 512) seg[i] = wsrc[i];
  __syncthreads();  // (also: every wave has left the previous chunk's moment pass)

  // ---- thread = frame: statistics from the waveform, the im2col row and the frame's column of the two transposed images
  {
    const int t = threadIdx.x;
    const bool ok = t < nt;
    unsigned xb[C0_KW];
    float x[C0_KW];
#pragma unroll
    for (int k = 0; k < C0_KW; ++k) { xb[k] = ok ? (unsigned)seg[t * stride + k] : 0u; x[k] = __uint_as_float(xb[k] << 16); }
    float mean = gc[10];
#pragma unroll
    for (int k = 0; k < C0_KW; ++k) mean = fmaf(gc[k], x[k], mean);
    float var = gc[11];
#pragma unroll
    for (int k = 0; k < C0_KW; ++k) {
      float r = gc[12 + k];
#pragma unroll
      for (int j = 0; j < C0_KW; ++j) r = fmaf(gc[22 + k * C0_KW + j], x[j], r);
      var = fmaf(r, x[k], var);
    }
    const float rstd = ok ? rsqrtf(fmaxf(var, 0.f) + eps) : 0.f;
    if (!ok) mean = 0.f;
    stt[t] = make_float2(rstd, -mean * rstd);
    const unsigned one = ok ? 0x3f80u : 0u;
    *reinterpret_cast<uint4*>(xc + t * 32) = make_uint4(xb[0] | (xb[1] << 16), xb[2] | (xb[3] << 16), xb[4] | (xb[5] << 16), xb[6] | (xb[7] << 16));
    *reinterpret_cast<uint4*>(xc + t * 32 + 16) = make_uint4(xb[8] | (xb[9] << 16), one, 0u, 0u);
#pragma unroll
    for (int k = 0; k < C0_KW; ++k) {
      *reinterpret_cast<bf16_t*>(xr + k * C0L_XTB + t * 2) = (bf16_t)xb[k];
      *reinterpret_cast<bf16_t*>(xt + k * C0L_XTB + t * 2) = f2bf(rstd * x[k]);
    }
    const bf16_t mh = f2bf(mean);
    *reinterpret_cast<bf16_t*>(xr + 10 * C0L_XTB + t * 2) = (bf16_t)one;
    *reinterpret_cast<bf16_t*>(xr + 11 * C0L_XTB + t * 2) = mh;
    *reinterpret_cast<bf16_t*>(xr + 12 * C0L_XTB + t * 2) = f2bf(mean - bf2f(mh));
    *reinterpret_cast<bf16_t*>(xt + 10 * C0L_XTB + t * 2) = f2bf(rstd);
    *reinterpret_cast<bf16_t*>(xt + 11 * C0L_XTB + t * 2) = (bf16_t)one;
  }

  __syncthreads();  // images and statistics complete; the segment is dead: the frame loop may write the frame partials

  const unsigned trb = (unsigned)((4 * q + (li >> 2)) * 128 + 8 * (li & 1));
  const int gsw = 2 * (li >> 2);
  const int nft = (nt + 31) >> 5;
  for (int ft = 0; ft < nft; ++ft) {
    if (ft + 1 < nft) {
      gdma(ft + 1, (ft + 1) & 1);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
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
    if (li <= 11) {
      const uint2 x0 = *reinterpret_cast<const uint2*>(xt + li * C0L_XTB + (32 * ft + 4 * q) * 2);
      const uint2 x1 = *reinterpret_cast<const uint2*>(xt + li * C0L_XTB + (32 * ft + 16 + 4 * q) * 2);
      xf.v = make_uint4(x0.x, x0.y, x1.x, x1.y);
    }
    // the lane's eight frames: 32 ft + 4 q + i and 32 ft + 16 + 4 q + i
    float rs[8], nm[8];
    {
      const float4* p0 = reinterpret_cast<const float4*>(stt + 32 * ft + 4 * q);
      const float4* p1 = reinterpret_cast<const float4*>(stt + 32 * ft + 16 + 4 * q);
      const float4 v0 = p0[0], v1 = p0[1], v2 = p1[0], v3 = p1[1];
      rs[0] = v0.x; nm[0] = v0.y; rs[1] = v0.z; nm[1] = v0.w; rs[2] = v1.x; nm[2] = v1.y; rs[3] = v1.z; nm[3] = v1.w;
      rs[4] = v2.x; nm[4] = v2.y; rs[5] = v2.z; nm[5] = v2.w; rs[6] = v3.x; nm[6] = v3.y; rs[7] = v3.z; nm[7] = v3.w;
    }
    float s1p[8], s2p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s1p[i] = 0.f; s2p[i] = 0.f; }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const unsigned goff = (unsigned)((((2 * ct + ((li & 3) >> 1)) ^ gsw) & 7) * 16);
      const c0_bf16x4_t g0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((c0_lds_b4_ptr)(gt + goff));
      const c0_bf16x4_t g1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((c0_lds_b4_ptr)(gt + goff + 16 * 128));
      const c0_f32x4_t zero = c0_f32x4_t{0.f, 0.f, 0.f, 0.f};
      const c0_f32x4_t y0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.b, wf[ct].b, zero, 0, 0, 0);
      const c0_f32x4_t y1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.b, wf[ct].b, zero, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      float xh[8], z[8], u[8], d[8];
      float2 cell[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) { xh[i] = fmaf(y0[i], rs[i], nm[i]); xh[4 + i] = fmaf(y1[i], rs[4 + i], nm[4 + i]); }
#pragma unroll
      for (int i = 0; i < 8; ++i) z[i] = fmaf(xh[i], gm[ct], bt[ct]);
#pragma unroll
      for (int i = 0; i < 8; ++i) u[i] = __builtin_amdgcn_fmed3f(fmaf(z[i], GT_INV_H, -GT_LO * GT_INV_H), 0.f, (float)(GT_N - 1));
#pragma unroll
      for (int i = 0; i < 8; ++i) cell[i] = tab[(int)u[i]];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        d[i] = (float)g0[i] * fmaf(cell[i].x, z[i], cell[i].y);
        d[4 + i] = (float)g1[i] * fmaf(cell[4 + i].x, z[4 + i], cell[4 + i].y);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float h = d[i] * gm[ct];
        s1p[i] += h;
        s2p[i] = fmaf(h, xh[i], s2p[i]);
        a2[ct] = fmaf(d[i], xh[i], a2[ct]);
      }
      C0U4 dz;
      dz.u[0] = pack_bf16x2(d[0], d[1]); dz.u[1] = pack_bf16x2(d[2], d[3]);
      dz.u[2] = pack_bf16x2(d[4], d[5]); dz.u[3] = pack_bf16x2(d[6], d[7]);
      pacc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dz.b, xf.b, pacc[ct], 0, 0, 0);
    }
    // frame sums over the wave's 64 channels: 16 lanes of a DPP row hold one frame's partials
#pragma unroll
    for (int i = 0; i < 8; ++i) { s1p[i] = c0_row_sum(s1p[i]); s2p[i] = c0_row_sum(s2p[i]); }
    if (li == 0) {
      float4* o0 = reinterpret_cast<float4*>(s12 + wave * C0L_FR + 32 * ft + 4 * q);
      float4* o1 = reinterpret_cast<float4*>(s12 + wave * C0L_FR + 32 * ft + 16 + 4 * q);
      o0[0] = make_float4(s1p[0], s2p[0], s1p[1], s2p[1]); o0[1] = make_float4(s1p[2], s2p[2], s1p[3], s2p[3]);
      o1[0] = make_float4(s1p[4], s2p[4], s1p[5], s2p[5]); o1[1] = make_float4(s1p[6], s2p[6], s1p[7], s2p[7]);
    }
  }
  __syncthreads();  // every wave's frame partials are in LDS; the gradient buffers are free

  // a_t = rstd^2 s2_t, b_t = rstd s1_t (means over the C channels), thread = frame
  {
    const int t = threadIdx.x;
    float s1 = 0.f, s2 = 0.f;
    if (t < nft * 32) {
#pragma unroll
      for (int w = 0; w < 8; ++w) { const float2 v = s12[w * C0L_FR + t]; s1 += v.x; s2 += v.y; }
    }
    const float rstd = stt[t].x;  // 0 for frames past the end
    const float invC = 1.f / (float)C;
    stt[t] = t < nt ? make_float2(rstd * rstd * s2 * invC, rstd * s1 * invC) : make_float2(0.f, 0.f);
  }
  __syncthreads();

  // ---- weighted waveform moments: D[j][k] = sum_t wgt_j(t) x''[t][j] x''[t][k]; rows j <= 12 weighted by a_t, row 13 = b_t
#pragma unroll
  for (int ss = 0; ss < 2; ++ss) {
    const int fb = 32 * (2 * wave + ss) + 8 * q;  // the lane's 8 frames = k slots 8 q .. 8 q + 7 of both operands
    C0U4 bf; bf.v = make_uint4(0, 0, 0, 0);
    if (li <= 12) bf.v = *reinterpret_cast<const uint4*>(xr + li * C0L_XTB + fb * 2);
    const float4* wp = reinterpret_cast<const float4*>(stt + fb);
    float wv[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float4 v = wp[e]; wv[2 * e] = li == 13 ? v.y : v.x; wv[2 * e + 1] = li == 13 ? v.w : v.z; }
    float hi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned bits = (bf.u[e >> 1] >> (16 * (e & 1))) & 0xffffu;
      const float base = li <= 12 ? __uint_as_float(bits << 16) : (li == 13 ? 1.f : 0.f);
      const float v = wv[e] * base;
      hi[e] = bf2f(f2bf(v));
      lo[e] = v - hi[e];
    }
    C0U4 ah, al;
#pragma unroll
    for (int e = 0; e < 4; ++e) { ah.u[e] = pack_bf16x2(hi[2 * e], hi[2 * e + 1]); al.u[e] = pack_bf16x2(lo[2 * e], lo[2 * e + 1]); }
    macc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah.b, bf.b, macc, 0, 0, 0);
    macc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al.b, bf.b, macc, 0, 0, 0);
  }
  }  // chunks of this workgroup

  float* out = part + ((long)b * gridDim.x + blockIdx.x) * (long)C0L_NV;
  // T / sum dz rstd / dbeta: D[c][k'], lane holds channels 16 ct + 4 q + i, column k' = li
  if (li <= 11) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i) out[(long)li * C + wave * 64 + 16 * ct + 4 * q + i] = pacc[ct][i];
  }
  {  // dgamma: the four q groups of a channel through the wave's own (now free) gradient buffer
    float* wsc = reinterpret_cast<float*>(gbuf);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) wsc[q * 64 + ct * 16 + li] = a2[ct];
  }
  __syncthreads();
  {
    const float* wsc = reinterpret_cast<const float*>(gbuf);
    out[(long)12 * C + wave * 64 + lane] = (wsc[lane] + wsc[64 + lane]) + (wsc[128 + lane] + wsc[192 + lane]);
  }
  __syncthreads();  // (the moment partials below reuse the first waves' buffers)

  float* md = reinterpret_cast<float*>(c0sm + C0L_OFF_G);  // [8 waves][16][16]
#pragma unroll
  for (int i = 0; i < 4; ++i) md[wave * 256 + (4 * q + i) * 16 + li] = macc[i];
  __syncthreads();
  if (threadIdx.x < 256) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += md[w * 256 + threadIdx.x];
    out[(long)C0L_NROW * C + threadIdx.x] = s;
  }
}

// ---- forward of the same block on the matrix cores ----------------------------------------------------------------------
// The VALU form (conv0.hip: a wave owns a frame, two wave reductions per frame, 10 FMAs per output) wrote the Large step's
// 2.1 GB at 2.6 TB/s.  Here: frame statistics from the waveform (thread = frame, the Gram form above), the conv as one MFMA per
// 16 channels x 16 frames with the WEIGHTS as the A operand, so that D[channel][frame] leaves a lane with one frame (column
// li) and, with tile ct's rows mapped to channels 16 (r >> 2) + 4 ct + (r & 3), sixteen CONSECUTIVE channels 16 q .. 16 q + 15
// of the wave's 64 across its four tiles: 32 bytes per frame and lane, two 16-byte stores; the four q groups of a frame
// fill a 128-byte line.
#define C0F_OFF_XC 0                                // [512][16] bf16 im2col rows (taps, 1.0, zeros)
#define C0F_OFF_ST (C0L_FR * 32)                    // [512] float2 (rstd_t, -mean_t rstd_t)
#define C0F_OFF_TAB (C0F_OFF_ST + C0L_FR * 8)       // GELU chords
#define C0F_OFF_SEG (C0F_OFF_TAB + GT_N * 8)        // the chunk's waveform segment, (511 * 8 + 10) samples at most
#define C0F_SMEM (C0F_OFF_SEG + 8208)

__global__ __launch_bounds__(512, 2) void conv0_ln_fwd_mfma_kernel(const bf16_t* __restrict__ wav, const bf16_t* __restrict__ W,
    const bf16_t* __restrict__ cbias, const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta,
    const float* __restrict__ gc, bf16_t* __restrict__ out, long T, int T0, int stride, float eps, const float2* __restrict__ gtab) {
  constexpr int C = 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char c0sm[];
  unsigned char* xc = c0sm + C0F_OFF_XC;
  float2* stt = reinterpret_cast<float2*>(c0sm + C0F_OFF_ST);
  float2* tab = reinterpret_cast<float2*>(c0sm + C0F_OFF_TAB);
  bf16_t* seg = reinterpret_cast<bf16_t*>(c0sm + C0F_OFF_SEG);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, q = lane >> 4;
  const int b = blockIdx.y, t0 = blockIdx.x * C0L_FR;
  const int nt = min(C0L_FR, T0 - t0);
  const bf16_t* wsrc = wav + (long)b * T + (long)t0 * stride;
  const int nseg = (nt - 1) * stride + C0_KW;
  for (int i = threadIdx.x; i < nseg; i += 512) seg[i] = wsrc[i];
  for (int i = threadIdx.x; i < GT_N / 2; i += 512) reinterpret_cast<float4*>(tab)[i] = reinterpret_cast<const float4*>(gtab)[i];

  // weights (+ conv bias as tap 10) as A fragments: row li of tile ct = channel 16 (li >> 2) + 4 ct + (li & 3) of the wave's 64
  C0U4 wf[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int c = wave * 64 + 16 * (li >> 2) + 4 * ct + (li & 3);
    wf[ct].v = make_uint4(0, 0, 0, 0);
    if (q == 0) wf[ct].v = *reinterpret_cast<const uint4*>(W + (long)c * C0_KW);
    if (q == 1) {
      wf[ct].u[0] = *reinterpret_cast<const unsigned*>(W + (long)c * C0_KW + 8);
      wf[ct].u[1] = cbias ? (unsigned)cbias[c] : 0u;
    }
  }
  // affine of the lane's sixteen output channels wave * 64 + 16 q + (4 ct + i)
  float gm[16], bt[16];
  {
    const int c0 = wave * 64 + 16 * q;
    C0U4 g0, g1, b0, b1;
    g0.v = *reinterpret_cast<const uint4*>(gamma + c0); g1.v = *reinterpret_cast<const uint4*>(gamma + c0 + 8);
    b0.v = *reinterpret_cast<const uint4*>(beta + c0); b1.v = *reinterpret_cast<const uint4*>(beta + c0 + 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int sh = 16 * (e & 1);
      gm[e] = __uint_as_float(((g0.u[e >> 1] >> sh) & 0xffffu) << 16); gm[8 + e] = __uint_as_float(((g1.u[e >> 1] >> sh) & 0xffffu) << 16);
      bt[e] = __uint_as_float(((b0.u[e >> 1] >> sh) & 0xffffu) << 16); bt[8 + e] = __uint_as_float(((b1.u[e >> 1] >> sh) & 0xffffu) << 16);
    }
  }
  __syncthreads();
  {  // thread = frame: statistics and the im2col row
    const int t = threadIdx.x;
    const bool ok = t < nt;
    unsigned xb[C0_KW];
    float x[C0_KW];
#pragma unroll
    for (int k = 0; k < C0_KW; ++k) { xb[k] = ok ? (unsigned)seg[t * stride + k] : 0u; x[k] = __uint_as_float(xb[k] << 16); }
    float mean = gc[10];
#pragma unroll
    for (int k = 0; k < C0_KW; ++k) mean = fmaf(gc[k], x[k], mean);
    float var = gc[11];
#pragma unroll
    for (int k = 0; k < C0_KW; ++k) {
      float r = gc[12 + k];
#pragma unroll
      for (int j = 0; j < C0_KW; ++j) r = fmaf(gc[22 + k * C0_KW + j], x[j], r);
      var = fmaf(r, x[k], var);
    }
    const float rstd = ok ? rsqrtf(fmaxf(var, 0.f) + eps) : 0.f;
    stt[t] = make_float2(rstd, ok ? -mean * rstd : 0.f);
    *reinterpret_cast<uint4*>(xc + t * 32) = make_uint4(xb[0] | (xb[1] << 16), xb[2] | (xb[3] << 16), xb[4] | (xb[5] << 16), xb[6] | (xb[7] << 16));
    *reinterpret_cast<uint4*>(xc + t * 32 + 16) = make_uint4(xb[8] | (xb[9] << 16), ok ? 0x3f80u : 0u, 0u, 0u);
  }
  __syncthreads();

  bf16_t* orow = out + ((long)b * T0 + t0) * C + wave * 64 + 16 * q;
  const int nft = (nt + 31) >> 5;
  for (int ft = 0; ft < nft; ++ft) {
    const int f0 = 32 * ft + li, f1 = f0 + 16;
    C0U4 x0, x1;
    x0.v = make_uint4(0, 0, 0, 0); x1.v = x0.v;
    if (q < 2) {
      x0.v = *reinterpret_cast<const uint4*>(xc + f0 * 32 + q * 16);
      x1.v = *reinterpret_cast<const uint4*>(xc + f1 * 32 + q * 16);
    }
    const float2 s0 = stt[f0], s1 = stt[f1];
    unsigned o0[8], o1[8];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const c0_f32x4_t zero = c0_f32x4_t{0.f, 0.f, 0.f, 0.f};
      const c0_f32x4_t y0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct].b, x0.b, zero, 0, 0, 0);
      const c0_f32x4_t y1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct].b, x1.b, zero, 0, 0, 0);
      float z[8], u[8], r[8];
      float2 cell[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        z[i] = fmaf(fmaf(y0[i], s0.x, s0.y), gm[4 * ct + i], bt[4 * ct + i]);
        z[4 + i] = fmaf(fmaf(y1[i], s1.x, s1.y), gm[4 * ct + i], bt[4 * ct + i]);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) u[i] = __builtin_amdgcn_fmed3f(fmaf(z[i], GT_INV_H, -GT_LO * GT_INV_H), 0.f, (float)(GT_N - 1));
#pragma unroll
      for (int i = 0; i < 8; ++i) cell[i] = tab[(int)u[i]];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 8; ++i) r[i] = fmaf(cell[i].x, z[i], cell[i].y);
      o0[2 * ct] = pack_bf16x2(r[0], r[1]); o0[2 * ct + 1] = pack_bf16x2(r[2], r[3]);
      o1[2 * ct] = pack_bf16x2(r[4], r[5]); o1[2 * ct + 1] = pack_bf16x2(r[6], r[7]);
    }
    if (f0 < nt) {
      uint4* p = reinterpret_cast<uint4*>(orow + (long)f0 * C);
      p[0] = make_uint4(o0[0], o0[1], o0[2], o0[3]); p[1] = make_uint4(o0[4], o0[5], o0[6], o0[7]);
    }
    if (f1 < nt) {
      uint4* p = reinterpret_cast<uint4*>(orow + (long)f1 * C);
      p[0] = make_uint4(o1[0], o1[1], o1[2], o1[3]); p[1] = make_uint4(o1[4], o1[5], o1[6], o1[7]);
    }
  }
}

int conv0_ln_fwd_mfma_launch(const void* wav, const void* W, const void* cbias, const void* gamma, const void* beta, void* out,
                             float* gc, long T, int T0, int stride, int B, float eps, const float2* tab0, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)conv0_ln_fwd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, C0F_SMEM) != hipSuccess)
      return WL_ELAUNCH;
    attr = true;
  }
  const int nchunk = (T0 + C0L_FR - 1) / C0L_FR;
  WL_LAUNCH(conv0_ln_gram_kernel, dim3(1), dim3(1024), 0, st, (const bf16_t*)W, (const bf16_t*)cbias, gc, 512);
  WL_LAUNCH(conv0_ln_fwd_mfma_kernel, dim3((unsigned)nchunk, (unsigned)B), dim3(512), C0F_SMEM, st, (const bf16_t*)wav,
            (const bf16_t*)W, (const bf16_t*)cbias, (const bf16_t*)gamma, (const bf16_t*)beta, gc, (bf16_t*)out, T, T0, stride, eps,
            tab0);
  return wl_check_launch();
}

// ---- GroupNorm-mode forward (WavLM-Base: the headline step's first kernel) in the same form: the statistics are per
// (batch row, channel) and arrive from the Gram pass (conv0.hip), so the affine is one fma with the lane's sixteen channels'
// (rstd gamma, beta - mean rstd gamma).  The VALU form (10 FMAs per output) wrote the 1.57 GB at 3.9 TB/s.
#define C0G_OFF_XC 0
#define C0G_OFF_TAB (C0L_FR * 32)
#define C0G_OFF_SEG (C0G_OFF_TAB + GT_N * 8)
#define C0G_SMEM (C0G_OFF_SEG + 8208)

__global__ __launch_bounds__(512, 2) void conv0_gn_fwd_mfma_kernel(const bf16_t* __restrict__ wav, const bf16_t* __restrict__ W,
    const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta, const float* __restrict__ stats, bf16_t* __restrict__ out,
    long T, int T0, int stride, const float2* __restrict__ gtab) {
  constexpr int C = 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char c0sm[];
  unsigned char* xc = c0sm + C0G_OFF_XC;
  float2* tab = reinterpret_cast<float2*>(c0sm + C0G_OFF_TAB);
  bf16_t* seg = reinterpret_cast<bf16_t*>(c0sm + C0G_OFF_SEG);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, q = lane >> 4;
  const int b = blockIdx.y, t0 = blockIdx.x * C0L_FR;
  const int nt = min(C0L_FR, T0 - t0);
  const bf16_t* wsrc = wav + (long)b * T + (long)t0 * stride;
  const int nseg = (nt - 1) * stride + C0_KW;
  for (int i = threadIdx.x; i < nseg; i += 512) seg[i] = wsrc[i];
  for (int i = threadIdx.x; i < GT_N / 2; i += 512) reinterpret_cast<float4*>(tab)[i] = reinterpret_cast<const float4*>(gtab)[i];
  C0U4 wf[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int c = wave * 64 + 16 * (li >> 2) + 4 * ct + (li & 3);
    wf[ct].v = make_uint4(0, 0, 0, 0);
    if (q == 0) wf[ct].v = *reinterpret_cast<const uint4*>(W + (long)c * C0_KW);
    if (q == 1) wf[ct].u[0] = *reinterpret_cast<const unsigned*>(W + (long)c * C0_KW + 8);
  }
  float zs[16], zb[16];
  {
    const int c0 = wave * 64 + 16 * q;
    C0U4 g0, g1, b0, b1;
    g0.v = *reinterpret_cast<const uint4*>(gamma + c0); g1.v = *reinterpret_cast<const uint4*>(gamma + c0 + 8);
    b0.v = *reinterpret_cast<const uint4*>(beta + c0); b1.v = *reinterpret_cast<const uint4*>(beta + c0 + 8);
    const float4* sp = reinterpret_cast<const float4*>(stats + ((long)b * C + c0) * 2);  // (mean, rstd) x 16
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const C0U4& gg = e < 8 ? g0 : g1;
      const C0U4& bb = e < 8 ? b0 : b1;
      const int sh = 16 * (e & 1), w = (e & 7) >> 1;
      const float gm = __uint_as_float(((gg.u[w] >> sh) & 0xffffu) << 16), bt = __uint_as_float(((bb.u[w] >> sh) & 0xffffu) << 16);
      const float4 st2 = sp[e >> 1];
      const float mean = (e & 1) ? st2.z : st2.x, rstd = (e & 1) ? st2.w : st2.y;
      zs[e] = rstd * gm; zb[e] = bt - mean * rstd * gm;
    }
  }
  __syncthreads();
  {  // thread = frame: the im2col row
    const int t = threadIdx.x;
    const bool ok = t < nt;
    unsigned xb[C0_KW];
#pragma unroll
    for (int k = 0; k < C0_KW; ++k) xb[k] = ok ? (unsigned)seg[t * stride + k] : 0u;
    *reinterpret_cast<uint4*>(xc + t * 32) = make_uint4(xb[0] | (xb[1] << 16), xb[2] | (xb[3] << 16), xb[4] | (xb[5] << 16), xb[6] | (xb[7] << 16));
    *reinterpret_cast<uint4*>(xc + t * 32 + 16) = make_uint4(xb[8] | (xb[9] << 16), 0u, 0u, 0u);
  }
  __syncthreads();

  bf16_t* orow = out + ((long)b * T0 + t0) * C + wave * 64 + 16 * q;
  const int nft = (nt + 31) >> 5;
  for (int ft = 0; ft < nft; ++ft) {
    const int f0 = 32 * ft + li, f1 = f0 + 16;
    C0U4 x0, x1;
    x0.v = make_uint4(0, 0, 0, 0); x1.v = x0.v;
    if (q < 2) {
      x0.v = *reinterpret_cast<const uint4*>(xc + f0 * 32 + q * 16);
      x1.v = *reinterpret_cast<const uint4*>(xc + f1 * 32 + q * 16);
    }
    unsigned o0[8], o1[8];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const c0_f32x4_t zero = c0_f32x4_t{0.f, 0.f, 0.f, 0.f};
      const c0_f32x4_t y0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct].b, x0.b, zero, 0, 0, 0);
      const c0_f32x4_t y1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct].b, x1.b, zero, 0, 0, 0);
      float z[8], u[8], r[8];
      float2 cell[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        z[i] = fmaf(y0[i], zs[4 * ct + i], zb[4 * ct + i]);
        z[4 + i] = fmaf(y1[i], zs[4 * ct + i], zb[4 * ct + i]);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) u[i] = __builtin_amdgcn_fmed3f(fmaf(z[i], GT_INV_H, -GT_LO * GT_INV_H), 0.f, (float)(GT_N - 1));
#pragma unroll
      for (int i = 0; i < 8; ++i) cell[i] = tab[(int)u[i]];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 8; ++i) r[i] = fmaf(cell[i].x, z[i], cell[i].y);
      o0[2 * ct] = pack_bf16x2(r[0], r[1]); o0[2 * ct + 1] = pack_bf16x2(r[2], r[3]);
      o1[2 * ct] = pack_bf16x2(r[4], r[5]); o1[2 * ct + 1] = pack_bf16x2(r[6], r[7]);
    }
    if (f0 < nt) {
      uint4* p = reinterpret_cast<uint4*>(orow + (long)f0 * C);
      p[0] = make_uint4(o0[0], o0[1], o0[2], o0[3]); p[1] = make_uint4(o0[4], o0[5], o0[6], o0[7]);
    }
    if (f1 < nt) {
      uint4* p = reinterpret_cast<uint4*>(orow + (long)f1 * C);
      p[0] = make_uint4(o1[0], o1[1], o1[2], o1[3]); p[1] = make_uint4(o1[4], o1[5], o1[6], o1[7]);
    }
  }
}

int conv0_gn_fwd_mfma_launch(const void* wav, const void* W, const void* gamma, const void* beta, const float* stats, void* out,
                             long T, int T0, int stride, int B, const float2* tab0, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)conv0_gn_fwd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, C0G_SMEM) != hipSuccess)
      return WL_ELAUNCH;
    attr = true;
  }
  const int nchunk = (T0 + C0L_FR - 1) / C0L_FR;
  WL_LAUNCH(conv0_gn_fwd_mfma_kernel, dim3((unsigned)nchunk, (unsigned)B), dim3(512), C0G_SMEM, st, (const bf16_t*)wav,
            (const bf16_t*)W, (const bf16_t*)gamma, (const bf16_t*)beta, stats, (bf16_t*)out, T, T0, stride, tab0);
  return wl_check_launch();
}

// ---- the Gram pass of the GroupNorm-mode forward on the matrix cores: Q[k] = sum_t x[s t + k], XX[j][k] = sum_t x[s t + j] x[s t + k]
// as X'^T X' with X' = (taps, 1).  bf16 x bf16 products are exact in fp32, so this is the VALU pass's arithmetic in another
// order.  A workgroup covers 4096 frames (the VALU form: 512, two serial LDS loops of 256 frames per sum, 62 us per step and
// 8 x as many partial records for the statistics kernel to walk).  partx[(b * gridDim.x + blockIdx.x)][112]: Q[10], XX[10][10].
#define C0GR_FR 4096
__global__ __launch_bounds__(256) void conv0_gram_mfma_kernel(const bf16_t* __restrict__ wav, float* __restrict__ partx, long T,
                                                              int T0, int stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char c0sm[];
  bf16_t* seg = reinterpret_cast<bf16_t*>(c0sm);
  __shared__ float md[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, q = lane >> 4;
  const int b = blockIdx.y, t0 = blockIdx.x * C0GR_FR;
  const int nt = min(C0GR_FR, T0 - t0);
  const bf16_t* wsrc = wav + (long)b * T + (long)t0 * stride;
  const int nseg = (nt - 1) * stride + C0_KW;
  for (int i = threadIdx.x; i < nseg; i += 256) seg[i] = wsrc[i];
  __syncthreads();
  c0_f32x4_t acc = c0_f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int nst = (nt + 31) >> 5;
  for (int s = wave; s < nst; s += 4) {
    C0U4 f; f.v = make_uint4(0, 0, 0, 0);
    if (li <= C0_KW) {
      unsigned v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int t = 32 * s + 8 * q + e;
        v[e] = t < nt ? (li < C0_KW ? (unsigned)seg[t * stride + li] : 0x3f80u) : 0u;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) f.u[e] = v[2 * e] | (v[2 * e + 1] << 16);
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.b, f.b, acc, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) md[wave][(4 * q + i) * 16 + li] = acc[i];  // D[j][k]: rows j = 4 q + i, column k = li
  __syncthreads();
  if (threadIdx.x < 110) {
    const int idx = threadIdx.x;
    const int o = idx < C0_KW ? C0_KW * 16 + idx : ((idx - C0_KW) / C0_KW) * 16 + (idx - C0_KW) % C0_KW;
    partx[((long)b * gridDim.x + blockIdx.x) * C0_NX + idx] = (md[0][o] + md[1][o]) + (md[2][o] + md[3][o]);
  }
}

// returns the number of partial records per batch row (what conv0_stats_from_gram_kernel walks)
int conv0_gram_mfma_launch(const void* wav, float* partx, long T, int T0, int stride, int B, hipStream_t st, int* nrec) {
  const int nblk = (T0 + C0GR_FR - 1) / C0GR_FR;
  const size_t smem = ((size_t)(C0GR_FR - 1) * stride + C0_KW) * 2 + 16;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)conv0_gram_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024) != hipSuccess)
      return WL_ELAUNCH;
    attr = true;
  }
  WL_LAUNCH(conv0_gram_mfma_kernel, dim3((unsigned)nblk, (unsigned)B), dim3(256), smem, st, (const bf16_t*)wav, partx, T, T0, stride);
  *nrec = nblk;
  return wl_check_launch();
}

// red[slice][C0L_NV] (double) = sum over the workgroups slice, slice + 32, ... of their partial records
__global__ __launch_bounds__(256) void conv0_ln2_reduce_kernel(const float* __restrict__ part, int nblk, double* __restrict__ red) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= C0L_NV) return;
  double s = 0.0;
  for (int blk = blockIdx.y; blk < nblk; blk += C0L_SLICES) s += part[(long)blk * C0L_NV + col];
  red[(long)blockIdx.y * C0L_NV + col] = s;
}

// 16 channels per workgroup: thread (channel cl, row r); rows 0..9 -> dW[c][r], 10 -> d(conv bias), 11 -> dbeta, 12 -> dgamma
__global__ __launch_bounds__(256) void conv0_ln2_finish_kernel(const double* __restrict__ red, const bf16_t* __restrict__ W,
    const bf16_t* __restrict__ cbias, const bf16_t* __restrict__ gamma, bf16_t* __restrict__ dW, bf16_t* __restrict__ dcbias,
    bf16_t* __restrict__ dgamma, bf16_t* __restrict__ dbeta, float gscale) {
  constexpr int C = 512;
  __shared__ double mom[256];
  __shared__ double tot[16][16];
  const int cl = threadIdx.x & 15, r = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
  {
    double s = 0.0;
    for (int sl = 0; sl < C0L_SLICES; ++sl) s += red[(long)sl * C0L_NV + C0L_NROW * C + threadIdx.x];
    mom[threadIdx.x] = s;
  }
  if (r < C0L_NROW) {
    double s = 0.0;
    for (int sl = 0; sl < C0L_SLICES; ++sl) s += red[(long)sl * C0L_NV + (long)r * C + c];
    tot[r][cl] = s;
  }
  __syncthreads();
  if (r <= 10) {
    if (r == 10 && !dcbias) return;
    const int k = r;  // column of the moment matrix: taps 0..9, 10 = the ones column
    const double cb = cbias ? (double)bf2f(cbias[c]) : 0.0;
    double t3 = cb * mom[10 * 16 + k] - (mom[11 * 16 + k] + mom[12 * 16 + k]);
    for (int j = 0; j < C0_KW; ++j) t3 += (double)bf2f(W[(long)c * C0_KW + j]) * mom[j * 16 + k];
    const double v = (double)gscale * ((double)bf2f(gamma[c]) * tot[k][cl] - mom[13 * 16 + k] - t3);
    if (r < 10) dW[(long)c * C0_KW + k] = f2bf((float)v);
    else dcbias[c] = f2bf((float)v);
  } else if (r == 11) {
    dbeta[c] = f2bf((float)((double)gscale * tot[11][cl]));
  } else if (r == 12) {
    dgamma[c] = f2bf((float)((double)gscale * tot[12][cl]));
  }
}

uint64_t conv0_ln_bwd_mfma_workspace_bytes(int B, int T0) {
  const uint64_t nblk = (uint64_t)B * (uint64_t)((T0 + C0L_FR - 1) / C0L_FR);
  return nblk * C0L_NV * sizeof(float) + (uint64_t)C0L_SLICES * C0L_NV * sizeof(double) + 128 * sizeof(float);
}

int conv0_ln_bwd_mfma_launch(const void* wav, const void* W, const void* cbias, const void* gamma, const void* beta, const void* g,
                             void* dW, void* dcbias, void* dgamma, void* dbeta, void* workspace, long T, int T0, int stride, int B,
                             float eps, float gscale, const float2* tab1, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)conv0_ln_bwd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, C0L_SMEM) != hipSuccess)
      return WL_ELAUNCH;
    attr = true;
  }
  const int nchunk = (T0 + C0L_FR - 1) / C0L_FR;
  // chunks per workgroup: one workgroup per CU at a time (LDS), so the pass takes ceil(workgroups / CUs) * cpb chunk times;
  // the largest cpb <= 8 that does not add to that (fewer partial records to write and to reduce)
  int cpb = 1;
  {
    const int cus = 256;
    long best = -1;
    for (int c = 1; c <= 8; ++c) {
      const long wgs = (long)B * ((nchunk + c - 1) / c);
      const long cost = ((wgs + cus - 1) / cus) * c;
      if (best < 0 || cost <= best) { best = cost; cpb = c; }
    }
  }
  const int nx = (nchunk + cpb - 1) / cpb;
  const long nblk = (long)B * nx;
  // workspace: red (double, first: 8-byte aligned) | part | gc
  double* red = (double*)workspace;
  float* part = (float*)(red + (long)C0L_SLICES * C0L_NV);
  float* gc = part + nblk * C0L_NV;
  WL_LAUNCH(conv0_ln_gram_kernel, dim3(1), dim3(1024), 0, st, (const bf16_t*)W, (const bf16_t*)cbias, gc, 512);
  WL_LAUNCH(conv0_ln_bwd_mfma_kernel, dim3((unsigned)nx, (unsigned)B), dim3(512), C0L_SMEM, st, (const bf16_t*)wav,
            (const bf16_t*)W, (const bf16_t*)cbias, (const bf16_t*)gamma, (const bf16_t*)beta, (const bf16_t*)g, gc, part, T, T0,
            stride, eps, tab1, nchunk, cpb);
  int rc = wl_check_launch();
  if (rc != WL_OK) return rc;
  WL_LAUNCH(conv0_ln2_reduce_kernel, dim3((C0L_NV + 255) / 256, C0L_SLICES), dim3(256), 0, st, part, (int)nblk, red);
  WL_LAUNCH(conv0_ln2_finish_kernel, dim3(512 / 16), dim3(256), 0, st, red, (const bf16_t*)W, (const bf16_t*)cbias,
            (const bf16_t*)gamma, (bf16_t*)dW, (bf16_t*)dcbias, (bf16_t*)dgamma, (bf16_t*)dbeta, gscale);
  return wl_check_launch();
}
