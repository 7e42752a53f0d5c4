// Kernel-side GEMM parameter block + the shared epilogue (bias / GELU / GELU' / residual / accumulate).
#pragma once
#include <stdlib.h>
#include "common.hpp"
#include "../../include/wavlm_hip.h"

// one problem of a grouped split-K launch (gemm_pp.hip): operands, slab base and tile geometry; vbase = first work item
struct GemmGrp { const void* A; const void* B; float* ws; long lda, ldb; int M, N, tiles_m, tiles_n, vbase;
                 // the member's final epilogue (in-kernel split-K fix-up, gemm_w4.hip): C (+)= alpha * sum of the partial sums
                 void* C; long ldc; int c_dtype, accumulate; float alpha; };

struct GemmP {
  const void* A; const void* B; void* C;
  int M, N, K, KB;
  long lda, ldb, ldc, sA_kb, sB_kb;
  int batch_i;
  long sA_o, sA_i, sB_o, sB_i, sC_o, sC_i;
  float alpha; int epi; int c_dtype;
  const void* bias; int bias_dtype; long sBias_o, sBias_i;
  void* aux; int aux_dtype; long ld_aux, sAux_o, sAux_i;
  const void* res; int res_dtype; long ld_res, sRes_o, sRes_i;
  int accumulate;
  int split_k; float* ws;
  int tiles_m, tiles_n;
  int vtotal, nbatch, skew;  // persistent launch (ping-pong kernels): virtual block count, batch count, start skew
  int swz_r;                 // > 0: tile ids walk super-rows of swz_r tile rows column by column (gemm_tile_rc)
  int patch_m;               // split-K work order: tile-row patch height (0: tiles only, split outermost)
  int ngrp;                  // > 0: grouped launch, the problems are grp[0 .. ngrp) (A/B/M/N/lda/ldb/ws above are unused)
  // fix_epoch != 0 (grouped launch of gemm_w4_kernel): in-kernel split-K fix-up -- the LAST workgroup of a tile to finish its K
  // range adds the other splits' slabs to its own partial sum and writes the final result; no reduction launch follows
  unsigned fix_epoch; unsigned* fix_cnt; unsigned* fix_flag;
  int sk_ks, sk_wgs, sk_tiles, sk_s, sk_lm, sk_spread;  // sk_ks > 0: balanced grouped launch (gemm_sk_plan): K steps per tile, workgroups, tiles of all members, main splits, main run
  const float4* gtab;        // GELU / GELU' chord table in global memory (fast epilogue 3 copies it to LDS), or null
  float* colsum_part;        // fused column sums of C (ping-pong kernels, fast epilogues): partial rows [tiles_m][N], or null
  GemmGrp grp[4];
};

// Balanced grouped weight-gradient launch.  T tiles (all members), KS K steps each, G workgroups, s = G / T: the one-round
// split runs T * s workgroups and leaves R = G - T * s CUs idle (Base: 108 x 2 = 216 of 256).  Here the T * s MAIN workgroups
// take the K steps [j Lm, (j + 1) Lm) of their tile, j < s -- in the same order and lockstep as the one-round split, which is
// what lets the tiles of a row / column share their operand panels in L2 -- and the R TAIL workgroups the last
// KS - s Lm steps of ceil(T / R) tiles each, one after the other; Lm is chosen so that both kinds finish together
// (c = what a tile's prologue + epilogue cost, in K steps).  Every tile has s + 1 partial sums (slabs).
// (A plain stream-K partition -- the tile-major line of K steps cut into G equal runs -- was built first and measured 40 %
//  SLOWER than the one-round split, profiles/r04/ab_streamk_tile_major.txt: neighbouring workgroups then work at different
//  K offsets, no operand panel is ever shared, and the launch becomes HBM-bound at 2.65 GB instead of 0.59 GB.)
#define GEMM_SK_SEG_COST 8
struct GemmSk { int s, R, q, Lm; };   // main splits, tail workgroups, tiles per tail workgroup (max), main run length
static inline bool gemm_sk_plan(long T, long KS, int G, GemmSk& k) {
  if (T <= 0 || T >= G) return false;
  k.s = (int)(G / T); k.R = (int)(G - T * k.s);
  if (k.R == 0 || k.s < 1) return false;
  k.q = (int)((T + k.R - 1) / k.R);
  static const long c_env = [] { const char* e = getenv("WAVLM_SK_SEG_COST"); return e && *e ? atol(e) : 0l; }();   // lab switch
  const long c = c_env > 0 ? c_env : GEMM_SK_SEG_COST;
  const long L0 = (k.q * (KS + c) - c) / (1 + (long)k.s * k.q);   // main run == tail workgroup's total, rounded down
  long Lm = 0, best = 0;
  for (long L = L0; L <= L0 + 1; ++L) {
    if (L < 8 || KS - L * k.s < 1) continue;
    const long tail = k.q * (KS - L * k.s + c), cost = L + c > tail ? L + c : tail;
    if (Lm == 0 || cost < best) { Lm = L; best = cost; }
  }
  if (Lm == 0 || best * 100 > 97 * ((KS + k.s - 1) / k.s + c)) return false;   // not worth a third kind of slab
  k.Lm = (int)Lm;
  return true;
}

// ---- tile order inside an XCD (round 5) ----
// A persistent launch hands XCD x a contiguous run of tile ids, and per round the XCD's 32 CUs work on 32 CONSECUTIVE ids at
// the same time -- they progress through K roughly in lockstep, so an operand panel is fetched into the XCD's L2 once per
// round for all the tiles that share it.  With row-major ids the 32 tiles are (32 / tiles_n) rows x tiles_n columns: at
// N = 4096 (16 column tiles of 256) two A panels and ALL sixteen B panels per round -- the PMC pass of round 5 shows 1283 MB
// fetched per fc1 / fc2-dX launch of WavLM-Large against 598 MB algorithmic (2.1 x), 3.4 TB/s: those launches are bound by
// that traffic.  Walking super-rows of R tile rows column by column makes the 32 tiles an R x (32 / R) block: the panels per
// round drop from 2 + 16 to 4 + 8 (N = 4096), from 4 + 8 to 8 + 4 at 192 x 384 tiles and N = 3072 (the B panels are twice as
// tall there).  R is chosen on the host (gemm_pick_swizzle) to minimise panel bytes per round; 0 = the row-major order.
__host__ __device__ inline void gemm_tile_rc(int tile, int tiles_m, int tiles_n, int R, int& tm, int& tn) {
  if (R <= 1) { tm = tile / tiles_n; tn = tile - tm * tiles_n; return; }
  const int per = R * tiles_n;
  const int sr = tile / per;
  int rows = tiles_m - sr * R; if (rows > R) rows = R;
  const int w = tile - sr * per;
  tn = w / rows; tm = sr * R + (w - tn * rows);
}
// R in {1, 2, 4, 8, 16, 32} minimising (tile rows x R + tile columns x min(32 / R, tiles_n)) -- the operand rows an XCD fetches
// per round; ties go to the smaller R (R = 1 is the row-major order itself)
// Only where the B operand as a whole does not fit an XCD's 4 MiB L2 (b_bytes >= 6 MB: measured -- WavLM-Large's N = 4096 /
// 3072 at K = 1024, 8 / 6 MB: fetch -13 ... -26 %, launches -3 ... -6 %; WavLM-Base's N = 3072 at K = 768, 4.7 MB: B stays
// resident either way, fetch +4 %, time unchanged; profiles/r05/gemm_swizzle.txt).
static inline int gemm_pick_swizzle(int tile_m, int tile_n, int tiles_m, int tiles_n, long b_bytes) {
  static const int off = getenv("WAVLM_GEMM_SWIZZLE") && getenv("WAVLM_GEMM_SWIZZLE")[0] == '0';
  if (off || tiles_n <= 1 || tiles_m <= 1 || b_bytes < 6l * 1000 * 1000) return 0;
  long best = -1; int br = 0;
  for (int R = 1; R <= 32; R *= 2) {
    if (R > tiles_m) break;
    int cols = (32 + R - 1) / R; if (cols > tiles_n) cols = tiles_n;
    // rows actually covered when the block is narrower than 32 / R columns: 32 tiles = rows x cols
    int rows = R; if (cols * R < 32) { rows = (32 + cols - 1) / cols; if (rows > tiles_m) rows = tiles_m; }
    const long cost = (long)tile_m * rows + (long)tile_n * cols;
    if (best < 0 || cost < best) { best = cost; br = R; }
  }
  return br <= 1 ? 0 : br;
}

static inline GemmP make_gemm_params(const wavlm_gemm_desc* d) {
  GemmP p;
  p.A = d->A; p.B = d->B; p.C = d->C;
  p.M = d->M; p.N = d->N; p.K = d->K; p.KB = d->KB < 1 ? 1 : d->KB;
  p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.sA_kb = d->sA_kb; p.sB_kb = d->sB_kb;
  p.batch_i = d->batch_i < 1 ? 1 : d->batch_i;
  p.sA_o = d->sA_o; p.sA_i = d->sA_i; p.sB_o = d->sB_o; p.sB_i = d->sB_i; p.sC_o = d->sC_o; p.sC_i = d->sC_i;
  p.alpha = d->alpha; p.epi = d->epi; p.c_dtype = d->c_dtype;
  p.bias = d->bias; p.bias_dtype = d->bias_dtype; p.sBias_o = d->sBias_o; p.sBias_i = d->sBias_i;
  p.aux = d->aux; p.aux_dtype = d->aux_dtype; p.ld_aux = d->ld_aux; p.sAux_o = d->sAux_o; p.sAux_i = d->sAux_i;
  p.gtab = nullptr;
  p.colsum_part = nullptr;
  p.res = d->res; p.res_dtype = d->res_dtype; p.ld_res = d->ld_res; p.sRes_o = d->sRes_o; p.sRes_i = d->sRes_i;
  p.accumulate = d->accumulate;
  p.split_k = d->split_k < 1 ? 1 : d->split_k;
  p.ws = (float*)d->workspace;
  p.tiles_m = 0; p.tiles_n = 0; p.vtotal = 0; p.nbatch = 1; p.skew = 0; p.patch_m = 0; p.ngrp = 0;
  p.sk_ks = 0; p.sk_wgs = 0; p.sk_tiles = 0; p.sk_s = 0; p.sk_lm = 0; p.sk_spread = 0;
  p.swz_r = 0;
  p.fix_epoch = 0; p.fix_cnt = nullptr; p.fix_flag = nullptr;
  return p;
}

// final epilogue for output element (m, n) of batch (zo, zi); acc is the raw fp32 contraction
__device__ __forceinline__ void gemm_epi_final(const GemmP& p, int zo, int zi, int m, int n, float acc) {
  float v = p.alpha * acc;
  if (p.bias) v += ld_elem(p.bias, (long)zo * p.sBias_o + (long)zi * p.sBias_i + n, p.bias_dtype);
  if (p.epi == 1) {
    if (p.aux) st_elem(p.aux, (long)zo * p.sAux_o + (long)zi * p.sAux_i + (long)m * p.ld_aux + n, p.aux_dtype, v);
    v = gelu_f(v);
  } else if (p.epi == 2) {
    const float u = ld_elem(p.aux, (long)zo * p.sAux_o + (long)zi * p.sAux_i + (long)m * p.ld_aux + n, p.aux_dtype);
    v *= gelu_grad_f(u);
  } else if (p.epi == 3) {
    float gr;
    v = gelu_both_f(v, gr);
    if (p.aux) st_elem(p.aux, (long)zo * p.sAux_o + (long)zi * p.sAux_i + (long)m * p.ld_aux + n, p.aux_dtype, gr);
  } else if (p.epi == 4) {
    v *= ld_elem(p.aux, (long)zo * p.sAux_o + (long)zi * p.sAux_i + (long)m * p.ld_aux + n, p.aux_dtype);
  }
  if (p.res) v += ld_elem(p.res, (long)zo * p.sRes_o + (long)zi * p.sRes_i + (long)m * p.ld_res + n, p.res_dtype);
  const long ci = (long)zo * p.sC_o + (long)zi * p.sC_i + (long)m * p.ldc + n;
  if (p.accumulate) v += ld_elem(p.C, ci, p.c_dtype);
  st_elem(p.C, ci, p.c_dtype, v);
}

// store from a main kernel: either a raw fp32 slab (split-K) or the final epilogue
__device__ __forceinline__ void gemm_store(const GemmP& p, int z, int split, int m, int n, float acc) {
  if (p.split_k > 1) {
    p.ws[((long)z * p.split_k + split) * ((long)p.M * p.N) + (long)m * p.N + n] = acc;
  } else {
    gemm_epi_final(p, z / p.batch_i, z % p.batch_i, m, n, acc);
  }
}


// ---- 8-wide row-vector epilogue (all row starts 16-byte aligned; see vec_epilogue_ok) ---------------------------
__device__ __forceinline__ void ld8_dt(const void* p, long idx, int dt, float (&v)[8]) {
  if (dt == WL_F32) {
    const float4 a = *reinterpret_cast<const float4*>((const float*)p + idx);
    const float4 b = *reinterpret_cast<const float4*>((const float*)p + idx + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
    const uint4 a = *reinterpret_cast<const uint4*>((const bf16_t*)p + idx);
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
    v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
  }
}
__device__ __forceinline__ void st8_dt(void* p, long idx, int dt, const float (&v)[8]) {
  if (dt == WL_F32) {
    *reinterpret_cast<float4*>((float*)p + idx) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>((float*)p + idx + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    uint4 o;
    o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
    o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>((bf16_t*)p + idx) = o;
  }
}

// columns n .. n+7 of output row m (n % 8 == 0).  A chunk that straddles N falls back to the scalar path.
__device__ __forceinline__ void gemm_store8(const GemmP& p, int zo, int zi, int z, int split, int m, int n,
                                            float (&acc)[8]) {
  if (n + 8 > p.N) {
    for (int e = 0; e < 8 && n + e < p.N; ++e) gemm_store(p, z, split, m, n + e, acc[e]);
    return;
  }
  if (p.split_k > 1) {
    float* dst = p.ws + ((long)z * p.split_k + split) * ((long)p.M * p.N) + (long)m * p.N + n;
    *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4*>(dst + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    return;
  }
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = p.alpha * acc[e];
  if (p.bias) {
    float b[8];
    ld8_dt(p.bias, (long)zo * p.sBias_o + (long)zi * p.sBias_i + n, p.bias_dtype, b);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += b[e];
  }
  if (p.epi == 1) {
    if (p.aux) st8_dt(p.aux, (long)zo * p.sAux_o + (long)zi * p.sAux_i + (long)m * p.ld_aux + n, p.aux_dtype, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
  } else if (p.epi == 2) {
    float u[8];
    ld8_dt(p.aux, (long)zo * p.sAux_o + (long)zi * p.sAux_i + (long)m * p.ld_aux + n, p.aux_dtype, u);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= gelu_grad_f(u[e]);
  } else if (p.epi == 3) {
    float gr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = gelu_both_f(v[e], gr[e]);
    if (p.aux) st8_dt(p.aux, (long)zo * p.sAux_o + (long)zi * p.sAux_i + (long)m * p.ld_aux + n, p.aux_dtype, gr);
  } else if (p.epi == 4) {
    float u[8];
    ld8_dt(p.aux, (long)zo * p.sAux_o + (long)zi * p.sAux_i + (long)m * p.ld_aux + n, p.aux_dtype, u);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= u[e];
  }
  if (p.res) {
    float r[8];
    ld8_dt(p.res, (long)zo * p.sRes_o + (long)zi * p.sRes_i + (long)m * p.ld_res + n, p.res_dtype, r);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += r[e];
  }
  const long ci = (long)zo * p.sC_o + (long)zi * p.sC_i + (long)m * p.ldc + n;
  if (p.accumulate) {
    float c[8];
    ld8_dt(p.C, ci, p.c_dtype, c);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += c[e];
  }
  st8_dt(p.C, ci, p.c_dtype, v);
}

// Specialised 8-wide store for the cases that carry the step (bf16 C, bf16 bias / aux / residual, no accumulate, no
// split-K): KIND 0 = alpha * acc (+ bias); 3 = + GELU with GELU'(pre-activation) stored to aux; 4 = * aux.
// The generic gemm_store8 decides all of that per 8-element chunk at run time; with one block per CU the epilogue is
// exposed, and the probe in tools/gemm_epi_probe.py showed it bound by the CU's own instruction stream (7.5 us per tile
// even with 8 CUs active), not by the HBM write burst.
// GELU and GELU' together from one 8-byte LDS read: cell i of the table holds the chord (slope a, intercept b) of the normal
// CDF Phi over [x_i, x_i + h), h = 16 / GT2_N, x_0 = -8:  Phi(x) ~ a x + b,  gelu(x) = x Phi(x),  gelu'(x) = Phi(x) + x phi(x)
// with phi = Phi' ~ a (the chord's slope is phi at the cell's midpoint).  Errors at 4096 cells: Phi 4.6e-7, gelu 1e-6 absolute;
// gelu' <= |x^2 phi(x)| h / 2 = 5.7e-4 absolute (a seventh of bf16's half ulp at 1: the outputs here are bf16).  The table is
// the same 32 KiB as the float4 form it replaces (chords of gelu AND gelu' per cell, 2048 cells: 16 bytes per lookup -- the
// random 16-byte reads of eight waves were half of the epilogue's LDS time).  The erf evaluation (v_rcp, v_exp, ~12 more
// VALU) on every element of fc1's and the conv stack's outputs ran in the EXPOSED epilogue of one-block-per-CU GEMMs: ~45 us
// of a 199 us fc1 launch.
#define GT4_N 2048            // float4 cells of the table's storage (32 KiB)
#define GT2_N (2 * GT4_N)     // chord cells
__device__ __forceinline__ float gelu_both_tab(const float4* tab4, float x, float& grad) {
  const float2* tab = reinterpret_cast<const float2*>(tab4);
  float u = fmaf(x, GT2_N / 16.0f, 8.0f * (GT2_N / 16.0f));
  u = __builtin_amdgcn_fmed3f(u, 0.f, (float)(GT2_N - 1));
  const float2 t = tab[(int)u];
  const float ph = fmaf(t.x, x, t.y);
  grad = fmaf(x, t.x, ph);
  return x * ph;
}
const float4* wl_gelu_tab4(hipStream_t st);  // device address of the table (gemm_bf16.hip; filled on first use)
// every thread of the block: global table -> LDS (32 KiB); the caller synchronises before and after
__device__ __forceinline__ void gelu_tab_stage(const float4* g, float4* lds) {
  for (int i = threadIdx.x; i < GT4_N; i += blockDim.x) lds[i] = g[i];
}

// vout (optional): receives the eight final values as stored (before rounding to bf16) -- fused column sums
// aux_pre (KIND 4, optional): the eight bf16 aux values of this chunk, already loaded by the caller (gemm_pp3's epilogue
// issues a 32-row block's six loads before it stages the block: one round trip per block instead of one per chunk)
// bias_pre (optional): the chunk's eight bias values as floats (an LDS address); res_pre (optional): its eight bf16 residual
// values, already loaded by the caller
__device__ __forceinline__ void gemm_unpack8(const uint4& a, float (&u)[8]) {
  u[0] = __uint_as_float(a.x << 16); u[1] = __uint_as_float(a.x & 0xffff0000u);
  u[2] = __uint_as_float(a.y << 16); u[3] = __uint_as_float(a.y & 0xffff0000u);
  u[4] = __uint_as_float(a.z << 16); u[5] = __uint_as_float(a.z & 0xffff0000u);
  u[6] = __uint_as_float(a.w << 16); u[7] = __uint_as_float(a.w & 0xffff0000u);
}
template <int KIND>
__device__ __forceinline__ void gemm_store8_fast(const GemmP& p, int zo, int zi, int m, int n, const float (&acc)[8],
                                                 const float4* tab = nullptr, float* vout = nullptr, const uint4* aux_pre = nullptr,
                                                 const float* bias_pre = nullptr, const uint4* res_pre = nullptr) {
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = p.alpha * acc[e];
  if (p.bias) {
    float b[8];
    if (bias_pre) {
      const float4 b0 = *reinterpret_cast<const float4*>(bias_pre), b1 = *reinterpret_cast<const float4*>(bias_pre + 4);
      b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
    } else ld8_dt(p.bias, (long)zo * p.sBias_o + (long)zi * p.sBias_i + n, WL_BF16, b);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += b[e];
  }
  if constexpr (KIND == 3) {
    float gr[8];
    if (tab) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = gelu_both_tab(tab, v[e], gr[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = gelu_both_f(v[e], gr[e]);
    }
    if (p.aux) st8_dt(p.aux, (long)zo * p.sAux_o + (long)zi * p.sAux_i + (long)m * p.ld_aux + n, WL_BF16, gr);
  } else if constexpr (KIND == 4) {
    float u[8];
    if (aux_pre) {
      gemm_unpack8(*aux_pre, u);
    } else {
      ld8_dt(p.aux, (long)zo * p.sAux_o + (long)zi * p.sAux_i + (long)m * p.ld_aux + n, WL_BF16, u);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= u[e];
  }
  if (p.res) {  // bf16 residual (uniform branch): the gradient another consumer of the same tensor produced
    float r[8];
    if (res_pre) gemm_unpack8(*res_pre, r);
    else ld8_dt(p.res, (long)zo * p.sRes_o + (long)zi * p.sRes_i + (long)m * p.ld_res + n, WL_BF16, r);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += r[e];
  }
  st8_dt(p.C, (long)zo * p.sC_o + (long)zi * p.sC_i + (long)m * p.ldc + n, WL_BF16, v);
  if (vout) {
#pragma unroll
    for (int e = 0; e < 8; ++e) vout[e] = v[e];
  }
}

// Fused column sums of C (the bias gradient of the linear whose output gradient this GEMM produces, e.g. fc1's from the
// GELU'-multiplying dX GEMM of fc2: WavLM/WavLM.py:732-737): the row-vector epilogue writes each final 8-vector back into
// the wave's staging slice (zeros for rows / columns past the edge); after a 32-row block lane c adds up column c of the
// slice.  NCOL = columns per wave (64 or 96), LD = slice row stride in floats.
template <int NCOL, int LD>
__device__ __forceinline__ void gemm_colsum_block(const float* ep, int lane, float (&cs)[2]) {
  asm volatile("" ::: "memory");  // same wave: LDS executes in order, the compiler must keep the order too
  float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
#pragma unroll
  for (int r = 0; r < 32; r += 2) { a0 += ep[r * LD + lane]; a1 += ep[(r + 1) * LD + lane]; }
  cs[0] += a0 + a1;
  if constexpr (NCOL > 64) {
    if (lane < NCOL - 64) {
#pragma unroll
      for (int r = 0; r < 32; r += 2) { b0 += ep[r * LD + 64 + lane]; b1 += ep[(r + 1) * LD + 64 + lane]; }
      cs[1] += b0 + b1;
    }
  }
  asm volatile("" ::: "memory");
}
// the two waves that share a column range (wm = 0, 1) meet in LDS; wm = 0 writes the tile's partial row segment
template <int NCOL>
__device__ __forceinline__ void gemm_colsum_finish(const GemmP& p, float* xch, int wave, int wm, int lane, int tm, int nw,
                                                   const float (&cs)[2]) {
  float* mine = xch + wave * 128;
  mine[lane] = cs[0];
  if (NCOL > 64 && lane < NCOL - 64) mine[64 + lane] = cs[1];
  __syncthreads();
  if (wm == 0) {
    const float* other = xch + (wave + 4) * 128;  // waves are numbered wm * 4 + wn
    float* dst = p.colsum_part + (long)tm * p.N;
    if (nw + lane < p.N) dst[nw + lane] = cs[0] + other[lane];
    if (NCOL > 64 && lane < NCOL - 64 && nw + 64 + lane < p.N) dst[nw + 64 + lane] = cs[1] + other[64 + lane];
  }
}

// epilogue class of a launch: 0 scalar stores, 1 generic row vectors, 2 / 3 / 4 specialised for epi 0 / 3 / 4
static inline int gemm_epilogue_class(const wavlm_gemm_desc* d, bool vec) {
  if (!vec) return 0;
  const bool fast = d->split_k <= 1 && (!d->res || (d->res_dtype == WL_BF16 && d->ld_res % 8 == 0)) && !d->accumulate &&
                    d->c_dtype == WL_BF16 && d->N % 8 == 0 &&
                    (!d->bias || d->bias_dtype == WL_BF16) &&
                    (d->epi == 0 || (d->epi == 3 && (!d->aux || d->aux_dtype == WL_BF16)) ||
                     (d->epi == 4 && d->aux && d->aux_dtype == WL_BF16));
  if (!fast) return 1;
  return d->epi == 0 ? 2 : d->epi == 3 ? 3 : 4;
}

// flattened reduction-tile range [t0, t1) owned by split `s` (tiles = KB * ceil(K / BK))
__device__ __forceinline__ void gemm_split_range(int total_tiles, int split_k, int s, int& t0, int& t1) {
  const int per = (total_tiles + split_k - 1) / split_k;
  t0 = s * per; t1 = t0 + per; if (t1 > total_tiles) t1 = total_tiles; if (t0 > t1) t0 = t1;
}
