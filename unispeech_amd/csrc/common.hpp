// Shared device helpers for the WavLM hot-path kernels (gfx950 / CDNA4 only).
// Wave = 64 lanes everywhere in this tree; nothing here is written for 32-wide warps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <mutex>

#define WL_MAX_DEVICES 64  // per-device lazily initialised tables (one process may drive several GPUs)

#define WL_OK 0
#define WL_EINVAL (-1)
#define WL_ELAUNCH (-2)
#define WL_ENOMEM (-3)

enum { WL_F32 = 0, WL_BF16 = 1 };

typedef unsigned short bf16_t;  // raw bf16 bits

__device__ __forceinline__ float bf2f(bf16_t v) {
  return __uint_as_float(((unsigned)v) << 16);
}
// fp32 -> bf16, round-to-nearest-even: gfx950 has v_cvt_pk_bf16_f32; a C++ conversion to __bf16 selects it
__device__ __forceinline__ bf16_t f2bf(float f) {
  const __bf16 h = (__bf16)f;
  return __builtin_bit_cast(bf16_t, h);
}
// two fp32 -> packed bf16x2 (a in the low half): one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

// typed scalar access by runtime dtype (0 = f32, 1 = bf16)
__device__ __forceinline__ float ld_elem(const void* p, long i, int dt) {
  return dt == WL_F32 ? ((const float*)p)[i] : bf2f(((const bf16_t*)p)[i]);
}
__device__ __forceinline__ void st_elem(void* p, long i, int dt, float v) {
  if (dt == WL_F32) ((float*)p)[i] = v; else ((bf16_t*)p)[i] = f2bf(v);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// ---- 64-lane wave reductions ----
// Within a 16-lane row: four DPP adds (quad_perm xor 1, xor 2, row_half_mirror, row_mirror -- the compiler folds each into
// one v_add_f32_dpp); across the four rows: v_readlane of one lane per row and three adds on uniform values.  ~12
// instructions and ~60 cycles of latency.  The former butterfly over __shfl_xor compiled to six DEPENDENT ds_bpermute_b32
// (LDS round trips, ~900 cycles per reduction): with four reductions per frame and one wave per SIMD that was the whole
// run time of the LayerNorm-mode conv0 backward (4.2 us per frame).  All 64 lanes must be active (every caller reduces
// with the full wave); the result is wave-uniform.
template <int CTRL> __device__ __forceinline__ float wl_dpp_f32(float v) {  // lanes whose source is invalid read 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ float wl_dpp_keep_f32(float v) {  // ... keep their own value
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wl_lane_f32(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += wl_dpp_f32<0xB1>(v);    // quad_perm [1,0,3,2]
  v += wl_dpp_f32<0x4E>(v);    // quad_perm [2,3,0,1]
  v += wl_dpp_f32<0x141>(v);   // row_half_mirror: lane i <-> 7 - i, the other quad of its 8
  v += wl_dpp_f32<0x140>(v);   // row_mirror: lane i <-> 15 - i, the other half of its row
  return (wl_lane_f32(v, 0) + wl_lane_f32(v, 16)) + (wl_lane_f32(v, 32) + wl_lane_f32(v, 48));
}
// sum over each aligned group of 8 lanes, in every lane of the group (three DPP adds)
__device__ __forceinline__ float wl_sum8(float v) {
  v += wl_dpp_f32<0xB1>(v);
  v += wl_dpp_f32<0x4E>(v);
  v += wl_dpp_f32<0x141>(v);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, wl_dpp_keep_f32<0xB1>(v));
  v = fmaxf(v, wl_dpp_keep_f32<0x4E>(v));
  v = fmaxf(v, wl_dpp_keep_f32<0x141>(v));
  v = fmaxf(v, wl_dpp_keep_f32<0x140>(v));
  return fmaxf(fmaxf(wl_lane_f32(v, 0), wl_lane_f32(v, 16)), fmaxf(wl_lane_f32(v, 32), wl_lane_f32(v, 48)));
}
// exchange between the two 32-lane halves of a wave with gfx950's v_permlane32_swap_b32 (VALU, a few cycles) instead of
// ds_bpermute_b32 (an LDS round trip): lo / hi = the value of lane (l & 31) / (l & 31) + 32 in every lane.  Inline asm with
// its own wait states (through the builtin the compiler folded the two results of a swap of two copies of one value).
__device__ __forceinline__ void wl_halves(float x, float& lo, float& hi) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  lo = a; hi = b;
}
__device__ __forceinline__ float wl_sum_xor32(float x) { float lo, hi; wl_halves(x, lo, hi); return lo + hi; }
__device__ __forceinline__ float wl_max_xor32(float x) { float lo, hi; wl_halves(x, lo, hi); return fmaxf(lo, hi); }
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- erf GELU, as torch.nn.functional.gelu(approximate='none') ----
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute, branch-free: one v_rcp, one v_exp, five FMAs)
// instead of libm's erff (two polynomial branches, ~2x the VALU work -- it showed as ~30 % of an fc1 GEMM launch).
// z = x / sqrt(2); returns erf(z) and e = exp(-z^2) = exp(-x^2 / 2), which the derivative reuses.
__device__ __forceinline__ float erf_as(float z, float& e) {
  const float a = fabsf(z);
  const float t = __frcp_rn(fmaf(0.3275911f, a, 1.0f));
  float pl = fmaf(1.061405429f, t, -1.453152027f);
  pl = fmaf(pl, t, 1.421413741f);
  pl = fmaf(pl, t, -0.284496736f);
  pl = fmaf(pl, t, 0.254829592f);
  e = __expf(-a * a);
  return copysignf(fmaf(-pl * t, e, 1.0f), z);
}
__device__ __forceinline__ float gelu_f(float x) {
  float e;
  return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f, e));
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  float e;
  const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752440f, e));
  return fmaf(x * 0.39894228040143267794f, e, cdf);
}
// value and derivative from one erf / exp evaluation (forward epilogues that store gelu'(pre-activation) for backward)
__device__ __forceinline__ float gelu_both_f(float x, float& grad) {
  float e;
  const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752440f, e));
  grad = fmaf(x * 0.39894228040143267794f, e, cdf);
  return x * cdf;
}

// ---- cheap stateless dropout masks -------------------------------------------------------------------------------
// word(row, col) = drop_mix(row_word + col_word): the two words are strong multiplicative hashes (3 x v_mul_lo_u32,
// quarter rate) evaluated once per row / once per column and reused; per element only the multiply-free drop_mix runs
// (one xor-shift + one 24-bit multiply, full rate).  Each 32-bit word yields two keep decisions (16-bit halves >= th16).
// Philox4x32-10 (below) costs ~80 quarter-rate multiplies per
// 8 elements and was ~80 % of the LayerNorm kernels' time with dropout on.
__device__ __forceinline__ unsigned hash32(unsigned x) {
  x *= 0x9E3779B1u;
  x ^= x >> 15; x *= 0x85EBCA77u;
  x ^= x >> 13; x *= 0xC2B2AE3Du;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ unsigned drop_mix(unsigned x) {
  // one xor-shift (a single SDWA instruction) + one FULL-RATE 24-bit multiply (v_mul_u32_u24): 2 VALU slots instead of
  // the 5 of the former two-round xorshift/shift-add mix -- and better mixed: on 4096 rows x 800 decisions the
  // column-pair |correlation| of the decisions is 0.0125 mean / 0.09 max (sampling noise 0.0156; the old mix: 0.024 /
  // 0.98 -- column words that happened to lie close gave near-identical masks), row pairs 0.028 mean / 0.22 max
  // (noise 0.035; old 0.039 / 0.33).  tools: the numpy experiment is summarised in DESIGN.md section 4.2.
  x ^= x >> 16;
  // inline asm: through __umul24 the compiler distributes the implied 24-bit mask over the xor and spends a third
  // instruction (v_and_b32) on bits the hardware multiply ignores anyway
  unsigned r;
  asm("v_mul_u32_u24 %0, 0xb5297b, %1" : "=v"(r) : "v"(x));
  return r;
}
static inline unsigned drop_thresh16(float p) {  // keep iff 16-bit half >= th; 0 = dropout off
  if (p <= 0.f) return 0u;
  double t = (double)p * 65536.0 + 0.5;
  if (t > 65535.0) t = 65535.0;
  return t < 1.0 ? 1u : (unsigned)t;
}
static inline float drop_scale16(unsigned th16) { return th16 ? (float)(1.0 / (1.0 - (double)th16 / 65536.0)) : 1.f; }

// ---- Philox4x32-10 counter RNG (dropout masks are regenerated in backward, never stored) ----
struct Philox4 { unsigned x, y, z, w; };
__device__ __forceinline__ Philox4 philox4x32_10(unsigned long long seed, unsigned long long ctr) {
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
  unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = 0x243F6A88u, c3 = 0x85A308D3u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
    const unsigned n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    const unsigned n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  Philox4 o; o.x = c0; o.y = c1; o.z = c2; o.w = c3; return o;
}
// keep-mask for element index `idx` of a tensor: one Philox call covers 4 consecutive elements.
// keep iff u32 >= thresh where thresh = p * 2^32.
__device__ __forceinline__ bool dropout_keep(unsigned long long seed, unsigned long long idx, unsigned thresh) {
  const Philox4 r = philox4x32_10(seed, idx >> 2);
  const unsigned sel = (unsigned)(idx & 3);
  const unsigned v = sel == 0 ? r.x : sel == 1 ? r.y : sel == 2 ? r.z : r.w;
  return v >= thresh;
}

// Launch-status bookkeeping.  hipGetLastError() is per-thread *sticky state shared with every other HIP user in the
// process* (torch's allocator leaves e.g. hipErrorNotReady behind after an event query), so each launch first
// clears it, launches, and records only its own status; an entry point reports the union of its launches.
static thread_local int wl_launch_failed = 0;
#define WL_LAUNCH(...)                                            \
  do {                                                            \
    (void)hipGetLastError();                                      \
    hipLaunchKernelGGL(__VA_ARGS__);                              \
    const hipError_t wl_e_ = hipGetLastError();                   \
    if (wl_e_ != hipSuccess) {                                    \
      wl_launch_failed = 1;                                       \
      fprintf(stderr, "[wavlm_hip] kernel launch failed: %s (%s:%d)\n", hipGetErrorString(wl_e_), __FILE__, __LINE__); \
    }                                                             \
  } while (0)
// per-launch timing of a kernel class while wavlm_prof_enable(1) (gemm_bf16.hip): HIP events on the launch stream
int wl_prof_begin(int cls, int dtype, double flops, double bytes, hipStream_t st);
void wl_prof_end(int i, hipStream_t st);
struct WlProfScope {  // events around everything the enclosing entry point launches after this line
  int i; hipStream_t st;
  WlProfScope(int cls, int dtype, double flops, double bytes, hipStream_t s) : i(wl_prof_begin(cls, dtype, flops, bytes, s)), st(s) {}
  ~WlProfScope() { wl_prof_end(i, st); }
};

static inline int wl_check_launch() {
  const int e = wl_launch_failed;
  wl_launch_failed = 0;
  return e ? WL_ELAUNCH : WL_OK;
}

// layer.hip: tell the registered gradient listener (wavlm_dp_set_listener) that an accumulation into the caller's gradient
// buffer [base, base + bytes) has been enqueued on `stream`; a no-op without a listener
void wl_notify_grad(const void* base, uint64_t bytes, void* stream);
static inline uint64_t wl_esize(int dt) { return dt == WL_BF16 ? 2 : 4; }
// attn_fused.hip: wavlm_attn_fused_bwd with dtab (+)= when dtab_accumulate != 0
// (pstore: the forward's probability store, wavlm_attn_fused_fwd_p, or NULL = recompute)
int wl_attn_fused_bwd_ex(const void* qkv, const void* O, const void* dO, const float* lse, const float* gate,
                         const float* tab, const uint8_t* kpm, const void* pstore, uint64_t pstore_bytes, void* dqkv, float* dgate,
                         float* dtab, int dtab_accumulate,
                         void* dbias, int32_t dbias_dtype, int32_t dbias_accumulate, int32_t B, int32_t H, int32_t T,
                         int32_t head_dim, float scale, float p_drop, uint64_t seed, void* workspace, uint64_t ws_bytes,
                         void* stream);
// rowops.hip: out[c] (+)= sum over nblk rows of part[nblk][n]
int wl_colsum_finish(const float* part, int nblk, int n, void* out, int out_dtype, int accumulate, hipStream_t st);

// ---- deferred finishing launches (rowops.hip) ----
// Kernels that reduce over rows leave per-block partial rows and need a small finishing launch (pure load latency, ~5 us,
// ~100 of them per training step).  A caller that issues a fixed sequence of such kernels (layer.hip: one encoder block's
// backward) installs a list; while it is installed the producers APPEND their finishing work to it instead of launching,
// and wl_fin_flush() runs everything as ONE launch.  Each segment is summed exactly as its stand-alone finishing kernel
// would (same slices, same order): results are bit-identical.  out[r * rep_stride + c] for r < rep (the gate's 4 x replicated
// rows), (+)= if accumulate.  The partial buffers must stay untouched until the flush.
struct WlFinSeg { const float* part; void* out; long stride, rep_stride; int nblk, n, out_dtype, accumulate, rep, blk0; };
#define WL_FIN_MAX 16
struct WlFinList { WlFinSeg s[WL_FIN_MAX]; int n, blocks; };
void wl_fin_defer(WlFinList* l);      // thread-local; nullptr ends the deferral
bool wl_fin_active();                 // a list is installed: producers leave the listener notification to the flush
bool wl_fin_add(const float* part, int nblk, long stride, int n, void* out, int out_dtype, int accumulate, int rep = 1,
                long rep_stride = 0);  // false: nothing installed (or full) -- the caller launches its own finish
int wl_fin_flush(WlFinList& l, void* stream);   // one launch for the whole list + the gradient-listener notifications
