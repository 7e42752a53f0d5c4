// Convolutional position embedding, activation side, as a DIRECT grouped convolution on the matrix cores
// (WavLM/WavLM.py:514-527, 577-579: x + gelu(pos_conv(x))[:, :T]).
//
// Round 1 ran it as G*B "overlapping-row" GEMMs: A row t = the K*Cg contiguous elements of the group-major, time-padded
// activation copy starting at frame t.  That form re-fetches every activation element once per tap: 4.7 GB of
// L2 -> LDS fill per launch at the Base step shape for 75 MB of distinct data, and the launch ran at the L2 rate
// (519 us, 0.43 PF/s).  Here one workgroup owns a (batch, group, frame-segment): its whole input window
// ((BM + K - 1) frames x Cg channels, <= 100 KB) is loaded into LDS ONCE and every tap reads its A fragments from
// there at a shifted row; only the weights (K*Cg x Cg per group, shared by all batches of the group and L2-resident:
// workgroup index -> XCD keeps a group on one XCD) stream through a double-buffered LDS chunk.
//
// MFMA: v_mfma_f32_16x16x32_bf16, because Cg = 48 (Base) is 3 x 16: no padded output columns (the 128 x 64 tile of
// the GEMM form wasted 25 %).  The contraction index (tap, channel) is cut into 32-wide slices of FOUR CONSECUTIVE TAPS
// x one block of 8 channels: lane l of the A fragment (16 frames x 32 k) holds frame (l & 15) at tap tau4 + (l >> 4),
// channels 8 cb .. +8 -- one ds_read_b128 at window row (frame + tap); the B fragment (32 k x 16 columns) holds column
// (l & 15) at the same (tap, channels).
// The convolution is Toeplitz in (frame, tap): the A fragment of frame block mt at taps tau4 + 16 is the fragment of
// block mt + 1 at taps tau4.  So the slices are walked in groups (cb, rho = tau4 mod 16) of J = K / 16 slices
// tau4 = rho + 16 j: the group needs only MT + J - 1 distinct A fragments (row block rb = mt + j) for its MT * J
// (fragment, slice) pairs -- 19 reads instead of 96 at MT = 12.  (The first version read every pair: LDS-bound,
// 288 us per launch against 125 us of MFMA time.)  Per group a wave reads MT + J - 1 A and J * NT B fragments for
// MT * J * NT MFMAs.
// Weights arrive in a matching image (wavlm_posconv_weight_fwd layout 1): per (group of channels g, cb, rho) one
// contiguous blob [j][tap - tau4][column][8 channels], which is exactly the order the B fragments are read in, so it
// moves global -> LDS as plain 1 KiB LDS-DMA pieces (double buffered) and is read back without bank conflicts.
// Activation rows in LDS are padded by 8 elements (row stride 112 B at Cg = 48, 144 B at Cg = 64): the 16 lanes of an A
// fragment read touch 16 distinct 16-byte bank groups.
//
// Epilogue: accumulators -> the wave's own LDS slice (same padded rows) -> 16-byte row vectors, so that bias,
// GELU / GELU' (chord table, gemm_common.hpp), the residual and the two outputs move as 96 / 128-byte row segments.
// Serves the forward (W = Wf, gelu, aux = pre-activation, res = x) and the backward-data pass (W = Wb over the
// group-major dy * gelu', res = dy) -- the same calls PosConvFn made to wavlm_gemm.
#include "tile_loaders.hpp"
#include "gemm_common.hpp"
#include "../../include/wavlm_hip.h"

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((address_space(3))) bf16x4_t* lds_b4_ptr;

struct PcdP {
  const bf16_t* xg; const bf16_t* W; const bf16_t* bias; const bf16_t* res; bf16_t* out; bf16_t* aux;
  const float4* gtab;
  int B, G, T, Tp, D, K, nseg, gelu;
};

template <int CG, int MT, int J, int NW>
__global__ __launch_bounds__(NW * 64, 1) void posconv_direct_kernel(PcdP p) {
  constexpr int NTH = NW * 64;
  constexpr int NT = CG / 16;        // 16-column blocks
  constexpr int C8 = CG / 8;         // 16-byte vectors per activation row = channel blocks
  constexpr int XR = CG + 8;         // padded LDS row (elements)
  constexpr int XRB = XR * 2;
  constexpr int RW = 16 * MT;        // frames per wave
  constexpr int BM = NW * RW;        // frames per workgroup
  constexpr int NA = MT + J - 1;     // distinct A fragments of a group
  constexpr int GB = J * 4 * CG * 16;  // bytes of one weight blob: [j][4 taps][CG columns][8 channels]
  constexpr int NG = C8 * 4;         // groups (channel block, rho / 4)
  constexpr int PPW = GB / 1024 / NW;  // 1 KiB DMA pieces per wave
  static_assert(GB % (1024 * NW) == 0, "a blob is whole 1 KiB pieces per wave");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int WIN = BM + 16 * J - 1;
  unsigned char* xs = smem;
  unsigned char* wbuf = smem + (size_t)((WIN * XRB + 1023) & ~1023);  // 2 x GB

  const int blk = blockIdx.x;
  const int seg = blk % p.nseg, bg = blk / p.nseg;
  const int g = bg % p.G, b = bg / p.G;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lm = lane & 15, q = lane >> 4;
  const int t0 = seg * BM;  // first output frame of the workgroup (= first window row in padded coordinates)

  // ---- weight blobs: LDS-DMA, 1 KiB per wave-instruction
  const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.W) + (size_t)g * NG * GB;
  auto wdma = [&](int grp, int st) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int piece = wave * PPW + i;
      __builtin_amdgcn_global_load_lds((gas_ptr)(wsrc + (size_t)grp * GB + piece * 1024 + lane * 16),
                                       (las_ptr)(wbuf + (size_t)st * GB + piece * 1024), 16, 0, 0);
    }
  };
  wdma(0, 0);
  // ---- input window -> LDS (rows past the padded length are zero)
  {
    const bf16_t* src = p.xg + (long)bg * p.Tp * CG;
    const int nv = WIN * C8;
    for (int v0 = threadIdx.x; v0 < nv; v0 += NTH * 8) {
      uint4 r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int v = v0 + NTH * u;
        const int row = v / C8, c8 = v - row * C8;
        r[u] = make_uint4(0, 0, 0, 0);
        if (v < nv && t0 + row < p.Tp) r[u] = *reinterpret_cast<const uint4*>(src + (long)(t0 + row) * CG + c8 * 8);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int v = v0 + NTH * u;
        const int row = v / C8, c8 = v - row * C8;
        if (v < nv) *reinterpret_cast<uint4*>(xs + (size_t)row * XRB + c8 * 16) = r[u];
      }
    }
  }

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const unsigned char* abase = xs + (size_t)(wave * RW + lm + q) * XRB;  // + (4 r4) rows + cb * 16 bytes per group
  const unsigned boff = (unsigned)((q * CG + lm) * 16);                   // + ((j * 4) * CG + 16 nt) * 16
  __syncthreads();

  for (int grp = 0; grp < NG; ++grp) {
    if (grp + 1 < NG) wdma(grp + 1, (grp + 1) & 1);  // that buffer was last read in group grp - 1: every wave has passed the barrier since
    const int cb = grp >> 2, r4 = grp & 3;
    const unsigned char* ab = abase + (size_t)(4 * r4) * XRB + cb * 16;
    const unsigned char* wb = wbuf + (size_t)(grp & 1) * GB + boff;
    U4 af[NA];
#pragma unroll
    for (int rb = 0; rb < NA; ++rb) af[rb].v = *reinterpret_cast<const uint4*>(ab + rb * 16 * XRB);
#pragma unroll
    for (int j = 0; j < J; ++j) {
      U4 bf[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bf[nt].v = *reinterpret_cast<const uint4*>(wb + (j * 4 * CG + 16 * nt) * 16);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt + j].b, bf[nt].b, acc[mt][nt], 0, 0, 0);
    }
    __syncthreads();  // (waits for this wave's DMA pieces of the next blob, then for everybody's)
  }

  // ---- epilogue.  The loop's last barrier has passed: the window and the weight buffers are dead.
  float4* tabL = reinterpret_cast<float4*>(smem + (size_t)BM * XRB);
  if (p.gelu) {
    gelu_tab_stage(p.gtab, tabL);
    __syncthreads();
  }
  unsigned char* st = smem + (size_t)wave * RW * XRB;
  float bias_n[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bias_n[nt] = p.bias ? bf2f(p.bias[g * CG + 16 * nt + lm]) : 0.f;
  const int tw = t0 + wave * RW;  // first frame of the wave
  // D[m][n] of a 16 x 16 block: lane holds rows 4 q + i (i = 0..3), column lm
  auto stage = [&](auto second_c) __attribute__((always_inline)) {
    constexpr bool SECOND = decltype(second_c)::value;  // false: pre-activation (or the plain result); true: gelu value
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = acc[mt][nt][i];
          if constexpr (!SECOND) {
            v += bias_n[nt];
            if (p.gelu) {  // aux takes the pre-activation itself (what epi = 1 of wavlm_gemm stores)
              float gr;
              acc[mt][nt][i] = gelu_both_tab(tabL, v, gr);
            }
          }
          *reinterpret_cast<bf16_t*>(st + (size_t)(16 * mt + 4 * q + i) * XRB + (16 * nt + lm) * 2) = f2bf(v);
        }
  };
  auto flush = [&](bf16_t* dst, const bf16_t* res) __attribute__((always_inline)) {
    for (int v = lane; v < RW * C8; v += 64) {
      const int row = v / C8, c8 = v - row * C8;
      const int t = tw + row;
      if (t < p.T) {
        U4 o; o.v = *reinterpret_cast<const uint4*>(st + (size_t)row * XRB + c8 * 16);
        const long gi = ((long)b * p.T + t) * p.D + g * CG + c8 * 8;
        if (res) {
          U4 r; r.v = *reinterpret_cast<const uint4*>(res + gi);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = __uint_as_float(o.u[e] << 16) + __uint_as_float(r.u[e] << 16);
            const float hi = __uint_as_float(o.u[e] & 0xffff0000u) + __uint_as_float(r.u[e] & 0xffff0000u);
            o.u[e] = pack_bf16x2(lo, hi);
          }
        }
        *reinterpret_cast<uint4*>(dst + gi) = o.v;
      }
    }
  };
  // LDS operations of one wave execute in order, so the wave's private slice needs no barrier -- but the COMPILER must
  // not move the 2-byte stores across the 16-byte loads of the same bytes (different access types: type-based alias
  // analysis sees no conflict)
#define PCD_FENCE() asm volatile("" ::: "memory")
  if (p.gelu) {
    stage(std::false_type{});          // pre-activation -> aux; the accumulators now hold gelu(pre-activation)
    PCD_FENCE();
    if (p.aux) flush(p.aux, nullptr);
    PCD_FENCE();
    stage(std::true_type{});
    PCD_FENCE();
    flush(p.out, p.res);
  } else {
    stage(std::false_type{});
    PCD_FENCE();
    flush(p.out, p.res);
  }
}

template <int CG, int MT, int J, int NW>
static int pcd_launch(const PcdP& p, hipStream_t st) {
  constexpr int BM = NW * 16 * MT;
  const size_t win = ((size_t)(BM + 16 * J - 1) * (CG + 8) * 2 + 1023) & ~(size_t)1023;
  const size_t wb = (size_t)2 * J * 4 * CG * 16;
  size_t smem = win + wb;
  const size_t epi = (size_t)BM * (CG + 8) * 2 + (size_t)GT4_N * sizeof(float4);
  if (smem < epi) smem = epi;
  if (smem > 160 * 1024) return WL_EINVAL;
  static size_t allowed = 0;  // per instantiation
  if (smem > allowed) {
    if (hipFuncSetAttribute((const void*)posconv_direct_kernel<CG, MT, J, NW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem) != hipSuccess)
      return WL_ELAUNCH;
    allowed = smem;
  }
  WL_LAUNCH((posconv_direct_kernel<CG, MT, J, NW>), dim3((unsigned)(p.B * p.G * p.nseg)), dim3(NW * 64), smem, st, p);
  return wl_check_launch();
}

// ---------------------------------------------------------------------------------------------- weight gradient
// dw[g][n][tap][ci] = sum_{b, t} xg[b, g, t + tap, ci] * du[b, t, g*Cg + n]  (du = dy * gelu'(pre-activation), read from its
// group-major copy).  The GEMM form (M = K*Cg, N = Cg, contraction over B*T, 128 x 64 tiles) ran at 0.27 PF/s: 830 us at
// the Base step shape, 1.9 ms at the Large one -- the same per-tap re-fetch of the activations as the forward had.
//
// Here the contraction index t is the MFMA's k: D[ci][n] (16 x 16) += A[ci][32 t] * B[32 t][n] for one tap, with both
// operands "k-strided" in LDS ([frame][channel] rows) and read with the transposing ds_read_b64_tr_b16.  Toeplitz again:
// the A fragment of tap + 32 at frames t0.. is the fragment of tap at frames t0 + 32.., so a wave owns the FOUR taps
// rho, rho + 32, rho + 64, rho + 96 of one residue and walks t in steps of 32: each step reads one new A fragment set
// (NCI of them) and keeps three from the steps before -- 4x fewer activation reads than taps x steps -- plus NT B
// fragments, for 4 * NCI * NT MFMAs.  A workgroup = NW / NH residues of one channel group (at Cg = 64 the 256
// accumulator registers of a residue are split over NH = 2 waves by output column); 32 / (NW / NH) workgroups cover
// the residues, BS of them split the batch: every CU gets one workgroup.  The BS partial sums leave as fp32 slabs and are
// added by wavlm_posconv_weight_bwd's first kernel (deterministic, no atomics).
// Frames stream through LDS in chunks of TCH (activation rows TCH + 127, gradient rows TCH), double buffered through
// registers (rows are padded to Cg + 8 elements, which a DMA image cannot be).
struct PdwP {
  const bf16_t* xg; const bf16_t* dug; float* part;
  int B, G, T, Tp, du_off, K, BS, bchunk;
};

template <int CG, int NW, int TCH, int NH>
__global__ __launch_bounds__(NW * 64, 1) void posconv_dw_kernel(PdwP p) {
  constexpr int NTH = NW * 64;
  constexpr int NCI = CG / 16, NT = CG / 16 / NH, C8 = CG / 8;  // NH waves share a residue, each takes NT of the column blocks
  constexpr int NRES = NW / NH;                                // residues per workgroup
  constexpr int XR = CG + 8, XRB = XR * 2;
  constexpr int XROWS = TCH + 127;
  constexpr int STG = (XROWS + TCH) * XRB;       // bytes of one stage: activation rows, then gradient rows
  constexpr int NVX = XROWS * C8, NVS = (XROWS + TCH) * C8;
  constexpr int NVT = (NVS + NTH - 1) / NTH;
  constexpr int NST = TCH / 32;                  // steps per chunk
  constexpr int RQ = 32 / NRES;                  // workgroups per (channel group, batch split)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int gbs = blockIdx.x % (p.G * p.BS), rq = blockIdx.x / (p.G * p.BS);  // the RQ workgroups that share the same
  const int bs = gbs % p.BS, g = gbs / p.BS;                                  // data sit on one XCD (G * BS % 8 == 0)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, q = lane >> 4;
  const int rho = rq * NRES + wave / NH, nh = wave % NH;
  const int b0 = bs * p.bchunk, b1 = min(p.B, b0 + p.bchunk);
  const int nch = (p.T + TCH - 1) / TCH;
  const int nwork = (b1 - b0) * nch;

  f32x4_t acc[4][NCI][NT];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int c = 0; c < NCI; ++c)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[j][c][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  u32x4_t sr[NVT];
  // work item w = (batch, frame chunk) -> registers (rows past the padded length / past T are zero)
#define PDW_LOAD(W)                                                                                   \
  {                                                                                                   \
    const int bb = b0 + (W) / nch, tc = (W) - ((W) / nch) * nch;                                       \
    const bf16_t* xsrc = p.xg + ((long)bb * p.G + g) * p.Tp * CG;                                      \
    const bf16_t* dsrc = p.dug + (((long)bb * p.G + g) * p.Tp + p.du_off) * CG;                        \
    _Pragma("unroll") for (int u = 0; u < NVT; ++u) {                                                  \
      const int v = threadIdx.x + NTH * u;                                                            \
      const bool isx = v < NVX;                                                                       \
      const int vv = isx ? v : v - NVX;                                                               \
      const int row = vv / C8, c8 = vv - row * C8;                                                    \
      const int fr = tc * TCH + row;                                                                  \
      const bool ok = v < NVS && (isx ? fr < p.Tp : fr < p.T);                                        \
      const bf16_t* src = (isx ? xsrc : dsrc) + (long)(ok ? fr : 0) * CG + c8 * 8;                     \
      u32x4_t val = *reinterpret_cast<const u32x4_t*>(src);                                           \
      if (!ok) val = u32x4_t{0u, 0u, 0u, 0u};                                                         \
      sr[u] = val;                                                                                    \
    }                                                                                                 \
  }
#define PDW_STORE(ST)                                                                                 \
  _Pragma("unroll") for (int u = 0; u < NVT; ++u) {                                                    \
    const int v = threadIdx.x + NTH * u;                                                              \
    const int row = v / C8, c8 = v - row * C8;                                                        \
    if (v < NVS) *reinterpret_cast<u32x4_t*>(smem + (size_t)(ST) * STG + (size_t)row * XRB + c8 * 16) = sr[u]; \
  }
  // lane's part of a transposing fragment read: frame row 8 q + (li >> 2), channels 4 (li & 3) .. +4
  const unsigned troff = (unsigned)((8 * q + (li >> 2)) * XRB + (li & 3) * 8);
  auto frag = [&](const unsigned char* base) __attribute__((always_inline)) -> bf16x8_t {
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_ptr)(base));
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_ptr)(base + 4 * XRB));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };

  if (nwork > 0) {
    PDW_LOAD(0)
    PDW_STORE(0)
  }
  __syncthreads();
  for (int w = 0; w < nwork; ++w) {
    const bool more = w + 1 < nwork;
    if (more) PDW_LOAD(w + 1)
    const unsigned char* xb = smem + (size_t)(w & 1) * STG + troff + (size_t)rho * XRB;  // activation frame 0 of tap rho
    const unsigned char* db = smem + (size_t)(w & 1) * STG + (size_t)XROWS * XRB + troff;
    bf16x8_t F[4][NCI];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int c = 0; c < NCI; ++c) F[u][c] = frag(xb + (size_t)(32 * u) * XRB + c * 32);
#pragma unroll
    for (int s = 0; s < NST; ++s) {
#pragma unroll
      for (int c = 0; c < NCI; ++c) F[(s + 3) & 3][c] = frag(xb + (size_t)(32 * (s + 3)) * XRB + c * 32);
      bf16x8_t Bf[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) Bf[n] = frag(db + (size_t)(32 * s) * XRB + (nh * NT + n) * 32);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < NCI; ++c)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[j][c][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F[(s + j) & 3][c], Bf[n], acc[j][c][n], 0, 0, 0);
    }
    if (more) PDW_STORE((w + 1) & 1)  // that stage was last read for work item w - 1: every wave has passed the barrier since
    __syncthreads();
  }
  // D[m][n] of a 16 x 16 block: lane holds rows (channels) 4 q + i, column (output column) li
  float* out = p.part + ((long)bs * p.G + g) * CG * ((long)p.K * CG);
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int c = 0; c < NCI; ++c)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int col = 16 * (nh * NT + n) + li, tap = rho + 32 * j;
        *reinterpret_cast<f32x4_t*>(out + ((long)col * p.K + tap) * CG + 16 * c + 4 * q) = acc[j][c][n];
      }
#undef PDW_LOAD
#undef PDW_STORE
}

template <int CG, int NW, int TCH, int NH>
static int pdw_launch(const PdwP& p, hipStream_t st) {
  const size_t smem = (size_t)2 * (TCH + 127 + TCH) * (CG + 8) * 2;
  if (smem > 160 * 1024) return WL_EINVAL;
  static size_t allowed = 0;
  if (smem > allowed) {
    if (hipFuncSetAttribute((const void*)posconv_dw_kernel<CG, NW, TCH, NH>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem) != hipSuccess)
      return WL_ELAUNCH;
    allowed = smem;
  }
  WL_LAUNCH((posconv_dw_kernel<CG, NW, TCH, NH>), dim3((unsigned)(p.G * p.BS * (32 / (NW / NH)))), dim3(NW * 64), smem, st, p);
  return wl_check_launch();
}

extern "C" {

int wavlm_posconv_direct_supported(int32_t Cg, int32_t K, int32_t T) {
  if (T <= 0 || K != 128) return 0;  // J = K / 16 = 8 slices per group is a template argument
  return (Cg == 48 || Cg == 64) ? 1 : 0;
}

// out[b, t, g*Cg + n] = (res) + f(sum_{tap, ci} xg[b, g, t + tap, ci] * w[g, n, tap, ci] (+ bias)),  f = gelu or id;
// W = the layout-1 image of w (wavlm_posconv_weight_fwd)
int wavlm_posconv_direct(const void* xg, const void* W, const void* bias, const void* res, void* out, void* aux,
                         int32_t B, int32_t G, int32_t T, int32_t Tp, int32_t Cg, int32_t K, int32_t gelu, void* stream) {
  if (!xg || !W || !out || B <= 0 || G <= 0 || T <= 0 || Tp < T + K - 1) return WL_EINVAL;
  if (!wavlm_posconv_direct_supported(Cg, K, T)) return WL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  PcdP p;
  p.xg = (const bf16_t*)xg; p.W = (const bf16_t*)W; p.bias = (const bf16_t*)bias; p.res = (const bf16_t*)res;
  p.out = (bf16_t*)out; p.aux = (bf16_t*)aux; p.gtab = nullptr;
  p.B = B; p.G = G; p.T = T; p.Tp = Tp; p.D = G * Cg; p.K = K; p.gelu = gelu ? 1 : 0;
  if (p.gelu) {
    p.gtab = wl_gelu_tab4(st);
    if (!p.gtab) return WL_ELAUNCH;
  }
  // frames per workgroup: the candidate (64 MT) that wastes the fewest padded frames, larger tiles first on ties
  if (Cg == 48) {
    const int cand[3] = {12, 8, 6};
    int best = 0; long bw = -1;
    for (int i = 0; i < 3; ++i) {
      const int bm = 64 * cand[i];
      const long tot = (long)((T + bm - 1) / bm) * bm;
      if (bw < 0 || tot < bw) { bw = tot; best = i; }
    }
    const int mt = cand[best];
    p.nseg = (T + 64 * mt - 1) / (64 * mt);
    // eight waves of half the height rather than four (one per SIMD) of the full: two waves per SIMD cover each other's
    // fragment-read and barrier stalls -- 213 -> 179 us at 768 frames, for 1.7x the LDS reads per MFMA
    if (mt == 12) return pcd_launch<48, 6, 8, 8>(p, st);
    if (mt == 8) return pcd_launch<48, 4, 8, 8>(p, st);
    return pcd_launch<48, 3, 8, 8>(p, st);
  } else {
    const int cand[2] = {8, 6};
    int best = 0; long bw = -1;
    for (int i = 0; i < 2; ++i) {
      const int bm = 64 * cand[i];
      const long tot = (long)((T + bm - 1) / bm) * bm;
      if (bw < 0 || tot < bw) { bw = tot; best = i; }
    }
    const int mt = cand[best];
    p.nseg = (T + 64 * mt - 1) / (64 * mt);
    if (mt == 8) return pcd_launch<64, 4, 8, 8>(p, st);
    return pcd_launch<64, 3, 8, 8>(p, st);
  }
}


// number of fp32 slabs wavlm_posconv_dw_direct writes (the batch is split that many ways)
int wavlm_posconv_dw_direct_splits(int32_t Cg, int32_t G) {
  if (Cg == 48) return (G * 4) % 8 == 0 ? 4 : 0;
  if (Cg == 64) return (G * 2) % 8 == 0 ? 2 : 0;
  return 0;
}

// part[split][g][n][tap][ci] (fp32, `splits` slabs of G*Cg*K*Cg) = partial sums over the batch of
// sum_t xg[b, g, t + tap, ci] * dug[b, g, du_off + t, n];  K = 128, Cg = 48 or 64
int wavlm_posconv_dw_direct(const void* xg, const void* dug, float* part, int32_t B, int32_t G, int32_t T, int32_t Tp,
                            int32_t du_off, int32_t Cg, int32_t K, void* stream) {
  if (!xg || !dug || !part || B <= 0 || G <= 0 || T <= 0 || K != 128 || Tp < T + K - 1 || du_off < 0 || du_off + T > Tp)
    return WL_EINVAL;
  const int bs = wavlm_posconv_dw_direct_splits(Cg, G);
  if (bs == 0) return WL_EINVAL;
  PdwP p;
  p.xg = (const bf16_t*)xg; p.dug = (const bf16_t*)dug; p.part = part;
  p.B = B; p.G = G; p.T = T; p.Tp = Tp; p.du_off = du_off; p.K = K; p.BS = bs; p.bchunk = (B + bs - 1) / bs;
  if (Cg == 48) return pdw_launch<48, 8, 256, 1>(p, (hipStream_t)stream);
  return pdw_launch<64, 8, 128, 2>(p, (hipStream_t)stream);  // 256 accumulator registers per residue: two waves share it
}
}  // extern "C"
