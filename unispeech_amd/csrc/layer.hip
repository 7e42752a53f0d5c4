// One transformer encoder block per C call (include/wavlm_hip.h, wavlm_layer_desc).
//
// Host code only: the block's kernel sequence is fixed (gate -> q|k|v GEMM -> fused attention -> out_proj -> LayerNorm ->
// fc1 (+GELU) -> fc2 -> LayerNorm, and its mirror image), so it is issued from here through the library's own entry points
// instead of from ~35 Python autograd nodes.  What the Python launch thread pays per block and direction is one descriptor,
// one ctypes call and one autograd node; what this file pays per kernel is the hipLaunchKernel itself.  Reference: the
// per-layer body of TransformerSentenceEncoderLayer.forward (WavLM/WavLM.py:694-742) and of MultiheadAttention.forward
// (WavLM/modules.py:417-563), which the reference's trainer runs once per micro-batch (src/fairseq/trainer.py:697-760).
//
// Memory: `saved` (forward -> backward) and `workspace` (temporaries of one call) are carved here with fixed layouts, so
// the caller allocates two buffers per call, not twenty.
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "common.hpp"
#include "wavlm_hip.h"

// ---- gradient listener: which slice of the caller's gradient arena has just received an accumulation (in stream order) ----
static std::atomic<wavlm_grad_listener> g_listener{nullptr};
static std::atomic<void*> g_listener_user{nullptr};
void wl_notify_grad(const void* base, uint64_t bytes, void* stream) {
  const wavlm_grad_listener cb = g_listener.load(std::memory_order_acquire);
  if (cb && base && bytes) cb(base, bytes, stream, g_listener_user.load(std::memory_order_acquire));
}
extern "C" void wavlm_dp_set_listener(wavlm_grad_listener cb, void* user) {
  g_listener.store(nullptr, std::memory_order_release);
  g_listener_user.store(user, std::memory_order_release);
  g_listener.store(cb, std::memory_order_release);
}

namespace {

inline uint64_t rup256(uint64_t v) { return (v + 255) & ~(uint64_t)255; }

struct Carver {  // bump allocator over a caller-provided region (256-byte aligned pieces)
  char* base; uint64_t off, cap; bool ok;
  Carver(void* p, uint64_t bytes) : base((char*)p), off(0), cap(bytes), ok(p != nullptr) {}
  void* take(uint64_t bytes) {
    void* r = base + off;
    off += rup256(bytes);
    if (off > cap) ok = false;
    return r;
  }
};

struct Saved {  // forward -> backward
  float *gate, *ga, *gb, *lse, *mean1, *rstd1, *mean2, *rstd2;
  void *qkv, *O, *s1, *h1, *x1, *u, *hact, *s2;
  void* pstore; uint64_t pstore_b;   // attention probabilities (attn_store_p)
  uint64_t bytes;
};

// post-LN: s1 = x + drop(a), x1 = LN1(s1), s2 = x1 + drop(f).  pre-LN: s1 = x + drop(r_in) (only stored when r_in is given:
// otherwise it IS x), h1 = LN1(s1), x1 = h2 = LN2(y); s2 is the output y itself.
Saved carve_saved(const wavlm_layer_desc* d, void* base) {
  Saved s;
  memset(&s, 0, sizeof(s));
  Carver c(base ? base : (void*)256, ~(uint64_t)0 >> 1);  // base == NULL: sizes only
  const uint64_t n = (uint64_t)d->B * d->T, bht = (uint64_t)d->B * d->H * d->T;
  const bool gated = d->Wgate != nullptr;
  if (gated) { s.gate = (float*)c.take(bht * 4); s.ga = (float*)c.take(bht * 4); s.gb = (float*)c.take(bht * 4); }
  else if (d->tab) s.gate = (float*)c.take(bht * 4);  // relative position bias without gru_rel_pos: gate == 1
  s.lse = (float*)c.take(bht * 4);
  s.mean1 = (float*)c.take(n * 4); s.rstd1 = (float*)c.take(n * 4);
  s.mean2 = (float*)c.take(n * 4); s.rstd2 = (float*)c.take(n * 4);
  s.qkv = c.take(n * 3 * d->D * 2);
  s.O = c.take(n * d->D * 2);
  if (!d->pre_ln || d->r_in) s.s1 = c.take(n * d->D * 2);
  if (d->pre_ln) s.h1 = c.take(n * d->D * 2);
  s.x1 = c.take(n * d->D * 2);
  s.u = c.take(n * d->F * 2);
  s.hact = c.take(n * d->F * 2);
  if (!d->pre_ln) s.s2 = c.take(n * d->D * 2);
  if (d->attn_store_p == 2) {   // the forward's dropout decisions as bit words (nothing to keep without attention dropout)
    s.pstore_b = d->p_attn > 0.f ? wavlm_attn_fused_dbits_bytes(d->B, d->H, d->T) : 0;
    if (s.pstore_b) s.pstore = c.take(s.pstore_b);
  } else
  if (d->attn_store_p) {   // (0 bytes: this T is not supported by the stored form -- recompute)
    s.pstore_b = wavlm_attn_fused_pstore_bytes(d->B, d->H, d->T);
    if (s.pstore_b) s.pstore = c.take(s.pstore_b);
  }
  s.bytes = c.off;
  return s;
}

bool desc_ok(const wavlm_layer_desc* d) {
  if (!d || d->B <= 0 || d->T <= 0 || d->D <= 0 || d->H <= 0 || d->F <= 0) return false;
  if (d->D != d->H * 64 || (d->D & 7) || (d->F & 7)) return false;
  if (!d->Wqkv || !d->bqkv || !d->Wo || !d->bo || !d->W1 || !d->b1 || !d->W2 || !d->b2) return false;
  if (!d->ln1_g || !d->ln1_b || !d->ln2_g || !d->ln2_b) return false;
  if ((d->Wgate != nullptr) != (d->bgate != nullptr) || (d->Wgate != nullptr) != (d->grep_a != nullptr)) return false;
  if (d->Wgate && !d->tab) return false;
  if (d->param_dtype != WL_BF16 && d->param_dtype != WL_F32) return false;
  return true;
}

void gemm_base(wavlm_gemm_desc& g) {
  memset(&g, 0, sizeof(g));
  g.dtype = WL_BF16; g.c_dtype = WL_BF16; g.KB = 1; g.batch_o = 1; g.batch_i = 1; g.alpha = 1.f; g.split_k = 1;
}

// y[n, N] = epi(x[n, K] W[N, K]^T + b)
int lin_fwd(const void* x, const void* W, const void* b, void* y, int64_t n, int N, int K, int pdt, int epi, void* aux,
            void* stream) {
  wavlm_gemm_desc g; gemm_base(g);
  g.M = (int32_t)n; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N;
  g.A = x; g.B = W; g.C = y; g.bias = b; g.bias_dtype = pdt; g.epi = epi;
  if (aux) { g.aux = aux; g.aux_dtype = WL_BF16; g.ld_aux = N; }
  return wavlm_gemm(&g, stream);
}

// dx[n, K] = (dy[n, N] W[N, K]) (* aux) (+ res); colsum (optional, param dtype): (+)= column sums of dx
int lin_dx(const void* dy, const void* W, void* dx, int64_t n, int N, int K, int epi, const void* aux, const void* res,
           void* colsum, int pdt, void* ws, uint64_t wsb, void* stream) {
  wavlm_gemm_desc g; gemm_base(g);
  g.M = (int32_t)n; g.N = K; g.K = N; g.lda = N; g.ldb = K; g.ldc = K; g.transB = 1;
  g.A = dy; g.B = W; g.C = dx; g.epi = epi;
  if (aux) { g.aux = const_cast<void*>(aux); g.aux_dtype = WL_BF16; g.ld_aux = K; }
  if (res) { g.res = res; g.res_dtype = WL_BF16; g.ld_res = K; }
  if (colsum) { g.colsum = colsum; g.colsum_dtype = pdt; g.colsum_accumulate = 1; g.workspace = ws; g.ws_bytes = wsb; }
  return wavlm_gemm(&g, stream);
}
uint64_t lin_dx_colsum_ws(int64_t n, int K) {
  wavlm_gemm_desc g; gemm_base(g);
  g.M = (int32_t)n; g.N = K; g.colsum = (void*)256;
  return wavlm_gemm_workspace_bytes(&g);
}

// ---- weight gradients: dW[N, K] += dy[n, N]^T x[n, K], grouped as unispeech_amd/functional.py WgradGroup does ----
struct WG { const void* dy; const void* x; void* dW; int N, K; };

int env_int(const char* name, int dflt) { const char* e = getenv(name); return e && *e ? atoi(e) : dflt; }

// split-K factor of a grouped launch: the largest that keeps tiles * split within ONE round of the 256-block persistent
// grid (measured: every extra round costs a slab store and a pipeline refill per CU); 0 = run the members singly
// blocks of a persistent GEMM grid (== ops.grid_blocks): the data-parallel reducer keeps some CUs free for RCCL
long grid_blocks() { return 256 - wavlm_get_reserved_cus(); }
int grouped_split(long tiles, long ktiles) {
  static const int forced = env_int("WAVLM_WGRAD_SPLIT", 0);
  if (forced > 0) return forced < 2 ? 2 : forced;
  long s = grid_blocks() / (tiles > 0 ? tiles : 1);
  if (s > ktiles / 8) s = ktiles / 8;
  if (s > 64) s = 64;
  return s >= 2 ? (int)s : 0;
}
// `split_k` of a grouped launch's members = slabs per member: the one-round split; the balanced launch (gemm_common.hpp:
// gemm_sk_plan, WAVLM_WGRAD_STREAMK=1) needs one more (== ops.grouped_slabs)
#if defined(WAVLM_EXPERIMENTAL)
// exported by the lab library ONLY: how ops.grouped_slabs learns that the balanced launch exists in the library it loaded
// (it used to infer it from WAVLM_HIP_LIB being set, which is also how one points at a product build elsewhere)
extern "C" int wavlm_lab_build(void) { return 1; }
#endif
int grouped_slabs(long tiles, long ktiles) {
#if defined(WAVLM_EXPERIMENTAL)   // (the balanced launch exists only in the lab library)
  static const bool sk = env_int("WAVLM_WGRAD_STREAMK", 0) != 0 && env_int("WAVLM_WGRAD_SPLIT", 0) <= 0;
#else
  constexpr bool sk = false;
#endif
  int split = grouped_split(tiles, ktiles); if (split < 2) split = 2;
  const long G = grid_blocks();
  if (sk && tiles < G && tiles * ktiles >= 8 * G) { const int s = (int)(G / tiles) + 1; if (s > split) split = s; }
  return split;
}
int single_split(int M, int N, long ktiles) {
  if (M >= 256 && N >= 256) {
    const long tiles = (long)((M + 255) / 256) * ((N + 255) / 256);
    long s = grid_blocks() / tiles; if (s < 1) s = 1;
    if (s > ktiles / 8) s = ktiles / 8;
    if (s > 64) s = 64;
    return s < 1 ? 1 : (int)s;
  }
  const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
  long s = (768 + tiles - 1) / tiles;
  if (s > ktiles) s = ktiles;
  if (s < 1) s = 1;
  return s > 64 ? 64 : (int)s;
}
void wg_desc(wavlm_gemm_desc& g, const WG& w, int64_t n, int pdt, int split) {
  gemm_base(g);
  g.c_dtype = pdt; g.M = w.N; g.N = w.K; g.K = (int32_t)n; g.transA = 1; g.transB = 1;
  g.lda = w.N; g.ldb = w.K; g.ldc = w.K; g.A = w.dy; g.B = w.x; g.C = w.dW; g.accumulate = 1; g.split_k = split;
}
uint64_t wg_ws_bytes(const WG* it, int cnt, int64_t n) {  // upper bound over every grouping the launcher may choose
  uint64_t tot = 0;
  const long kt = (long)((n + 63) / 64);
  for (int i = 0; i < cnt; ++i) {
    int s = single_split(it[i].N, it[i].K, kt);
    const int gs = grouped_slabs((long)((it[i].N + 255) / 256) * ((it[i].K + 255) / 256), kt);  // alone in a group: the most slabs it can see
    if (gs > s) s = gs;
    tot += rup256((uint64_t)s * it[i].N * it[i].K * 4);
  }
  return tot;
}
int wgrads(const WG* it, int cnt, int64_t n, int pdt, void* ws, uint64_t wsb, void* stream) {
  // anything but the literal "0" is on -- as unispeech_amd/functional.py reads it (atoi("auto") would be 0: the fused block
  // would silently stop grouping while the composed path still groups, and the two paths would no longer issue the same kernels)
  static const bool grouping = [] { const char* e = getenv("WAVLM_WGRAD_GROUPING"); return !(e && e[0] == '0' && !e[1]); }();
  const long kt = (long)((n + 63) / 64);
  int i = 0;
  while (i < cnt) {
    // greedy sub-group in arrival order: members whose tiles still fit one round with a split >= 2 (at most four)
    int j = i; long tiles = 0;
    while (grouping && j < cnt && j - i < 4) {
      const long t = (long)((it[j].N + 255) / 256) * ((it[j].K + 255) / 256);
      if (j > i && grouped_split(tiles + t, kt) < 2) break;
      tiles += t; ++j;
    }
    if (j == i) j = i + 1;
    const int m = j - i;
    Carver c(ws, wsb);
    if (m >= 2) {
      const int split = grouped_slabs(tiles, kt);
      wavlm_gemm_desc g[4];
      for (int k = 0; k < m; ++k) {
        wg_desc(g[k], it[i + k], n, pdt, split);
        g[k].ws_bytes = wavlm_gemm_workspace_bytes(&g[k]);
        g[k].workspace = c.take(g[k].ws_bytes);
      }
      if (!c.ok) return WL_EINVAL;
      const int rc = wavlm_gemm_grouped(g, m, stream);
      if (rc != WL_OK) return rc;
    } else {
      wavlm_gemm_desc g;
      wg_desc(g, it[i], n, pdt, single_split(it[i].N, it[i].K, kt));
      if (g.split_k > 1) { g.ws_bytes = wavlm_gemm_workspace_bytes(&g); g.workspace = c.take(g.ws_bytes); if (!c.ok) return WL_EINVAL; }
      const int rc = wavlm_gemm(&g, stream);
      if (rc != WL_OK) return rc;
    }
    i = j;
  }
  return WL_OK;
}

#define RC(expr) do { const int rc_ = (expr); if (rc_ != WL_OK) return rc_; } while (0)

struct BwdWs {
  void *dsa, *df, *du, *dh, *dxa, *da, *dO, *dqkv;
  float* dgate;
  void *ln_ws, *ln_ws2, *cs_ws, *cs_ws2, *attn_ws, *gate_ws, *wg_ws;   // (two LayerNorm / column-sum partial buffers: the
  uint64_t ln_b, cs_b, attn_b, gate_b, wg_b, bytes;                     //  finishing launches are deferred to the end of the block)
};
BwdWs carve_bwd(const wavlm_layer_desc* d, void* base, uint64_t cap) {
  BwdWs w; memset(&w, 0, sizeof(w));
  Carver c(base ? base : (void*)256, base ? cap : (~(uint64_t)0 >> 1));
  const uint64_t n = (uint64_t)d->B * d->T, D = d->D, F = d->F;
  w.dsa = c.take(n * D * 2);                                  // LayerNorm-2 dx (post-LN: towards x1; pre-LN: towards s1)
  w.df = d->p_drop > 0.f ? c.take(n * D * 2) : w.dsa;         // ... and its residual-branch gradient
  w.du = c.take(n * F * 2);
  w.dh = c.take(n * D * 2);                                   // gradient of the fc1 input
  w.dxa = c.take(n * D * 2);                                  // LayerNorm-1 dx
  w.da = d->p_drop > 0.f ? c.take(n * D * 2) : w.dxa;
  w.dO = c.take(n * D * 2);
  w.dqkv = c.take(n * 3 * D * 2);
  if (d->tab) w.dgate = (float*)c.take((uint64_t)d->B * d->H * d->T * 4);
  w.ln_b = wavlm_layernorm_bwd_workspace_bytes(d->D); w.ln_ws = c.take(w.ln_b); w.ln_ws2 = c.take(w.ln_b);
  w.cs_b = lin_dx_colsum_ws((int64_t)n, d->F);
  if (w.cs_b < wavlm_colsum_workspace_bytes(d->D)) w.cs_b = wavlm_colsum_workspace_bytes(d->D);
  w.cs_ws = c.take(w.cs_b); w.cs_ws2 = c.take(w.cs_b);
  w.attn_b = wavlm_attn_fused_bwd_workspace_bytes(d->B, d->H, d->T); w.attn_ws = c.take(w.attn_b);
  w.gate_b = wavlm_gate_bwd_workspace_bytes(d->H, 64); w.gate_ws = c.take(w.gate_b);
  const WG it[4] = {{nullptr, nullptr, nullptr, d->D, d->F}, {nullptr, nullptr, nullptr, d->F, d->D},
                    {nullptr, nullptr, nullptr, d->D, d->D}, {nullptr, nullptr, nullptr, 3 * d->D, d->D}};
  w.wg_b = wg_ws_bytes(it, 4, (int64_t)n); w.wg_ws = c.take(w.wg_b);
  w.bytes = c.off;
  if (base && !c.ok) w.bytes = ~(uint64_t)0;
  return w;
}

}  // namespace

extern "C" {

uint64_t wavlm_layer_saved_bytes(const wavlm_layer_desc* d) { return desc_ok(d) ? carve_saved(d, nullptr).bytes : 0; }

uint64_t wavlm_layer_fwd_workspace_bytes(const wavlm_layer_desc* d) {
  if (!desc_ok(d)) return 0;
  return rup256((uint64_t)d->B * d->T * d->D * 2) + (d->saved ? 0 : carve_saved(d, nullptr).bytes);
}

uint64_t wavlm_layer_bwd_workspace_bytes(const wavlm_layer_desc* d) { return desc_ok(d) ? carve_bwd(d, nullptr, 0).bytes : 0; }

int wavlm_encoder_layer_fwd(const wavlm_layer_desc* d, void* stream) {
  if (!desc_ok(d) || !d->x || !d->y || !d->workspace) return WL_EINVAL;
  if (d->pre_ln && !d->r_out) return WL_EINVAL;
  if (d->ws_bytes < wavlm_layer_fwd_workspace_bytes(d)) return WL_EINVAL;
  const int64_t n = (int64_t)d->B * d->T;
  const int D = d->D, F = d->F, pdt = d->param_dtype;
  Carver wc(d->workspace, d->ws_bytes);
  void* tmp = wc.take((uint64_t)n * D * 2);  // out_proj output, then (post-LN) the feed-forward output
  void* sbase = d->saved;
  if (sbase) { if (d->saved_bytes < carve_saved(d, nullptr).bytes) return WL_EINVAL; }
  else sbase = wc.take(carve_saved(d, nullptr).bytes);
  if (!wc.ok) return WL_EINVAL;
  const Saved s = carve_saved(d, sbase);

  // ---- what the attention block reads
  const void* ain = d->x;
  if (d->pre_ln) {
    RC(wavlm_layernorm_fwd(d->x, d->r_in, s.h1, d->r_in ? s.s1 : nullptr, s.mean1, s.rstd1, d->ln1_g, d->ln1_b, n, D, d->eps1,
                           WL_BF16, pdt, 0, d->r_in ? d->p_drop : 0.f, d->seed_r1, 0.f, 0, stream));
    ain = s.h1;
  }
  const float* gate = nullptr;
  if (d->Wgate) {
    RC(wavlm_gate_fwd(ain, d->Wgate, d->bgate, d->grep_a, s.gate, s.ga, s.gb, d->B, d->T, d->H, 64, WL_BF16, pdt, stream));
    gate = s.gate;
  } else if (d->tab) {
    return WL_EINVAL;  // (ungated relative position bias: the caller takes the composed path)
  }
  RC(lin_fwd(ain, d->Wqkv, d->bqkv, s.qkv, n, 3 * D, D, pdt, 0, nullptr, stream));
  // (inference -- d->saved == NULL -- never stores probabilities: nothing will read them)
  RC(wavlm_attn_fused_fwd_p(s.qkv, s.O, s.lse, gate, d->tab, d->kpm, d->saved ? s.pstore : nullptr, s.pstore_b, d->B, d->H, d->T, 64,
                            d->scale, d->p_attn, d->seed_attn, stream));
  RC(lin_fwd(s.O, d->Wo, d->bo, tmp, n, D, D, pdt, 0, nullptr, stream));
  if (!d->pre_ln) {
    RC(wavlm_layernorm_fwd(d->x, tmp, s.x1, s.s1, s.mean1, s.rstd1, d->ln1_g, d->ln1_b, n, D, d->eps1, WL_BF16, pdt, 0,
                           d->p_drop, d->seed_r1, 0.f, 0, stream));
    RC(lin_fwd(s.x1, d->W1, d->b1, s.hact, n, F, D, pdt, 3, s.u, stream));
    RC(lin_fwd(s.hact, d->W2, d->b2, tmp, n, D, F, pdt, 0, nullptr, stream));
    RC(wavlm_layernorm_fwd(s.x1, tmp, d->y, s.s2, s.mean2, s.rstd2, d->ln2_g, d->ln2_b, n, D, d->eps2, WL_BF16, pdt, 0,
                           d->p_drop, d->seed_r2, 0.f, 0, stream));
  } else {
    const void* s1 = d->r_in ? s.s1 : d->x;
    RC(wavlm_layernorm_fwd(s1, tmp, s.x1, d->y, s.mean2, s.rstd2, d->ln2_g, d->ln2_b, n, D, d->eps2, WL_BF16, pdt, 0,
                           d->p_drop, d->seed_r2, 0.f, 0, stream));
    RC(lin_fwd(s.x1, d->W1, d->b1, s.hact, n, F, D, pdt, 3, s.u, stream));
    RC(lin_fwd(s.hact, d->W2, d->b2, d->r_out, n, D, F, pdt, 0, nullptr, stream));
  }
  return WL_OK;
}

int wavlm_encoder_layer_bwd(const wavlm_layer_desc* d, void* stream) {
  if (!desc_ok(d) || !d->x || !d->dy || !d->dx || !d->saved || !d->workspace) return WL_EINVAL;
  if (!d->dWqkv || !d->dbqkv || !d->dWo || !d->dbo || !d->dW1 || !d->db1 || !d->dW2 || (!d->db2 && !d->pre_ln)) return WL_EINVAL;
  if (!d->dln1_g || !d->dln1_b || !d->dln2_g || !d->dln2_b) return WL_EINVAL;
  if (d->Wgate && (!d->dWgate || !d->dbgate || !d->dgrep_a)) return WL_EINVAL;
  if (d->tab && (!d->dtab || !d->Wgate)) return WL_EINVAL;
  if (d->pre_ln && (!d->y || !d->dr_out || (d->r_in && !d->dr_in))) return WL_EINVAL;
  if (d->saved_bytes < carve_saved(d, nullptr).bytes) return WL_EINVAL;
  const int64_t n = (int64_t)d->B * d->T;
  const int D = d->D, F = d->F, pdt = d->param_dtype;
  const Saved s = carve_saved(d, d->saved);
  const BwdWs w = carve_bwd(d, d->workspace, d->ws_bytes);
  if (w.bytes == ~(uint64_t)0) return WL_EINVAL;
  const bool drop = d->p_drop > 0.f;
  // the ~8 finishing launches of the block (LayerNorm parameter sums x 2, three bias column sums, the gate's sums) become
  // ONE at the end: the producers append to this list while it is installed (common.hpp)
  WlFinList fin;
  struct Defer { Defer(WlFinList* l) { wl_fin_defer(l); } ~Defer() { wl_fin_defer(nullptr); } } defer(&fin);

  // ---- feed-forward half
  const void* df;       // gradient of the feed-forward output
  const void* dres;     // gradient reaching the fc1 input past the feed-forward branch (post-LN), NULL for pre-LN
  if (!d->pre_ln) {
    RC(wavlm_layernorm_bwd(d->dy, s.s2, s.mean2, s.rstd2, d->ln2_g, d->ln2_b, w.dsa, drop ? w.df : nullptr, nullptr, d->dln2_g,
                           d->dln2_b, d->db2, n, D, WL_BF16, pdt, 0, d->p_drop, d->seed_r2, 0.f, 0, 1.f, 1, 0, w.ln_ws, w.ln_b, stream));
    df = drop ? w.df : w.dsa;
    dres = w.dsa;
  } else {
    df = d->dr_out;
    dres = nullptr;
    // db2: normally delivered by the LayerNorm that consumed r_out (the next block's LN1 backward through db2_prev, or the
    // encoder's final LayerNorm); a caller whose r_out went elsewhere passes db2 and gets the column sums here
    if (d->db2)
      RC(wavlm_colsum(df, n, D, D, WL_BF16, nullptr, nullptr, d->db2, pdt, 1, w.cs_ws2, w.cs_b, stream));
  }
  RC(lin_dx(df, d->W2, w.du, n, D, F, 4, s.u, nullptr, d->db1, pdt, w.cs_ws, w.cs_b, stream));   // du = (df W2) * gelu'(u); db1 += colsum
  RC(lin_dx(w.du, d->W1, w.dh, n, F, D, 0, nullptr, dres, nullptr, pdt, nullptr, 0, stream));     // dh = du W1 (+ dres)

  // ---- attention half
  const void* da;       // gradient of the out_proj output
  const void* dxres;    // gradient reaching the block input past the attention branch
  if (!d->pre_ln) {
    RC(wavlm_layernorm_bwd(w.dh, s.s1, s.mean1, s.rstd1, d->ln1_g, d->ln1_b, w.dxa, drop ? w.da : nullptr, nullptr, d->dln1_g,
                           d->dln1_b, d->dbo, n, D, WL_BF16, pdt, 0, d->p_drop, d->seed_r1, 0.f, 0, 1.f, 1, 0, w.ln_ws2, w.ln_b, stream));
    da = drop ? w.da : w.dxa;
    dxres = w.dxa;
  } else {
    // LN2 backward: dy = dh (gradient of h2), the residual stream's gradient d->dy is added inside and reaches the branch too
    RC(wavlm_layernorm_bwd(w.dh, d->y, s.mean2, s.rstd2, d->ln2_g, d->ln2_b, w.dsa, drop ? w.da : nullptr, d->dy, d->dln2_g,
                           d->dln2_b, d->dbo, n, D, WL_BF16, pdt, 0, d->p_drop, d->seed_r2, 0.f, 0, 1.f, 1, 1, w.ln_ws, w.ln_b, stream));
    da = drop ? w.da : w.dsa;
    dxres = nullptr;    // (pre-LN: the residual stream's gradient w.dsa joins in the LN1 backward below)
  }
  RC(lin_dx(da, d->Wo, w.dO, n, D, D, 0, nullptr, nullptr, nullptr, pdt, nullptr, 0, stream));
  RC(wl_attn_fused_bwd_ex(s.qkv, s.O, w.dO, s.lse, d->Wgate ? s.gate : nullptr, d->tab, d->kpm, s.pstore, s.pstore_b, w.dqkv, w.dgate, d->dtab,
                          d->dtab_accumulate, d->dbqkv, pdt, 1, d->B, d->H, d->T, 64, d->scale, d->p_attn, d->seed_attn, w.attn_ws,
                          w.attn_b, stream));
  const void* ain = d->pre_ln ? s.h1 : d->x;
  void* dain = d->pre_ln ? w.dxa : d->dx;   // gradient of the attention block's input
  RC(lin_dx(w.dqkv, d->Wqkv, dain, n, 3 * D, D, 0, nullptr, dxres, nullptr, pdt, nullptr, 0, stream));

  // ---- the four weight gradients (arrival order of the composed path: fc2, fc1, out_proj, q|k|v)
  const WG it[4] = {{df, s.hact, d->dW2, D, F}, {w.du, s.x1, d->dW1, F, D}, {da, s.O, d->dWo, D, D}, {w.dqkv, ain, d->dWqkv, 3 * D, D}};
  RC(wgrads(it, 4, n, pdt, w.wg_ws, w.wg_b, stream));

  if (d->Wgate)
    RC(wavlm_gate_bwd(w.dgate, ain, d->Wgate, d->grep_a, s.ga, s.gb, dain, d->dWgate, d->dbgate, d->dgrep_a, d->B, d->T, d->H, 64,
                      WL_BF16, pdt, 1 | 2, w.gate_ws, w.gate_b, stream));
  if (d->pre_ln) {
    // LN1 backward: dy = gradient of h1; the residual stream's gradient (w.dsa) is added inside and reaches r_in as well
    const void* s1 = d->r_in ? s.s1 : d->x;
    RC(wavlm_layernorm_bwd(w.dxa, s1, s.mean1, s.rstd1, d->ln1_g, d->ln1_b, d->dx, (d->r_in && drop) ? d->dr_in : nullptr, w.dsa,
                           d->dln1_g, d->dln1_b, d->r_in ? d->db2_prev : nullptr, n, D, WL_BF16, pdt, 0, d->r_in ? d->p_drop : 0.f,
                           d->seed_r1, 0.f, 0, 1.f, 1, d->r_in ? 1 : 0, w.ln_ws2, w.ln_b, stream));
    if (d->r_in && !drop)   // without dropout the branch gradient IS dx
      RC(wavlm_axpby(d->dx, WL_BF16, d->dr_in, WL_BF16, n * D, 1.f, 0.f, stream));
  }
  wl_fin_defer(nullptr);
  return wl_fin_flush(fin, stream);
}

}  // extern "C"
