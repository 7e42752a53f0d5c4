/*
 * wavlm_hip.h -- C ABI of libwavlm_hip.so: the MI355X (gfx950) kernels of the WavLM / UniSpeech
 * pre-training hot path.
 *
 * The reference (microsoft/UniSpeech) has NO native interface on this path: every FLOP is a stock
 * PyTorch op called from Python.  Each entry point below therefore cites the reference *Python*
 * call site whose arithmetic it replaces (paths relative to the reference root).  A maintainer
 * binds the library with ctypes (INTEGRATION.md shows the stub); unispeech_amd/_lib.py is that
 * binding.
 *
 * Conventions (all entry points):
 *   - plain device pointers + sizes; no torch types; `stream` is a hipStream_t passed as void*
 *   - returns 0 on success, <0 on error (-1 bad argument, -2 launch failure); never throws,
 *     never allocates, never synchronises -- with ONE exception: the first GELU-epilogue launch on a device fills a
 *     32 KiB chord table and waits for it (once per device and process; every later launch reads one atomic pointer)
 *   - dtype codes: 0 = float32, 1 = bfloat16 (raw 16-bit)
 *   - activations are channel-last: [B, T, C] row-major ("rows" = frames, contiguous channels)
 */
#ifndef WAVLM_HIP_H
#define WAVLM_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define WAVLM_HIP_ABI_VERSION 21
int wavlm_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Dense contraction (MFMA for bf16, exact-f32 VALU tile for the parity mode).
 *   C[zo,zi][m,n] = epi( alpha * sum_{kb<KB} sum_{k<K} A[zo,zi,kb](m,k) * B[zo,zi,kb](n,k) + bias[n] )
 * A(m,k) lives at A + zo*sA_o + zi*sA_i + kb*sA_kb + (transA ? k*lda + m : m*lda + k); same for B.
 * lda may be SMALLER than K (overlapping rows): that is how a strided Conv1d over a channel-last
 * activation becomes a GEMM with no im2col.
 * Replaces: F.linear (WavLM/WavLM.py:348, modules.py:540-563 q/k/v/out, WavLM.py:671-672 fc1/fc2),
 * nn.Conv1d of the feature extractor (WavLM/WavLM.py:401,499-500), the grouped pos_conv
 * (WavLM/WavLM.py:514-527), torch.bmm inside F.multi_head_attention_forward, the cosine-logit
 * product (src/fairseq/models/wavlm/wavlm.py:431) and all their autograd backward contractions.
 * epi: 0 none | 1 gelu (pre-activation stored to aux if aux != NULL) | 2 multiply by gelu'(aux)
 *      | 3 gelu (gelu'(pre-activation) stored to aux if aux != NULL) | 4 multiply by aux
 *      (3 / 4 are the training pair: forward pays three extra VALU ops per element, backward saves an erf + exp)
 * then: + res (if res != NULL), + old C (if accumulate).
 * split_k > 1: the KB range is cut into split_k slabs written to `workspace` (f32) and summed by a
 * second kernel (deterministic, no atomics).  workspace bytes >= wavlm_gemm_workspace_bytes().
 * ------------------------------------------------------------------------------------------ */
typedef struct wavlm_gemm_desc {
  int32_t dtype;      /* element type of A and B */
  int32_t c_dtype;    /* element type of C */
  int32_t M, N, K, KB;
  int32_t transA, transB;
  int64_t lda, ldb, ldc;
  int64_t sA_kb, sB_kb;
  int32_t batch_o, batch_i;
  int64_t sA_o, sA_i, sB_o, sB_i, sC_o, sC_i;
  const void* A;
  const void* B;
  void* C;
  float alpha;
  int32_t epi;
  const void* bias; int32_t bias_dtype; int64_t sBias_o, sBias_i;
  void* aux; int32_t aux_dtype; int64_t ld_aux, sAux_o, sAux_i;
  const void* res; int32_t res_dtype; int64_t ld_res, sRes_o, sRes_i;
  int32_t accumulate;
  int32_t split_k;
  void* workspace; uint64_t ws_bytes;
  /* optional: colsum[n] (+)= sum_m C[m][n] over the rows of the (un-batched, un-split) result, in colsum_dtype -- the bias
   * gradient of the linear whose output gradient C is (fc1's from fc2's GELU'-multiplying dX GEMM, WavLM/WavLM.py:732-737).
   * Computed inside the GEMM's epilogue where the kernel supports it (per-tile partial rows in `workspace` + one finishing
   * launch), otherwise by a column-sum pass over C; either way `workspace` must hold wavlm_gemm_workspace_bytes(). */
  void* colsum; int32_t colsum_dtype; int32_t colsum_accumulate;
} wavlm_gemm_desc;

uint64_t wavlm_gemm_workspace_bytes(const wavlm_gemm_desc* d);
int wavlm_gemm(const wavlm_gemm_desc* d, void* stream);
/* n descriptors in one call.  Weight-gradient-shaped problems (transA && transB, same K and split_k >= 2, no epilogue
 * extras, each with its own workspace) run as ONE grouped split-K launch + one slab reduction each -- the four dW of an
 * encoder layer (modules.py q/k/v/out_proj, WavLM.py:732-737 fc1/fc2) share the reduction length B*T; anything else is
 * executed as n wavlm_gemm calls.  Results are identical either way.  Callers should pick `split_k` so that all members' tiles x
 * split_k fit ONE round of the persistent grid (256 CUs minus wavlm_set_reserved_cus: what unispeech_amd/ops.py grouped_split
 * and csrc/layer.hip do).  Under WAVLM_WGRAD_STREAMK=1 `split_k` may be one larger than that (= the fp32 slabs the workspace
 * holds): the library then runs a balanced partition in which the CUs the one-round split leaves idle take the K tail of
 * every tile (measured neutral: off by default). */
int wavlm_gemm_grouped(const wavlm_gemm_desc* d, int32_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Row kernels (one 64-lane wave per row, wave-shuffle reductions).
 * ------------------------------------------------------------------------------------------ */

/* y = dropout_out( act( LayerNorm( x + dropout_in(r) ) ) ).  r may be NULL.  s (optional) receives the
 * pre-norm sum, mean/rstd (optional) the row statistics -- backward needs all three.  act: 0 none, 1 gelu.
 * Replaces nn.LayerNorm / Fp32LayerNorm + the residual add + nn.Dropout around it
 * (WavLM/WavLM.py:342, 582-584, 702-703, 726-729, 739-740; WavLM/modules.py:30-42). D % 8 == 0, D <= 2048. */
int wavlm_layernorm_fwd(const void* x, const void* r, void* y, void* s, float* mean, float* rstd, const void* gamma,
                        const void* beta, int64_t rows, int32_t D, float eps, int32_t dtype, int32_t param_dtype,
                        int32_t act, float p_in, uint64_t seed_in, float p_out, uint64_t seed_out, void* stream);
uint64_t wavlm_layernorm_bwd_workspace_bytes(int32_t D);
/* dx: gradient of x (and of the sum s); dr (optional): gradient of r (dx through the input-dropout mask);
 * dgamma/dbeta in param dtype; dy is scaled by grad_scale first (GradMultiply, WavLM/modules.py:60-69).
 * dr_colsum (optional, [D], param dtype, same accumulate flag): column sums of the gradient of r, i.e. the bias
 * gradient of the nn.Linear whose output r is (out_proj / fc2 of a post-LN block, WavLM/WavLM.py:726-740) -- it falls
 * out of the pass that already reduces dgamma / dbeta over rows.
 * dx_add (optional, same shape / dtype as dx): added into dx -- the gradient that reaches x past the LayerNorm, i.e. the
 * residual stream of a layer_norm_first block (WavLM/WavLM.py:702-724: `residual = x; x = layer_norm(x); ...; x =
 * residual + x`), so the two contributions never meet in a separate add kernel.  dr / dr_colsum do not include it unless
 * dr_incl_add != 0: then the SUM s = x + dropout(r) is itself the residual stream (a pre-LN block whose residual add is
 * fused into the following LayerNorm) and the gradient arriving at s reaches r as well (through the dropout mask). */
int wavlm_layernorm_bwd(const void* dy, const void* s, const float* mean, const float* rstd, const void* gamma,
                        const void* beta, void* dx, void* dr, const void* dx_add, void* dgamma, void* dbeta,
                        void* dr_colsum, int64_t rows, int32_t D, int32_t dtype, int32_t param_dtype, int32_t act,
                        float p_in, uint64_t seed_in, float p_out, uint64_t seed_out, float grad_scale,
                        int32_t accumulate_params, int32_t dr_incl_add, void* workspace, uint64_t ws_bytes, void* stream);
/* The same with a segmented dx: row r of the b-th run of dx_seg_rows rows is written at row r + b * dx_seg_gap of dx (the
 * caller zero-fills the gaps) -- the zero-padded per-utterance layout that the data-gradient GEMMs of the conv layer in
 * front of this LayerNorm read (layer_norm extractor mode: WavLM/WavLM.py:403-418), so that no padded copy of the gradient is
 * made.  dx_seg_rows = 0: contiguous.  D in {512, 768, 1024} only (WL_EINVAL otherwise). */
int wavlm_layernorm_bwd_seg(const void* dy, const void* s, const float* mean, const float* rstd, const void* gamma,
                            const void* beta, void* dx, void* dr, const void* dx_add, void* dgamma, void* dbeta,
                            void* dr_colsum, int64_t rows, int32_t D, int32_t dtype, int32_t param_dtype, int32_t act,
                            float p_in, uint64_t seed_in, float p_out, uint64_t seed_out, float grad_scale,
                            int32_t accumulate_params, int32_t dr_incl_add, int32_t dx_seg_rows, int32_t dx_seg_gap,
                            void* workspace, uint64_t ws_bytes, void* stream);

/* out[c] (+)= sum over rows of x[row, c]; a row counts iff (!include || include[row]) && (!exclude || !exclude[row]).
 * Bias gradients of every nn.Linear / conv bias, and d(mask_emb) (src/fairseq/models/wavlm/wavlm.py:401). */
uint64_t wavlm_colsum_workspace_bytes(int32_t N);
int wavlm_colsum(const void* x, int64_t rows, int32_t N, int64_t ld, int32_t dtype, const uint8_t* include_mask,
                 const uint8_t* exclude_mask, void* out, int32_t out_dtype, int32_t accumulate, void* workspace,
                 uint64_t ws_bytes, void* stream);

/* y[row] = zero[row] ? 0 : (sel[row] ? emb (0 if emb == NULL) : x[row]).  apply_mask's x[mask] = mask_emb and the
 * encoder's x[padding_mask] = 0 in one pass (WavLM/WavLM.py:286, 574-575), and their backward. */
int wavlm_select_rows(const void* x, void* y, const uint8_t* sel, const void* emb, const uint8_t* zero, int64_t rows,
                      int32_t D, int32_t dtype, int32_t emb_dtype, void* stream);
/* dst[i] = idx[i] >= 0 ? src[idx[i]] : 0.  x[masked_indices] (wavlm.py:541,557) and its scatter-back. */
int wavlm_gather_rows(const void* src, const int32_t* idx, void* dst, int64_t n_out, int32_t D, int32_t dtype,
                      void* stream);
/* Operand images of the feature extractor's Conv1d weights (WavLM/WavLM.py:378-504: conv_layers.{i}.0.weight
 * [Cout, Cin, k]) for every layer of the stack in ONE launch: Wf[l] = [Cout][k * Cin] (forward GEMM over overlapping
 * rows) and, if Wb[l] != NULL, the s[l] stride-phase images of the data-gradient GEMMs back to back, phase r as
 * [Cin][J_r * Cout] with J_r = ceil((k - r) / s) taps newest first.  wavlm_conv_wgrad_scatter is the way back for the
 * weight gradients: W[l][co][ci][kk] (+)= Wf[l][co][kk * Cin + ci] (here W = the gradient tensors, Wf = the GEMM outputs). */
#define WL_CONV_RELAYOUT_MAX 8
typedef struct {
  const void* W[WL_CONV_RELAYOUT_MAX];
  void* Wf[WL_CONV_RELAYOUT_MAX];
  void* Wb[WL_CONV_RELAYOUT_MAX];
  int32_t Cout[WL_CONV_RELAYOUT_MAX], Cin[WL_CONV_RELAYOUT_MAX], k[WL_CONV_RELAYOUT_MAX], s[WL_CONV_RELAYOUT_MAX];
  int32_t n_layers, dtype;
} wavlm_conv_relayout_desc;
int wavlm_conv_weights_relayout(const wavlm_conv_relayout_desc* d, void* stream);
int wavlm_conv_wgrad_scatter(const wavlm_conv_relayout_desc* d, int32_t accumulate, void* stream);
/* y = a*x + b*y */
int wavlm_axpby(const void* x, int32_t x_dtype, void* y, int32_t y_dtype, int64_t n, float a, float b, void* stream);
/* y *= scalar[0] * extra, scalar on the device (upstream loss gradient without a host sync) */
int wavlm_scale_dev(void* y, int32_t dtype, int64_t n, const float* scalar, float extra, void* stream);
/* y = dropout(x; p, seed): counter-based (Philox4x32-10) mask, regenerated (not stored) for the backward. */
int wavlm_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, int32_t dtype, void* stream);
/* y = x + dropout(r, p, seed): residual add of a pre-LN block (unispeech_sat.py:1088-1111 `x = residual + dropout(x)`);
 * same mask as wavlm_dropout for the same seed (the backward of the dropped branch is wavlm_dropout(dy)) */
int wavlm_dropout_add(const void* x, const void* r, void* y, int64_t n, float p, uint64_t seed, int32_t dtype, void* stream);
/* out[0] = scale * sum(x^2): features_pen (wavlm.py:486) and the global gradient norm (utils.py:338-388). */
uint64_t wavlm_sumsq_workspace_bytes(void);
int wavlm_sumsq(const void* x, int32_t dtype, int64_t n, float scale, float* out, void* workspace, uint64_t ws_bytes,
                void* stream);

/* ------------------------------------------------------------------------------------------
 * conv0: Conv1d(1->C, k=10, stride) + GroupNorm(C, C) + GELU, channel-last output [B, T0, C]
 * (WavLM/WavLM.py:401-428 block(is_group_norm=True); Fp32GroupNorm WavLM/modules.py:45-57).
 * stats[B, C, 2] = (mean, rstd) saved for backward.  kw must be 10, C <= 512.
 * ------------------------------------------------------------------------------------------ */
uint64_t wavlm_conv0_gn_workspace_bytes(int32_t B, int64_t T, int32_t C, int32_t stride);
int wavlm_conv0_gn_gelu_fwd(const void* wav, int32_t wav_dtype, const void* W, const void* gamma, const void* beta,
                            int32_t param_dtype, void* out, int32_t out_dtype, float* stats, int32_t B, int64_t T,
                            int32_t C, int32_t kw, int32_t stride, float eps, void* workspace, uint64_t ws_bytes,
                            void* stream);
uint64_t wavlm_conv0_gn_bwd_workspace_bytes(int32_t B, int64_t T, int32_t C, int32_t stride);
int wavlm_conv0_gn_gelu_bwd(const void* wav, int32_t wav_dtype, const void* W, const void* gamma, const void* beta,
                            int32_t param_dtype, const void* g, int32_t g_dtype, const float* stats, void* dW,
                            void* dgamma, void* dbeta, int32_t B, int64_t T, int32_t C, int32_t kw, int32_t stride,
                            float gscale, void* workspace, uint64_t ws_bytes, void* stream);

/* extractor_mode = "layer_norm" (WavLM-Large), block 0: Conv1d(1 -> C, k = 10) -> LayerNorm over the C channels of each
 * frame -> GELU, channel-last (WavLM/WavLM.py:403-418 with mode "layer_norm"; Fp32LayerNorm WavLM/modules.py:31-43).
 * One fused pass forward; backward (dW, dgamma, dbeta; no input gradient) recomputes conv and statistics.  With bf16
 * operands and C = 512 both run on the matrix cores; their frame statistics come from second moments of the parameters over
 * the channels, which the forward keeps in its (small) workspace: wavlm_conv0_ln_fwd_workspace_bytes() bytes. */
uint64_t wavlm_conv0_ln_fwd_workspace_bytes(void);
int wavlm_conv0_ln_gelu_fwd(const void* wav, int32_t wav_dtype, const void* W, const void* conv_bias, const void* gamma,
                            const void* beta, int32_t param_dtype, void* out, int32_t out_dtype, int32_t B, int64_t T,
                            int32_t C, int32_t kw, int32_t stride, float eps, void* workspace, uint64_t ws_bytes,
                            void* stream);
uint64_t wavlm_conv0_ln_bwd_workspace_bytes(int32_t B, int64_t T, int32_t C, int32_t stride);
/* conv_bias / dconv_bias: the Conv1d bias of conv_bias=True configurations and its gradient (both optional, [C]) */
int wavlm_conv0_ln_gelu_bwd(const void* wav, int32_t wav_dtype, const void* W, const void* conv_bias, const void* gamma,
                            const void* beta, int32_t param_dtype, const void* g, int32_t g_dtype, void* dW,
                            void* dconv_bias, void* dgamma, void* dbeta, int32_t B, int64_t T, int32_t C, int32_t kw,
                            int32_t stride, float eps, float gscale, void* workspace, uint64_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Gated relative-position-bias attention (WavLM/modules.py:417-455, 504-563).
 * ------------------------------------------------------------------------------------------ */
/* rel[h][d] = emb[bucket[d]][h], d = (j - i) + T - 1 in [0, 2T-2]; bucket[] is the host-computed int table */
int wavlm_relpos_gather(const void* emb, int32_t emb_dtype, const int32_t* bucket, float* tab, int32_t H, int32_t L,
                        void* stream);
int wavlm_relpos_scatter(const float* dtab, const int32_t* bucket, void* demb, int32_t emb_dtype, int32_t H, int32_t L,
                         int32_t num_buckets, void* stream);
/* gate[b,h,t] = ga*(gb*grep_a[h] - 1) + 2 from the un-projected layer input x[B,T,H*hd] (modules.py:523-533) */
int wavlm_gate_fwd(const void* x, const void* W, const void* bias, const void* grep_a, float* gate, float* ga, float* gb,
                   int32_t B, int32_t T, int32_t H, int32_t hd, int32_t dtype, int32_t param_dtype, void* stream);
uint64_t wavlm_gate_bwd_workspace_bytes(int32_t H, int32_t hd);
int wavlm_gate_bwd(const float* dgate, const void* x, const void* W, const void* grep_a, const float* ga,
                   const float* gb, void* dx, void* dW, void* dbias, void* dgrep_a, int32_t B, int32_t T, int32_t H,
                   int32_t hd, int32_t dtype, int32_t param_dtype, int32_t accumulate_params, void* workspace,
                   uint64_t ws_bytes, void* stream);  /* accumulate_params bit 0: dW / dbias / dgrep_a (+)= (gradient sinks);
                                           bit 1: dx (+)= (dx already holds the gradient another consumer of x produced) */
/* P = dropout(softmax_j(S + gate_i*rel[h, j-i] + keypad)); lse saved.  S: [B*H, T, ldS], P: [B*H, T, ldP]; T <= 1024 */
int wavlm_attn_softmax_fwd(const void* S, void* P, float* lse, const float* gate, const float* tab, const uint8_t* kpm,
                           int32_t B, int32_t H, int32_t T, int64_t ldS, int64_t ldP, int32_t s_dtype, int32_t p_dtype,
                           float p_drop, uint64_t seed, void* stream);
uint64_t wavlm_attn_softmax_bwd_workspace_bytes(int32_t B, int32_t H, int32_t T);
int wavlm_attn_softmax_bwd(const void* S, const void* dP, const float* lse, const float* gate, const float* tab,
                           const uint8_t* kpm, void* dS, float* dgate, float* dtab, int32_t dtab_accumulate, int32_t B,
                           int32_t H, int32_t T, int64_t ldS, int64_t ldP, int32_t s_dtype, int32_t p_dtype,
                           float p_drop, uint64_t seed, void* workspace, uint64_t ws_bytes, void* stream);

/* Fused attention (bf16, head_dim 64): scores, Toeplitz bias, key padding, online softmax, dropout and PV in one
 * kernel; nothing of size [B*H, T, T] is written.  qkv: packed [B, T, 3*H*64]; O: [B, T, H*64]; lse: [B*H, T].
 * Replaces F.multi_head_attention_forward + the materialised gated bias (WavLM/modules.py:504-563). */
int wavlm_attn_fused_fwd(const void* qkv, void* O, float* lse, const float* gate, const float* tab, const uint8_t* kpm,
                         int32_t B, int32_t H, int32_t T, int32_t head_dim, float scale, float p_drop, uint64_t seed,
                         void* stream);
uint64_t wavlm_attn_fused_bwd_workspace_bytes(int32_t B, int32_t H, int32_t T);
/* dqkv [B, T, 3*H*64], dgate [B, H, T], dtab [H, 2T-1] from dO and the forward's (qkv, O, lse).
 * dbias (optional, [3*H*64], dbias_dtype, (+)= if dbias_accumulate): column sums of dqkv over all B*T rows, i.e. the bias
 * gradient of the packed q|k|v projection (WavLM/modules.py:504-520 in_proj_bias) -- it falls out of the kernels that produce
 * dq / dk / dv (per-block column sums + one finishing launch) instead of a separate pass over dqkv. */
int wavlm_attn_fused_bwd(const void* qkv, const void* O, const void* dO, const float* lse, const float* gate,
                         const float* tab, const uint8_t* kpm, void* dqkv, float* dgate, float* dtab, void* dbias,
                         int32_t dbias_dtype, int32_t dbias_accumulate, int32_t B, int32_t H, int32_t T, int32_t head_dim,
                         float scale, float p_drop, uint64_t seed, void* workspace, uint64_t ws_bytes, void* stream);
/* The same pair with STORED PROBABILITIES (round 5): the forward additionally writes its softmax probabilities -- fp16,
 * relative to the running row maximum of their key tile, dropout decision in the sign bit -- and those maxima into the opaque
 * `pstore` (wavlm_attn_fused_pstore_bytes: 4 KiB per 32 query rows x 64 keys, 453 MB per layer at B = 32, H = 12, T = 749),
 * and the backward kernels read them instead of recomputing scores, bias, exponentials and dropout words from (q, k, lse,
 * seed): the element pass of both backward kernels shrinks to convert / scale / multiply and each loses its score MFMA.
 * The reference keeps the same tensor (`attn_output_weights` after dropout, F.multi_head_attention_forward under
 * WavLM/modules.py:504-563, fp32 / fp16 [B*H, T, T]) -- here it is a memory-for-VALU trade the 288 GB part can afford.
 * wavlm_attn_fused_pstore_bytes returns 0 for T > 1024: recompute only.
 * pstore == NULL: exactly wavlm_attn_fused_fwd / _bwd (recompute; the low-memory mode).  A backward with pstore must be
 * given the pstore its own forward wrote (same B, H, T, gate, tab, kpm, p_drop, seed).
 * BIT mode (ABI 21, round 6; opt-in, measured neutral): a pstore of exactly wavlm_attn_fused_dbits_bytes (one bit per (row, key) of
 * the padded grid: 27 MB per layer at B = 32, H = 12, T = 749; 256-byte aligned) makes the forward store only its dropout
 * DECISIONS -- it evaluates the hash anyway -- and both backward kernels select with the stored bits (two instructions per
 * element) instead of evaluating the hash twice more; scores, bias and exponentials are recomputed as without a pstore.  The
 * reference keeps the dropout mask inside F.dropout's autograd node (WavLM/modules.py:504-563 under
 * F.multi_head_attention_forward, one byte per element). */
uint64_t wavlm_attn_fused_pstore_bytes(int32_t B, int32_t H, int32_t T);
uint64_t wavlm_attn_fused_dbits_bytes(int32_t B, int32_t H, int32_t T);
int wavlm_attn_fused_fwd_p(const void* qkv, void* O, float* lse, const float* gate, const float* tab, const uint8_t* kpm,
                           void* pstore, uint64_t pstore_bytes, int32_t B, int32_t H, int32_t T, int32_t head_dim, float scale,
                           float p_drop, uint64_t seed, void* stream);
int wavlm_attn_fused_bwd_p(const void* qkv, const void* O, const void* dO, const float* lse, const float* gate,
                           const float* tab, const uint8_t* kpm, const void* pstore, uint64_t pstore_bytes, void* dqkv,
                           float* dgate, float* dtab, void* dbias, int32_t dbias_dtype, int32_t dbias_accumulate, int32_t B,
                           int32_t H, int32_t T, int32_t head_dim, float scale, float p_drop, uint64_t seed, void* workspace,
                           uint64_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * pos_conv weight side: weight_norm(dim=2) -> GEMM weight images, and its backward (WavLM/WavLM.py:514-527)
 * ------------------------------------------------------------------------------------------ */
uint64_t wavlm_posconv_weight_workspace_bytes(int32_t D, int32_t Cg, int32_t K);
/* w[co, ci, k] = g[k] * v[co, ci, k] / ||v[:, :, k]|| as the two operand images of the convolution (Wf: forward; Wb:
 * gradient of x, taps flipped, channels / columns swapped), G * Cg * K * Cg elements each.  layout 0: [g][column][tap][channel]
 * (B operand of the overlapping-row GEMM form); layout 1 (K % 16 == 0, Cg % 8 == 0): the image wavlm_posconv_direct
 * streams, [g][channel / 8][(tap % 16) / 4][tap / 16][tap % 4][column][channel % 8]. */
int wavlm_posconv_weight_fwd(const void* v, const void* g, int32_t param_dtype, void* Wf, void* Wb, int32_t out_dtype,
                             float* norm, int32_t D, int32_t Cg, int32_t K, int32_t layout, void* workspace,
                             uint64_t ws_bytes, void* stream);
/* dWf: fp32 gradient of the layout-0 forward image, as `nsplit` slabs of G*Cg*K*Cg elements that are summed first
 * (1 slab from the GEMM form, wavlm_posconv_dw_direct_splits() from the direct kernel) -> dv, dg in param dtype */
int wavlm_posconv_weight_bwd(const float* dWf, const void* v, const void* g, const float* norm, int32_t param_dtype,
                             void* dv, void* dg, int32_t D, int32_t Cg, int32_t K, int32_t nsplit, void* workspace,
                             uint64_t ws_bytes, void* stream);
/* x[B,T,D] (optionally * gelu'(aux)) -> group-major, time-padded out[B,G,Tp,D/G]; nat_out optional natural copy */
int wavlm_posconv_group_major(const void* x, const void* aux, void* out, void* nat_out, int32_t B, int32_t T, int32_t D,
                              int32_t G, int32_t left_pad, int32_t Tp, int32_t dtype, int32_t aux_is_grad, void* stream);

/* The grouped convolution itself as a direct convolution (bf16; Cg = 48 or 64; K = 128): one workgroup per (batch,
 * group, frame segment) keeps its input window in LDS, the group's weights stream through it.
 *   out[b, t, g*Cg + n] = res[b, t, g*Cg + n] + f( sum_{tap, ci} xg[b, g, t + tap, ci] * w[g, n, tap, ci] + bias[g*Cg + n] )
 * xg [B, G, Tp, Cg] is wavlm_posconv_group_major's output (Tp >= T + K - 1), W one of wavlm_posconv_weight_fwd's LAYOUT-1 images.
 * gelu != 0: f = GELU and aux (optional, [B, T, G*Cg]) receives the pre-activation for the backward pass -- the
 * forward, nn.Conv1d(groups=16, k=128) + SamePad + GELU + the residual add of WavLM/WavLM.py:577-579; gelu == 0 with
 * W = Wb over the group-major dy * gelu': the gradient of x.  bias / res / aux may be NULL.
 * wavlm_posconv_direct_supported tells whether the shape is covered (otherwise: wavlm_gemm, overlapping-row form). */
int wavlm_posconv_direct_supported(int32_t Cg, int32_t K, int32_t T);
int wavlm_posconv_direct(const void* xg, const void* W, const void* bias, const void* res, void* out, void* aux,
                         int32_t B, int32_t G, int32_t T, int32_t Tp, int32_t Cg, int32_t K, int32_t gelu, void* stream);
/* The weight gradient of the same convolution, directly from the two group-major copies (bf16; Cg = 48 / 64; K = 128):
 *   part[s][g][n][tap][ci] = sum over the s-th part of the batch and all t of xg[b, g, t + tap, ci] * dug[b, g, du_off + t, n]
 * fp32, wavlm_posconv_dw_direct_splits(Cg, G) slabs (0: shape not covered) for wavlm_posconv_weight_bwd to add up.
 * Replaces the weight-gradient pass of nn.Conv1d(groups=16, k=128) in WavLM/WavLM.py:514-527's backward. */
int wavlm_posconv_dw_direct_splits(int32_t Cg, int32_t G);
int wavlm_posconv_dw_direct(const void* xg, const void* dug, float* part, int32_t B, int32_t G, int32_t T, int32_t Tp,
                            int32_t du_off, int32_t Cg, int32_t K, void* stream);

/* ------------------------------------------------------------------------------------------
 * Masked-prediction loss (src/fairseq/models/wavlm/wavlm.py:426-438; criterions/wavlm_criterion.py:52-138)
 * ------------------------------------------------------------------------------------------ */
int wavlm_l2norm_fwd(const void* x, int32_t x_dtype, void* y, int32_t y_dtype, float* inv_norm, int64_t rows,
                     int32_t D, float eps, void* stream);
int wavlm_l2norm_bwd(const void* dy, const void* y, int32_t y_dtype, const float* inv_norm, void* dx, int32_t x_dtype,
                     int64_t rows, int32_t D, void* stream);
/* per row: loss = logsumexp(logits) - logits[target]; correct = (logits[target] is the row max);
 * dlogits (optional) = weight * (softmax - onehot), zero-filled up to ld_dlogits */
int wavlm_ce_rows(const float* logits, const int32_t* target, float* loss_rows, float* correct_rows, void* dlogits,
                  int32_t d_dtype, int64_t S, int32_t V, int64_t ld_logits, int64_t ld_dlogits, float weight,
                  void* stream);

/* Sampled-instance cosine logits + BCE: the utterance-contrastive head of UniSpeech-SAT
 * (src/fairseq/models/unispeech_sat/unispeech_sat.py:487-557 sample_instances / compute_nce, 701-737 compute_pred_spk;
 * the same gathered-cosine shape as wav2vec 2.0's sample_negatives, models/wav2vec/wav2vec2.py:474-553).
 * X, Y: L2-normalised rows (X = Y for UniSpeech-SAT); idx[s, n]: row of Y gathered for logit n of row s (the host draws
 * the indices with the reference's torch.randint calls); out[s, n] = scale * <X[s], Y[idx[s, n]]>; mask_equal: logits
 * n >= 1 whose gathered row equals the row of column 0 become -inf (wav2vec 2.0's neg_is_pos, wav2vec2.py:535-551).  rows_wsum is both halves of its backward:
 * out[j] (+)= sum_{e in [off[j], off[j+1])} w[e] * Y[src[e]]. */
int wavlm_gather_dot(const void* X, const void* Y, int32_t dtype, const int32_t* idx, float* out, int64_t S, int32_t N,
                     int32_t D, float scale, int32_t mask_equal, void* stream);
int wavlm_rows_wsum(const void* Y, int32_t dtype, const int32_t* src, const float* w, const int32_t* off, void* out,
                    int32_t out_dtype, int64_t rows, int32_t D, int32_t accumulate, void* stream);
uint64_t wavlm_bce_workspace_bytes(void);
int wavlm_bce_logits(const float* logits, const uint8_t* targets, float* dlogits, float* out, int64_t n, float gscale,
                     void* workspace, uint64_t ws_bytes, void* stream);
/* Gated linear unit over the last dimension: y[rows, F] = x[:, :F] * g(x[:, F:]) and its backward dx[rows, 2F].
 * gate: 0 sigmoid = nn.GLU (target_glu, src/fairseq/models/wavlm/wavlm.py:322-327, 529-531: Linear(F, 2F) + GLU on the label
 * embeddings) | 1 swish (GLU_Linear(.., "swish"): the feed-forward fc1 under activation_fn = "glu", WavLM/modules.py:99-129,
 * WavLM/WavLM.py:668-669, 707-708) | 2 relu | 3 gelu | 4 bilinear. */
int wavlm_glu_fwd(const void* x, void* y, int64_t rows, int32_t F, int32_t dtype, int32_t gate, void* stream);
int wavlm_glu_bwd(const void* x, const void* dy, void* dx, int64_t rows, int32_t F, int32_t dtype, int32_t gate, void* stream);
/* Elementwise feed-forward activations other than the erf GELU of the GEMM epilogues (utils.get_activation_fn,
 * src/fairseq/utils.py:533-555; WavLM/modules.py:144-160): kind 1 relu | 2 gelu_accurate (tanh form,
 * src/fairseq/modules/gelu.py:14-19) | 3 tanh | 4 erf gelu.  The backward takes the pre-activation x: dx = dy * act'(x). */
int wavlm_act_fwd(const void* x, void* y, int64_t n, int32_t dtype, int32_t kind, void* stream);
int wavlm_act_bwd(const void* x, const void* dy, void* dx, int64_t n, int32_t dtype, int32_t kind, void* stream);
uint64_t wavlm_sum_workspace_bytes(void);
int wavlm_sum_f32(const float* x, int64_t n, float* out, void* workspace, uint64_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused optimizer step (src/fairseq/optim/adam.py:148-228, fp16_optimizer.py:106-218, utils.py:338-388)
 * ------------------------------------------------------------------------------------------ */
/* grad is scaled by grad_mult (* grad_mult_dev[0] when given: a device scalar such as 1 / global sample size) and
 * clipped to max_norm using the device scalar *gnorm_sq (sum of squares of the unscaled gradient) -- no host sync. */
int wavlm_adam_step(float* p, float* m, float* v, const void* grad, int32_t grad_dtype, void* p_lowp,
                    int32_t lowp_dtype, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                    int64_t step, float grad_mult, const float* grad_mult_dev, const float* gnorm_sq, float max_norm,
                    void* stream);

/* ------------------------------------------------------------------------------------------
 * Utterance / noise mixing of a collated batch (replaces the per-sample numpy loop of
 * src/fairseq/data/audio/utterance_mixing_dataset.py:373-438 mixing_collated_audios; all random draws stay on the host).
 * dst[B, T] (fp32, != src) = src with `ops` applied in the reference's in-place row order; ops: n_ops x 8 int32
 * (row, kind 0 = batch row / 1 = noise segment, src row | noise offset, c_start, s_start, c_len, src length, float32
 * bits of 10^(snr/10)), sorted by row; op_begin[B + 1].  scale = sqrt(mean(dst_row^2) / (mean(src^2) * gain)), 0 if the
 * source power is 0.  normalize: rows that were mixed get (x - mean) / sqrt(var + eps) over the whole row (F.layer_norm).
 * dst_lowp (optional): bf16 copy of the result (the Trainer's waveform cast, trainer.py:1141-1152).
 * ------------------------------------------------------------------------------------------ */
uint64_t wavlm_mix_workspace_bytes(int32_t B, int64_t T);
int wavlm_mix_utterances(const float* src, float* dst, void* dst_lowp, int32_t B, int64_t T, const int32_t* ops,
                         int32_t n_ops, const int32_t* op_begin, const float* noise, int32_t normalize, float eps,
                         void* workspace, uint64_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Gumbel-softmax vector quantiser on projected logits [n, G*V] (replaces the per-row part of
 * src/fairseq/modules/gumbel_vector_quantizer.py:157-213; the weight projection is a wavlm_gemm, the codebook product a
 * wavlm_gather_rows with the returned indices).
 *   fwd: idx[n*G] = g*V + argmax (training: of (logits + gumbel)/tau, else of the raw logits); ysoft[n*G, V] =
 *        softmax((logits + gumbel)/tau) (training only); noise = the Gumbel draws [n*G, V] (fp32) or NULL for a device
 *        counter hash of `seed`; part[wavlm_gumbel_vq_partial_rows(n)][2*G*V] = per-wave column sums of softmax(raw
 *        logits) and of the hard one-hot (sum the rows with wavlm_colsum).
 *   perplexity: sums[2][G*V] -> out[0] = prob_perplexity, out[1] = code_perplexity, dA[G*V] = d prob_ppl / d avg_probs.
 *   bwd: dlogits = ysoft*(dret - <ysoft,dret>)/tau (if ysoft/dret given) + dppl[0] * softmax(raw)*(dA - <softmax,dA>)/n
 *        (if dppl given).  G <= 4, V <= 320.
 * ------------------------------------------------------------------------------------------ */
uint64_t wavlm_gumbel_vq_partial_rows(int64_t n);
int wavlm_gumbel_vq_fwd(const void* logits, int32_t dtype, const float* noise, uint64_t seed, float tau, int32_t training,
                        int64_t n, int32_t G, int32_t V, float* ysoft, int32_t* idx, float* part, void* stream);
int wavlm_vq_perplexity(const float* sums, int64_t n, int32_t G, int32_t V, float* out, float* dA, void* stream);
int wavlm_gumbel_vq_bwd(const void* logits, int32_t dtype, const float* ysoft, const float* dret, const float* dA,
                        const float* dppl, float tau, int64_t n, int32_t G, int32_t V, void* dlogits, void* stream);

/* ------------------------------------------------------------------------------------------
 * One transformer encoder block per call (WavLM/WavLM.py:615-742 TransformerSentenceEncoderLayer.forward + the
 * MultiheadAttention it owns, WavLM/modules.py:417-563; fairseq twin models/unispeech_sat/unispeech_sat.py:1040-1139).
 * The kernel sequence of a block is fixed, so the host issues it from C: one call (and one autograd node) per block and
 * direction instead of ~35 -- the reference's training loop calls the model once per micro-batch (trainer.py:697-760)
 * and the launch thread must stay ahead of a 35 ms step.
 *   forward, post-LN (pre_ln = 0):  a = out_proj(attn(x)); x1 = LN1(x + drop(a)); f = fc2(gelu(fc1 x1)); y = LN2(x1 + drop(f))
 *   forward, pre-LN (pre_ln = 1), residual adds fused into the LayerNorms: s1 = x + drop(r_in) (r_in = the previous
 *     block's feed-forward output, NULL for the first block: s1 = x); h1 = LN1(s1); a = out_proj(attn(h1));
 *     y = s1 + drop(a); h2 = LN2(y); r_out = fc2(gelu(fc1 h2))  -- the caller adds r_out in the next block / final LN.
 * attn = gate (gru_rel_pos, optional) -> packed q|k|v projection -> wavlm_attn_fused_fwd.  bf16, head_dim 64 only.
 * `saved` holds what backward needs (wavlm_layer_saved_bytes; forward with saved == NULL = inference: nothing kept, the
 * workspace must then hold wavlm_layer_fwd_workspace_bytes, which includes that region).  Backward accumulates EVERY
 * parameter gradient into the given gradient pointers ((+)=, the flat gradient arena of the optimizer), returns the input
 * gradient(s) and (+)= or = the gradient of the relative-position table.
 * All parameters share param_dtype; activations are bf16.  Dropout seeds: one per mask, regenerated in backward.
 * ------------------------------------------------------------------------------------------ */
typedef struct wavlm_layer_desc {
  int32_t B, T, D, H, F;           /* batch, frames, model width, heads (D / H == 64), feed-forward width */
  int32_t pre_ln;                  /* 0 post-LN (Base), 1 pre-LN with fused residual adds (Large) */
  int32_t param_dtype;             /* 0 f32 | 1 bf16 */
  int32_t dtab_accumulate;         /* backward: dtab (+)= instead of = */
  float eps1, eps2, scale;         /* LayerNorm epsilons, q scaling (head_dim^-0.5) */
  float p_drop, p_attn;            /* residual dropout, attention dropout (0 in eval) */
  int32_t attn_store_p;            /* 2: the attention keeps its dropout decisions (bit words) in `saved` for backward;
                                      1: its probabilities (wavlm_attn_fused_fwd_p); 0: backward recomputes everything
                                      (the default).  Same value in forward and backward. */
  uint64_t seed_r1, seed_r2, seed_attn;
  /* parameters (and, backward only, their gradient accumulators) */
  const void *Wqkv, *bqkv, *Wo, *bo, *W1, *b1, *W2, *b2, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
  const void *Wgate, *bgate, *grep_a;           /* grep_linear [8, 64], [8]; grep_a [H]; all NULL: gate == 1 */
  void *dWqkv, *dbqkv, *dWo, *dbo, *dW1, *db1, *dW2, *db2, *dln1_g, *dln1_b, *dln2_g, *dln2_b, *dWgate, *dbgate, *dgrep_a;
  void* db2_prev;                  /* pre-LN backward: gradient of the bias of the linear that produced r_in (or NULL) */
  const float* tab;                /* [H, 2T-1] Toeplitz generator of the position bias, or NULL */
  const uint8_t* kpm;              /* [B, T] key padding (1 = padded) or NULL */
  /* activations */
  const void* x;                   /* [B, T, D] block input (pre-LN: the residual stream) */
  const void* r_in;                /* pre-LN: [B, T, D] pending feed-forward output of the previous block, or NULL */
  void* y;                         /* [B, T, D] block output (pre-LN: the residual stream after the attention branch) */
  void* r_out;                     /* pre-LN: [B, T, D] this block's feed-forward output */
  void* saved; uint64_t saved_bytes;
  void* workspace; uint64_t ws_bytes;
  /* backward */
  const void* dy;                  /* gradient of y */
  const void* dr_out;              /* pre-LN: gradient of r_out */
  void* dx;                        /* gradient of x */
  void* dr_in;                     /* pre-LN: gradient of r_in (if r_in != NULL) */
  float* dtab;                     /* [H, 2T-1] (if tab != NULL) */
} wavlm_layer_desc;
uint64_t wavlm_layer_saved_bytes(const wavlm_layer_desc* d);
uint64_t wavlm_layer_fwd_workspace_bytes(const wavlm_layer_desc* d);   /* for saved == NULL add wavlm_layer_saved_bytes */
uint64_t wavlm_layer_bwd_workspace_bytes(const wavlm_layer_desc* d);
int wavlm_encoder_layer_fwd(const wavlm_layer_desc* d, void* stream);
int wavlm_encoder_layer_bwd(const wavlm_layer_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * Data-parallel seam below Python.  The reference reduces gradients AFTER the whole backward, from one flat buffer
 * (src/fairseq/distributed/legacy_distributed_data_parallel.py:76-165); here backward kernels accumulate straight into
 * the caller's flat gradient arena, and a reducer that wants to start a bucket's all-reduce while backward is still running
 * has to learn which slice has just been written.  Every entry point that ACCUMULATES into a caller-designated parameter
 * gradient -- wavlm_gemm / wavlm_gemm_grouped with `accumulate` or an accumulating `colsum`, wavlm_layernorm_bwd with
 * accumulate_params (dgamma, dbeta, dr_colsum), wavlm_colsum with accumulate, wavlm_gate_bwd (bit 0), wavlm_attn_fused_bwd
 * (dbias), wavlm_conv_wgrad_scatter with accumulate, and through them wavlm_encoder_layer_bwd -- calls the listener with
 * (base, bytes, stream, user) right after it has ENQUEUED the kernels: the slice is complete in `stream` order (record an
 * event on `stream`, make the communication stream wait for it).  One call per accumulation: a parameter used twice in
 * one forward is reported twice.  The callback runs on the launch thread and must not block.  cb == NULL: none.
 * ------------------------------------------------------------------------------------------ */
typedef void (*wavlm_grad_listener)(const void* base, uint64_t bytes, void* stream, void* user);
void wavlm_dp_set_listener(wavlm_grad_listener cb, void* user);

/* The transport half (ABI v20, csrc/dp_rccl.hip): one RCCL communicator per process (one process per GPU), the reference's
 * all-reduce of the flat gradient buffer (src/fairseq/distributed/legacy_distributed_data_parallel.py:82-120,
 * src/fairseq/distributed/utils.py:273-288) bucket by bucket, overlapped with backward:
 *   wavlm_dp_unique_id(id128)             rank 0: a fresh ncclUniqueId (128 bytes) for the caller to hand to the other ranks
 *                                         (over whatever it bootstraps with: MPI, a TCP store, torch.distributed)
 *   wavlm_dp_init(rank, world, id128, average)   ncclCommInitRank on the CURRENT device + a high-priority communication
 *                                         stream; average != 0: ncclAvg (the reference divides by the world size), else ncclSum
 *   wavlm_dp_bucket_ready(base, count, dtype, compute_stream)   the `count` elements at `base` (WL_F32 / WL_BF16) are
 *                                         complete in compute_stream order: all-reduced in place on the communication stream
 *                                         behind an event on compute_stream.  Every rank reports the same buckets in the same
 *                                         order (RCCL matches collectives by issue order).  Call it from the gradient listener.
 *   wavlm_dp_finish(compute_stream)       compute_stream waits for every bucket reported since the last finish (before the
 *                                         optimizer reads the gradients); no host synchronisation
 *   wavlm_dp_destroy()
 * RCCL is resolved at the first of these calls (the copy the process already carries, else librccl.so): libwavlm_hip.so has no
 * link-time dependency on it.  WL_ELAUNCH (-2): RCCL missing or a HIP / RCCL call failed; WL_EINVAL (-1): bad argument, not
 * initialised, initialised twice. */
int wavlm_dp_unique_id(void* id128);
int wavlm_dp_init(int32_t rank, int32_t world, const void* id128, int32_t average);
int wavlm_dp_bucket_ready(void* base, uint64_t count, int32_t dtype, void* compute_stream);
int wavlm_dp_finish(void* compute_stream);
int wavlm_dp_destroy(void);

/* ------------------------------------------------------------------------------------------
 * Measurement aid (bench.py roofline leg): HIP events around every wavlm_gemm launch while enabled.
 * ------------------------------------------------------------------------------------------ */
void wavlm_prof_enable(int on);
/* 0: automatic tile choice, 1: force the 128x128 tile (A/B measurements only) */
void wavlm_gemm_set_variant(int v);
/* Data-parallel runs: leave `n` of the 256 CUs out of every PERSISTENT GEMM launch (grid 256 - n), so that the RCCL
 * kernels of the gradient all-reduce (side stream; replaces the blocking all-reduce after backward of
 * src/fairseq/distributed/legacy_distributed_data_parallel.py:132-165) find free CUs while backward is still running.
 * 0 (default) = use the whole chip.  Values are clamped to [0, 64]. */
void wavlm_set_reserved_cus(int n);
int wavlm_get_reserved_cus(void);
int wavlm_prof_collect(int dtype, double* total_ms, double* total_flops);
/* algorithmic HBM bytes of the recorded launches: every operand, output and epilogue tensor counted once */
double wavlm_prof_collect_bytes(int dtype);
/* Kernel classes timed besides wavlm_gemm (class 0) while profiling is enabled: the fused attention forward / backward
 * (backward = dQ + dK/dV kernels + their finishing launches), the conv0 stage forward / backward (either norm mode) and
 * the LayerNorm row kernels.  Returns the number of recorded calls of class `cls`; totals: duration [ms], algorithmic
 * FLOPs (attention: 4 B H T^2 hd forward, 10 B H T^2 hd backward) and algorithmic HBM bytes (every input / output tensor
 * of the call once).  Blocks until the recorded launches have finished. */
#define WL_PROF_GEMM 0
#define WL_PROF_ATTN_FWD 1
#define WL_PROF_ATTN_BWD 2
#define WL_PROF_CONV0_FWD 3
#define WL_PROF_CONV0_BWD 4
#define WL_PROF_LN_FWD 5
#define WL_PROF_LN_BWD 6
int wavlm_prof_collect_class(int cls, double* total_ms, double* total_flops, double* total_bytes);
/* one text line per recorded wavlm_gemm launch into `path`: M N K KB trans epi split batches ms gflop (tools/gemm_step_table.py:
 * which shapes of a real step sit where against the MFMA peak).  Returns the number of lines, < 0 on error. */
int wavlm_prof_dump(const char* path);

#ifdef __cplusplus
}
#endif
#endif
