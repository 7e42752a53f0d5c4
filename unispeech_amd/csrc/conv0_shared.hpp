// Constants shared by conv0.hip and conv0_bwd_mfma.hip (the latter is a translation unit of its own: build.py compiles it
// with -fno-slp-vectorize).
#pragma once
#define C0_KW 10
#define C0_TCH 512      // time steps per block (forward passes)
#define C0_TCH_BWD 1024 // time steps per block (backward passes)
// GELU / GELU' chord tables (conv0.hip): 2048 cells over [-8, 8), cell = (slope, intercept)
#define GT_N 2048
#define GT_LO (-8.0f)
#define GT_INV_H (GT_N / 16.0f)
#define C0_NQ (1 + C0_KW)   // chunk partials of the GroupNorm-mode backward: A, P[0..9]
#define C0_NX 112           // waveform partials: Q[10], XX[10][10]
// GroupNorm-mode backward on the matrix cores (bf16 waveform / parameters / gradient, C = 512); tab1 = device address of the
// GELU' table; returns a WL_* code
int conv0_bwd_mfma_launch(const void* wav, const void* W, const void* gamma, const void* beta, const float* stats, const void* g,
                          float* part, float* partx, long T, int T0, int stride, float gscale, int nchunk, int B,
                          const float2* tab1, hipStream_t st);
// LayerNorm-mode backward on the matrix cores (same instantiation): workspace size and launch (gram constants, the pass, the
// two reduction kernels); dW / dcbias (may be null) / dgamma / dbeta are written in bf16
uint64_t conv0_ln_bwd_mfma_workspace_bytes(int B, int T0);
int conv0_ln_bwd_mfma_launch(const void* wav, const void* W, const void* cbias, const void* gamma, const void* beta, const void* g,
                             void* dW, void* dcbias, void* dgamma, void* dbeta, void* workspace, long T, int T0, int stride, int B,
                             float eps, float gscale, const float2* tab1, hipStream_t st);
// ... and the forward: gc = 128 floats of workspace (the parameters' second moments over the channels)
int conv0_ln_fwd_mfma_launch(const void* wav, const void* W, const void* cbias, const void* gamma, const void* beta, void* out,
                             float* gc, long T, int T0, int stride, int B, float eps, const float2* tab0, hipStream_t st);
// GroupNorm-mode forward (apply pass) on the matrix cores; stats[B][C][2] = (mean, rstd) from the Gram pass
int conv0_gn_fwd_mfma_launch(const void* wav, const void* W, const void* gamma, const void* beta, const float* stats, void* out,
                             long T, int T0, int stride, int B, const float2* tab0, hipStream_t st);
// ... and its Gram pass (bf16 waveform): partx[B][*nrec][112]
int conv0_gram_mfma_launch(const void* wav, float* partx, long T, int T0, int stride, int B, hipStream_t st, int* nrec);
