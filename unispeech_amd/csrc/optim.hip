// Fused optimizer step over flat arenas.
//
// The reference performs per-tensor: bf16 grad -> fp32 copy, global L2 norm, clip multiply, deferred
// multiply_grads factor, Adam moments, decoupled weight decay, parameter update, fp32 -> bf16 copy-back
// (src/fairseq/optim/fp16_optimizer.py:106-218, optim/adam.py:148-228, utils.py:338-388): about 44 B/param in
// hundreds of launches.  Here: one L2-norm reduction over the flat gradient arena (wavlm_sumsq) and ONE update
// kernel reading grad + fp32 (p, m, v), writing fp32 (p, m, v) + the low-precision parameter copy = 28 B/param.
// The clip coefficient is derived on device from the norm scalar, so the step needs no host synchronisation.
#include "common.hpp"
#include "../../include/wavlm_hip.h"

__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
    const void* __restrict__ grad, int gdt, void* __restrict__ p_lowp, int ldt, long n, float lr, float beta1, float beta2,
    float eps, float wd, float step_size, float grad_mult, const float* __restrict__ grad_mult_dev, const float* __restrict__ gnorm_sq, float max_norm) {
  float gm = grad_mult_dev ? grad_mult * grad_mult_dev[0] : grad_mult;
  if (gnorm_sq && max_norm > 0.f) {
    const float total = sqrtf(gnorm_sq[0]) * fabsf(gm);
    const float clip = fminf(1.f, max_norm / (total + 1e-6f));
    gm *= clip;
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float g = ld_elem(grad, i, gdt) * gm;
    const float mi = beta1 * m[i] + (1.f - beta1) * g;
    const float vi = beta2 * v[i] + (1.f - beta2) * g * g;
    float pi = p[i];
    if (wd != 0.f) pi += pi * (-wd * lr);
    pi += -step_size * (mi / (sqrtf(vi) + eps));
    m[i] = mi; v[i] = vi; p[i] = pi;
    if (p_lowp) st_elem(p_lowp, i, ldt, pi);
  }
}

extern "C" {

// One Adam(W-style, as fairseq's Adam) step over n contiguous elements.  `step` is 1-based.
// grad is multiplied by grad_mult (* grad_mult_dev[0] if given) and, if gnorm_sq != NULL and max_norm > 0, by
// min(1, max_norm / (sqrt(*gnorm_sq) * |grad_mult (* grad_mult_dev[0] if given)| + 1e-6)).
int wavlm_adam_step(float* p, float* m, float* v, const void* grad, int32_t grad_dtype, void* p_lowp,
                    int32_t lowp_dtype, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                    int64_t step, float grad_mult, const float* grad_mult_dev, const float* gnorm_sq, float max_norm, void* stream) {
  if (!p || !m || !v || !grad || n < 0 || step < 1) return WL_EINVAL;
  if (n == 0) return WL_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr * sqrt(bc2) / bc1);
  long grid = (n + 255) / 256; if (grid > 16384) grid = 16384;
  WL_LAUNCH(adam_step_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, m, v, grad,
                     (int)grad_dtype, p_lowp, (int)lowp_dtype, (long)n, lr, beta1, beta2, eps, weight_decay, step_size,
                     grad_mult, grad_mult_dev, gnorm_sq, max_norm);
  return wl_check_launch();
}

}  // extern "C"
