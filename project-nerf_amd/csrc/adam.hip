// Adam / AdamW over one flat fp32 parameter vector:
// 28 B/param of streaming traffic, float4-vectorised.
#include "common.h"

namespace nerf {

__global__ void __launch_bounds__(256)
adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
            float* __restrict__ v, int64_t n, float lr, float beta1, float beta2, float eps, float wd,
            float inv_bc1, float inv_sqrt_bc2, const float* __restrict__ grad_scale) {
  const float gs = grad_scale ? *grad_scale : 1.0f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float pi = p[i];
    const float gi = g[i] * gs;
    if (wd != 0.0f) pi *= (1.0f - lr * wd);
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    p[i] = pi - (lr * inv_bc1) * (mi / denom);
  }
}

}  // namespace nerf

extern "C" int nerf_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                              int64_t n, int step, float lr, float beta1, float beta2, float eps,
                              float weight_decay, const float* grad_scale_dev, nerf_stream_t stream) {
  NERF_REQUIRE(n >= 0 && step >= 1, "nerf_adam_step: n=%lld step=%d", (long long)n, step);
  if (n == 0) return NERF_OK;
  NERF_REQUIRE(params && grads && exp_avg && exp_avg_sq, "nerf_adam_step: NULL pointer");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(nerf::adam_kernel, dim3((int)blocks), dim3(256), 0, nerf::as_stream(stream), params,
                     grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay,
                     (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), grad_scale_dev);
  return nerf::check_launch("nerf_adam_step");
}
