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

// ---------------------------------------------------------------------------------------------
// Fused regulariser + clip + AdamW for flat hash-table parameters (replaces the
// torch sequence of reference run.py:611-630: TV-L1 over the flat vector, loss.backward() of it,
// clip_grad_norm_(max_norm) and AdamW.step()).  Two streaming passes:
//   pass 1  g_i += w/(n-1) * (sign(p_i - p_{i-1}) - sign(p_{i+1} - p_i));  normsq += g_i^2
//   pass 2  AdamW with g * min(1, max_norm / (sqrt(normsq) + 1e-6))
// ---------------------------------------------------------------------------------------------
namespace nerf {

__device__ __forceinline__ float sgn(float x) { return (x > 0.0f) - (x < 0.0f); }

__global__ void __launch_bounds__(256)
tv_normsq_kernel(const float* __restrict__ p, float* __restrict__ g, int64_t n, float tv_scale,
                 float* __restrict__ normsq) {
  float local = 0.0f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float gi = g[i];
    if (tv_scale != 0.0f) {
      const float pi = p[i];
      float t = 0.0f;
      if (i > 0) t += sgn(pi - p[i - 1]);          // d/dp_i |p_i - p_{i-1}|
      if (i + 1 < n) t -= sgn(p[i + 1] - pi);      // d/dp_i |p_{i+1} - p_i|
      gi += tv_scale * t;
      g[i] = gi;
    }
    local += gi * gi;
  }
  local = wave_sum(local);
  if ((threadIdx.x & 63) == 0) atomicAdd(normsq, local);
}

__global__ void __launch_bounds__(256)
adamw_clip_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                  int64_t n, float lr, float beta1, float beta2, float eps, float wd, float inv_bc1,
                  float inv_sqrt_bc2, const float* __restrict__ normsq, float max_norm, float extra_scale) {
  float gs = extra_scale;
  if (normsq != nullptr && max_norm > 0.0f) {
    const float coef = max_norm / (sqrtf(*normsq * extra_scale * extra_scale) + 1e-6f);   // torch clip_grad_norm_
    gs *= coef < 1.0f ? coef : 1.0f;
  }
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float pi = p[i];
    const float gi = g[i] * gs;
    if (wd != 0.0f) pi *= (1.0f - lr * wd);
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = pi - (lr * inv_bc1) * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
  }
}

}  // namespace nerf

extern "C" int nerf_tv_normsq(const float* params, float* grads, int64_t n, float tv_weight, float* normsq_dev,
                              nerf_stream_t stream) {
  NERF_REQUIRE(n >= 0 && normsq_dev, "nerf_tv_normsq: bad arguments");
  if (hipMemsetAsync(normsq_dev, 0, sizeof(float), nerf::as_stream(stream)) != hipSuccess)
    return nerf::fail(NERF_ELAUNCH, "nerf_tv_normsq: memset failed");
  if (n == 0) return NERF_OK;
  NERF_REQUIRE(params && grads, "nerf_tv_normsq: NULL pointer");
  const float tv_scale = n > 1 ? tv_weight / (float)(n - 1) : 0.0f;      // d/dp of mean|p[1:] - p[:-1]| * w
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(nerf::tv_normsq_kernel, dim3((int)blocks), dim3(256), 0, nerf::as_stream(stream), params, grads, n,
                     tv_scale, normsq_dev);
  return nerf::check_launch("nerf_tv_normsq");
}

extern "C" int nerf_adamw_clip_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                    int step, float lr, float beta1, float beta2, float eps, float weight_decay,
                                    const float* normsq_dev, float max_norm, float grad_scale, nerf_stream_t stream) {
  NERF_REQUIRE(n >= 0 && step >= 1, "nerf_adamw_clip_step: n=%lld step=%d", (long long)n, step);
  if (n == 0) return NERF_OK;
  NERF_REQUIRE(params && grads && exp_avg && exp_avg_sq, "nerf_adamw_clip_step: NULL pointer");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(nerf::adamw_clip_kernel, dim3((int)blocks), dim3(256), 0, nerf::as_stream(stream), params, grads,
                     exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, (float)(1.0 / bc1),
                     (float)(1.0 / sqrt(bc2)), normsq_dev, max_norm, grad_scale);
  return nerf::check_launch("nerf_adamw_clip_step");
}
