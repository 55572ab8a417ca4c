// Stand-alone Fourier feature kernel.  fp32 in / fp32 out with
// full-range sinf/cosf: arguments reach 2^14 * pi * 1.5, far outside the range of
// the hardware v_sin_f32 approximations.  The fused decoder kernel has its own
// in-register version; this one backs FourierRepresentation.forward.
#include "common.h"

namespace nerf {

__global__ void __launch_bounds__(256)
fourier_kernel(const float* __restrict__ x, int64_t n, int dim, int n_freq, float* __restrict__ out) {
  const int width = dim + 2 * dim * n_freq;
  const int64_t total = n * (int64_t)width;
  for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < total;
       g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = g / width;
    const int col = (int)(g - row * width);
    float v;
    if (col < dim) {
      v = x[row * dim + col];
    } else {
      const int c = col - dim;
      const int band = c / (2 * dim);
      const int rem = c - band * 2 * dim;
      const int a = rem % dim;
      // (x * 2^band) * pi, both products rounded to fp32 (src/embeddings.py:30-31)
      const float arg = mul_rn(mul_rn(x[row * dim + a], (float)(1u << band)), 3.14159265358979323846f);
      v = rem < dim ? sinf(arg) : cosf(arg);
    }
    out[g] = v;
  }
}

}  // namespace nerf

extern "C" int nerf_fourier_encode(const float* x, int64_t n, int dim, int n_freq, float* out,
                                   nerf_stream_t stream) {
  NERF_REQUIRE(n >= 0 && dim >= 1 && dim <= 8 && n_freq >= 0 && n_freq <= 24,
               "nerf_fourier_encode: n=%lld dim=%d n_freq=%d", (long long)n, dim, n_freq);
  NERF_REQUIRE(n == 0 || (x && out), "nerf_fourier_encode: NULL pointer");
  if (n == 0) return NERF_OK;
  int64_t blocks = (n * (dim + 2 * dim * n_freq) + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(nerf::fourier_kernel, dim3((int)blocks), dim3(256), 0, nerf::as_stream(stream), x, n,
                     dim, n_freq, out);
  return nerf::check_launch("nerf_fourier_encode");
}
