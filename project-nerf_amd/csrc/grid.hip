// Occupancy-grid refresh helpers (reference src/renderer.py:35-132):
// lattice generation and the overwrite / running-max + threshold + active count pass.
#include "common.h"

namespace nerf {

// node i of torch.linspace(-b, b, res): walk up from the start in the lower half, down from the
// end in the upper half (one rounding each)
__device__ __forceinline__ float lattice_node(int i, int res, float bound, float step) {
  return i < res / 2 ? __builtin_fmaf(step, (float)i, -bound) : __builtin_fmaf(-step, (float)(res - 1 - i), bound);
}

__global__ void __launch_bounds__(256)
lattice_kernel(float bound, int res, float step, float* __restrict__ out) {
  const int64_t total = (int64_t)res * res * res;
  for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int iz = (int)(g % res), iy = (int)((g / res) % res), ix = (int)(g / ((int64_t)res * res));
    out[g * 3 + 0] = lattice_node(ix, res, bound, step);   // 'ij' meshgrid, x slowest (renderer.py:53-54)
    out[g * 3 + 1] = lattice_node(iy, res, bound, step);
    out[g * 3 + 2] = lattice_node(iz, res, bound, step);
  }
}

__global__ void __launch_bounds__(256)
grid_update_kernel(const float* __restrict__ cur, float* __restrict__ grid, uint8_t* __restrict__ binary,
                   int64_t n, float decay, int dynamic, float threshold, unsigned long long* __restrict__ count) {
  unsigned long long local = 0;
  for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < n; g += (int64_t)gridDim.x * blockDim.x) {
    float v = cur[g];
    if (dynamic) v = fmaxf(grid[g] * decay, v);          // renderer.py:122-125
    grid[g] = v;
    const bool on = v > threshold;
    binary[g] = on ? 1 : 0;
    local += on ? 1 : 0;
  }
  // one atomic per wave
  for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(count, local);
}

}  // namespace nerf

using namespace nerf;

extern "C" int nerf_grid_lattice(float bound, int resolution, float* pts_out, nerf_stream_t stream) {
  NERF_REQUIRE(bound > 0.0f && resolution >= 2 && pts_out, "nerf_grid_lattice: bad arguments");
  const float step = (bound - (-bound)) / (float)(resolution - 1);
  hipLaunchKernelGGL(lattice_kernel, dim3(2048), dim3(256), 0, as_stream(stream), bound, resolution, step, pts_out);
  return check_launch("nerf_grid_lattice");
}

extern "C" int nerf_grid_update(const float* sigma, float* grid, uint8_t* binary_grid, int64_t n_cells,
                                float decay, int dynamic, float threshold, unsigned long long* active_count,
                                nerf_stream_t stream) {
  NERF_REQUIRE(n_cells > 0 && sigma && grid && binary_grid && active_count, "nerf_grid_update: bad arguments");
  if (hipMemsetAsync(active_count, 0, sizeof(unsigned long long), as_stream(stream)) != hipSuccess)
    return fail(NERF_ELAUNCH, "nerf_grid_update: memset failed");
  int64_t blocks = (n_cells + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(grid_update_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), sigma, grid, binary_grid,
                     n_cells, decay, dynamic, threshold, active_count);
  return check_launch("nerf_grid_update");
}
