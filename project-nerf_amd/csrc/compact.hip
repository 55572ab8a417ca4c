// Occupancy-masked sampling with in-kernel compaction.
// Replaces, for the density-grid branch of render_rays (reference src/renderer.py:303-343), the
// chain  sample -> pts -> get_active_mask -> boolean-index gather -> model -> zeros + masked scatter:
// one kernel generates the depths, forms the points, looks them up in the occupancy bitfield and
// appends the ACTIVE samples' (pts, dirs) to compact arrays; `slot_of_sample` records where each
// sample went (-1 = skipped).  The compositing kernels take that map and treat skipped samples as
// sigma = 0, rgb = 0 (exactly what the reference's zero-filled scatter produces), so no dense
// [R*S, 3] tensors are materialised.  Slots are reserved with ONE returning atomic per 1024-thread
// workgroup and pass (wave ballots -> LDS -> prefix): one per wave meant ~16 k same-address
// atomics per 2 M samples, which serialise in L2 and were most of the kernel's time.  Slot order is
// arbitrary but consistent within a call.
#include "common.h"

namespace nerf {

__device__ __forceinline__ float lin01(int i, int n, float step) {
  if (i < n / 2) return mul_rn(step, (float)i);
  return __builtin_fmaf(-step, (float)(n - 1 - i), 1.0f);
}
__device__ __forceinline__ float depth_plain(int i, int n, float step, float near_p, float far_p) {
  const float t = lin01(i, n, step);
  return add_rn(mul_rn(near_p, sub_rn(1.0f, t)), mul_rn(far_p, t));
}

constexpr int kCompactThreads = 1024;
__global__ void __launch_bounds__(kCompactThreads)
sample_compact_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ u,
                      int64_t n_rays, int S, float near_p, float far_p, float step,
                      const uint8_t* __restrict__ grid, int res, float bound, float scale,
                      float* __restrict__ z_out, int* __restrict__ slot_of_sample,
                      float* __restrict__ pts_c, float* __restrict__ dirs_c, unsigned* __restrict__ count) {
  const int64_t total = n_rays * (int64_t)S;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // uniform trip count so that every lane takes part in the ballot
  for (int64_t g0 = blockIdx.x * (int64_t)blockDim.x; g0 < total; g0 += stride) {
    const int64_t g = g0 + threadIdx.x;
    bool active = false;
    float px = 0.f, py = 0.f, pz = 0.f, vx = 0.f, vy = 0.f, vz = 0.f;
    if (g < total) {
      const int64_t r = g / S;
      const int s = (int)(g - r * S);
      float z = depth_plain(s, S, step, near_p, far_p);
      if (u != nullptr) {
        float lo = z, hi = z;
        if (s > 0) lo = mul_rn(0.5f, add_rn(z, depth_plain(s - 1, S, step, near_p, far_p)));
        if (s < S - 1) hi = mul_rn(0.5f, add_rn(depth_plain(s + 1, S, step, near_p, far_p), z));
        z = add_rn(lo, mul_rn(sub_rn(hi, lo), u[g]));
      }
      z_out[g] = z;
      const float dx = rays_d[r * 3 + 0], dy = rays_d[r * 3 + 1], dz = rays_d[r * 3 + 2];
      px = add_rn(rays_o[r * 3 + 0], mul_rn(dx, z));
      py = add_rn(rays_o[r * 3 + 1], mul_rn(dy, z));
      pz = add_rn(rays_o[r * 3 + 2], mul_rn(dz, z));
      const int64_t ix = (int64_t)mul_rn(add_rn(px, bound), scale);
      const int64_t iy = (int64_t)mul_rn(add_rn(py, bound), scale);
      const int64_t iz = (int64_t)mul_rn(add_rn(pz, bound), scale);
      if (ix >= 0 && ix < res && iy >= 0 && iy < res && iz >= 0 && iz < res)
        active = grid[(ix * res + iy) * res + iz] != 0;
      if (active) {
        const float nrm = sqrtf(add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz)));
        vx = dx / nrm; vy = dy / nrm; vz = dz / nrm;
      }
    }
    const unsigned long long ballot = __ballot(active);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ unsigned wave_count[kCompactThreads / 64], block_base;
    if (lane == 0) wave_count[wave] = (unsigned)__popcll(ballot);
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned tot = 0;
#pragma unroll
      for (int w = 0; w < kCompactThreads / 64; ++w) tot += wave_count[w];
      block_base = tot ? atomicAdd(count, tot) : 0u;
    }
    __syncthreads();
    unsigned base = block_base;
    for (int w = 0; w < wave; ++w) base += wave_count[w];
    __syncthreads();                                   // wave_count / block_base are rewritten by the next pass
    if (g < total) {
      int slot = -1;
      if (active) {
        slot = (int)(base + (unsigned)__popcll(ballot & ((1ull << lane) - 1ull)));
        pts_c[(size_t)slot * 3 + 0] = px; pts_c[(size_t)slot * 3 + 1] = py; pts_c[(size_t)slot * 3 + 2] = pz;
        dirs_c[(size_t)slot * 3 + 0] = vx; dirs_c[(size_t)slot * 3 + 1] = vy; dirs_c[(size_t)slot * 3 + 2] = vz;
      }
      slot_of_sample[g] = slot;
    }
  }
}

}  // namespace nerf

using namespace nerf;

extern "C" int nerf_sample_compact(const float* rays_o, const float* rays_d, const float* u, int64_t n_rays,
                                   int n_samples, float near_plane, float far_plane, const uint8_t* binary_grid,
                                   int resolution, float bound, float* z_out, int* slot_of_sample, float* pts_compact,
                                   float* dirs_compact, unsigned* active_count, nerf_stream_t stream) {
  NERF_REQUIRE(n_rays >= 0 && n_samples >= 2 && resolution > 0 && bound > 0.0f, "nerf_sample_compact: bad sizes");
  NERF_REQUIRE(active_count != nullptr, "nerf_sample_compact: active_count is NULL");
  if (hipMemsetAsync(active_count, 0, sizeof(unsigned), as_stream(stream)) != hipSuccess)
    return fail(NERF_ELAUNCH, "nerf_sample_compact: memset failed");
  if (n_rays == 0) return NERF_OK;
  NERF_REQUIRE(rays_o && rays_d && binary_grid && z_out && slot_of_sample && pts_compact && dirs_compact,
               "nerf_sample_compact: NULL pointer");
  const float step = 1.0f / (float)(n_samples - 1);
  const float scale = (float)((double)resolution / (2.0 * (double)bound));
  int64_t blocks = (n_rays * n_samples + kCompactThreads - 1) / kCompactThreads;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sample_compact_kernel, dim3((int)blocks), dim3(kCompactThreads), 0, as_stream(stream), rays_o, rays_d, u, n_rays,
                     n_samples, near_plane, far_plane, step, binary_grid, resolution, bound, scale, z_out, slot_of_sample,
                     pts_compact, dirs_compact, active_count);
  return check_launch("nerf_sample_compact");
}
