// Occupancy-masked sampling with in-kernel compaction.
// Replaces, for the density-grid branch of render_rays (reference src/renderer.py:303-343), the
// chain  sample -> pts -> get_active_mask -> boolean-index gather -> model -> zeros + masked scatter:
// one kernel generates the depths, forms the points, looks them up in the occupancy bitfield and
// appends the ACTIVE samples' (pts, dirs) to compact arrays; `slot_of_sample` records where each
// sample went (-1 = skipped).  The compositing kernels take that map and treat skipped samples as
// sigma = 0, rgb = 0 (exactly what the reference's zero-filled scatter produces), so no dense
// [R*S, 3] tensors are materialised.  Slots are reserved with ONE returning atomic per 1024-thread
// workgroup and pass of 4096 samples (wave ballots -> LDS -> one wave scans the 64 counts): same-address
// atomics serialise in L2 (one per wave: ~16 k per 2 M samples, most of the kernel's time in round 1;
// one per 1024 samples: still 2048 of them, about half of the 38 us the kernel took).  Slot order is
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
constexpr int kCompactPer = 4;                          // samples per thread and pass: 4096 samples share one atomic
constexpr int kCompactWaves = kCompactThreads / 64;
static_assert(kCompactPer * kCompactWaves == 64, "the (k, wave) counts of a pass are scanned by one wave");
__global__ void __launch_bounds__(kCompactThreads)
sample_compact_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ u,
                      int64_t n_rays, int S, float near_p, float far_p, float step,
                      const uint8_t* __restrict__ grid, int res, float bound, float scale,
                      float* __restrict__ z_out, int* __restrict__ slot_of_sample,
                      float* __restrict__ pts_c, float* __restrict__ dirs_c, unsigned* __restrict__ count,
                      uint64_t key, uint64_t counter, int draw, uint64_t first_sample) {
  const int64_t total = n_rays * (int64_t)S;
  const int64_t span = (int64_t)kCompactThreads * kCompactPer;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool small = total < (int64_t)1 << 31;
  __shared__ unsigned seg[64];                          // per (k, wave): active count, then first slot
  // uniform trip count so that every lane takes part in the ballots
  for (int64_t g0 = blockIdx.x * span; g0 < total; g0 += (int64_t)gridDim.x * span) {
    float px[kCompactPer], py[kCompactPer], pz[kCompactPer];
    unsigned long long ballot[kCompactPer];
#pragma unroll
    for (int k = 0; k < kCompactPer; ++k) {
      const int64_t g = g0 + (int64_t)k * kCompactThreads + threadIdx.x;
      bool active = false;
      px[k] = py[k] = pz[k] = 0.f;
      if (g < total) {
        const int64_t r = small ? (int64_t)((unsigned)g / (unsigned)S) : g / S;     // 64-bit division is emulated: ~100 instructions
        const int s = (int)(g - r * S);
        float z = depth_plain(s, S, step, near_p, far_p);
        if (u != nullptr || draw) {
          // draw: the jitter comes from the counter-based generator (common.h) instead of a [R,S] tensor of uniforms
          const float uu = u != nullptr ? u[g] : squares_uniform(counter, first_sample + (uint64_t)g, key);
          float lo = z, hi = z;
          if (s > 0) lo = mul_rn(0.5f, add_rn(z, depth_plain(s - 1, S, step, near_p, far_p)));
          if (s < S - 1) hi = mul_rn(0.5f, add_rn(depth_plain(s + 1, S, step, near_p, far_p), z));
          z = add_rn(lo, mul_rn(sub_rn(hi, lo), uu));
        }
        z_out[g] = z;
        px[k] = add_rn(rays_o[r * 3 + 0], mul_rn(rays_d[r * 3 + 0], z));
        py[k] = add_rn(rays_o[r * 3 + 1], mul_rn(rays_d[r * 3 + 1], z));
        pz[k] = add_rn(rays_o[r * 3 + 2], mul_rn(rays_d[r * 3 + 2], z));
        // .long() truncates toward zero: (-1, res) is exactly the range of values whose index lands in [0, res)
        const float fx = mul_rn(add_rn(px[k], bound), scale), fy = mul_rn(add_rn(py[k], bound), scale),
                    fz = mul_rn(add_rn(pz[k], bound), scale), fres = (float)res;
        if (fx > -1.0f && fx < fres && fy > -1.0f && fy < fres && fz > -1.0f && fz < fres)
          active = grid[((int64_t)((int)fx * res + (int)fy)) * res + (int)fz] != 0;
      }
      ballot[k] = __ballot(active);
      if (lane == 0) seg[k * kCompactWaves + wave] = (unsigned)__popcll(ballot[k]);
    }
    __syncthreads();
    if (wave == 0) {                                    // exclusive scan of the 64 counts; ONE returning atomic per pass
      const unsigned mine = seg[lane];
      unsigned incl = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
      }
      unsigned base = 0;
      if (lane == 63 && incl != 0) base = atomicAdd(count, incl);
      base = __shfl(base, 63);
      seg[lane] = base + incl - mine;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kCompactPer; ++k) {
      const int64_t g = g0 + (int64_t)k * kCompactThreads + threadIdx.x;
      if (g < total) {
        int slot = -1;
        if ((ballot[k] >> lane) & 1ull) {
          slot = (int)(seg[k * kCompactWaves + wave] + (unsigned)__popcll(ballot[k] & ((1ull << lane) - 1ull)));
          const int64_t r = small ? (int64_t)((unsigned)g / (unsigned)S) : g / S;
          const float dx = rays_d[r * 3 + 0], dy = rays_d[r * 3 + 1], dz = rays_d[r * 3 + 2];
          const float nrm = sqrtf(add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz)));
          pts_c[(size_t)slot * 3 + 0] = px[k]; pts_c[(size_t)slot * 3 + 1] = py[k]; pts_c[(size_t)slot * 3 + 2] = pz[k];
          dirs_c[(size_t)slot * 3 + 0] = dx / nrm; dirs_c[(size_t)slot * 3 + 1] = dy / nrm; dirs_c[(size_t)slot * 3 + 2] = dz / nrm;
        }
        slot_of_sample[g] = slot;
      }
    }
    __syncthreads();                                    // seg is rewritten by the next pass
  }
}

}  // namespace nerf

using namespace nerf;

static int sample_compact_impl(const float* rays_o, const float* rays_d, const float* u, int64_t n_rays,
                               int n_samples, float near_plane, float far_plane, const uint8_t* binary_grid,
                               int resolution, float bound, float* z_out, int* slot_of_sample, float* pts_compact,
                               float* dirs_compact, unsigned* active_count, nerf_stream_t stream,
                               uint64_t key, uint64_t counter, int draw, uint64_t first_sample = 0) {
  NERF_REQUIRE(n_rays >= 0 && n_samples >= 2 && resolution > 0 && resolution <= 32768 && bound > 0.0f, "nerf_sample_compact: bad sizes");
  NERF_REQUIRE(active_count != nullptr, "nerf_sample_compact: active_count is NULL");
  if (hipMemsetAsync(active_count, 0, sizeof(unsigned), as_stream(stream)) != hipSuccess)
    return fail(NERF_ELAUNCH, "nerf_sample_compact: memset failed");
  if (n_rays == 0) return NERF_OK;
  NERF_REQUIRE(rays_o && rays_d && binary_grid && z_out && slot_of_sample && pts_compact && dirs_compact,
               "nerf_sample_compact: NULL pointer");
  const float step = 1.0f / (float)(n_samples - 1);
  const float scale = (float)((double)resolution / (2.0 * (double)bound));
  const int64_t span = (int64_t)kCompactThreads * kCompactPer;
  int64_t blocks = (n_rays * n_samples + span - 1) / span;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(sample_compact_kernel, dim3((int)blocks), dim3(kCompactThreads), 0, as_stream(stream), rays_o, rays_d, u, n_rays,
                     n_samples, near_plane, far_plane, step, binary_grid, resolution, bound, scale, z_out, slot_of_sample,
                     pts_compact, dirs_compact, active_count, key, counter, draw, first_sample);
  return check_launch("nerf_sample_compact");
}

extern "C" int nerf_sample_compact(const float* rays_o, const float* rays_d, const float* u, int64_t n_rays,
                                   int n_samples, float near_plane, float far_plane, const uint8_t* binary_grid,
                                   int resolution, float bound, float* z_out, int* slot_of_sample, float* pts_compact,
                                   float* dirs_compact, unsigned* active_count, nerf_stream_t stream) {
  return sample_compact_impl(rays_o, rays_d, u, n_rays, n_samples, near_plane, far_plane, binary_grid, resolution, bound, z_out,
                             slot_of_sample, pts_compact, dirs_compact, active_count, stream, 0, 0, 0);
}

extern "C" int nerf_sample_compact_jitter_shard(const float* rays_o, const float* rays_d, uint64_t seed, uint64_t counter,
                                                int64_t first_ray, int64_t n_rays, int n_samples, float near_plane, float far_plane,
                                                const uint8_t* binary_grid, int resolution, float bound, float* z_out,
                                                int* slot_of_sample, float* pts_compact, float* dirs_compact,
                                                unsigned* active_count, nerf_stream_t stream) {
  NERF_REQUIRE(first_ray >= 0 && counter < ((uint64_t)1 << 24) && (first_ray + n_rays) * (int64_t)n_samples < ((int64_t)1 << 40),
               "nerf_sample_compact_jitter: counter / batch out of range");
  return sample_compact_impl(rays_o, rays_d, nullptr, n_rays, n_samples, near_plane, far_plane, binary_grid, resolution, bound,
                             z_out, slot_of_sample, pts_compact, dirs_compact, active_count, stream, squares_key(seed), counter, 1,
                             (uint64_t)first_ray * (uint64_t)n_samples);
}

extern "C" int nerf_sample_compact_jitter(const float* rays_o, const float* rays_d, uint64_t seed, uint64_t counter,
                                          int64_t n_rays, int n_samples, float near_plane, float far_plane,
                                          const uint8_t* binary_grid, int resolution, float bound, float* z_out,
                                          int* slot_of_sample, float* pts_compact, float* dirs_compact,
                                          unsigned* active_count, nerf_stream_t stream) {
  return nerf_sample_compact_jitter_shard(rays_o, rays_d, seed, counter, 0, n_rays, n_samples, near_plane, far_plane, binary_grid,
                                          resolution, bound, z_out, slot_of_sample, pts_compact, dirs_compact, active_count, stream);
}
