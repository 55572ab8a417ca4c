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
                      uint64_t key, uint64_t counter, int draw, uint64_t first_sample, unsigned* __restrict__ zero_out,
                      unsigned* __restrict__ tickets, unsigned* __restrict__ count_host, unsigned seq) {
  if (zero_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *zero_out = 0u;      // the NEXT call's counter (chained form)
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
  // chained form: the workgroup that finishes last hands the count to the host -- two words of host-mapped pinned memory, the call's
  // sequence number written last behind a system-scope fence -- instead of a copy launch and an event behind the kernel (each of those
  // packets cost the stream a ~5-us gap of its own).  Tickets in two stages (32 groups: same-address atomics retire serially).
  if (tickets != nullptr && threadIdx.x == 0) {       // (this workgroup's atomics on *count have returned: their values were used)
    const unsigned B = gridDim.x, g = blockIdx.x % 32u, in_group = (B - g + 31u) / 32u;
    if (atomicAdd(&tickets[1 + g], 1u) == in_group - 1) {
      atomicExch(&tickets[1 + g], 0u);
      if (atomicAdd(&tickets[0], 1u) == (B < 32u ? B : 32u) - 1) {
        atomicExch(&tickets[0], 0u);
        volatile unsigned* host = count_host;
        host[0] = atomicOr(count, 0u);
        __threadfence_system();
        host[1] = seq;
      }
    }
  }
}

// ---- ordered form (option "deterministic"): slots in SAMPLE ORDER, whatever the order the workgroups ran in ----
// The single-pass kernel above reserves each pass's slots with a returning atomic: which pass draws which range of the
// compact arrays depends on scheduling, and with it the order in which every later kernel sums over samples (MFMA
// contractions over the sample index, workgroup partials).  Three launches instead: (1) depths, ray points, occupancy
// lookup -> active flag per sample (slot_of_sample: 0 / -1) and the active count of every (pass, k, wave) segment of 64
// samples; (2) one workgroup scans the segment counts in order; (3) slots = segment start + rank inside the wave's
// ballot, points recomputed from the stored depth (the same two rounded operations: the same bits).
__global__ void __launch_bounds__(kCompactThreads)
sample_compact_count_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ u,
                            int64_t n_rays, int S, float near_p, float far_p, float step, const uint8_t* __restrict__ grid, int res,
                            float bound, float scale, float* __restrict__ z_out, int* __restrict__ slot_of_sample,
                            unsigned* __restrict__ seg_count, uint64_t key, uint64_t counter, int draw, uint64_t first_sample) {
  const int64_t total = n_rays * (int64_t)S;
  const int64_t span = (int64_t)kCompactThreads * kCompactPer;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool small = total < (int64_t)1 << 31;
  for (int64_t g0 = blockIdx.x * span; g0 < total; g0 += (int64_t)gridDim.x * span) {
    const int64_t pass = g0 / span;
#pragma unroll
    for (int k = 0; k < kCompactPer; ++k) {
      const int64_t g = g0 + (int64_t)k * kCompactThreads + threadIdx.x;
      bool active = false;
      if (g < total) {
        const int64_t r = small ? (int64_t)((unsigned)g / (unsigned)S) : g / S;
        const int s = (int)(g - r * S);
        float z = depth_plain(s, S, step, near_p, far_p);
        if (u != nullptr || draw) {
          const float uu = u != nullptr ? u[g] : squares_uniform(counter, first_sample + (uint64_t)g, key);
          float lo = z, hi = z;
          if (s > 0) lo = mul_rn(0.5f, add_rn(z, depth_plain(s - 1, S, step, near_p, far_p)));
          if (s < S - 1) hi = mul_rn(0.5f, add_rn(depth_plain(s + 1, S, step, near_p, far_p), z));
          z = add_rn(lo, mul_rn(sub_rn(hi, lo), uu));
        }
        z_out[g] = z;
        const float px = add_rn(rays_o[r * 3 + 0], mul_rn(rays_d[r * 3 + 0], z));
        const float py = add_rn(rays_o[r * 3 + 1], mul_rn(rays_d[r * 3 + 1], z));
        const float pz = add_rn(rays_o[r * 3 + 2], mul_rn(rays_d[r * 3 + 2], z));
        const float fx = mul_rn(add_rn(px, bound), scale), fy = mul_rn(add_rn(py, bound), scale),
                    fz = mul_rn(add_rn(pz, bound), scale), fres = (float)res;
        if (fx > -1.0f && fx < fres && fy > -1.0f && fy < fres && fz > -1.0f && fz < fres)
          active = grid[((int64_t)((int)fx * res + (int)fy)) * res + (int)fz] != 0;
        slot_of_sample[g] = active ? 0 : -1;
      }
      const unsigned long long ballot = __ballot(active);
      if (lane == 0) seg_count[pass * 64 + k * kCompactWaves + wave] = (unsigned)__popcll(ballot);
    }
  }
}

// exclusive scan of n_seg counts in place (one workgroup of 1024; a carried total between rounds); *count = the total
__global__ void __launch_bounds__(1024) sample_compact_scan_kernel(unsigned* __restrict__ seg, int64_t n_seg, unsigned* __restrict__ count) {
  __shared__ unsigned wave_tot[16];
  __shared__ unsigned carry;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < n_seg; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const unsigned mine = i < n_seg ? seg[i] : 0u;
    unsigned incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned v = __shfl_up(incl, off);
      if (lane >= off) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned before = carry;
    for (int w = 0; w < wave; ++w) before += wave_tot[w];
    if (i < n_seg) seg[i] = before + incl - mine;
    __syncthreads();
    if (threadIdx.x == 1023) carry = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *count = carry;
}

__global__ void __launch_bounds__(kCompactThreads)
sample_compact_place_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, int64_t n_rays, int S,
                            const float* __restrict__ z, int* __restrict__ slot_of_sample, const unsigned* __restrict__ seg_start,
                            float* __restrict__ pts_c, float* __restrict__ dirs_c) {
  const int64_t total = n_rays * (int64_t)S;
  const int64_t span = (int64_t)kCompactThreads * kCompactPer;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool small = total < (int64_t)1 << 31;
  for (int64_t g0 = blockIdx.x * span; g0 < total; g0 += (int64_t)gridDim.x * span) {
    const int64_t pass = g0 / span;
#pragma unroll
    for (int k = 0; k < kCompactPer; ++k) {
      const int64_t g = g0 + (int64_t)k * kCompactThreads + threadIdx.x;
      const bool active = g < total && slot_of_sample[g] == 0;
      const unsigned long long ballot = __ballot(active);
      if (!active) continue;
      const int slot = (int)(seg_start[pass * 64 + k * kCompactWaves + wave] + (unsigned)__popcll(ballot & ((1ull << lane) - 1ull)));
      const int64_t r = small ? (int64_t)((unsigned)g / (unsigned)S) : g / S;
      const float zz = z[g];
      const float dx = rays_d[r * 3 + 0], dy = rays_d[r * 3 + 1], dz = rays_d[r * 3 + 2];
      const float nrm = sqrtf(add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz)));
      pts_c[(size_t)slot * 3 + 0] = add_rn(rays_o[r * 3 + 0], mul_rn(dx, zz));
      pts_c[(size_t)slot * 3 + 1] = add_rn(rays_o[r * 3 + 1], mul_rn(dy, zz));
      pts_c[(size_t)slot * 3 + 2] = add_rn(rays_o[r * 3 + 2], mul_rn(dz, zz));
      dirs_c[(size_t)slot * 3 + 0] = dx / nrm; dirs_c[(size_t)slot * 3 + 1] = dy / nrm; dirs_c[(size_t)slot * 3 + 2] = dz / nrm;
      slot_of_sample[g] = slot;
    }
  }
}

}  // namespace nerf

using namespace nerf;

static int sample_compact_impl(const float* rays_o, const float* rays_d, const float* u, int64_t n_rays,
                               int n_samples, float near_plane, float far_plane, const uint8_t* binary_grid,
                               int resolution, float bound, float* z_out, int* slot_of_sample, float* pts_compact,
                               float* dirs_compact, unsigned* active_count, nerf_stream_t stream,
                               uint64_t key, uint64_t counter, int draw, uint64_t first_sample = 0, void* scratch = nullptr,
                               size_t scratch_bytes = 0, unsigned* next_count = nullptr, unsigned* tickets = nullptr,
                               unsigned* count_host = nullptr, unsigned seq = 0) {
  NERF_REQUIRE(n_rays >= 0 && n_samples >= 2 && resolution > 0 && resolution <= 32768 && bound > 0.0f, "nerf_sample_compact: bad sizes");
  NERF_REQUIRE(active_count != nullptr, "nerf_sample_compact: active_count is NULL");
  // chained form (next_count): active_count was cleared by the previous call of the chain (or by the caller), this call's kernel
  // clears *next_count -- no fill launch per call
  if (next_count == nullptr && hipMemsetAsync(active_count, 0, sizeof(unsigned), as_stream(stream)) != hipSuccess)
    return fail(NERF_ELAUNCH, "nerf_sample_compact: memset failed");
  if (n_rays == 0 && tickets == nullptr) {
    if (next_count != nullptr && hipMemsetAsync(next_count, 0, sizeof(unsigned), as_stream(stream)) != hipSuccess)
      return fail(NERF_ELAUNCH, "nerf_sample_compact: memset failed");
    return NERF_OK;
  }                                                    // (publishing form with no rays: one workgroup that publishes a count of zero)
  NERF_REQUIRE(n_rays == 0 || (rays_o && rays_d && binary_grid && z_out && slot_of_sample && pts_compact && dirs_compact),
               "nerf_sample_compact: NULL pointer");
  const float step = 1.0f / (float)(n_samples - 1);
  const float scale = (float)((double)resolution / (2.0 * (double)bound));
  const int64_t span = (int64_t)kCompactThreads * kCompactPer;
  int64_t blocks = (n_rays * n_samples + span - 1) / span;
  const int64_t n_seg = blocks * 64;                     // before the cap: one (k, wave) segment per 64 samples of every pass
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  if (scratch != nullptr) {                              // ordered form: slots in sample order
    NERF_REQUIRE(scratch_bytes >= (size_t)n_seg * sizeof(unsigned) && ((uintptr_t)scratch & 3) == 0,
                 "nerf_sample_compact_ordered: scratch of %zu bytes, %zu needed", scratch_bytes, (size_t)n_seg * sizeof(unsigned));
    unsigned* seg = static_cast<unsigned*>(scratch);
    hipLaunchKernelGGL(sample_compact_count_kernel, dim3((int)blocks), dim3(kCompactThreads), 0, as_stream(stream), rays_o, rays_d, u, n_rays,
                       n_samples, near_plane, far_plane, step, binary_grid, resolution, bound, scale, z_out, slot_of_sample, seg, key, counter,
                       draw, first_sample);
    hipLaunchKernelGGL(sample_compact_scan_kernel, dim3(1), dim3(1024), 0, as_stream(stream), seg, n_seg, active_count);
    hipLaunchKernelGGL(sample_compact_place_kernel, dim3((int)blocks), dim3(kCompactThreads), 0, as_stream(stream), rays_o, rays_d, n_rays,
                       n_samples, z_out, slot_of_sample, seg, pts_compact, dirs_compact);
    return check_launch("nerf_sample_compact_ordered");
  }
  hipLaunchKernelGGL(sample_compact_kernel, dim3((int)blocks), dim3(kCompactThreads), 0, as_stream(stream), rays_o, rays_d, u, n_rays,
                     n_samples, near_plane, far_plane, step, binary_grid, resolution, bound, scale, z_out, slot_of_sample,
                     pts_compact, dirs_compact, active_count, key, counter, draw, first_sample, next_count, tickets, count_host, seq);
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

// The same call as a link of a CHAIN of calls on one stream.  chain_state: NERF_COMPACT_CHAIN_WORDS device words, zeroed ONCE by the
// caller: two counters used alternately (this call counts in word `turn` and its kernel clears word `turn ^ 1` for the next link: no
// fill launch per call) and the tickets of the workgroup that finishes last, which writes the count to count_host[0] and then
// `seq` to count_host[1] (host-mapped pinned memory; NULL: no publication): no copy launch, no event -- the host polls for its seq.
extern "C" int nerf_sample_compact_jitter_chain(const float* rays_o, const float* rays_d, uint64_t seed, uint64_t counter,
                                                int64_t first_ray, int64_t n_rays, int n_samples, float near_plane, float far_plane,
                                                const uint8_t* binary_grid, int resolution, float bound, float* z_out,
                                                int* slot_of_sample, float* pts_compact, float* dirs_compact,
                                                unsigned* chain_state, int turn, unsigned* count_host, unsigned seq, nerf_stream_t stream) {
  NERF_REQUIRE(first_ray >= 0 && counter < ((uint64_t)1 << 24) && (first_ray + n_rays) * (int64_t)n_samples < ((int64_t)1 << 40),
               "nerf_sample_compact_jitter_chain: counter / batch out of range");
  NERF_REQUIRE(chain_state != nullptr && (turn == 0 || turn == 1), "nerf_sample_compact_jitter_chain: chain_state NULL or turn not 0 / 1");
  return sample_compact_impl(rays_o, rays_d, nullptr, n_rays, n_samples, near_plane, far_plane, binary_grid, resolution, bound,
                             z_out, slot_of_sample, pts_compact, dirs_compact, chain_state + turn, stream, squares_key(seed), counter, 1,
                             (uint64_t)first_ray * (uint64_t)n_samples, nullptr, 0, chain_state + (turn ^ 1),
                             count_host != nullptr ? chain_state + 2 : nullptr, count_host, seq);
}

extern "C" int nerf_sample_compact_jitter(const float* rays_o, const float* rays_d, uint64_t seed, uint64_t counter,
                                          int64_t n_rays, int n_samples, float near_plane, float far_plane,
                                          const uint8_t* binary_grid, int resolution, float bound, float* z_out,
                                          int* slot_of_sample, float* pts_compact, float* dirs_compact,
                                          unsigned* active_count, nerf_stream_t stream) {
  return nerf_sample_compact_jitter_shard(rays_o, rays_d, seed, counter, 0, n_rays, n_samples, near_plane, far_plane, binary_grid,
                                          resolution, bound, z_out, slot_of_sample, pts_compact, dirs_compact, active_count, stream);
}

extern "C" size_t nerf_sample_compact_ordered_scratch_bytes(int64_t n_rays, int n_samples) {
  if (n_rays <= 0 || n_samples <= 0) return 0;
  const int64_t span = (int64_t)kCompactThreads * kCompactPer;
  return (size_t)((n_rays * n_samples + span - 1) / span) * 64 * sizeof(unsigned);
}

extern "C" int nerf_sample_compact_ordered(const float* rays_o, const float* rays_d, const float* u, int draw, uint64_t seed, uint64_t counter,
                                           int64_t first_ray, int64_t n_rays, int n_samples, float near_plane, float far_plane,
                                           const uint8_t* binary_grid, int resolution, float bound, float* z_out, int* slot_of_sample,
                                           float* pts_compact, float* dirs_compact, unsigned* active_count, void* scratch,
                                           size_t scratch_bytes, nerf_stream_t stream) {
  NERF_REQUIRE(first_ray >= 0 && counter < ((uint64_t)1 << 24) && (first_ray + n_rays) * (int64_t)n_samples < ((int64_t)1 << 40),
               "nerf_sample_compact_ordered: counter / batch out of range");
  NERF_REQUIRE(n_rays == 0 || scratch != nullptr, "nerf_sample_compact_ordered: scratch is NULL");
  return sample_compact_impl(rays_o, rays_d, u, n_rays, n_samples, near_plane, far_plane, binary_grid, resolution, bound, z_out,
                             slot_of_sample, pts_compact, dirs_compact, active_count, stream, squares_key(seed), counter,
                             (u == nullptr && draw) ? 1 : 0, (uint64_t)first_ray * (uint64_t)n_samples, scratch, scratch_bytes);
}
