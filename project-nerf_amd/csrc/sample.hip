// Stratified depths, ray points and occupancy lookup.
// Built with -ffp-contract=off: the reference's eager ops round after every
// multiply and add, and sample / voxel indices must match it bit for bit.
#include "common.h"

namespace nerf {

// torch.linspace(0,1,S) on CPU: step = 1/(S-1) in fp32; the lower half walks up
// from 0, the upper half walks down from 1, each with ONE rounding.
__device__ __forceinline__ float linspace01(int i, int n, float step) {
  if (i < n / 2) return mul_rn(step, (float)i);
  return __builtin_fmaf(-step, (float)(n - 1 - i), 1.0f);
}

__device__ __forceinline__ float plain_depth(int i, int n, float step, float near_p, float far_p) {
  const float t = linspace01(i, n, step);
  // near*(1-t) + far*t, three roundings (src/renderer.py:190)
  return add_rn(mul_rn(near_p, sub_rn(1.0f, t)), mul_rn(far_p, t));
}

// jittered depth of sample i given its uniform draw u (src/renderer.py:195-199: mids / upper / lower)
__device__ __forceinline__ float jitter_depth(int i, int n, float step, float near_p, float far_p, float u) {
  const float zi = plain_depth(i, n, step, near_p, far_p);
  float lo = zi, hi = zi;
  if (i > 0) lo = mul_rn(0.5f, add_rn(zi, plain_depth(i - 1, n, step, near_p, far_p)));
  if (i < n - 1) hi = mul_rn(0.5f, add_rn(plain_depth(i + 1, n, step, near_p, far_p), zi));
  return add_rn(lo, mul_rn(sub_rn(hi, lo), u));
}

__device__ __forceinline__ float sample_depth(int i, int n, float step, float near_p, float far_p,
                                              const float* u_row) {
  if (u_row == nullptr) return plain_depth(i, n, step, near_p, far_p);
  return jitter_depth(i, n, step, near_p, far_p, u_row[i]);
}

__global__ void __launch_bounds__(256)
sample_rays_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                   const float* __restrict__ u, int64_t n_rays, int n_samples, float near_p,
                   float far_p, float step, float* __restrict__ z_out, float* __restrict__ pts_out,
                   float* __restrict__ dirs_out) {
  const int64_t total = n_rays * (int64_t)n_samples;
  for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < total;
       g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = g / n_samples;
    const int s = (int)(g - r * n_samples);
    const float z = sample_depth(s, n_samples, step, near_p, far_p, u ? u + r * n_samples : nullptr);
    z_out[g] = z;
    if (pts_out != nullptr || dirs_out != nullptr) {
      const float ox = rays_o[r * 3 + 0], oy = rays_o[r * 3 + 1], oz = rays_o[r * 3 + 2];
      const float dx = rays_d[r * 3 + 0], dy = rays_d[r * 3 + 1], dz = rays_d[r * 3 + 2];
      if (pts_out != nullptr) {
        pts_out[g * 3 + 0] = add_rn(ox, mul_rn(dx, z));
        pts_out[g * 3 + 1] = add_rn(oy, mul_rn(dy, z));
        pts_out[g * 3 + 2] = add_rn(oz, mul_rn(dz, z));
      }
      if (dirs_out != nullptr) {
        const float nrm =
            sqrtf(add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz)));
        dirs_out[g * 3 + 0] = (dx / nrm);
        dirs_out[g * 3 + 1] = (dy / nrm);
        dirs_out[g * 3 + 2] = (dz / nrm);
      }
    }
  }
}

__global__ void __launch_bounds__(256)
active_mask_kernel(const float* __restrict__ pts, int64_t n, const uint8_t* __restrict__ grid, int res,
                   float bound, float scale, uint8_t* __restrict__ mask_out,
                   int64_t* __restrict__ idx_out) {
  for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < n;
       g += (int64_t)gridDim.x * blockDim.x) {
    int64_t v[3];
    bool inside = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      // (pts + bound) * scale, then .long(): truncation toward zero (src/renderer.py:145)
      const float f = mul_rn(add_rn(pts[g * 3 + a], bound), scale);
      v[a] = (int64_t)f;
      inside = inside && v[a] >= 0 && v[a] < res;
      if (idx_out != nullptr) idx_out[g * 3 + a] = v[a];
    }
    uint8_t m = 0;
    if (inside) m = grid[(v[0] * res + v[1]) * res + v[2]] != 0;
    mask_out[g] = m;
  }
}

// Batch sampling from GPU-resident frames: the body of
// BlenderDataset.sample_random_rays after the three index draws (reference src/dataset.py:150-171)
// -- camera-space direction of the pixel (no +0.5 centre offset, -y, -z), rotation by c2w[:3,:3],
// normalisation, origin = c2w[:3,3] * scene_scale, RGBA fetch -- as one kernel instead of a
// batched 3x3 GEMM plus a dozen elementwise launches.
struct GatherArgs {
  const float* images;
  const float* poses;
  int H, W;
  float half_w, half_h, focal, scene_scale;
  const float* bg;
  float* rays_o;
  float* rays_d;
  float* rgba;
  float* target;
};

// ray r <- pixel (im, py, px): direction, origin, RGBA and (optionally) the composited target
__device__ __forceinline__ void gather_one(const GatherArgs& a, int64_t r, int64_t im, int64_t py, int64_t px) {
  const float* __restrict__ images = a.images;
  const float* __restrict__ poses = a.poses;
  float* __restrict__ rays_o = a.rays_o;
  float* __restrict__ rays_d = a.rays_d;
  float* __restrict__ rgba = a.rgba;
  float* __restrict__ target = a.target;
  const float* __restrict__ bg = a.bg;
  const int H = a.H, W = a.W;
  const float half_w = a.half_w, half_h = a.half_h, focal = a.focal, scene_scale = a.scene_scale;
  {
    const float* c2w = poses + im * 16;
    const float x = sub_rn((float)px, half_w) / focal;
    const float y = -(sub_rn((float)py, half_h) / focal);
    const float z = -1.0f;
    float d[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
      d[i] = add_rn(add_rn(mul_rn(c2w[4 * i + 0], x), mul_rn(c2w[4 * i + 1], y)), mul_rn(c2w[4 * i + 2], z));
    const float nrm = sqrtf(add_rn(add_rn(mul_rn(d[0], d[0]), mul_rn(d[1], d[1])), mul_rn(d[2], d[2])));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      rays_d[r * 3 + i] = d[i] / nrm;
      const float o = c2w[4 * i + 3];
      rays_o[r * 3 + i] = scene_scale != 1.0f ? mul_rn(o, scene_scale) : o;
    }
    const float4 c = *reinterpret_cast<const float4*>(images + ((im * H + py) * W + px) * 4);
    if (rgba != nullptr) *reinterpret_cast<float4*>(rgba + r * 4) = c;
    if (target != nullptr) {
      // target = rgb * a + bg * (1 - a), every product and sum rounded on its own (run.py:317-322)
      const float rest = sub_rn(1.0f, c.w);
      target[r * 3 + 0] = add_rn(mul_rn(c.x, c.w), mul_rn(bg[0], rest));
      target[r * 3 + 1] = add_rn(mul_rn(c.y, c.w), mul_rn(bg[1], rest));
      target[r * 3 + 2] = add_rn(mul_rn(c.z, c.w), mul_rn(bg[2], rest));
    }
  }
}

__global__ void __launch_bounds__(256)
gather_rays_kernel(const GatherArgs a, const int64_t* __restrict__ img_idx, const int64_t* __restrict__ pix_y,
                   const int64_t* __restrict__ pix_x, int64_t batch) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < batch; r += (int64_t)gridDim.x * blockDim.x) {
    // pix_y == NULL: img_idx holds ONE flat draw over all pixels of all frames (image, row, column)
    int64_t im = img_idx[r], py, px;
    if (pix_y != nullptr) { py = pix_y[r]; px = pix_x[r]; }
    else { px = im % a.W; py = (im / a.W) % a.H; im = im / ((int64_t)a.W * a.H); }
    gather_one(a, r, im, py, px);
  }
}

// One kernel for the data side of a training step (reference run.py:314-322 + renderer.py:186-201):
// per ray one uniform draw over all pixels of all frames -> origin, direction, composited target; per sample
// one uniform draw -> jittered stratified depth.  Thread per (ray, sample); sample 0 also forms the ray.
__global__ void __launch_bounds__(256)
train_batch_kernel(const GatherArgs a, uint64_t n_pixels, uint64_t key, uint64_t counter, int64_t batch, int n_samples,
                   float near_p, float far_p, float step, int perturb, float* __restrict__ z_out, int64_t first_ray) {
  // first_ray: this call forms rays [first_ray, first_ray + batch) of a larger (global) batch -- the draws are indexed by
  // the GLOBAL ray / sample number, the outputs by the local one
  const int64_t total = batch * (int64_t)n_samples;
  for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = g / n_samples;
    const int s = (int)(g - r * n_samples);
    if (perturb) {
      const uint64_t gg = (uint64_t)(g + first_ray * n_samples);
      const float u = (float)(squares32((counter << 40) + gg, key) >> 8) * 5.9604644775390625e-08f;   // [0, 1), 24 bits
      z_out[g] = jitter_depth(s, n_samples, step, near_p, far_p, u);
    } else {
      z_out[g] = plain_depth(s, n_samples, step, near_p, far_p);
    }
    if (s == 0) {
      const uint64_t c0 = (counter << 40) + ((uint64_t)1 << 39) + 2 * (uint64_t)(r + first_ray);
      const uint64_t r64 = ((uint64_t)squares32(c0, key) << 32) | squares32(c0 + 1, key);
      const uint64_t idx = __umul64hi(r64, n_pixels);            // uniform over [0, n_pixels)
      const int64_t px = (int64_t)(idx % (uint64_t)a.W), py = (int64_t)((idx / (uint64_t)a.W) % (uint64_t)a.H);
      gather_one(a, r, (int64_t)(idx / ((uint64_t)a.W * a.H)), py, px);
    }
  }
}

static inline int grid_for(int64_t work, int block) {
  int64_t b = (work + block - 1) / block;
  if (b > 256 * 8) b = 256 * 8;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace nerf

using namespace nerf;

extern "C" int nerf_sample_rays(const float* rays_o, const float* rays_d, const float* u,
                                int64_t n_rays, int n_samples, float near_plane, float far_plane,
                                float* z_out, float* pts_out, float* dirs_out, nerf_stream_t stream) {
  NERF_REQUIRE(n_rays >= 0 && n_samples >= 2, "nerf_sample_rays: n_rays=%lld n_samples=%d",
               (long long)n_rays, n_samples);
  if (n_rays == 0) return NERF_OK;
  NERF_REQUIRE(z_out != nullptr, "nerf_sample_rays: z_out is NULL");
  NERF_REQUIRE((pts_out == nullptr && dirs_out == nullptr) || (rays_o && rays_d),
               "nerf_sample_rays: rays_o/rays_d required when pts/dirs are requested");
  const float step = 1.0f / (float)(n_samples - 1);
  hipLaunchKernelGGL(sample_rays_kernel, dim3(grid_for(n_rays * n_samples, 256)), dim3(256), 0,
                     as_stream(stream), rays_o, rays_d, u, n_rays, n_samples, near_plane, far_plane,
                     step, z_out, pts_out, dirs_out);
  return check_launch("nerf_sample_rays");
}

extern "C" int nerf_active_mask(const float* pts, int64_t n, const uint8_t* binary_grid,
                                int resolution, float bound, uint8_t* mask_out, int64_t* idx_out,
                                nerf_stream_t stream) {
  NERF_REQUIRE(n >= 0 && resolution > 0 && bound > 0.0f, "nerf_active_mask: bad sizes");
  NERF_REQUIRE(n == 0 || (pts && binary_grid && mask_out), "nerf_active_mask: NULL pointer");
  if (n == 0) return NERF_OK;
  // the Python double res/(2*bound) is demoted to fp32 before the multiply
  const float scale = (float)((double)resolution / (2.0 * (double)bound));
  hipLaunchKernelGGL(active_mask_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), pts,
                     n, binary_grid, resolution, bound, scale, mask_out, idx_out);
  return check_launch("nerf_active_mask");
}

extern "C" int nerf_gather_rays(const float* images, const float* poses, const int64_t* img_idx, const int64_t* pix_y,
                                const int64_t* pix_x, int64_t batch, int n_images, int H, int W, float focal,
                                float scene_scale, float* rays_o, float* rays_d, float* rgba, nerf_stream_t stream) {
  NERF_REQUIRE(batch >= 0 && n_images > 0 && H > 0 && W > 0 && focal > 0.0f, "nerf_gather_rays: bad sizes");
  if (batch == 0) return NERF_OK;
  NERF_REQUIRE(images && poses && img_idx && pix_y && pix_x && rays_o && rays_d && rgba, "nerf_gather_rays: NULL pointer");
  NERF_REQUIRE((((uintptr_t)images | (uintptr_t)rgba) & 15) == 0, "nerf_gather_rays: images / rgba must be 16-byte aligned");
  const GatherArgs a{images, poses, H, W, (float)(W * 0.5), (float)(H * 0.5), focal, scene_scale, nullptr, rays_o, rays_d, rgba, nullptr};
  hipLaunchKernelGGL(gather_rays_kernel, dim3(grid_for(batch, 256)), dim3(256), 0, as_stream(stream), a, img_idx, pix_y, pix_x, batch);
  return check_launch("nerf_gather_rays");
}

extern "C" int nerf_gather_batch(const float* images, const float* poses, const int64_t* flat_idx, int64_t batch, int n_images,
                                 int H, int W, float focal, float scene_scale, const float* bg, float* rays_o, float* rays_d,
                                 float* rgba, float* target, nerf_stream_t stream) {
  NERF_REQUIRE(batch >= 0 && n_images > 0 && H > 0 && W > 0 && focal > 0.0f, "nerf_gather_batch: bad sizes");
  if (batch == 0) return NERF_OK;
  NERF_REQUIRE(images && poses && flat_idx && rays_o && rays_d && (rgba || target), "nerf_gather_batch: NULL pointer");
  NERF_REQUIRE((target == nullptr) == (bg == nullptr), "nerf_gather_batch: target and bg go together");
  NERF_REQUIRE((((uintptr_t)images | (uintptr_t)rgba) & 15) == 0, "nerf_gather_batch: images / rgba must be 16-byte aligned");
  const GatherArgs a{images, poses, H, W, (float)(W * 0.5), (float)(H * 0.5), focal, scene_scale, bg, rays_o, rays_d, rgba, target};
  hipLaunchKernelGGL(gather_rays_kernel, dim3(grid_for(batch, 256)), dim3(256), 0, as_stream(stream), a, flat_idx, nullptr, nullptr, batch);
  return check_launch("nerf_gather_batch");
}

static int train_batch_impl(const float* images, const float* poses, int n_images, int H, int W, float focal,
                            float scene_scale, const float* bg, uint64_t seed, uint64_t counter, int64_t first_ray, int64_t batch,
                            int n_samples, float near_plane, float far_plane, int perturb, float* rays_o, float* rays_d,
                            float* rgba, float* target, float* z_out, nerf_stream_t stream);

extern "C" int nerf_train_batch(const float* images, const float* poses, int n_images, int H, int W, float focal,
                                float scene_scale, const float* bg, uint64_t seed, uint64_t counter, int64_t batch,
                                int n_samples, float near_plane, float far_plane, int perturb, float* rays_o, float* rays_d,
                                float* rgba, float* target, float* z_out, nerf_stream_t stream) {
  return train_batch_impl(images, poses, n_images, H, W, focal, scene_scale, bg, seed, counter, 0, batch, n_samples, near_plane,
                          far_plane, perturb, rays_o, rays_d, rgba, target, z_out, stream);
}

extern "C" int nerf_train_batch_shard(const float* images, const float* poses, int n_images, int H, int W, float focal,
                                      float scene_scale, const float* bg, uint64_t seed, uint64_t counter, int64_t first_ray,
                                      int64_t batch, int n_samples, float near_plane, float far_plane, int perturb, float* rays_o,
                                      float* rays_d, float* rgba, float* target, float* z_out, nerf_stream_t stream) {
  NERF_REQUIRE(first_ray >= 0, "nerf_train_batch_shard: first_ray=%lld", (long long)first_ray);
  return train_batch_impl(images, poses, n_images, H, W, focal, scene_scale, bg, seed, counter, first_ray, batch, n_samples,
                          near_plane, far_plane, perturb, rays_o, rays_d, rgba, target, z_out, stream);
}

static int train_batch_impl(const float* images, const float* poses, int n_images, int H, int W, float focal,
                            float scene_scale, const float* bg, uint64_t seed, uint64_t counter, int64_t first_ray, int64_t batch,
                            int n_samples, float near_plane, float far_plane, int perturb, float* rays_o, float* rays_d,
                            float* rgba, float* target, float* z_out, nerf_stream_t stream) {
  NERF_REQUIRE(batch >= 0 && n_images > 0 && H > 0 && W > 0 && focal > 0.0f && n_samples >= 2, "nerf_train_batch: bad sizes");
  NERF_REQUIRE(counter < ((uint64_t)1 << 24) && (first_ray + batch) * (int64_t)n_samples < ((int64_t)1 << 39),
               "nerf_train_batch: counter / batch out of range");
  if (batch == 0) return NERF_OK;
  NERF_REQUIRE(images && poses && rays_o && rays_d && z_out && (rgba || target), "nerf_train_batch: NULL pointer");
  NERF_REQUIRE((target == nullptr) == (bg == nullptr), "nerf_train_batch: target and bg go together");
  NERF_REQUIRE((((uintptr_t)images | (uintptr_t)rgba) & 15) == 0, "nerf_train_batch: images / rgba must be 16-byte aligned");
  const uint64_t key = squares_key(seed);
  const GatherArgs a{images, poses, H, W, (float)(W * 0.5), (float)(H * 0.5), focal, scene_scale, bg, rays_o, rays_d, rgba, target};
  const float step = 1.0f / (float)(n_samples - 1);
  hipLaunchKernelGGL(train_batch_kernel, dim3(grid_for(batch * (int64_t)n_samples, 256)), dim3(256), 0, as_stream(stream), a,
                     (uint64_t)n_images * H * W, key, counter, batch, n_samples, near_plane, far_plane, step, perturb, z_out, first_ray);
  return check_launch("nerf_train_batch");
}
