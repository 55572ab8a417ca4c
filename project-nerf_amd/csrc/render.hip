// Evaluation entry of the vanilla field: render_rays without perturbation for ANY number of rays as one
// launch chain (SURVEY 8(b)3 `nerf_render_rays_fwd`; reference render_image, src/renderer.py:387-418, whose
// Python chunk loop exists only to bound activation memory).  Chunking is a property of the launch chain
// here: per chunk three kernels (stratified depths -> fused Fourier + decoder -> compositing) on the caller's
// stream, reusing ONE caller-provided workspace -- no allocation, no host synchronisation, graph-capturable.
#include "common.h"

extern "C" size_t nerf_render_rays_workspace_bytes(int64_t chunk_rays, int n_samples) {
  if (chunk_rays <= 0 || n_samples <= 0) return 0;
  const size_t n = (size_t)chunk_rays * (size_t)n_samples;
  return ((n * 4 + 255) / 256 * 256) * 2 + (n * 12 + 255) / 256 * 256;   // z, sigma, rgb
}

extern "C" int nerf_render_rays_fwd(const void* packed, const float* rays_o, const float* rays_d, int64_t n_rays,
                                    int n_samples, float near_plane, float far_plane, const float* bg, int64_t bg_rows,
                                    int64_t chunk_rays, void* workspace, float* out_rgb, float* out_depth, float* out_acc,
                                    nerf_stream_t stream) {
  NERF_REQUIRE(n_rays >= 0 && n_samples >= 2 && chunk_rays > 0, "nerf_render_rays_fwd: n_rays=%lld n_samples=%d chunk=%lld",
               (long long)n_rays, n_samples, (long long)chunk_rays);
  if (n_rays == 0) return NERF_OK;
  NERF_REQUIRE(packed && rays_o && rays_d && workspace && out_rgb && out_depth && out_acc, "nerf_render_rays_fwd: NULL pointer");
  NERF_REQUIRE(((uintptr_t)workspace & 255) == 0, "nerf_render_rays_fwd: workspace must be 256-byte aligned");
  NERF_REQUIRE(bg == nullptr || bg_rows == 1 || bg_rows == n_rays, "nerf_render_rays_fwd: bg_rows=%lld", (long long)bg_rows);
  NERF_REQUIRE(chunk_rays * (int64_t)n_samples < ((int64_t)1 << 31), "nerf_render_rays_fwd: chunk too large");
  const size_t n = (size_t)chunk_rays * (size_t)n_samples, plane = (n * 4 + 255) / 256 * 256;
  char* w = static_cast<char*>(workspace);
  float* z = reinterpret_cast<float*>(w);
  float* sigma = reinterpret_cast<float*>(w + plane);
  float* rgb = reinterpret_cast<float*>(w + 2 * plane);
  for (int64_t r0 = 0; r0 < n_rays; r0 += chunk_rays) {
    const int64_t r = n_rays - r0 < chunk_rays ? n_rays - r0 : chunk_rays;
    int rc = nerf_sample_rays(rays_o + r0 * 3, rays_d + r0 * 3, nullptr, r, n_samples, near_plane, far_plane, z, nullptr, nullptr, stream);
    if (rc != NERF_OK) return rc;
    rc = nerf_mlp_fwd(packed, rays_o + r0 * 3, rays_d + r0 * 3, z, r * n_samples, n_samples, rgb, sigma, nullptr, stream);
    if (rc != NERF_OK) return rc;
    rc = nerf_composite_fwd(rgb, sigma, z, rays_d + r0 * 3, bg == nullptr ? nullptr : (bg_rows > 1 ? bg + r0 * 3 : bg), bg_rows > 1 ? r : bg_rows,
                            nullptr, r, n_samples, out_rgb + r0 * 3, out_depth + r0, out_acc + r0, nullptr, nullptr, stream);
    if (rc != NERF_OK) return rc;
  }
  return NERF_OK;
}
