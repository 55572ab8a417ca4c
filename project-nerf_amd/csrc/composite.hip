// Alpha compositing along rays: one wavefront per ray, samples
// on lanes (K = ceil(S/64) consecutive samples per lane), transmittance as a
// wave-level exclusive product scan on DPP, weighted sums as wave reductions.
// HBM-bound: 20 B/sample in, 20 B/ray out (forward).
#include "common.h"

namespace nerf {

constexpr int kMaxPerLane = 16;  // S <= 1024 (instantiated for 1, 2, 3, 4, 6, 8, 12, 16 samples per lane)

template <int K>
struct RayCtx {
  float e[K];      // exp(-sigma*delta)
  float alpha[K];  // 1 - e
  float q[K];      // 1 - alpha + 1e-10
  float T[K];      // exclusive transmittance
  float z[K];
  float delta[K];
};

// loads sigma/z for the lane's K samples of ray r and builds alpha / transmittance
// slot map: sample (r, s) lives at row slots[r*S+s] of the compact rgb/sigma arrays, or is skipped
// (slot < 0: sigma = 0, rgb = 0); without a map the arrays are dense [R,S]
__device__ __forceinline__ int64_t row_of(const int* __restrict__ slots, int64_t dense_index) {
  return slots == nullptr ? dense_index : (int64_t)slots[dense_index];
}

template <int K>
__device__ __forceinline__ void ray_setup(const float* __restrict__ sigma, const float* __restrict__ z,
                                          const float* __restrict__ rays_d, const int* __restrict__ slots,
                                          int64_t r, int S, int lane, RayCtx<K>& c, float* sig_out, int64_t* row_out) {
  const float dx = rays_d[r * 3 + 0], dy = rays_d[r * 3 + 1], dz = rays_d[r * 3 + 2];
  const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
  const int s0 = lane * K;
  float zn[K + 1];
#pragma unroll
  for (int k = 0; k <= K; ++k) {
    const int s = s0 + k;
    zn[k] = s < S ? z[r * S + s] : 0.0f;
  }
  float local = 1.0f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int s = s0 + k;
    const bool valid = s < S;
    const int64_t row = valid ? row_of(slots, r * S + s) : -1;
    row_out[k] = row;
    const float sg = row >= 0 ? sigma[row] : 0.0f;
    sig_out[k] = sg;
    float dl = (s < S - 1) ? (zn[k + 1] - zn[k]) : 1e10f;   // src/renderer.py:213-214
    dl = dl * dnorm;
    const float e = valid ? expf(-sg * dl) : 1.0f;
    c.z[k] = zn[k];
    c.delta[k] = dl;
    c.e[k] = e;
    c.alpha[k] = 1.0f - e;
    c.q[k] = valid ? (1.0f - c.alpha[k] + 1e-10f) : 1.0f;
    local *= c.q[k];
  }
  // exclusive scan over lanes of the per-lane products, then walk the lane's own samples
  const float incl = wave_inclusive_prod(local);
  float run = dpp_row_shr<1>(incl, 1.0f);
  // row_shr does not cross 16-lane rows: patch lanes 16/32/48 from the previous row's last lane
  {
    const float p15 = lane_read(incl, 15), p31 = lane_read(incl, 31), p47 = lane_read(incl, 47);
    if (lane == 16) run = p15;
    if (lane == 32) run = p31;
    if (lane == 48) run = p47;
    if (lane == 0) run = 1.0f;
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    c.T[k] = run;
    run *= c.q[k];
  }
}

template <int K>
__global__ void __launch_bounds__(256)
composite_fwd_kernel(const float* __restrict__ rgb, const float* __restrict__ sigma,
                     const float* __restrict__ z, const float* __restrict__ rays_d,
                     const float* __restrict__ bg, int64_t bg_rows, const float* __restrict__ extra,
                     const int* __restrict__ slots, int64_t R, int S, float* __restrict__ out_rgb,
                     float* __restrict__ out_depth, float* __restrict__ out_acc, float* __restrict__ extra_map,
                     float* __restrict__ weights_out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < R; r += nwave) {
    RayCtx<K> c;
    float sg[K];
    int64_t row[K];
    ray_setup<K>(sigma, z, rays_d, slots, r, S, lane, c, sg, row);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, ad = 0.f, aw = 0.f, x0 = 0.f, x1 = 0.f, x2 = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int s = lane * K + k;
      if (s < S) {
        const float w = c.alpha[k] * c.T[k];
        if (row[k] >= 0) {
          const float* p = rgb + row[k] * 3;
          a0 += w * p[0];
          a1 += w * p[1];
          a2 += w * p[2];
        }
        ad += w * c.z[k];
        aw += w;
        if (extra != nullptr && row[k] >= 0) {
          const float* q = extra + row[k] * 3;
          x0 += w * q[0];
          x1 += w * q[1];
          x2 += w * q[2];
        }
        if (weights_out != nullptr) weights_out[r * S + s] = w;
      }
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
    ad = wave_sum(ad); aw = wave_sum(aw);
    if (extra != nullptr) { x0 = wave_sum(x0); x1 = wave_sum(x1); x2 = wave_sum(x2); }
    if (lane == 0) {
      if (bg != nullptr) {
        const float* b = bg + (bg_rows > 1 ? r * 3 : 0);
        const float rest = 1.0f - aw;
        a0 += rest * b[0];
        a1 += rest * b[1];
        a2 += rest * b[2];
      }
      out_rgb[r * 3 + 0] = a0;
      out_rgb[r * 3 + 1] = a1;
      out_rgb[r * 3 + 2] = a2;
      out_depth[r] = ad;
      out_acc[r] = aw;
      if (extra_map != nullptr) {
        extra_map[r * 3 + 0] = x0;
        extra_map[r * 3 + 1] = x1;
        extra_map[r * 3 + 2] = x2;
      }
    }
  }
}

// dL/dw_i = G_i = g_rgb.(c_i - bg) + g_depth z_i + g_acc + g_extra.x_i
// dL/dalpha_i = G_i T_i - (sum_{k>i} G_k w_k) / q_i ;  dL/dsigma_i = dL/dalpha_i * delta_i * e_i
template <int K>
__global__ void __launch_bounds__(256)
composite_bwd_kernel(const float* __restrict__ rgb, const float* __restrict__ sigma,
                     const float* __restrict__ z, const float* __restrict__ rays_d,
                     const float* __restrict__ bg, int64_t bg_rows, const float* __restrict__ extra,
                     const float* __restrict__ g_rgb, const float* __restrict__ g_depth,
                     const float* __restrict__ g_acc, const float* __restrict__ g_extra,
                     const int* __restrict__ slots, int64_t R, int S,
                     float* __restrict__ d_rgb, float* __restrict__ d_sigma, float* __restrict__ d_extra) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < R; r += nwave) {
    RayCtx<K> c;
    float sg[K];
    int64_t row[K];
    ray_setup<K>(sigma, z, rays_d, slots, r, S, lane, c, sg, row);
    const float gr0 = g_rgb[r * 3 + 0], gr1 = g_rgb[r * 3 + 1], gr2 = g_rgb[r * 3 + 2];
    const float gd = g_depth ? g_depth[r] : 0.0f;
    float ga = g_acc ? g_acc[r] : 0.0f;
    if (bg != nullptr) {
      const float* b = bg + (bg_rows > 1 ? r * 3 : 0);
      ga -= gr0 * b[0] + gr1 * b[1] + gr2 * b[2];
    }
    float gx0 = 0.f, gx1 = 0.f, gx2 = 0.f;
    if (g_extra != nullptr) { gx0 = g_extra[r * 3 + 0]; gx1 = g_extra[r * 3 + 1]; gx2 = g_extra[r * 3 + 2]; }
    float G[K], w[K];
    float local = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int s = lane * K + k;
      G[k] = 0.0f;
      w[k] = 0.0f;
      if (s < S) {
        w[k] = c.alpha[k] * c.T[k];
        float g = gd * c.z[k] + ga;
        if (row[k] >= 0) {
          const float* p = rgb + row[k] * 3;
          g += gr0 * p[0] + gr1 * p[1] + gr2 * p[2];
          float* o = d_rgb + row[k] * 3;
          o[0] = w[k] * gr0;
          o[1] = w[k] * gr1;
          o[2] = w[k] * gr2;
        }
        if (extra != nullptr && g_extra != nullptr && row[k] >= 0) {
          const float* q = extra + row[k] * 3;
          g += gx0 * q[0] + gx1 * q[1] + gx2 * q[2];
          if (d_extra != nullptr) {
            float* oe = d_extra + row[k] * 3;
            oe[0] = w[k] * gx0;
            oe[1] = w[k] * gx1;
            oe[2] = w[k] * gx2;
          }
        }
        G[k] = g;
        local += g * w[k];
      }
    }
    // suffix sum over later samples: lanes after this one, then within the lane
    float after = wave_exclusive_suffix_sum(local);  // sum over lanes > this lane
#pragma unroll
    for (int k = K - 1; k >= 0; --k) {
      const int s = lane * K + k;
      if (s < S && row[k] >= 0) {
        const float dalpha = G[k] * c.T[k] - after / c.q[k];
        d_sigma[row[k]] = dalpha * c.delta[k] * c.e[k];
      }
      after += G[k] * w[k];
    }
  }
}

// Training step around the compositing (reference run.py:324-337: render_rays -> nn.MSELoss ->
// loss.backward(), the compositing part): one pass per ray computes the pixel, its squared error,
// g = 2 (pixel - target) loss_weight and the gradients of rgb / sigma -- the forward and backward kernels
// above fused with the four elementwise / reduction launches of the loss in between.
// amax_out (optional): running maximum of the vanilla decoder's output-layer derivatives
// |d_rgb rgb (1 - rgb)| and |d_sigma| (sigma > 0), which nerf_mlp_bwd_dgrad_ex takes instead of its own pass.
// extra / d_extra / reg_out (optional, Part 3 / 4): a per-sample 3-vector e_s (the displacement delta_x) composited
// with the same weights, m = sum_s w_s e_s (render_rays' extras['mean_delta_x'], src/renderer.py:363-380), and the
// regulariser reg_weight * |m|^2 added to the objective (run.py:1838: mean(mean_delta_x^2) * deformation_reg_weight
// with reg_weight = deformation_reg_weight / (3 n_rays)): its gradient reaches e_s (d_extra) AND sigma (through w_s).
template <int K>
__global__ void __launch_bounds__(256)
composite_mse_bwd_kernel(const float* __restrict__ rgb, const float* __restrict__ sigma, const float* __restrict__ z,
                         const float* __restrict__ rays_d, const float* __restrict__ bg, int64_t bg_rows,
                         const float* __restrict__ target, float loss_weight, const int* __restrict__ slots, int64_t R, int S,
                         float* __restrict__ pred_out, float* __restrict__ loss_out, float* __restrict__ d_rgb,
                         float* __restrict__ d_sigma, float* __restrict__ amax_out,
                         const float* __restrict__ extra, float reg_weight, float* __restrict__ d_extra,
                         float* __restrict__ reg_out, float* __restrict__ extra_map, unsigned* __restrict__ sum_ws) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  float loss_local = 0.0f, amax = 0.0f, reg_local = 0.0f;
  for (int64_t r = wave; r < R; r += nwave) {
    RayCtx<K> c;
    float sg[K];
    int64_t row[K];
    ray_setup<K>(sigma, z, rays_d, slots, r, S, lane, c, sg, row);
    float col[K][3], w[K];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, aw = 0.f;
    float m0 = 0.f, m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int s = lane * K + k;
      w[k] = 0.0f;
      col[k][0] = col[k][1] = col[k][2] = 0.0f;
      if (s < S) {
        w[k] = c.alpha[k] * c.T[k];
        if (row[k] >= 0) {
          const float* p = rgb + row[k] * 3;
          col[k][0] = p[0]; col[k][1] = p[1]; col[k][2] = p[2];
          a0 += w[k] * p[0]; a1 += w[k] * p[1]; a2 += w[k] * p[2];
          if (extra != nullptr) {
            const float* e = extra + row[k] * 3;
            m0 += w[k] * e[0]; m1 += w[k] * e[1]; m2 += w[k] * e[2];
          }
        }
        aw += w[k];
      }
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); aw = wave_sum(aw);
    float gm0 = 0.f, gm1 = 0.f, gm2 = 0.f;
    if (extra != nullptr) {
      m0 = wave_sum(m0); m1 = wave_sum(m1); m2 = wave_sum(m2);
      gm0 = 2.0f * reg_weight * m0; gm1 = 2.0f * reg_weight * m1; gm2 = 2.0f * reg_weight * m2;
      if (lane == 0) {
        reg_local += reg_weight * (m0 * m0 + m1 * m1 + m2 * m2);
        if (extra_map != nullptr) { extra_map[r * 3 + 0] = m0; extra_map[r * 3 + 1] = m1; extra_map[r * 3 + 2] = m2; }
      }
    }
    float b0 = 0.f, b1 = 0.f, b2 = 0.f;
    if (bg != nullptr) {
      const float* b = bg + (bg_rows > 1 ? r * 3 : 0);
      b0 = b[0]; b1 = b[1]; b2 = b[2];
      const float rest = 1.0f - aw;
      a0 += rest * b0; a1 += rest * b1; a2 += rest * b2;
    }
    const float e0 = a0 - target[r * 3 + 0], e1 = a1 - target[r * 3 + 1], e2 = a2 - target[r * 3 + 2];
    if (lane == 0) {
      loss_local += (e0 * e0 + e1 * e1 + e2 * e2) * loss_weight;
      if (pred_out != nullptr) { pred_out[r * 3 + 0] = a0; pred_out[r * 3 + 1] = a1; pred_out[r * 3 + 2] = a2; }
    }
    const float gr0 = 2.0f * e0 * loss_weight, gr1 = 2.0f * e1 * loss_weight, gr2 = 2.0f * e2 * loss_weight;
    const float ga = -(gr0 * b0 + gr1 * b1 + gr2 * b2);
    float G[K], local = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int s = lane * K + k;
      G[k] = 0.0f;
      if (s < S) {
        float g = ga;
        if (row[k] >= 0) {
          g += gr0 * col[k][0] + gr1 * col[k][1] + gr2 * col[k][2];
          if (extra != nullptr) {
            const float* e = extra + row[k] * 3;
            g += gm0 * e[0] + gm1 * e[1] + gm2 * e[2];
            float* oe = d_extra + row[k] * 3;
            oe[0] = w[k] * gm0; oe[1] = w[k] * gm1; oe[2] = w[k] * gm2;
          }
          float* o = d_rgb + row[k] * 3;
          const float d0 = w[k] * gr0, d1 = w[k] * gr1, d2 = w[k] * gr2;
          o[0] = d0; o[1] = d1; o[2] = d2;
          if (amax_out != nullptr)
            amax = fmaxf(amax, fmaxf(fabsf(d0 * col[k][0] * (1.0f - col[k][0])),
                                     fmaxf(fabsf(d1 * col[k][1] * (1.0f - col[k][1])), fabsf(d2 * col[k][2] * (1.0f - col[k][2])))));
        }
        G[k] = g;
        local += g * w[k];
      }
    }
    float after = wave_exclusive_suffix_sum(local);
#pragma unroll
    for (int k = K - 1; k >= 0; --k) {
      const int s = lane * K + k;
      if (s < S && row[k] >= 0) {
        const float dalpha = G[k] * c.T[k] - after / c.q[k];
        const float ds = dalpha * c.delta[k] * c.e[k];
        d_sigma[row[k]] = ds;
        if (amax_out != nullptr && sg[k] > 0.0f) amax = fmaxf(amax, fabsf(ds));
      }
      after += G[k] * w[k];
    }
  }
  // one atomic per workgroup for the loss and for the maximum (same-address atomics serialise in L2)
  __shared__ float part[12];
  loss_local = wave_sum(loss_local);
  reg_local = wave_sum(reg_local);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
  if (lane == 0) { part[threadIdx.x >> 6] = loss_local; part[4 + (threadIdx.x >> 6)] = amax; part[8 + (threadIdx.x >> 6)] = reg_local; }
  __syncthreads();
  if (sum_ws != nullptr) {
    // no atomics at all: the workgroup's three partials go to sum_ws (plain stores), composite_sums_kernel adds them in workgroup
    // order afterwards.  Same-address atomics retire one after the other (~20 ns each): two per workgroup on 512 workgroups
    // were most of this kernel's time
    if (threadIdx.x == 0) {
      float* ws = reinterpret_cast<float*>(sum_ws);
      ws[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
      ws[gridDim.x + blockIdx.x] = (part[8] + part[9]) + (part[10] + part[11]);
      ws[2 * gridDim.x + blockIdx.x] = fmaxf(fmaxf(part[4], part[5]), fmaxf(part[6], part[7]));
    }
    return;
  }
  if (threadIdx.x == 0) {
    atomicAdd(loss_out, (part[0] + part[1]) + (part[2] + part[3]));
    if (reg_out != nullptr) atomicAdd(reg_out, (part[8] + part[9]) + (part[10] + part[11]));
    if (amax_out != nullptr) {
      const float m = fmaxf(fmaxf(part[4], part[5]), fmaxf(part[6], part[7]));
      if (m == m && m < 3.0e38f) atomicMax(reinterpret_cast<unsigned*>(amax_out), __builtin_bit_cast(unsigned, m));
    }
  }
}

// the partials of composite_mse_bwd_kernel, added in workgroup order (one workgroup; the same bits every run)
__global__ void __launch_bounds__(256) composite_sums_kernel(const float* __restrict__ ws, int n_blocks, float* __restrict__ loss_out,
                                                             float* __restrict__ reg_out, float* __restrict__ amax_out) {
  __shared__ float part[3][4];
  float s0 = 0.0f, s1 = 0.0f, m = 0.0f;
  for (int b = threadIdx.x; b < n_blocks; b += 256) {
    s0 += ws[b];
    s1 += ws[n_blocks + b];
    const float v = ws[2 * n_blocks + b];
    if (v == v && v < 3.0e38f) m = fmaxf(m, v);
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = s0; part[1][threadIdx.x >> 6] = s1; part[2][threadIdx.x >> 6] = m; }
  __syncthreads();
  if (threadIdx.x == 0) {
    *loss_out += (part[0][0] + part[0][1]) + (part[0][2] + part[0][3]);
    if (reg_out != nullptr) *reg_out += (part[1][0] + part[1][1]) + (part[1][2] + part[1][3]);
    if (amax_out != nullptr) *amax_out = fmaxf(*amax_out, fmaxf(fmaxf(part[2][0], part[2][1]), fmaxf(part[2][2], part[2][3])));
  }
}

// samples per lane: the smallest instantiated K with 64 K >= S (lanes past S are masked in the kernels)
static int per_lane(int S) {
  const int k = (S + 63) / 64;
  return k <= 4 ? k : (k <= 6 ? 6 : (k <= 8 ? 8 : (k <= 12 ? 12 : 16)));
}

}  // namespace nerf

using namespace nerf;

#define DISPATCH_K(K, KERNEL, ...)                                                        \
  switch (K) {                                                                            \
    case 1: hipLaunchKernelGGL(KERNEL<1>, grid, dim3(256), 0, as_stream(stream), __VA_ARGS__); break; \
    case 2: hipLaunchKernelGGL(KERNEL<2>, grid, dim3(256), 0, as_stream(stream), __VA_ARGS__); break; \
    case 3: hipLaunchKernelGGL(KERNEL<3>, grid, dim3(256), 0, as_stream(stream), __VA_ARGS__); break; \
    case 4: hipLaunchKernelGGL(KERNEL<4>, grid, dim3(256), 0, as_stream(stream), __VA_ARGS__); break; \
    case 6: hipLaunchKernelGGL(KERNEL<6>, grid, dim3(256), 0, as_stream(stream), __VA_ARGS__); break; \
    case 8: hipLaunchKernelGGL(KERNEL<8>, grid, dim3(256), 0, as_stream(stream), __VA_ARGS__); break; \
    case 12: hipLaunchKernelGGL(KERNEL<12>, grid, dim3(256), 0, as_stream(stream), __VA_ARGS__); break; \
    default: hipLaunchKernelGGL(KERNEL<16>, grid, dim3(256), 0, as_stream(stream), __VA_ARGS__); break; \
  }

static int composite_fwd_impl(const float* rgb, const float* sigma, const float* z, const float* rays_d,
                              const float* bg, int64_t bg_rows, const float* extra, const int* slots,
                              int64_t n_rays, int n_samples, float* out_rgb, float* out_depth, float* out_acc,
                              float* extra_map, float* weights_out, nerf_stream_t stream) {
  NERF_REQUIRE(n_rays >= 0 && n_samples >= 1 && n_samples <= 64 * kMaxPerLane,
               "nerf_composite_fwd: n_rays=%lld n_samples=%d (max %d)", (long long)n_rays, n_samples,
               64 * kMaxPerLane);
  if (n_rays == 0) return NERF_OK;
  NERF_REQUIRE(rgb && sigma && z && rays_d && out_rgb && out_depth && out_acc,
               "nerf_composite_fwd: NULL pointer");
  NERF_REQUIRE(bg == nullptr || bg_rows == 1 || bg_rows == n_rays, "nerf_composite_fwd: bg_rows=%lld",
               (long long)bg_rows);
  NERF_REQUIRE((extra == nullptr) == (extra_map == nullptr), "nerf_composite_fwd: extra/extra_map mismatch");
  int64_t blocks = (n_rays + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;
  const dim3 grid((int)blocks);
  DISPATCH_K(per_lane(n_samples), composite_fwd_kernel, rgb, sigma, z, rays_d, bg, bg_rows, extra, slots, n_rays,
             n_samples, out_rgb, out_depth, out_acc, extra_map, weights_out);
  return check_launch("nerf_composite_fwd");
}

extern "C" int nerf_composite_fwd(const float* rgb, const float* sigma, const float* z,
                                  const float* rays_d, const float* bg, int64_t bg_rows,
                                  const float* extra, int64_t n_rays, int n_samples, float* out_rgb,
                                  float* out_depth, float* out_acc, float* extra_map,
                                  float* weights_out, nerf_stream_t stream) {
  return composite_fwd_impl(rgb, sigma, z, rays_d, bg, bg_rows, extra, nullptr, n_rays, n_samples, out_rgb, out_depth,
                            out_acc, extra_map, weights_out, stream);
}

extern "C" int nerf_composite_fwd_indexed(const float* rgb_compact, const float* sigma_compact, const int* slot_of_sample,
                                          const float* z, const float* rays_d, const float* bg, int64_t bg_rows,
                                          int64_t n_rays, int n_samples, float* out_rgb, float* out_depth,
                                          float* out_acc, nerf_stream_t stream) {
  NERF_REQUIRE(n_rays == 0 || slot_of_sample != nullptr, "nerf_composite_fwd_indexed: slot map is NULL");
  return composite_fwd_impl(rgb_compact, sigma_compact, z, rays_d, bg, bg_rows, nullptr, slot_of_sample, n_rays, n_samples,
                            out_rgb, out_depth, out_acc, nullptr, nullptr, stream);
}

static int composite_bwd_impl(const float* rgb, const float* sigma, const float* z, const float* rays_d,
                              const float* bg, int64_t bg_rows, const float* extra, const float* g_rgb,
                              const float* g_depth, const float* g_acc, const float* g_extra, const int* slots,
                              int64_t n_rays, int n_samples, float* d_rgb, float* d_sigma, float* d_extra,
                              nerf_stream_t stream) {
  NERF_REQUIRE(n_rays >= 0 && n_samples >= 1 && n_samples <= 64 * kMaxPerLane,
               "nerf_composite_bwd: n_rays=%lld n_samples=%d", (long long)n_rays, n_samples);
  if (n_rays == 0) return NERF_OK;
  NERF_REQUIRE(rgb && sigma && z && rays_d && g_rgb && d_rgb && d_sigma, "nerf_composite_bwd: NULL pointer");
  NERF_REQUIRE(bg == nullptr || bg_rows == 1 || bg_rows == n_rays, "nerf_composite_bwd: bg_rows=%lld",
               (long long)bg_rows);
  int64_t blocks = (n_rays + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;
  const dim3 grid((int)blocks);
  DISPATCH_K(per_lane(n_samples), composite_bwd_kernel, rgb, sigma, z, rays_d, bg, bg_rows, extra, g_rgb,
             g_depth, g_acc, g_extra, slots, n_rays, n_samples, d_rgb, d_sigma, d_extra);
  return check_launch("nerf_composite_bwd");
}

extern "C" int nerf_composite_bwd(const float* rgb, const float* sigma, const float* z,
                                  const float* rays_d, const float* bg, int64_t bg_rows,
                                  const float* extra, const float* g_rgb, const float* g_depth,
                                  const float* g_acc, const float* g_extra, int64_t n_rays,
                                  int n_samples, float* d_rgb, float* d_sigma, float* d_extra,
                                  nerf_stream_t stream) {
  return composite_bwd_impl(rgb, sigma, z, rays_d, bg, bg_rows, extra, g_rgb, g_depth, g_acc, g_extra, nullptr, n_rays,
                            n_samples, d_rgb, d_sigma, d_extra, stream);
}

extern "C" int nerf_composite_bwd_indexed(const float* rgb_compact, const float* sigma_compact, const int* slot_of_sample,
                                          const float* z, const float* rays_d, const float* bg, int64_t bg_rows,
                                          const float* g_rgb, const float* g_depth, const float* g_acc, int64_t n_rays,
                                          int n_samples, float* d_rgb_compact, float* d_sigma_compact,
                                          nerf_stream_t stream) {
  NERF_REQUIRE(n_rays == 0 || slot_of_sample != nullptr, "nerf_composite_bwd_indexed: slot map is NULL");
  return composite_bwd_impl(rgb_compact, sigma_compact, z, rays_d, bg, bg_rows, nullptr, g_rgb, g_depth, g_acc, nullptr,
                            slot_of_sample, n_rays, n_samples, d_rgb_compact, d_sigma_compact, nullptr, stream);
}

// workgroups of the fused compositing + loss backward: composite_wgs_per_cu per CU of THIS device (at least one)
static int64_t mse_blocks(int64_t n_rays, bool partials) {
  int n_cu = 256;
  (void)device_cu_count(&n_cu);
  // with partial sums (no atomics) the kernel can spread over the chip: eight workgroups of four waves per CU
  const int per_cu = partials ? 8 : (options().composite_wgs_per_cu < 1 ? 1 : options().composite_wgs_per_cu);
  int64_t cap = (int64_t)n_cu * per_cu;
  if (cap > kOrderedSumMaxBlocks) cap = kOrderedSumMaxBlocks;
  const int64_t blocks = (n_rays + 3) / 4;
  return blocks < cap ? blocks : cap;
}
static int finish_sums(const float* sum_ws, unsigned n_blocks, float* loss, float* reg, float* amax, nerf_stream_t stream) {
  static_assert(NERF_SUM_WS_FLOATS >= 3 * kOrderedSumMaxBlocks, "three partials per workgroup");
  if (sum_ws == nullptr) return NERF_OK;
  hipLaunchKernelGGL(composite_sums_kernel, dim3(1), dim3(256), 0, as_stream(stream), sum_ws, (int)n_blocks, loss, reg, amax);
  return check_launch("nerf_composite_mse_bwd (sums)");
}

extern "C" int nerf_composite_mse_bwd(const float* rgb, const float* sigma, const int* slot_of_sample, const float* z,
                                      const float* rays_d, const float* bg, int64_t bg_rows, const float* target,
                                      float loss_weight, int64_t n_rays, int n_samples, float* pred_out, float* loss_accum,
                                      float* d_rgb, float* d_sigma, float* amax_accum, float* sum_ws, nerf_stream_t stream) {
  NERF_REQUIRE(n_rays >= 0 && n_samples >= 1 && n_samples <= 64 * kMaxPerLane,
               "nerf_composite_mse_bwd: n_rays=%lld n_samples=%d (max %d)", (long long)n_rays, n_samples, 64 * kMaxPerLane);
  if (n_rays == 0) return NERF_OK;
  NERF_REQUIRE(rgb && sigma && z && rays_d && target && loss_accum && d_rgb && d_sigma, "nerf_composite_mse_bwd: NULL pointer");
  NERF_REQUIRE(bg == nullptr || bg_rows == 1 || bg_rows == n_rays, "nerf_composite_mse_bwd: bg_rows=%lld", (long long)bg_rows);
  const dim3 grid((unsigned)mse_blocks(n_rays, sum_ws != nullptr));
  DISPATCH_K(per_lane(n_samples), composite_mse_bwd_kernel, rgb, sigma, z, rays_d, bg, bg_rows, target, loss_weight,
             slot_of_sample, n_rays, n_samples, pred_out, loss_accum, d_rgb, d_sigma, amax_accum,
             (const float*)nullptr, 0.0f, (float*)nullptr, (float*)nullptr, (float*)nullptr, reinterpret_cast<unsigned*>(sum_ws));
  if (int rc = check_launch("nerf_composite_mse_bwd"); rc != NERF_OK) return rc;
  return finish_sums(sum_ws, grid.x, loss_accum, nullptr, amax_accum, stream);
}

extern "C" int nerf_composite_mse_reg_bwd(const float* rgb, const float* sigma, const int* slot_of_sample, const float* z,
                                          const float* rays_d, const float* bg, int64_t bg_rows, const float* target,
                                          float loss_weight, const float* extra, float reg_weight, int64_t n_rays, int n_samples,
                                          float* pred_out, float* extra_map, float* loss_accum, float* reg_accum, float* d_rgb,
                                          float* d_sigma, float* d_extra, float* sum_ws, nerf_stream_t stream) {
  NERF_REQUIRE(n_rays >= 0 && n_samples >= 1 && n_samples <= 64 * kMaxPerLane,
               "nerf_composite_mse_reg_bwd: n_rays=%lld n_samples=%d (max %d)", (long long)n_rays, n_samples, 64 * kMaxPerLane);
  if (n_rays == 0) return NERF_OK;
  NERF_REQUIRE(rgb && sigma && z && rays_d && target && loss_accum && d_rgb && d_sigma && extra && d_extra && reg_accum,
               "nerf_composite_mse_reg_bwd: NULL pointer");
  NERF_REQUIRE(bg == nullptr || bg_rows == 1 || bg_rows == n_rays, "nerf_composite_mse_reg_bwd: bg_rows=%lld", (long long)bg_rows);
  const dim3 grid((unsigned)mse_blocks(n_rays, sum_ws != nullptr));
  DISPATCH_K(per_lane(n_samples), composite_mse_bwd_kernel, rgb, sigma, z, rays_d, bg, bg_rows, target, loss_weight,
             slot_of_sample, n_rays, n_samples, pred_out, loss_accum, d_rgb, d_sigma, (float*)nullptr, extra, reg_weight, d_extra,
             reg_accum, extra_map, reinterpret_cast<unsigned*>(sum_ws));
  if (int rc = check_launch("nerf_composite_mse_reg_bwd"); rc != NERF_OK) return rc;
  return finish_sums(sum_ws, grid.x, loss_accum, reg_accum, nullptr, stream);
  return check_launch("nerf_composite_mse_reg_bwd");
}
