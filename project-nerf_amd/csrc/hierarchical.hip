// Hierarchical (inverse-CDF) fine sampling, opt-in extension.
// NOT in the reference (grep for sample_pdf / searchsorted / importance over the tree: 0 hits);
// BASELINE.json names "64 coarse + 128 fine", so the build provides it following Mildenhall et al.
// 2020, section 5.2 -- PARITY UNPINNED, checked against oracle/nerf_oracle.py::sample_pdf.
//
// One wavefront per ray: bins = mid-points of the coarse depths, pdf = coarse weights[1:-1] + 1e-5,
// cdf by a wave-level prefix sum, each fine sample inverts the cdf by binary search in LDS, then
// coarse + fine depths are merged by a bitonic sort in LDS.
#include "common.h"

namespace nerf {

constexpr int kMaxCoarse = 256, kMaxTotal = 1024;

__device__ __forceinline__ float wave_inclusive_sum(float v) {
  v += dpp_row_shr<1>(v, 0.0f);
  v += dpp_row_shr<2>(v, 0.0f);
  v += dpp_row_shr<4>(v, 0.0f);
  v += dpp_row_shr<8>(v, 0.0f);
  const int lane = __lane_id();
  const float r0 = lane_read(v, 15), r1 = lane_read(v, 31), r2 = lane_read(v, 47);
  const int row = lane >> 4;
  return v + (row == 0 ? 0.0f : (row == 1 ? r0 : (row == 2 ? r0 + r1 : r0 + r1 + r2)));
}

__global__ void __launch_bounds__(64)
sample_pdf_kernel(const float* __restrict__ z, const float* __restrict__ weights, const float* __restrict__ u,
                  int64_t R, int S, int NF, int P, float* __restrict__ z_out) {
  __shared__ float cdf[kMaxCoarse];
  __shared__ float bins[kMaxCoarse];
  __shared__ float vals[kMaxTotal];
  const int lane = threadIdx.x;
  const int nb = S - 1;            // bin edges = cdf entries
  const int K = (nb + 63) / 64;
  for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
    const float* zr = z + r * S;
    const float* wr = weights + r * S;
    // pdf over the S-2 interior weights; entry e (1..nb-1) of the cdf = sum_{i<e} pdf_i
    float local[4], sum = 0.0f;
    for (int k = 0; k < K; ++k) {
      const int i = lane * K + k;                       // pdf index 0..S-3
      local[k] = i < S - 2 ? wr[i + 1] + 1e-5f : 0.0f;
      sum += local[k];
    }
    const float incl = wave_inclusive_sum(sum);
    const float total = lane_read(incl, 63);
    float run = incl - sum;                             // exclusive prefix of this lane
    for (int k = 0; k < K; ++k) {
      const int i = lane * K + k;
      if (i < nb) {
        cdf[i] = run / total;                           // cdf[0] = 0
        bins[i] = 0.5f * (zr[i + 1] + zr[i]);
      }
      run += local[k];
    }
    __syncthreads();
    const float ustep = NF > 1 ? 1.0f / (float)(NF - 1) : 0.0f;
    for (int j = lane; j < NF; j += 64) {
      float uj;
      if (u != nullptr) uj = u[r * NF + j];
      else uj = j < NF / 2 ? ustep * (float)j : __builtin_fmaf(-ustep, (float)(NF - 1 - j), 1.0f);   // torch.linspace(0,1,NF)
      int lo = 0, hi = nb;                              // searchsorted(cdf, u, right=True)
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] <= uj) lo = mid + 1; else hi = mid;
      }
      const int below = lo - 1 > 0 ? lo - 1 : 0, above = lo < nb - 1 ? lo : nb - 1;
      float denom = cdf[above] - cdf[below];
      if (denom < 1e-5f) denom = 1.0f;
      const float t = (uj - cdf[below]) / denom;
      vals[S + j] = bins[below] + t * (bins[above] - bins[below]);
    }
    for (int i = lane; i < S; i += 64) vals[i] = zr[i];
    for (int i = S + NF + lane; i < P; i += 64) vals[i] = __builtin_inff();
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = lane; i < P; i += 64) {
          const int l = i ^ j;
          if (l > i) {
            const float a = vals[i], b = vals[l];
            const bool up = (i & k) == 0;
            if ((a > b) == up) { vals[i] = b; vals[l] = a; }
          }
        }
        __syncthreads();
      }
    }
    for (int i = lane; i < S + NF; i += 64) z_out[r * (S + NF) + i] = vals[i];
    __syncthreads();
  }
}

}  // namespace nerf

extern "C" int nerf_sample_pdf(const float* z_coarse, const float* weights, const float* u, int64_t n_rays,
                               int n_coarse, int n_fine, float* z_out, nerf_stream_t stream) {
  using namespace nerf;
  NERF_REQUIRE(n_rays >= 0 && n_coarse >= 3 && n_coarse <= kMaxCoarse && n_fine >= 1 && n_coarse + n_fine <= kMaxTotal,
               "nerf_sample_pdf: n_coarse=%d (3..%d), n_fine=%d, total <= %d", n_coarse, kMaxCoarse, n_fine, kMaxTotal);
  if (n_rays == 0) return NERF_OK;
  NERF_REQUIRE(z_coarse && weights && z_out, "nerf_sample_pdf: NULL pointer");
  int P = 2;
  while (P < n_coarse + n_fine) P <<= 1;
  int64_t blocks = n_rays < 65536 ? n_rays : 65536;
  hipLaunchKernelGGL(sample_pdf_kernel, dim3((int)blocks), dim3(64), 0, as_stream(stream), z_coarse, weights, u, n_rays,
                     n_coarse, n_fine, P, z_out);
  return check_launch("nerf_sample_pdf");
}
