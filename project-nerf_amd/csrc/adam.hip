// Adam / AdamW over one flat fp32 parameter vector:
// 28 B/param of streaming traffic, float4-vectorised.
#include "common.h"

namespace nerf {

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, float lr, float beta1, float beta2,
                                            float eps, float wd, float inv_bc1, float inv_sqrt_bc2) {
  float pi = p;
  if (wd != 0.0f) pi *= (1.0f - lr * wd);
  const float mi = beta1 * m + (1.0f - beta1) * g;
  const float vi = beta2 * v + (1.0f - beta2) * g * g;
  m = mi;
  v = vi;
  p = pi - (lr * inv_bc1) * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
}

// 16-byte accesses on the 4-aligned body, scalar tail; VEC = false when a base pointer is not 16-byte aligned
template <bool VEC>
__device__ __forceinline__ void adam_sweep(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                           float* __restrict__ v, int64_t n, float gs, float lr, float beta1, float beta2,
                                           float eps, float wd, float inv_bc1, float inv_sqrt_bc2, _Float16* __restrict__ shadow = nullptr) {
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = VEC ? n / 4 : 0;
  for (int64_t i = tid; i < n4; i += stride) {
    f4 pp = reinterpret_cast<f4*>(p)[i], mm = reinterpret_cast<f4*>(m)[i], vv = reinterpret_cast<f4*>(v)[i];
    const f4 gg = reinterpret_cast<const f4*>(g)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pe = pp[e], me = mm[e], ve = vv[e];
      adam_update(pe, gg[e] * gs, me, ve, lr, beta1, beta2, eps, wd, inv_bc1, inv_sqrt_bc2);
      pp[e] = pe; mm[e] = me; vv[e] = ve;
    }
    reinterpret_cast<f4*>(m)[i] = mm;
    reinterpret_cast<f4*>(v)[i] = vv;
    reinterpret_cast<f4*>(p)[i] = pp;
    if (shadow != nullptr) {                  // fp16 copy of the updated parameters (the hash forward gathers from it)
      typedef _Float16 h4 __attribute__((ext_vector_type(4)));
      h4 hh = {(_Float16)pp[0], (_Float16)pp[1], (_Float16)pp[2], (_Float16)pp[3]};
      reinterpret_cast<h4*>(shadow)[i] = hh;
    }
  }
  for (int64_t i = 4 * n4 + tid; i < n; i += stride) {
    adam_update(p[i], g[i] * gs, m[i], v[i], lr, beta1, beta2, eps, wd, inv_bc1, inv_sqrt_bc2);
    if (shadow != nullptr) shadow[i] = (_Float16)p[i];
  }
}

template <bool VEC>
__global__ void __launch_bounds__(256)
adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
            float* __restrict__ v, int64_t n, float lr, float beta1, float beta2, float eps, float wd,
            float inv_bc1, float inv_sqrt_bc2, const float* __restrict__ grad_scale) {
  adam_sweep<VEC>(p, g, m, v, n, grad_scale ? *grad_scale : 1.0f, lr, beta1, beta2, eps, wd, inv_bc1, inv_sqrt_bc2);
}

__host__ inline bool aligned16(const void* a, const void* b, const void* c, const void* d) {
  return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15) == 0;
}

}  // namespace nerf

extern "C" int nerf_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                              int64_t n, int step, float lr, float beta1, float beta2, float eps,
                              float weight_decay, const float* grad_scale_dev, nerf_stream_t stream) {
  NERF_REQUIRE(n >= 0 && step >= 1, "nerf_adam_step: n=%lld step=%d", (long long)n, step);
  if (n == 0) return NERF_OK;
  NERF_REQUIRE(params && grads && exp_avg && exp_avg_sq, "nerf_adam_step: NULL pointer");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  int64_t blocks = (n / 4 + 255) / 256 + 1;
  if (blocks > 2048) blocks = 2048;
  if (nerf::aligned16(params, grads, exp_avg, exp_avg_sq))
    hipLaunchKernelGGL(nerf::adam_kernel<true>, dim3((int)blocks), dim3(256), 0, nerf::as_stream(stream), params,
                       grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay,
                       (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), grad_scale_dev);
  else
    hipLaunchKernelGGL(nerf::adam_kernel<false>, dim3((int)blocks), dim3(256), 0, nerf::as_stream(stream), params,
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

__device__ __forceinline__ float tv_term(float prev, float cur, float next, bool has_prev, bool has_next) {
  float t = 0.0f;
  if (has_prev) t += sgn(cur - prev);          // d/dp_i |p_i - p_{i-1}|
  if (has_next) t -= sgn(next - cur);          // d/dp_i |p_{i+1} - p_i|
  return t;
}

template <bool VEC>
__global__ void __launch_bounds__(256)
tv_normsq_kernel(const float* __restrict__ p, float* __restrict__ g, int64_t n, float tv_scale, float grad_scale,
                 float* __restrict__ normsq, int64_t seg) {
  // seg: the vector is up to four tables of seg elements back to back (Part 4's three deformation grids in one launch): the
  // total variation does not couple the last element of a table with the first of the next
  auto starts = [seg](int64_t e) { return e == 0 || e == seg || e == 2 * seg || e == 3 * seg; };
  // grad_scale (1/world after a summing all-reduce) applies to the DATA gradient only: the TV term is
  // a function of the replicated parameters and must not be divided by the world size
  const bool rewrite = tv_scale != 0.0f || grad_scale != 1.0f;
  float local = 0.0f;
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = VEC ? n / 4 : 0;
  for (int64_t i = tid; i < n4; i += stride) {
    f4 gg = reinterpret_cast<f4*>(g)[i];
    gg[0] *= grad_scale; gg[1] *= grad_scale; gg[2] *= grad_scale; gg[3] *= grad_scale;
    if (tv_scale != 0.0f) {
      const f4 pp = reinterpret_cast<const f4*>(p)[i];
      const int64_t e0 = 4 * i;
      const bool has_prev = !starts(e0), has_next = e0 + 4 < n && !starts(e0 + 4);     // seg is a multiple of 4 on this path
      const float before = has_prev ? p[e0 - 1] : 0.0f, after = has_next ? p[e0 + 4] : 0.0f;
      gg[0] += tv_scale * tv_term(before, pp[0], pp[1], has_prev, true);
      gg[1] += tv_scale * tv_term(pp[0], pp[1], pp[2], true, true);
      gg[2] += tv_scale * tv_term(pp[1], pp[2], pp[3], true, true);
      gg[3] += tv_scale * tv_term(pp[2], pp[3], after, true, has_next);
    }
    if (rewrite) reinterpret_cast<f4*>(g)[i] = gg;
    local += gg[0] * gg[0] + gg[1] * gg[1] + gg[2] * gg[2] + gg[3] * gg[3];
  }
  for (int64_t i = 4 * n4 + tid; i < n; i += stride) {
    float gi = g[i] * grad_scale;
    if (tv_scale != 0.0f) {
      const bool has_prev = !starts(i), has_next = i + 1 < n && !starts(i + 1);
      gi += tv_scale * tv_term(has_prev ? p[i - 1] : 0.0f, p[i], has_next ? p[i + 1] : 0.0f, has_prev, has_next);
    }
    if (rewrite) g[i] = gi;
    local += gi * gi;
  }
  // one partial per workgroup, summed in workgroup order by the last one to finish (common.h::ordered_block_sum): the same
  // bits on every run and on every data-parallel replica -- the clip coefficient derived from it must not differ between ranks
  __shared__ float part[4];
  local = wave_sum(local);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = local;
  __syncthreads();
  const float val[1] = {(part[0] + part[1]) + (part[2] + part[3])};
  float* const out[1] = {normsq};
  ordered_block_sum<1>(val, out, reinterpret_cast<unsigned*>(normsq + 1));
}

template <bool VEC>
__global__ void __launch_bounds__(256)
adamw_clip_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                  int64_t n, float lr, float beta1, float beta2, float eps, float wd, float inv_bc1,
                  float inv_sqrt_bc2, const float* __restrict__ normsq, float max_norm, float extra_scale,
                  _Float16* __restrict__ shadow) {
  float gs = extra_scale;
  if (normsq != nullptr && max_norm > 0.0f) {
    const float coef = max_norm / (sqrtf(*normsq * extra_scale * extra_scale) + 1e-6f);   // torch clip_grad_norm_
    gs *= coef < 1.0f ? coef : 1.0f;
  }
  adam_sweep<VEC>(p, g, m, v, n, gs, lr, beta1, beta2, eps, wd, inv_bc1, inv_sqrt_bc2, shadow);
}

// ---- the same two passes WITHOUT rewriting the gradient (38.5 instead of 42 bytes per parameter) --------------------------
// pass 1 leaves, instead of g + TV term (4 bytes written, 4 read back), the SIGNS it is made of: code_i = 1 + sign(p[i+1] - p[i])
// in two bits (0 behind the last element of a table), four elements per byte; the TV term of element i is
// tv_scale * (s[i-1] - s[i]).  Pass 2 rebuilds g * grad_scale + TV term from the codes -- from the OLD parameters' signs, whatever
// its neighbours have already written.
__device__ __forceinline__ int tv_code_at(const uint8_t* __restrict__ codes, int64_t i) {
  return i < 0 ? 0 : (int)((codes[i >> 2] >> (2 * (int)(i & 3))) & 3u) - 1;
}

// generic form (any alignment, any n): one chunk of four elements per thread and round
__global__ void __launch_bounds__(256)
tv_normsq_codes_kernel(const float* __restrict__ p, const float* __restrict__ g, int64_t n, float tv_scale, float grad_scale,
                       float* __restrict__ normsq, int64_t seg, uint8_t* __restrict__ codes, int accumulate) {
  auto starts = [seg](int64_t e) { return e == 0 || e == seg || e == 2 * seg || e == 3 * seg; };
  float local = 0.0f;
  const int64_t n4 = (n + 3) / 4;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e0 = 4 * q;
    float pp[6], gg[4];                        // pp[0] = p[e0-1] ... pp[5] = p[e0+4]
#pragma unroll
    for (int e = 0; e < 4; ++e) { pp[1 + e] = e0 + e < n ? p[e0 + e] : 0.0f; gg[e] = e0 + e < n ? g[e0 + e] : 0.0f; }
    if (tv_scale != 0.0f) {
      pp[0] = e0 > 0 ? p[e0 - 1] : 0.0f;
      pp[5] = e0 + 4 < n ? p[e0 + 4] : 0.0f;
      unsigned byte = 0;
      int s_prev = !starts(e0) ? (int)sgn(pp[1] - pp[0]) : 0;      // s[e0-1]: 0 across a seam
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int64_t i = e0 + e;
        const int s_cur = (i + 1 < n && !starts(i + 1)) ? (int)sgn(pp[2 + e] - pp[1 + e]) : 0;
        if (i < n) {
          const float gi = gg[e] * grad_scale + tv_scale * (float)(s_prev - s_cur);
          local += gi * gi;
          byte |= (unsigned)(s_cur + 1) << (2 * e);
        }
        s_prev = s_cur;
      }
      codes[q] = (uint8_t)byte;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float gi = gg[e] * grad_scale; local += gi * gi; }     // elements past n are zero
    }
  }
  __shared__ float part[4];
  local = wave_sum(local);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = local;
  __syncthreads();
  const float val[1] = {(part[0] + part[1]) + (part[2] + part[3])};
  float* const out[1] = {normsq};
  ordered_block_sum<1>(val, out, reinterpret_cast<unsigned*>(normsq + 1), accumulate != 0);
}

// the tables' form: 16-byte aligned, n a multiple of 4 and below 2^31.  kU chunks per thread and round, every load issued before the
// first is used and NO branch around a load (the neighbours p[e0-1], p[e0+4] come from clamped addresses; a table seam can only lie on
// a chunk's border since seg is a multiple of 4: two 32-bit seam tests per chunk).  The generic form above spent its time on 64-bit
// seam tests per element and on one chunk's two loads in flight per thread: 2.4 TB/s.
// 1024 threads per workgroup, at most one workgroup per CU: every workgroup ends with a same-address ticket atomic, and those
// retire one after the other (~20 ns each: 1024 workgroups of 256 spent 20 us of an 80-us launch queueing there; 4096: 60 us)
constexpr int kTvFastThreads = 1024;
template <bool TV>
__global__ void __launch_bounds__(kTvFastThreads)
tv_normsq_codes_fast_kernel(const float* __restrict__ p, const float* __restrict__ g, int n4, float tv_scale, float grad_scale,
                            float* __restrict__ normsq, int seg, uint8_t* __restrict__ codes, int halo, int accumulate) {
  // halo (a PIECE of one table: the sharded optimiser's slice): bit 0 -- p[-1] belongs to the same table, bit 1 -- p[n] does; the
  // piece's first TV term then reaches back to p[-1] (its sign is also left in codes[-1] for the AdamW pass), its last one on to p[n]
  constexpr int kU = 4;
  const f4* __restrict__ p4 = reinterpret_cast<const f4*>(p);
  const f4* __restrict__ g4 = reinterpret_cast<const f4*>(g);
  const int span = gridDim.x * blockDim.x, n = 4 * n4;
  float local = 0.0f;
  for (int q0 = blockIdx.x * blockDim.x + threadIdx.x; q0 < n4; q0 += kU * span) {
    f4 a[kU], b[kU];
    float lo[kU], hi[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int q = min(q0 + u * span, n4 - 1);           // past the end: the last chunk again (masked below)
      a[u] = p4[q];
      b[u] = g4[q];
      if (TV) {
        lo[u] = p[max(4 * q - 1, (halo & 1) ? -1 : 0)];
        hi[u] = p[min(4 * q + 4, (halo & 2) ? n : n - 1)];
      }
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int q = q0 + u * span, e0 = 4 * q;
      if (q >= n4) break;
      float g0, g1, g2, g3;
      if (TV) {
        const bool first = (e0 == 0 && !(halo & 1)) || e0 == seg || e0 == 2 * seg || e0 == 3 * seg;
        const int e4 = e0 + 4;
        const bool last = (e4 == n && !(halo & 2)) || (e4 != n && (e4 == seg || e4 == 2 * seg || e4 == 3 * seg));
        const int s_in = first ? 0 : (int)sgn(a[u][0] - lo[u]), s0 = (int)sgn(a[u][1] - a[u][0]), s1 = (int)sgn(a[u][2] - a[u][1]),
                  s2 = (int)sgn(a[u][3] - a[u][2]), s3 = last ? 0 : (int)sgn(hi[u] - a[u][3]);
        g0 = b[u][0] * grad_scale + tv_scale * (float)(s_in - s0);
        g1 = b[u][1] * grad_scale + tv_scale * (float)(s0 - s1);
        g2 = b[u][2] * grad_scale + tv_scale * (float)(s1 - s2);
        g3 = b[u][3] * grad_scale + tv_scale * (float)(s2 - s3);
        codes[q] = (uint8_t)((s0 + 1) | ((s1 + 1) << 2) | ((s2 + 1) << 4) | ((s3 + 1) << 6));
        if (q == 0 && (halo & 1)) codes[-1] = (uint8_t)((s_in + 1) << 6);      // the sign the AdamW pass reads as the piece's s[-1]
      } else {
        g0 = b[u][0] * grad_scale; g1 = b[u][1] * grad_scale; g2 = b[u][2] * grad_scale; g3 = b[u][3] * grad_scale;
      }
      local += (g0 * g0 + g1 * g1) + (g2 * g2 + g3 * g3);
    }
  }
  __shared__ float part[kTvFastThreads / 64];
  local = wave_sum(local);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = local;
  __syncthreads();
  float sum = 0.0f;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < kTvFastThreads / 64; ++w) sum += part[w];
  }
  const float val[1] = {sum};
  float* const out[1] = {normsq};
  ordered_block_sum<1>(val, out, reinterpret_cast<unsigned*>(normsq + 1), accumulate != 0);
}

// One small parameter group (a tiny MLP's few thousand weights) in ONE launch: squared norm of the scaled gradient, clip coefficient,
// AdamW -- instead of a zeroing launch, a norm launch and an AdamW launch of ~4.5 us each for 45 KB of data.  EVERY workgroup sums
// the whole gradient itself (threads in a fixed stride, waves in order: the same bits in every workgroup and every run -- the vector
// sits in L2) and then steps its own kSmallOptThreads elements: no exchange between workgroups, one element per thread in the update.
// zero_grads: the gradient is left zeroed for the next step's accumulating backward (not while other workgroups may still read it:
// only with one workgroup, i.e. n <= kSmallOptThreads -- the launcher falls back to a memset otherwise).
constexpr int kSmallOptThreads = 1024;
__global__ void __launch_bounds__(kSmallOptThreads)
clip_adamw_small_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int n, float lr,
                        float beta1, float beta2, float eps, float wd, float inv_bc1, float inv_sqrt_bc2, float max_norm, float grad_scale,
                        float* __restrict__ normsq_out, int zero_grads) {
  float local = 0.0f;
  for (int i = threadIdx.x; i < n; i += kSmallOptThreads) { const float gi = g[i] * grad_scale; local += gi * gi; }
  __shared__ float part[kSmallOptThreads / 64];
  __shared__ float total;
  local = wave_sum(local);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    float sum = 0.0f;
#pragma unroll
    for (int w = 0; w < kSmallOptThreads / 64; ++w) sum += part[w];
    total = sum;
    if (normsq_out != nullptr && blockIdx.x == 0) *normsq_out = sum;
  }
  __syncthreads();
  float clip = 1.0f;
  if (max_norm > 0.0f) {
    const float coef = max_norm / (sqrtf(total) + 1e-6f);         // torch clip_grad_norm_
    clip = coef < 1.0f ? coef : 1.0f;
  }
  const int i = blockIdx.x * kSmallOptThreads + threadIdx.x;
  if (i < n) {
    float pp = p[i], mm = m[i], vv = v[i];
    adam_update(pp, g[i] * grad_scale * clip, mm, vv, lr, beta1, beta2, eps, wd, inv_bc1, inv_sqrt_bc2);
    p[i] = pp; m[i] = mm; v[i] = vv;
    if (zero_grads) g[i] = 0.0f;
  }
}

__global__ void __launch_bounds__(256)
adamw_clip_tv_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                     float lr, float beta1, float beta2, float eps, float wd, float inv_bc1, float inv_sqrt_bc2,
                     const float* __restrict__ normsq, float max_norm, float grad_scale, const uint8_t* __restrict__ codes,
                     int64_t tv_split, float tv_scale_lo, float tv_scale_hi, _Float16* __restrict__ shadow, int vec,
                     int64_t lr_split, float lr_hi, int halo_lo) {
  float clip = 1.0f;
  if (normsq != nullptr && max_norm > 0.0f) {
    const float coef = max_norm / (sqrtf(*normsq) + 1e-6f);       // torch clip_grad_norm_; normsq is of the scaled gradient + TV term
    clip = coef < 1.0f ? coef : 1.0f;
  }
  const int64_t n4 = (n + 3) / 4;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e0 = 4 * q;
    const bool whole = vec && e0 + 4 <= n;
    float pp[4], gg[4], mm[4], vv[4];
    if (whole) {
      const f4 a = reinterpret_cast<const f4*>(p)[q], b = reinterpret_cast<const f4*>(g)[q], c = reinterpret_cast<const f4*>(m)[q],
               d = reinterpret_cast<const f4*>(v)[q];
#pragma unroll
      for (int e = 0; e < 4; ++e) { pp[e] = a[e]; gg[e] = b[e]; mm[e] = c[e]; vv[e] = d[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool in = e0 + e < n;
        pp[e] = in ? p[e0 + e] : 0.0f; gg[e] = in ? g[e0 + e] : 0.0f; mm[e] = in ? m[e0 + e] : 0.0f; vv[e] = in ? v[e0 + e] : 0.0f;
      }
    }
    int s_prev = 0;
    unsigned byte = 0x55u;                       // all codes 1 = sign 0
    if (codes != nullptr) {
      byte = codes[q];
      s_prev = (q > 0 || halo_lo) ? (int)(codes[q - 1] >> 6) - 1 : 0;      // halo_lo: a piece inside a table, codes[-1] holds s[-1]
    }
    const float tv_scale = e0 < tv_split ? tv_scale_lo : tv_scale_hi;       // the splits are multiples of 4: one choice per chunk
    const float lr_q = e0 < lr_split ? lr : lr_hi;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int s_cur = (int)((byte >> (2 * e)) & 3u) - 1;
      const float gi = (gg[e] * grad_scale + tv_scale * (float)(s_prev - s_cur)) * clip;
      adam_update(pp[e], gi, mm[e], vv[e], lr_q, beta1, beta2, eps, wd, inv_bc1, inv_sqrt_bc2);
      s_prev = s_cur;
    }
    if (whole) {
      f4 a = {pp[0], pp[1], pp[2], pp[3]}, c = {mm[0], mm[1], mm[2], mm[3]}, d = {vv[0], vv[1], vv[2], vv[3]};
      reinterpret_cast<f4*>(m)[q] = c;
      reinterpret_cast<f4*>(v)[q] = d;
      reinterpret_cast<f4*>(p)[q] = a;
      if (shadow != nullptr) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        h4 hh = {(_Float16)pp[0], (_Float16)pp[1], (_Float16)pp[2], (_Float16)pp[3]};
        reinterpret_cast<h4*>(shadow)[q] = hh;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (e0 + e < n) {
          p[e0 + e] = pp[e]; m[e0 + e] = mm[e]; v[e0 + e] = vv[e];
          if (shadow != nullptr) shadow[e0 + e] = (_Float16)pp[e];
        }
    }
  }
}

__global__ void __launch_bounds__(256) f32_to_f16_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (_Float16)src[i];
}

}  // namespace nerf

static int tv_normsq_impl(const float* params, float* grads, int64_t n, float tv_weight, float grad_scale,
                          float* normsq_dev, bool zero_first, nerf_stream_t stream, int n_tables = 1);

extern "C" int nerf_tv_normsq(const float* params, float* grads, int64_t n, float tv_weight, float grad_scale,
                              float* normsq_dev, nerf_stream_t stream) {
  return tv_normsq_impl(params, grads, n, tv_weight, grad_scale, normsq_dev, true, stream);
}

extern "C" int nerf_tv_normsq_accum(const float* params, float* grads, int64_t n, float tv_weight, float grad_scale,
                                    float* normsq_dev, nerf_stream_t stream) {
  return tv_normsq_impl(params, grads, n, tv_weight, grad_scale, normsq_dev, false, stream);
}

extern "C" int nerf_tv_normsq_accum_tables(const float* params, float* grads, int64_t n, int n_tables, float tv_weight,
                                           float grad_scale, float* normsq_dev, nerf_stream_t stream) {
  return tv_normsq_impl(params, grads, n, tv_weight, grad_scale, normsq_dev, false, stream, n_tables);
}

static int tv_normsq_impl(const float* params, float* grads, int64_t n, float tv_weight, float grad_scale,
                          float* normsq_dev, bool zero_first, nerf_stream_t stream, int n_tables) {
  NERF_REQUIRE(n >= 0 && normsq_dev && n_tables >= 1 && n_tables <= 4 && n % n_tables == 0, "nerf_tv_normsq: bad arguments");
  const int64_t seg = n / n_tables;
  NERF_REQUIRE(n_tables == 1 || seg % 4 == 0, "nerf_tv_normsq_accum_tables: %lld elements per table (a multiple of 4)", (long long)seg);
  // normsq_dev: NERF_NORMSQ_WS_FLOATS floats -- [0] the squared norm, then the ordered sum's workspace (tickets, one partial per workgroup)
  if (zero_first && hipMemsetAsync(normsq_dev, 0, sizeof(float) * (1 + nerf::kOrderedSumTickets), nerf::as_stream(stream)) != hipSuccess)   // the norm and the tickets
    return nerf::fail(NERF_ELAUNCH, "nerf_tv_normsq: memset failed");
  if (n == 0) return NERF_OK;
  NERF_REQUIRE(params && grads, "nerf_tv_normsq: NULL pointer");
  const float tv_scale = seg > 1 ? tv_weight / (float)(seg - 1) : 0.0f;      // d/dp of mean|p[1:] - p[:-1]| * w, per table
  int64_t blocks = (n / 4 + 255) / 256 + 1;
  if (blocks > 1024) blocks = 1024;      // measured: 512 +6 %, 256 +50 %, 2048 +15 % (one same-address atomic per workgroup against HBM streams in flight)
  static_assert(NERF_NORMSQ_WS_FLOATS >= 1 + nerf::kOrderedSumTickets + 4096, "the norm, the tickets, one partial per workgroup");
  if ((((uintptr_t)params | (uintptr_t)grads) & 15) == 0)
    hipLaunchKernelGGL(nerf::tv_normsq_kernel<true>, dim3((int)blocks), dim3(256), 0, nerf::as_stream(stream), params, grads, n,
                       tv_scale, grad_scale, normsq_dev, seg);
  else
    hipLaunchKernelGGL(nerf::tv_normsq_kernel<false>, dim3((int)blocks), dim3(256), 0, nerf::as_stream(stream), params, grads, n,
                       tv_scale, grad_scale, normsq_dev, seg);
  return nerf::check_launch("nerf_tv_normsq");
}

static int adamw_clip_impl(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                           int step, float lr, float beta1, float beta2, float eps, float weight_decay,
                           const float* normsq_dev, float max_norm, float grad_scale, void* shadow_f16, nerf_stream_t stream);

extern "C" int nerf_adamw_clip_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                    int step, float lr, float beta1, float beta2, float eps, float weight_decay,
                                    const float* normsq_dev, float max_norm, float grad_scale, nerf_stream_t stream) {
  return adamw_clip_impl(params, grads, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, weight_decay, normsq_dev, max_norm,
                         grad_scale, nullptr, stream);
}

extern "C" int nerf_adamw_clip_step_shadow(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                           int step, float lr, float beta1, float beta2, float eps, float weight_decay,
                                           const float* normsq_dev, float max_norm, float grad_scale, void* params_f16_out,
                                           nerf_stream_t stream) {
  NERF_REQUIRE(params_f16_out != nullptr && ((uintptr_t)params_f16_out & 7) == 0, "nerf_adamw_clip_step_shadow: params_f16_out NULL or unaligned");
  return adamw_clip_impl(params, grads, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, weight_decay, normsq_dev, max_norm,
                         grad_scale, params_f16_out, stream);
}

extern "C" int nerf_f32_to_f16(const float* src, void* dst_f16, int64_t n, nerf_stream_t stream) {
  NERF_REQUIRE(n >= 0, "nerf_f32_to_f16: n=%lld", (long long)n);
  if (n == 0) return NERF_OK;
  NERF_REQUIRE(src && dst_f16, "nerf_f32_to_f16: NULL pointer");
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(nerf::f32_to_f16_kernel, dim3((int)blocks), dim3(256), 0, nerf::as_stream(stream), src, static_cast<_Float16*>(dst_f16), n);
  return nerf::check_launch("nerf_f32_to_f16");
}

static int adamw_clip_impl(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                           int step, float lr, float beta1, float beta2, float eps, float weight_decay,
                           const float* normsq_dev, float max_norm, float grad_scale, void* shadow_f16, nerf_stream_t stream) {
  _Float16* shadow = static_cast<_Float16*>(shadow_f16);
  NERF_REQUIRE(n >= 0 && step >= 1, "nerf_adamw_clip_step: n=%lld step=%d", (long long)n, step);
  if (n == 0) return NERF_OK;
  NERF_REQUIRE(params && grads && exp_avg && exp_avg_sq, "nerf_adamw_clip_step: NULL pointer");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  int64_t blocks = (n / 4 + 255) / 256 + 1;
  if (blocks > 4096) blocks = 4096;
  if (nerf::aligned16(params, grads, exp_avg, exp_avg_sq))
    hipLaunchKernelGGL(nerf::adamw_clip_kernel<true>, dim3((int)blocks), dim3(256), 0, nerf::as_stream(stream), params, grads,
                       exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, (float)(1.0 / bc1),
                       (float)(1.0 / sqrt(bc2)), normsq_dev, max_norm, grad_scale, shadow);
  else
    hipLaunchKernelGGL(nerf::adamw_clip_kernel<false>, dim3((int)blocks), dim3(256), 0, nerf::as_stream(stream), params, grads,
                       exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, (float)(1.0 / bc1),
                       (float)(1.0 / sqrt(bc2)), normsq_dev, max_norm, grad_scale, shadow);
  return nerf::check_launch("nerf_adamw_clip_step");
}

// ---- TV + norm and clip + AdamW without the gradient rewrite (see tv_normsq_codes_kernel) ----
extern "C" size_t nerf_tv_codes_bytes(int64_t n) { return n > 0 ? (size_t)((n + 3) / 4) : 0; }

extern "C" int nerf_tv_normsq_codes(const float* params, const float* grads, int64_t n, int n_tables, float tv_weight, float grad_scale,
                                    float* normsq_dev, int accumulate, void* tv_codes, nerf_stream_t stream) {
  NERF_REQUIRE(n >= 0 && normsq_dev && n_tables >= 1 && n_tables <= 4 && n % n_tables == 0, "nerf_tv_normsq_codes: bad arguments");
  if (n == 0) {        // nothing to add; a storing call still leaves the norm defined
    if (!accumulate && hipMemsetAsync(normsq_dev, 0, sizeof(float), nerf::as_stream(stream)) != hipSuccess)
      return nerf::fail(NERF_ELAUNCH, "nerf_tv_normsq_codes: memset failed");
    return NERF_OK;
  }
  const int64_t seg = n / n_tables;
  NERF_REQUIRE(n_tables == 1 || seg % 4 == 0, "nerf_tv_normsq_codes: %lld elements per table (a multiple of 4)", (long long)seg);
  NERF_REQUIRE(params && grads && (tv_weight == 0.0f || tv_codes), "nerf_tv_normsq_codes: NULL pointer");
  const float tv_scale = seg > 1 ? tv_weight / (float)(seg - 1) : 0.0f;
  int64_t blocks = ((n + 3) / 4 + 255) / 256;
  const int64_t cap = 1024;   // generic form; partials: NERF_NORMSQ_WS_FLOATS
  if (blocks > cap) blocks = cap;
  const bool fast = (((uintptr_t)params | (uintptr_t)grads) & 15) == 0 && n % 4 == 0 && n < ((int64_t)1 << 31) - 16;
  uint8_t* codes = static_cast<uint8_t*>(tv_codes);
  int n_cu = 256;
  (void)nerf::device_cu_count(&n_cu);
  int64_t fblocks = (n / 4 + 4 * nerf::kTvFastThreads - 1) / (4 * nerf::kTvFastThreads);          // four chunks per thread and round
  const int64_t fcap = nerf::options().tv_blocks > 0 && nerf::options().tv_blocks <= 4096 ? nerf::options().tv_blocks : n_cu;
  if (fblocks > fcap) fblocks = fcap;
  if (fblocks < 1) fblocks = 1;
  if (fast && tv_scale != 0.0f)
    hipLaunchKernelGGL(nerf::tv_normsq_codes_fast_kernel<true>, dim3((int)fblocks), dim3(nerf::kTvFastThreads), 0, nerf::as_stream(stream),
                       params, grads, (int)(n / 4), tv_scale, grad_scale, normsq_dev, (int)seg, codes, 0, accumulate);
  else if (fast)
    hipLaunchKernelGGL(nerf::tv_normsq_codes_fast_kernel<false>, dim3((int)fblocks), dim3(nerf::kTvFastThreads), 0, nerf::as_stream(stream),
                       params, grads, (int)(n / 4), 0.0f, grad_scale, normsq_dev, (int)seg, codes, 0, accumulate);
  else
    hipLaunchKernelGGL(nerf::tv_normsq_codes_kernel, dim3((int)blocks), dim3(256), 0, nerf::as_stream(stream), params, grads, n, tv_scale,
                       grad_scale, normsq_dev, seg, codes, accumulate);
  return nerf::check_launch("nerf_tv_normsq_codes");
}

extern "C" int nerf_adamw_clip_step_tv(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int step, float lr,
                                       float beta1, float beta2, float eps, float weight_decay, const float* normsq_dev, float max_norm,
                                       float grad_scale, const void* tv_codes, int64_t tv_split, float tv_weight_lo, int64_t seg_lo,
                                       float tv_weight_hi, int64_t seg_hi, int64_t lr_split, float lr_hi, void* params_f16_out,
                                       nerf_stream_t stream) {
  NERF_REQUIRE(n >= 0 && step >= 1 && tv_split >= 0 && seg_lo >= 0 && seg_hi >= 0, "nerf_adamw_clip_step_tv: n=%lld step=%d", (long long)n, step);
  NERF_REQUIRE((tv_split % 4 == 0 || tv_split >= n) && (lr_split % 4 == 0 || lr_split >= n),
               "nerf_adamw_clip_step_tv: tv_split / lr_split must be multiples of 4 (or past the end)");
  if (n == 0) return NERF_OK;
  NERF_REQUIRE(params && grads && exp_avg && exp_avg_sq, "nerf_adamw_clip_step_tv: NULL pointer");
  NERF_REQUIRE(params_f16_out == nullptr || ((uintptr_t)params_f16_out & 7) == 0, "nerf_adamw_clip_step_tv: params_f16_out unaligned");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const float lo = (tv_codes && seg_lo > 1) ? tv_weight_lo / (float)(seg_lo - 1) : 0.0f;
  const float hi = (tv_codes && seg_hi > 1) ? tv_weight_hi / (float)(seg_hi - 1) : 0.0f;
  int64_t blocks = ((n + 3) / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  const int vec = nerf::aligned16(params, grads, exp_avg, exp_avg_sq);
  hipLaunchKernelGGL(nerf::adamw_clip_tv_kernel, dim3((int)blocks), dim3(256), 0, nerf::as_stream(stream), params, grads, exp_avg, exp_avg_sq,
                     n, lr, beta1, beta2, eps, weight_decay, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), normsq_dev, max_norm, grad_scale,
                     static_cast<const uint8_t*>(tv_codes), tv_split, lo, hi, static_cast<_Float16*>(params_f16_out), vec,
                     lr_split <= 0 ? n : lr_split, lr_hi, 0);
  return nerf::check_launch("nerf_adamw_clip_step_tv");
}

// ---- the same two passes on a PIECE [params, params + n) of ONE table of table_elems elements (the sharded optimiser of the
// data-parallel engines: every rank steps its slice of the flat table buffer).  halo bit 0: params[-1] belongs to the same table and
// holds its current value, bit 1: params[n] does -- the TV terms at the piece's ends then equal the whole-table pass.  tv_codes points
// at the piece's first code byte inside a buffer that has at least one byte before it (codes[-1] receives the sign of
// params[0] - params[-1]).  n a multiple of 4, 16-byte aligned pointers.
extern "C" int nerf_tv_normsq_codes_piece(const float* params, const float* grads, int64_t n, int64_t table_elems, int halo, float tv_weight,
                                          float grad_scale, float* normsq_dev, int accumulate, void* tv_codes, nerf_stream_t stream) {
  NERF_REQUIRE(n >= 0 && normsq_dev && table_elems >= n && halo >= 0 && halo <= 3, "nerf_tv_normsq_codes_piece: bad arguments");
  if (n == 0) {
    if (!accumulate && hipMemsetAsync(normsq_dev, 0, sizeof(float), nerf::as_stream(stream)) != hipSuccess)
      return nerf::fail(NERF_ELAUNCH, "nerf_tv_normsq_codes_piece: memset failed");
    return NERF_OK;
  }
  NERF_REQUIRE(params && grads && (tv_weight == 0.0f || tv_codes), "nerf_tv_normsq_codes_piece: NULL pointer");
  NERF_REQUIRE((((uintptr_t)params | (uintptr_t)grads) & 15) == 0 && n % 4 == 0 && n < ((int64_t)1 << 31) - 16,
               "nerf_tv_normsq_codes_piece: 16-byte aligned pointers and n a multiple of 4");
  const float tv_scale = table_elems > 1 ? tv_weight / (float)(table_elems - 1) : 0.0f;
  int n_cu = 256;
  (void)nerf::device_cu_count(&n_cu);
  int64_t fblocks = (n / 4 + 4 * nerf::kTvFastThreads - 1) / (4 * nerf::kTvFastThreads);
  if (fblocks > n_cu) fblocks = n_cu;
  if (tv_scale != 0.0f)
    hipLaunchKernelGGL(nerf::tv_normsq_codes_fast_kernel<true>, dim3((int)fblocks), dim3(nerf::kTvFastThreads), 0, nerf::as_stream(stream),
                       params, grads, (int)(n / 4), tv_scale, grad_scale, normsq_dev, (int)n, static_cast<uint8_t*>(tv_codes), halo, accumulate);
  else
    hipLaunchKernelGGL(nerf::tv_normsq_codes_fast_kernel<false>, dim3((int)fblocks), dim3(nerf::kTvFastThreads), 0, nerf::as_stream(stream),
                       params, grads, (int)(n / 4), 0.0f, grad_scale, normsq_dev, (int)n, static_cast<uint8_t*>(tv_codes), 0, accumulate);
  return nerf::check_launch("nerf_tv_normsq_codes_piece");
}

extern "C" int nerf_adamw_clip_step_tv_piece(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int step,
                                             float lr, float beta1, float beta2, float eps, float weight_decay, const float* normsq_dev,
                                             float max_norm, float grad_scale, const void* tv_codes, float tv_weight, int64_t table_elems,
                                             int halo_lo, void* params_f16_out, nerf_stream_t stream) {
  NERF_REQUIRE(n >= 0 && step >= 1 && table_elems >= n, "nerf_adamw_clip_step_tv_piece: n=%lld step=%d", (long long)n, step);
  if (n == 0) return NERF_OK;
  NERF_REQUIRE(params && grads && exp_avg && exp_avg_sq, "nerf_adamw_clip_step_tv_piece: NULL pointer");
  NERF_REQUIRE(params_f16_out == nullptr || ((uintptr_t)params_f16_out & 7) == 0, "nerf_adamw_clip_step_tv_piece: params_f16_out unaligned");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const float sc = (tv_codes && table_elems > 1) ? tv_weight / (float)(table_elems - 1) : 0.0f;
  int64_t blocks = ((n + 3) / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  const int vec = nerf::aligned16(params, grads, exp_avg, exp_avg_sq);
  hipLaunchKernelGGL(nerf::adamw_clip_tv_kernel, dim3((int)blocks), dim3(256), 0, nerf::as_stream(stream), params, grads, exp_avg, exp_avg_sq,
                     n, lr, beta1, beta2, eps, weight_decay, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), normsq_dev, max_norm, grad_scale,
                     static_cast<const uint8_t*>(tv_codes), n, sc, sc, static_cast<_Float16*>(params_f16_out), vec, n, lr,
                     (tv_codes && halo_lo) ? 1 : 0);
  return nerf::check_launch("nerf_adamw_clip_step_tv_piece");
}

// ---- one small group (n <= 65536: a tiny MLP's weights) in ONE launch (n / 1024 workgroups, each summing the whole gradient itself):
// squared norm of grads * grad_scale (fixed order), global-norm
// clip, AdamW (clip_grad_norm_ + AdamW.step() of reference run.py:624-629 for the decoder group).  normsq_out (nullable) receives
// the squared norm; zero_grads: the gradient vector is left zeroed (the next backward accumulates into it).
extern "C" int nerf_clip_adamw_small(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int step, float lr,
                                     float beta1, float beta2, float eps, float weight_decay, float max_norm, float grad_scale,
                                     float* normsq_out, int zero_grads, nerf_stream_t stream) {
  NERF_REQUIRE(n >= 0 && n <= 65536 && step >= 1, "nerf_clip_adamw_small: n=%lld (at most 65536) step=%d", (long long)n, step);
  if (n == 0) return NERF_OK;
  NERF_REQUIRE(params && grads && exp_avg && exp_avg_sq, "nerf_clip_adamw_small: NULL pointer");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const int blocks = (int)((n + nerf::kSmallOptThreads - 1) / nerf::kSmallOptThreads);
  hipLaunchKernelGGL(nerf::clip_adamw_small_kernel, dim3(blocks), dim3(nerf::kSmallOptThreads), 0, nerf::as_stream(stream), params, grads,
                     exp_avg, exp_avg_sq, (int)n, lr, beta1, beta2, eps, weight_decay, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), max_norm,
                     grad_scale, normsq_out, (zero_grads && blocks == 1) ? 1 : 0);
  if (int rc = nerf::check_launch("nerf_clip_adamw_small"); rc != NERF_OK) return rc;
  if (zero_grads && blocks > 1 && hipMemsetAsync(grads, 0, sizeof(float) * (size_t)n, nerf::as_stream(stream)) != hipSuccess)
    return nerf::fail(NERF_ELAUNCH, "nerf_clip_adamw_small: memset failed");
  return NERF_OK;
}
