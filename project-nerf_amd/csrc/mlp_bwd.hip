// Backward (dgrad) chain of the fused decoder (autograd of reference src/decoders.py:68-87).
//
// Mirror image of mlp_fwd.hip: the same 512-thread workgroup / 32-samples-per-wave register
// chain, fed by the TRANSPOSED weight stream.  Starting from d(rgb), d(sigma) it walks
//   rgb_layer^T -> view_layer^T -> (feature_layer | sigma_layer)^T -> pts_layers 7..1 ^T
// applying the ReLU bitmasks the forward stashed, and writes every pre-activation gradient
// as a blocked bf16 image: those are the A operands of the weight-gradient kernel
// (mlp_wgrad.hip).  Input gradients of the Fourier codes are never formed (positions are
// not trainable), so pts_layers.0^T and the code columns of layers 4 / view are skipped.
#include "mlp_chain.h"
#include "mlp_stash.h"

namespace nerf {
using namespace plan;

struct BwdArgs {
  const char* packed;
  const float* rgb;       // forward outputs [n,3], [n]
  const float* sigma;
  const float* d_rgb;     // upstream gradients [n,3], [n]
  const float* d_sigma;
  int64_t n, n_pad;
  const uint4* st_mask;   // relu bits written by the forward
  __bf16* dsmall;         // nat [n_pad,16]
  __bf16* dhv;            // blocked [n_pad,128]
  __bf16* dfeat;          // blocked [n_pad,256]
  __bf16* dh;             // 8 x blocked [n_pad,256]
  float* amax;            // 8-bit images: max |output-layer derivative| of the launch (workspace slot read by wgrad)
  const float* amax_src;  // where the dgrad kernel reads it: == amax (bwd_amax_kernel ran) or the caller's value
};

// output-layer derivatives of one sample: sigmoid' and relu' applied to the upstream gradients
__device__ __forceinline__ void out_derivs(const BwdArgs& a, int64_t n, float& g0, float& g1, float& g2, float& gs) {
  const float r0 = a.rgb[n * 3 + 0], r1 = a.rgb[n * 3 + 1], r2 = a.rgb[n * 3 + 2];
  g0 = a.d_rgb[n * 3 + 0] * r0 * (1.0f - r0);
  g1 = a.d_rgb[n * 3 + 1] * r1 * (1.0f - r1);
  g2 = a.d_rgb[n * 3 + 2] * r2 * (1.0f - r2);
  gs = a.sigma[n] > 0.0f ? a.d_sigma[n] : 0.0f;
}

// amax of the dgrad chain's inputs: the e5m2 gradient images are divided by a power of two derived
// from it (mlp_stash.h::grad_image_scale).  Non-negative floats order like their bit patterns.
__global__ void __launch_bounds__(256) bwd_amax_kernel(const BwdArgs a) {
  float m = 0.0f;
  for (int64_t n = blockIdx.x * 256 + threadIdx.x; n < a.n; n += (int64_t)gridDim.x * 256) {
    float g0, g1, g2, gs;
    out_derivs(a, n, g0, g1, g2, gs);
    m = fmaxf(fmaxf(m, fabsf(g0)), fmaxf(fmaxf(fabsf(g1), fabsf(g2)), fabsf(gs)));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    if (m == m && m < 3.0e38f) atomicMax(reinterpret_cast<unsigned*>(a.amax), __builtin_bit_cast(unsigned, m));
  }
}

__global__ void __launch_bounds__(kChainThreads, 2) mlp_bwd_kernel(const BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;

  WeightRing<true> ring;
  ring.init(a.packed + kPackBwdOff, smem + kBiasLdsBytes, wave, lane);
  ring.prologue();
  const char* a_base = nullptr;

  const int64_t n_tiles = a.n_pad / kTileSamples;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const bool more = tile + gridDim.x < n_tiles;
    const int64_t wave_tile = tile * 8 + wave;
    const int64_t n = wave_tile * kWaveSamples + col;
    const bool live = n < a.n;

    // ---- output-layer derivatives: sigmoid' and relu' ----
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, gs = 0.f;
    if (live) out_derivs(a, n, g0, g1, g2, gs);
    bf16x8 small;
#pragma unroll
    for (int j = 0; j < 8; ++j) small[j] = (__bf16)0.0f;
    if (half == 0) {
      small[0] = (__bf16)g0; small[1] = (__bf16)g1; small[2] = (__bf16)g2; small[3] = (__bf16)gs;
    }
    stash_nat(a.dsmall, wave_tile, 1, 0, col, half, small);

    uint4 mask;
    auto load_mask = [&](int layer) { mask = a.st_mask[(tile * 9 + layer) * kChainThreads + tid]; };
    // epilogue: optional relu mask (bits of the layer whose output this gradient belongs to),
    // bf16 operand for the next step, blocked stash for wgrad
    auto grad_epi = [&](bf16x8* out, __bf16* stash, int width, bool masked) {
      return [=, &mask](auto mc, f32x16 acc) {
        constexpr int m = decltype(mc)::value;
        if (masked) {
          const uint32_t words[4] = {mask.x, mask.y, mask.z, mask.w};
          const uint32_t bits = words[m >> 1] >> (16 * (m & 1));
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = (bits >> r) & 1u ? acc[r] : 0.0f;
        }
        acc_to_operand(acc, out[2 * m], out[2 * m + 1]);
        stash_block(stash, wave_tile, width / 32, m, col, half, out[2 * m], out[2 * m + 1]);
      };
    };

    bf16x8 gA[16], gB[16];
    // ---- rgb_layer^T: d(hv_pre) = relu'(hv) * W_rgb^T d(rgb_pre) ----
    {
      bf16x8 in[1];
      in[0] = small;
      if (half == 0) in[0][3] = (__bf16)0.0f;   // column 3 carries d(sigma_pre), not an rgb row
      load_mask(8);
      run_step<true, B_RGB, 1, true>(ring, a_base, more, in, nullptr, half, grad_epi(gA, a.dhv, 128, true));
    }
    // ---- view_layer^T (feature columns only): d(feat) ----
    {
      bf16x8 in[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) in[i] = gA[i];
      run_step<true, B_VIEW, 8, true>(ring, a_base, more, in, nullptr, half, grad_epi(gB, a.dfeat, 256, false));
    }
    // ---- (feature_layer | sigma_layer)^T: d(h7_pre) ----
    {
      bf16x8 in[17];
#pragma unroll
      for (int i = 0; i < 16; ++i) in[i] = gB[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) in[16][j] = (__bf16)0.0f;
      if (half == 0) in[16][0] = (__bf16)gs;
      load_mask(7);
      run_step<true, B_HEAD, 17, true>(ring, a_base, more, in, nullptr, half, grad_epi(gA, a.dh + 7 * a.n_pad * 256, 256, true));
    }
    // ---- pts_layers.7 .. 1 transposed: d(h_{l-1}_pre) ----
    load_mask(6);
    run_step<true, B_PTS7, 16, true>(ring, a_base, more, gA, nullptr, half, grad_epi(gB, a.dh + 6 * a.n_pad * 256, 256, true));
    load_mask(5);
    run_step<true, B_PTS6, 16, true>(ring, a_base, more, gB, nullptr, half, grad_epi(gA, a.dh + 5 * a.n_pad * 256, 256, true));
    load_mask(4);
    run_step<true, B_PTS5, 16, true>(ring, a_base, more, gA, nullptr, half, grad_epi(gB, a.dh + 4 * a.n_pad * 256, 256, true));
    load_mask(3);
    run_step<true, B_PTS4, 16, true>(ring, a_base, more, gB, nullptr, half, grad_epi(gA, a.dh + 3 * a.n_pad * 256, 256, true));
    load_mask(2);
    run_step<true, B_PTS3, 16, true>(ring, a_base, more, gA, nullptr, half, grad_epi(gB, a.dh + 2 * a.n_pad * 256, 256, true));
    load_mask(1);
    run_step<true, B_PTS2, 16, true>(ring, a_base, more, gB, nullptr, half, grad_epi(gA, a.dh + 1 * a.n_pad * 256, 256, true));
    load_mask(0);
    run_step<true, B_PTS1, 16, true>(ring, a_base, more, gA, nullptr, half, grad_epi(gB, a.dh + 0 * a.n_pad * 256, 256, true));
  }
}

}  // namespace nerf

#include <stddef.h>
#include "mlp_stream_asm.h"

namespace nerf {

static_assert(offsetof(BwdArgs, n_pad) == 48 && offsetof(BwdArgs, st_mask) == 56 && offsetof(BwdArgs, dhv) == 72 &&
              offsetof(BwdArgs, dfeat) == 80 && offsetof(BwdArgs, dh) == 88, "kernarg offsets used by the stream asm");

// dgrad chain as one hand-scheduled asm statement per wave and tile (gen_stream_asm.py): this kernel
// forms the output-layer derivatives and hands the pass its two natural-order operands.
// IMG16: bf16 gradient images (stash_block / stash_nat) instead of the e5m2 ones (option stash_fp8).
template <bool IMG16>
__global__ void __launch_bounds__(kChainThreads, 2) mlp_bwd_stream_kernel(const BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;

  WeightRing<true> ring;
  ring.init(a.packed + kPackBwdOff, smem + kBiasLdsBytes, wave, lane);
  ring.template issue<0>(0);
  ring.template issue<1>(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const unsigned ab0 = lds_addr(smem + kBiasLdsBytes) + 16u * lane, ab1 = ab0 + kRingSlotBytes;
  const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds_addr(smem + kBiasLdsBytes) + 1024u * wave);
  const unsigned voff = 1024u * wave + 16u * lane;
  const char* src = a.packed + kPackBwdOff;
  const void* karg = (const void*)__builtin_amdgcn_kernarg_segment_ptr();

  // e5m2 gradient images, divided by a power of two that puts the launch's largest output-layer
  // derivative in [64, 128) (the chain itself runs on unscaled bf16; wgrad multiplies the scale back)
  float gscale = 1.0f;
  if constexpr (!IMG16) {
    set_fp8_saturate();
    const float amax_in = *a.amax_src;
    if (a.amax_src != a.amax && blockIdx.x == 0 && tid == 0) *a.amax = amax_in;   // the wgrad pass reads the workspace slot
    gscale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, grad_image_scale(amax_in))));
  }
  const int64_t n_tiles = a.n_pad / kTileSamples;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t wave_tile = tile * 8 + wave;
    const int64_t n = wave_tile * kWaveSamples + col;
    const bool live = n < a.n;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, gs = 0.f;
    if (live) out_derivs(a, n, g0, g1, g2, gs);
    bf16x8 small, in_rgb, in_sigma;
#pragma unroll
    for (int j = 0; j < 8; ++j) small[j] = in_sigma[j] = (__bf16)0.0f;
    if (half == 0) {
      small[0] = (__bf16)g0; small[1] = (__bf16)g1; small[2] = (__bf16)g2; small[3] = (__bf16)gs;
      in_sigma[0] = (__bf16)gs;
    }
    if constexpr (IMG16) stash_nat(a.dsmall, wave_tile, 1, 0, col, half, small);
    else stash_nat8<true>(reinterpret_cast<char*>(a.dsmall), wave_tile, 1, 0, col, half, small, gscale);
    in_rgb = small;
    in_rgb[3] = (__bf16)0.0f;   // column 3 carries d(sigma_pre), not an rgb row

    const unsigned lane32 = IMG16 ? block_lane_offset(col, half) : block8_lane_offset(col, half);
    const unsigned blk = IMG16 ? 2048u : 1024u;
    const unsigned so8 = (unsigned)wave_tile * (8u * blk) + lane32, so4 = (unsigned)wave_tile * (4u * blk) + lane32;
    const unsigned mo0 = (unsigned)tile * (72u * 512u * 2u) + 2u * tid;
    const bool more = tile + gridDim.x < n_tiles;
    const unsigned mo0n = more ? (unsigned)(tile + gridDim.x) * (72u * 512u * 2u) + 2u * tid : mo0;
    const unsigned first = __builtin_amdgcn_readfirstlane(tile == (int64_t)blockIdx.x ? 1u : 0u);
    if constexpr (IMG16) bwd16_stream_pass(ab0, ab1, in_rgb, in_sigma, src, voff, ldsw, so8, so4, mo0, mo0n, first, karg, gscale);
    else bwd_stream_pass(ab0, ab1, in_rgb, in_sigma, src, voff, ldsw, so8, so4, mo0, mo0n, first, karg, gscale);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace nerf

using namespace nerf;

int nerf_launch_wgrad(const char* stash, const StashLayout& sl, const char* work, const BwdLayout& bl,
                      int64_t n, float* grads, int part, hipStream_t stream, size_t zero_lo, size_t zero_hi);   // mlp_wgrad.hip

extern "C" size_t nerf_mlp_bwd_workspace_bytes(int64_t n) { return n > 0 ? bwd_layout(n).total : 0; }

static int launch_dgrad(const void* packed, const void* stash, const float* rgb, const float* sigma,
                        const float* d_rgb, const float* d_sigma, int64_t n, void* workspace,
                        const float* amax_dev, nerf_stream_t stream) {
  NERF_REQUIRE(n > 0 && n < (int64_t)1 << 31, "nerf_mlp_bwd: n=%lld out of range", (long long)n);
  NERF_REQUIRE(packed && stash && rgb && sigma && d_rgb && d_sigma && workspace, "nerf_mlp_bwd: NULL pointer");
  NERF_REQUIRE(((uintptr_t)packed & 255) == 0 && ((uintptr_t)stash & 255) == 0 && ((uintptr_t)workspace & 255) == 0,
               "nerf_mlp_bwd: packed/stash/workspace must be 256-byte aligned");
  const StashLayout sl = stash_layout(n);
  const BwdLayout bl = bwd_layout(n);
  BwdArgs a{};
  a.packed = static_cast<const char*>(packed);
  a.rgb = rgb; a.sigma = sigma; a.d_rgb = d_rgb; a.d_sigma = d_sigma;
  a.n = n; a.n_pad = bl.n_pad;
  a.st_mask = reinterpret_cast<const uint4*>(static_cast<const char*>(stash) + sl.mask);
  char* w = static_cast<char*>(workspace);
  a.dsmall = reinterpret_cast<__bf16*>(w + bl.dsmall);
  a.dhv = reinterpret_cast<__bf16*>(w + bl.dhv);
  a.dfeat = reinterpret_cast<__bf16*>(w + bl.dfeat);
  a.dh = reinterpret_cast<__bf16*>(w + bl.dh);
  a.amax = reinterpret_cast<float*>(w + bl.amax);
  a.amax_src = amax_dev != nullptr ? amax_dev : a.amax;
  int n_cu = 0;
  if (int rc = device_cu_count(&n_cu); rc != NERF_OK) return rc;
  const bool stream_family = chain_use_stream(n, true);
  const void* kernel = stream_family ? (bl.fp8 ? (const void*)mlp_bwd_stream_kernel<false> : (const void*)mlp_bwd_stream_kernel<true>)
                                     : (const void*)mlp_bwd_kernel;
  if (int rc = ensure_dynamic_lds(kernel, kChainLds, "nerf_mlp_bwd"); rc != NERF_OK) return rc;
  const int64_t tiles = bl.n_pad / kTileSamples;
  if (options().chain_grid > 0 && options().chain_grid < n_cu) n_cu = options().chain_grid;
  const int grid = (int)(tiles < n_cu ? tiles : n_cu);
  if (bl.fp8 && amax_dev == nullptr) {
    if (hipMemsetAsync(a.amax, 0, sizeof(float), as_stream(stream)) != hipSuccess)
      return fail(NERF_ELAUNCH, "nerf_mlp_bwd: memset failed");
    const int64_t want = (n + 1023) / 1024;
    hipLaunchKernelGGL(bwd_amax_kernel, dim3((int)(want < 1024 ? want : 1024)), dim3(256), 0, as_stream(stream), a);
  }
  if (stream_family && bl.fp8)
    hipLaunchKernelGGL(mlp_bwd_stream_kernel<false>, dim3(grid), dim3(kChainThreads), kChainLds, as_stream(stream), a);
  else if (stream_family)
    hipLaunchKernelGGL(mlp_bwd_stream_kernel<true>, dim3(grid), dim3(kChainThreads), kChainLds, as_stream(stream), a);
  else
    hipLaunchKernelGGL(mlp_bwd_kernel, dim3(grid), dim3(kChainThreads), kChainLds, as_stream(stream), a);
  return check_launch("nerf_mlp_bwd (dgrad chain)");
}

static int launch_wgrad(const void* stash, const void* workspace, int64_t n, float* grads_f32, int part,
                        nerf_stream_t stream) {
  NERF_REQUIRE(grads_f32 != nullptr, "nerf_mlp_bwd: grads_f32 is NULL");
  NERF_REQUIRE(part >= 0 && part <= 2, "nerf_mlp_bwd_wgrad_part: part=%d (0 all, 1 late layers, 2 early layers)", part);
  const size_t lo = part == 1 ? (size_t)plan::kW4 : 0, hi = part == 2 ? (size_t)plan::kW4 : (size_t)plan::kParamCount;
  if (n == 0) {
    if (hipMemsetAsync(grads_f32 + lo, 0, sizeof(float) * (hi - lo), as_stream(stream)) != hipSuccess)
      return fail(NERF_ELAUNCH, "nerf_mlp_bwd: memset failed");
    return NERF_OK;
  }
  NERF_REQUIRE(stash && workspace, "nerf_mlp_bwd: NULL pointer");
  // the range is zeroed inside only if the launch flushes with atomics (the partial-tile form overwrites it)
  return nerf_launch_wgrad(static_cast<const char*>(stash), stash_layout(n), static_cast<const char*>(workspace),
                           bwd_layout(n), n, grads_f32, part, as_stream(stream), lo, hi);
}

extern "C" int nerf_mlp_bwd_dgrad(const void* packed, const void* stash, const float* rgb, const float* sigma,
                                  const float* d_rgb, const float* d_sigma, int64_t n, void* workspace,
                                  nerf_stream_t stream) {
  if (n == 0) return NERF_OK;
  return launch_dgrad(packed, stash, rgb, sigma, d_rgb, d_sigma, n, workspace, nullptr, stream);
}

extern "C" int nerf_mlp_bwd_dgrad_ex(const void* packed, const void* stash, const float* rgb, const float* sigma,
                                     const float* d_rgb, const float* d_sigma, int64_t n, void* workspace,
                                     const float* amax_dev, nerf_stream_t stream) {
  if (n == 0) return NERF_OK;
  return launch_dgrad(packed, stash, rgb, sigma, d_rgb, d_sigma, n, workspace, amax_dev, stream);
}

extern "C" int nerf_mlp_bwd_wgrad(const void* stash, const void* workspace, int64_t n, float* grads_f32,
                                  nerf_stream_t stream) {
  return launch_wgrad(stash, workspace, n, grads_f32, 0, stream);
}

extern "C" int64_t nerf_mlp_wgrad_part_split(void) { return plan::kW4; }

extern "C" int nerf_mlp_bwd_wgrad_part(const void* stash, const void* workspace, int64_t n, float* grads_f32, int part,
                                       nerf_stream_t stream) {
  return launch_wgrad(stash, workspace, n, grads_f32, part, stream);
}

extern "C" int nerf_mlp_bwd(const void* packed, const void* stash, const float* rgb, const float* sigma,
                            const float* d_rgb, const float* d_sigma, int64_t n, float* grads_f32,
                            void* workspace, nerf_stream_t stream) {
  NERF_REQUIRE(n >= 0, "nerf_mlp_bwd: n=%lld", (long long)n);
  if (n > 0) {
    const int rc = launch_dgrad(packed, stash, rgb, sigma, d_rgb, d_sigma, n, workspace, nullptr, stream);
    if (rc != NERF_OK) return rc;
  }
  return launch_wgrad(stash, workspace, n, grads_f32, 0, stream);
}
