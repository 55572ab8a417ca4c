// Instant-NGP decoder: two bias-free tiny MLPs on bf16 MFMA.
// Replaces tinycudann's FullyFusedMLP networks behind InstantNeRFDecoder (reference
// src/decoders.py:90-162):
//   sigma-net  32 -> 64 (relu) -> 16 (linear);  sigma = softplus(h[0] - 5)
//   colour-net [h (16) | dir code (27)] -> 64 (relu) -> 64 (relu) -> 3 (sigmoid)
// PARITY UNPINNED against tinycudann itself (source and binary absent); the checker is the build's
// own CPU restatement (oracle/nerf_oracle.py::instant_decoder).
//
// Same register chain as the 8x256 decoder (mlp_chain.h): 32 samples per wave on the MFMA
// column, activations carried as accumulator tiles -> bf16 B fragments.  All 26 (forward) /
// 20 (transposed) weight fragments stay resident in LDS for the whole launch.
// Parameter vector (fp32, [out,in] row-major, bias-free):
//   sigma_net : W1 [64,32] | W2 [16,64]                       = 3072
//   color_net : W1 [64,48] (cols 43..47 unused) | W2 [64,64] | W3 [16,64] (rows 3..15 unused) = 8192
#include "mlp_chain.h"
#include "mlp_wgrad.h"

namespace nerf {

constexpr int kSW1 = 0, kSW2 = 2048, kCW1 = 3072, kCW2 = 6144, kCW3 = 10240, kIParams = 11264;
constexpr int kIFwdFrags = 26, kIBwdFrags = 20;
constexpr size_t kIPackBytes = (size_t)(kIFwdFrags + kIBwdFrags) * 1024;
constexpr int kIThreads = 256, kITile = 128;

// fragment index of (step, m-tile, k-step); forward steps 0..4 = S1 S2 C1 C2 C3, backward 5..9 = C3t C2t C1t S2t S1t
struct IStep { int mt, ks_acc, ks_nat, frag0; };
constexpr IStep istep(int s) {
  switch (s) {
    case 0: return {2, 0, 2, 0};     // S1: hash(32, nat) -> 64
    case 1: return {1, 4, 0, 4};     // S2: 64 -> 16
    case 2: return {2, 1, 2, 8};     // C1: [h16 | denc(32 nat)] -> 64
    case 3: return {2, 4, 0, 14};    // C2: 64 -> 64
    case 4: return {1, 4, 0, 22};    // C3: 64 -> 3
    case 5: return {2, 0, 1, 26};    // C3^T: d(rgb_pre) (nat) -> d(hc2)
    case 6: return {2, 4, 0, 28};    // C2^T
    case 7: return {1, 4, 0, 36};    // C1^T (h16 rows only)
    case 8: return {2, 1, 0, 40};    // S2^T: d(h16) -> d(hs1)
    default: return {1, 4, 0, 42};   // S1^T: d(hs1) -> d(hash features)
  }
}

__device__ __forceinline__ int isrc(int step, int row, int k, bool nat) {
  switch (step) {
    case 0: return kSW1 + row * 32 + k;
    case 1: return row < 16 ? kSW2 + row * 64 + k : -1;
    case 2: return nat ? (k < 27 ? kCW1 + row * 48 + 16 + k : -1) : (k < 16 ? kCW1 + row * 48 + k : -1);
    case 3: return kCW2 + row * 64 + k;
    case 4: return row < 3 ? kCW3 + row * 64 + k : -1;
    case 5: return k < 3 ? kCW3 + k * 64 + row : -1;
    case 6: return kCW2 + k * 64 + row;
    case 7: return row < 16 ? kCW1 + k * 48 + row : -1;
    case 8: return k < 16 ? kSW2 + k * 64 + row : -1;
    default: return row < 32 ? kSW1 + k * 32 + row : -1;
  }
}

__global__ void __launch_bounds__(256) ipack_kernel(const float* __restrict__ params, __bf16* __restrict__ packed) {
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < (kIFwdFrags + kIBwdFrags) * 64; t += gridDim.x * blockDim.x) {
    const int frag = t >> 6, lane = t & 63;
    int step = 0;
    for (int s = 0; s < 10; ++s) if (frag >= istep(s).frag0) step = s;
    const IStep st = istep(step);
    const int ksn = st.ks_acc + st.ks_nat, rel = frag - st.frag0, mt = rel / ksn, ks = rel % ksn;
    const int row = mt * 32 + (lane & 31), h = lane >> 5;
    const bool nat = ks >= st.ks_acc;
    bf16x8 out;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = nat ? 16 * (ks - st.ks_acc) + 8 * h + j : 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
      const int src = isrc(step, row, k, nat);
      out[j] = (__bf16)(src >= 0 ? params[src] : 0.0f);
    }
    *reinterpret_cast<bf16x8*>(packed + (size_t)frag * 512 + lane * 8) = out;
  }
}

struct IArgs {
  const char* packed;
  const __bf16* hash_nat;   // nat blocks [n_pad,32] from nerf_hash_encode_fwd
  const float* dirs;        // [n,3] unit view directions
  const float* x_enc;       // encoded entry (InstantNeRFDecoder.forward(x_enc, d_enc)): [n,32] hash features and
  const float* d_enc;       // [n,27] direction codes given by the caller; NULL: hash_nat image + dirs
  __bf16* hash_nat_out;     // encoded entry, training: the feature image the wgrad pass reads is written here
  int64_t n, n_pad;
  float* rgb;               // [n,3]
  float* sigma;             // [n]
  // training stash (blocked images) and relu bits
  __bf16* hs1; __bf16* h16; __bf16* denc; __bf16* hc1; __bf16* hc2;
  uint4* mask;              // [tiles][256]
  // backward
  const float* d_rgb; const float* d_sigma;
  __bf16* dzs1; __bf16* dzs2; __bf16* dzc1; __bf16* dzc2; __bf16* dsmall;
  float* d_feat;            // [n,32] fp32
  float2* grad_lm;          // instead of d_feat: level-major gradients [16][n] float2 (the binned hash backward's input)
  unsigned* amax_bits;      // kAmaxSlots words: running maxima of |d_feat| as fp32 bits (the scatter's fixed-point scale)
  float* zero_grads;        // backward: the weight-gradient vector [kIParams], cleared here for the wgrad launch that follows and ADDS
};

template <int STEP, int KS, class Epi>
__device__ __forceinline__ void istep_run(const char* wbase, const bf16x8 (&b)[KS], Epi&& epi) {
  constexpr IStep st = istep(STEP);
  static_assert(KS == st.ks_acc + st.ks_nat, "k-steps");
  static_for<st.mt>([&](auto mc) {
    constexpr int m = decltype(mc)::value;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    acc = mtile_mfma<KS>(wbase, st.frag0 + m * KS, b, acc);
    epi(mc, acc);
  });
}

template <bool TRAIN>
__global__ void __launch_bounds__(kIThreads) imlp_fwd_kernel(const IArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
  for (int i = tid; i < kIFwdFrags * 64; i += kIThreads)
    reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(a.packed)[i];
  __syncthreads();
  const char* wbase = smem + lane * 16;
  const int64_t n_tiles = a.n_pad / kITile;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t wt = tile * 4 + wave, n = wt * 32 + col;
    const bool live = n < a.n;
    const int64_t nc = live ? n : a.n - 1;
    bf16x8 hin[2], denc[2];
    if (a.x_enc != nullptr) {
      // already-encoded inputs (src/decoders.py:136-162 as a stand-alone operator)
      encoded_operand<2, 32>(a.x_enc + nc * 32, half, hin);
      encoded_operand<2, plan::kDirDim>(a.d_enc + nc * plan::kDirDim, half, denc);
      if constexpr (TRAIN) {
        stash_nat(a.hash_nat_out, wt, 2, 0, col, half, hin[0]);
        stash_nat(a.hash_nat_out, wt, 2, 1, col, half, hin[1]);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        hin[ks] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(a.hash_nat) + ((wt * 2 + ks) * 64 + 2 * col + half) * 16);
      fourier_operand<2, plan::kDirDim>(a.dirs[nc * 3 + 0], a.dirs[nc * 3 + 1], a.dirs[nc * 3 + 2], half, denc);
    }
    if constexpr (TRAIN) {
      stash_nat(a.denc, wt, 2, 0, col, half, denc[0]);
      stash_nat(a.denc, wt, 2, 1, col, half, denc[1]);
    }
    uint32_t mw[3] = {0, 0, 0};
    auto relu_epi = [&](bf16x8* out, __bf16* stash, int layer) {
      return [=, &mw](auto mc, f32x16 acc) {
        constexpr int m = decltype(mc)::value;
        uint32_t bits = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) { bits |= (acc[r] > 0.0f ? 1u : 0u) << r; acc[r] = fmaxf(acc[r], 0.0f); }
        mw[layer] |= bits << (16 * m);
        acc_to_operand(acc, out[2 * m], out[2 * m + 1]);
        if constexpr (TRAIN) stash_block(stash, wt, 2, m, col, half, out[2 * m], out[2 * m + 1]);
      };
    };
    bf16x8 hs1[4], h16[2], hc1[4], hc2[4];
    istep_run<0, 2>(wbase, hin, relu_epi(hs1, a.hs1, 0));
    float h0 = 0.0f;
    istep_run<1, 4>(wbase, hs1, [&](auto, f32x16 acc) {
      h0 = acc[0];
      acc_to_operand(acc, h16[0], h16[1]);
      if constexpr (TRAIN) stash_block(a.h16, wt, 1, 0, col, half, h16[0], h16[1]);
    });
    if (live && half == 0) {
      const float x = h0 - 5.0f;                               // decoders.py:153
      a.sigma[n] = x > 20.0f ? x : log1pf(expf(x));            // F.softplus (threshold 20)
    }
    {
      bf16x8 cat[3] = {h16[0], denc[0], denc[1]};
      istep_run<2, 3>(wbase, cat, relu_epi(hc1, a.hc1, 1));
    }
    istep_run<3, 4>(wbase, hc1, relu_epi(hc2, a.hc2, 2));
    istep_run<4, 4>(wbase, hc2, [&](auto, f32x16 acc) {
      if (live && half == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a.rgb[n * 3 + c] = 1.0f / (1.0f + __expf(-acc[c]));
      }
    });
    if constexpr (TRAIN) a.mask[tile * kIThreads + tid] = make_uint4(mw[0], mw[1], mw[2], 0);
  }
}

__global__ void __launch_bounds__(kIThreads) imlp_bwd_kernel(const IArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
  for (int i = tid; i < kIBwdFrags * 64; i += kIThreads)
    reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(a.packed + kIFwdFrags * 1024)[i];
  __syncthreads();
  const char* wbase = smem + lane * 16 - kIFwdFrags * 1024;   // istep().frag0 counts from the forward stream
  const int64_t n_tiles = a.n_pad / kITile;
  if (a.zero_grads != nullptr)                                // instead of a fill launch before this one (4.5 us + its gap)
    for (int i = blockIdx.x * kIThreads + tid; i < kIParams; i += gridDim.x * kIThreads) a.zero_grads[i] = 0.0f;
  float amax = 0.0f;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t wt = tile * 4 + wave, n = wt * 32 + col;
    const bool live = n < a.n;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, gs = 0.f;
    if (live) {
      const float r0 = a.rgb[n * 3 + 0], r1 = a.rgb[n * 3 + 1], r2 = a.rgb[n * 3 + 2];
      g0 = a.d_rgb[n * 3 + 0] * r0 * (1.0f - r0);
      g1 = a.d_rgb[n * 3 + 1] * r1 * (1.0f - r1);
      g2 = a.d_rgb[n * 3 + 2] * r2 * (1.0f - r2);
      gs = a.d_sigma[n] * -expm1f(-a.sigma[n]);                // softplus'(x) = sigmoid(x) = 1 - exp(-softplus(x)), no cancellation
    }
    bf16x8 small;
#pragma unroll
    for (int j = 0; j < 8; ++j) small[j] = (__bf16)0.0f;
    if (half == 0) { small[0] = (__bf16)g0; small[1] = (__bf16)g1; small[2] = (__bf16)g2; }
    stash_nat(a.dsmall, wt, 1, 0, col, half, small);
    const uint4 mask = a.mask[tile * kIThreads + tid];
    auto grad_epi = [&](bf16x8* out, __bf16* stash, uint32_t bits32) {
      return [=](auto mc, f32x16 acc) {
        constexpr int m = decltype(mc)::value;
        const uint32_t bits = bits32 >> (16 * m);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = (bits >> r) & 1u ? acc[r] : 0.0f;
        acc_to_operand(acc, out[2 * m], out[2 * m + 1]);
        stash_block(stash, wt, 2, m, col, half, out[2 * m], out[2 * m + 1]);
      };
    };
    bf16x8 gc2[4], gc1[4], g16[2], gs1[4];
    { bf16x8 in[1] = {small}; istep_run<5, 1>(wbase, in, grad_epi(gc2, a.dzc2, mask.z)); }
    istep_run<6, 4>(wbase, gc2, grad_epi(gc1, a.dzc1, mask.y));
    istep_run<7, 4>(wbase, gc1, [&](auto, f32x16 acc) {
      if (half == 0) acc[0] += gs;                              // row 0 of h also feeds sigma
      acc_to_operand(acc, g16[0], g16[1]);
      stash_block(a.dzs2, wt, 1, 0, col, half, g16[0], g16[1]);
    });
    { bf16x8 in[1] = {g16[0]}; istep_run<8, 1>(wbase, in, grad_epi(gs1, a.dzs1, mask.x)); }
    istep_run<9, 4>(wbase, gs1, [&](auto, f32x16 acc) {
      if (!live) return;
      if (a.grad_lm != nullptr) {
        // registers 4g..4g+3 = features 8g + 4 half + (0..3) = levels 4g + 2 half and + 1: two float2 per group, each store
        // instruction covers 32 consecutive points of one level (256 contiguous bytes)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int lvl = 4 * g + 2 * half;
          a.grad_lm[(int64_t)lvl * a.n + n] = make_float2(acc[4 * g], acc[4 * g + 1]);
          a.grad_lm[(int64_t)(lvl + 1) * a.n + n] = make_float2(acc[4 * g + 2], acc[4 * g + 3]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) amax = fmaxf(amax, fabsf(acc[r]));
        return;
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        *reinterpret_cast<f32x4*>(a.d_feat + n * 32 + 8 * g + 4 * half) = v;
      }
      if (a.amax_bits != nullptr) {
#pragma unroll
        for (int r = 0; r < 16; ++r) amax = fmaxf(amax, fabsf(acc[r]));
      }
    });
  }
  // one atomic per workgroup at most, spread over kAmaxSlots words (common.h): one per wave on ONE word -- 4096 of them -- doubled this
  // kernel's time, one per workgroup on one word still cost ~8 us of queueing at the kernel's tail
  if (a.amax_bits != nullptr) publish_amax_slots(amax, a.amax_bits);
}

struct ILayout {
  int64_t n_pad;
  size_t hash_nat, hs1, h16, denc, hc1, hc2, mask, dzs1, dzs2, dzc1, dzc2, dsmall, slab, total;
};
static ILayout ilayout(int64_t n) {
  ILayout s{};
  s.n_pad = (n + kITile - 1) / kITile * kITile;
  const size_t np = (size_t)s.n_pad;
  size_t o = 0;
  s.hash_nat = o; o += np * 32 * 2;
  s.hs1 = o; o += np * 64 * 2;
  s.h16 = o; o += np * 32 * 2;
  s.denc = o; o += np * 32 * 2;
  s.hc1 = o; o += np * 64 * 2;
  s.hc2 = o; o += np * 64 * 2;
  s.mask = o; o += (np / kITile) * kIThreads * 16;
  s.dzs1 = o; o += np * 64 * 2;
  s.dzs2 = o; o += np * 32 * 2;
  s.dzc1 = o; o += np * 64 * 2;
  s.dzc2 = o; o += np * 64 * 2;
  s.dsmall = o; o += np * 16 * 2;
  o = (o + 255) / 256 * 256;
  s.slab = o; o += kSmallSlabBytes;          // partial tiles of the weight-gradient launch (option "deterministic")
  s.total = (o + 255) / 256 * 256;
  return s;
}

static IArgs iargs(const void* packed, void* ws, const float* dirs, int64_t n, float* rgb, float* sigma) {
  const ILayout l = ilayout(n);
  char* w = static_cast<char*>(ws);
  IArgs a{};
  a.packed = static_cast<const char*>(packed);
  a.hash_nat = reinterpret_cast<const __bf16*>(w + l.hash_nat);
  a.dirs = dirs; a.n = n; a.n_pad = l.n_pad; a.rgb = rgb; a.sigma = sigma;
  a.hs1 = reinterpret_cast<__bf16*>(w + l.hs1); a.h16 = reinterpret_cast<__bf16*>(w + l.h16);
  a.denc = reinterpret_cast<__bf16*>(w + l.denc); a.hc1 = reinterpret_cast<__bf16*>(w + l.hc1);
  a.hc2 = reinterpret_cast<__bf16*>(w + l.hc2); a.mask = reinterpret_cast<uint4*>(w + l.mask);
  a.dzs1 = reinterpret_cast<__bf16*>(w + l.dzs1); a.dzs2 = reinterpret_cast<__bf16*>(w + l.dzs2);
  a.dzc1 = reinterpret_cast<__bf16*>(w + l.dzc1); a.dzc2 = reinterpret_cast<__bf16*>(w + l.dzc2);
  a.dsmall = reinterpret_cast<__bf16*>(w + l.dsmall);
  return a;
}

static int grid_for_tiles(int64_t tiles) {
  int n_cu = 0;
  if (device_cu_count(&n_cu) != NERF_OK) return -1;
  const int64_t cap = (int64_t)n_cu * 4;
  return (int)(tiles < cap ? tiles : cap);
}

}  // namespace nerf

using namespace nerf;

extern "C" size_t nerf_imlp_packed_bytes(void) { return kIPackBytes; }
extern "C" size_t nerf_imlp_workspace_bytes(int64_t n) { return n > 0 ? ilayout(n).total : 0; }
extern "C" size_t nerf_imlp_hash_operand_offset(int64_t n) { return n > 0 ? ilayout(n).hash_nat : 0; }

extern "C" int nerf_imlp_pack(const float* params_f32, void* packed, nerf_stream_t stream) {
  NERF_REQUIRE(params_f32 && packed && ((uintptr_t)packed & 255) == 0, "nerf_imlp_pack: bad pointer");
  hipLaunchKernelGGL(ipack_kernel, dim3(16), dim3(256), 0, as_stream(stream), params_f32, static_cast<__bf16*>(packed));
  return check_launch("nerf_imlp_pack");
}

extern "C" int nerf_imlp_fwd(const void* packed, void* workspace, const float* dirs, int64_t n, float* rgb,
                             float* sigma, int train, nerf_stream_t stream) {
  NERF_REQUIRE(n >= 0, "nerf_imlp_fwd: n=%lld", (long long)n);
  if (n == 0) return NERF_OK;
  NERF_REQUIRE(packed && workspace && dirs && rgb && sigma && ((uintptr_t)workspace & 255) == 0, "nerf_imlp_fwd: bad pointer");
  const IArgs a = iargs(packed, workspace, dirs, n, rgb, sigma);
  const int grid = grid_for_tiles(a.n_pad / kITile);
  if (grid <= 0) return fail(NERF_ELAUNCH, "nerf_imlp_fwd: cannot query device");
  if (train) hipLaunchKernelGGL(imlp_fwd_kernel<true>, dim3(grid), dim3(kIThreads), kIFwdFrags * 1024, as_stream(stream), a);
  else hipLaunchKernelGGL(imlp_fwd_kernel<false>, dim3(grid), dim3(kIThreads), kIFwdFrags * 1024, as_stream(stream), a);
  return check_launch("nerf_imlp_fwd");
}

extern "C" int nerf_imlp_fwd_encoded(const void* packed, void* workspace, const float* x_enc, const float* d_enc, int64_t n,
                                     float* rgb, float* sigma, int train, nerf_stream_t stream) {
  NERF_REQUIRE(n >= 0, "nerf_imlp_fwd_encoded: n=%lld", (long long)n);
  if (n == 0) return NERF_OK;
  NERF_REQUIRE(packed && workspace && x_enc && d_enc && rgb && sigma && ((uintptr_t)workspace & 255) == 0, "nerf_imlp_fwd_encoded: bad pointer");
  IArgs a = iargs(packed, workspace, nullptr, n, rgb, sigma);
  a.x_enc = x_enc; a.d_enc = d_enc;
  a.hash_nat_out = const_cast<__bf16*>(a.hash_nat);
  const int grid = grid_for_tiles(a.n_pad / kITile);
  if (grid <= 0) return fail(NERF_ELAUNCH, "nerf_imlp_fwd_encoded: cannot query device");
  if (train) hipLaunchKernelGGL(imlp_fwd_kernel<true>, dim3(grid), dim3(kIThreads), kIFwdFrags * 1024, as_stream(stream), a);
  else hipLaunchKernelGGL(imlp_fwd_kernel<false>, dim3(grid), dim3(kIThreads), kIFwdFrags * 1024, as_stream(stream), a);
  return check_launch("nerf_imlp_fwd_encoded");
}

static int imlp_bwd_impl(const void* packed, void* workspace, const float* rgb, const float* sigma, const float* d_rgb,
                         const float* d_sigma, int64_t n, float* grads_f32, float* d_feat, float2* grad_lm, unsigned* amax_bits,
                         nerf_stream_t stream);

extern "C" int nerf_imlp_bwd(const void* packed, void* workspace, const float* rgb, const float* sigma,
                             const float* d_rgb, const float* d_sigma, int64_t n, float* grads_f32,
                             float* d_feat, nerf_stream_t stream) {
  NERF_REQUIRE(n == 0 || d_feat != nullptr, "nerf_imlp_bwd: d_feat is NULL");
  return imlp_bwd_impl(packed, workspace, rgb, sigma, d_rgb, d_sigma, n, grads_f32, d_feat, nullptr, nullptr, stream);
}

// row-major feature gradients AND their largest magnitude (amax_bits: device u32, max-accumulated fp32 bits; the caller zeroes it):
// what the speculative hash backward needs without the level-major copy (whose stores cost this kernel 33 us on 200 k points)
extern "C" int nerf_imlp_bwd_amax(const void* packed, void* workspace, const float* rgb, const float* sigma,
                                  const float* d_rgb, const float* d_sigma, int64_t n, float* grads_f32,
                                  float* d_feat, void* amax_bits, nerf_stream_t stream) {
  NERF_REQUIRE(n == 0 || (d_feat != nullptr && amax_bits != nullptr), "nerf_imlp_bwd_amax: NULL output");
  return imlp_bwd_impl(packed, workspace, rgb, sigma, d_rgb, d_sigma, n, grads_f32, d_feat, nullptr, static_cast<unsigned*>(amax_bits), stream);
}

extern "C" int nerf_imlp_bwd_lm(const void* packed, void* workspace, const float* rgb, const float* sigma,
                                const float* d_rgb, const float* d_sigma, int64_t n, float* grads_f32,
                                void* grad_lm, void* amax_bits, nerf_stream_t stream) {
  NERF_REQUIRE(n == 0 || (grad_lm != nullptr && amax_bits != nullptr), "nerf_imlp_bwd_lm: NULL output");
  return imlp_bwd_impl(packed, workspace, rgb, sigma, d_rgb, d_sigma, n, grads_f32, nullptr, static_cast<float2*>(grad_lm),
                       static_cast<unsigned*>(amax_bits), stream);
}

static int imlp_bwd_impl(const void* packed, void* workspace, const float* rgb, const float* sigma, const float* d_rgb,
                         const float* d_sigma, int64_t n, float* grads_f32, float* d_feat, float2* grad_lm, unsigned* amax_bits,
                         nerf_stream_t stream) {
  NERF_REQUIRE(n >= 0 && grads_f32, "nerf_imlp_bwd: bad arguments");
  if (n == 0) {
    if (hipMemsetAsync(grads_f32, 0, sizeof(float) * kIParams, as_stream(stream)) != hipSuccess)
      return fail(NERF_ELAUNCH, "nerf_imlp_bwd: memset failed");
    return NERF_OK;
  }
  NERF_REQUIRE(packed && workspace && rgb && sigma && d_rgb && d_sigma && (d_feat || grad_lm), "nerf_imlp_bwd: NULL pointer");
  IArgs a = iargs(packed, workspace, nullptr, n, const_cast<float*>(rgb), const_cast<float*>(sigma));
  a.d_rgb = d_rgb; a.d_sigma = d_sigma; a.d_feat = d_feat; a.grad_lm = grad_lm; a.amax_bits = amax_bits;
  a.zero_grads = grads_f32;            // the dgrad kernel clears the vector the wgrad launch behind it adds to
  const int grid = grid_for_tiles(a.n_pad / kITile);
  if (grid <= 0) return fail(NERF_ELAUNCH, "nerf_imlp_bwd: cannot query device");
  hipLaunchKernelGGL(imlp_bwd_kernel, dim3(grid), dim3(kIThreads), kIBwdFrags * 1024, as_stream(stream), a);
  int rc = check_launch("nerf_imlp_bwd (dgrad)");
  if (rc != NERF_OK) return rc;
  // weight gradients: five small jobs on the shared split-K kernel
  const ILayout l = ilayout(n);
  const char* w = static_cast<const char*>(workspace);
  WgradArgs wa{};
  auto job = [&](size_t a_off, int a_bytes, int mt_a, size_t b_off, int nt_acc, size_t bn_off, int nt_nat, int kind) {
    WgradJob j{};
    j.a = w + a_off; j.a_bytes = a_bytes; j.mt_a = mt_a;
    if (nt_acc) { j.b_acc = w + b_off; j.b_acc_bytes = nt_acc * 2048; j.nt_acc = nt_acc; }
    if (nt_nat) { j.b_nat = w + bn_off; j.b_nat_bytes = nt_nat * 2048; j.nt_nat = nt_nat; }
    j.bias_nat_col = -1; j.kind = kind;
    return j;
  };
  { WgradJob j = job(l.dzs1, 4096, 2, 0, 0, l.hash_nat, 1, 6); j.w_off = kSW1; j.w_ld = 32; j.o_valid = 64; j.nat_valid = 32; wa.jobs[0] = j; }
  { WgradJob j = job(l.dzs2, 2048, 1, l.hs1, 2, 0, 0, 7); j.w_off = kSW2; j.w_ld = 64; j.o_valid = 16; j.acc_valid = 64; wa.jobs[1] = j; }
  { WgradJob j = job(l.dzc1, 4096, 2, l.h16, 1, l.denc, 1, 8); j.w_off = kCW1; j.w_ld = 48; j.o_valid = 64; j.acc_valid = 16; j.nat_valid = 27; j.nat_col0 = 16; wa.jobs[2] = j; }
  { WgradJob j = job(l.dzc2, 4096, 2, l.hc1, 2, 0, 0, 7); j.w_off = kCW2; j.w_ld = 64; j.o_valid = 64; j.acc_valid = 64; wa.jobs[3] = j; }
  { WgradJob j = job(l.dsmall, 1024, 1, l.hc2, 2, 0, 0, 9); j.a_nat = 1; j.split_n = 1; j.w_off = kCW3; j.w_ld = 64; j.o_valid = 3; j.acc_valid = 64; wa.jobs[4] = j; }
  wa.n_jobs = 5;
  if (options().deterministic)      // partial tiles summed in workgroup order instead of one float atomic per weight and workgroup
    return wgrad_launch(wa, n, grads_f32, as_stream(stream), reinterpret_cast<float*>(static_cast<char*>(workspace) + l.slab), kSmallSlabBytes);
  return wgrad_launch(wa, n, grads_f32, as_stream(stream));
}
