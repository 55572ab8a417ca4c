// Part 4 dual-hash dynamic field: the small networks of reference src/core.py:282-352 as fused
// bf16-MFMA register chains, replacing tinycudann FullyFusedMLP (HashDeformationDecoder, src/decoders.py:264-318;
// InstantNeRFDecoder at pos_dim 32 + 21, src/decoders.py:136-162 with src/core.py:222) and the nn.Linear
// TimeModulationNetwork (src/decoders.py:321-371).  PARITY of the tcnn internals UNPINNED (library absent); the
// checker is the reference's own Part 4 code around the stand-in tinycudann (golden g14).
//
//   deformation chain (one kernel per direction), per sample:
//     tcode = Fourier_10(t')                                   [t | sin | cos ...] 21 columns  (src/embeddings.py:28-32)
//     tm    = sigmoid(W_T2 relu(W_T1 tcode + b_T1) + b_T2)     time modulation, 64                 (decoders.py:368-371)
//     df    = sum_k w_k(t') feat_k,  w = normalised triangle weights around t = 0, .5, 1       (core.py:313-332)
//     dx    = scale * W_D3 relu(W_D2 relu(W_D1 [df | tm]))     bias-free 88 -> 64 -> 64 -> 3     (decoders.py:313-316)
//     x_c   = x + dx                                                                             (core.py:341)
//   canonical chain: InstantNeRFDecoder on [hash(x_c) (32) | tcode (21)] and the direction code  (core.py:344-349)
//
// Same register chain as imlp.hip / the 8x256 decoder (mlp_chain.h): 32 samples per wave on the MFMA column,
// accumulator tiles -> 16-bit B fragments, all weight fragments resident in LDS.  The FORWARD chains contract fp16
// operands (v_mfma_f32_32x32x16_f16, what tinycudann's FullyFusedMLP computes in): delta_x moves x_canonical inside a
// hash grid whose finest cells are 4e-4 wide, and bf16's 8 mantissa bits put ~1e-3 of rounding on a displacement of
// 0.2 -- several cells; fp16 keeps it below one.  The backward chains and the training images stay bf16 (gradients span
// more binades than fp16 holds without a loss scale): every layer input is stashed as a blocked bf16 image for the
// shared split-K weight-gradient kernel (mlp_wgrad.hip).
//
// Parameter vector (fp32, [out,in] row-major; the layouts of the module's state dict, concatenated):
//   T1W [64,21] T1b [64] T2W [64,64] T2b [64]              time_modulation.net.{0,2}.{weight,bias}
//   D1 [64,96] D2 [64,64] D3 [16,64]                        deform_decoder.deform_net.params (cols: 24 hash | 64 tm | 8 pad)
//   S1 [64,64] S2 [16,64]                                   decoder.sigma_net.params       (cols: 32 hash | 21 tcode | pad)
//   C1 [64,48] C2 [64,64] C3 [16,64]                        decoder.color_net.params
//   scale [1]                                               deform_decoder.displacement_scale
#include "mlp_chain.h"
#include "mlp_wgrad.h"

namespace nerf {
namespace p4 {

constexpr int kT1W = 0, kT1b = 1344, kT2W = 1408, kT2b = 5504, kD1 = 5568, kD2 = 11712, kD3 = 15808;
constexpr int kS1 = 16832, kS2 = 20928, kC1 = 21952, kC2 = 25024, kC3 = 29120, kScale = 30144, kParams = 30145;
constexpr int kTimeDim = 21, kHashDeform = 24, kDirDim = 27;
constexpr int kThreads = 256, kTile = 128;

// (m-tiles, k-steps fed by the previous step's accumulators, natural-order k-steps, first fragment)
struct Step { int mt, ks_acc, ks_nat, frag0; };
enum { T1, T2, D1, D2, D3, D3t, D2t, D1tT, D1tH, T2t, S1, S2, C1, C2, C3, C3t, C2t, C1t, S2t, S1t, kSteps };
constexpr Step step_of(int s) {
  switch (s) {
    case T1: return {2, 0, 2, 0};      // tcode (32 nat) -> 64
    case T2: return {2, 4, 0, 4};      // 64 -> 64
    case D1: return {2, 4, 2, 12};     // [tm (64) | df (32 nat)] -> 64
    case D2: return {2, 4, 0, 24};
    case D3: return {1, 4, 0, 32};     // 64 -> 3
    case D3t: return {2, 0, 1, 36};    // d(dx raw) (16 nat) -> d(hd2)
    case D2t: return {2, 4, 0, 38};
    case D1tT: return {2, 4, 0, 46};   // -> d(tm)
    case D1tH: return {1, 4, 0, 54};   // -> d(df)
    case T2t: return {2, 4, 0, 58};    // d(tm_pre) -> d(ht1)
    case S1: return {2, 0, 4, 66};     // [hash (32) | tcode (32)] nat -> 64
    case S2: return {1, 4, 0, 74};     // 64 -> 16
    case C1: return {2, 1, 2, 78};     // [h16 | dir code (32 nat)] -> 64
    case C2: return {2, 4, 0, 84};
    case C3: return {1, 4, 0, 92};
    case C3t: return {2, 0, 1, 96};
    case C2t: return {2, 4, 0, 98};
    case C1t: return {1, 4, 0, 106};   // h16 rows only
    case S2t: return {2, 1, 0, 110};
    default: return {1, 4, 0, 112};    // S1t: hash rows only
  }
}
constexpr int kFrags = 116;
constexpr int kDeformFwd0 = 0, kDeformFwdN = 36, kDeformBwd0 = 36, kDeformBwdN = 30;
constexpr int kCanonFwd0 = 66, kCanonFwdN = 30, kCanonBwd0 = 96, kCanonBwdN = 20;
constexpr size_t kPackBytes = (size_t)kFrags * 1024 + 256;   // + T2 bias table (64 fp32)
constexpr size_t kPackBiasOff = (size_t)kFrags * 1024;

// parameter index of weight (row, k) of a step, or -1 (zero padding); nat: k is a natural-order column
__device__ __forceinline__ int src_of(int step, int row, int k, bool nat) {
  switch (step) {
    case T1: return k < kTimeDim ? kT1W + row * 21 + k : (k == kTimeDim ? kT1b + row : -1);   // column 21 of the code is the constant 1
    case T2: return kT2W + row * 64 + k;
    case D1: return nat ? (k < kHashDeform ? kD1 + row * 96 + k : -1) : kD1 + row * 96 + kHashDeform + k;
    case D2: return kD2 + row * 64 + k;
    case D3: return row < 3 ? kD3 + row * 64 + k : -1;
    case D3t: return k < 3 ? kD3 + k * 64 + row : -1;
    case D2t: return kD2 + k * 64 + row;
    case D1tT: return kD1 + k * 96 + kHashDeform + row;
    case D1tH: return row < kHashDeform ? kD1 + k * 96 + row : -1;
    case T2t: return kT2W + k * 64 + row;
    case S1: return k < 32 + kTimeDim ? kS1 + row * 64 + k : -1;
    case S2: return row < 16 ? kS2 + row * 64 + k : -1;
    case C1: return nat ? (k < kDirDim ? kC1 + row * 48 + 16 + k : -1) : (k < 16 ? kC1 + row * 48 + k : -1);
    case C2: return kC2 + row * 64 + k;
    case C3: return row < 3 ? kC3 + row * 64 + k : -1;
    case C3t: return k < 3 ? kC3 + k * 64 + row : -1;
    case C2t: return kC2 + k * 64 + row;
    case C1t: return row < 16 ? kC1 + k * 48 + row : -1;
    case S2t: return k < 16 ? kS2 + k * 64 + row : -1;
    default: return row < 32 ? kS1 + k * 64 + row : -1;
  }
}

__global__ void __launch_bounds__(256) pack_kernel(const float* __restrict__ params, char* __restrict__ packed) {
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < kFrags * 64; t += gridDim.x * blockDim.x) {
    const int frag = t >> 6, lane = t & 63;
    int step = 0;
    for (int s = 0; s < kSteps; ++s) if (frag >= step_of(s).frag0) step = s;
    const Step st = step_of(step);
    const int ksn = st.ks_acc + st.ks_nat, rel = frag - st.frag0, mt = rel / ksn, ks = rel % ksn;
    const int row = mt * 32 + (lane & 31), h = lane >> 5;
    const bool nat = ks >= st.ks_acc;
    const bool fwd = step <= D3 || (step >= S1 && step <= C3);       // forward chains: fp16 fragments; backward: bf16
    unsigned short out[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = nat ? 16 * (ks - st.ks_acc) + 8 * h + j : 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
      const int src = src_of(step, row, k, nat);
      const float v = src >= 0 ? params[src] : 0.0f;
      out[j] = fwd ? __builtin_bit_cast(unsigned short, (_Float16)v) : __builtin_bit_cast(unsigned short, (__bf16)v);
    }
    uint4 bits;
    bits.x = out[0] | ((unsigned)out[1] << 16); bits.y = out[2] | ((unsigned)out[3] << 16);
    bits.z = out[4] | ((unsigned)out[5] << 16); bits.w = out[6] | ((unsigned)out[7] << 16);
    *reinterpret_cast<uint4*>(packed + (size_t)frag * 1024 + lane * 16) = bits;
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) reinterpret_cast<float*>(packed + kPackBiasOff)[threadIdx.x] = params[kT2b + threadIdx.x];
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// 1-D Fourier code of the time stamp, element f of [t | sin(2^0 pi t) | cos(2^0 pi t) | sin(2^1 pi t) | ...] (21 columns);
// ONE: column 21 = 1 (bias column of the time-modulation layer), else 0.  v[ks][j] = column 16 ks + 8 half + j.
template <int KS, bool ONE>
__device__ __forceinline__ void time_values(float t, int half, float (&v)[KS][8]) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int f = 16 * ks + 8 * half + j;                          // this lane's column: one sin/cos evaluation
      const int c = f > 0 ? f - 1 : 0;
      const float trig = sincos_rev(t, (float)(1u << (c >> 1)), (c & 1) ? 0.25f : 0.0f);
      v[ks][j] = f == 0 ? t : (f < kTimeDim ? trig : ((ONE && f == kTimeDim) ? 1.0f : 0.0f));
    }
  }
}
// Fourier code of a unit direction (src/embeddings.py:28-32, L = 4: 27 columns, zero pad), as fourier_operand computes it
template <int KS>
__device__ __forceinline__ void dir_values(float x0, float x1, float x2, int half, float (&v)[KS][8]) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const FeatSpec s0 = feat_spec<kDirDim>(16 * ks + j), s1 = feat_spec<kDirDim>(16 * ks + 8 + j);
      const float v0 = s0.raw == 3 || s0.raw == 2 ? 0.0f : feat_eval<kDirDim>(s0, x0, x1, x2);
      const float v1 = s1.raw == 3 || s1.raw == 2 ? 0.0f : feat_eval<kDirDim>(s1, x0, x1, x2);
      v[ks][j] = half ? v1 : v0;
    }
  }
}
template <int KS>
__device__ __forceinline__ void cast_values(const float (&v)[KS][8], f16x8 (&h)[KS], bf16x8 (&b)[KS]) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) { h[ks][j] = (_Float16)v[ks][j]; b[ks][j] = (__bf16)v[ks][j]; }
}

// forward m-tile on fp16 operands (the same rotating fragment window as mlp_chain.h::mtile_mfma)
template <int KS>
__device__ __forceinline__ f32x16 mtile_mfma16(const char* a_base, int frag_off, const f16x8 (&b)[KS], f32x16 acc) {
  constexpr int D = KS < kAhead ? KS : kAhead;
  f16x8 win[D];
#pragma unroll
  for (int i = 0; i < D; ++i) win[i] = *reinterpret_cast<const f16x8*>(a_base + (frag_off + i) * 1024);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const f16x8 cur = win[ks % D];
    if (ks + D < KS) win[ks % D] = *reinterpret_cast<const f16x8*>(a_base + (frag_off + ks + D) * 1024);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur, b[ks], acc, 0, 0, 0);
  }
  return acc;
}
template <int STEP, int KS, class Epi>
__device__ __forceinline__ void run16(const char* wbase, const f16x8 (&b)[KS], Epi&& epi) {
  constexpr Step st = step_of(STEP);
  static_assert(KS == st.ks_acc + st.ks_nat, "k-steps");
  static_for<st.mt>([&](auto mc) {
    constexpr int m = decltype(mc)::value;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    acc = mtile_mfma16<KS>(wbase, st.frag0 + m * KS, b, acc);
    epi(mc, acc);
  });
}
__device__ __forceinline__ void acc_to_operand16(const f32x16& acc, f16x8& lo, f16x8& hi) {
#pragma unroll
  for (int j = 0; j < 8; ++j) { lo[j] = (_Float16)acc[j]; hi[j] = (_Float16)acc[8 + j]; }
}
__device__ __forceinline__ f16x8 load_nat16(const __bf16* img, int64_t wt, int n_ks, int ks, int col, int half) {
  return *reinterpret_cast<const f16x8*>(reinterpret_cast<const char*>(img) + ((wt * n_ks + ks) * 64 + 2 * col + half) * 16);
}

template <int STEP, int KS, class Epi>
__device__ __forceinline__ void run(const char* wbase, const bf16x8 (&b)[KS], Epi&& epi) {
  constexpr Step st = step_of(STEP);
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

__device__ __forceinline__ bf16x8 load_nat(const __bf16* img, int64_t wt, int n_ks, int ks, int col, int half) {
  return *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(img) + ((wt * n_ks + ks) * 64 + 2 * col + half) * 16);
}
// the two B fragments of m-tile m of a blocked image (what stash_block wrote)
__device__ __forceinline__ void load_block(const __bf16* img, int64_t wt, int n_mtiles, int m, int col, int half, bf16x8& lo, bf16x8& hi) {
  const char* p = reinterpret_cast<const char*>(img) + (wt * n_mtiles + m) * 2048 + block_lane_offset(col, half);
  lo = *reinterpret_cast<const bf16x8*>(p);
  hi = *reinterpret_cast<const bf16x8*>(p + 128);
}

struct DeformArgs {
  const char* packed;
  const __bf16* feat[3];   // nat images [n_pad,32] of the three deformation grids at x', FP16 (features 0..23 valid)
  const float* t;          // [n] per-sample time stamp t' (after the optional noise)
  const float* blend;      // [n,3] explicit grid weights or NULL: triangle weights of t' (core.py:324-332)
  const float* x;          // [n,3] sample positions (x_c = x + dx)
  const float* params;     // flat fp32 parameters (displacement_scale is read from here)
  int64_t n, n_pad;
  float* dx;               // [n,3]
  float* xc;               // [n,3]
  float* raw;              // [n,3] dx / scale (training)
  // training stash
  __bf16* tc; __bf16* ht1; __bf16* tm; __bf16* df; __bf16* hd1; __bf16* hd2;
  uint4* mask;             // [tiles][256]: T1, D1, D2 relu bits
  float* wts;              // [n_pad,4] the blend weights used (training)
  // backward
  const float* d_dx;       // [n,3]
  __bf16* dzt1; __bf16* dzt2; __bf16* dzd1; __bf16* dzd2; __bf16* dsmall;
  float* d_feat[3];        // [n,24] fp32 gradients for the three grids' scatter
  float* g_scale;          // parameter-gradient slot of displacement_scale (accumulated)
  unsigned* sum_ws;        // non-null (option "deterministic"): workspace of common.h::ordered_block_sum for g_scale
  unsigned* amax_bits;     // non-null: kAmaxSlots words that max-accumulate the largest |d_feat| as fp32 bits (common.h; the hash scatter's scale)
  float2* grad_lm;         // non-null: INSTEAD of d_feat, level-major gradients [3 * 12][n] (virtual level = grid * 12 + level): the
                           // layout the binned hash scatter reads coalesced (its count pass writes it otherwise)
};

__device__ __forceinline__ void triangle_weights(float t, float (&w)[3]) {
  // clamp(1 - |t - a| / 0.5, 0, 1) around a = 0, 0.5, 1, normalised with + 1e-8 (reference src/core.py:324-332)
  const float a[3] = {0.0f, 0.5f, 1.0f};
  float s = 1e-8f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    w[k] = fminf(fmaxf(1.0f - fabsf(t - a[k]) / 0.5f, 0.0f), 1.0f);
    s += w[k];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) w[k] = w[k] / s;
}

template <bool TRAIN>
__global__ void __launch_bounds__(kThreads) deform_fwd_kernel(const DeformArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
  for (int i = tid; i < kDeformFwdN * 64; i += kThreads)
    reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(a.packed + kDeformFwd0 * 1024)[i];
  float* bias_lds = reinterpret_cast<float*>(smem + kDeformFwdN * 1024);
  if (tid < 64) bias_lds[tid] = reinterpret_cast<const float*>(a.packed + kPackBiasOff)[tid];
  __syncthreads();
  const char* wbase = smem + lane * 16 - kDeformFwd0 * 1024;
  const float scale = a.params[kScale];
  const int64_t n_tiles = a.n_pad / kTile;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t wt = tile * 4 + wave, n = wt * 32 + col;
    const bool live = n < a.n;
    const int64_t nc = live ? n : a.n - 1;
    const float t = a.t[nc];
    f16x8 tc[2];
    bf16x8 tc_b[2];
    {
      float v[2][8];
      time_values<2, true>(t, half, v);
      cast_values<2>(v, tc, tc_b);
    }
    uint32_t mw[3] = {0, 0, 0};
    // relu epilogue: fp16 operand of the next step; training: bf16 image of the same values for the weight-gradient pass
    auto relu_epi = [&](f16x8* out, __bf16* stash, int layer) {
      return [=, &mw](auto mc, f32x16 acc) {
        constexpr int m = decltype(mc)::value;
        uint32_t bits = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) { bits |= (acc[r] > 0.0f ? 1u : 0u) << r; acc[r] = fmaxf(acc[r], 0.0f); }
        mw[layer] |= bits << (16 * m);
        acc_to_operand16(acc, out[2 * m], out[2 * m + 1]);
        if constexpr (TRAIN) {
          bf16x8 lo, hi;
          acc_to_operand(acc, lo, hi);
          stash_block(stash, wt, 2, m, col, half, lo, hi);
        }
      };
    };
    // ---- time modulation (decoders.py:368-371) ----
    f16x8 ht1[4], tm[4];
    run16<T1, 2>(wbase, tc, relu_epi(ht1, a.ht1, 0));
    run16<T2, 4>(wbase, ht1, [&](auto mc, f32x16 acc) {
      constexpr int m = decltype(mc)::value;
      const f32x16 b = bias_tile(bias_lds, 32 * m, half);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 1.0f / (1.0f + __expf(-(acc[r] + b[r])));
      acc_to_operand16(acc, tm[2 * m], tm[2 * m + 1]);
      if constexpr (TRAIN) {
        bf16x8 lo, hi;
        acc_to_operand(acc, lo, hi);
        stash_block(a.tm, wt, 2, m, col, half, lo, hi);
      }
    });
    // ---- tri-grid blend (core.py:313-336) ----
    float w[3];
    if (a.blend != nullptr) { w[0] = a.blend[nc * 3 + 0]; w[1] = a.blend[nc * 3 + 1]; w[2] = a.blend[nc * 3 + 2]; }
    else triangle_weights(t, w);
    f16x8 df[2];
    bf16x8 df_b[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const f16x8 f0 = load_nat16(a.feat[0], wt, 2, ks, col, half), f1 = load_nat16(a.feat[1], wt, 2, ks, col, half),
                  f2 = load_nat16(a.feat[2], wt, 2, ks, col, half);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool valid = 16 * ks + 8 * half + j < kHashDeform;      // the hash forward leaves features 24..31 unwritten
        const float v = valid ? w[0] * (float)f0[j] + w[1] * (float)f1[j] + w[2] * (float)f2[j] : 0.0f;
        df[ks][j] = (_Float16)v;
        df_b[ks][j] = (__bf16)v;
      }
    }
    if constexpr (TRAIN) {
      stash_nat(a.tc, wt, 2, 0, col, half, tc_b[0]);
      stash_nat(a.tc, wt, 2, 1, col, half, tc_b[1]);
      stash_nat(a.df, wt, 2, 0, col, half, df_b[0]);
      stash_nat(a.df, wt, 2, 1, col, half, df_b[1]);
      if (half == 0) *reinterpret_cast<f32x4*>(a.wts + n * 4) = f32x4{w[0], w[1], w[2], 0.0f};
    }
    // ---- displacement decoder (decoders.py:313-316) ----
    f16x8 hd1[4], hd2[4];
    {
      f16x8 cat[6] = {tm[0], tm[1], tm[2], tm[3], df[0], df[1]};
      run16<D1, 6>(wbase, cat, relu_epi(hd1, a.hd1, 1));
    }
    run16<D2, 4>(wbase, hd1, relu_epi(hd2, a.hd2, 2));
    run16<D3, 4>(wbase, hd2, [&](auto, f32x16 acc) {
      if (live && half == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float d = acc[c] * scale;
          a.dx[n * 3 + c] = d;
          a.xc[n * 3 + c] = a.x[n * 3 + c] + d;
          if constexpr (TRAIN) a.raw[n * 3 + c] = acc[c];
        }
      }
    });
    if constexpr (TRAIN) a.mask[tile * kThreads + tid] = make_uint4(mw[0], mw[1], mw[2], 0);
  }
}

__global__ void __launch_bounds__(kThreads) deform_bwd_kernel(const DeformArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
  for (int i = tid; i < kDeformBwdN * 64; i += kThreads)
    reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(a.packed + kDeformBwd0 * 1024)[i];
  __syncthreads();
  const char* wbase = smem + lane * 16 - kDeformBwd0 * 1024;
  const float scale = a.params[kScale];
  float gscale_local = 0.0f, amax = 0.0f;
  const int64_t n_tiles = a.n_pad / kTile;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t wt = tile * 4 + wave, n = wt * 32 + col;
    const bool live = n < a.n;
    float g[3] = {0.f, 0.f, 0.f};
    if (live) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float d = a.d_dx[n * 3 + c];
        g[c] = d * scale;                                           // dx = raw * scale
        if (half == 0) gscale_local += d * a.raw[n * 3 + c];
      }
    }
    bf16x8 small;
#pragma unroll
    for (int j = 0; j < 8; ++j) small[j] = (__bf16)0.0f;
    if (half == 0) { small[0] = (__bf16)g[0]; small[1] = (__bf16)g[1]; small[2] = (__bf16)g[2]; }
    stash_nat(a.dsmall, wt, 1, 0, col, half, small);
    const uint4 mask = a.mask[tile * kThreads + tid];
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
    bf16x8 gd2[4], gd1[4], gt2[4], gt1[4];
    { bf16x8 in[1] = {small}; run<D3t, 1>(wbase, in, grad_epi(gd2, a.dzd2, mask.z)); }
    run<D2t, 4>(wbase, gd2, grad_epi(gd1, a.dzd1, mask.y));
    // d(tm) -> through the sigmoid: d(tm_pre) = d(tm) tm (1 - tm), with the stashed (bf16) gate values
    run<D1tT, 4>(wbase, gd1, [&](auto mc, f32x16 acc) {
      constexpr int m = decltype(mc)::value;
      bf16x8 lo, hi;
      load_block(a.tm, wt, 2, m, col, half, lo, hi);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float s = (float)(r < 8 ? lo[r] : hi[r - 8]);
        acc[r] *= s * (1.0f - s);
      }
      acc_to_operand(acc, gt2[2 * m], gt2[2 * m + 1]);
      stash_block(a.dzt2, wt, 2, m, col, half, gt2[2 * m], gt2[2 * m + 1]);
    });
    // d(df) -> the three grids, each weighted by its blend weight
    run<D1tH, 4>(wbase, gd1, [&](auto, f32x16 acc) {
      if (live) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(a.wts + n * 4);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
          for (int q = 0; q < 3; ++q) {                              // rows 8q + 4 half + (0..3) < 24
            f32x4 v = {acc[4 * q] * w[k], acc[4 * q + 1] * w[k], acc[4 * q + 2] * w[k], acc[4 * q + 3] * w[k]};
            if (a.grad_lm != nullptr) {                              // levels 4q + 2 half, + 1 of grid k
              float2* lm = a.grad_lm + (int64_t)(k * (kHashDeform / 2) + 4 * q + 2 * half) * a.n + n;
              lm[0] = make_float2(v[0], v[1]);
              lm[a.n] = make_float2(v[2], v[3]);
            } else *reinterpret_cast<f32x4*>(a.d_feat[k] + n * kHashDeform + 8 * q + 4 * half) = v;
          }
        }
        if (a.amax_bits != nullptr) {             // the largest STORED gradient (the scale the counting form would find)
          float m = 0.0f;
#pragma unroll
          for (int r = 0; r < 12; ++r) m = fmaxf(m, fabsf(acc[r]));
          amax = fmaxf(amax, m * fmaxf(fmaxf(w[0], w[1]), w[2]));
        }
      }
    });
    run<T2t, 4>(wbase, gt2, grad_epi(gt1, a.dzt1, mask.x));
    (void)gt1;
  }
  // one atomic per workgroup (same-address float atomics retire one after the other in L2)
  __shared__ float gscale_part[kThreads / 64];
  gscale_local = wave_sum(gscale_local);
  if (lane == 0) gscale_part[threadIdx.x >> 6] = gscale_local;
  __syncthreads();
  float sum = 0.0f;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) sum += gscale_part[w];
  }
  if (a.sum_ws != nullptr) {               // option "deterministic": the workgroups' sums added in workgroup order
    const float val[1] = {sum};
    float* const out[1] = {a.g_scale};
    ordered_block_sum<1>(val, out, a.sum_ws);
  } else if (threadIdx.x == 0 && sum != 0.0f) atomicAdd(a.g_scale, sum);
  if (a.amax_bits != nullptr) publish_amax_slots(amax, a.amax_bits);
}

// ------------------------------------------------------------------------------------------------ canonical chain
struct CanonArgs {
  const char* packed;
  const __bf16* hash_nat;  // nat [n_pad,32] FP16 from nerf_hash_encode_fwd_nat (nat_dtype 1) at x_c
  const float* t;          // [n]
  const float* dirs;       // [n,3] unit view directions
  int64_t n, n_pad;
  float* rgb; float* sigma;
  __bf16* sin_nat;         // training: [n_pad,64] nat image [hash | tcode] (the sigma-net's input, wgrad operand)
  __bf16* hs1; __bf16* h16; __bf16* denc; __bf16* hc1; __bf16* hc2;
  uint4* mask;
  const float* d_rgb; const float* d_sigma;
  __bf16* dzs1; __bf16* dzs2; __bf16* dzc1; __bf16* dzc2; __bf16* dsmall;
  float* d_feat;           // [n,32]
  unsigned* amax_bits;     // non-null: kAmaxSlots words that max-accumulate the largest |d_feat| as fp32 bits (common.h; the hash scatter's scale)
  float2* grad_lm;         // non-null: INSTEAD of d_feat, level-major gradients [16][n]
};

template <bool TRAIN>
__global__ void __launch_bounds__(kThreads) canon_fwd_kernel(const CanonArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
  for (int i = tid; i < kCanonFwdN * 64; i += kThreads)
    reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(a.packed + kCanonFwd0 * 1024)[i];
  __syncthreads();
  const char* wbase = smem + lane * 16 - kCanonFwd0 * 1024;
  const int64_t n_tiles = a.n_pad / kTile;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t wt = tile * 4 + wave, n = wt * 32 + col;
    const bool live = n < a.n;
    const int64_t nc = live ? n : a.n - 1;
    f16x8 sin[4], denc[2];
    bf16x8 sin_b[4], denc_b[2];
    sin[0] = load_nat16(a.hash_nat, wt, 2, 0, col, half);
    sin[1] = load_nat16(a.hash_nat, wt, 2, 1, col, half);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) sin_b[ks][j] = (__bf16)(float)sin[ks][j];
    {
      float v[2][8];
      f16x8 tc[2];
      bf16x8 tc_b[2];
      time_values<2, false>(a.t[nc], half, v);                      // bias-free network: the pad columns stay zero
      cast_values<2>(v, tc, tc_b);
      sin[2] = tc[0]; sin[3] = tc[1];
      sin_b[2] = tc_b[0]; sin_b[3] = tc_b[1];
      dir_values<2>(a.dirs[nc * 3 + 0], a.dirs[nc * 3 + 1], a.dirs[nc * 3 + 2], half, v);
      cast_values<2>(v, denc, denc_b);
    }
    if constexpr (TRAIN) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) stash_nat(a.sin_nat, wt, 4, ks, col, half, sin_b[ks]);
      stash_nat(a.denc, wt, 2, 0, col, half, denc_b[0]);
      stash_nat(a.denc, wt, 2, 1, col, half, denc_b[1]);
    }
    uint32_t mw[3] = {0, 0, 0};
    auto relu_epi = [&](f16x8* out, __bf16* stash, int layer) {
      return [=, &mw](auto mc, f32x16 acc) {
        constexpr int m = decltype(mc)::value;
        uint32_t bits = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) { bits |= (acc[r] > 0.0f ? 1u : 0u) << r; acc[r] = fmaxf(acc[r], 0.0f); }
        mw[layer] |= bits << (16 * m);
        acc_to_operand16(acc, out[2 * m], out[2 * m + 1]);
        if constexpr (TRAIN) {
          bf16x8 lo, hi;
          acc_to_operand(acc, lo, hi);
          stash_block(stash, wt, 2, m, col, half, lo, hi);
        }
      };
    };
    f16x8 hs1[4], h16[2], hc1[4], hc2[4];
    run16<S1, 4>(wbase, sin, relu_epi(hs1, a.hs1, 0));
    float h0 = 0.0f;
    run16<S2, 4>(wbase, hs1, [&](auto, f32x16 acc) {
      h0 = acc[0];
      acc_to_operand16(acc, h16[0], h16[1]);
      if constexpr (TRAIN) {
        bf16x8 lo, hi;
        acc_to_operand(acc, lo, hi);
        stash_block(a.h16, wt, 1, 0, col, half, lo, hi);
      }
    });
    if (live && half == 0) {
      const float x = h0 - 5.0f;                                     // decoders.py:153
      a.sigma[n] = x > 20.0f ? x : log1pf(expf(x));
    }
    {
      f16x8 cat[3] = {h16[0], denc[0], denc[1]};
      run16<C1, 3>(wbase, cat, relu_epi(hc1, a.hc1, 1));
    }
    run16<C2, 4>(wbase, hc1, relu_epi(hc2, a.hc2, 2));
    run16<C3, 4>(wbase, hc2, [&](auto, f32x16 acc) {
      if (live && half == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a.rgb[n * 3 + c] = 1.0f / (1.0f + __expf(-acc[c]));
      }
    });
    if constexpr (TRAIN) a.mask[tile * kThreads + tid] = make_uint4(mw[0], mw[1], mw[2], 0);
  }
}

__global__ void __launch_bounds__(kThreads) canon_bwd_kernel(const CanonArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
  for (int i = tid; i < kCanonBwdN * 64; i += kThreads)
    reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(a.packed + kCanonBwd0 * 1024)[i];
  __syncthreads();
  const char* wbase = smem + lane * 16 - kCanonBwd0 * 1024;
  const int64_t n_tiles = a.n_pad / kTile;
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
      gs = a.d_sigma[n] * -expm1f(-a.sigma[n]);                      // softplus' = 1 - exp(-softplus)
    }
    bf16x8 small;
#pragma unroll
    for (int j = 0; j < 8; ++j) small[j] = (__bf16)0.0f;
    if (half == 0) { small[0] = (__bf16)g0; small[1] = (__bf16)g1; small[2] = (__bf16)g2; }
    stash_nat(a.dsmall, wt, 1, 0, col, half, small);
    const uint4 mask = a.mask[tile * kThreads + tid];
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
    { bf16x8 in[1] = {small}; run<C3t, 1>(wbase, in, grad_epi(gc2, a.dzc2, mask.z)); }
    run<C2t, 4>(wbase, gc2, grad_epi(gc1, a.dzc1, mask.y));
    run<C1t, 4>(wbase, gc1, [&](auto, f32x16 acc) {
      if (half == 0) acc[0] += gs;                                   // row 0 of h also feeds sigma
      acc_to_operand(acc, g16[0], g16[1]);
      stash_block(a.dzs2, wt, 1, 0, col, half, g16[0], g16[1]);
    });
    { bf16x8 in[1] = {g16[0]}; run<S2t, 1>(wbase, in, grad_epi(gs1, a.dzs1, mask.x)); }
    run<S1t, 4>(wbase, gs1, [&](auto, f32x16 acc) {
      if (live) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
          if (a.grad_lm != nullptr) {                                // levels 4q + 2 half, + 1
            float2* lm = a.grad_lm + (int64_t)(4 * q + 2 * half) * a.n + n;
            lm[0] = make_float2(v[0], v[1]);
            lm[a.n] = make_float2(v[2], v[3]);
          } else *reinterpret_cast<f32x4*>(a.d_feat + n * 32 + 8 * q + 4 * half) = v;
        }
        if (a.amax_bits != nullptr) {
#pragma unroll
          for (int r = 0; r < 16; ++r) amax = fmaxf(amax, fabsf(acc[r]));
        }
      }
    });
  }
  if (a.amax_bits != nullptr) publish_amax_slots(amax, a.amax_bits);
}

// ------------------------------------------------------------------------------------------------ per-sample inputs
// t' and x' of every compacted sample (core.py:289-297): t = times[ray of the sample]; training with use_coord_noise
// adds N(0, std) noise from the counter-based generator, keyed by (seed, counter) and the sample's index in the
// GLOBAL batch (first_sample + g), so data-parallel shards draw what one GPU would
__device__ __forceinline__ void normal_pair(uint64_t counter, uint64_t index, uint64_t key, float& z0, float& z1) {
  const float u1 = fmaxf(squares_uniform(counter, 2 * index, key), 5.9604644775390625e-08f);
  const float u2 = squares_uniform(counter, 2 * index + 1, key);
  const float r = sqrtf(-2.0f * __logf(u1));
  z0 = r * __builtin_amdgcn_sinf(u2 + 0.25f);      // v_sin_f32 takes revolutions: cos(2 pi u2)
  z1 = r * __builtin_amdgcn_sinf(u2);
}

__global__ void __launch_bounds__(256)
prep_kernel(const int* __restrict__ slots, const float* __restrict__ pts, const float* __restrict__ times, int64_t total, int S,
            float std_x, float std_t, uint64_t key, uint64_t counter, uint64_t first_sample,
            float* __restrict__ x_out, float* __restrict__ t_out) {
  for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int slot = slots != nullptr ? slots[g] : (int)g;
    if (slot < 0) continue;
    float t = times[S > 0 ? g / S : g];
    float x[3] = {pts[(int64_t)slot * 3 + 0], pts[(int64_t)slot * 3 + 1], pts[(int64_t)slot * 3 + 2]};
    if (std_x > 0.0f || std_t > 0.0f) {
      float z0, z1, z2, z3;
      normal_pair(counter, 2 * (first_sample + (uint64_t)g), key, z0, z1);
      normal_pair(counter, 2 * (first_sample + (uint64_t)g) + 1, key, z2, z3);
      if (std_x > 0.0f) { x[0] += z0 * std_x; x[1] += z1 * std_x; x[2] += z2 * std_x; }
      if (std_t > 0.0f) t = fminf(fmaxf(t + z3 * std_t, 0.0f), 1.0f);
    }
    if (x_out != nullptr) { x_out[(int64_t)slot * 3 + 0] = x[0]; x_out[(int64_t)slot * 3 + 1] = x[1]; x_out[(int64_t)slot * 3 + 2] = x[2]; }
    t_out[slot] = t;
  }
}

struct Layout {
  int64_t n_pad;
  size_t feat[3], canon_nat, tc, ht1, tm, df, hd1, hd2, dmask, wts, raw, dzt1, dzt2, dzd1, dzd2, dsmall_d, dfeat[3];
  size_t sin_nat, hs1, h16, denc, hc1, hc2, cmask, dzs1, dzs2, dzc1, dzc2, dsmall_c, dfeat_c, sum_ws, slab, total;
};
static Layout layout(int64_t n) {
  Layout s{};
  s.n_pad = (n + kTile - 1) / kTile * kTile;
  const size_t np = (size_t)s.n_pad;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) / 256 * 256; return at; };
  for (int k = 0; k < 3; ++k) s.feat[k] = take(np * 32 * 2);
  s.canon_nat = take(np * 32 * 2);
  s.tc = take(np * 32 * 2); s.ht1 = take(np * 64 * 2); s.tm = take(np * 64 * 2); s.df = take(np * 32 * 2);
  s.hd1 = take(np * 64 * 2); s.hd2 = take(np * 64 * 2); s.dmask = take((np / kTile) * kThreads * 16);
  s.wts = take(np * 16); s.raw = take(np * 12);
  s.dzt1 = take(np * 64 * 2); s.dzt2 = take(np * 64 * 2); s.dzd1 = take(np * 64 * 2); s.dzd2 = take(np * 64 * 2);
  s.dsmall_d = take(np * 16 * 2);
  for (int k = 0; k < 3; ++k) s.dfeat[k] = take(np * kHashDeform * 4);
  s.sin_nat = take(np * 64 * 2); s.hs1 = take(np * 64 * 2); s.h16 = take(np * 32 * 2); s.denc = take(np * 32 * 2);
  s.hc1 = take(np * 64 * 2); s.hc2 = take(np * 64 * 2); s.cmask = take((np / kTile) * kThreads * 16);
  s.dzs1 = take(np * 64 * 2); s.dzs2 = take(np * 32 * 2); s.dzc1 = take(np * 64 * 2); s.dzc2 = take(np * 64 * 2);
  s.dsmall_c = take(np * 16 * 2); s.dfeat_c = take(np * 32 * 4);
  // option "deterministic": ordered sum of the displacement-scale gradient, partial tiles of the weight-gradient launches
  s.sum_ws = take(ordered_sum_ws_words(1) * sizeof(unsigned)); s.slab = take(kSmallSlabBytes);
  s.total = o;
  return s;
}

static DeformArgs deform_args(const void* packed, const float* params, void* ws, int64_t n) {
  const Layout l = layout(n);
  char* w = static_cast<char*>(ws);
  auto B = [&](size_t off) { return reinterpret_cast<__bf16*>(w + off); };
  DeformArgs a{};
  a.packed = static_cast<const char*>(packed);
  a.params = params; a.n = n; a.n_pad = l.n_pad;
  for (int k = 0; k < 3; ++k) { a.feat[k] = B(l.feat[k]); a.d_feat[k] = reinterpret_cast<float*>(w + l.dfeat[k]); }
  a.tc = B(l.tc); a.ht1 = B(l.ht1); a.tm = B(l.tm); a.df = B(l.df); a.hd1 = B(l.hd1); a.hd2 = B(l.hd2);
  a.mask = reinterpret_cast<uint4*>(w + l.dmask);
  a.wts = reinterpret_cast<float*>(w + l.wts); a.raw = reinterpret_cast<float*>(w + l.raw);
  a.dzt1 = B(l.dzt1); a.dzt2 = B(l.dzt2); a.dzd1 = B(l.dzd1); a.dzd2 = B(l.dzd2); a.dsmall = B(l.dsmall_d);
  return a;
}

static CanonArgs canon_args(const void* packed, void* ws, int64_t n) {
  const Layout l = layout(n);
  char* w = static_cast<char*>(ws);
  auto B = [&](size_t off) { return reinterpret_cast<__bf16*>(w + off); };
  CanonArgs a{};
  a.packed = static_cast<const char*>(packed);
  a.n = n; a.n_pad = l.n_pad;
  a.hash_nat = B(l.canon_nat); a.sin_nat = B(l.sin_nat); a.hs1 = B(l.hs1); a.h16 = B(l.h16); a.denc = B(l.denc);
  a.hc1 = B(l.hc1); a.hc2 = B(l.hc2); a.mask = reinterpret_cast<uint4*>(w + l.cmask);
  a.dzs1 = B(l.dzs1); a.dzs2 = B(l.dzs2); a.dzc1 = B(l.dzc1); a.dzc2 = B(l.dzc2); a.dsmall = B(l.dsmall_c);
  a.d_feat = reinterpret_cast<float*>(w + l.dfeat_c);
  return a;
}

static int grid_for(int64_t tiles) {
  int n_cu = 0;
  if (device_cu_count(&n_cu) != NERF_OK) return -1;
  const int64_t cap = (int64_t)n_cu * 4;
  return (int)(tiles < cap ? tiles : cap);
}

}  // namespace p4
}  // namespace nerf

using namespace nerf;
using namespace nerf::p4;

extern "C" int64_t nerf_p4_param_count(void) { return kParams; }
extern "C" size_t nerf_p4_packed_bytes(void) { return kPackBytes; }
extern "C" size_t nerf_p4_workspace_bytes(int64_t n) { return n > 0 ? layout(n).total : 0; }
extern "C" size_t nerf_p4_workspace_offset(int64_t n, int which) {
  if (n <= 0) return 0;
  const Layout l = layout(n);
  switch (which) {
    case 0: case 1: case 2: return l.feat[which];        // nat images of the three deformation grids (hash forward outputs)
    case 3: return l.canon_nat;                            // nat image of the canonical grid at x_c
    case 4: case 5: case 6: return l.dfeat[which - 4];    // d features of the three deformation grids [n,24] fp32
    case 7: return l.dfeat_c;                              // d features of the canonical grid [n,32] fp32
    default: return (size_t)-1;
  }
}

extern "C" int nerf_p4_pack(const float* params_f32, void* packed, nerf_stream_t stream) {
  NERF_REQUIRE(params_f32 && packed && ((uintptr_t)packed & 255) == 0, "nerf_p4_pack: bad pointer");
  hipLaunchKernelGGL(p4::pack_kernel, dim3(32), dim3(256), 0, as_stream(stream), params_f32, static_cast<char*>(packed));
  return check_launch("nerf_p4_pack");
}

extern "C" int nerf_p4_sample_inputs(const int* slot_of_sample, const float* pts_compact, const float* ray_times, int64_t n_rays,
                                     int n_samples, float coord_noise_std, float time_noise_std, uint64_t seed, uint64_t counter,
                                     int64_t first_ray, float* x_deform, float* t_deform, nerf_stream_t stream) {
  NERF_REQUIRE(n_rays >= 0 && n_samples >= 0 && first_ray >= 0 && counter < ((uint64_t)1 << 24), "nerf_p4_sample_inputs: bad sizes");
  const int64_t total = n_samples > 0 ? n_rays * n_samples : n_rays;
  if (total == 0) return NERF_OK;
  NERF_REQUIRE(pts_compact && ray_times && t_deform, "nerf_p4_sample_inputs: NULL pointer");
  NERF_REQUIRE(n_samples == 0 || slot_of_sample != nullptr, "nerf_p4_sample_inputs: slot map is NULL");
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(p4::prep_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), slot_of_sample, pts_compact, ray_times, total,
                     n_samples, coord_noise_std, time_noise_std, squares_key(seed ^ 0x6e6f697365ull), counter,   // its own stream: the depth jitter of
                                                                                                     // nerf_sample_compact_jitter draws from squares_key(seed)
                     (uint64_t)first_ray * (uint64_t)(n_samples > 0 ? n_samples : 1), x_deform, t_deform);
  return check_launch("nerf_p4_sample_inputs");
}

extern "C" int nerf_p4_deform_fwd(const void* packed, const float* params_f32, void* workspace, const float* pts, const float* t_deform,
                                  const float* blend, int64_t n, float* delta_x, float* x_canonical, int train, nerf_stream_t stream) {
  NERF_REQUIRE(n >= 0, "nerf_p4_deform_fwd: n=%lld", (long long)n);
  if (n == 0) return NERF_OK;
  NERF_REQUIRE(packed && params_f32 && workspace && pts && t_deform && delta_x && x_canonical && ((uintptr_t)workspace & 255) == 0,
               "nerf_p4_deform_fwd: bad pointer");
  DeformArgs a = deform_args(packed, params_f32, workspace, n);
  a.t = t_deform; a.blend = blend; a.x = pts; a.dx = delta_x; a.xc = x_canonical;
  const int grid = grid_for(a.n_pad / kTile);
  if (grid <= 0) return fail(NERF_ELAUNCH, "nerf_p4_deform_fwd: cannot query device");
  const int lds = kDeformFwdN * 1024 + 256;
  if (train) hipLaunchKernelGGL(p4::deform_fwd_kernel<true>, dim3(grid), dim3(kThreads), lds, as_stream(stream), a);
  else hipLaunchKernelGGL(p4::deform_fwd_kernel<false>, dim3(grid), dim3(kThreads), lds, as_stream(stream), a);
  return check_launch("nerf_p4_deform_fwd");
}

extern "C" int nerf_p4_canon_fwd(const void* packed, void* workspace, const float* t_deform, const float* dirs, int64_t n, float* rgb,
                                 float* sigma, int train, nerf_stream_t stream) {
  NERF_REQUIRE(n >= 0, "nerf_p4_canon_fwd: n=%lld", (long long)n);
  if (n == 0) return NERF_OK;
  NERF_REQUIRE(packed && workspace && t_deform && dirs && rgb && sigma && ((uintptr_t)workspace & 255) == 0, "nerf_p4_canon_fwd: bad pointer");
  CanonArgs a = canon_args(packed, workspace, n);
  a.t = t_deform; a.dirs = dirs; a.rgb = rgb; a.sigma = sigma;
  const int grid = grid_for(a.n_pad / kTile);
  if (grid <= 0) return fail(NERF_ELAUNCH, "nerf_p4_canon_fwd: cannot query device");
  if (train) hipLaunchKernelGGL(p4::canon_fwd_kernel<true>, dim3(grid), dim3(kThreads), kCanonFwdN * 1024, as_stream(stream), a);
  else hipLaunchKernelGGL(p4::canon_fwd_kernel<false>, dim3(grid), dim3(kThreads), kCanonFwdN * 1024, as_stream(stream), a);
  return check_launch("nerf_p4_canon_fwd");
}

static WgradJob make_job(const char* w, size_t a_off, int a_bytes, int mt_a, size_t b_off, int nt_acc, size_t bn_off, int nt_nat, int kind) {
  WgradJob j{};
  j.a = w + a_off; j.a_bytes = a_bytes; j.mt_a = mt_a;
  if (nt_acc) { j.b_acc = w + b_off; j.b_acc_bytes = nt_acc * 2048; j.nt_acc = nt_acc; }
  if (nt_nat) { j.b_nat = w + bn_off; j.b_nat_bytes = nt_nat * 2048; j.nt_nat = nt_nat; }
  j.bias_nat_col = -1; j.kind = kind;
  return j;
}

// Backward of the canonical chain: d_feat (workspace slot 7) = d loss / d hash features; the five weight gradients are
// ACCUMULATED into grads_f32 (the caller zeroes the vector once per step: several passes -- data batch, regulariser
// probes -- add into it)
extern "C" int nerf_p4_canon_bwd(const void* packed, void* workspace, const float* rgb, const float* sigma, const float* d_rgb,
                                 const float* d_sigma, int64_t n, float* grads_f32, void* amax_bits, void* grad_lm, nerf_stream_t stream) {
  NERF_REQUIRE(n >= 0 && grads_f32, "nerf_p4_canon_bwd: bad arguments");
  if (n == 0) return NERF_OK;
  NERF_REQUIRE(packed && workspace && rgb && sigma && d_rgb && d_sigma, "nerf_p4_canon_bwd: NULL pointer");
  CanonArgs a = canon_args(packed, workspace, n);
  a.rgb = const_cast<float*>(rgb); a.sigma = const_cast<float*>(sigma); a.d_rgb = d_rgb; a.d_sigma = d_sigma;
  a.amax_bits = static_cast<unsigned*>(amax_bits); a.grad_lm = static_cast<float2*>(grad_lm);
  const int grid = grid_for(a.n_pad / kTile);
  if (grid <= 0) return fail(NERF_ELAUNCH, "nerf_p4_canon_bwd: cannot query device");
  hipLaunchKernelGGL(p4::canon_bwd_kernel, dim3(grid), dim3(kThreads), kCanonBwdN * 1024, as_stream(stream), a);
  if (int rc = check_launch("nerf_p4_canon_bwd (dgrad)"); rc != NERF_OK) return rc;
  const Layout l = layout(n);
  const char* w = static_cast<const char*>(workspace);
  WgradArgs wa{};
  { WgradJob j = make_job(w, l.dzs1, 4096, 2, 0, 0, l.sin_nat, 2, 12); j.w_off = kS1; j.w_ld = 64; j.o_valid = 64; j.nat_valid = 32 + kTimeDim; wa.jobs[0] = j; }
  { WgradJob j = make_job(w, l.dzs2, 2048, 1, l.hs1, 2, 0, 0, 7); j.w_off = kS2; j.w_ld = 64; j.o_valid = 16; j.acc_valid = 64; wa.jobs[1] = j; }
  { WgradJob j = make_job(w, l.dzc1, 4096, 2, l.h16, 1, l.denc, 1, 8); j.w_off = kC1; j.w_ld = 48; j.o_valid = 64; j.acc_valid = 16; j.nat_valid = kDirDim; j.nat_col0 = 16; wa.jobs[2] = j; }
  { WgradJob j = make_job(w, l.dzc2, 4096, 2, l.hc1, 2, 0, 0, 7); j.w_off = kC2; j.w_ld = 64; j.o_valid = 64; j.acc_valid = 64; wa.jobs[3] = j; }
  { WgradJob j = make_job(w, l.dsmall_c, 1024, 1, l.hc2, 2, 0, 0, 9); j.a_nat = 1; j.split_n = 1; j.w_off = kC3; j.w_ld = 64; j.o_valid = 3; j.acc_valid = 64; wa.jobs[4] = j; }
  wa.n_jobs = 5;
  if (options().deterministic)
    return wgrad_launch(wa, n, grads_f32, as_stream(stream), reinterpret_cast<float*>(const_cast<char*>(w) + l.slab), kSmallSlabBytes);
  return wgrad_launch(wa, n, grads_f32, as_stream(stream));
}

// Backward of the deformation chain from d loss / d delta_x: d features of the three grids (workspace slots 4..6, already
// multiplied by the blend weights), the time-modulation and displacement-decoder weight gradients and the gradient of
// displacement_scale ACCUMULATED into grads_f32
extern "C" int nerf_p4_deform_bwd(const void* packed, const float* params_f32, void* workspace, const float* d_delta_x, int64_t n,
                                  float* grads_f32, void* amax_bits, void* grad_lm, nerf_stream_t stream) {
  NERF_REQUIRE(n >= 0 && grads_f32, "nerf_p4_deform_bwd: bad arguments");
  if (n == 0) return NERF_OK;
  NERF_REQUIRE(packed && params_f32 && workspace && d_delta_x, "nerf_p4_deform_bwd: NULL pointer");
  DeformArgs a = deform_args(packed, params_f32, workspace, n);
  a.d_dx = d_delta_x; a.g_scale = grads_f32 + kScale; a.amax_bits = static_cast<unsigned*>(amax_bits);
  a.grad_lm = static_cast<float2*>(grad_lm);
  const int grid = grid_for(a.n_pad / kTile);
  if (grid <= 0) return fail(NERF_ELAUNCH, "nerf_p4_deform_bwd: cannot query device");
  const bool det = options().deterministic != 0;
  const Layout l = layout(n);
  if (det) {
    NERF_REQUIRE(grid <= kOrderedSumMaxBlocks, "nerf_p4_deform_bwd: %d workgroups (ordered sum: at most %d)", grid, kOrderedSumMaxBlocks);
    a.sum_ws = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + l.sum_ws);
    if (hipMemsetAsync(a.sum_ws, 0, sizeof(unsigned) * kOrderedSumTickets, as_stream(stream)) != hipSuccess)
      return fail(NERF_ELAUNCH, "nerf_p4_deform_bwd: memset failed");
  }
  hipLaunchKernelGGL(p4::deform_bwd_kernel, dim3(grid), dim3(kThreads), kDeformBwdN * 1024, as_stream(stream), a);
  if (int rc = check_launch("nerf_p4_deform_bwd (dgrad)"); rc != NERF_OK) return rc;
  const char* w = static_cast<const char*>(workspace);
  WgradArgs wa{};
  { WgradJob j = make_job(w, l.dzt1, 4096, 2, 0, 0, l.tc, 1, 6); j.w_off = kT1W; j.w_ld = 21; j.o_valid = 64; j.nat_valid = kTimeDim;
    j.bias_off = kT1b; j.bias_nat_col = kTimeDim; wa.jobs[0] = j; }
  { WgradJob j = make_job(w, l.dzt2, 4096, 2, l.ht1, 2, 0, 0, 10); j.ones = 1; j.w_off = kT2W; j.w_ld = 64; j.o_valid = 64; j.acc_valid = 64;
    j.bias_off = kT2b; wa.jobs[1] = j; }
  { WgradJob j = make_job(w, l.dzd1, 4096, 2, l.tm, 2, l.df, 1, 11); j.w_off = kD1; j.w_ld = 96; j.o_valid = 64; j.acc_valid = 64;
    j.acc_col0 = kHashDeform; j.nat_valid = kHashDeform; j.nat_col0 = 0; wa.jobs[2] = j; }
  { WgradJob j = make_job(w, l.dzd2, 4096, 2, l.hd1, 2, 0, 0, 7); j.w_off = kD2; j.w_ld = 64; j.o_valid = 64; j.acc_valid = 64; wa.jobs[3] = j; }
  { WgradJob j = make_job(w, l.dsmall_d, 1024, 1, l.hd2, 2, 0, 0, 9); j.a_nat = 1; j.split_n = 1; j.w_off = kD3; j.w_ld = 64; j.o_valid = 3; j.acc_valid = 64; wa.jobs[4] = j; }
  wa.n_jobs = 5;
  if (det) return wgrad_launch(wa, n, grads_f32, as_stream(stream), reinterpret_cast<float*>(static_cast<char*>(workspace) + l.slab), kSmallSlabBytes);
  return wgrad_launch(wa, n, grads_f32, as_stream(stream));
}
