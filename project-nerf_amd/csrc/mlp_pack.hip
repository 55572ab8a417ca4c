// fp32 reference-layout parameters -> MFMA-fragment-ordered bf16 weight streams
// (forward stream, transposed stream for dgrad) + fp32 bias table.  See mlp_plan.h.
#include "common.h"
#include "mlp_plan.h"

namespace nerf {
using namespace plan;

struct FragTable {
  // per stream fragment: step kind and (m-tile, k-step) packed as kind<<16 | mt<<8 | ks
  int v[kFwdFrags + kBwdFrags];
};

constexpr FragTable make_frag_table() {
  FragTable t{};
  int f = 0;
  for (int kind = 0; kind < kNumKinds; ++kind) {
    const int ks = step_ks(kind);
    for (int m = 0; m < step_of(kind).mt; ++m)
      for (int k = 0; k < ks; ++k) t.v[f++] = (kind << 16) | (m << 8) | k;
  }
  return t;
}
__constant__ FragTable g_frag_table = make_frag_table();

// flat parameter index feeding A[row][k] of a step, or -1 for structural zeros
__device__ __forceinline__ int src_index(int kind, int row, int k, bool nat) {
  if (kind <= F_PTS7) {
    const int l = kind;
    const int in_dim = pts_in_dim(l);
    int col;
    if (l == 0) col = k < 63 ? k : -1;                      // all natural (Fourier code)
    else if (l == 4) col = nat ? (k < 63 ? 256 + k : -1) : k;
    else col = k;
    return col < 0 ? -1 : pts_weight_off(l) + row * in_dim + col;
  }
  switch (kind) {
    case F_HEAD: return row < 256 ? kWFeat + row * 256 + k : (row == 256 ? kWSigma + k : -1);
    case F_VIEW: return nat ? (k < 27 ? kWView + row * 283 + 256 + k : -1) : kWView + row * 283 + k;
    case F_RGB: return row < 3 ? kWRgb + row * 128 + k : -1;
    case B_RGB: return k < 3 ? kWRgb + k * 128 + row : -1;
    case B_VIEW: return kWView + k * 283 + row;
    case B_HEAD: return nat ? (k == 0 ? kWSigma + row : -1) : kWFeat + k * 256 + row;
    default: {
      const int l = 7 - (kind - B_PTS7);                    // B_PTS7..B_PTS1
      return pts_weight_off(l) + k * pts_in_dim(l) + row;
    }
  }
}

// forward stream in the operand layout of v_mfma_f32_16x16x32_bf16: lane (q = lane >> 4, r = lane & 15) of
// fragment k of m-tile mt holds A[row 32 mt + 16 (k & 1) + r][8 elements of the 32-deep k-step kk = k >> 1]:
// natural steps element j = column 32 kk + 8 q + j; accumulator-fed steps j = column 32 kk + 16 (j >> 2) + 4 q + (j & 3)
// (the order in which two stacked 16x16 accumulator tiles become the next B operand)
__global__ void __launch_bounds__(256) pack16_kernel(const float* __restrict__ params, __bf16* __restrict__ packed_fwd16) {
  const int total = kFwdFrags * 64;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int frag = t >> 6, lane = t & 63;
    const int desc = g_frag_table.v[frag];
    const int kind = desc >> 16, mt = (desc >> 8) & 0xFF, ks = desc & 0xFF;
    const Step st = step_of(kind);
    const int row = mt * 32 + 16 * (ks & 1) + (lane & 15), q = lane >> 4;
    const bool nat = ks >= st.ks_acc;
    const int kk = nat ? (ks - st.ks_acc) >> 1 : ks >> 1;
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    bf16x8 out;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = nat ? 32 * kk + 8 * q + j : 32 * kk + 16 * (j >> 2) + 4 * q + (j & 3);
      const int src = src_index(kind, row, k, nat);
      out[j] = (__bf16)(src >= 0 ? params[src] : 0.0f);
    }
    *reinterpret_cast<bf16x8*>(packed_fwd16 + (size_t)frag * 512 + lane * 8) = out;
  }
}

__global__ void __launch_bounds__(256)
pack_kernel(const float* __restrict__ params, __bf16* __restrict__ packed_fwd,
            __bf16* __restrict__ packed_bwd, float* __restrict__ bias) {
  const int total = (kFwdFrags + kBwdFrags) * 64;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int frag = t >> 6, lane = t & 63;
    const int desc = g_frag_table.v[frag];
    const int kind = desc >> 16, mt = (desc >> 8) & 0xFF, ks = desc & 0xFF;
    const Step st = step_of(kind);
    const int row = mt * 32 + (lane & 31), h = lane >> 5;
    const bool nat = ks >= st.ks_acc;
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    bf16x8 out;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int k;
      if (nat) k = 16 * (ks - st.ks_acc) + 8 * h + j;
      else k = 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
      const int src = src_index(kind, row, k, nat);
      out[j] = (__bf16)(src >= 0 ? params[src] : 0.0f);
    }
    __bf16* dst = frag < kFwdFrags ? packed_fwd + (size_t)frag * 512 : packed_bwd + (size_t)(frag - kFwdFrags) * 512;
    *reinterpret_cast<bf16x8*>(dst + lane * 8) = out;
  }
  // fp32 bias table
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < kBiasFloats; i += gridDim.x * blockDim.x) {
    float b = 0.0f;
    if (i < 2048) b = params[pts_bias_off(i >> 8) + (i & 255)];
    else if (i < 2048 + 288) { const int r = i - 2048; b = r < 256 ? params[kBFeat + r] : (r == 256 ? params[kBSigma] : 0.0f); }
    else if (i < 2048 + 288 + 128) b = params[kBView + (i - 2048 - 288)];
    else { const int r = i - 2048 - 288 - 128; b = r < 3 ? params[kBRgb + r] : 0.0f; }
    bias[i] = b;
  }
}

}  // namespace nerf

extern "C" size_t nerf_mlp_packed_bytes(void) { return nerf::plan::kPackBytes; }

extern "C" int nerf_mlp_pack_streams(const float* params_f32, void* packed, int which, nerf_stream_t stream) {
  using namespace nerf;
  NERF_REQUIRE(params_f32 && packed, "nerf_mlp_pack: NULL pointer");
  NERF_REQUIRE(((uintptr_t)packed & 255) == 0, "nerf_mlp_pack: packed buffer must be 256-byte aligned");
  NERF_REQUIRE(which >= 1 && which <= 3, "nerf_mlp_pack_streams: which=%d (1 training streams + biases, 2 inference stream, 3 all)", which);
  char* base = static_cast<char*>(packed);
  if (which & 1)
    hipLaunchKernelGGL(pack_kernel, dim3(512), dim3(256), 0, as_stream(stream), params_f32,
                       reinterpret_cast<__bf16*>(base + plan::kPackFwdOff),
                       reinterpret_cast<__bf16*>(base + plan::kPackBwdOff),
                       reinterpret_cast<float*>(base + plan::kPackBiasOff));
  if (which & 2)
    hipLaunchKernelGGL(pack16_kernel, dim3(304), dim3(256), 0, as_stream(stream), params_f32,
                       reinterpret_cast<__bf16*>(base + plan::kPackFwd16Off));
  return check_launch("nerf_mlp_pack");
}

extern "C" int nerf_mlp_pack(const float* params_f32, void* packed, nerf_stream_t stream) {
  return nerf_mlp_pack_streams(params_f32, packed, 3, stream);
}
