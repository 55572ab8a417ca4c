// Shared helpers for the gfx950 kernels (error reporting, launch checks, wave ops).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/nerf_hip.h"

namespace nerf {

constexpr int kWave = 64;  // CDNA wavefront width

int fail(int code, const char* fmt, ...);  // records nerf_last_error(), returns code

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(NERF_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return NERF_OK;
}

inline hipStream_t as_stream(nerf_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Development options (api.cpp): defaults come from the environment, read ONCE per process
// (NERF_CHAIN_LEGACY, NERF_WGRAD_DEBUG, ...); nerf_set_option() changes them afterwards.
struct Options {
  int chain_legacy = 0;          // compiler-scheduled chain kernels instead of the asm streams
  int fwd_cycles = 0;            // print shader cycles per pass of the forward stream kernel
  int wgrad_overhead = 98304;    // span cost model, bf16 images: fixed share per ring iteration (bytes)
  int wgrad_bw_x16 = 192;        // span cost model, 8-bit images: DMA bytes per 16 shader cycles and CU
  int wgrad_fixed = 2000;        // ... and the fixed cycles per ring iteration
  int wgrad_debug = 0;           // skeleton timing: 1 no compute, 2 no DMA, 4 no flush
  int wgrad_small_span = 32;     // tiny-MLP wgrad kernel: least wave tiles per workgroup span
  int wgrad_small_cap = 3;       // ... and most workgroups per CU
  int wgrad_only = -1;           // keep one job kind
  int hash_bwd_only_level = -1;  // time one level's atomics
  int wgrad_big_only = 0;        // 1: the Instant tiny-MLP weight gradients on the decoder's one-workgroup-per-CU kernel (A/B)
  int wgrad_k16 = 0;             // 1: the K = 16 8-bit MFMA in the owner-mode wgrad jobs (A/B against the K = 64 form)
  int wgrad_atomic = 0;          // 1: flush the split-K partial sums with float atomics at every size (A/B)
  int hash_bwd_atomic = 0;       // 1: the atomic form of the hash scatter even when a workspace is given (A/B)
  int infer_shape32 = 0;         // inference on the 32x32x16 MFMA stream instead of the 16x16x32 one (A/B)
  int infer64 = 1;               // inference with 64 samples per wave, four waves per workgroup (activations in AGPRs): +2 %, the same bits; 0: A/B
  int stash_fp8 = 0;             // 1: 8-bit training images (e4m3 / e5m2) in the asm-stream family instead of bf16
  int chain_grid = 0;            // > 0: cap the chain kernels' workgroup count (timing below the power limit; CU partition with wgrad_grid)
  int wgrad_grid = 0;            // > 0: cap the decoder weight-gradient kernel's workgroup count (two half-batches in flight)
  int composite_wgs_per_cu = 2;  // fused compositing + loss backward: workgroups per CU.  Every workgroup ends with same-address atomics (loss, regulariser,
                                 // maximum), which retire one after the other in L2: 8 per CU 58 us, 4: 35, 2: 27, 1: 29 (8192 rays x 64, Part 4 step)
  int hash_xcd = 1;              // 1: hash-grid gather kernels launch XCD-aware (levels x and x + 8 on XCD x); 0: level-major 2-D launch (A/B)
  int hash_fwd_lds_kb = 36;      // dynamic LDS per hash-forward workgroup (occupancy throttle, see nerf_hash_encode_fwd); 0: none
  int tv_blocks = 0;             // > 0: workgroups (of 1024 threads) of the TV + squared-norm pass; 0: one per CU (A/B)
  int deterministic = 0;         // 1: every sum whose order depends on scheduling takes an ordered form -- compaction slots in sample order,
                                 // tiny-MLP weight gradients through partial tiles, hash bins never cut, d x from the hash grid level by level,
                                 // scalar sums (loss, regulariser, displacement-scale gradient) per workgroup and in workgroup order: two runs
                                 // of the same step give the same bits (reference semantics: a plain sum, run.py:1941-1944)
};
Options& options();

// Per-device launch facts (api.cpp), keyed by the calling thread's current HIP device.
int device_cu_count(int* n_cu);
int ensure_dynamic_lds(const void* kernel, int bytes, const char* what);

#define NERF_REQUIRE(cond, ...) \
  do {                          \
    if (!(cond)) return ::nerf::fail(NERF_EINVAL, __VA_ARGS__); \
  } while (0)

// Counter-based generator ("squares", Widynski 2020: four rounds of squaring a 64-bit counter x key): every
// (step, element) pair owns its draw, so one kernel can draw the batch's pixels AND the stratified jitter
// without any state -- the reference draws them with torch.randint / torch.rand (dataset.py:147-150,
// renderer.py:198); the distributions are the same, the streams are not.
__device__ __forceinline__ uint32_t squares32(uint64_t ctr, uint64_t key) {
  uint64_t x = ctr * key, y = x, z = y + key;
  x = x * x + y; x = (x >> 32) | (x << 32);
  x = x * x + z; x = (x >> 32) | (x << 32);
  x = x * x + y; x = (x >> 32) | (x << 32);
  return (uint32_t)((x * x + z) >> 32);
}

// key of the generator from a user seed: splitmix64 of the seed, forced odd
inline uint64_t squares_key(uint64_t seed) {
  uint64_t key = seed + 0x9E3779B97F4A7C15ull;
  key = (key ^ (key >> 30)) * 0xBF58476D1CE4E5B9ull;
  key = (key ^ (key >> 27)) * 0x94D049BB133111EBull;
  return (key ^ (key >> 31)) | 1ull;
}
// uniform in [0, 1) with 24 bits: draw `index` of step `counter` (< 2^24; index < 2^40)
__device__ __forceinline__ float squares_uniform(uint64_t counter, uint64_t index, uint64_t key) {
  return (float)(squares32((counter << 40) + index, key) >> 8) * 5.9604644775390625e-08f;
}

// ---- individually rounded fp32 ops: never contracted into FMAs (hipcc defaults to
// -ffp-contract=fast; the reference's eager ops round after every multiply and add) ----
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}

// ---- wave-level scans on DPP (row_shr within 16-lane rows, then readlane across rows) ----
// v_mov_dpp row_shr:n  -> lane i receives lane i-n of its 16-lane row, `identity` when i-n < 0.
template <int N>
__device__ __forceinline__ float dpp_row_shr(float v, float identity) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity),
                                                              __builtin_bit_cast(int, v),
                                                              0x110 + N, 0xF, 0xF, false));
}
template <int N>
__device__ __forceinline__ float dpp_row_shl(float v, float identity) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity),
                                                              __builtin_bit_cast(int, v),
                                                              0x100 + N, 0xF, 0xF, false));
}
__device__ __forceinline__ float lane_read(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// inclusive product over the 64 lanes, in lane order
__device__ __forceinline__ float wave_inclusive_prod(float v) {
  v *= dpp_row_shr<1>(v, 1.0f);
  v *= dpp_row_shr<2>(v, 1.0f);
  v *= dpp_row_shr<4>(v, 1.0f);
  v *= dpp_row_shr<8>(v, 1.0f);
  const int lane = __lane_id();
  const float r0 = lane_read(v, 15), r1 = lane_read(v, 31), r2 = lane_read(v, 47);
  const float p1 = r0, p2 = r0 * r1, p3 = p2 * r2;
  const int row = lane >> 4;
  const float carry = row == 0 ? 1.0f : (row == 1 ? p1 : (row == 2 ? p2 : p3));
  return v * carry;
}

// inclusive suffix sum: lane i gets sum_{k>=i} v_k
__device__ __forceinline__ float wave_inclusive_suffix_sum(float v) {
  v += dpp_row_shl<1>(v, 0.0f);
  v += dpp_row_shl<2>(v, 0.0f);
  v += dpp_row_shl<4>(v, 0.0f);
  v += dpp_row_shl<8>(v, 0.0f);
  const int lane = __lane_id();
  const float r1 = lane_read(v, 16), r2 = lane_read(v, 32), r3 = lane_read(v, 48);
  const float s3 = r3, s2 = r3 + r2, s1 = s2 + r1;
  const int row = lane >> 4;
  const float carry = row == 3 ? 0.0f : (row == 2 ? s3 : (row == 1 ? s2 : s1));
  return v + carry;
}

// exclusive suffix sum: lane i gets sum_{k>i} v_k (shifted scan: no `inclusive - own`
// cancellation when a lane's own term dwarfs everything behind it)
__device__ __forceinline__ float wave_exclusive_suffix_sum(float v) {
  const float incl = wave_inclusive_suffix_sum(v);
  float ex = dpp_row_shl<1>(incl, 0.0f);
  const float n16 = lane_read(incl, 16), n32 = lane_read(incl, 32), n48 = lane_read(incl, 48);
  const int lane = __lane_id();
  if (lane == 15) ex = n16;
  if (lane == 31) ex = n32;
  if (lane == 47) ex = n48;
  return ex;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ---- ordered sum of V values per workgroup of a 1-D launch (instead of V same-address float atomics per workgroup) ----
// Every workgroup publishes its values (thread 0 passes them) and takes a ticket; the workgroup that draws the last
// ticket adds all partials IN WORKGROUP ORDER and accumulates the totals into (accumulate false: stores them to) *out[v] (null:
// skipped) -- the same bits whatever the order the workgroups finished in.  Tickets in two stages: workgroup b takes one of group
// b % 32, the last of a group takes one of the master ticket -- 1024 workgroups that finish together queue 32 deep on 33 addresses
// instead of 1024 deep on one (same-address atomics retire one after the other, ~20 ns each: 5 us at the tail of a 25-us launch on
// 256 workgroups).  ws: kOrderedSumTickets ticket words (zero before the first launch; the last workgroups leave them zero again),
// then V * gridDim.x partials.  Every access to ws is a returning device-scope atomic (coherent across the XCDs' L2s without a
// cache write-back); a ticket is taken only after the atomics before it have returned.
// Called by all threads of the workgroup (blockDim.x >= 64).
constexpr int kOrderedSumMaxBlocks = 4096, kOrderedSumGroups = 32, kOrderedSumTickets = 1 + kOrderedSumGroups;
inline size_t ordered_sum_ws_words(int n_values) { return kOrderedSumTickets + (size_t)n_values * kOrderedSumMaxBlocks; }
template <int V>
__device__ __forceinline__ void ordered_block_sum(const float (&val)[V], float* const (&out)[V], unsigned* ws, bool accumulate = true) {
  __shared__ unsigned last_block;
  const unsigned B = gridDim.x;
  unsigned* part = ws + kOrderedSumTickets;
  if (threadIdx.x == 0) {
    unsigned seen = 0;
#pragma unroll
    for (int v = 0; v < V; ++v) seen |= atomicExch(&part[v * B + blockIdx.x], __float_as_uint(val[v]));
    asm volatile("" ::"v"(seen) : "memory");      // the exchanges have returned: the partials sit at their coherent home
    const unsigned g = blockIdx.x % kOrderedSumGroups, in_group = (B - g + kOrderedSumGroups - 1) / kOrderedSumGroups;
    unsigned last = 0u;
    if (atomicAdd(&ws[1 + g], 1u) == in_group - 1) {                   // last of its group
      atomicExch(&ws[1 + g], 0u);
      const unsigned groups = B < (unsigned)kOrderedSumGroups ? B : (unsigned)kOrderedSumGroups;
      last = atomicAdd(&ws[0], 1u) == groups - 1 ? 1u : 0u;
    }
    last_block = last;
  }
  __syncthreads();
  if (last_block == 0u || threadIdx.x >= 64) return;
#pragma unroll
  for (int v = 0; v < V; ++v) {
    float s = 0.0f;
    for (unsigned b = threadIdx.x; b < B; b += 64) s += __uint_as_float(atomicOr(&part[v * B + b], 0u));
    s = wave_sum(s);
    if (threadIdx.x == 0 && out[v] != nullptr) *out[v] = accumulate ? *out[v] + s : s;   // !accumulate: no zeroing launch before the call
  }
  if (threadIdx.x == 0) atomicExch(&ws[0], 0u);
}

// ---- largest |value| of a launch, for a consumer in a LATER launch: kAmaxSlots words instead of one.  A workgroup max-accumulates
// its value (fp32 bits) into word (workgroup & 31): 1024 workgroups that finish together queue 32 deep on 32 addresses instead of
// 1024 deep on one (same-address atomics retire one after the other, ~20 ns each: 8 us at the tail of a 10-us kernel).  The
// consumer takes the maximum of all words.  Called by all threads of the workgroup; amax: the thread's own maximum (>= 0).
constexpr int kAmaxSlots = 32;
static_assert(kAmaxSlots == NERF_AMAX_WORDS, "include/nerf_hip.h");
__device__ __forceinline__ void publish_amax_slots(float amax, unsigned* __restrict__ slots) {
  __shared__ unsigned wg_amax;
  if (threadIdx.x == 0) wg_amax = 0;
  __syncthreads();
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
  if ((threadIdx.x & 63) == 0 && amax > 0.0f && amax <= 3.0e38f) atomicMax(&wg_amax, __builtin_bit_cast(unsigned, amax));
  __syncthreads();
  unsigned* slot = slots + (blockIdx.x & (kAmaxSlots - 1));
  if (threadIdx.x == 0 && wg_amax > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, wg_amax);
}

}  // namespace nerf
