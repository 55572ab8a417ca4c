// Building blocks of the decoder chain kernels (forward and dgrad): LDS weight ring fed by
// global_load_lds_dwordx4, the per-m-tile MFMA loop, accumulator -> bf16 operand conversion,
// and the in-register Fourier codes.  See mlp_plan.h for the data layout.
#pragma once
#include <type_traits>
#include <utility>
#include "common.h"
#include "mlp_plan.h"

namespace nerf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kChainThreads = 512;                 // 8 waves, 2 per SIMD
constexpr int kWaveSamples = 32;                   // one 32-column MFMA tile per wave
constexpr int kTileSamples = 8 * kWaveSamples;     // 256 samples per workgroup pass
constexpr int kRingSlotBytes = plan::kChunkFrags * 1024;
constexpr int kRingBytes = 2 * kRingSlotBytes;     // 128 KiB
// LDS map: [bias table | ring slot 0 | ring slot 1].  The bias table sits first so that its
// reads are `small base + 16-bit immediate`; ring reads carry the slot in the address VGPR.
constexpr int kBiasLdsBytes = ((plan::kBiasFloats * 4 + 1023) / 1024) * 1024;
constexpr int kChainLds = kBiasLdsBytes + kRingBytes;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// ---------------------------------------------------------------------------
// Weight ring.  Chunk c of the stream lives in slot (phase & 1); `advance<C>()`
// is called right before the first m-tile of chunk C is consumed:
//   barrier  -> every wave's DMA for chunk C has landed (hipcc drains vmcnt before
//               __syncthreads) and every wave is done reading the other slot
//   issue    -> DMA chunk C+1 (or chunk 0 of the next pass) into the other slot
// ---------------------------------------------------------------------------
template <bool BWD>
struct WeightRing {
  static constexpr const plan::Chunks& chunks() { return BWD ? plan::kBwdChunks : plan::kFwdChunks; }
  const char* stream;   // packed fragment stream in global memory
  char* lds;            // ring base
  int slot;             // slot holding the chunk being consumed
  int wave, lane;

  __device__ __forceinline__ void init(const char* s, char* l, int w, int ln) {
    stream = s; lds = l; slot = 1; wave = w; lane = ln;
  }
  template <int C>
  __device__ __forceinline__ void issue(int dst_slot) const {
    constexpr int frag0 = chunks().chunk_frag0[C];
    constexpr int count = chunks().chunk_count[C];
    // wave-uniform source base (SGPR pair) + per-lane 32-bit offset; the empty asm keeps the
    // compiler from hoisting ~160 loop-invariant 64-bit addresses out of the tile loop
    const char* sbase = stream;
    asm volatile("" : "+s"(sbase));
    sbase += (size_t)(frag0 + wave) * 1024;
    char* dst = lds + dst_slot * kRingSlotBytes + wave * 1024;
    const uint32_t voff = (uint32_t)lane * 16u;
    // every wave issues ceil(count/8) pieces unconditionally: the tail pieces read the next
    // chunk's fragments (the stream is padded by 8 KiB) into unused slot space
#pragma unroll
    for (int i = 0; i < (count + 7) / 8; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(sbase + i * 8192 + voff), (lptr_t)(dst + i * 8192), 16, 0, 0);
  }
  __device__ __forceinline__ void prologue() { issue<0>(0); }
  // Returns the LDS address this lane reads its A fragments of chunk C from.
  // STORES = vector-memory instructions (stash stores) this wave issued since it issued the
  // DMA of chunk C: vmcnt retires in issue order, so vmcnt(STORES) proves the DMA has landed
  // without draining the (slow) stores.  Under-counting is safe, over-counting is not.
  template <int C, int STORES>
  __device__ __forceinline__ const char* advance(bool more_passes) {
    constexpr int n = chunks().n_chunks;
    static_assert(STORES >= 0 && STORES < 48, "vmcnt immediate");
    __builtin_amdgcn_sched_barrier(0);   // also bounds the scheduler's regions (compile time)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STORES) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    slot ^= 1;
    // (ablation on MI355X, 65536x128 samples: skipping these DMA issues -> 1476 TFLOP/s vs 1253;
    //  skipping the wait changes nothing: the ~77-cycle issue cost of each 1-KiB LDS-DMA piece,
    //  64 pieces per 4096-cycle chunk per CU, is what the ring costs, not its latency)
    if constexpr (C + 1 < n) issue<C + 1>(slot ^ 1);
    else if (more_passes) issue<0>(slot ^ 1);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return lds + slot * kRingSlotBytes + lane * 16;
  }
};

// ---------------------------------------------------------------------------
// One 32-row output tile: acc = bias; acc += A[m][ks] * B[ks] over all k-steps.
// ---------------------------------------------------------------------------
// A fragments are read kAhead k-steps ahead of the MFMA that consumes them; the
// sched_group_barrier sequence pins that software pipeline (hipcc otherwise serialises
// ds_read -> wait -> mfma on one fragment register when VGPRs are tight).
constexpr int kAhead = 3;
template <int KS>
__device__ __forceinline__ f32x16 mtile_mfma(const char* a_base, int frag_off, const bf16x8 (&b)[KS], f32x16 acc) {
  // explicit rotating window of kAhead fragments, in program order: read k+kAhead, then MFMA k
  constexpr int D = KS < kAhead ? KS : kAhead;
  bf16x8 win[D];
#pragma unroll
  for (int i = 0; i < D; ++i) win[i] = *reinterpret_cast<const bf16x8*>(a_base + (frag_off + i) * 1024);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const bf16x8 cur = win[ks % D];
    if (ks + D < KS) win[ks % D] = *reinterpret_cast<const bf16x8*>(a_base + (frag_off + ks + D) * 1024);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur, b[ks], acc, 0, 0, 0);
  }
  return acc;
}

__device__ __forceinline__ unsigned lds_addr(const char* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}

}  // namespace nerf
#include "mlp_mtile_asm.h"
namespace nerf {

// accumulator rows of register r in lane-half h: (r&3) + 8*(r>>2) + 4h  -> bias as 4 x float4
__device__ __forceinline__ f32x16 bias_tile(const float* bias_lds, int row0, int half) {
  f32x16 acc;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(bias_lds + row0 + 8 * g + 4 * half);
    acc[4 * g + 0] = b[0]; acc[4 * g + 1] = b[1]; acc[4 * g + 2] = b[2]; acc[4 * g + 3] = b[3];
  }
  return acc;
}

// ReLU as a signed integer max on the fp32 bit pattern (negative floats are negative integers,
// -0.0 -> +0): one v_max_i32 per element where fmaxf costs two v_max_f32 (hipcc canonicalises
// MFMA results before fmaxf).  NaNs pass through unchanged if positive-signed.
__device__ __forceinline__ float relu_bits(float x) {
  const int i = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, i > 0 ? i : 0);
}
template <bool RELU>
__device__ __forceinline__ void acc_to_operand_relu(const f32x16& acc, bf16x8& lo, bf16x8& hi) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    lo[j] = (__bf16)(RELU ? relu_bits(acc[j]) : acc[j]);
    hi[j] = (__bf16)(RELU ? relu_bits(acc[8 + j]) : acc[8 + j]);
  }
}

// fp32 accumulator tile -> the two bf16 B fragments (k-steps 2m, 2m+1) of the next step
__device__ __forceinline__ void acc_to_operand(const f32x16& acc, bf16x8& lo, bf16x8& hi) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    lo[j] = (__bf16)acc[j];
    hi[j] = (__bf16)acc[8 + j];
  }
}

template <int N, class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// stash stores (16-byte global stores) the epilogue of m-tile group g issues when training
template <bool BWD>
constexpr int group_stores(int g) {
  int acc = 0;
  for (int s = 0; s < plan::stream_steps(BWD); ++s) {
    const int kind = plan::stream_first(BWD) + s;
    const int mt = plan::step_of(kind).mt;
    if (g < acc + mt) {
      const int m = g - acc;
      if (kind == plan::F_RGB) return 0;
      if (kind == plan::F_HEAD && m == 8) return 0;
      return 2;
    }
    acc += mt;
  }
  return 0;
}
// stash stores issued while the chunk BEFORE the one opened by group g was consumed
template <bool BWD>
constexpr int prev_chunk_stores(int g) {
  const plan::Chunks& ch = WeightRing<BWD>::chunks();
  const int c = ch.group_chunk[g];
  if (c == 0) return 0;            // wraps across tile passes: drain everything
  int n = 0;
  for (int i = 0; i < ch.n_groups; ++i) n += ch.group_chunk[i] == c - 1 ? group_stores<BWD>(i) : 0;
  return n;
}

// Runs one GEMM step; epi(mc, acc) consumes each finished 32-row tile.
// STASH: the epilogues issue their stash stores (training); false for inference.
template <bool BWD, int KIND, int KS, bool STASH, class Epi>
__device__ __forceinline__ void run_step(WeightRing<BWD>& ring, const char*& a_base, bool more_passes,
                                         const bf16x8 (&b)[KS], const float* bias_lds, int half, Epi&& epi) {
  constexpr plan::Step st = plan::step_of(KIND);
  static_assert(KS == st.ks_acc + st.ks_nat, "operand k-steps");
  static_for<st.mt>([&](auto mc) {
    constexpr int m = decltype(mc)::value;
    constexpr int g = plan::group_of(KIND, m);
    constexpr const plan::Chunks& ch = WeightRing<BWD>::chunks();
    if constexpr (ch.group_first[g])
      a_base = ring.template advance<ch.group_chunk[g], STASH ? prev_chunk_stores<BWD>(g) : 0>(more_passes);
    f32x16 acc;
    if constexpr (BWD) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    } else {
      acc = bias_tile(bias_lds, plan::bias_off(KIND) + 32 * m, half);
    }
    acc = mtile_asm<KS>(lds_addr(a_base) + ch.group_off[g] * 1024, b, acc);
    epi(mc, acc);
  });
}

// Blocked stash image of a [n, 32*MT] bf16 matrix: one 2-KiB block per (32-sample wave tile,
// m-tile), eight 256-byte segments of four samples each; lane (c, h) writes its 16 accumulator
// rows as bf16 (the two B fragments lo, hi) at block_lane_offset(c, h) and + 128.  One wave store
// instruction therefore fills eight whole 128-byte lines (16-byte pieces at a 32-byte stride cost
// +45 % cycles in the stash-writing kernels), and the 32 lanes of one ds_read_b64_tr_b16 pass in
// the wgrad kernel (4 samples x h x lo/hi x 8-byte group) tile one 256-byte segment: no bank
// conflicts.  The wgrad kernel DMAs blocks into LDS verbatim.
__device__ __forceinline__ unsigned block_lane_offset(int col, int half) {
  return 256u * (col >> 2) + 64u * half + 16u * (col & 3);
}
__device__ __forceinline__ void stash_block(__bf16* base, int64_t wave_tile, int n_mtiles, int m, int col, int half,
                                            const bf16x8& lo, const bf16x8& hi) {
  char* p = reinterpret_cast<char*>(base) + (wave_tile * n_mtiles + m) * 2048 + block_lane_offset(col, half);
  // non-temporal: the images are streamed out once and read back by another kernel; plain stores
  // made the training step 7 % slower (they displace the weight stream in L2); sc0/sc1 variants
  // of the store were measured too: `nt` alone is what matters
  __builtin_nontemporal_store(lo, reinterpret_cast<bf16x8*>(p));
  __builtin_nontemporal_store(hi, reinterpret_cast<bf16x8*>(p + 128));
}
// natural-order operand (Fourier codes, output gradients): 1-KiB block per (wave tile, k-step),
// lane (c, h) owns bytes [(2c+h)*16, +16) = features 16ks + 8h + (0..7)
__device__ __forceinline__ void stash_nat(__bf16* base, int64_t wave_tile, int n_ks, int ks, int col, int half,
                                          const bf16x8& v) {
  char* p = reinterpret_cast<char*>(base) + ((wave_tile * n_ks + ks) * 64 + 2 * col + half) * 16;
  __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(p));
}

// ---- 8-bit images (asm-stream family, gen_stream_asm.py) ----
// acc-type block: 1 KiB per (32-sample wave tile, m-tile); lane (c, h) owns 16 bytes at 32c + 16h =
// its 16 accumulator rows (r&3) + 8(r>>2) + 4h, r = byte index.  One wave store instruction fills the
// whole block; the wgrad kernel's ds_read_b64_tr_b8 reads a 16-sample x 32-row operand from 512
// contiguous bytes (conflict-free).
__device__ __forceinline__ unsigned block8_lane_offset(int col, int half) { return 32u * col + 16u * half; }

// bf16x8 -> 8 x e4m3 (BF8 = false) or e5m2 (true), each divided by `scale`; MODE.FP16_OVFL must be set
// (saturation).  The s_nop keeps the half-register writes apart (dst_sel forwarding hazard).
template <bool BF8>
__device__ __forceinline__ u32x2 cvt8_bf16x8(const bf16x8& v, float scale) {
  const u32x4 w = __builtin_bit_cast(u32x4, v);
  unsigned r0 = 0u, r1 = 0u;
  const unsigned w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
  if constexpr (BF8)
    asm volatile("v_cvt_scalef32_pk_bf8_bf16 %0, %2, %6\n\tv_cvt_scalef32_pk_bf8_bf16 %1, %4, %6\n\ts_nop 1\n\t"
                 "v_cvt_scalef32_pk_bf8_bf16 %0, %3, %6 op_sel:[0,0,1]\n\tv_cvt_scalef32_pk_bf8_bf16 %1, %5, %6 op_sel:[0,0,1]\n\ts_nop 1"
                 : "+v"(r0), "+v"(r1) : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(scale));
  else
    asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %2, %6\n\tv_cvt_scalef32_pk_fp8_bf16 %1, %4, %6\n\ts_nop 1\n\t"
                 "v_cvt_scalef32_pk_fp8_bf16 %0, %3, %6 op_sel:[0,0,1]\n\tv_cvt_scalef32_pk_fp8_bf16 %1, %5, %6 op_sel:[0,0,1]\n\ts_nop 1"
                 : "+v"(r0), "+v"(r1) : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(scale));
  return u32x2{r0, r1};
}
// natural-order operand, 8-bit: 512-byte block per (wave tile, k-step); sample c's 16-byte record is
// [lane-half 0: features 16ks + 0..7 | lane-half 1: features 16ks + 8..15]
template <bool BF8>
__device__ __forceinline__ void stash_nat8(char* base, int64_t wave_tile, int n_ks, int ks, int col, int half,
                                           const bf16x8& v, float scale) {
  char* p = base + (wave_tile * n_ks + ks) * 512 + 16 * col + 8 * half;
  __builtin_nontemporal_store(cvt8_bf16x8<BF8>(v, scale), reinterpret_cast<u32x2*>(p));
}
__device__ __forceinline__ void set_fp8_saturate() {
  __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);   // hwreg(HW_REG_MODE, 23, 1) = FP16_OVFL
}

// ---------------------------------------------------------------------------
// Fourier code of one coordinate triple, element f of
// [x(3) | sin(2^0 pi x)(3) | cos(2^0 pi x)(3) | sin(2^1 pi x)(3) | ...]  (src/embeddings.py:28-32).
// The reference evaluates sin(fl(fl(x*2^b)*pi_f32)); we reproduce that argument exactly:
// y = x*2^b is exact, r0 = fract(y/2) is exact, and the rounding error of the reference's
// fp32 product (recovered with one fma) is added back as a correction in revolutions.
// v_sin_f32 takes revolutions; cos = sin(r + 1/4).
// ---------------------------------------------------------------------------
__device__ __forceinline__ float sincos_rev(float xa, float scale, float phase) {
  constexpr float kPiF = 3.14159274101257324f;            // fp32(pi)
  constexpr float kPiErr = 8.74227766e-8f;                // fp32(pi) - pi
  constexpr float kInv2Pi = 0.15915494309189535f;
  const float y = xa * scale;                              // exact (power-of-two scale)
  const float t = y * kPiF;                                // the reference's rounded argument
  const float resid = __builtin_fmaf(y, kPiF, -t);         // y*pi_f - t, exact
  const float corr = (y * kPiErr - resid) * kInv2Pi;       // (t - y*pi) / 2pi
  const float r = __builtin_amdgcn_fractf(0.5f * y) + (corr + phase);
  return __builtin_amdgcn_sinf(r);
}

struct FeatSpec { int axis; float scale; float phase; int raw; };   // raw: 0 trig, 1 coordinate, 2 const 1, 3 zero
template <int VALID>
constexpr FeatSpec feat_spec(int f) {
  if (f < 3) return {f, 1.0f, 0.0f, 1};
  if (f == VALID) return {0, 1.0f, 0.0f, 2};              // pad column = 1 (bias column of wgrad)
  if (f > VALID) return {0, 1.0f, 0.0f, 3};
  const int c = f - 3, band = c / 6, rem = c % 6;
  return {rem % 3, (float)(1 << band), rem >= 3 ? 0.25f : 0.0f, 0};
}

template <int VALID>
__device__ __forceinline__ float feat_eval(const FeatSpec s, float x0, float x1, float x2) {
  const float xa = s.axis == 0 ? x0 : (s.axis == 1 ? x1 : x2);
  if (s.raw == 1) return xa;
  if (s.raw == 2) return 1.0f;
  if (s.raw == 3) return 0.0f;
  return sincos_rev(xa, s.scale, s.phase);
}

// B fragments (natural k order) of a Fourier code with KS k-steps; feature f = 16*ks + 8*half + j
template <int KS, int VALID>
__device__ __forceinline__ void fourier_operand(float x0, float x1, float x2, int half, bf16x8 (&out)[KS]) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      constexpr int dummy = 0; (void)dummy;
      const FeatSpec s0 = feat_spec<VALID>(16 * ks + j), s1 = feat_spec<VALID>(16 * ks + 8 + j);
      float v;
      if (s0.raw == 0 && s1.raw == 0) {
        // both halves evaluate a trig feature: select the parameters, evaluate once
        const float xa0 = s0.axis == 0 ? x0 : (s0.axis == 1 ? x1 : x2);
        const float xa1 = s1.axis == 0 ? x0 : (s1.axis == 1 ? x1 : x2);
        v = sincos_rev(half ? xa1 : xa0, half ? s1.scale : s0.scale, half ? s1.phase : s0.phase);
      } else {
        const float v0 = feat_eval<VALID>(s0, x0, x1, x2), v1 = feat_eval<VALID>(s1, x0, x1, x2);
        v = half ? v1 : v0;
      }
      out[ks][j] = (__bf16)v;
    }
  }
}

// B fragments (natural k order) of an ALREADY ENCODED feature row (the BaseDecoder.forward(x_enc, d_enc)
// entry of the reference, src/decoders.py:68-87): feature f = 16*ks + 8*half + j; column VALID carries
// the constant 1 of the wgrad bias trick, columns beyond it are zero
template <int KS, int VALID>
__device__ __forceinline__ void encoded_operand(const float* __restrict__ row, int half, bf16x8 (&out)[KS]) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int f = 16 * ks + 8 * half + j;
      out[ks][j] = (__bf16)(f < VALID ? row[f] : (f == VALID ? 1.0f : 0.0f));
    }
  }
}

}  // namespace nerf
