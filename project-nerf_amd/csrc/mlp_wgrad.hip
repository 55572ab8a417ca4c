// Weight gradients of the fused decoder: dW[o][i] = sum_n dPre[n][o] * In[n][i]  (+ bias sums).
//
// A (pre-activation gradients, written by mlp_bwd.hip) and B (activations, stashed by
// mlp_fwd.hip) are blocked bf16 images: per 32-sample wave tile one 2-KiB block per 32
// features (or 1 KiB per 16 for the Fourier codes).  A workgroup streams consecutive wave
// tiles of ONE layer through a 4-stage LDS ring (global_load_lds_dwordx4, blocks copied
// verbatim), and reads MFMA operands with ds_read_b64_tr_b16: the blocks hold "sample on
// lane, 4 consecutive features per 8 bytes", the transposing read returns "feature on lane,
// 4 consecutive samples" = the A/B fragment of a contraction over samples.  Wave w owns
// output rows 32w..32w+31 x all input columns (fp32 accumulators stay in registers for the
// whole span), bias gradients fall out of an all-ones column.  Every workgroup stores its partial
// tile with plain stores and wgrad_reduce_kernel sums a job's tiles in order (small launches: float
// atomics).  HBM-bound: every stashed byte is read exactly once (9.8 KB per sample: sigma_layer's
// gradient rides on the feature job's stream of h7).  The decoder's bf16 jobs run in run_job16 (piece
// counts as template constants, straight-line ring loop), everything else in the generic run_job.
#include <stdlib.h>
#include "mlp_chain.h"
#include "mlp_stash.h"
#include "mlp_wgrad.h"

namespace nerf {
using namespace plan;

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;

// LDS-DMA issued from inline asm: hipcc orders every LDS read it can see behind ALL pending
// LDS-DMA writes it knows of (s_waitcnt vmcnt(0)), which would drain the prefetch ring at every
// fragment read.  Hidden in asm, the DMA -> read ordering is ours (counted vmcnt + s_barrier in
// the stage loop) and the transposing reads stay ordinary builtins the compiler can schedule.
// M0 carries the wave-uniform LDS destination; it is saved/restored around the instruction.
// `nt`: every stashed byte is read exactly once -- streaming it past L2 keeps the gradient
// accumulators (atomics) and the next step's weight stream resident (training step -3 %).
// (Moving the refill of the freed slot behind the first k-step's MFMAs: no change, 0.54 ms.)
__device__ __forceinline__ void dma_1k(const char* gsrc_lane, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc_lane), "s"(lds_dst) : "memory");
}

// A/B fragment of the contraction over samples: two transposing reads STRIDE bytes apart
// (samples 4t + 0..3, t = 0, 1) of this lane's feature column
template <int STRIDE>
__device__ __forceinline__ bf16x8 tr_frag(const char* p) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(p + STRIDE));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// LDS ring geometry of a stage: the decoder's jobs need 16 + 16 + 4 KiB, the Instant tiny-MLP jobs 4 + 4 + 2 KiB
struct BigStage { static constexpr int A = kWgStageA, B = kWgStageB, Bytes = kWgStageBytes; };
struct SmallStage { static constexpr int A = 4096, B = 4096, N = 2048, Bytes = 4096 + 4096 + 2048; };
// the Part 4 tiny-MLP jobs (kinds 6..12): two natural-order tiles (the sigma-net's [hash | time code] input)
struct SmallStageP4 { static constexpr int A = 4096, B = 4096, N = 4096, Bytes = 4096 + 4096 + 4096; };

struct LaneGeo {
  int lane, wave, fhalf, off_acc, off_nat;
  int off_acc8, off_nat8;   // 8-bit images (ds_read_b64_tr_b8)
};

// 8-bit operand of the contraction over samples: ONE transposing read returns, for the lane's feature,
// the 8 consecutive samples of its k-half (tools/probe/fp8_probe.hip pins the instruction's lane map)
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) i32x2* lds_i32x2_t;
__device__ __forceinline__ long tr_frag8(const char* p) {
  return __builtin_bit_cast(long, __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2_t)(p)));
}
// lane r of an acc-type 8-bit operand holds feature row phi(r) of its 32-row tile (the record of
// lane-half r>>4 is 16 consecutive accumulator registers); natural-order operands are the identity
__device__ __forceinline__ int phi8(int r) { return (r & 3) + 8 * ((r & 15) >> 2) + 4 * (r >> 4); }

// Partial sums of one span -> float atomics.  C layout: row = (r&3) + 8(r>>2) + 4(lane>>5) of the wave's
// 32-row tile, column = lane&31 of column tile k (two 128-byte segments per wave instruction).
template <int NT_ACC, int NT_NAT, bool ONES, bool SPLIT, bool FP8, int NT>
__device__ __forceinline__ void flush_tiles(const WgradArgs& args, const WgradJob& job, const f32x16 (&acc)[NT], const LaneGeo g,
                                            bool active) {
  const int wave = g.wave, lane = g.lane;
  if (!active || (args.debug & 4)) return;
  const int c32 = lane & 31, hrow = lane >> 5;
  const int m_tile = SPLIT ? 0 : wave;
  // 8-bit images: acc-type operands carry their tile's rows in the order phi8; the sums come back in
  // units of (gradient scale) x (activation scale), bias sums (all-ones operand) of the gradient scale alone
  const int c_acc = FP8 ? phi8(c32) : c32;
  float w_scale = 1.0f, b_scale = 1.0f;
  if constexpr (FP8) {
    b_scale = grad_image_scale(*args.amax);
    w_scale = b_scale * kActScale;
  }
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    int col = -1, bias_here = 0;
    bool use = true;
    if constexpr (SPLIT) {
      if (k == 0) { use = wave < NT_ACC; const int i = wave * 32 + c_acc; if (i < job.acc_valid) col = job.acc_col0 + i; }
      else { use = ONES && wave == NT_ACC % 8; bias_here = (c32 == 0); }
    } else {
      if (k < NT_ACC) { const int i = k * 32 + c_acc; if (i < job.acc_valid) col = job.acc_col0 + i; }
      else if (k < NT_ACC + NT_NAT) {
        const int i = (k - NT_ACC) * 32 + c32;
        if (i < job.nat_valid) col = job.nat_col0 + i;
        bias_here = (i == job.bias_nat_col);
      } else bias_here = (c32 == 0);
    }
    if (!use) continue;
    const bool ones_tile = SPLIT ? k == 1 : k == NT_ACC + NT_NAT;
    if (args.slab != nullptr) {      // plain stores into this workgroup's partial tile (two 128-byte segments per instruction)
      float* tile = args.slab + job.slab_off + (long long)((int)blockIdx.x - job.part0) * job.p_stride;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hrow;
        const int o = 32 * m_tile + ((FP8 && !SPLIT) ? phi8(row) : row) - job.o_row0;
        if (o >= 0 && o < job.o_valid) {
          if (col >= 0) tile[o * job.w_ld + col] = acc[k][r] * w_scale;
          if (bias_here) tile[job.o_valid * job.w_ld + o] = acc[k][r] * (ones_tile ? b_scale : w_scale);
        }
      }
      continue;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * hrow;
      const int o = 32 * m_tile + ((FP8 && !SPLIT) ? phi8(row) : row) - job.o_row0;
      if (o >= 0 && o < job.o_valid) {
        if (col >= 0) atomicAdd(args.grads + job.w_off + o * job.w_ld + col, acc[k][r] * w_scale);
        if (bias_here) atomicAdd(args.grads + job.bias_off + o, acc[k][r] * (ones_tile ? b_scale : w_scale));
      }
    }
  }
}

// 8-bit images, owner mode: the same span loop, software-pipelined inside the wave.  The fragments of
// k-step s+1 (or of the next stage's first k-step: the stage barrier already covers stage wt+1) are read
// while the MFMAs of k-step s run, so the matrix pipe is not left waiting for LDS at every k-step
// (compute-only time of the unpipelined loop: 0.30 ms against a 0.17-0.20 ms MFMA floor at 4096 x 64).
template <int NT_ACC, int NT_NAT, bool ONES>
__device__ __forceinline__ void run_job8(const WgradArgs& args, const WgradJob job, int wt0, int wt1,
                                         char* smem, const LaneGeo g) {
  constexpr int NB = NT_ACC + NT_NAT, NT = NB + (ONES ? 1 : 0);
  const int wave = g.wave, lane = g.lane;
  const bool active = wave < job.mt_a;
  f32x16 acc[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;

  const int pieces_a = job.a_bytes >> 10, pieces_b = job.b_acc_bytes >> 10, pieces_n = job.b_nat_bytes >> 10;
  const int pieces = pieces_a + pieces_b + pieces_n;
  const int per_wave = (pieces + 7) >> 3;    // every wave issues exactly this many (tail duplicates)
  const unsigned smem_lds = lds_addr(smem);
  auto issue = [&](int wt) {
    const unsigned stage = smem_lds + (wt & (kWgStages - 1)) * kWgStageBytes;
    for (int i = 0; i < per_wave; ++i) {
      int pc = wave + 8 * i;
      pc = pc < pieces ? pc : pieces - 1;
      const char* src;
      unsigned dst;
      if (pc < pieces_a) {
        src = job.a + (size_t)wt * job.a_bytes + pc * 1024;
        dst = stage + pc * 1024;
      } else if (pc < pieces_a + pieces_b) {
        const int o = pc - pieces_a;
        src = job.b_acc + (size_t)wt * job.b_acc_bytes + o * 1024;
        dst = stage + kWgStageA + o * 1024;
      } else {
        const int o = pc - pieces_a - pieces_b;
        src = job.b_nat + (size_t)wt * job.b_nat_bytes + o * 1024;
        dst = stage + kWgStageA + kWgStageB + o * 1024;
      }
      dma_1k(src + lane * 16, __builtin_amdgcn_readfirstlane(dst));
    }
  };
  auto wait_in_flight = [&](int stages) {   // all but `stages` newest stages of this wave have landed
    const int outstanding = stages * per_wave;
    if (outstanding >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (outstanding >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (outstanding >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (outstanding >= 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (outstanding >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (outstanding >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (outstanding >= 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  const int half_a = job.a_bytes >> 1, half_b = job.b_acc_bytes >> 1, half_n = job.b_nat_bytes >> 1;
  // fragments of k-step s (0..3) of the stage at `stage`: A tile of this wave, NB column tiles
  auto load = [&](const char* stage, int s, long& a, long (&b)[NB > 0 ? NB : 1]) {
    const int t = s >> 1, ss = s & 1;
    a = tr_frag8(stage + wave * 1024 + g.off_acc8 + t * half_a + 512 * ss);
#pragma unroll
    for (int k = 0; k < NT_ACC; ++k) b[k] = tr_frag8(stage + kWgStageA + g.off_acc8 + t * half_b + k * 1024 + 512 * ss);
#pragma unroll
    for (int k = 0; k < NT_NAT; ++k)
      b[NT_ACC + k] = tr_frag8(stage + kWgStageA + kWgStageB + g.off_nat8 + t * half_n + k * 1024 + 256 * ss);
  };
  const long ones8 = 0x3838383838383838L;   // 8 x e4m3 1.0
  const bool compute = active && !(args.debug & 1);

  if (!args.k16) {
    // ONE v_mfma_scale_f32_32x32x64_f8f6f4 per column tile and stage instead of four 32x32x16: the K = 64 form runs
    // at twice the rate (64 cycles for 4x the K; tools/probe/f8f6f4_probe.hip).  Its 32-byte operands are the four
    // K = 16 fragments of the stage side by side -- any k order is fine as long as A and B agree, and both come from
    // the same transposing reads.  A = e5m2 (cbsz 1), B = e4m3 (blgp 0), block scales 2^0 (E8M0 127).
    auto frag64 = [&](const char* p, int half, int step) {
      i32x8 v;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const long f = tr_frag8(p + (s4 >> 1) * half + step * (s4 & 1));
        v[2 * s4] = (int)f;
        v[2 * s4 + 1] = (int)(f >> 32);
      }
      return v;
    };
    auto col_tile = [&](const char* stage, int k) {
      return k < NT_ACC ? frag64(stage + kWgStageA + g.off_acc8 + k * 1024, half_b, 512)
                        : frag64(stage + kWgStageA + kWgStageB + g.off_nat8 + (k - NT_ACC) * 1024, half_n, 256);
    };
    i32x8 ones64;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones64[e] = 0x38383838;
    __builtin_amdgcn_s_barrier();   // previous span's readers are done with the ring
    if (wt0 + 0 < wt1) issue(wt0 + 0);
    if (wt0 + 1 < wt1) issue(wt0 + 1);
    if (wt0 + 2 < wt1) issue(wt0 + 2);
    for (int wt = wt0; wt < wt1; ++wt) {
      wait_in_flight((wt + 1 < wt1) + (wt + 2 < wt1));
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (wt + 3 < wt1 && !(args.debug & 2)) issue(wt + 3);
      if (!compute) continue;
      const char* stage = smem + (wt & (kWgStages - 1)) * kWgStageBytes;
      const i32x8 a = frag64(stage + wave * 1024 + g.off_acc8, half_a, 512);
      i32x8 b_cur = NB > 0 ? col_tile(stage, 0) : ones64;
      static_for<NB>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        i32x8 b_nxt = ones64;
        if constexpr (k + 1 < NB) b_nxt = col_tile(stage, k + 1);       // the next tile's reads ride under this MFMA
        acc[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b_cur, acc[k], 1, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        b_cur = b_nxt;
      });
      if constexpr (ONES) acc[NB] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, ones64, acc[NB], 1, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    flush_tiles<NT_ACC, NT_NAT, ONES, false, true, NT>(args, job, acc, g, active);
    return;
  }

  __builtin_amdgcn_s_barrier();   // previous span's readers are done with the ring
  if (wt0 + 0 < wt1) issue(wt0 + 0);
  if (wt0 + 1 < wt1) issue(wt0 + 1);
  if (wt0 + 2 < wt1) issue(wt0 + 2);
  wait_in_flight((wt0 + 1 < wt1) + (wt0 + 2 < wt1));
  __builtin_amdgcn_s_barrier();   // stage wt0 has landed for every wave
  asm volatile("" ::: "memory");
  long a_cur = 0, b_cur[NB > 0 ? NB : 1];
  if (compute) load(smem + (wt0 & (kWgStages - 1)) * kWgStageBytes, 0, a_cur, b_cur);
  for (int wt = wt0; wt < wt1; ++wt) {
    // stage wt+1 must be visible before this iteration's last k-step prefetches from it; the same
    // barrier says every wave has finished reading stage wt-1, whose slot the next DMA refills
    wait_in_flight(wt + 2 < wt1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wt + 3 < wt1 && !(args.debug & 2)) issue(wt + 3);
    if (!compute) continue;
    const char* stage = smem + (wt & (kWgStages - 1)) * kWgStageBytes;
    const char* next = smem + ((wt + 1) & (kWgStages - 1)) * kWgStageBytes;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      long a_nxt = 0, b_nxt[NB > 0 ? NB : 1];
      if (s < 3) load(stage, s + 1, a_nxt, b_nxt);
      else if (wt + 1 < wt1) load(next, 0, a_nxt, b_nxt);
#pragma unroll
      for (int k = 0; k < NB; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf8_fp8(a_cur, b_cur[k], acc[k], 0, 0, 0);
      if constexpr (ONES) acc[NB] = __builtin_amdgcn_mfma_f32_32x32x16_bf8_fp8(a_cur, ones8, acc[NB], 0, 0, 0);
      // interleave: the next k-step's reads ride in the gaps of this k-step's MFMAs
      static_for<NT>([&](auto kc) {
        constexpr int k = decltype(kc)::value, reads = NB + 1;
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, reads / NT + (k < reads % NT ? 1 : 0), 0);
      });
      __builtin_amdgcn_sched_barrier(0);
      a_cur = a_nxt;
#pragma unroll
      for (int k = 0; k < NB; ++k) b_cur[k] = b_nxt[k];
    }
  }
  flush_tiles<NT_ACC, NT_NAT, ONES, false, true, NT>(args, job, acc, g, active);
}

// LDS-DMA of one 1-KiB piece with a wave-uniform base: lane l copies base[voff_l .. +16) to LDS m0 + 16 l
__device__ __forceinline__ void dma_1k_s(const char* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt"
               : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

// The bf16-image jobs of the vanilla decoder (kinds 0..5), straight-line form.  The generic run_job below decides at run time,
// every ring iteration, how many pieces a wave copies, from which of the three images, and which s_waitcnt immediate fits: some
// 500 shader cycles of scalar branches per iteration that all eight waves spend together behind the barrier (skeleton launch
// without DMA and MFMAs: 0.28 us per iteration, 0.12 ms per step).  Here the piece counts are template constants (PA / PB / PN
// 1-KiB pieces per wave tile of the A image, the blocked B image and the natural-order B image), every wave's pieces are
// resolved once per span into (wave-uniform base, LDS offset, stride), the steady state is wait(2 PER) - barrier - PER copies -
// MFMAs without a branch, and the last three iterations of a span are peeled.
// SIG: the merged feature + sigma job -- both contract against the SAME h7 image, so sigma_layer's gradient rides on the
// feature job's stream instead of re-reading 16 KiB per wave tile in a job of its own (one ring iteration and 5 % of the
// kernel's bytes less per wave tile).  The 16-wide natural gradient block (job.a2) lands in the stage's unused N region; wave
// w contracts it with column tile w (a tenth accumulator tile); its row o2_row summed over the samples is sigma_layer's bias
// gradient: the wave whose turn it is (stage number mod 8) adds its operand registers on the vector ALU.
template <int PA, int PB, int PN, int NT_ACC, int NT_NAT, bool ONES, bool SPLIT, bool SIG = false, class Stage = BigStage>
__device__ __forceinline__ void run_job16(const WgradArgs& args, const WgradJob job, int wt0, int wt1, char* smem, const LaneGeo g) {
  static_assert(!SIG || (PN == 0 && !SPLIT && NT_ACC == 8), "the merged job uses the N region and one column tile per wave");
  constexpr int NT = SPLIT ? 2 : NT_ACC + NT_NAT + (ONES ? 1 : 0);
  constexpr int PIECES = PA + PB + PN + (SIG ? 1 : 0), FULL = PIECES / 8, REM = PIECES % 8, PER = FULL + (REM ? 1 : 0);
  const int wave = g.wave, lane = g.lane;
  const bool active = SPLIT ? (wave <= NT_ACC) : (wave < PA / 2);
  const bool compute = active && !(args.debug & 1), copy = !(args.debug & 2);
  const bool extra = wave < REM;        // waves 0 .. REM-1 copy one piece more than the others
  f32x16 acc[NT], acc_sig;
#pragma unroll
  for (int k = 0; k < NT; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc_sig[r] = 0.0f;
  float sig_bias = 0.0f;

  // this wave's pieces: (wave-uniform source, LDS offset in the stage, bytes per wave tile of that image)
  const char* base[PER];
  unsigned dst[PER];
  int stride[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    int pc = wave + 8 * i;
    pc = pc < PIECES ? pc : PIECES - 1;
    if (pc < PA) {
      stride[i] = PA * 1024;
      base[i] = job.a + pc * 1024;
      dst[i] = pc * 1024;
    } else if (pc < PA + PB) {
      stride[i] = PB * 1024;
      base[i] = job.b_acc + (pc - PA) * 1024;
      dst[i] = Stage::A + (pc - PA) * 1024;
    } else if (SIG && pc == PIECES - 1) {
      stride[i] = 1024;
      base[i] = job.a2;
      dst[i] = Stage::A + Stage::B;
    } else {
      stride[i] = PN * 1024;
      base[i] = job.b_nat + (pc - PA - PB) * 1024;
      dst[i] = Stage::A + Stage::B + (pc - PA - PB) * 1024;
    }
    base[i] += (size_t)wt0 * stride[i];
  }
  const unsigned smem_lds = lds_addr(smem), voff = lane * 16;
  auto issue = [&](int wt) {            // stages are issued in order: base[] walks with them
    const unsigned stage = smem_lds + (wt & (kWgStages - 1)) * Stage::Bytes;
#pragma unroll
    for (int i = 0; i < FULL; ++i) {
      dma_1k_s(base[i], voff, stage + dst[i]);
      base[i] += stride[i];
    }
    if constexpr (REM != 0) {
      if (extra) {
        dma_1k_s(base[FULL], voff, stage + dst[FULL]);
        base[FULL] += stride[FULL];
      }
    }
  };
  auto wait_stages = [&](auto stages) {   // all but the `stages` newest stages of this wave have landed
    constexpr int S = decltype(stages)::value;
    if constexpr (REM != 0) {
      if (extra) wait_vm<S * (FULL + 1)>();
      else wait_vm<S * FULL>();
    } else wait_vm<S * FULL>();
  };

  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
  auto mfmas = [&](int wt) {
    const char* stage = smem + (wt & (kWgStages - 1)) * Stage::Bytes;
    const char* pa = SPLIT ? stage + g.off_nat - 1024 * g.fhalf : stage + wave * 2048 + g.off_acc;
    const char* pb = stage + Stage::A + g.off_acc + (SPLIT ? (wave < NT_ACC ? wave : 0) * 2048 : 0);
    const char* pn = stage + Stage::A + Stage::B + g.off_nat;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af = SPLIT ? tr_frag<128>(pa + 512 * s) : tr_frag<256>(pa + 1024 * s);
      if (SPLIT && g.fhalf) {
#pragma unroll
        for (int e = 0; e < 8; ++e) af[e] = (__bf16)0.0f;
      }
      if constexpr (SPLIT) {
        if (wave < NT_ACC) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, tr_frag<256>(pb + 1024 * s), acc[0], 0, 0, 0);
        if (ONES && wave == NT_ACC % 8) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, ones, acc[1], 0, 0, 0);
      } else {
        // all fragment reads of the k-step first, then the MFMAs: hipcc otherwise recycles one
        // fragment register (read -> wait -> mfma), exposing the LDS latency per MFMA
        bf16x8 bf[NT_ACC + NT_NAT + 1];
#pragma unroll
        for (int k = 0; k < NT_ACC; ++k) bf[k] = tr_frag<256>(pb + k * 2048 + 1024 * s);
#pragma unroll
        for (int k = 0; k < NT_NAT; ++k) bf[NT_ACC + k] = tr_frag<128>(pn + k * 2048 + 512 * s);
        bf[NT_ACC + NT_NAT] = ones;
#pragma unroll
        for (int k = 0; k < NT; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[k], acc[k], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * (1 + NT_ACC + NT_NAT), 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SIG) {
          // after the k-step's own MFMAs, in the registers they freed (the kernel sits on the 256-register limit)
          bf16x8 a2 = tr_frag<128>(pn - 1024 * g.fhalf + 512 * s);      // natural 16-wide block: the upper feature half reads as zeros
          if (g.fhalf) {
#pragma unroll
            for (int e = 0; e < 8; ++e) a2[e] = (__bf16)0.0f;
          }
          const bf16x8 b2 = tr_frag<256>(pb + wave * 2048 + 1024 * s);  // column tile `wave` of h7 once more (register arrays cannot be indexed by the wave)
          acc_sig = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc_sig, 0, 0, 0);
          if ((wt & 7) == wave) {      // lane r (r < 16) of each half holds 8 consecutive samples of gradient column r
            const s16x8 bits = __builtin_bit_cast(s16x8, a2);
#pragma unroll
            for (int e = 0; e < 8; ++e) sig_bias += __builtin_bit_cast(float, (unsigned)(unsigned short)bits[e] << 16);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };

  __builtin_amdgcn_s_barrier();   // previous span's readers are done with the ring
  if (copy) {
    if (wt0 + 0 < wt1) issue(wt0 + 0);
    if (wt0 + 1 < wt1) issue(wt0 + 1);
    if (wt0 + 2 < wt1) issue(wt0 + 2);
  }
  int wt = wt0;
  for (; wt + 3 < wt1; ++wt) {    // steady state: stages wt+1 and wt+2 stay in flight, the slot of stage wt-1 is refilled
    wait_stages(std::integral_constant<int, 2>{});
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (copy) issue(wt + 3);
    if (compute) mfmas(wt);
  }
  for (; wt < wt1; ++wt) {        // the span's last three stages: nothing left to issue
    const int left = wt1 - 1 - wt;
    if (left >= 2) wait_stages(std::integral_constant<int, 2>{});
    else if (left == 1) wait_stages(std::integral_constant<int, 1>{});
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (compute) mfmas(wt);
  }
  flush_tiles<NT_ACC, NT_NAT, ONES, SPLIT, false, NT>(args, job, acc, g, active);
  if constexpr (SIG) {
    // row o2_row of the tenth tile = d sigma_layer.weight[wave*32 + c32]; the bias sum: lanes o2_row and 32 + o2_row of every
    // wave hold partial sums over that wave's stages -> scratch behind the ring -> wave 0 adds the eight in order
    float* red = reinterpret_cast<float*>(smem + kWgStages * Stage::Bytes);
    const float both = sig_bias + __shfl(sig_bias, (lane + 32) & 63);
    if (lane == job.o2_row) red[wave] = both;
    __builtin_amdgcn_s_barrier();
    if (!(args.debug & 4)) {
      const int c32 = lane & 31, hrow = lane >> 5;
      float w = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if ((r & 3) + 8 * (r >> 2) + 4 * hrow == job.o2_row) w = acc_sig[r];
      const bool mine = ((job.o2_row >> 2) & 1) == hrow;      // C layout: row = (r&3) + 8(r>>2) + 4 hrow
      float bias = 0.0f;
      if (wave == 0 && lane == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) bias += red[k];
      }
      if (args.slab != nullptr) {
        float* tile = args.slab + job.slab_off + (long long)((int)blockIdx.x - job.part0) * job.p_stride + job.o_valid * job.w_ld + job.o_valid;
        if (mine) tile[wave * 32 + c32] = w;
        if (wave == 0 && lane == 0) tile[job.acc_valid] = bias;
      } else {
        if (mine) atomicAdd(args.grads + job.w2_off + wave * 32 + c32, w);
        if (wave == 0 && lane == 0) atomicAdd(args.grads + job.bias2_off, bias);
      }
    }
    __builtin_amdgcn_s_barrier();   // the scratch is free for the next span
  }
}

// One job span [wt0, wt1) of one layer.  OWNER mode (SPLIT = false): wave w owns output rows
// 32w.. x all NT = NT_ACC + NT_NAT + ONES column tiles.  SPLIT mode (single 16-row natural A
// block, dsmall): wave w owns column tile w (w < NT_ACC) and wave NT_ACC % 8 the ones tile.
// FP8: the images are 8-bit (A = e5m2 gradients, B = e4m3 activations); one ring stage then holds TWO wave
// tiles (64 samples = four k-steps): same bytes per stage as one bf16 wave tile, half the iterations.
template <int NT_ACC, int NT_NAT, bool ONES, bool SPLIT, bool FP8, class Stage = BigStage>
__device__ __forceinline__ void run_job(const WgradArgs& args, const WgradJob job, int wt0, int wt1,
                                        char* smem, const LaneGeo g) {
  constexpr int NT = SPLIT ? 2 : NT_ACC + NT_NAT + (ONES ? 1 : 0);
  const int wave = g.wave, lane = g.lane;
  const bool active = SPLIT ? (wave <= NT_ACC) : (wave < job.mt_a);
  f32x16 acc[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;

  const int pieces_a = job.a_bytes >> 10, pieces_b = job.b_acc_bytes >> 10, pieces_n = job.b_nat_bytes >> 10;
  const int pieces = pieces_a + pieces_b + pieces_n;
  const int per_wave = (pieces + 7) >> 3;    // every wave issues exactly this many (tail duplicates)
  const unsigned smem_lds = lds_addr(smem);
  auto issue = [&](int wt) {
    const unsigned stage = smem_lds + (wt & (kWgStages - 1)) * Stage::Bytes;
    for (int i = 0; i < per_wave; ++i) {
      int pc = wave + 8 * i;
      pc = pc < pieces ? pc : pieces - 1;
      const char* src;
      unsigned dst;
      if (pc < pieces_a) {
        src = job.a + (size_t)wt * job.a_bytes + pc * 1024;
        dst = stage + pc * 1024;
      } else if (pc < pieces_a + pieces_b) {
        const int o = pc - pieces_a;
        src = job.b_acc + (size_t)wt * job.b_acc_bytes + o * 1024;
        dst = stage + Stage::A + o * 1024;
      } else {
        const int o = pc - pieces_a - pieces_b;
        src = job.b_nat + (size_t)wt * job.b_nat_bytes + o * 1024;
        dst = stage + Stage::A + Stage::B + o * 1024;
      }
      dma_1k(src + lane * 16, __builtin_amdgcn_readfirstlane(dst));
    }
  };
  auto wait_in_flight = [&](int stages) {   // all but `stages` newest stages of this wave have landed
    const int outstanding = stages * per_wave;
    if (outstanding >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (outstanding >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (outstanding >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (outstanding >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (outstanding >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (outstanding >= 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
  const long ones8 = 0x3838383838383838L;   // 8 x e4m3 1.0

  __builtin_amdgcn_s_barrier();   // previous span's readers are done with the ring
  if (wt0 + 0 < wt1) issue(wt0 + 0);
  if (wt0 + 1 < wt1) issue(wt0 + 1);
  if (wt0 + 2 < wt1) issue(wt0 + 2);
  for (int wt = wt0; wt < wt1; ++wt) {
    wait_in_flight((wt + 1 < wt1) + (wt + 2 < wt1));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wt + 3 < wt1 && !(args.debug & 2)) issue(wt + 3);
    if (!active || (args.debug & 1)) continue;
    const char* stage = smem + (wt & (kWgStages - 1)) * Stage::Bytes;
    const char* pa = SPLIT ? stage + g.off_nat - 1024 * g.fhalf : stage + wave * 2048 + g.off_acc;
    const char* pb = stage + Stage::A + g.off_acc + (SPLIT ? (wave < NT_ACC ? wave : 0) * 2048 : 0);
    const char* pn = stage + Stage::A + Stage::B + g.off_nat;
    if constexpr (FP8) {
      const int half_a = job.a_bytes >> 1, half_b = job.b_acc_bytes >> 1, half_n = job.b_nat_bytes >> 1;
      const char* pa8 = SPLIT ? stage + g.off_nat8 - 512 * g.fhalf : stage + wave * 1024 + g.off_acc8;
      const char* pb8 = stage + Stage::A + g.off_acc8 + (SPLIT ? (wave < NT_ACC ? wave : 0) * 1024 : 0);
      const char* pn8 = stage + Stage::A + Stage::B + g.off_nat8;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int t = s >> 1, ss = s & 1;
        long af = SPLIT ? tr_frag8(pa8 + t * half_a + 256 * ss) : tr_frag8(pa8 + t * half_a + 512 * ss);
        if (SPLIT && g.fhalf) af = 0;
        if constexpr (SPLIT) {
          if (wave < NT_ACC)
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf8_fp8(af, tr_frag8(pb8 + t * half_b + 512 * ss), acc[0], 0, 0, 0);
          if (ONES && wave == NT_ACC % 8) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf8_fp8(af, ones8, acc[1], 0, 0, 0);
        } else {
          long bf[NT_ACC + NT_NAT + 1];
#pragma unroll
          for (int k = 0; k < NT_ACC; ++k) bf[k] = tr_frag8(pb8 + t * half_b + k * 1024 + 512 * ss);
#pragma unroll
          for (int k = 0; k < NT_NAT; ++k) bf[NT_ACC + k] = tr_frag8(pn8 + t * half_n + k * 1024 + 256 * ss);
          bf[NT_ACC + NT_NAT] = ones8;
#pragma unroll
          for (int k = 0; k < NT; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf8_fp8(af, bf[k], acc[k], 0, 0, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1 + NT_ACC + NT_NAT, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      continue;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af = SPLIT ? tr_frag<128>(pa + 512 * s) : tr_frag<256>(pa + 1024 * s);
      if (SPLIT && g.fhalf) {
#pragma unroll
        for (int e = 0; e < 8; ++e) af[e] = (__bf16)0.0f;
      }
      if constexpr (SPLIT) {
        if (wave < NT_ACC) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, tr_frag<256>(pb + 1024 * s), acc[0], 0, 0, 0);
        if (ONES && wave == NT_ACC % 8) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, ones, acc[1], 0, 0, 0);
      } else {
        // all fragment reads of the k-step first, then the MFMAs: hipcc otherwise recycles one
        // fragment register (read -> wait -> mfma), exposing the LDS latency per MFMA
        bf16x8 bf[NT_ACC + NT_NAT + 1];
#pragma unroll
        for (int k = 0; k < NT_ACC; ++k) bf[k] = tr_frag<256>(pb + k * 2048 + 1024 * s);
#pragma unroll
        for (int k = 0; k < NT_NAT; ++k) bf[NT_ACC + k] = tr_frag<128>(pn + k * 2048 + 512 * s);
        bf[NT_ACC + NT_NAT] = ones;
#pragma unroll
        for (int k = 0; k < NT; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[k], acc[k], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * (1 + NT_ACC + NT_NAT), 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  flush_tiles<NT_ACC, NT_NAT, ONES, SPLIT, FP8, NT>(args, job, acc, g, active);
}

// FP8 = false: bf16 training images (the default); true: the 8-bit opt-in.  Two instantiations, so that a profile of a run
// that uses both (bench.py times the 8-bit step as a labelled secondary) reports them as two kernels
template <bool FP8>
__global__ void __launch_bounds__(512, 2) mlp_wgrad_kernel(const WgradArgs args) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  LaneGeo g;
  g.lane = threadIdx.x & 63;
  g.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // transposing-read lane geometry: 16-lane group grp covers features 16*fhalf.. of k-half hh;
  // lane 4q+p supplies sample (8hh + q) [+4 for the second read], features 4p..4p+3
  const int grp = g.lane >> 4, i16 = g.lane & 15, q = i16 >> 2, p = i16 & 3;
  const int hh = grp >> 1;
  g.fhalf = grp & 1;
  // blocked image (mlp_chain.h::stash_block): sample c = 8hh + q (+16s, +4 second read), lane-half p&1,
  // fragment fhalf, 8-byte row group p>>1
  g.off_acc = 512 * hh + 128 * g.fhalf + 64 * (p & 1) + 16 * q + 8 * (p >> 1);   // + 1024*s + 2048*block
  g.off_nat = 32 * (8 * hh + q) + 16 * (p >> 1) + 8 * (p & 1) + 1024 * g.fhalf;  // + 512*s + 2048*pair
  // 8-bit images (ds_read_b64_tr_b8): lane 2q8+p8 of a 16-lane group supplies sample 8hh + q8, 8-byte piece p8
  // of the 16-byte record of lane-half (acc) / k-step (nat) fhalf
  const int q8 = i16 >> 1, p8 = i16 & 1;
  g.off_acc8 = 256 * hh + 32 * q8 + 16 * g.fhalf + 8 * p8;            // + 512*ss + 1024*block + wave-tile half
  g.off_nat8 = 512 * g.fhalf + 16 * (8 * hh + q8) + 8 * p8;          // + 256*ss + 1024*pair + wave-tile half

  // this workgroup's span of the cost line
  const long long lo = args.total_cost * blockIdx.x / gridDim.x;
  const long long hi = args.total_cost * (blockIdx.x + 1) / gridDim.x;
  for (int j = 0; j < args.n_jobs; ++j) {
    const WgradJob job = args.jobs[j];   // by value: fields live in SGPRs, not re-read per use
    const long long j0 = job.cost0, j1 = job.cost0 + (long long)job.cost * args.wave_tiles;
    if (hi <= j0 || lo >= j1) continue;
    // wave tile t belongs to the workgroup whose span contains its first cost unit
    const long long a0 = lo > j0 ? lo - j0 : 0, a1 = (hi < j1 ? hi : j1) - j0;
    const int wt0 = (int)((a0 + job.cost - 1) / job.cost), wt1 = (int)((a1 + job.cost - 1) / job.cost);
    if (wt0 >= wt1) continue;
    if constexpr (FP8) {          // 8-bit images (vanilla decoder, asm-stream family, option stash_fp8)
      switch (job.kind) {
        case 0: run_job8<8, 0, true>(args, job, wt0, wt1, smem, g); break;
        case 1: run_job8<8, 2, false>(args, job, wt0, wt1, smem, g); break;
        case 2: run_job8<0, 2, false>(args, job, wt0, wt1, smem, g); break;
        case 3: run_job8<8, 1, false>(args, job, wt0, wt1, smem, g); break;
        case 4: run_job<8, 0, true, true, true>(args, job, wt0, wt1, smem, g); break;
        default: run_job<4, 0, true, true, true>(args, job, wt0, wt1, smem, g); break;
      }
    } else {
    switch (job.kind) {
      case 0: run_job16<16, 16, 0, 8, 0, true, false>(args, job, wt0, wt1, smem, g); break;    // 256x256 (+bias)
      case 1: run_job16<16, 16, 4, 8, 2, false, false>(args, job, wt0, wt1, smem, g); break;   // pts_layers.4
      case 2: run_job16<16, 0, 4, 0, 2, false, false>(args, job, wt0, wt1, smem, g); break;    // pts_layers.0
      case 3: run_job16<8, 16, 2, 8, 1, false, false>(args, job, wt0, wt1, smem, g); break;    // view_layer
      case 4: run_job16<1, 16, 0, 8, 0, true, true>(args, job, wt0, wt1, smem, g); break;      // sigma_layer
      case 5: run_job16<1, 8, 0, 4, 0, true, true>(args, job, wt0, wt1, smem, g); break;       // rgb_layer
      case 13: run_job16<16, 16, 0, 8, 0, true, false, true>(args, job, wt0, wt1, smem, g); break;   // feature_layer + sigma_layer
      case 6: run_job<0, 1, false, false, false>(args, job, wt0, wt1, smem, g); break;   // instant sigma-net layer 1
      case 7: run_job<2, 0, false, false, false>(args, job, wt0, wt1, smem, g); break;   // instant 64-wide layers
      case 8: run_job<1, 1, false, false, false>(args, job, wt0, wt1, smem, g); break;   // instant colour-net layer 1
      default: run_job<2, 0, false, true, false>(args, job, wt0, wt1, smem, g); break;   // instant rgb layer
    }
    }
  }
}

// The Instant tiny-MLP jobs (kinds 6..9) on their own: 10 KiB stages and few accumulators, so that several
// workgroups share a CU -- the loop is barrier- and DMA-latency-bound on such small stages, a second and third
// workgroup fill the waits of the first (the decoder's kernel holds 256 VGPRs and 144 KiB of LDS: one per CU)
// P4 = false: the Part 2 Instant jobs (kinds 6..9); true: the Part 4 field's jobs (kinds 6..12, p4mlp.hip) -- its own
// instantiation so that the three-tile jobs' registers do not lower the occupancy of the Instant step's kernel
// (launch bounds: the Instant jobs fit 80 VGPRs = three workgroups per CU, 46.7 -> 44.3 us; the Part 4 jobs spill there, 48 -> 52 us)
template <bool P4>
__global__ void __launch_bounds__(512, P4 ? 4 : 6) mlp_wgrad_small_kernel(const WgradArgs args) {
  using Stage = std::conditional_t<P4, SmallStageP4, SmallStage>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  LaneGeo g;
  g.lane = threadIdx.x & 63;
  g.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = g.lane >> 4, i16 = g.lane & 15, q = i16 >> 2, p = i16 & 3;
  const int hh = grp >> 1;
  g.fhalf = grp & 1;
  g.off_acc = 512 * hh + 128 * g.fhalf + 64 * (p & 1) + 16 * q + 8 * (p >> 1);
  g.off_nat = 32 * (8 * hh + q) + 16 * (p >> 1) + 8 * (p & 1) + 1024 * g.fhalf;
  g.off_acc8 = g.off_nat8 = 0;
  const long long lo = args.total_cost * blockIdx.x / gridDim.x;
  const long long hi = args.total_cost * (blockIdx.x + 1) / gridDim.x;
  for (int j = 0; j < args.n_jobs; ++j) {
    const WgradJob job = args.jobs[j];
    const long long j0 = job.cost0, j1 = job.cost0 + (long long)job.cost * args.wave_tiles;
    if (hi <= j0 || lo >= j1) continue;
    const long long a0 = lo > j0 ? lo - j0 : 0, a1 = (hi < j1 ? hi : j1) - j0;
    const int wt0 = (int)((a0 + job.cost - 1) / job.cost), wt1 = (int)((a1 + job.cost - 1) / job.cost);
    if (wt0 >= wt1) continue;
    // straight-line ring loop (run_job16): piece counts as template constants; the 1-KiB pieces of a stage are A (PA), blocked B
    // (PB), natural-order B (PN)
    if constexpr (P4) {
      switch (job.kind) {
        case 10: run_job16<4, 4, 0, 2, 0, true, false, false, Stage>(args, job, wt0, wt1, smem, g); continue;    // time modulation layer 2 (+ bias)
        case 11: run_job16<4, 4, 2, 2, 1, false, false, false, Stage>(args, job, wt0, wt1, smem, g); continue;   // displacement decoder layer 1
        case 12: run_job16<4, 0, 4, 0, 2, false, false, false, Stage>(args, job, wt0, wt1, smem, g); continue;   // sigma-net layer 1 on [hash | time code]
        default: break;
      }
    }
    switch (job.kind) {
      case 6: run_job16<4, 0, 2, 0, 1, false, false, false, Stage>(args, job, wt0, wt1, smem, g); break;   // sigma-net layer 1
      case 7:                                                                                           // 64-wide layers: 64 or 16 outputs
        if (job.mt_a == 2) run_job16<4, 4, 0, 2, 0, false, false, false, Stage>(args, job, wt0, wt1, smem, g);
        else run_job16<2, 4, 0, 2, 0, false, false, false, Stage>(args, job, wt0, wt1, smem, g);
        break;
      case 8: run_job16<4, 2, 2, 1, 1, false, false, false, Stage>(args, job, wt0, wt1, smem, g); break;   // colour-net layer 1
      default: run_job16<1, 4, 0, 2, 0, false, true, false, Stage>(args, job, wt0, wt1, smem, g); break;   // rgb layer
    }
  }
}

// slab mode: grads[parameter] = sum over the job's partial tiles, in tile order (the same sum every run)
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const WgradArgs args) {
  const WgradJob& job = args.jobs[blockIdx.y];
  const int n_w = job.o_valid * job.w_ld, n_b = n_w + job.o_valid, n_all = n_b + job.n2;
  const float* tile = args.slab + job.slab_off;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_all; i += gridDim.x * blockDim.x) {
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int t = 0;
    for (; t + 4 <= job.n_parts; t += 4) {       // four independent loads in flight
      s0 += tile[(long long)(t + 0) * job.p_stride + i];
      s1 += tile[(long long)(t + 1) * job.p_stride + i];
      s2 += tile[(long long)(t + 2) * job.p_stride + i];
      s3 += tile[(long long)(t + 3) * job.p_stride + i];
    }
    for (; t < job.n_parts; ++t) s0 += tile[(long long)t * job.p_stride + i];
    const float sum = (s0 + s1) + (s2 + s3);
    // which parameter this element of the tile is; the tiny-MLP launches ADD (one writer per parameter: the same bits every run)
    float* dst = i < n_w ? args.grads + job.w_off + i
               : i < n_b ? ((job.ones || job.bias_nat_col >= 0) ? args.grads + job.bias_off + (i - n_w) : nullptr)   // bias-free jobs: no such row
               : i < n_all - 1 ? args.grads + job.w2_off + (i - n_b)             // merged job: second output behind the bias sums
                               : args.grads + job.bias2_off;
    if (dst != nullptr) *dst = args.slab_accumulate ? *dst + sum : sum;
  }
}

}  // namespace nerf

using namespace nerf;

// part 0: all twelve layer jobs; part 1: pts_layers.4 .. rgb_layer (parameters [kW4, end)); part 2:
// pts_layers.0 .. 3 (parameters [0, kW4)).  Parts 1 and 2 let a data-parallel caller all-reduce the
// first range while the second is still being computed.
int nerf_launch_wgrad(const char* stash, const StashLayout& sl, const char* work, const BwdLayout& bl,
                      int64_t n, float* grads, int part, hipStream_t stream, size_t zero_lo, size_t zero_hi) {
  WgradArgs args{};
  const size_t np = (size_t)sl.n_pad, eb = sl.fp8 ? 1 : 2;
  if (sl.fp8 != bl.fp8) return fail(NERF_EINVAL, "nerf_mlp_bwd: stash and workspace disagree on the image width");
  // job byte counts are per ring stage: one 32-sample wave tile of bf16 images or TWO wave tiles of
  // 8-bit images -- the same numbers either way
  args.amax = sl.fp8 ? reinterpret_cast<const float*>(work + bl.amax) : nullptr;
  const char* xenc = stash + sl.xenc;
  const char* denc = stash + sl.denc;
  auto st_h = [&](int l) { return stash + sl.h + (size_t)l * np * 256 * eb; };
  auto dh = [&](int l) { return work + bl.dh + (size_t)l * np * 256 * eb; };
  int nj = 0;
  auto add = [&](WgradJob j) {
    args.jobs[nj++] = j;
  };
  // pts_layers.0: dH0 x xenc (bias from the code's constant-one column 63)
  {
    WgradJob j{};
    j.a = dh(0); j.a_bytes = 16384; j.mt_a = 8;
    j.b_nat = xenc; j.b_nat_bytes = 4096; j.nt_nat = 2;
    j.w_off = kW0; j.w_ld = 63; j.o_valid = 256; j.nat_valid = 63; j.nat_col0 = 0;
    j.bias_off = kB0; j.bias_nat_col = 63; j.kind = 2;
    add(j);
  }
  for (int l = 1; l < 8; ++l) {
    WgradJob j{};
    j.a = dh(l); j.a_bytes = 16384; j.mt_a = 8;
    j.b_acc = st_h(l - 1); j.b_acc_bytes = 16384; j.nt_acc = 8;
    j.w_off = pts_weight_off(l); j.w_ld = pts_in_dim(l); j.o_valid = 256; j.acc_valid = 256;
    j.bias_off = pts_bias_off(l);
    if (l == 4) {
      j.b_nat = xenc; j.b_nat_bytes = 4096; j.nt_nat = 2; j.nat_valid = 63; j.nat_col0 = 256; j.bias_nat_col = 63;
      j.kind = 1;
    } else {
      j.ones = 1; j.bias_nat_col = -1; j.kind = 0;
    }
    add(j);
  }
  {  // feature_layer: dFeat x h7
    WgradJob j{};
    j.a = work + bl.dfeat; j.a_bytes = 16384; j.mt_a = 8;
    j.b_acc = st_h(7); j.b_acc_bytes = 16384; j.nt_acc = 8; j.ones = 1; j.bias_nat_col = -1;
    j.w_off = kWFeat; j.w_ld = 256; j.o_valid = 256; j.acc_valid = 256; j.bias_off = kBFeat; j.kind = 0;
    if (!sl.fp8) {   // bf16 images: sigma_layer (dsmall[:,3] x h7) rides on this job's stream of h7
      j.a2 = work + bl.dsmall; j.w2_off = kWSigma; j.bias2_off = kBSigma; j.o2_row = 3; j.n2 = 257; j.kind = 13;
    }
    add(j);
  }
  if (sl.fp8) {  // sigma_layer: dsmall[:,3] x h7
    WgradJob j{};
    j.a = work + bl.dsmall; j.a_bytes = 1024; j.a_nat = 1; j.mt_a = 1; j.split_n = 1;
    j.b_acc = st_h(7); j.b_acc_bytes = 16384; j.nt_acc = 8; j.ones = 1; j.bias_nat_col = -1;
    j.w_off = kWSigma; j.w_ld = 256; j.o_row0 = 3; j.o_valid = 1; j.acc_valid = 256; j.bias_off = kBSigma; j.kind = 4;
    add(j);
  }
  {  // view_layer: dHv x [feat | denc] (bias from the direction code's constant-one column 27)
    WgradJob j{};
    j.a = work + bl.dhv; j.a_bytes = 8192; j.mt_a = 4;
    j.b_acc = stash + sl.feat; j.b_acc_bytes = 16384; j.nt_acc = 8;
    j.b_nat = denc; j.b_nat_bytes = 2048; j.nt_nat = 1; j.nat_valid = 27; j.nat_col0 = 256; j.bias_nat_col = 27;
    j.w_off = kWView; j.w_ld = 283; j.o_valid = 128; j.acc_valid = 256; j.bias_off = kBView; j.kind = 3;
    add(j);
  }
  {  // rgb_layer: dsmall[:,0:3] x hv
    WgradJob j{};
    j.a = work + bl.dsmall; j.a_bytes = 1024; j.a_nat = 1; j.mt_a = 1; j.split_n = 1;
    j.b_acc = stash + sl.hv; j.b_acc_bytes = 8192; j.nt_acc = 4; j.ones = 1; j.bias_nat_col = -1;
    j.w_off = kWRgb; j.w_ld = 128; j.o_row0 = 0; j.o_valid = 3; j.acc_valid = 128; j.bias_off = kBRgb; j.kind = 5;
    add(j);
  }
  if (part != 0) {                     // jobs are in parameter order: 0..3 = pts_layers.0..3
    const int lo = part == 1 ? 4 : 0, hi = part == 1 ? nj : 4;
    for (int j = lo; j < hi; ++j) args.jobs[j - lo] = args.jobs[j];
    nj = hi - lo;
  }
  args.n_jobs = nj;
  float* slab = bl.slab_bytes ? reinterpret_cast<float*>(const_cast<char*>(work) + bl.slab) : nullptr;
  return wgrad_launch(args, n, grads, stream, slab, bl.slab_bytes, zero_lo, zero_hi);
}

int nerf::wgrad_launch(WgradArgs& args, int64_t n, float* grads, hipStream_t stream, float* slab, size_t slab_bytes,
                       size_t zero_lo, size_t zero_hi) {
  int nj = args.n_jobs;
  // span cost of one wave tile = its bytes + a fixed per-iteration share (barrier, counted waits,
  // DMA issue, LDS reads + MFMAs of the stage).  Measured on MI355X: the iteration time is nearly
  // independent of the stage's bytes (9..36 KB), so the fixed share dominates (sweep: 4 KB ->
  // 0.88 ms, 16 KB -> 0.64, 96 KB -> 0.53); with byte-only costs the workgroups owning the narrow
  // layers (rgb: 9 KB per wave tile) ran 3x more iterations and finished last (1.11 ms).
  const int overhead = options().wgrad_overhead;
  for (int j = 0; j < nj; ++j) {
    WgradJob& jb = args.jobs[j];
    const int bytes = jb.a_bytes + jb.b_acc_bytes + jb.b_nat_bytes + (jb.a2 ? 1024 : 0);
    jb.cost = bytes + overhead;
    if (args.amax != nullptr) {
      // 8-bit images: a stage costs max(DMA time, MFMA time) + a fixed share, in shader cycles.  DMA: the
      // CU's share of the HBM stream (wgrad_bw_x16 bytes per 16 cycles); MFMA: 32 cycles each on the
      // busiest SIMD (waves w and w+4 share one; four k-steps per stage)
      const int nt = jb.nt_acc + jb.nt_nat + jb.ones;
      const int per_simd = jb.split_n ? (jb.nt_acc > 4 ? 3 : 2) : (jb.mt_a > 4 ? 2 : 1) * nt;
      // owner-mode jobs run one K = 64 MFMA (64 cycles) per column tile and stage, the split ones four K = 16 (32 each)
      const int t_mfma = (jb.split_n || options().wgrad_k16 ? 32 * 4 : 64) * per_simd, t_dma = bytes * 16 / options().wgrad_bw_x16;
      jb.cost = (t_mfma > t_dma ? t_mfma : t_dma) + options().wgrad_fixed;
    }
  }
  args.wave_tiles = args.amax != nullptr ? (int)((n + 63) / 64) : (int)((n + 31) / 32);   // ring stages per job
  long long c = 0;
  for (int j = 0; j < nj; ++j) {
    args.jobs[j].cost0 = c;
    c += (long long)args.jobs[j].cost * args.wave_tiles;
  }
  args.total_cost = c;
  args.grads = grads;
  args.debug = options().wgrad_debug;
  args.k16 = options().wgrad_k16;
  if (options().wgrad_only >= 0) {   // development aid: keep one job kind
    const int kind = options().wgrad_only;
    int m = 0;
    for (int j = 0; j < args.n_jobs; ++j) if (args.jobs[j].kind == kind) args.jobs[m++] = args.jobs[j];
    args.n_jobs = m;
    nj = m;
    long long cc = 0;
    for (int j = 0; j < nj; ++j) { args.jobs[j].cost0 = cc; cc += (long long)args.jobs[j].cost * args.wave_tiles; }
    args.total_cost = cc;
  }

  int n_cu = 0;
  if (int rc = device_cu_count(&n_cu); rc != NERF_OK) return rc;
  const bool fp8 = args.amax != nullptr;
  if (int rc = ensure_dynamic_lds(fp8 ? (const void*)mlp_wgrad_kernel<true> : (const void*)mlp_wgrad_kernel<false>, kWgLds + kWgScratch,
                                  "nerf_mlp_bwd (wgrad)"); rc != NERF_OK) return rc;
  // the tiny-MLP jobs (kinds >= 6: Instant, Part 4) run on the small-stage kernel, several workgroups per CU
  bool small = args.amax == nullptr && !options().wgrad_big_only, p4 = false;
  for (int j = 0; j < args.n_jobs; ++j) {
    const WgradJob& jb = args.jobs[j];
    p4 = p4 || (jb.kind >= 10 && jb.kind <= 12);
    small = small && jb.kind >= 6 && jb.a_bytes <= SmallStage::A && jb.b_acc_bytes <= SmallStage::B && jb.b_nat_bytes <= SmallStageP4::N;
  }
  if (p4 && !small) return fail(NERF_EINVAL, "wgrad: Part 4 job kinds run on the small-stage kernel only");
  long long want = (long long)args.wave_tiles * nj / 4;   // at least ~4 wave tiles per span
  if (options().wgrad_grid > 0 && options().wgrad_grid < n_cu) n_cu = options().wgrad_grid;
  int grid = (int)(want < 1 ? 1 : (want > n_cu ? n_cu : want));
  if (small) {
    // every workgroup ends its span with one float atomic per weight of the job: short spans on many workgroups turn the
    // launch into an atomic storm on a few thousand addresses (54 k samples on 768 workgroups: 47 us, most of it the flush)
    const long long span = options().wgrad_small_span > 0 ? options().wgrad_small_span : 4;
    // (partial tiles -- option "deterministic" -- one workgroup per CU: the tiles fit kSmallSlabBytes)
    const long long cap = (long long)(slab != nullptr ? 1 : (options().wgrad_small_cap > 0 ? options().wgrad_small_cap : 3)) * n_cu;
    const long long want3 = (long long)args.wave_tiles * nj / span;
    grid = (int)(want3 < 1 ? 1 : (want3 > cap ? cap : want3));
  }
  // slab mode: which workgroups hold a partial tile of which job -- the kernel's own span arithmetic, replayed
  args.slab = nullptr;
  args.slab_accumulate = small ? 1 : 0;       // the tiny-MLP launches ADD to grads (several passes share one gradient vector)
  if (slab != nullptr && args.n_jobs > 0 && args.total_cost > 0) {
    bool ok = true;
    long long off = 0;
    for (int j = 0; j < args.n_jobs && ok; ++j) {
      WgradJob& jb = args.jobs[j];
      const long long j0 = jb.cost0, j1 = jb.cost0 + (long long)jb.cost * args.wave_tiles;
      int first = -1, last = -1, count = 0;
      for (int b = 0; b < grid; ++b) {
        const long long lo = args.total_cost * b / grid, hi = args.total_cost * (b + 1) / grid;
        if (hi <= j0 || lo >= j1) continue;
        const long long a0 = lo > j0 ? lo - j0 : 0, a1 = (hi < j1 ? hi : j1) - j0;
        const int wt0 = (int)((a0 + jb.cost - 1) / jb.cost), wt1 = (int)((a1 + jb.cost - 1) / jb.cost);
        if (wt0 >= wt1) continue;
        if (first < 0) first = b;
        last = b;
        ++count;
      }
      ok = count > 0 && count == last - first + 1;          // contiguous (a tiny launch can leave holes: atomics then)
      jb.part0 = first;
      jb.n_parts = count;
      jb.p_stride = (jb.o_valid * jb.w_ld + jb.o_valid + jb.n2 + 63) / 64 * 64;
      jb.slab_off = off;
      off += (long long)count * jb.p_stride;
    }
    if (ok && (size_t)off * sizeof(float) <= slab_bytes) args.slab = slab;
    // the tiny jobs leave their matrices' pad columns unwritten (the atomic flush never touches them): zero tiles first
    if (args.slab != nullptr && small && hipMemsetAsync(slab, 0, (size_t)off * sizeof(float), stream) != hipSuccess)
      return fail(NERF_ELAUNCH, "tiny-MLP wgrad: memset failed");
    if (args.slab == nullptr && small) return fail(NERF_EINVAL, "tiny-MLP wgrad (option \"deterministic\"): partial tiles need %zu bytes, %zu given%s",
                                (size_t)off * sizeof(float), slab_bytes, ok ? "" : " (a job's workgroups are not contiguous)");
  }
  if (args.slab == nullptr && zero_hi > zero_lo &&
      hipMemsetAsync(grads + zero_lo, 0, sizeof(float) * (zero_hi - zero_lo), stream) != hipSuccess)
    return fail(NERF_ELAUNCH, "nerf_mlp_bwd: memset failed");
  if (small) {
    if (p4) hipLaunchKernelGGL(mlp_wgrad_small_kernel<true>, dim3(grid), dim3(512), kWgStages * SmallStageP4::Bytes, stream, args);
    else hipLaunchKernelGGL(mlp_wgrad_small_kernel<false>, dim3(grid), dim3(512), kWgStages * SmallStage::Bytes, stream, args);
    if (args.slab != nullptr) {
      if (int rc = check_launch("tiny-MLP wgrad"); rc != NERF_OK) return rc;
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(16, args.n_jobs), dim3(256), 0, stream, args);
      return check_launch("tiny-MLP wgrad (reduce)");
    }
    return check_launch("tiny-MLP wgrad");
  }
  if (fp8) hipLaunchKernelGGL(mlp_wgrad_kernel<true>, dim3(grid), dim3(512), kWgLds + kWgScratch, stream, args);
  else hipLaunchKernelGGL(mlp_wgrad_kernel<false>, dim3(grid), dim3(512), kWgLds + kWgScratch, stream, args);
  if (args.slab != nullptr) {
    if (int rc = check_launch("nerf_mlp_bwd (wgrad)"); rc != NERF_OK) return rc;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(160, args.n_jobs), dim3(256), 0, stream, args);   // 80: 14.7 us, 160: 12.7, 320: 14.3
    return check_launch("nerf_mlp_bwd (wgrad reduce)");
  }
  return check_launch("nerf_mlp_bwd (wgrad)");
}
