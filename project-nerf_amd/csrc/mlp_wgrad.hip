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
// whole span), bias gradients fall out of an all-ones column.  Partial sums are flushed with
// float atomics (two 128-byte segments per wave instruction).
// HBM-bound: every stashed byte is read exactly once (about 10.4 KB per sample).
#include "mlp_chain.h"
#include "mlp_stash.h"

namespace nerf {
using namespace plan;

constexpr int kMaxJobs = 12;
constexpr int kWgStages = 4;
constexpr int kWgStageA = 16 * 1024, kWgStageB = 16 * 1024, kWgStageN = 4 * 1024;
constexpr int kWgStageBytes = kWgStageA + kWgStageB + kWgStageN;   // 36 KiB
constexpr int kWgLds = kWgStages * kWgStageBytes;                 // 144 KiB
constexpr int kMaxTiles = 10;                                      // n-tiles a wave accumulates

struct WgradJob {
  const char* a;        // A image
  const char* b_acc;    // blocked activations (or null)
  const char* b_nat;    // Fourier-code blocks (or null)
  int a_bytes;          // A bytes per wave tile
  int b_acc_bytes, b_nat_bytes;
  int a_nat;            // A is one 16-wide natural block (dsmall)
  int mt_a;             // 32-row tiles of A
  int nt_acc, nt_nat, ones;
  int split_n;          // single-m-tile job: wave w owns n-tiles {w, w+8}
  int w_off, w_ld;      // dW[o][i] -> grads[w_off + (o - o_row0) * w_ld + col]
  int o_row0, o_valid;
  int acc_valid, acc_col0;
  int nat_valid, nat_col0;
  int bias_off, bias_nat_col;   // bias_nat_col < 0: bias comes from the ones tile
  long long cost0;      // prefix sum of cost (bytes per wave tile * wave tiles) before this job
  int cost;             // bytes per wave tile
};

struct WgradArgs {
  WgradJob jobs[kMaxJobs];
  int n_jobs;
  int wave_tiles;
  long long total_cost;
  float* grads;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// Transposing LDS reads as inline asm: hipcc orders every LDS read it can see behind ALL
// pending LDS-DMA writes (s_waitcnt vmcnt(0)), which would drain the prefetch ring on every
// fragment.  The DMA -> read ordering is done by hand instead (counted vmcnt + s_barrier in
// the stage loop); the reads and their lgkmcnt wait live in one asm statement.
__device__ __forceinline__ unsigned lds_addr(const char* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
// N transposing reads + one wait in a single statement (outputs early-clobber)
__device__ __forceinline__ void tr_read12(const unsigned (&ad)[12], s16x4 (&o)[12]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %12\n\tds_read_b64_tr_b16 %1, %13\n\tds_read_b64_tr_b16 %2, %14\n\t"
      "ds_read_b64_tr_b16 %3, %15\n\tds_read_b64_tr_b16 %4, %16\n\tds_read_b64_tr_b16 %5, %17\n\t"
      "ds_read_b64_tr_b16 %6, %18\n\tds_read_b64_tr_b16 %7, %19\n\tds_read_b64_tr_b16 %8, %20\n\t"
      "ds_read_b64_tr_b16 %9, %21\n\tds_read_b64_tr_b16 %10, %22\n\tds_read_b64_tr_b16 %11, %23\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
        "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11])
      : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7]),
        "v"(ad[8]), "v"(ad[9]), "v"(ad[10]), "v"(ad[11])
      : "memory");
}
__device__ __forceinline__ void tr_read10(const unsigned (&ad)[10], s16x4 (&o)[10]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %10\n\tds_read_b64_tr_b16 %1, %11\n\tds_read_b64_tr_b16 %2, %12\n\t"
      "ds_read_b64_tr_b16 %3, %13\n\tds_read_b64_tr_b16 %4, %14\n\tds_read_b64_tr_b16 %5, %15\n\t"
      "ds_read_b64_tr_b16 %6, %16\n\tds_read_b64_tr_b16 %7, %17\n\tds_read_b64_tr_b16 %8, %18\n\t"
      "ds_read_b64_tr_b16 %9, %19\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
        "=&v"(o[8]), "=&v"(o[9])
      : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7]),
        "v"(ad[8]), "v"(ad[9])
      : "memory");
}
__device__ __forceinline__ bf16x8 frag_of(const s16x4& lo, const s16x4& hi) {
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

__global__ void __launch_bounds__(512, 2) mlp_wgrad_kernel(const WgradArgs args) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // transposing-read lane geometry (see header comment of mlp_chain.h stash_block)
  const int grp = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
  const int hh = grp >> 1, fhalf = grp & 1;
  const int off_acc = 64 * (8 * hh + q) + 32 * (p & 1) + 16 * fhalf + 8 * (p >> 1);   // + 1024*s, block*2048
  const int off_nat = 32 * (8 * hh + q) + 16 * (p >> 1) + 8 * (p & 1) + 1024 * fhalf;  // + 512*s, pair*2048

  // this workgroup's span of the cost line
  const long long lo = args.total_cost * blockIdx.x / gridDim.x;
  const long long hi = args.total_cost * (blockIdx.x + 1) / gridDim.x;

  for (int j = 0; j < args.n_jobs; ++j) {
    const WgradJob& job = args.jobs[j];
    const long long j0 = job.cost0, j1 = job.cost0 + (long long)job.cost * args.wave_tiles;
    if (hi <= j0 || lo >= j1) continue;
    // wave tile t belongs to the workgroup whose span contains its first cost unit
    const long long a0 = lo > j0 ? lo - j0 : 0, a1 = (hi < j1 ? hi : j1) - j0;
    const int wt0 = (int)((a0 + job.cost - 1) / job.cost), wt1 = (int)((a1 + job.cost - 1) / job.cost);
    if (wt0 >= wt1) continue;

    const int nt_total = job.nt_acc + job.nt_nat + job.ones;
    int m_tile, n_first, n_step, n_count;
    if (job.split_n) {
      m_tile = 0; n_first = wave; n_step = 8; n_count = wave < nt_total ? (nt_total - wave + 7) / 8 : 0;
    } else {
      m_tile = wave; n_first = 0; n_step = 1; n_count = wave < job.mt_a ? nt_total : 0;
    }

    f32x16 acc[kMaxTiles];
#pragma unroll
    for (int k = 0; k < kMaxTiles; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;

    const int pieces_a = job.a_bytes >> 10, pieces_b = job.b_acc_bytes >> 10, pieces_n = job.b_nat_bytes >> 10;
    const int pieces = pieces_a + pieces_b + pieces_n;
    const int per_wave = (pieces + 7) >> 3;    // every wave issues exactly this many (tail duplicates)
    auto issue = [&](int wt) {
      char* stage = smem + (wt & (kWgStages - 1)) * kWgStageBytes;
      for (int i = 0; i < per_wave; ++i) {
        int pc = wave + 8 * i;
        pc = pc < pieces ? pc : pieces - 1;
        const char* src;
        char* dst;
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
        __builtin_amdgcn_global_load_lds((gptr_t)(src + lane * 16), (lptr_t)dst, 16, 0, 0);
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

    __builtin_amdgcn_s_barrier();   // previous job's readers are done with the ring
    if (wt0 + 0 < wt1) issue(wt0 + 0);
    if (wt0 + 1 < wt1) issue(wt0 + 1);
    if (wt0 + 2 < wt1) issue(wt0 + 2);
    for (int wt = wt0; wt < wt1; ++wt) {
      const int ahead = (wt + 1 < wt1) + (wt + 2 < wt1);
      wait_in_flight(ahead);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (wt + 3 < wt1) issue(wt + 3);
      const char* stage = smem + (wt & (kWgStages - 1)) * kWgStageBytes;
      if (n_count > 0) {
        const unsigned st = lds_addr(stage);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          // operand addresses: pair (first read, second read = +4 samples) per fragment
          auto tile_addr = [&](int k, unsigned& a0, unsigned& a1) {
            const int nt = n_first + k * n_step;
            if (k < n_count && nt < job.nt_acc) {
              a0 = st + kWgStageA + nt * 2048 + off_acc + 1024 * s; a1 = a0 + 256;
            } else if (k < n_count && nt < job.nt_acc + job.nt_nat) {
              a0 = st + kWgStageA + kWgStageB + (nt - job.nt_acc) * 2048 + off_nat + 512 * s; a1 = a0 + 128;
            } else {
              a0 = st; a1 = st;     // ones tile / unused slot: any valid address
            }
          };
          unsigned ad0[12];
          s16x4 r0[12];
          if (job.a_nat) { ad0[0] = st + off_nat - 1024 * fhalf + 512 * s; ad0[1] = ad0[0] + 128; }
          else { ad0[0] = st + m_tile * 2048 + off_acc + 1024 * s; ad0[1] = ad0[0] + 256; }
#pragma unroll
          for (int k = 0; k < 5; ++k) tile_addr(k, ad0[2 + 2 * k], ad0[3 + 2 * k]);
          tr_read12(ad0, r0);
          bf16x8 af = frag_of(r0[0], r0[1]);
          if (job.a_nat && fhalf) {
#pragma unroll
            for (int e = 0; e < 8; ++e) af[e] = (__bf16)0.0f;
          }
          bf16x8 ones;
#pragma unroll
          for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
#pragma unroll
          for (int k = 0; k < 5; ++k) {
            if (k < n_count) {
              const int nt = n_first + k * n_step;
              const bf16x8 bfrag = nt < job.nt_acc + job.nt_nat ? frag_of(r0[2 + 2 * k], r0[3 + 2 * k]) : ones;
              acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfrag, acc[k], 0, 0, 0);
            }
          }
          if (n_count > 5) {
            unsigned ad1[10];
            s16x4 r1[10];
#pragma unroll
            for (int k = 0; k < 5; ++k) tile_addr(5 + k, ad1[2 * k], ad1[2 * k + 1]);
            tr_read10(ad1, r1);
#pragma unroll
            for (int k = 0; k < 5; ++k) {
              if (5 + k < n_count) {
                const int nt = n_first + (5 + k) * n_step;
                const bf16x8 bfrag = nt < job.nt_acc + job.nt_nat ? frag_of(r1[2 * k], r1[2 * k + 1]) : ones;
                acc[5 + k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfrag, acc[5 + k], 0, 0, 0);
              }
            }
          }
        }
      }
    }

    // ---- flush: row o = 32*m_tile + (r&3) + 8*(r>>2) + 4*hh, column = 32*nt + (lane&31) ----
    const int c32 = lane & 31, hrow = lane >> 5;
#pragma unroll
    for (int k = 0; k < kMaxTiles; ++k) {
      if (k < n_count) {
        const int nt = n_first + k * n_step;
        int col = -1, bias_here = 0;
        if (nt < job.nt_acc) {
          const int i = nt * 32 + c32;
          if (i < job.acc_valid) col = job.acc_col0 + i;
        } else if (nt < job.nt_acc + job.nt_nat) {
          const int i = (nt - job.nt_acc) * 32 + c32;
          if (i < job.nat_valid) col = job.nat_col0 + i;
          bias_here = (i == job.bias_nat_col);
        } else {
          bias_here = (c32 == 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = 32 * m_tile + (r & 3) + 8 * (r >> 2) + 4 * hrow - job.o_row0;
          if (o >= 0 && o < job.o_valid) {
            if (col >= 0) atomicAdd(args.grads + job.w_off + o * job.w_ld + col, acc[k][r]);
            if (bias_here) atomicAdd(args.grads + job.bias_off + o, acc[k][r]);
          }
        }
      }
    }
  }
}

}  // namespace nerf

using namespace nerf;

int nerf_launch_wgrad(const char* stash, const StashLayout& sl, const char* work, const BwdLayout& bl,
                      int64_t n, float* grads, hipStream_t stream) {
  WgradArgs args{};
  const size_t np = (size_t)sl.n_pad;
  const char* xenc = stash + sl.xenc;
  const char* denc = stash + sl.denc;
  auto st_h = [&](int l) { return stash + sl.h + (size_t)l * np * 512; };
  auto dh = [&](int l) { return work + bl.dh + (size_t)l * np * 512; };
  int nj = 0;
  auto add = [&](WgradJob j) {
    j.cost = j.a_bytes + j.b_acc_bytes + j.b_nat_bytes;
    args.jobs[nj++] = j;
  };
  // pts_layers.0: dH0 x xenc (bias from the code's constant-one column 63)
  {
    WgradJob j{};
    j.a = dh(0); j.a_bytes = 16384; j.mt_a = 8;
    j.b_nat = xenc; j.b_nat_bytes = 4096; j.nt_nat = 2;
    j.w_off = kW0; j.w_ld = 63; j.o_valid = 256; j.nat_valid = 63; j.nat_col0 = 0;
    j.bias_off = kB0; j.bias_nat_col = 63;
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
    } else {
      j.ones = 1; j.bias_nat_col = -1;
    }
    add(j);
  }
  {  // feature_layer: dFeat x h7
    WgradJob j{};
    j.a = work + bl.dfeat; j.a_bytes = 16384; j.mt_a = 8;
    j.b_acc = st_h(7); j.b_acc_bytes = 16384; j.nt_acc = 8; j.ones = 1; j.bias_nat_col = -1;
    j.w_off = kWFeat; j.w_ld = 256; j.o_valid = 256; j.acc_valid = 256; j.bias_off = kBFeat;
    add(j);
  }
  {  // sigma_layer: dsmall[:,3] x h7
    WgradJob j{};
    j.a = work + bl.dsmall; j.a_bytes = 1024; j.a_nat = 1; j.mt_a = 1; j.split_n = 1;
    j.b_acc = st_h(7); j.b_acc_bytes = 16384; j.nt_acc = 8; j.ones = 1; j.bias_nat_col = -1;
    j.w_off = kWSigma; j.w_ld = 256; j.o_row0 = 3; j.o_valid = 1; j.acc_valid = 256; j.bias_off = kBSigma;
    add(j);
  }
  {  // view_layer: dHv x [feat | denc] (bias from the direction code's constant-one column 27)
    WgradJob j{};
    j.a = work + bl.dhv; j.a_bytes = 8192; j.mt_a = 4;
    j.b_acc = stash + sl.feat; j.b_acc_bytes = 16384; j.nt_acc = 8;
    j.b_nat = denc; j.b_nat_bytes = 2048; j.nt_nat = 1; j.nat_valid = 27; j.nat_col0 = 256; j.bias_nat_col = 27;
    j.w_off = kWView; j.w_ld = 283; j.o_valid = 128; j.acc_valid = 256; j.bias_off = kBView;
    add(j);
  }
  {  // rgb_layer: dsmall[:,0:3] x hv
    WgradJob j{};
    j.a = work + bl.dsmall; j.a_bytes = 1024; j.a_nat = 1; j.mt_a = 1; j.split_n = 1;
    j.b_acc = stash + sl.hv; j.b_acc_bytes = 8192; j.nt_acc = 4; j.ones = 1; j.bias_nat_col = -1;
    j.w_off = kWRgb; j.w_ld = 128; j.o_row0 = 0; j.o_valid = 3; j.acc_valid = 128; j.bias_off = kBRgb;
    add(j);
  }
  args.n_jobs = nj;
  args.wave_tiles = (int)((n + 31) / 32);
  long long c = 0;
  for (int j = 0; j < nj; ++j) {
    args.jobs[j].cost0 = c;
    c += (long long)args.jobs[j].cost * args.wave_tiles;
  }
  args.total_cost = c;
  args.grads = grads;

  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
      return fail(NERF_ELAUNCH, "nerf_mlp_bwd: cannot query device");
    n_cu = prop.multiProcessorCount;
    if (hipFuncSetAttribute((const void*)mlp_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kWgLds) != hipSuccess)
      return fail(NERF_ELAUNCH, "nerf_mlp_bwd: cannot raise dynamic LDS limit to %d", kWgLds);
  }
  long long want = (long long)args.wave_tiles * nj / 4;   // at least ~4 wave tiles per span
  int grid = (int)(want < 1 ? 1 : (want > n_cu ? n_cu : want));
  hipLaunchKernelGGL(mlp_wgrad_kernel, dim3(grid), dim3(512), kWgLds, stream, args);
  return check_launch("nerf_mlp_bwd (wgrad)");
}
