// Job table of the split-K weight-gradient kernel (mlp_wgrad.hip), shared by the vanilla decoder
// and the Instant tiny-MLP backward passes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nerf {

constexpr int kMaxJobs = 12;
constexpr int kWgStages = 4;
constexpr int kWgStageA = 16 * 1024, kWgStageB = 16 * 1024, kWgStageN = 4 * 1024;
constexpr int kWgStageBytes = kWgStageA + kWgStageB + kWgStageN;   // 36 KiB
constexpr int kWgLds = kWgStages * kWgStageBytes;                 // 144 KiB
constexpr int kWgScratch = 256;                                    // behind the ring: cross-wave sums at the end of a span
constexpr int kMaxTiles = 10;                                      // n-tiles a wave accumulates
// option "deterministic": partial tiles of the tiny-MLP weight-gradient launches (at most one workgroup per CU then:
// (256 + kMaxJobs) tiles of at most 6208 floats), part of the imlp / Part 4 workspaces
constexpr size_t kSmallSlabBytes = 8u << 20;

struct WgradJob {
  const char* a;        // A image
  const char* b_acc;    // blocked activations (or null)
  const char* b_nat;    // Fourier-code blocks (or null)
  int a_bytes;          // A bytes per wave tile
  int b_acc_bytes, b_nat_bytes;
  int a_nat;            // A is one 16-wide natural block (dsmall)
  int mt_a;             // 32-row tiles of A
  int nt_acc, nt_nat, ones;
  int split_n;          // single natural A block (dsmall): column tiles are split over the waves
  int kind;             // template instantiation of run_job (see the switch in the kernel)
  int w_off, w_ld;      // dW[o][i] -> grads[w_off + (o - o_row0) * w_ld + col]
  int o_row0, o_valid;
  int acc_valid, acc_col0;
  int nat_valid, nat_col0;
  int bias_off, bias_nat_col;   // bias_nat_col < 0: bias comes from the ones tile
  // second output of the merged feature + sigma job (a2 != null, bf16 images): the 16-wide natural gradient block whose row
  // o2_row contracts with the same B image into grads[w2_off .. + acc_valid) and its sum over samples into grads[bias2_off];
  // the partial tile carries n2 = acc_valid + 1 extra floats behind the bias sums
  const char* a2;
  int w2_off, bias2_off, o2_row, n2;
  long long cost0;      // prefix sum of cost (bytes per wave tile * wave tiles) before this job
  int cost;             // bytes per wave tile
  // slab mode (WgradArgs::slab): workgroups part0 .. part0 + n_parts - 1 hold a partial tile of this job, tile
  // t at slab[slab_off + t * p_stride], element (o, col) at o * w_ld + col, bias o at o_valid * w_ld + o
  long long slab_off;
  int part0, n_parts, p_stride;
};

struct WgradArgs {
  WgradJob jobs[kMaxJobs];
  int n_jobs;
  int wave_tiles;       // ring stages per job: 32-sample wave tiles (bf16) or 64-sample pairs (8-bit)
  long long total_cost;
  float* grads;
  float* slab;          // non-null: partial tiles go here with plain stores, wgrad_reduce_kernel sums them
  int slab_accumulate;  // reduce pass: 1 grads += sum of the tiles (tiny-MLP launches), 0 grads = sum
  const float* amax;    // non-null: 8-bit images; *amax = the dgrad launch's largest output-layer derivative
  int k16;              // 1: four 32x32x16 MFMAs per stage instead of one 32x32x64 (8-bit images; A/B)
  int debug;            // development aid (NERF_WGRAD_DEBUG): bit0 skip MFMA/LDS reads, bit1 skip DMA, bit2 skip flush
};


// fills cost0 / total_cost / wave_tiles / grads and launches; jobs[0..n_jobs) must be set
// slab: partial-tile memory of slab_bytes (mlp_stash.h::kSlabBytes) or null for the atomic flush; in slab mode
// every parameter of the launched jobs is OVERWRITTEN (no memset needed), in atomic mode accumulated
// zero_lo / zero_hi: the parameter range [zero_lo, zero_hi) of grads is zeroed here when (and only when) the launch
// flushes with atomics -- the slab form overwrites every parameter of its jobs, so it needs no memset
int wgrad_launch(WgradArgs& args, int64_t n_samples, float* grads, hipStream_t stream, float* slab = nullptr,
                 size_t slab_bytes = 0, size_t zero_lo = 0, size_t zero_hi = 0);

}  // namespace nerf
