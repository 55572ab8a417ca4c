// Byte layout of the training stash (written by nerf_mlp_fwd, read by nerf_mlp_bwd) and of
// the backward workspace (dgrad chain -> wgrad).  All matrices are blocked images with the
// sample count rounded up to whole 256-sample tiles; every offset is 256-byte aligned.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include "common.h"

namespace nerf {

// Two families of chain kernels: hand-scheduled asm streams (gen_stream_asm.py, the default) and
// the compiler-scheduled kernels built from mlp_chain.h::run_step.
//   inference        : stream (+10 % render FPS)
//   training (stash) : stream (fwd 0.33 -> 0.29 ms, dgrad 0.31 -> 0.28 ms once the stash stores are
//                      non-temporal; before that the stream dgrad lost 0.07 ms waiting on its mask
//                      loads behind the stores).  The streams' 32-bit image offsets cover 2^22
//                      samples per launch; larger launches take the compiler-scheduled kernels.
//   option chain_legacy (env NERF_CHAIN_LEGACY=1, or nerf_set_option) selects the compiler-scheduled
//   family everywhere (development aid, tests).
// Forward and backward of one step must decide alike: the ReLU
// mask words differ between the families (stream: 16 bits per lane and m-tile, bit q / 8+q =
// rows 2q / 2q+1; compiler-scheduled: 16 bits per m-tile, bit r = accumulator register r).
inline bool chain_use_stream(int64_t n, bool training) {
  if (options().chain_legacy) return false;
  return !training || n <= ((int64_t)1 << 22);
}

// Element width of the training images.  The asm-stream family writes 8-bit images: e4m3
// activations (forward) and e5m2 pre-activation gradients (dgrad), both consumed by
// v_mfma_f32_32x32x16_bf8_fp8 in the wgrad kernel -- half the bytes of every image crosses HBM
// (step traffic 5.8 -> 3.0 GB at 4096 x 64 samples).  The compiler-scheduled family keeps bf16.
// DEFAULT: bf16 images -- every MFMA of the training step then contracts bf16 operands, the precision BASELINE.json
// configs[1] names.  Option stash_fp8 (NERF_STASH_FP8=1) selects the 8-bit images (narrower than the config's
// precision: reported as a labelled secondary figure by bench.py, never as the headline).
inline bool stash_fp8(int64_t n) { return options().stash_fp8 != 0 && chain_use_stream(n, true); }
// divisor of the activation images before the e4m3 conversion (a power of two).  1: the image
// saturates at 448 and flushes below 2^-10; the chain's own bf16 values are unaffected.
constexpr float kActScale = 1.0f;

struct StashLayout {
  int64_t n_pad;
  int fp8;          // 1: 8-bit images (1 byte per element), 0: bf16
  size_t xenc, h, feat, hv, denc, mask, total;
};

inline StashLayout stash_layout(int64_t n) {
  StashLayout s{};
  s.n_pad = (n + 255) / 256 * 256;
  s.fp8 = stash_fp8(n) ? 1 : 0;
  const size_t np = (size_t)s.n_pad, eb = s.fp8 ? 1 : 2;
  size_t o = 0;
  s.xenc = o; o += np * 64 * eb;
  s.h = o;    o += np * 256 * eb * 8;
  s.feat = o; o += np * 256 * eb;
  s.hv = o;   o += np * 128 * eb;
  s.denc = o; o += np * 32 * eb;
  s.mask = o; o += (np / 256) * 9 * 512 * 16;   // both families: 16 mask bits per lane and m-tile
  s.total = o;
  return s;
}

// backward workspace: blocked gradients w.r.t. pre-activations (bf16, or e5m2 divided by the
// power-of-two scale the dgrad launch derives from the largest output-layer derivative)
struct BwdLayout {
  int64_t n_pad;
  int fp8;
  size_t dsmall;   // nat [n_pad,16]: cols 0..2 d(rgb_pre), col 3 d(sigma_pre)
  size_t dhv;      // blocked [n_pad,128]
  size_t dfeat;    // blocked [n_pad,256]
  size_t dh;       // 8 x blocked [n_pad,256], layer l at dh + l * n_pad * 256 * element bytes
  size_t amax;     // one fp32: max |output-layer derivative| of this launch (8-bit images only)
  size_t slab;     // partial weight-gradient tiles of the split-K wgrad (slab_bytes; 0 = float atomics instead)
  size_t slab_bytes;
  size_t total;
};

// Partial sums of the split-K weight-gradient kernel: one tile per (workgroup, layer job) written with plain
// stores and summed by a second small kernel in a fixed order, instead of ~18 M float atomics per step (they
// retire per line request, ~20 G/s: 0.06 ms exposed at the end of the kernel).  Capacity: every workgroup's
// largest tile (pts_layers.4: 256 x 319 + 256 floats) plus one extra tile per job.  Launches below
// kSlabMinSamples keep the atomic flush (their grids are small and the workspace stays small).
constexpr int64_t kSlabMinSamples = 65536;
constexpr size_t kSlabMaxWorkgroups = 320, kSlabMaxTileFloats = 256 * 319 + 256;
constexpr size_t kSlabBytes = (kSlabMaxWorkgroups + 12) * kSlabMaxTileFloats * sizeof(float);

inline BwdLayout bwd_layout(int64_t n) {
  BwdLayout s{};
  s.n_pad = (n + 255) / 256 * 256;
  s.fp8 = stash_fp8(n) ? 1 : 0;
  const size_t np = (size_t)s.n_pad, eb = s.fp8 ? 1 : 2;
  size_t o = 0;
  s.dsmall = o; o += np * 16 * eb;
  s.dhv = o;    o += np * 128 * eb;
  s.dfeat = o;  o += np * 256 * eb;
  s.dh = o;     o += np * 256 * eb * 8;
  s.amax = o;   o += 256;
  s.slab = o;
  s.slab_bytes = (n >= kSlabMinSamples && !options().wgrad_atomic) ? kSlabBytes : 0;
  o += s.slab_bytes;
  s.total = o;
  return s;
}

// gradient-image divisor from the launch's amax: a power of two that puts amax in [64, 128)
// (e5m2 tops out at 57344: ~9 binades of headroom for growth along the chain, 20 below)
__host__ __device__ inline float grad_image_scale(float amax) {
  const unsigned bits = __builtin_bit_cast(unsigned, amax);
  int e = (int)((bits >> 23) & 0xffu) - 6;
  e = e < 1 ? 1 : (e > 254 ? 254 : e);
  return __builtin_bit_cast(float, (unsigned)e << 23);
}

}  // namespace nerf
