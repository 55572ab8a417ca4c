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
// mask words differ between the families (stream: one dword per lane and m-tile, bit q / 16+q =
// rows 2q / 2q+1; compiler-scheduled: 16 bits per m-tile, bit r = accumulator register r).
inline bool chain_use_stream(int64_t n, bool training) {
  if (options().chain_legacy) return false;
  return !training || n <= ((int64_t)1 << 22);
}

struct StashLayout {
  int64_t n_pad;
  size_t xenc, h, feat, hv, denc, mask, total;
};

inline StashLayout stash_layout(int64_t n) {
  StashLayout s{};
  s.n_pad = (n + 255) / 256 * 256;
  const size_t np = (size_t)s.n_pad;
  size_t o = 0;
  s.xenc = o; o += np * 64 * 2;
  s.h = o;    o += np * 256 * 2 * 8;
  s.feat = o; o += np * 256 * 2;
  s.hv = o;   o += np * 128 * 2;
  s.denc = o; o += np * 32 * 2;
  s.mask = o; o += (np / 256) * 9 * 512 * 32;
  s.total = o;
  return s;
}

// backward workspace: bf16 blocked gradients w.r.t. pre-activations
struct BwdLayout {
  int64_t n_pad;
  size_t dsmall;   // nat [n_pad,16]: cols 0..2 d(rgb_pre), col 3 d(sigma_pre)
  size_t dhv;      // blocked [n_pad,128]
  size_t dfeat;    // blocked [n_pad,256]
  size_t dh;       // 8 x blocked [n_pad,256], layer l at dh + l * n_pad * 512
  size_t total;
};

inline BwdLayout bwd_layout(int64_t n) {
  BwdLayout s{};
  s.n_pad = (n + 255) / 256 * 256;
  const size_t np = (size_t)s.n_pad;
  size_t o = 0;
  s.dsmall = o; o += np * 16 * 2;
  s.dhv = o;    o += np * 128 * 2;
  s.dfeat = o;  o += np * 256 * 2;
  s.dh = o;     o += np * 256 * 2 * 8;
  s.total = o;
  return s;
}

}  // namespace nerf
