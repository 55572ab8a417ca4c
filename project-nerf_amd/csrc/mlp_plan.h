// Compile-time plan of the fused 8x256 decoder shared by the weight
// packer, the forward chain kernel and the backward (dgrad) chain kernel.
//
// The decoder is evaluated as a chain of GEMM "steps"  Y[out, sample] = A[out, k] * X[k, sample]
// on v_mfma_f32_32x32x16_bf16 with the SAMPLE on the MFMA column (= lane) and features on
// the rows (= accumulator registers).  A step's fp32 accumulator tile, converted to bf16 in
// place, is already the B operand of the next step (no lane movement, no LDS): register
// 8s+j of lane-half h is row 16s + 8(j>>2) + 4h + (j&3) of the 32-row tile, so the weight
// matrix of the NEXT step is stored with its k index permuted the same way ("acc order").
// k-steps fed from features a lane computes itself (Fourier codes, output gradients) use the
// natural order k = 16*ks + 8h + j ("nat order").
//
// Weights are streamed as 1-KiB MFMA A-fragments (64 lanes x 8 bf16), in exactly the order
// the chain consumes them: step by step, m-tile by m-tile, k-step by k-step.  The stream
// is cut into chunks of <= 64 fragments at m-tile boundaries; chunks ping-pong through a
// 2 x 64 KiB LDS ring filled by global_load_lds_dwordx4.
#pragma once
#include <stdint.h>

namespace nerf {
namespace plan {

// ---- reference parameter vector (state_dict order, src/decoders.py:37-66) ----
constexpr int kPosDim = 63, kDirDim = 27, kHidden = 256, kViewDim = 128;
constexpr int kW0 = 0;                                     // pts_layers.0.weight [256,63]
constexpr int kB0 = kW0 + 256 * 63;
constexpr int kPlain = 256 * 256 + 256;                    // a 256->256 layer incl. bias
constexpr int kW1 = kB0 + 256;                             // pts_layers.1..3
constexpr int kW4 = kW1 + 3 * kPlain;                      // pts_layers.4.weight [256,319]
constexpr int kB4 = kW4 + 256 * 319;
constexpr int kW5 = kB4 + 256;                             // pts_layers.5..7
constexpr int kWSigma = kW5 + 3 * kPlain;                  // sigma_layer [1,256]
constexpr int kBSigma = kWSigma + 256;
constexpr int kWFeat = kBSigma + 1;                        // feature_layer [256,256]
constexpr int kBFeat = kWFeat + 256 * 256;
constexpr int kWView = kBFeat + 256;                       // view_layer [128,283]
constexpr int kBView = kWView + 128 * 283;
constexpr int kWRgb = kBView + 128;                        // rgb_layer [3,128]
constexpr int kBRgb = kWRgb + 3 * 128;
constexpr int kParamCount = kBRgb + 3;
static_assert(kParamCount == 595844, "parameter count must match the reference decoder");

constexpr int pts_weight_off(int l) {
  return l == 0 ? kW0 : (l < 4 ? kW1 + (l - 1) * kPlain : (l == 4 ? kW4 : kW5 + (l - 5) * kPlain));
}
constexpr int pts_in_dim(int l) { return l == 0 ? 63 : (l == 4 ? 319 : 256); }
constexpr int pts_bias_off(int l) { return pts_weight_off(l) + 256 * pts_in_dim(l); }

// ---- chain steps ----
enum Kind : int {
  F_PTS0 = 0, F_PTS1, F_PTS2, F_PTS3, F_PTS4, F_PTS5, F_PTS6, F_PTS7, F_HEAD, F_VIEW, F_RGB,  // forward
  B_RGB, B_VIEW, B_HEAD, B_PTS7, B_PTS6, B_PTS5, B_PTS4, B_PTS3, B_PTS2, B_PTS1,              // dgrad
  kNumKinds
};
constexpr int kFwdSteps = 11, kBwdSteps = 10;

struct Step {
  int mt;      // 32-row output tiles
  int ks_acc;  // k-steps (16 wide) taken from the previous step's accumulators
  int ks_nat;  // k-steps generated in registers (natural k order)
};

constexpr Step step_of(int kind) {
  switch (kind) {
    case F_PTS0: return {8, 0, 4};
    case F_PTS4: return {8, 16, 4};
    case F_HEAD: return {9, 16, 0};   // 256 feature rows + 1 sigma row (tile 8, row 0)
    case F_VIEW: return {4, 16, 2};
    case F_RGB: return {1, 8, 0};
    case B_RGB: return {4, 0, 1};     // d(hv)   = W_rgb^T  d(rgb_pre)
    case B_VIEW: return {8, 8, 0};    // d(feat) = W_view[:, :256]^T d(hv_pre)
    case B_HEAD: return {8, 16, 1};   // d(h7)   = W_feat^T d(feat) + W_sigma^T d(sigma_pre)
    default: return {8, 16, 0};       // plain 256 -> 256 (forward or transposed)
  }
}
constexpr int step_ks(int kind) { return step_of(kind).ks_acc + step_of(kind).ks_nat; }
constexpr int step_frags(int kind) { return step_of(kind).mt * step_ks(kind); }

constexpr int stream_first(bool bwd) { return bwd ? (int)B_RGB : (int)F_PTS0; }
constexpr int stream_steps(bool bwd) { return bwd ? kBwdSteps : kFwdSteps; }

constexpr int stream_frags(bool bwd) {
  int n = 0;
  for (int s = 0; s < stream_steps(bwd); ++s) n += step_frags(stream_first(bwd) + s);
  return n;
}
constexpr int kFwdFrags = stream_frags(false);   // 1184
constexpr int kBwdFrags = stream_frags(true);    // 1100
static_assert(kFwdFrags == 1184 && kBwdFrags == 1100, "stream sizes");

// first fragment of a step inside its stream
constexpr int step_frag0(int kind) {
  const bool bwd = kind >= B_RGB;
  int n = 0;
  for (int k = stream_first(bwd); k < kind; ++k) n += step_frags(k);
  return n;
}

// ---- LDS ring chunking (greedy, m-tile granular, <= 64 fragments) ----
constexpr int kChunkFrags = 64;
constexpr int kMaxGroups = 96;   // m-tiles per stream (78 fwd, 76 bwd)
constexpr int kMaxChunks = 32;

struct Chunks {
  int n_groups;
  int n_chunks;
  int group_chunk[kMaxGroups];   // chunk of m-tile group g
  int group_off[kMaxGroups];     // fragment offset of the group inside its chunk
  bool group_first[kMaxGroups];  // group opens a new chunk
  int chunk_frag0[kMaxChunks];   // first stream fragment of the chunk
  int chunk_count[kMaxChunks];   // fragments in the chunk
};

constexpr Chunks make_chunks(bool bwd) {
  Chunks c{};
  int g = 0, chunk = -1, fill = kChunkFrags + 1, frag = 0;
  for (int s = 0; s < stream_steps(bwd); ++s) {
    const int kind = stream_first(bwd) + s;
    const int ks = step_ks(kind);
    for (int m = 0; m < step_of(kind).mt; ++m, ++g) {
      const bool open = fill + ks > kChunkFrags;
      if (open) {
        ++chunk;
        fill = 0;
        c.chunk_frag0[chunk] = frag;
        c.chunk_count[chunk] = 0;
      }
      c.group_chunk[g] = chunk;
      c.group_off[g] = fill;
      c.group_first[g] = open;
      fill += ks;
      frag += ks;
      c.chunk_count[chunk] += ks;
    }
  }
  c.n_groups = g;
  c.n_chunks = chunk + 1;
  return c;
}
constexpr Chunks kFwdChunks = make_chunks(false);
constexpr Chunks kBwdChunks = make_chunks(true);

// global m-tile group index of (kind, m)
constexpr int group_of(int kind, int m) {
  const bool bwd = kind >= B_RGB;
  int g = 0;
  for (int k = stream_first(bwd); k < kind; ++k) g += step_of(k).mt;
  return g + m;
}

// ---- forward bias table (fp32, added as the accumulator's initial value) ----
constexpr int bias_off(int kind) {
  return kind <= F_PTS7 ? 256 * kind : (kind == F_HEAD ? 2048 : (kind == F_VIEW ? 2048 + 288 : 2048 + 288 + 128));
}
constexpr int kBiasFloats = 2048 + 288 + 128 + 32;   // 2496

// ---- packed buffer layout (bytes) ----
constexpr size_t kFragBytes = 1024;
constexpr size_t kPackFwdOff = 0;
constexpr size_t kStreamPad = 8 * kFragBytes;   // DMA tail pieces may read this far past a stream
constexpr size_t kPackBwdOff = kPackFwdOff + (size_t)kFwdFrags * kFragBytes + kStreamPad;
constexpr size_t kPackBiasOff = kPackBwdOff + (size_t)kBwdFrags * kFragBytes + kStreamPad;
// forward stream for the 16x16x32 MFMA shape (inference): same fragment count and chunking; fragment k of an
// m-tile = 16-row half k & 1 at the 32-deep k-step k >> 1 (gen_stream_asm.py::fwd_groups)
constexpr size_t kPackFwd16Off = kPackBiasOff + ((kBiasFloats * 4 + 255) / 256) * 256;
constexpr size_t kPackBytes = kPackFwd16Off + (size_t)kFwdFrags * kFragBytes + kStreamPad;

// ---- training stash: blocked images [xenc 64 | h0..h7 8x256 | feat 256 | hv 128 | denc 32] per
// sample, followed by relu bitmasks; byte layout in mlp_stash.h, block shapes in mlp_chain.h.
constexpr int kStashXenc = 64, kStashDenc = 32;

}  // namespace plan
}  // namespace nerf
