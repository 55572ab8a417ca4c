// Multiresolution hash-grid encoding, forward gather and backward scatter-add.
// Replaces tinycudann's HashGrid encoding behind HashRepresentation (reference
// src/embeddings.py:39-93).  The third-party source is absent, so this follows the published
// Instant-NGP algorithm (Mueller et al. 2022, section 3) with the level table defined in
// oracle/nerf_oracle.py::hash_grid_levels -- PARITY UNPINNED against tinycudann itself; bit-exact
// indices and fp32-accurate features against the build's own CPU restatement.
//
// (Pairing the x-neighbour gathers of the forward on two lanes, as the backward does for its atomics,
// was measured: 0.17 vs 0.16 ms -- gathers are not bound by the line-request rate.)
// One thread per (point, level): normalise + clamp the point, 8 corner indices (dense below the
// hash-map budget, xor-prime hash above), trilinear blend of F = 2 features.  Gather-bound:
// 16 levels x 8 corners x 8 B (fp32 table) per point; the 52 MB table lives in the 256 MB
// Infinity Cache.  Backward: 16 float atomics per (point, level).
#include <stdlib.h>
#include "common.h"

namespace nerf {

constexpr int kMaxLevels = 16;

struct HashLevels {
  float scale[kMaxLevels];
  unsigned res[kMaxLevels];
  unsigned size[kMaxLevels];
  unsigned offset[kMaxLevels];
  unsigned dense[kMaxLevels];
  int n_levels;
  float bound;
};

struct Corner {
  unsigned idx[8];
  float w[8];
};

__device__ __forceinline__ Corner corners_of(const HashLevels& L, int lvl, float px, float py, float pz) {
  // HashRepresentation.forward: (x + bound) / (2 bound), clamp to [0, 1]  (embeddings.py:86-87)
  const float two_b = 2.0f * L.bound;
  float x01[3] = {add_rn(px, L.bound) / two_b, add_rn(py, L.bound) / two_b, add_rn(pz, L.bound) / two_b};
  unsigned cell[3];
  float frac[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float c = fminf(fmaxf(x01[a], 0.0f), 1.0f);
    const float pos = add_rn(mul_rn(c, L.scale[lvl]), 0.5f);
    const float fl = floorf(pos);
    frac[a] = sub_rn(pos, fl);
    cell[a] = (unsigned)fl;
  }
  Corner out;
  const unsigned res = L.res[lvl], size = L.size[lvl];
  const bool pow2 = (size & (size - 1)) == 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const unsigned gx = cell[0] + (c & 1), gy = cell[1] + ((c >> 1) & 1), gz = cell[2] + ((c >> 2) & 1);
    const float wx = (c & 1) ? frac[0] : sub_rn(1.0f, frac[0]);
    const float wy = (c & 2) ? frac[1] : sub_rn(1.0f, frac[1]);
    const float wz = (c & 4) ? frac[2] : sub_rn(1.0f, frac[2]);
    out.w[c] = mul_rn(mul_rn(wx, wy), wz);
    unsigned e;
    if (L.dense[lvl]) {
      // (gx + gy res + gz res^2) % size: every coordinate is at most res and size >= res^3, so the index is below 2 size --
      // one conditional subtraction gives the remainder (a runtime udiv costs ~30 VALU ops per corner)
      e = gx + gy * res + gz * res * res;
      e = e >= size ? e - size : e;
    }
    else {
      const unsigned h = (gx * 1u) ^ (gy * 2654435761u) ^ (gz * 805459861u);
      e = pow2 ? (h & (size - 1)) : (h % size);     // same value; a runtime udiv costs ~30 VALU ops per corner
    }
    out.idx[c] = e + L.offset[lvl];
  }
  return out;
}

// out_f32 [n, 2L] row-major (optional), out_nat: bf16 natural-order operand blocks (optional):
// [wave tile of 32 points][k-step ks = level/8][lane (c, h)][8] with feature 16ks + 8h + j
// TableT = float2: the fp32 parameters themselves.  TableT = half2_t: an fp16 copy kept by the optimiser
// (nerf_adamw_clip_step_shadow) -- half the bytes per gather and twice the entries per cache line; tinycudann
// itself evaluates its grid from fp16 parameters next to an fp32 master copy.
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
// table slices of the binned backward (see below): the forward can already count the corners per (level, slice)
constexpr unsigned kSliceLog2 = 12, kSlice = 1u << kSliceLog2;   // 4096 entries x 2 features x 8 B = 64 KiB of LDS
struct LevelBins { unsigned bin0[kMaxLevels + 1]; };             // first bin of every level; bin0[n_levels] = number of bins
__device__ __forceinline__ float2 table_entry(const float2* t, unsigned i) { return t[i]; }
__device__ __forceinline__ float2 table_entry(const half2_t* t, unsigned i) {
  const half2_t h = t[i];
  return make_float2((float)h[0], (float)h[1]);
}
// Which (level, point chunk) a workgroup of the gather kernels works on.  n_chunks > 0 -- XCD-aware 1-D launch of
// 8 * ceil(n_levels / 8) * n_chunks workgroups: the hardware deals consecutive workgroup ids to the eight XCDs in turn, so
// workgroup b runs on XCD b % 8; it takes level (b % 8) + 8 * ((b / 8) / n_chunks), i.e. EVERY lookup into the tables of levels
// x and x + 8 goes through the L2 of XCD x: one hashed level (2 MB of fp16 entries) and one coarse level per 4-MB L2, each
// table fetched from HBM once per launch instead of once per XCD (level-major 2-D launch: 335 MB fetched for 105 MB of
// gathers on 200 k points).  n_chunks == 0: the 2-D launch (blockIdx.y = level).
struct LevelChunk { int lvl, chunk, chunks; };
__device__ __forceinline__ LevelChunk level_chunk(int n_levels, int n_chunks) {
  if (n_chunks <= 0) return {(int)blockIdx.y, (int)blockIdx.x, (int)gridDim.x};
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  return {xcd + 8 * (j / n_chunks), j % n_chunks, n_chunks};
}
inline dim3 level_chunk_grid(int n_levels, int64_t blocks, bool xcd_aware) {
  return xcd_aware ? dim3((unsigned)(8 * ((n_levels + 7) / 8) * blocks)) : dim3((unsigned)blocks, (unsigned)n_levels);
}

// The eight entries of a cell, gathered with as few cache-line requests as the addresses allow.  Corners 2j and 2j+1 differ in x
// only: on a dense level their entries are neighbours (e, e + 1), on a hashed level with an even cell x they are e and e ^ 1
// (the hash XORs x in unmultiplied) -- either way one 8-byte (fp16 table) / 16-byte (fp32) load at min(e0, e1) returns both.
// Otherwise (hashed level, odd x: the carry changes higher bits) the second corner takes its own load.  The gather kernels are
// bound by the texture path's line rate (one 64-byte line per lane and request on the hashed levels), not by bytes.
struct half2x2 { half2_t a, b; };
__device__ __forceinline__ void table_pair(const float2* t, unsigned lo, float2& v0, float2& v1) {
  float4 q;
  __builtin_memcpy(&q, reinterpret_cast<const char*>(t + lo), 16);      // 8-byte aligned: two dwordx2 or one dwordx4
  v0 = make_float2(q.x, q.y);
  v1 = make_float2(q.z, q.w);
}
__device__ __forceinline__ void table_pair(const half2_t* t, unsigned lo, float2& v0, float2& v1) {
  half2x2 q;
  __builtin_memcpy(&q, reinterpret_cast<const char*>(t + lo), 8);       // 4-byte aligned dwordx2
  v0 = make_float2((float)q.a[0], (float)q.a[1]);
  v1 = make_float2((float)q.b[0], (float)q.b[1]);
}
template <class TableT>
__device__ __forceinline__ void gather_cell(const TableT* __restrict__ table, const Corner& c, float2 (&v)[8]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned e0 = c.idx[2 * j], e1 = c.idx[2 * j + 1], lo = e0 < e1 ? e0 : e1;
    if ((e0 > e1 ? e0 - e1 : e1 - e0) == 1u) {
      float2 p0, p1;
      table_pair(table, lo, p0, p1);                       // entries lo and lo + 1 = the two corners
      v[2 * j] = e0 == lo ? p0 : p1;
      v[2 * j + 1] = e0 == lo ? p1 : p0;
    } else {
      v[2 * j] = table_entry(table, e0);
      v[2 * j + 1] = table_entry(table, e1);
    }
  }
}

struct TableSet { int n; long long table_stride, out_stride; };     // n tables, entries between them, operand-image elements between their outputs

template <class TableT>
__global__ void __launch_bounds__(256)
hash_fwd_kernel(const float* __restrict__ pts, int64_t n, int64_t n_pad, const TableT* __restrict__ table, HashLevels L,
                float* __restrict__ out_f32, __bf16* __restrict__ out_nat, unsigned* __restrict__ idx_out, int nat_f16,
                unsigned* __restrict__ hist_count, LevelBins lb, int n_chunks, TableSet ts) {
  // hist_count (training, optional): corners per (level, slice) of the points p < n -- the first pass of the binned
  // backward (hash_bin_count_*), done here where the corner indices already exist; LDS histogram per workgroup (the dynamic
  // LDS allocation, >= 4 bytes per slice of the level), one global add per non-empty bin and workgroup
  extern __shared__ unsigned hist_lds[];
  // the operand image is padded to whole 128-point tiles: pad rows repeat the last point so that
  // every stashed value is finite (their gradients are zero downstream)
  // level-major: blockIdx.y = level, consecutive lanes = consecutive points = neighbouring samples of a
  // ray, so a wave's gathers at the coarse and middle levels fall into few cache lines (point-major,
  // 16 lanes of a wave hit 16 different level tables)
  const int64_t rows = out_nat != nullptr ? n_pad : n;
  // ts.n tables of the same level structure evaluated at the same points in one launch (Part 4's three deformation grids):
  // virtual level v = table * n_levels + level; table g reads ts.table_stride entries further and writes its own operand image
  const LevelChunk lc = level_chunk(L.n_levels * ts.n, n_chunks);
  if (lc.lvl >= L.n_levels * ts.n) return;
  const int lvl = lc.lvl % L.n_levels;
  table += (int64_t)(lc.lvl / L.n_levels) * ts.table_stride;
  if (out_nat != nullptr) out_nat += (int64_t)(lc.lvl / L.n_levels) * ts.out_stride;
  const unsigned hist_bins = hist_count != nullptr ? lb.bin0[lvl + 1] - lb.bin0[lvl] : 0u, lvl_offset = L.offset[lvl];
  if (hist_count != nullptr) {
    for (unsigned i = threadIdx.x; i < hist_bins; i += blockDim.x) hist_lds[i] = 0;
    __syncthreads();
  }
  for (int64_t p = lc.chunk * (int64_t)blockDim.x + threadIdx.x; p < rows; p += (int64_t)lc.chunks * blockDim.x) {
    const int64_t g = p * L.n_levels + lvl;
    const int64_t ps = p < n ? p : n - 1;
    const Corner c = corners_of(L, lvl, pts[ps * 3 + 0], pts[ps * 3 + 1], pts[ps * 3 + 2]);
    if (hist_count != nullptr && p < n) {
#pragma unroll
      for (int k = 0; k < 8; ++k) atomicAdd(&hist_lds[(c.idx[k] - lvl_offset) >> kSliceLog2], 1u);
    }
    float f0 = 0.0f, f1 = 0.0f;
    float2 v[8];
    gather_cell(table, c, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) {                           // the sums run over the corners in the reference's order
      f0 += c.w[k] * v[k].x;
      f1 += c.w[k] * v[k].y;
      if (idx_out != nullptr && p < n) idx_out[g * 8 + k] = c.idx[k];
    }
    if (out_f32 != nullptr && p < n) {
      out_f32[p * (2 * L.n_levels) + 2 * lvl + 0] = f0;
      out_f32[p * (2 * L.n_levels) + 2 * lvl + 1] = f1;
    }
    if (out_nat != nullptr) {
      const int f = 2 * lvl, ks = f >> 4, h = (f >> 3) & 1, j = f & 7;
      const int64_t wt = p >> 5;
      const int col = (int)(p & 31);
      const int n_ks = (2 * L.n_levels + 15) / 16;
      __bf16* dst = out_nat + ((wt * n_ks + ks) * 64 + 2 * col + h) * 8 + j;
      if (nat_f16) {                       // fp16 operand image (the Part 4 forward chains contract fp16 operands)
        reinterpret_cast<_Float16*>(dst)[0] = (_Float16)f0;
        reinterpret_cast<_Float16*>(dst)[1] = (_Float16)f1;
      } else {
        dst[0] = (__bf16)f0;
        dst[1] = (__bf16)f1;
      }
    }
  }
  if (hist_count != nullptr) {
    __syncthreads();
    for (unsigned i = threadIdx.x; i < hist_bins; i += blockDim.x)
      if (hist_lds[i] != 0) atomicAdd(hist_count + lb.bin0[lvl] + i, hist_lds[i]);
  }
}

// d_feat [n, 2L] fp32 -> scatter into d_table [E, 2].  Level-major work split (blockIdx.y = level,
// consecutive lanes = consecutive points, i.e. neighbouring samples of a ray): a wave's 64 x 16
// float atomics then fall into ONE level's table, mostly into neighbouring cells, instead of 16
// unrelated regions.  Levels whose whole table fits in LDS (<= kLdsEntries entries: the dense
// 16^3 and 24^3 levels, where thousands of samples hit each cell) are reduced in LDS first and
// flushed with one well-shaped (contiguous) global atomic per entry per workgroup.
// The larger levels use global float atomics, coalesced four lanes per (point, level) (see the
// kernel body).  Built, measured on 200 k points and dropped: entries sliced per XCD by
// HW_REG_XCC_ID so that every 128-byte line stays in one L2 -- no change; owner-computes (a
// workgroup per 16384-entry slice scans the whole batch into LDS) -- slower, the 32-fold
// recomputation of the corner hashes costs more than the atomics it removes; a packed-fp16
// gradient table (one atomic per corner) -- 1.29 ms against 0.72 ms for the coalesced fp32 form.
// Gradient with respect to the ENCODED POSITION (dynamic fields: x_canonical = x + delta_x is encoded, so the
// loss reaches the deformation through d features / d x; reference src/core.py:268-271, 341-344):
//   d f / d x01_a = scale_l * sum_corners table[corner] * (+1 | -1 along a) * prod_{b != a} w_b,
//   d x01 / d x = 1 / (2 bound) inside the box, 0 where HashRepresentation's clamp is active.
// One thread per (point, level), level-major like the forward; the 16 levels of a point meet in d_pts through
// float atomics (three per thread; d_pts is zeroed by the launcher).
template <class TableT>
__global__ void __launch_bounds__(256)
hash_bwd_input_kernel(const float* __restrict__ pts, int64_t n, const TableT* __restrict__ table, HashLevels L,
                      const float* __restrict__ d_feat, float* __restrict__ d_pts, int n_chunks, const float2* __restrict__ grad_lm) {
  const LevelChunk lc = level_chunk(L.n_levels, n_chunks);
  if (lc.lvl >= L.n_levels) return;
  const int lvl = lc.lvl;
  const float two_b = 2.0f * L.bound;
  for (int64_t p = lc.chunk * (int64_t)blockDim.x + threadIdx.x; p < n; p += (int64_t)lc.chunks * blockDim.x) {
    float g0, g1;
    if (grad_lm != nullptr) {                    // level-major [L][n] float2: this launch walks one level per workgroup -- coalesced
      const float2 g = grad_lm[(int64_t)lvl * n + p];
      g0 = g.x; g1 = g.y;
    } else { g0 = d_feat[p * (2 * L.n_levels) + 2 * lvl + 0]; g1 = d_feat[p * (2 * L.n_levels) + 2 * lvl + 1]; }
    if (g0 == 0.0f && g1 == 0.0f) continue;
    const float px = pts[p * 3 + 0], py = pts[p * 3 + 1], pz = pts[p * 3 + 2];
    const Corner c = corners_of(L, lvl, px, py, pz);
    // per-axis weights recovered from the same arithmetic as corners_of
    float frac[3], x01[3] = {add_rn(px, L.bound) / two_b, add_rn(py, L.bound) / two_b, add_rn(pz, L.bound) / two_b};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float cl = fminf(fmaxf(x01[a], 0.0f), 1.0f);
      const float pos = add_rn(mul_rn(cl, L.scale[lvl]), 0.5f);
      frac[a] = sub_rn(pos, floorf(pos));
    }
    float d[3] = {0.0f, 0.0f, 0.0f};
    float2 tv[8];
    gather_cell(table, c, tv);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float2 v = tv[k];
      const float gv = g0 * v.x + g1 * v.y;
      const float wx = (k & 1) ? frac[0] : 1.0f - frac[0], wy = (k & 2) ? frac[1] : 1.0f - frac[1], wz = (k & 4) ? frac[2] : 1.0f - frac[2];
      d[0] += gv * ((k & 1) ? 1.0f : -1.0f) * wy * wz;
      d[1] += gv * ((k & 2) ? 1.0f : -1.0f) * wx * wz;
      d[2] += gv * ((k & 4) ? 1.0f : -1.0f) * wx * wy;
    }
    const float s = L.scale[lvl] / two_b;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const bool inside = x01[a] >= 0.0f && x01[a] <= 1.0f;          // torch.clamp passes the gradient on its closed interval
      if (inside && d[a] != 0.0f) atomicAdd(d_pts + p * 3 + a, d[a] * s);
    }
  }
}

// Option "deterministic": the same gradient with the POINT on the thread, the levels walked in order and summed in registers --
// one store per coordinate instead of one float atomic per level (whose order across levels depends on scheduling).
template <class TableT>
__global__ void __launch_bounds__(256)
hash_bwd_input_ordered_kernel(const float* __restrict__ pts, int64_t n, const TableT* __restrict__ table, HashLevels L,
                              const float* __restrict__ d_feat, float* __restrict__ d_pts, int accumulate) {
  const float two_b = 2.0f * L.bound;
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
    const float px = pts[p * 3 + 0], py = pts[p * 3 + 1], pz = pts[p * 3 + 2];
    const float x01[3] = {add_rn(px, L.bound) / two_b, add_rn(py, L.bound) / two_b, add_rn(pz, L.bound) / two_b};
    float sum[3] = {0.0f, 0.0f, 0.0f};
    for (int lvl = 0; lvl < L.n_levels; ++lvl) {
      const float g0 = d_feat[p * (2 * L.n_levels) + 2 * lvl + 0], g1 = d_feat[p * (2 * L.n_levels) + 2 * lvl + 1];
      if (g0 == 0.0f && g1 == 0.0f) continue;
      const Corner c = corners_of(L, lvl, px, py, pz);
      float frac[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float cl = fminf(fmaxf(x01[a], 0.0f), 1.0f);
        const float pos = add_rn(mul_rn(cl, L.scale[lvl]), 0.5f);
        frac[a] = sub_rn(pos, floorf(pos));
      }
      float d[3] = {0.0f, 0.0f, 0.0f};
      float2 tv[8];
      gather_cell(table, c, tv);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float2 v = tv[k];
        const float gv = g0 * v.x + g1 * v.y;
        const float wx = (k & 1) ? frac[0] : 1.0f - frac[0], wy = (k & 2) ? frac[1] : 1.0f - frac[1], wz = (k & 4) ? frac[2] : 1.0f - frac[2];
        d[0] += gv * ((k & 1) ? 1.0f : -1.0f) * wy * wz;
        d[1] += gv * ((k & 2) ? 1.0f : -1.0f) * wx * wz;
        d[2] += gv * ((k & 4) ? 1.0f : -1.0f) * wx * wy;
      }
      const float s = L.scale[lvl] / two_b;
#pragma unroll
      for (int a = 0; a < 3; ++a) sum[a] += d[a] * s;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const bool inside = x01[a] >= 0.0f && x01[a] <= 1.0f;          // torch.clamp passes the gradient on its closed interval
      const float v = inside ? sum[a] : 0.0f;
      d_pts[p * 3 + a] = accumulate ? d_pts[p * 3 + a] + v : v;
    }
  }
}

constexpr int kLdsEntries = 16384;      // 128 KiB of float2
template <bool in_lds>
__global__ void __launch_bounds__(512)
hash_bwd_kernel(const float* __restrict__ pts, int64_t n, HashLevels L, int lvl0, const float* __restrict__ d_feat,
                float* __restrict__ d_table) {
  extern __shared__ __attribute__((aligned(16))) float lds_acc[];
  const int lvl = lvl0 + blockIdx.y;
  const unsigned size = L.size[lvl], offset = L.offset[lvl];
  if (in_lds) {
    for (unsigned i = threadIdx.x; i < 2 * size; i += blockDim.x) lds_acc[i] = 0.0f;
    __syncthreads();
  }
  if (in_lds) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
      const float g0 = d_feat[p * (2 * L.n_levels) + 2 * lvl + 0], g1 = d_feat[p * (2 * L.n_levels) + 2 * lvl + 1];
      if (g0 == 0.0f && g1 == 0.0f) continue;
      const Corner c = corners_of(L, lvl, pts[p * 3 + 0], pts[p * 3 + 1], pts[p * 3 + 2]);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        atomicAdd(lds_acc + 2 * (c.idx[k] - offset) + 0, c.w[k] * g0);
        atomicAdd(lds_acc + 2 * (c.idx[k] - offset) + 1, c.w[k] * g1);
      }
    }
  } else {
    // Global float atomics retire per 128-byte LINE REQUEST (~21 G/s on MI355X), not per lane: lanes
    // of one instruction that fall into the same line are merged (tools/probe/atomic_rate.hip: 2
    // lanes per line 42 G elements/s, 16 per line 324 G/s).  Four consecutive lanes therefore share
    // one (point, level): lane bit 0 = feature, bit 1 = the x offset of the corner.  Each of the four
    // instructions (one per (dy, dz)) then carries, per point, the two features of the two
    // x-neighbouring corners -- 16 adjacent bytes in 15 of 16 cases, since the x coordinate enters
    // the index with prime 1 -- i.e. 4 line requests per point and level instead of 16.
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < 4 * n; t += (int64_t)gridDim.x * blockDim.x) {
      const int64_t p = t >> 2;
      const int f = (int)(t & 1), dx = (int)((t >> 1) & 1);
      const float g = d_feat[p * (2 * L.n_levels) + 2 * lvl + f];
      if (g == 0.0f) continue;
      const Corner c = corners_of(L, lvl, pts[p * 3 + 0], pts[p * 3 + 1], pts[p * 3 + 2]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // corner index bit 0 is the x offset (corners_of); selects, not a runtime-indexed array
        const unsigned e = dx ? c.idx[2 * q + 1] : c.idx[2 * q];
        const float w = dx ? c.w[2 * q + 1] : c.w[2 * q];
        atomicAdd(d_table + 2 * (size_t)e + f, w * g);
      }
    }
  }
  if (in_lds) {
    __syncthreads();
    for (unsigned i = threadIdx.x; i < 2 * size; i += blockDim.x) {
      const float v = lds_acc[i];
      if (v != 0.0f) atomicAdd(d_table + 2 * (size_t)offset + i, v);
    }
  }
}

// ---- binned scatter (workspace form) --------------------------------------------------------------------
// Float atomics retire per line REQUEST wherever they land (tools/probe/atomic_scope.hip: 20 G/s, the same in a
// 0.5 MB table and under workgroup scope), and the hashed levels scatter every corner into its own line.  The
// workspace form turns the scatter into one partial sort: the contributions of a level are binned by table
// SLICE (kSlice entries = 64 KiB of float2), a workgroup then owns a slice, sums its bin in LDS and adds the
// slice to d_table with plain coalesced read-modify-writes -- no global float atomics on the hot part.
//   1 count   : corners per (level, slice), LDS histogram per workgroup, one global add per bin and workgroup
//   2 plan    : exclusive scan -> bin starts / write cursors; work items = (bin, <= kChunk records); a bin cut
//               into several items (dense coarse levels: thousands of samples per cell) flushes with atomics
//   3 scatter : records {slot in slice, w g0, w g1} packed into 8 B to cursor positions, ranges reserved per workgroup
//   4 reduce  : one workgroup per item sums its records into the LDS slice, flush of the non-zero entries
// The LDS sums are 64-bit FIXED POINT (ds_add_u64): ds_add_f32 retires one lane every three clocks whatever the
// addresses (tools/probe/lds_atomic_rate.hip: 0.33 lanes/clk/CU against 4.7 for ds_add_u64, 7.8 for ds_add_u32).
// The scale is a power of two taken from the largest |d_feat| of the call (found by the count pass), so that a
// term keeps 25 bits below that maximum (the packed record's width) and 2^37 terms cannot overflow;
// integer sums also make the result independent of the order of the records.
constexpr unsigned kChunk = 32768;             // records per work item
constexpr unsigned kMaxSlices = 4096;          // per level (LDS histogram); larger tables use the atomic form
constexpr int kMaxBins = 65536;
constexpr int kFixedBits = 25;                 // magnitude bits of a term (26-bit signed field of the packed record)

// Several tables of one level structure, scattered to from the same points, share one pass (Part 4's three deformation grids):
// VIRTUAL level v = table * n_levels + level; table t's entries start t * table_stride entries further in d_table, its feature
// gradients t * dfeat_stride floats further in d_feat.  One table: n_tables = 1, v = level.
constexpr int kMaxPlanLevels = 48;
struct BinPlan {
  int first, count;                            // virtual levels [first, first + count)
  int n_tables;
  unsigned table_stride;
  long long dfeat_stride;
  unsigned bin0[kMaxPlanLevels + 1];           // first bin of virtual level first + i; bin0[count] = number of bins
};

struct BinHeader {                             // start of the workspace
  unsigned n_items, n_records, amax_bits, n_overflow;  // amax_bits: largest |d_feat| of the call as fp32 bits
  unsigned lost, spec_total, ticket, pad1;     // speculative form: records dropped (overflow list full / plan did not fit); total capacity;
};                                             // ticket: workgroups of the call's last launch that are done
constexpr int kAmaxSlotWord = 32;              // words [32, 32 + kAmaxSlots) of the header area: a producer's running maxima of |d_feat| (common.h),
                                               // folded into amax_bits (and cleared) by the plan pass of the forms that do not count
constexpr int kSpecStatusWord = 16;            // the speculative form's last launch publishes the header here (workspace + 64 bytes) ...
struct BinItem {
  unsigned entry0, begin, end, atomic;         // first table entry of the slice, record range, flush: mode | live entries << 2
  unsigned bin;                                // the item's bin (speculative form: the range ends at the bin's cursor)
  unsigned stage0;                             // kFlushInt: first entry of the slice in the integer staging array
};
// Speculative form (nerf_hash_encode_bwd_ws_store_spec): NO count pass.  The bins' capacities come from the previous call's true
// counts (kept in the workspace) + 1/8 + 64; a record that does not fit its bin goes to an overflow list that a last small launch
// adds with float atomics.  Steady-state training batches fill their bins within a few percent from step to step.
constexpr unsigned kMaxOverflow = 1u << 20;    // overflow records live behind the bins' records; their bins in a parallel array
// flush modes of an item: read-modify-write of the non-zero sums (the slice belongs to this item, d_table holds other
// contributions: the accumulate form), float atomics (the bin was cut into several items), plain STORE of the whole slice
// (the overwrite form: d_table needs no zeroing and is not read back)
constexpr unsigned kFlushRmw = 0, kFlushAtomic = 1, kFlushStore = 2;
// option "deterministic": a cut bin's items add their 64-bit FIXED-POINT sums with integer atomics into a staging array (two words
// per table entry of the coarse dense levels -- the only bins worth cutting: a few MB in the slack of the record area); integer
// addition commutes, so the result does not depend on the order, and a last small launch converts the staged sums to floats.
// Without it a deterministic call could not cut bins at all: the two bins of Instant-NGP's level 0 -- hundreds of thousands of
// records each -- were then summed by one workgroup each (0.79 ms instead of 0.06 for the reduce launch).
constexpr unsigned kFlushInt = 3;
constexpr unsigned kItemFirstOfBin = 1u << 17;
// items of the dense (coarse) levels: consecutive samples of a ray sit in the same cell, so a bin's records come in runs of equal
// slots (16 at the coarsest level) -- read lane-adjacent, a wave's 64 adds pile up on a few LDS addresses (level 0 alone: 51 us of
// the reduce launch's 70).  For these items every lane takes EIGHT CONSECUTIVE records, sums equal neighbours in registers
// (integers: the same sums) and adds once per run; a run then meets at most three lanes of an instruction.
constexpr unsigned kItemRuns = 1u << 16, kItemLiveMask = 0x3fffu;
// one corner contribution, 8 bytes: bits [0,12) slot in the slice, [12,38) and [38,64) the two feature gradients as
// 26-bit signed fixed point at the call's scale (fixed_shift: the largest |d_feat| of the call keeps 25 bits, i.e. a
// resolution of 3e-8 of it -- finer than one fp32 ulp of that largest term; tinycudann accumulates these in fp16)
typedef unsigned long long BinRecord;
__device__ __forceinline__ BinRecord pack_record(unsigned slot, float g0, float g1, float scale) {
  const long long lim = (1ll << kFixedBits) - 1;
  long long a = __float2ll_rn(g0 * scale), b = __float2ll_rn(g1 * scale);
  a = a > lim ? lim : (a < -lim ? -lim : a);
  b = b > lim ? lim : (b < -lim ? -lim : b);
  return (BinRecord)slot | (((BinRecord)a & 0x3ffffffull) << 12) | ((BinRecord)b << 38);
}
__device__ __forceinline__ unsigned record_slot(BinRecord r) { return (unsigned)r & (kSlice - 1); }
__device__ __forceinline__ long long record_g0(BinRecord r) { return (long long)(r << 26) >> 38; }   // sign-extended bits [12,38)
__device__ __forceinline__ long long record_g1(BinRecord r) { return (long long)r >> 38; }

// records the workspace holds: every corner of every point and level, + a quarter and 64 per bin for the speculative form's margins
static size_t bin_record_capacity(int64_t n, int n_levels) { const size_t r = (size_t)n * 8 * (size_t)n_levels; return r + r / 4 + 64 * (size_t)kMaxBins; }
struct BinWorkspace {
  BinHeader* header;
  unsigned* count;                             // [bins]  (speculative form: the bins' capacities)
  unsigned* cursor;                            // [bins]
  unsigned* est;                               // [bins] true record counts of the last call (persistent between calls)
  unsigned* start;                             // [bins] first record of the bin (speculative form)
  BinItem* items;                              // [max_items]
  float2* grad_lm;                             // [levels][n] level-major copy of d_feat (the count pass writes it)
  BinRecord* records;                          // bin_record_capacity(n, L) records
  unsigned* overflow_bin;                      // [kMaxOverflow] bin of overflow record o = records[bin_record_capacity + o]
};

static size_t bin_max_items(int64_t n, int n_levels) { return (size_t)(bin_record_capacity(n, n_levels) / kChunk) + kMaxBins; }

static BinWorkspace carve(void* base, int64_t n, int n_levels) {
  char* p = static_cast<char*>(base);
  BinWorkspace w;
  w.header = reinterpret_cast<BinHeader*>(p);                 p += 256;
  w.count = reinterpret_cast<unsigned*>(p);                   p += sizeof(unsigned) * kMaxBins;
  w.cursor = reinterpret_cast<unsigned*>(p);                  p += sizeof(unsigned) * kMaxBins;
  w.est = reinterpret_cast<unsigned*>(p);                     p += sizeof(unsigned) * kMaxBins;          // fixed offsets: survive a change of n
  w.start = reinterpret_cast<unsigned*>(p);                   p += sizeof(unsigned) * kMaxBins;
  w.overflow_bin = reinterpret_cast<unsigned*>(p);            p += sizeof(unsigned) * kMaxOverflow;
  w.items = reinterpret_cast<BinItem*>(p);                    p += (sizeof(BinItem) * bin_max_items(n, n_levels) + 255) / 256 * 256;
  w.grad_lm = reinterpret_cast<float2*>(p);                   p += (sizeof(float2) * (size_t)n * (size_t)n_levels + 255) / 256 * 256;
  w.records = reinterpret_cast<BinRecord*>(p);
  return w;
}

static size_t bin_workspace_bytes(int64_t n, int n_levels) {
  return 256 + 4 * sizeof(unsigned) * kMaxBins + sizeof(unsigned) * kMaxOverflow +
         (sizeof(BinItem) * bin_max_items(n, n_levels) + 255) / 256 * 256 +
         (sizeof(float2) * (size_t)n * (size_t)n_levels + 255) / 256 * 256 + sizeof(BinRecord) * (bin_record_capacity(n, n_levels) + kMaxOverflow);
}

__device__ __forceinline__ bool point_gradient(const float* __restrict__ d_feat, int64_t p, int n_levels, int lvl, float& g0, float& g1) {
  const float2 g = *reinterpret_cast<const float2*>(d_feat + p * (2 * n_levels) + 2 * lvl);
  g0 = g.x;
  g1 = g.y;
  return g0 != 0.0f || g1 != 0.0f;
}

// LDS counter updates of a wave whose lanes mostly hit the SAME bin (the two coarsest levels: 2 and 3 slices; measured slower from
// 12 slices per level on): one add per distinct bin and wave instead of up to 64 queued on one address.  Called by the active lanes of a wave.
__device__ __forceinline__ void wave_count_add(unsigned* counters, unsigned bin) {
  unsigned long long todo = __ballot(1);
  const int lane = threadIdx.x & 63;
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const unsigned b = __shfl(bin, leader);
    const unsigned long long same = __ballot(bin == b);
    if (lane == leader) atomicAdd(&counters[b], (unsigned)__popcll(same));
    todo &= ~same;
  }
}
// ... returning each lane's position: the counter's value before the wave's add + the lane's rank among the lanes of its bin
__device__ __forceinline__ unsigned wave_count_take(unsigned* counters, unsigned bin) {
  unsigned long long todo = __ballot(1);
  const int lane = threadIdx.x & 63;
  unsigned slot = 0;
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const unsigned b = __shfl(bin, leader);
    const unsigned long long same = __ballot(bin == b);
    unsigned first = 0;
    if (lane == leader) first = atomicAdd(&counters[b], (unsigned)__popcll(same));
    first = __shfl(first, leader);
    if (bin == b) slot = first + (unsigned)__popcll(same & ((1ull << lane) - 1ull));
    todo &= ~same;
  }
  return slot;
}

__global__ void __launch_bounds__(512)
hash_bin_count_kernel(const float* __restrict__ pts, int64_t n, HashLevels L, BinPlan plan, const float* __restrict__ d_feat,
                      unsigned* __restrict__ count, BinHeader* __restrict__ header) {
  __shared__ unsigned hist[kMaxSlices];
  __shared__ unsigned wg_amax;
  const int lvl = plan.first + blockIdx.y;
  const unsigned bins = plan.bin0[blockIdx.y + 1] - plan.bin0[blockIdx.y], offset = L.offset[lvl];
  for (unsigned i = threadIdx.x; i < bins; i += blockDim.x) hist[i] = 0;
  if (threadIdx.x == 0) wg_amax = 0;
  __syncthreads();
  float amax = 0.0f;
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
    float g0, g1;
    if (!point_gradient(d_feat, p, L.n_levels, lvl, g0, g1)) continue;
    amax = fmaxf(amax, fmaxf(fabsf(g0), fabsf(g1)));
    const Corner c = corners_of(L, lvl, pts[p * 3 + 0], pts[p * 3 + 1], pts[p * 3 + 2]);
#pragma unroll
    for (int k = 0; k < 8; ++k) atomicAdd(&hist[(c.idx[k] - offset) >> kSliceLog2], 1u);
  }
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
  if ((threadIdx.x & 63) == 0 && amax > 0.0f && amax <= 3.0e38f) atomicMax(&wg_amax, __float_as_uint(amax));
  __syncthreads();
  // one contended global atomic per workgroup at most, none once a larger value is visible
  if (threadIdx.x == 0 && wg_amax > __hip_atomic_load(&header->amax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(&header->amax_bits, wg_amax);
  for (unsigned i = threadIdx.x; i < bins; i += blockDim.x)
    if (hist[i] != 0) atomicAdd(count + plan.bin0[blockIdx.y] + i, hist[i]);
}

// The same count with the POINT on the lane: a lane reads its point and its whole 8 L-byte row of d_feat once
// (the level-major form reads 8 bytes of every 128-byte row per level: a quarter of each sector fetched is used),
// walks the levels, and leaves a level-major copy of the gradients for the scatter pass.  Needs the histograms of
// all levels in LDS at once: used when the call has at most kPmBins bins (2048 for L16 / T2^19).
constexpr unsigned kPmBins = 8192;
// levels per workgroup row of the point-major count pass: all of them for a large batch (198 k points: 46 us, in rows of four
// 63 us -- four times the point reads and flushes), four for a small one (54 k points x 36 levels: 37 -> 18 us per launch)
inline int pm_levels_per_row(int64_t n, int n_levels) { return n >= 131072 ? n_levels : 4; }
__global__ void __launch_bounds__(256)
hash_bin_count_pm_kernel(const float* __restrict__ pts, int64_t n, HashLevels L, BinPlan plan, const float* __restrict__ d_feat,
                         unsigned* __restrict__ count, BinHeader* __restrict__ header, float2* __restrict__ grad_lm, int levels_per_row) {
  __shared__ unsigned hist[kPmBins];
  __shared__ unsigned wg_amax;
  // blockIdx.y: a chunk of levels_per_row levels -- a lane walking all 16 (36) levels alone is a long dependent chain, and a small
  // batch (54 k points) leaves less than one wave per SIMD to hide it
  const int li0 = blockIdx.y * levels_per_row, li1 = min(li0 + levels_per_row, plan.count);
  const unsigned bin_lo = plan.bin0[li0], bin_hi = plan.bin0[li1];
  for (unsigned i = bin_lo + threadIdx.x; i < bin_hi; i += blockDim.x) hist[i] = 0;
  if (threadIdx.x == 0) wg_amax = 0;
  __syncthreads();
  float amax = 0.0f;
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
    const float px = pts[p * 3 + 0], py = pts[p * 3 + 1], pz = pts[p * 3 + 2];
    for (int li = li0; li < li1; ++li) {
      const int vl = plan.first + li, tbl = vl / L.n_levels;
      const float2 g = *reinterpret_cast<const float2*>(d_feat + tbl * plan.dfeat_stride + p * (2 * L.n_levels) + 2 * (vl - tbl * L.n_levels));
      grad_lm[(int64_t)li * n + p] = g;
      if (g.x == 0.0f && g.y == 0.0f) continue;
      amax = fmaxf(amax, fmaxf(fabsf(g.x), fabsf(g.y)));
      const int lvl = vl - tbl * L.n_levels;
      const Corner c = corners_of(L, lvl, px, py, pz);
      const unsigned offset = L.offset[lvl], b0 = plan.bin0[li];
      if (plan.bin0[li + 1] - b0 <= 4u) {        // uniform per iteration: a level of at most four slices (the loop runs once per distinct bin)
#pragma unroll
        for (int k = 0; k < 8; ++k) wave_count_add(hist, b0 + ((c.idx[k] - offset) >> kSliceLog2));
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&hist[b0 + ((c.idx[k] - offset) >> kSliceLog2)], 1u);
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
  if ((threadIdx.x & 63) == 0 && amax > 0.0f && amax <= 3.0e38f) atomicMax(&wg_amax, __float_as_uint(amax));
  __syncthreads();
  if (threadIdx.x == 0 && wg_amax > __hip_atomic_load(&header->amax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(&header->amax_bits, wg_amax);
  for (unsigned i = bin_lo + threadIdx.x; i < bin_hi; i += blockDim.x)
    if (hist[i] != 0) atomicAdd(count + i, hist[i]);
}

// 2^s with |v| 2^s < 2^kFixedBits for every |v| <= amax; s clamped so that both 2^s and 2^-s are normal fp32
__device__ __forceinline__ int fixed_shift(unsigned amax_bits) {
  const int e = (int)(amax_bits >> 23) - 126;                     // amax < 2^e
  int s = kFixedBits - e;
  return s > 100 ? 100 : (s < -80 ? -80 : s);
}

// one workgroup of 1024: bins in rounds of 1024 with a carried total
__global__ void __launch_bounds__(1024)
hash_bin_plan_kernel(HashLevels L, BinPlan plan, unsigned* __restrict__ count, unsigned* __restrict__ cursor,
                     BinItem* __restrict__ items, BinHeader* __restrict__ header, int overwrite, unsigned chunk,
                     unsigned* __restrict__ est, unsigned* __restrict__ start, unsigned spec_capacity, int fold_amax,
                     unsigned dense_entries, int det_int) {
  // det_int (option "deterministic" with a staging array): only the bins of DENSE levels are cut (chunk), the others never
  // est != null, spec_capacity == 0 (counted forms): the bins' true counts are also left in est for a later speculative call.
  // spec_capacity > 0 (speculative form): NO counts exist -- a bin's capacity is est + est / 8 + 64 (written to count[], which the
  // later passes read as the bin's size), its records start at start[bin]; the reduce pass reads the true fill from cursor[]
  // chunk: records per work item -- kChunk, or 0xffffffff (option "deterministic"): a bin is never cut, so no slice is
  // flushed with float atomics; the coarse dense levels' bins then serialise on one workgroup each
  __shared__ unsigned scan_r[1024], scan_i[1024], wave_r[16], wave_i[16];
  __shared__ unsigned carry_r, carry_i;
  __shared__ unsigned s_bin0[kMaxPlanLevels + 1], s_size[kMaxLevels], s_offset[kMaxLevels], s_dense[kMaxLevels];
  const unsigned n_bins = plan.bin0[plan.count];
  if (threadIdx.x <= (unsigned)plan.count) s_bin0[threadIdx.x] = plan.bin0[threadIdx.x];
  if (threadIdx.x < (unsigned)L.n_levels) { s_size[threadIdx.x] = L.size[threadIdx.x]; s_offset[threadIdx.x] = L.offset[threadIdx.x]; s_dense[threadIdx.x] = L.dense[threadIdx.x]; }
  if (threadIdx.x == 0) carry_r = carry_i = 0;
  if (fold_amax && threadIdx.x < 64) {         // the producer's kAmaxSlots running maxima -> the call's amax_bits; slots cleared for the next call
    unsigned* slots = reinterpret_cast<unsigned*>(header) + kAmaxSlotWord;
    unsigned v = threadIdx.x < kAmaxSlots ? slots[threadIdx.x] : 0u;
    if (threadIdx.x < kAmaxSlots) slots[threadIdx.x] = 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, off));
    if (threadIdx.x == 0) header->amax_bits = max(header->amax_bits, v);
  }
  __syncthreads();
  // the first four rounds' counts are loaded before the first scan (one global-load latency instead of one per round)
  const unsigned* src = spec_capacity ? est : count;
  unsigned pre[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { const unsigned b = r * 1024 + threadIdx.x; pre[r] = b < n_bins ? src[b] : 0u; }
  for (unsigned base = 0; base < n_bins; base += 1024) {
    const unsigned b = base + threadIdx.x, round = base >> 10;
    unsigned c = round == 0 ? pre[0] : round == 1 ? pre[1] : round == 2 ? pre[2] : round == 3 ? pre[3] : (b < n_bins ? src[b] : 0u);
    if (b < n_bins) {
      if (spec_capacity) {
        c = c + (c >> 3) + 64u;
        count[b] = c;
      } else if (est != nullptr) est[b] = c;
    }
    unsigned chunk_b = chunk;
    if (det_int && b < n_bins) {
      int li = 0, hi_l = plan.count - 1;
      while (li < hi_l) { const int mid = (li + hi_l + 1) >> 1; if (s_bin0[mid] <= b) li = mid; else hi_l = mid - 1; }
      if (!s_dense[(plan.first + li) % L.n_levels]) chunk_b = 0xffffffffu;
    }
    // overwrite form: every bin gets an item (an empty bin's item stores a slice of zeros)
    const unsigned it = c == 0 ? ((b < n_bins && overwrite) ? 1u : 0u) : (unsigned)(((unsigned long long)c + chunk_b - 1) / chunk_b);
    // inclusive scan of both columns: within the wave by shuffles, the sixteen wave totals through LDS (two barriers per round
    // instead of twenty: the launch sits alone on the pass's critical path)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned sr = c, si = it;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned ur = __shfl_up(sr, d), ui = __shfl_up(si, d);
      if (lane >= d) { sr += ur; si += ui; }
    }
    if (lane == 63) { wave_r[wave] = sr; wave_i[wave] = si; }
    __syncthreads();
    unsigned before_r = 0, before_i = 0;
    for (int w = 0; w < wave; ++w) { before_r += wave_r[w]; before_i += wave_i[w]; }
    scan_r[threadIdx.x] = sr + before_r;
    scan_i[threadIdx.x] = si + before_i;
    __syncthreads();
    const unsigned r0 = carry_r + scan_r[threadIdx.x] - c;
    if (b < n_bins) {
      cursor[b] = r0;
      if (start != nullptr) start[b] = r0;
    }
    // the round's work items, written by ALL threads (item k belongs to the bin whose inclusive item scan first exceeds k): a bin of a
    // coarse dense level is cut into a few hundred items, which ONE thread used to write one after the other -- most of this
    // launch's time, and the launch sits alone on the pass's critical path
    const unsigned round_items = scan_i[1023];
    for (unsigned k = threadIdx.x; k < round_items; k += 1024) {
      unsigned lo = 0, hi = 1023;
      while (lo < hi) {
        const unsigned mid = (lo + hi) >> 1;
        if (scan_i[mid] > k) hi = mid; else lo = mid + 1;
      }
      const unsigned t = lo, bb = base + t;
      const unsigned it_t = scan_i[t] - (t ? scan_i[t - 1] : 0u), c_t = scan_r[t] - (t ? scan_r[t - 1] : 0u);
      const unsigned j = k - (scan_i[t] - it_t), rb = carry_r + scan_r[t] - c_t;
      int li = 0, hi_l = plan.count - 1;          // the bin's virtual level: last li with s_bin0[li] <= bb (binary search in LDS: a walk
      while (li < hi_l) {                          // over the kernel argument's array costs a dependent scalar load per step)
        const int mid = (li + hi_l + 1) >> 1;
        if (s_bin0[mid] <= bb) li = mid; else hi_l = mid - 1;
      }
      const unsigned first = (bb - s_bin0[li]) << kSliceLog2;
      const int tbl = (plan.first + li) / L.n_levels, lvl = plan.first + li - tbl * L.n_levels;
      const unsigned live = min(kSlice, s_size[lvl] - first);      // the level's last slice may be partial
      BinItem item;
      item.entry0 = tbl * plan.table_stride + s_offset[lvl] + first;
      item.begin = rb + j * chunk;                                 // (j > 0 only in bins that are cut by `chunk`)
      item.end = it_t > 1 ? rb + (unsigned)min((unsigned long long)c_t, (unsigned long long)(j + 1) * chunk) : rb + c_t;
      item.atomic = (it_t > 1 ? (det_int ? kFlushInt : kFlushAtomic) : (overwrite ? kFlushStore : kFlushRmw)) | (live << 2) |
                    (s_dense[lvl] ? kItemRuns : 0u) | (j == 0 ? kItemFirstOfBin : 0u);
      item.bin = bb;
      item.stage0 = tbl * dense_entries + s_offset[lvl] + first;
      items[carry_i + k] = item;
    }
    __syncthreads();
    if (threadIdx.x == 1023) {
      carry_r += scan_r[1023];
      carry_i += scan_i[1023];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    header->n_items = carry_i;
    header->n_records = carry_r;
    if (spec_capacity) {
      header->spec_total = carry_r;
      header->n_overflow = 0;
      header->lost = 0;
      if (carry_r > spec_capacity) {           // the estimates do not fit the workspace: nothing may be written
        header->n_items = 0;
        header->lost = 2;
      }
    }
  }
}

// STAGED (levels of at most kStagedBins slices, i.e. tables up to 2^20 entries): the round's 4096 records are
// grouped by bin in LDS first and leave as contiguous runs -- one bin's records of the round are adjacent in
// the workspace, so a wave's store covers a few whole lines instead of 64 separate 12-byte pieces.
constexpr unsigned kStagedBins = 256;
template <bool STAGED>
__global__ void __launch_bounds__(512, 8)
hash_bin_scatter_kernel(const float* __restrict__ pts, int64_t n, HashLevels L, BinPlan plan, const float* __restrict__ d_feat,
                        unsigned* __restrict__ cursor, BinRecord* __restrict__ records, const BinHeader* __restrict__ header,
                        const float2* __restrict__ grad_lm, const unsigned* __restrict__ count, float* __restrict__ zero_table,
                        int all_live, unsigned chunk, const unsigned* __restrict__ spec_start, unsigned* __restrict__ overflow_bin,
                        BinHeader* __restrict__ header_rw, unsigned overflow_base, unsigned long long* __restrict__ istage,
                        unsigned dense_entries) {
  // istage != null (option "deterministic"): the cut bins (dense levels only) are summed in the integer staging array: their slices
  // of THAT are zeroed here, d_table is written by the convert launch
  // spec_start != null (speculative form, staged levels only): bin b holds count[b] records from spec_start[b] on; a record past
  // that goes to the overflow list (records[overflow_base + o], its bin in overflow_bin[o])
  constexpr unsigned kBins = STAGED ? kStagedBins : kMaxSlices;
  __shared__ unsigned cnt[kBins], base[kBins];
  __shared__ unsigned start[STAGED ? kBins : 1], wave_sum[8], total;
  __shared__ BinRecord stage[STAGED ? 4096 : 1];
  __shared__ unsigned char bin_of[STAGED ? 4096 : 1];       // bin of every staged record (<= kStagedBins = 256 bins): its place in the workspace is
                                                            // base[bin] + (position in the stage - start[bin]); 4 KiB instead of 16 KiB of places:
                                                            // 39 KiB of LDS per workgroup, four workgroups per CU instead of three
  const int tbl = (plan.first + (int)blockIdx.y) / L.n_levels, lvl = plan.first + (int)blockIdx.y - tbl * L.n_levels;
  const unsigned bins = plan.bin0[blockIdx.y + 1] - plan.bin0[blockIdx.y], offset = L.offset[lvl];
  if (STAGED != (bins <= kStagedBins)) return;                 // the other instantiation owns this level
  d_feat += tbl * plan.dfeat_stride;
  if (istage != nullptr) {
    if (L.dense[lvl])
      for (unsigned b = blockIdx.x; b < bins; b += gridDim.x) {
        if (count[plan.bin0[blockIdx.y] + b] <= chunk) continue;
        const unsigned first = b << kSliceLog2, live = min(kSlice, L.size[lvl] - first);
        unsigned long long* dst = istage + 2 * ((size_t)tbl * dense_entries + offset + first);
        for (unsigned i = threadIdx.x; i < 2 * live; i += blockDim.x) dst[i] = 0ull;
      }
  } else if (zero_table != nullptr) {
    // overwrite form: a bin that was cut into several items is flushed with atomics by the reduce pass (a later launch),
    // so its slice is zeroed here -- the coarse dense levels in steady state, 1.8 MB of a 52 MB table
    for (unsigned b = blockIdx.x; b < bins; b += gridDim.x) {
      if (count[plan.bin0[blockIdx.y] + b] <= chunk) continue;
      const unsigned first = b << kSliceLog2, live = min(kSlice, L.size[lvl] - first);
      float2* dst = reinterpret_cast<float2*>(zero_table) + (size_t)tbl * plan.table_stride + offset + first;
      for (unsigned i = threadIdx.x; i < live; i += blockDim.x) dst[i] = make_float2(0.0f, 0.0f);
    }
  }
  const float scale = __uint_as_float((unsigned)(127 + fixed_shift(header->amax_bits)) << 23);
  for (unsigned i = threadIdx.x; i < bins; i += blockDim.x) cnt[i] = 0;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p0 = blockIdx.x * (int64_t)blockDim.x; p0 < n; p0 += stride) {     // uniform trip count: barriers inside
    const int64_t p = p0 + threadIdx.x;
    float g0 = 0.0f, g1 = 0.0f;
    bool live = false;
    if (p < n) {
      if (grad_lm != nullptr) {                 // coalesced: the count pass left the gradients level-major
        const float2 g = grad_lm[(int64_t)blockIdx.y * n + p];
        g0 = g.x;
        g1 = g.y;
        live = all_live || g0 != 0.0f || g1 != 0.0f;     // counted by the forward: every point owns its eight records
      } else live = point_gradient(d_feat, p, L.n_levels, lvl, g0, g1);
    }
    Corner c;
    unsigned slot[8];
    if (live) {
      c = corners_of(L, lvl, pts[p * 3 + 0], pts[p * 3 + 1], pts[p * 3 + 2]);
      if (bins <= 4u) {                          // uniform: a level of at most four slices (the loop runs once per distinct bin of the wave)
#pragma unroll
        for (int k = 0; k < 8; ++k) slot[k] = wave_count_take(cnt, (c.idx[k] - offset) >> kSliceLog2);
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) slot[k] = atomicAdd(&cnt[(c.idx[k] - offset) >> kSliceLog2], 1u);
      }
    }
    __syncthreads();
    if constexpr (STAGED) {
      // reserve the bins' runs in the workspace; exclusive scan of the counts = the runs' places in the stage
      const unsigned t = threadIdx.x;
      unsigned v = 0, fit = 0;
      if (t < kStagedBins) {
        v = t < bins ? cnt[t] : 0u;
        fit = v;
        if (v != 0) {
          const unsigned b0 = atomicAdd(cursor + plan.bin0[blockIdx.y] + t, v);
          base[t] = b0;
          if (spec_start != nullptr) {             // speculative form: how many of this round's records still fit the bin
            const unsigned lim = spec_start[plan.bin0[blockIdx.y] + t] + count[plan.bin0[blockIdx.y] + t];
            fit = b0 >= lim ? 0u : min(v, lim - b0);
          }
        }
        unsigned incl = v;
        for (int o = 1; o < 64; o <<= 1) {
          const unsigned up = __shfl_up(incl, o);
          if ((int)(t & 63) >= o) incl += up;
        }
        if ((t & 63) == 63) wave_sum[t >> 6] = incl;
        start[t] = incl - v;                                    // within the wave; the waves before are added below
      }
      __syncthreads();
      if (t < kStagedBins) {
        unsigned before = 0;
        for (unsigned w = 0; w < (t >> 6); ++w) before += wave_sum[w];
        start[t] += before;
        if (t < bins) cnt[t] = fit;               // until the copy below: records of this round that fit; zeroed after it
        if (t == kStagedBins - 1) total = start[t] + v;
      }
      __syncthreads();
      if (live) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const unsigned local = c.idx[k] - offset, b = local >> kSliceLog2, pos = start[b] + slot[k];
          stage[pos] = pack_record(local & (kSlice - 1), c.w[k] * g0, c.w[k] * g1, scale);
          bin_of[pos] = (unsigned char)b;
        }
      }
      __syncthreads();
      const unsigned staged = total;
      for (unsigned q = threadIdx.x; q < staged; q += blockDim.x) {
        const unsigned b = bin_of[q], idx = q - start[b];
        unsigned dst = base[b] + idx;
        if (idx >= cnt[b]) {                     // speculative form: the bin is full
          const unsigned o = atomicAdd(&header_rw->n_overflow, 1u);
          if (o >= kMaxOverflow) { header_rw->lost = 1; continue; }
          overflow_bin[o] = plan.bin0[blockIdx.y] + b;
          dst = overflow_base + o;
        }
        records[dst] = stage[q];
      }
      __syncthreads();
      for (unsigned i = threadIdx.x; i < bins; i += blockDim.x) cnt[i] = 0;
      __syncthreads();                          // stage, start, base and cnt are rewritten by the next round
    } else {
      for (unsigned i = threadIdx.x; i < bins; i += blockDim.x) {
        const unsigned v = cnt[i];
        if (v != 0) base[i] = atomicAdd(cursor + plan.bin0[blockIdx.y] + i, v);
        cnt[i] = 0;
      }
      __syncthreads();
      if (live) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const unsigned local = c.idx[k] - offset;
          records[base[local >> kSliceLog2] + slot[k]] = pack_record(local & (kSlice - 1), c.w[k] * g0, c.w[k] * g1, scale);
        }
      }
      __syncthreads();                          // base[] is rewritten by the next round
    }
  }
}

__global__ void __launch_bounds__(512)
hash_bin_reduce_kernel(const BinHeader* __restrict__ header, const BinItem* __restrict__ items, const BinRecord* __restrict__ records,
                       float* __restrict__ d_table, unsigned table_entries, const unsigned* __restrict__ spec_cursor,
                       const unsigned* __restrict__ spec_start, unsigned* __restrict__ est, unsigned long long* __restrict__ istage) {
  // spec_cursor != null (speculative form): an item's range was planned from the bin's CAPACITY; what was written ends at the
  // bin's cursor.  The bin's true count (cursor - start, overflow included) is left in est for the next call.
  __shared__ unsigned long long acc[2 * kSlice];            // 64 KiB of 64-bit fixed-point sums
  const int shift = fixed_shift(header->amax_bits);
  const float inv_scale = __uint_as_float((unsigned)(127 - shift) << 23);
  for (unsigned item_id = blockIdx.x; item_id < header->n_items; item_id += gridDim.x) {
    BinItem item = items[item_id];
    if (spec_cursor != nullptr) {
      const unsigned fill = spec_cursor[item.bin];
      if (threadIdx.x == 0 && item.begin == spec_start[item.bin]) est[item.bin] = fill - item.begin;
      item.end = min(item.end, fill);
      item.begin = min(item.begin, item.end);
    }
    for (unsigned i = threadIdx.x; i < 2 * kSlice; i += blockDim.x) acc[i] = 0ull;
    __syncthreads();
    // kUnroll independent record loads in flight per lane: the loop is latency-bound otherwise (49 records per lane)
    constexpr int kUnroll = 8;
    if (item.atomic & kItemRuns) {
      for (unsigned r0 = item.begin + kUnroll * threadIdx.x; r0 < item.end; r0 += kUnroll * blockDim.x) {
        BinRecord rec[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) rec[u] = records[r0 + u < item.end ? r0 + u : item.end - 1];
        unsigned cur = record_slot(rec[0]);
        long long s0 = record_g0(rec[0]), s1 = record_g1(rec[0]);
#pragma unroll
        for (int u = 1; u < kUnroll; ++u) {
          if (r0 + u >= item.end) break;
          const unsigned slot = record_slot(rec[u]);
          if (slot != cur) {
            if (s0 != 0) atomicAdd(&acc[2 * cur + 0], (unsigned long long)s0);
            if (s1 != 0) atomicAdd(&acc[2 * cur + 1], (unsigned long long)s1);
            cur = slot;
            s0 = s1 = 0;
          }
          s0 += record_g0(rec[u]);
          s1 += record_g1(rec[u]);
        }
        if (s0 != 0) atomicAdd(&acc[2 * cur + 0], (unsigned long long)s0);
        if (s1 != 0) atomicAdd(&acc[2 * cur + 1], (unsigned long long)s1);
      }
    } else
    for (unsigned r0 = item.begin + threadIdx.x; r0 < item.end; r0 += kUnroll * blockDim.x) {
      BinRecord rec[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const unsigned r = r0 + u * blockDim.x;
        rec[u] = records[r < item.end ? r : item.end - 1];
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        if (r0 + u * blockDim.x < item.end) {
          atomicAdd(&acc[2 * record_slot(rec[u]) + 0], (unsigned long long)record_g0(rec[u]));
          atomicAdd(&acc[2 * record_slot(rec[u]) + 1], (unsigned long long)record_g1(rec[u]));
        }
      }
    }
    __syncthreads();
    float2* dst = reinterpret_cast<float2*>(d_table) + item.entry0;
    const unsigned live = min((item.atomic >> 2) & kItemLiveMask, table_entries - item.entry0), mode = item.atomic & 3u;
    if (mode == kFlushStore) {
      for (unsigned i = threadIdx.x; i < live; i += blockDim.x)
        dst[i] = make_float2(__ll2float_rn((long long)acc[2 * i]) * inv_scale, __ll2float_rn((long long)acc[2 * i + 1]) * inv_scale);
    } else if (mode == kFlushAtomic) {
      for (unsigned i = threadIdx.x; i < live; i += blockDim.x) {
        const long long a0 = (long long)acc[2 * i], a1 = (long long)acc[2 * i + 1];
        if (a0 != 0) atomicAdd(&dst[i].x, __ll2float_rn(a0) * inv_scale);
        if (a1 != 0) atomicAdd(&dst[i].y, __ll2float_rn(a1) * inv_scale);
      }
    } else if (mode == kFlushInt) {             // integer sums: the same total in any order (option "deterministic")
      unsigned long long* sdst = istage + 2 * (size_t)item.stage0;
      for (unsigned i = threadIdx.x; i < 2 * live; i += blockDim.x)
        if (acc[i] != 0ull) atomicAdd(&sdst[i], acc[i]);
    } else {
      // the slice belongs to this workgroup for the whole launch: plain read-modify-write, all loads first
      static_assert(kSlice % 512 == 0, "flush");
      constexpr int kPer = kSlice / 512;
      float2 t[kPer];
#pragma unroll
      for (int u = 0; u < kPer; ++u) {
        const unsigned i = threadIdx.x + u * 512;
        t[u] = i < live ? dst[i] : make_float2(0.0f, 0.0f);
      }
#pragma unroll
      for (int u = 0; u < kPer; ++u) {
        const unsigned i = threadIdx.x + u * 512;
        const long long a0 = (long long)acc[2 * i], a1 = (long long)acc[2 * i + 1];
        if (i < live && (a0 != 0 || a1 != 0))
          dst[i] = make_float2(t[u].x + __ll2float_rn(a0) * inv_scale, t[u].y + __ll2float_rn(a1) * inv_scale);
      }
    }
    __syncthreads();
  }
}

// option "deterministic": the cut bins' staged integer sums -> d_table (stored, or added to what is there: the accumulating form);
// one workgroup per first item of a cut bin
__global__ void __launch_bounds__(256)
hash_bin_stage_convert_kernel(const BinHeader* __restrict__ header, const BinItem* __restrict__ items, const unsigned long long* __restrict__ istage,
                              float* __restrict__ d_table, unsigned table_entries, int accumulate) {
  const float inv_scale = __uint_as_float((unsigned)(127 - fixed_shift(header->amax_bits)) << 23);
  for (unsigned item_id = blockIdx.x; item_id < header->n_items; item_id += gridDim.x) {
    const BinItem item = items[item_id];
    if ((item.atomic & 3u) != kFlushInt || !(item.atomic & kItemFirstOfBin)) continue;
    const unsigned live = min((item.atomic >> 2) & kItemLiveMask, table_entries - item.entry0);
    float2* dst = reinterpret_cast<float2*>(d_table) + item.entry0;
    const unsigned long long* src = istage + 2 * (size_t)item.stage0;
    for (unsigned i = threadIdx.x; i < live; i += blockDim.x) {
      const float2 v = make_float2(__ll2float_rn((long long)src[2 * i]) * inv_scale, __ll2float_rn((long long)src[2 * i + 1]) * inv_scale);
      dst[i] = accumulate ? make_float2(dst[i].x + v.x, dst[i].y + v.y) : v;
    }
  }
}

// speculative form: the records that did not fit their bins, added with float atomics (a handful per step in steady state).  The
// call's LAST launch: the workgroup that finishes last publishes the header's eight status words (workspace + 64 bytes, and the
// caller's host-mapped status block) and clears the header -- the next speculative call starts from a clean header with no fill
// launch, and the host reads the status without a copy launch.
__global__ void __launch_bounds__(256)
hash_bin_overflow_kernel(BinHeader* __restrict__ header, const BinRecord* __restrict__ overflow, const unsigned* __restrict__ overflow_bin,
                         HashLevels L, BinPlan plan, float* __restrict__ d_table, unsigned* __restrict__ status_host) {
  const unsigned n = min(header->n_overflow, kMaxOverflow);
  const float inv_scale = __uint_as_float((unsigned)(127 - fixed_shift(header->amax_bits)) << 23);
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const BinRecord rec = overflow[i];
    const unsigned bin = overflow_bin[i];
    int li = 0;
    while (li + 1 < plan.count && plan.bin0[li + 1] <= bin) ++li;
    const int tbl = (plan.first + li) / L.n_levels, lvl = plan.first + li - tbl * L.n_levels;
    const unsigned entry = tbl * plan.table_stride + L.offset[lvl] + ((bin - plan.bin0[li]) << kSliceLog2) + record_slot(rec);
    const long long a0 = record_g0(rec), a1 = record_g1(rec);
    if (a0 != 0) atomicAdd(d_table + 2 * (size_t)entry + 0, __ll2float_rn(a0) * inv_scale);
    if (a1 != 0) atomicAdd(d_table + 2 * (size_t)entry + 1, __ll2float_rn(a1) * inv_scale);
  }
  __syncthreads();                                               // every thread of this workgroup has read the header
  __shared__ unsigned last;
  if (threadIdx.x == 0) last = atomicAdd(&header->ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (last == 0u || threadIdx.x >= 8) return;
  unsigned* h = reinterpret_cast<unsigned*>(header);
  const unsigned word = threadIdx.x >= 6 ? 0u : atomicOr(&h[threadIdx.x], 0u);     // the coherent copy (other launches' atomics)
  if (threadIdx.x < 7) {
    h[kSpecStatusWord + threadIdx.x] = word;
    if (status_host != nullptr) status_host[threadIdx.x] = word;
  }
  __threadfence_system();
  if (threadIdx.x == 7) {                        // word [7]: "published", written last (a host that cleared it before the call polls it)
    h[kSpecStatusWord + 7] = 1u;
    if (status_host != nullptr) status_host[7] = 1u;
  }
  atomicExch(&h[threadIdx.x], 0u);
}

static int fill_levels(HashLevels& L, int n_levels, const float* scale, const unsigned* res, const unsigned* size,
                       const unsigned* offset, const unsigned* dense, float bound) {
  if (n_levels < 1 || n_levels > kMaxLevels) return fail(NERF_EINVAL, "hash grid: n_levels=%d (1..16)", n_levels);
  L.n_levels = n_levels;
  L.bound = bound;
  for (int i = 0; i < n_levels; ++i) {
    L.scale[i] = scale[i]; L.res[i] = res[i]; L.size[i] = size[i]; L.offset[i] = offset[i]; L.dense[i] = dense[i];
    if (size[i] == 0) return fail(NERF_EINVAL, "hash grid: level %d has size 0", i);
  }
  return NERF_OK;
}

}  // namespace nerf

using namespace nerf;

static int hash_fwd_impl(const float* pts, int64_t n, const float* table, const void* table_f16, int n_levels,
                         const float* scale_host, const unsigned* res_host, const unsigned* size_host,
                         const unsigned* offset_host, const unsigned* dense_host, float bound,
                         float* out_f32, void* out_nat_bf16, unsigned* idx_out, nerf_stream_t stream, int nat_f16 = 0,
                         void* bwd_workspace = nullptr, size_t bwd_workspace_bytes = 0, TableSet ts = TableSet{1, 0, 0});

extern "C" int nerf_hash_encode_fwd(const float* pts, int64_t n, const float* table, int n_levels,
                                    const float* scale_host, const unsigned* res_host, const unsigned* size_host,
                                    const unsigned* offset_host, const unsigned* dense_host, float bound,
                                    float* out_f32, void* out_nat_bf16, unsigned* idx_out, nerf_stream_t stream) {
  return hash_fwd_impl(pts, n, table, nullptr, n_levels, scale_host, res_host, size_host, offset_host, dense_host, bound, out_f32,
                       out_nat_bf16, idx_out, stream);
}

extern "C" int nerf_hash_encode_fwd_f16(const float* pts, int64_t n, const void* table_f16, int n_levels,
                                        const float* scale_host, const unsigned* res_host, const unsigned* size_host,
                                        const unsigned* offset_host, const unsigned* dense_host, float bound,
                                        float* out_f32, void* out_nat_bf16, nerf_stream_t stream) {
  NERF_REQUIRE(table_f16 != nullptr || n == 0, "nerf_hash_encode_fwd_f16: NULL table");
  return hash_fwd_impl(pts, n, nullptr, table_f16, n_levels, scale_host, res_host, size_host, offset_host, dense_host, bound, out_f32,
                       out_nat_bf16, nullptr, stream);
}

extern "C" int nerf_hash_encode_fwd_nat(const float* pts, int64_t n, const float* table_f32, const void* table_f16, int n_levels,
                                        const float* scale_host, const unsigned* res_host, const unsigned* size_host,
                                        const unsigned* offset_host, const unsigned* dense_host, float bound,
                                        void* out_nat, int nat_dtype, nerf_stream_t stream) {
  NERF_REQUIRE((table_f32 == nullptr) != (table_f16 == nullptr), "nerf_hash_encode_fwd_nat: exactly one of table_f32 / table_f16");
  NERF_REQUIRE(nat_dtype == 0 || nat_dtype == 1, "nerf_hash_encode_fwd_nat: nat_dtype=%d (0 bf16, 1 fp16)", nat_dtype);
  NERF_REQUIRE(n == 0 || out_nat != nullptr, "nerf_hash_encode_fwd_nat: out_nat is NULL");
  return hash_fwd_impl(pts, n, table_f32, table_f16, n_levels, scale_host, res_host, size_host, offset_host, dense_host, bound,
                       nullptr, out_nat, nullptr, stream, nat_dtype);
}

extern "C" int nerf_hash_encode_fwd_nat_tables(const float* pts, int64_t n, const void* tables_f16, int n_tables, int64_t table_stride,
                                               int n_levels, const float* scale_host, const unsigned* res_host,
                                               const unsigned* size_host, const unsigned* offset_host, const unsigned* dense_host,
                                               float bound, void* out_nat, int64_t out_stride_bytes, int nat_dtype, nerf_stream_t stream) {
  NERF_REQUIRE(n_tables >= 1 && n_tables <= 8 && table_stride >= 0 && out_stride_bytes >= 0 && out_stride_bytes % 2 == 0,
               "nerf_hash_encode_fwd_nat_tables: n_tables=%d", n_tables);
  NERF_REQUIRE(nat_dtype == 0 || nat_dtype == 1, "nerf_hash_encode_fwd_nat_tables: nat_dtype=%d (0 bf16, 1 fp16)", nat_dtype);
  NERF_REQUIRE(n == 0 || (tables_f16 != nullptr && out_nat != nullptr), "nerf_hash_encode_fwd_nat_tables: NULL pointer");
  return hash_fwd_impl(pts, n, nullptr, tables_f16, n_levels, scale_host, res_host, size_host, offset_host, dense_host, bound,
                       nullptr, out_nat, nullptr, stream, nat_dtype, nullptr, 0, TableSet{n_tables, table_stride, out_stride_bytes / 2});
}

extern "C" int nerf_hash_encode_fwd_f16_hist(const float* pts, int64_t n, const void* table_f16, int n_levels,
                                             const float* scale_host, const unsigned* res_host, const unsigned* size_host,
                                             const unsigned* offset_host, const unsigned* dense_host, float bound,
                                             void* out_nat_bf16, void* bwd_workspace, size_t bwd_workspace_bytes, nerf_stream_t stream) {
  NERF_REQUIRE(n == 0 || (table_f16 && out_nat_bf16 && bwd_workspace), "nerf_hash_encode_fwd_f16_hist: NULL pointer");
  return hash_fwd_impl(pts, n, nullptr, table_f16, n_levels, scale_host, res_host, size_host, offset_host, dense_host, bound, nullptr,
                       out_nat_bf16, nullptr, stream, 0, bwd_workspace, bwd_workspace_bytes);
}

extern "C" int nerf_hash_encode_bwd_ws_slots(void* workspace, int64_t n, int n_levels, void** amax_bits_out, void** grad_lm_out) {
  NERF_REQUIRE(workspace && n > 0 && n_levels >= 1 && n_levels <= kMaxPlanLevels && amax_bits_out && grad_lm_out, "nerf_hash_encode_bwd_ws_slots: bad arguments");
  const BinWorkspace w = carve(workspace, n, n_levels);
  *amax_bits_out = reinterpret_cast<unsigned*>(w.header) + kAmaxSlotWord;       // kAmaxSlots words (common.h::publish_amax_slots)
  *grad_lm_out = w.grad_lm;
  return NERF_OK;
}

static int hash_fwd_impl(const float* pts, int64_t n, const float* table, const void* table_f16, int n_levels,
                         const float* scale_host, const unsigned* res_host, const unsigned* size_host,
                         const unsigned* offset_host, const unsigned* dense_host, float bound,
                         float* out_f32, void* out_nat_bf16, unsigned* idx_out, nerf_stream_t stream, int nat_f16,
                         void* bwd_workspace, size_t bwd_workspace_bytes, TableSet ts) {
  NERF_REQUIRE(n >= 0, "nerf_hash_encode_fwd: n=%lld", (long long)n);
  if (n == 0) return NERF_OK;
  NERF_REQUIRE(pts && (table || table_f16) && scale_host && res_host && size_host && offset_host && dense_host,
               "nerf_hash_encode_fwd: NULL pointer");
  NERF_REQUIRE(out_f32 || out_nat_bf16, "nerf_hash_encode_fwd: no output requested");
  HashLevels L;
  int rc = fill_levels(L, n_levels, scale_host, res_host, size_host, offset_host, dense_host, bound);
  if (rc != NERF_OK) return rc;
  const int64_t n_pad = (n + 127) / 128 * 128;
  int64_t blocks = (n_pad + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  // occupancy throttle: the kernel uses no LDS; asking for 36 KiB per workgroup leaves 4 workgroups (1024 threads) per CU,
  // i.e. about 1.3 level tables in flight on the chip instead of 2.6 -- the 2 MB tables of the hashed levels then stay in
  // the 4 MB L2 of each XCD (91 -> 85 us on 200 k points; 3 or 2 workgroups per CU: 98 us)
  const int lds_kb = options().hash_fwd_lds_kb;
  int lds = (lds_kb < 0 ? 0 : (lds_kb > 64 ? 64 : lds_kb)) * 1024;
  // bwd_workspace: this forward also counts the corners per (level, slice) into the binned backward's workspace (header and
  // counts are zeroed here), so that nerf_hash_encode_bwd_ws_store_precounted can skip its count pass
  unsigned* hist_count = nullptr;
  LevelBins lb{};
  if (bwd_workspace != nullptr) {
    NERF_REQUIRE(bwd_workspace_bytes >= bin_workspace_bytes(n, n_levels), "nerf_hash_encode_fwd_hist: workspace of %zu bytes, need %zu",
                 bwd_workspace_bytes, bin_workspace_bytes(n, n_levels));
    unsigned max_slices = 0;
    for (int i = 0; i < n_levels; ++i) {
      const unsigned slices = (size_host[i] + kSlice - 1) / kSlice;
      NERF_REQUIRE(slices <= kMaxSlices, "nerf_hash_encode_fwd_hist: level %d has %u slices (max %u)", i, slices, kMaxSlices);
      max_slices = slices > max_slices ? slices : max_slices;
      lb.bin0[i + 1] = lb.bin0[i] + slices;
    }
    NERF_REQUIRE(lb.bin0[n_levels] <= (unsigned)kMaxBins, "nerf_hash_encode_fwd_hist: %u bins (max %d)", lb.bin0[n_levels], kMaxBins);
    const BinWorkspace w = carve(bwd_workspace, n, n_levels);
    if (hipMemsetAsync(w.header, 0, 256 + sizeof(unsigned) * lb.bin0[n_levels], as_stream(stream)) != hipSuccess)
      return fail(NERF_ELAUNCH, "nerf_hash_encode_fwd_hist: memset failed");
    hist_count = w.count;
    if (lds < (int)(max_slices * sizeof(unsigned))) lds = (int)(max_slices * sizeof(unsigned));
  }
  const bool xcd = options().hash_xcd != 0;
  NERF_REQUIRE(ts.n >= 1 && (ts.n == 1 || (out_f32 == nullptr && idx_out == nullptr && bwd_workspace == nullptr && out_nat_bf16 != nullptr)),
               "nerf_hash_encode_fwd_nat_tables: several tables write operand images only");
  const dim3 grid = level_chunk_grid(n_levels * ts.n, blocks, xcd);
  const int n_chunks = xcd ? (int)blocks : 0;
  if (table_f16 != nullptr)
    hipLaunchKernelGGL(hash_fwd_kernel<half2_t>, grid, dim3(256), lds, as_stream(stream), pts, n, n_pad,
                       static_cast<const half2_t*>(table_f16), L, out_f32, static_cast<__bf16*>(out_nat_bf16), idx_out, nat_f16,
                       hist_count, lb, n_chunks, ts);
  else
    hipLaunchKernelGGL(hash_fwd_kernel<float2>, grid, dim3(256), lds, as_stream(stream), pts, n, n_pad,
                       reinterpret_cast<const float2*>(table), L, out_f32, static_cast<__bf16*>(out_nat_bf16), idx_out, nat_f16,
                       hist_count, lb, n_chunks, ts);
  return check_launch("nerf_hash_encode_fwd");
}

// option "deterministic": the integer staging array of the cut bins (kFlushInt) lives in the slack of the record area -- a counted call
// writes at most n * 8 * levels records of a capacity 1.25 x that + 64 per bin.  dense_entries: entries of the leading dense levels
// (the only levels whose bins are cut then).  ptr NULL: no room (or no dense level): no bin is cut at all, as before.
struct DetStage { unsigned long long* ptr; unsigned dense_entries; };
static DetStage det_stage(const BinWorkspace& w, int64_t n, int plan_levels, int n_levels, int n_tables, const unsigned* size_host,
                          const unsigned* dense_host) {
  DetStage d{nullptr, 0};
  if (!options().deterministic) return d;
  unsigned dense = 0;
  for (int i = 0; i < n_levels && dense_host[i]; ++i) dense += size_host[i];
  const size_t used = (size_t)n * 8 * (size_t)plan_levels, cap = bin_record_capacity(n, plan_levels);
  const size_t need = (size_t)n_tables * dense * 2;            // 64-bit words, one BinRecord's size each
  if (dense == 0 || used + need > cap) return d;
  static_assert(sizeof(BinRecord) == sizeof(unsigned long long), "staging words");
  d.ptr = reinterpret_cast<unsigned long long*>(w.records + used);
  d.dense_entries = dense;
  return d;
}

static int hash_bwd_impl(const float* pts, int64_t n, int n_levels, const float* scale_host,
                         const unsigned* res_host, const unsigned* size_host, const unsigned* offset_host,
                         const unsigned* dense_host, float bound, const float* d_feat, float* d_table,
                         int level0, int level1, nerf_stream_t stream, void* workspace = nullptr, size_t workspace_bytes = 0,
                         bool overwrite = false, int precounted = 0, void* status_host = nullptr) {
  // precounted: 0 the count pass runs here; 1 counts by the forward (nerf_hash_encode_fwd_f16_hist); 2 speculative: no counts at all
  // (capacities from the last call's true counts).  1 and 2: largest |gradient| and level-major gradients by nerf_imlp_bwd_lm
  const bool spec = precounted == 2;
  NERF_REQUIRE(level0 >= 0 && level0 <= level1 && level1 <= n_levels, "nerf_hash_encode_bwd: levels [%d, %d) of %d", level0, level1, n_levels);
  NERF_REQUIRE(n >= 0, "nerf_hash_encode_bwd: n=%lld", (long long)n);
  if (n == 0) return NERF_OK;
  NERF_REQUIRE(pts && (d_feat || precounted) && d_table && scale_host && res_host && size_host && offset_host && dense_host,
               "nerf_hash_encode_bwd: NULL pointer");
  HashLevels L;
  int rc = fill_levels(L, n_levels, scale_host, res_host, size_host, offset_host, dense_host, bound);
  if (rc != NERF_OK) return rc;
  rc = ensure_dynamic_lds((const void*)hash_bwd_kernel<true>, kLdsEntries * 8, "nerf_hash_encode_bwd");
  if (rc != NERF_OK) return rc;
  int n_small = 0;                      // leading levels whose table fits in LDS
  while (n_small < n_levels && size_host[n_small] <= (unsigned)kLdsEntries) ++n_small;
  // the workspace form bins EVERY level (the small dense ones too: their LDS pass is ds_add_f32-bound)
  const int first_big = level0;
  bool binned = workspace != nullptr && first_big < level1 && options().hash_bwd_only_level < 0 && !options().hash_bwd_atomic;
  BinPlan plan;
  if (binned) {
    plan.first = first_big;
    plan.count = level1 - first_big;
    plan.n_tables = 1;
    plan.table_stride = 0;
    plan.dfeat_stride = 0;
    plan.bin0[0] = 0;
    unsigned table_entries = 0;
    for (int i = 0; i < n_levels; ++i) table_entries = offset_host[i] + size_host[i] > table_entries ? offset_host[i] + size_host[i] : table_entries;
    for (int i = 0; i < plan.count; ++i) {
      const unsigned slices = (size_host[first_big + i] + kSlice - 1) / kSlice;
      if (slices > kMaxSlices) binned = false;
      plan.bin0[i + 1] = plan.bin0[i] + slices;
    }
    if (binned && plan.bin0[plan.count] > (unsigned)kMaxBins) binned = false;
    if (binned) {
      NERF_REQUIRE(workspace_bytes >= bin_workspace_bytes(n, n_levels), "nerf_hash_encode_bwd_ws: workspace of %zu bytes, need %zu",
                   workspace_bytes, bin_workspace_bytes(n, n_levels));
      NERF_REQUIRE(bin_record_capacity(n, n_levels) < 0xffffffffull, "nerf_hash_encode_bwd_ws: n=%lld too large for 32-bit record offsets", (long long)n);
      const BinWorkspace w = carve(workspace, n, n_levels);
      const unsigned n_bins = plan.bin0[plan.count];
      const bool whole = level0 == 0 && level1 == n_levels;      // the bins of a level range are numbered from 0: est[] only for whole calls
      NERF_REQUIRE(!spec || (whole && overwrite), "nerf_hash_encode_bwd_ws_store_spec: all levels, overwrite form");
      if (precounted == 0 && hipMemsetAsync(w.header, 0, 256 + sizeof(unsigned) * n_bins, as_stream(stream)) != hipSuccess)   // header + counts
        return fail(NERF_ELAUNCH, "nerf_hash_encode_bwd_ws: memset failed");
      int64_t bx = (n + 511) / 512;
      const int64_t bx_count = bx > 256 ? 256 : bx, bx_scatter = bx > 128 ? 128 : bx;
      // level-major gradients: written by the count pass here, or by nerf_imlp_bwd_lm (precounted forms called without d_feat)
      const bool point_major = precounted != 0 ? d_feat == nullptr : n_bins <= kPmBins;
      if (precounted != 0) {
        // counts by the forward (nerf_hash_encode_fwd_f16_hist), largest |gradient| and level-major gradients by the
        // decoder's backward (nerf_imlp_bwd_lm): nothing to do here
      } else if (point_major) {
        int64_t bpm = (n + 255) / 256;
        if (bpm > 1024) bpm = 1024;
        const int per_row = pm_levels_per_row(n, plan.count);
        hipLaunchKernelGGL(hash_bin_count_pm_kernel, dim3((int)bpm, (plan.count + per_row - 1) / per_row), dim3(256), 0, as_stream(stream),
                           pts, n, L, plan, d_feat, w.count, w.header, w.grad_lm, per_row);
      } else
        hipLaunchKernelGGL(hash_bin_count_kernel, dim3((int)bx_count, plan.count), dim3(512), 0, as_stream(stream), pts, n, L, plan, d_feat,
                           w.count, w.header);
      const float2* grad_lm = point_major ? w.grad_lm : nullptr;
      const DetStage ds = det_stage(w, n, n_levels, n_levels, 1, size_host, dense_host);
      const unsigned chunk = (options().deterministic && ds.ptr == nullptr) ? 0xffffffffu : kChunk;
      bool any_staged = false, any_direct = false;
      for (int i = 0; i < plan.count; ++i) (plan.bin0[i + 1] - plan.bin0[i] <= kStagedBins ? any_staged : any_direct) = true;
      NERF_REQUIRE(!spec || !any_direct, "nerf_hash_encode_bwd_ws_store_spec: a level with more than %u slices", kStagedBins);
      hipLaunchKernelGGL(hash_bin_plan_kernel, dim3(1), dim3(1024), 0, as_stream(stream), L, plan, w.count, w.cursor, w.items, w.header,
                         overwrite ? 1 : 0, chunk, whole ? w.est : nullptr, spec ? w.start : nullptr,
                         spec ? (unsigned)bin_record_capacity(n, n_levels) : 0u, precounted != 0 ? 1 : 0, ds.dense_entries, ds.ptr != nullptr);
      float* zero_table = overwrite ? d_table : nullptr;
      const unsigned* spec_start = spec ? w.start : nullptr;
      if (any_staged)
        hipLaunchKernelGGL(hash_bin_scatter_kernel<true>, dim3((int)bx_scatter, plan.count), dim3(512), 0, as_stream(stream), pts, n, L,
                           plan, d_feat, w.cursor, w.records, w.header, grad_lm, w.count, zero_table, precounted == 1 ? 1 : 0, chunk,
                           spec_start, w.overflow_bin, w.header, (unsigned)bin_record_capacity(n, n_levels), ds.ptr, ds.dense_entries);
      if (any_direct)
        hipLaunchKernelGGL(hash_bin_scatter_kernel<false>, dim3((int)bx_scatter, plan.count), dim3(512), 0, as_stream(stream), pts, n, L,
                           plan, d_feat, w.cursor, w.records, w.header, grad_lm, w.count, zero_table, precounted == 1 ? 1 : 0, chunk,
                           (const unsigned*)nullptr, w.overflow_bin, w.header, 0u, ds.ptr, ds.dense_entries);
      size_t grid = (size_t)n * 8 * plan.count / kChunk + n_bins;
      if (grid > 4096) grid = 4096;             // persistent beyond that: items are taken round-robin
      hipLaunchKernelGGL(hash_bin_reduce_kernel, dim3((unsigned)grid), dim3(512), 0, as_stream(stream), w.header, w.items, w.records,
                         d_table, table_entries, spec ? w.cursor : (const unsigned*)nullptr, spec_start, w.est, ds.ptr);
      if (ds.ptr != nullptr)
        hipLaunchKernelGGL(hash_bin_stage_convert_kernel, dim3(256), dim3(256), 0, as_stream(stream), w.header, w.items, ds.ptr, d_table,
                           table_entries, overwrite ? 0 : 1);
      if (spec)
        hipLaunchKernelGGL(hash_bin_overflow_kernel, dim3(64), dim3(256), 0, as_stream(stream), w.header,
                           w.records + bin_record_capacity(n, n_levels), w.overflow_bin, L, plan, d_table, static_cast<unsigned*>(status_host));
    }
  }
  if (options().deterministic && !binned && level0 < level1)
    return fail(NERF_EINVAL, "nerf_hash_encode_bwd: option \"deterministic\" needs the workspace form (nerf_hash_encode_bwd_ws*) -- the "
                             "other forms end in float atomics");
  if (precounted != 0 && !binned)
    return fail(NERF_EINVAL, "nerf_hash_encode_bwd_ws_store_precounted: the binned form is not available for this table shape / option set");
  if (overwrite && !binned && level0 < level1) {
    // the atomic forms accumulate: give them the zeroed range the overwrite contract promises
    const size_t e0 = offset_host[level0], e1 = offset_host[level1 - 1] + size_host[level1 - 1];
    if (hipMemsetAsync(d_table + 2 * e0, 0, sizeof(float) * 2 * (e1 - e0), as_stream(stream)) != hipSuccess)
      return fail(NERF_ELAUNCH, "nerf_hash_encode_bwd_ws_store: memset failed");
  }
  if (level0 < n_small && !binned) {
    const int hi = level1 < n_small ? level1 : n_small;
    int64_t bx = (n + 511) / 512;
    if (bx > 128) bx = 128;             // each workgroup flushes its whole LDS table once
    hipLaunchKernelGGL(hash_bwd_kernel<true>, dim3((int)bx, hi - level0), dim3(512), kLdsEntries * 8, as_stream(stream), pts, n, L,
                       level0, d_feat, d_table);
  }
  if (n_small < level1 && !binned) {
    int64_t bx = (4 * n + 511) / 512;
    if (bx > 1024) bx = 1024;
    int first = level0 > n_small ? level0 : n_small, count = level1 - first;
    if (options().hash_bwd_only_level >= 0) {   // development aid: time one level's atomics
      first = options().hash_bwd_only_level;
      count = 1;
      if (first < n_small || first >= level1) return check_launch("nerf_hash_encode_bwd");
    }
    hipLaunchKernelGGL(hash_bwd_kernel<false>, dim3((int)bx, count), dim3(512), 0, as_stream(stream), pts, n, L,
                       first, d_feat, d_table);
  }
  return check_launch("nerf_hash_encode_bwd");
}

extern "C" int nerf_hash_encode_bwd(const float* pts, int64_t n, int n_levels, const float* scale_host,
                                    const unsigned* res_host, const unsigned* size_host, const unsigned* offset_host,
                                    const unsigned* dense_host, float bound, const float* d_feat, float* d_table,
                                    nerf_stream_t stream) {
  return hash_bwd_impl(pts, n, n_levels, scale_host, res_host, size_host, offset_host, dense_host, bound, d_feat, d_table, 0,
                       n_levels, stream);
}

extern "C" int nerf_hash_encode_bwd_levels(const float* pts, int64_t n, int n_levels, const float* scale_host,
                                           const unsigned* res_host, const unsigned* size_host, const unsigned* offset_host,
                                           const unsigned* dense_host, float bound, const float* d_feat, float* d_table,
                                           int first_level, int end_level, nerf_stream_t stream) {
  return hash_bwd_impl(pts, n, n_levels, scale_host, res_host, size_host, offset_host, dense_host, bound, d_feat, d_table,
                       first_level, end_level, stream);
}

extern "C" size_t nerf_hash_encode_bwd_workspace_bytes(int64_t n, int n_levels) {
  if (n <= 0 || n_levels < 1 || n_levels > kMaxLevels) return 0;
  return bin_workspace_bytes(n, n_levels);
}

extern "C" int nerf_hash_encode_bwd_ws(const float* pts, int64_t n, int n_levels, const float* scale_host,
                                       const unsigned* res_host, const unsigned* size_host, const unsigned* offset_host,
                                       const unsigned* dense_host, float bound, const float* d_feat, float* d_table,
                                       int first_level, int end_level, void* workspace, size_t workspace_bytes,
                                       nerf_stream_t stream) {
  return hash_bwd_impl(pts, n, n_levels, scale_host, res_host, size_host, offset_host, dense_host, bound, d_feat, d_table,
                       first_level, end_level, stream, workspace, workspace_bytes);
}

extern "C" int nerf_hash_encode_bwd_ws_store(const float* pts, int64_t n, int n_levels, const float* scale_host,
                                             const unsigned* res_host, const unsigned* size_host, const unsigned* offset_host,
                                             const unsigned* dense_host, float bound, const float* d_feat, float* d_table,
                                             int first_level, int end_level, void* workspace, size_t workspace_bytes,
                                             nerf_stream_t stream) {
  NERF_REQUIRE(first_level >= 0 && first_level <= end_level && end_level <= n_levels, "nerf_hash_encode_bwd_ws_store: levels [%d, %d) of %d",
               first_level, end_level, n_levels);
  if (n == 0 && first_level < end_level) {            // nothing to scatter: the levels' range is still OVERWRITTEN (with zeros)
    NERF_REQUIRE(d_table && size_host && offset_host, "nerf_hash_encode_bwd_ws_store: NULL pointer");
    const size_t e0 = offset_host[first_level], e1 = offset_host[end_level - 1] + size_host[end_level - 1];
    if (hipMemsetAsync(d_table + 2 * e0, 0, sizeof(float) * 2 * (e1 - e0), as_stream(stream)) != hipSuccess)
      return fail(NERF_ELAUNCH, "nerf_hash_encode_bwd_ws_store: memset failed");
    return NERF_OK;
  }
  return hash_bwd_impl(pts, n, n_levels, scale_host, res_host, size_host, offset_host, dense_host, bound, d_feat, d_table,
                       first_level, end_level, stream, workspace, workspace_bytes, true);
}

extern "C" size_t nerf_hash_encode_bwd_tables_workspace_bytes(int64_t n, int n_levels, int n_tables) {
  if (n <= 0 || n_levels < 1 || n_levels > kMaxLevels || n_tables < 1 || n_levels * n_tables > kMaxPlanLevels) return 0;
  return bin_workspace_bytes(n, n_levels * n_tables);
}

// The overwrite-form scatter for n_tables tables of ONE level structure from the same points in one pass of count / plan /
// scatter / reduce launches (Part 4's three deformation grids: three passes of four small launches otherwise).
static int hash_bwd_tables_impl(const float* pts, int64_t n, int n_tables, int64_t table_stride, int n_levels,
                                const float* scale_host, const unsigned* res_host, const unsigned* size_host,
                                const unsigned* offset_host, const unsigned* dense_host, float bound,
                                const float* d_feat, int64_t dfeat_stride, float* d_table, void* workspace,
                                size_t workspace_bytes, nerf_stream_t stream, bool spec, void* status_host) {
  NERF_REQUIRE(n >= 0 && n_tables >= 1 && n_levels >= 1 && n_levels * n_tables <= kMaxPlanLevels && table_stride >= 0 && dfeat_stride >= 0,
               "nerf_hash_encode_bwd_ws_store_tables: n=%lld, %d tables of %d levels (at most %d virtual levels)", (long long)n, n_tables,
               n_levels, kMaxPlanLevels);
  NERF_REQUIRE(d_table && size_host && offset_host, "nerf_hash_encode_bwd_ws_store_tables: NULL pointer");
  unsigned entries = 0;
  for (int i = 0; i < n_levels; ++i) entries = offset_host[i] + size_host[i] > entries ? offset_host[i] + size_host[i] : entries;
  NERF_REQUIRE(n_tables == 1 || (uint64_t)table_stride >= entries, "nerf_hash_encode_bwd_ws_store_tables: table_stride %lld < %u entries",
               (long long)table_stride, entries);
  NERF_REQUIRE(((uint64_t)(n_tables - 1) * (uint64_t)table_stride + entries) < 0xffffffffull, "nerf_hash_encode_bwd_ws_store_tables: tables too large");
  if (n == 0) {                                      // nothing to scatter: every table is still OVERWRITTEN (with zeros)
    for (int t = 0; t < n_tables; ++t)
      if (hipMemsetAsync(d_table + 2 * (size_t)t * table_stride, 0, sizeof(float) * 2 * entries, as_stream(stream)) != hipSuccess)
        return fail(NERF_ELAUNCH, "nerf_hash_encode_bwd_ws_store_tables: memset failed");
    return NERF_OK;
  }
  NERF_REQUIRE(pts && (d_feat || spec) && workspace && scale_host && res_host && dense_host, "nerf_hash_encode_bwd_ws_store_tables: NULL pointer");
  HashLevels L;
  if (int rc = fill_levels(L, n_levels, scale_host, res_host, size_host, offset_host, dense_host, bound); rc != NERF_OK) return rc;
  BinPlan plan;
  plan.first = 0;
  plan.count = n_levels * n_tables;
  plan.n_tables = n_tables;
  plan.table_stride = (unsigned)table_stride;
  plan.dfeat_stride = dfeat_stride;
  plan.bin0[0] = 0;
  bool any_staged = false, any_direct = false;
  for (int v = 0; v < plan.count; ++v) {
    const unsigned slices = (size_host[v % n_levels] + kSlice - 1) / kSlice;
    NERF_REQUIRE(slices <= kMaxSlices, "nerf_hash_encode_bwd_ws_store_tables: level %d has %u slices (max %u)", v % n_levels, slices, kMaxSlices);
    plan.bin0[v + 1] = plan.bin0[v] + slices;
    (slices <= kStagedBins ? any_staged : any_direct) = true;
  }
  const unsigned n_bins = plan.bin0[plan.count];
  NERF_REQUIRE(n_bins <= kPmBins, "nerf_hash_encode_bwd_ws_store_tables: %u bins (this form holds all histograms in LDS: max %u)", n_bins, kPmBins);
  NERF_REQUIRE(workspace_bytes >= bin_workspace_bytes(n, plan.count), "nerf_hash_encode_bwd_ws_store_tables: workspace of %zu bytes, need %zu",
               workspace_bytes, bin_workspace_bytes(n, plan.count));
  NERF_REQUIRE((size_t)n * 8 * (size_t)plan.count < 0xffffffffull, "nerf_hash_encode_bwd_ws_store_tables: n=%lld too large for 32-bit record offsets",
               (long long)n);
  const BinWorkspace w = carve(workspace, n, plan.count);
  const DetStage ds = det_stage(w, n, plan.count, n_levels, n_tables, size_host, dense_host);
  const unsigned chunk = (options().deterministic && ds.ptr == nullptr) ? 0xffffffffu : kChunk;
  const unsigned capacity = (unsigned)bin_record_capacity(n, plan.count);
  const float2* grad_lm = w.grad_lm;
  if (spec) {
    // no count pass: capacities from the true counts the previous call on this workspace left in est[]; the chain's backward has
    // max-accumulated the largest |gradient| into the header (nerf_hash_encode_bwd_ws_slots) and hands d_feat over row-major
    NERF_REQUIRE(!any_direct && !options().deterministic, "nerf_hash_encode_bwd_ws_store_tables_spec: levels of at most %u slices, not with "
                 "option \"deterministic\"", kStagedBins);
    grad_lm = d_feat == nullptr ? w.grad_lm : nullptr;       // NULL d_feat: the producer wrote the level-major copy into the workspace
  } else {
    if (hipMemsetAsync(w.header, 0, 256 + sizeof(unsigned) * n_bins, as_stream(stream)) != hipSuccess)
      return fail(NERF_ELAUNCH, "nerf_hash_encode_bwd_ws_store_tables: memset failed");
    int64_t bpm = (n + 255) / 256;
    if (bpm > 1024) bpm = 1024;
    const int per_row = pm_levels_per_row(n, plan.count);
    hipLaunchKernelGGL(hash_bin_count_pm_kernel, dim3((int)bpm, (plan.count + per_row - 1) / per_row), dim3(256), 0, as_stream(stream), pts, n,
                       L, plan, d_feat, w.count, w.header, w.grad_lm, per_row);
  }
  // (the counted call leaves the bins' true counts in est[] for a later speculative one)
  hipLaunchKernelGGL(hash_bin_plan_kernel, dim3(1), dim3(1024), 0, as_stream(stream), L, plan, w.count, w.cursor, w.items, w.header, 1, chunk,
                     w.est, spec ? w.start : (unsigned*)nullptr, spec ? capacity : 0u, spec ? 1 : 0, ds.dense_entries, ds.ptr != nullptr);
  const unsigned* spec_start = spec ? w.start : nullptr;
  int64_t bx = (n + 511) / 512;
  const int64_t bx_scatter = bx > 128 ? 128 : bx;
  if (any_staged)
    hipLaunchKernelGGL(hash_bin_scatter_kernel<true>, dim3((int)bx_scatter, plan.count), dim3(512), 0, as_stream(stream), pts, n, L, plan,
                       d_feat, w.cursor, w.records, w.header, grad_lm, w.count, d_table, 0, chunk, spec_start, w.overflow_bin, w.header,
                       spec ? capacity : 0u, ds.ptr, ds.dense_entries);
  if (any_direct)
    hipLaunchKernelGGL(hash_bin_scatter_kernel<false>, dim3((int)bx_scatter, plan.count), dim3(512), 0, as_stream(stream), pts, n, L, plan,
                       d_feat, w.cursor, w.records, w.header, grad_lm, w.count, d_table, 0, chunk, (const unsigned*)nullptr, w.overflow_bin, w.header, 0u,
                       ds.ptr, ds.dense_entries);
  size_t grid = (size_t)n * 8 * plan.count / kChunk + n_bins;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(hash_bin_reduce_kernel, dim3((unsigned)grid), dim3(512), 0, as_stream(stream), w.header, w.items, w.records, d_table,
                     (unsigned)((uint64_t)(n_tables - 1) * (uint64_t)table_stride + entries), spec ? w.cursor : (const unsigned*)nullptr, spec_start,
                     w.est, ds.ptr);
  if (ds.ptr != nullptr)
    hipLaunchKernelGGL(hash_bin_stage_convert_kernel, dim3(256), dim3(256), 0, as_stream(stream), w.header, w.items, ds.ptr, d_table,
                       (unsigned)((uint64_t)(n_tables - 1) * (uint64_t)table_stride + entries), 0);
  if (spec)
    hipLaunchKernelGGL(hash_bin_overflow_kernel, dim3(64), dim3(256), 0, as_stream(stream), w.header, w.records + capacity, w.overflow_bin, L,
                       plan, d_table, static_cast<unsigned*>(status_host));
  return check_launch("nerf_hash_encode_bwd_ws_store_tables");
}

extern "C" int nerf_hash_encode_bwd_ws_store_tables(const float* pts, int64_t n, int n_tables, int64_t table_stride, int n_levels,
                                                    const float* scale_host, const unsigned* res_host, const unsigned* size_host,
                                                    const unsigned* offset_host, const unsigned* dense_host, float bound,
                                                    const float* d_feat, int64_t dfeat_stride, float* d_table, void* workspace,
                                                    size_t workspace_bytes, nerf_stream_t stream) {
  return hash_bwd_tables_impl(pts, n, n_tables, table_stride, n_levels, scale_host, res_host, size_host, offset_host, dense_host, bound, d_feat,
                              dfeat_stride, d_table, workspace, workspace_bytes, stream, false, nullptr);
}

// ... its speculative form (see nerf_hash_encode_bwd_ws_store_spec): no count pass, capacities from the previous call's true counts
// on this workspace; d_feat row-major, its largest magnitude max-accumulated into the slot nerf_hash_encode_bwd_ws_slots(workspace, n,
// n_levels * n_tables, ...) names by the producer (nerf_p4_deform_bwd)
extern "C" int nerf_hash_encode_bwd_ws_store_tables_spec(const float* pts, int64_t n, int n_tables, int64_t table_stride, int n_levels,
                                                         const float* scale_host, const unsigned* res_host, const unsigned* size_host,
                                                         const unsigned* offset_host, const unsigned* dense_host, float bound,
                                                         const float* d_feat, int64_t dfeat_stride, float* d_table, void* workspace,
                                                         size_t workspace_bytes, void* status_host, nerf_stream_t stream) {
  NERF_REQUIRE(n > 0, "nerf_hash_encode_bwd_ws_store_tables_spec: n=%lld", (long long)n);
  return hash_bwd_tables_impl(pts, n, n_tables, table_stride, n_levels, scale_host, res_host, size_host, offset_host, dense_host, bound, d_feat,
                              dfeat_stride, d_table, workspace, workspace_bytes, stream, true, status_host);
}

extern "C" int nerf_hash_encode_bwd_ws_store_precounted(const float* pts, int64_t n, int n_levels, const float* scale_host,
                                                        const unsigned* res_host, const unsigned* size_host,
                                                        const unsigned* offset_host, const unsigned* dense_host, float bound,
                                                        float* d_table, void* workspace, size_t workspace_bytes, nerf_stream_t stream) {
  NERF_REQUIRE(n > 0 && workspace != nullptr, "nerf_hash_encode_bwd_ws_store_precounted: n=%lld, workspace %p", (long long)n, workspace);
  return hash_bwd_impl(pts, n, n_levels, scale_host, res_host, size_host, offset_host, dense_host, bound, nullptr, d_table, 0, n_levels,
                       stream, workspace, workspace_bytes, true, 1);
}

// The speculative form: NO count pass -- bin capacities from the true counts the previous call (counted or speculative, all levels,
// on this workspace) left behind.  Call nerf_hash_encode_bwd_spec_begin before the decoder's backward (which max-accumulates the
// largest |gradient| into the workspace's slot: nerf_imlp_bwd_amax with row-major d_feat [n, 2L] handed over here, or nerf_imlp_bwd_lm
// with the level-major copy in the workspace and d_feat NULL), then this.  A record that does not fit its bin is
// added with a float atomic at the end; *header (nerf_hash_encode_bwd_spec_status) tells how many, and whether any was lost.
extern "C" int nerf_hash_encode_bwd_spec_begin(void* workspace, nerf_stream_t stream) {
  NERF_REQUIRE(workspace != nullptr, "nerf_hash_encode_bwd_spec_begin: NULL workspace");
  if (hipMemsetAsync(workspace, 0, 256, as_stream(stream)) != hipSuccess)        // header, status block, the producer's amax words
    return fail(NERF_ELAUNCH, "nerf_hash_encode_bwd_spec_begin: memset failed");
  return NERF_OK;
}

extern "C" int nerf_hash_encode_bwd_ws_store_spec(const float* pts, int64_t n, int n_levels, const float* scale_host,
                                                  const unsigned* res_host, const unsigned* size_host, const unsigned* offset_host,
                                                  const unsigned* dense_host, float bound, const float* d_feat, float* d_table,
                                                  void* workspace, size_t workspace_bytes, void* status_host, nerf_stream_t stream) {
  NERF_REQUIRE(n > 0 && workspace != nullptr, "nerf_hash_encode_bwd_ws_store_spec: n=%lld, workspace %p", (long long)n, workspace);
  NERF_REQUIRE(!options().deterministic, "nerf_hash_encode_bwd_ws_store_spec: overflow records end in float atomics (option \"deterministic\" is set)");
  return hash_bwd_impl(pts, n, n_levels, scale_host, res_host, size_host, offset_host, dense_host, bound, d_feat, d_table, 0, n_levels,
                       stream, workspace, workspace_bytes, true, 2, status_host);
}

// device address of the 8-word status block the LAST speculative call published: [0] items, [1] record capacity planned, [2] largest
// |gradient| bits, [3] records that went to the overflow list, [4] != 0: records were LOST (the gradient of that call is incomplete: 1
// the overflow list was full, 2 the estimates did not fit the workspace), [5] capacity planned.  The same eight words go to the
// status_host block given to the call (host-mapped pinned memory: readable once an event recorded behind the call has completed).
extern "C" const void* nerf_hash_encode_bwd_spec_status(const void* workspace) {
  return workspace == nullptr ? nullptr : static_cast<const char*>(workspace) + sizeof(unsigned) * kSpecStatusWord;
}

static int hash_bwd_input_impl(const float* pts, int64_t n, const float* table, const void* table_f16, int n_levels,
                               const float* scale_host, const unsigned* res_host, const unsigned* size_host,
                               const unsigned* offset_host, const unsigned* dense_host, float bound,
                               const float* d_feat, float* d_pts, nerf_stream_t stream, int accumulate = 0,
                               const float2* grad_lm = nullptr) {
  NERF_REQUIRE(n >= 0, "nerf_hash_encode_bwd_input: n=%lld", (long long)n);
  if (n == 0) return NERF_OK;
  NERF_REQUIRE(pts && (table || table_f16) && (d_feat || grad_lm) && d_pts && scale_host && res_host && size_host && offset_host && dense_host,
               "nerf_hash_encode_bwd_input: NULL pointer");
  HashLevels L;
  int rc = fill_levels(L, n_levels, scale_host, res_host, size_host, offset_host, dense_host, bound);
  if (rc != NERF_OK) return rc;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  NERF_REQUIRE(!(options().deterministic && grad_lm != nullptr), "nerf_hash_encode_bwd_input_lm: not with option \"deterministic\"");
  if (options().deterministic) {          // levels summed in order per point: no float atomics
    if (table_f16 != nullptr)
      hipLaunchKernelGGL(hash_bwd_input_ordered_kernel<half2_t>, dim3((int)blocks), dim3(256), 0, as_stream(stream), pts, n,
                         static_cast<const half2_t*>(table_f16), L, d_feat, d_pts, accumulate);
    else
      hipLaunchKernelGGL(hash_bwd_input_ordered_kernel<float2>, dim3((int)blocks), dim3(256), 0, as_stream(stream), pts, n,
                         reinterpret_cast<const float2*>(table), L, d_feat, d_pts, accumulate);
    return check_launch("nerf_hash_encode_bwd_input (ordered)");
  }
  if (!accumulate && hipMemsetAsync(d_pts, 0, sizeof(float) * 3 * (size_t)n, as_stream(stream)) != hipSuccess)
    return fail(NERF_ELAUNCH, "nerf_hash_encode_bwd_input: memset failed");
  const bool xcd = options().hash_xcd != 0;
  const dim3 grid = level_chunk_grid(n_levels, blocks, xcd);
  if (table_f16 != nullptr)
    hipLaunchKernelGGL(hash_bwd_input_kernel<half2_t>, grid, dim3(256), 0, as_stream(stream), pts, n,
                       static_cast<const half2_t*>(table_f16), L, d_feat, d_pts, xcd ? (int)blocks : 0, grad_lm);
  else
    hipLaunchKernelGGL(hash_bwd_input_kernel<float2>, grid, dim3(256), 0, as_stream(stream), pts, n,
                       reinterpret_cast<const float2*>(table), L, d_feat, d_pts, xcd ? (int)blocks : 0, grad_lm);
  return check_launch("nerf_hash_encode_bwd_input");
}

extern "C" int nerf_hash_encode_bwd_input(const float* pts, int64_t n, const float* table, int n_levels,
                                          const float* scale_host, const unsigned* res_host, const unsigned* size_host,
                                          const unsigned* offset_host, const unsigned* dense_host, float bound,
                                          const float* d_feat, float* d_pts, nerf_stream_t stream) {
  return hash_bwd_input_impl(pts, n, table, nullptr, n_levels, scale_host, res_host, size_host, offset_host, dense_host, bound, d_feat,
                             d_pts, stream);
}

// the same, ADDED to d_pts (a gradient that reaches the positions by another path as well -- Part 4: the displacement regulariser's --
// is already there: no zeroing launch, no separate add)
extern "C" int nerf_hash_encode_bwd_input_f16_accum(const float* pts, int64_t n, const void* table_f16, int n_levels,
                                                    const float* scale_host, const unsigned* res_host, const unsigned* size_host,
                                                    const unsigned* offset_host, const unsigned* dense_host, float bound,
                                                    const float* d_feat, float* d_pts, nerf_stream_t stream) {
  return hash_bwd_input_impl(pts, n, nullptr, table_f16, n_levels, scale_host, res_host, size_host, offset_host, dense_host, bound,
                             d_feat, d_pts, stream, 1);
}

// ... from LEVEL-MAJOR feature gradients [n_levels][n] float2 (what a producer writes into the hash backward's workspace for the
// speculative scatter, nerf_hash_encode_bwd_ws_slots); accumulate != 0: added to d_pts
extern "C" int nerf_hash_encode_bwd_input_lm_f16(const float* pts, int64_t n, const void* table_f16, int n_levels,
                                                 const float* scale_host, const unsigned* res_host, const unsigned* size_host,
                                                 const unsigned* offset_host, const unsigned* dense_host, float bound,
                                                 const void* grad_lm, float* d_pts, int accumulate, nerf_stream_t stream) {
  NERF_REQUIRE(n == 0 || grad_lm != nullptr, "nerf_hash_encode_bwd_input_lm_f16: grad_lm is NULL");
  return hash_bwd_input_impl(pts, n, nullptr, table_f16, n_levels, scale_host, res_host, size_host, offset_host, dense_host, bound,
                             nullptr, d_pts, stream, accumulate, static_cast<const float2*>(grad_lm));
}

extern "C" int nerf_hash_encode_bwd_input_f16(const float* pts, int64_t n, const void* table_f16, int n_levels,
                                              const float* scale_host, const unsigned* res_host, const unsigned* size_host,
                                              const unsigned* offset_host, const unsigned* dense_host, float bound,
                                              const float* d_feat, float* d_pts, nerf_stream_t stream) {
  return hash_bwd_input_impl(pts, n, nullptr, table_f16, n_levels, scale_host, res_host, size_host, offset_host, dense_host, bound,
                             d_feat, d_pts, stream);
}
