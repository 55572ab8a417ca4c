/* nerf_hip.h -- C ABI of the MI355X (gfx950) NeRF volumetric-rendering hot path.
 *
 * Drop-in boundary for CV-Project2025/Project-NeRF.  The reference has no FFI of
 * its own: the path sits behind two Python operator surfaces (src/renderer.py,
 * src/core.py + src/abstract.py) and, for the Instant variants, behind the
 * third-party `tinycudann` bindings.  Every entry point below names the
 * reference interface (file:line) it replaces.  INTEGRATION.md shows the ctypes
 * stubs a maintainer of the reference would add.
 *
 * Conventions (all functions):
 *   - extern "C", plain pointers and sizes; no torch / C++ types.
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`.
 *   - pointers are borrowed for the duration of the call only; no hidden
 *     allocation: the caller passes outputs and workspaces.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are
 *     asynchronous w.r.t. the host and safe to capture in a hipGraph.
 *   - return 0 on success, a negative NERF_E* code otherwise; never throws.
 *     nerf_last_error() returns a thread-local, NUL-terminated description.
 *   - fp32 tensors are row-major and densely packed.
 */
#ifndef NERF_HIP_H
#define NERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NERF_OK 0
#define NERF_EINVAL (-22)  /* bad argument (shape, alignment, NULL) */
#define NERF_ELAUNCH (-5)  /* HIP launch / runtime failure          */
#define NERF_ENOSYS (-38)  /* configuration not compiled in         */

/* 2 (round 2 -> 3): nerf_tv_normsq gained `grad_scale`, the training images of the decoder changed format (see
 * nerf_mlp_fwd).  A host built against another version must refuse to run: compare nerf_abi_version() with the
 * NERF_ABI_VERSION it was compiled against (project-nerf_amd/_lib.py does; INTEGRATION.md shows the check). */
/* 3 (round 3 -> 4): the squared-norm scalar of nerf_tv_normsq* became a workspace of NERF_NORMSQ_WS_FLOATS floats (the sum over
 * workgroups is taken in workgroup order: the same bits on every run and every data-parallel replica);
 * nerf_composite_mse_bwd / nerf_composite_mse_reg_bwd gained `sum_ws`; the imlp / Part 4 workspaces grew (partial tiles);
 * option "deterministic" and nerf_sample_compact_ordered are new. */
#define NERF_ABI_VERSION 3

typedef void* nerf_stream_t;

const char* nerf_last_error(void);
int nerf_abi_version(void);

/* Development options (kernel-family selection, timing skeletons).  Defaults are read from the
 * environment ONCE per process (NERF_CHAIN_LEGACY, NERF_FWD_CYCLES, NERF_WGRAD_OVH,
 * NERF_WGRAD_DEBUG, NERF_WGRAD_ONLY, NERF_HASH_BWD_ONLY_LEVEL, NERF_STASH_FP8); afterwards they
 * change only through nerf_set_option.  Names: "chain_legacy", "fwd_cycles", "wgrad_overhead",
 * "wgrad_debug", "wgrad_only", "hash_bwd_only_level", "hash_bwd_atomic", "wgrad_atomic", "wgrad_k16", "wgrad_big_only", "stash_fp8", "infer_shape32", "infer64", "wgrad_bw_x16", "wgrad_fixed", "chain_grid", "wgrad_grid", "hash_fwd_lds_kb",
 * "hash_xcd", "composite_wgs_per_cu", "deterministic".  No hot-path launch reads the environment.
 *
 * "deterministic" (NERF_DETERMINISTIC; default 0).  The reference's gradients are plain sums (loss.backward(), run.py:1941-1944);
 * several kernels here take such sums in an order that depends on scheduling (float atomics, slots reserved with returning
 * atomics).  With the option on, every such sum has a fixed order and two runs of the same step produce the same bits:
 *   - compaction: use nerf_sample_compact_ordered (slots in sample order) instead of nerf_sample_compact*;
 *   - nerf_imlp_bwd / nerf_p4_canon_bwd / nerf_p4_deform_bwd: weight gradients through partial tiles summed in workgroup order
 *     (inside the workspace), the displacement-scale gradient through an ordered sum;
 *   - nerf_hash_encode_bwd_ws*: a bin that is cut into several work items (the coarse dense levels) is summed with 64-bit INTEGER
 *     atomics on the records' fixed-point terms in a staging array inside the workspace (integer addition commutes: any order gives
 *     the same bits) and converted by a last small launch; no float atomics anywhere.  The forms without a workspace return
 *     NERF_EINVAL;
 *   - nerf_hash_encode_bwd_input*: point on the thread, levels summed in order;
 *   - nerf_composite_mse*_bwd: pass `sum_ws` (the loss and regulariser sums; they do not enter the gradients; the engines always do).
 * Always ordered, option or not: the vanilla decoder's weight gradients, nerf_tv_normsq*.  Measured cost per step: Instant +0.085 ms
 * (0.514 -> 0.599), Part 4 +0.10 ms (0.730 -> 0.832): ordered compaction (three launches), partial tiles of the tiny-MLP weight
 * gradients, the staging pass (profiles/r04_deterministic_cost.txt). */
int nerf_set_option(const char* name, int value);
int nerf_get_option(const char* name, int* value_out);

/* ---- a1+a2: stratified depths and ray points ------------------------------
 * replaces sample_stratified (src/renderer.py:186-201) and the pts/view_dirs
 * block of render_rays (src/renderer.py:291-299).
 *   rays_o, rays_d [R,3]; u [R,S] uniform jitter or NULL (no perturbation);
 *   z_out [R,S]; pts_out / dirs_out [R*S,3] or NULL (skip materialisation).
 * z is bit-exact w.r.t. the reference's CPU path for the same `u`. */
int nerf_sample_rays(const float* rays_o, const float* rays_d, const float* u,
                     int64_t n_rays, int n_samples, float near_plane, float far_plane,
                     float* z_out, float* pts_out, float* dirs_out, nerf_stream_t stream);

/* ---- f1: batch sampling from device-resident frames ------------------------------
 * replaces the body of BlenderDataset.sample_random_rays after its three index draws
 * (src/dataset.py:150-171): pixel -> camera direction ((x - W/2)/f, -(y - H/2)/f, -1), rotation by
 * c2w[:3,:3], normalisation, origin = c2w[:3,3] * scene_scale, RGBA fetch.
 *   images [n_images,H,W,4] fp32, poses [n_images,4,4] fp32 row-major, img_idx / pix_y / pix_x [batch]
 *   int64 (the caller draws them; values must be in range) -> rays_o [batch,3], rays_d [batch,3]
 *   unit, rgba [batch,4]. */
int nerf_gather_rays(const float* images, const float* poses, const int64_t* img_idx, const int64_t* pix_y,
                     const int64_t* pix_x, int64_t batch, int n_images, int H, int W, float focal,
                     float scene_scale, float* rays_o, float* rays_d, float* rgba, nerf_stream_t stream);

/* the same with ONE index per ray, drawn uniformly over all pixels of all frames (flat_idx in
 * [0, n_images*H*W): image = idx / (H*W), row = idx / W % H, column = idx % W), and with the training
 * target formed in the same pass: target = rgb * a + bg * (1 - a) (run.py:317-322, run.py:588-594);
 * bg [3] and target [batch,3] go together (both NULL: no target); rgba [batch,4] optional. */
int nerf_gather_batch(const float* images, const float* poses, const int64_t* flat_idx, int64_t batch, int n_images,
                      int H, int W, float focal, float scene_scale, const float* bg, float* rays_o, float* rays_d,
                      float* rgba, float* target, nerf_stream_t stream);

/* ---- f1 + a1: the data side of one training step in one kernel --------------------------
 * replaces sample_random_rays (src/dataset.py:140-171: the three index draws and everything after
 * them), the target compositing (run.py:317-322) and sample_stratified with perturb (src/renderer.py:
 * 186-201, its torch.rand included): per ray one uniform draw over all pixels of all frames, per
 * sample one uniform jitter draw, both from a counter-based generator keyed by (seed, counter) --
 * same distributions as the reference's torch.randint / torch.rand, different streams; pass a new
 * `counter` (< 2^24) every step.  perturb 0: plain stratified depths.  Outputs rays_o / rays_d [batch,3],
 * z [batch,n_samples], target [batch,3] (with bg [3]) and/or rgba [batch,4]. */
int nerf_train_batch(const float* images, const float* poses, int n_images, int H, int W, float focal,
                     float scene_scale, const float* bg, uint64_t seed, uint64_t counter, int64_t batch,
                     int n_samples, float near_plane, float far_plane, int perturb, float* rays_o, float* rays_d,
                     float* rgba, float* target, float* z_out, nerf_stream_t stream);

/* the same for rays [first_ray, first_ray + batch) of a larger batch: ranks that pass the same seed and counter and
 * their shard of a global batch draw, between them, exactly the batch one rank draws with the global size (SURVEY 8(e):
 * "each rank draws its own shard from the same seeded index stream so the union equals the 1-GPU batch") */
int nerf_train_batch_shard(const float* images, const float* poses, int n_images, int H, int W, float focal,
                     float scene_scale, const float* bg, uint64_t seed, uint64_t counter, int64_t first_ray, int64_t batch,
                     int n_samples, float near_plane, float far_plane, int perturb, float* rays_o, float* rays_d,
                     float* rgba, float* target, float* z_out, nerf_stream_t stream);

/* ---- a3: occupancy lookup ---------------------------------------------------
 * replaces DensityGrid.get_active_mask (src/renderer.py:134-166).
 *   pts [N,3]; binary_grid [res,res,res] bytes (torch.bool storage);
 *   mask_out [N] bytes (0/1); idx_out [N,3] int64 voxel indices or NULL.
 * Index arithmetic is bit-exact: fp32 add, fp32 multiply by (float)(res/(2*bound)),
 * truncation toward zero. */
int nerf_active_mask(const float* pts, int64_t n, const uint8_t* binary_grid, int resolution,
                     float bound, uint8_t* mask_out, int64_t* idx_out, nerf_stream_t stream);

/* ---- a1-a4: occupancy-masked sampling with in-kernel compaction ---------------------
 * replaces the density-grid branch of render_rays (src/renderer.py:303-343): sample_stratified
 * + pts + get_active_mask + boolean-index gather (+ the zero-filled scatter, which the *_indexed
 * compositing entry points make unnecessary).
 *   z_out [R,S]; slot_of_sample [R*S] int32: row of the sample in the compact arrays, -1 = skipped;
 *   pts_compact / dirs_compact [capacity >= R*S rows, 3] (only the first *active_count rows are
 *   written; dirs are unit view directions); active_count: device u32.
 * z and the voxel test are bit-exact w.r.t. nerf_sample_rays + nerf_active_mask.  resolution <= 32768. */
int nerf_sample_compact(const float* rays_o, const float* rays_d, const float* u, int64_t n_rays,
                        int n_samples, float near_plane, float far_plane, const uint8_t* binary_grid,
                        int resolution, float bound, float* z_out, int* slot_of_sample, float* pts_compact,
                        float* dirs_compact, unsigned* active_count, nerf_stream_t stream);
/* the same with the jitter drawn in the kernel (one uniform per sample from the counter-based generator of
 * nerf_train_batch, keyed by (seed, counter); counter < 2^24): what `perturb=True` costs the reference a
 * torch.rand([R,S]) for (src/renderer.py:198).  Same distribution, not the same stream. */
int nerf_sample_compact_jitter(const float* rays_o, const float* rays_d, uint64_t seed, uint64_t counter,
                               int64_t n_rays, int n_samples, float near_plane, float far_plane,
                               const uint8_t* binary_grid, int resolution, float bound, float* z_out,
                               int* slot_of_sample, float* pts_compact, float* dirs_compact,
                               unsigned* active_count, nerf_stream_t stream);
/* rays [first_ray, first_ray + n_rays) of a larger batch: sample g of this call draws uniform (first_ray * n_samples + g)
 * of step `counter`, so data-parallel ranks that compact the shards of ONE global batch (same seed and counter, first_ray =
 * the shard's first ray) jitter exactly as one GPU would with the whole batch. */
int nerf_sample_compact_jitter_shard(const float* rays_o, const float* rays_d, uint64_t seed, uint64_t counter,
                                     int64_t first_ray, int64_t n_rays, int n_samples, float near_plane, float far_plane,
                                     const uint8_t* binary_grid, int resolution, float bound, float* z_out,
                                     int* slot_of_sample, float* pts_compact, float* dirs_compact,
                                     unsigned* active_count, nerf_stream_t stream);
/* ... as a link of a CHAIN of calls on one stream.  chain_state: NERF_COMPACT_CHAIN_WORDS device u32 words, zeroed ONCE by the caller:
 * two counters used alternately (this call counts in word `turn` (0 / 1) and its kernel clears word `turn ^ 1` for the next link: no
 * fill launch per call) and the tickets of the workgroup that finishes last, which writes the count to count_host[0] and then `seq` to
 * count_host[1] (two u32 of host-mapped pinned memory; NULL: no publication).  No copy launch and no event behind the kernel: the
 * host polls count_host[1] for its seq.  A host that queues batch k + 1 before it runs step k reads word `turn` of call k before
 * call k + 2 is queued. */
#define NERF_COMPACT_CHAIN_WORDS 40
int nerf_sample_compact_jitter_chain(const float* rays_o, const float* rays_d, uint64_t seed, uint64_t counter,
                                     int64_t first_ray, int64_t n_rays, int n_samples, float near_plane, float far_plane,
                                     const uint8_t* binary_grid, int resolution, float bound, float* z_out,
                                     int* slot_of_sample, float* pts_compact, float* dirs_compact,
                                     unsigned* chain_state, int turn, unsigned* count_host, unsigned seq, nerf_stream_t stream);

/* the same compaction with the slots in SAMPLE ORDER (option "deterministic": the single-pass kernels reserve slots with a returning
 * atomic per 4096 samples, so the order of the compact arrays -- and with it the order of every later sum over samples -- depends
 * on scheduling).  u != NULL: jitter from u [R,S]; u NULL and draw != 0: drawn as nerf_sample_compact_jitter_shard does from
 * (seed, counter, first_ray); else the plain depths.  scratch: nerf_sample_compact_ordered_scratch_bytes(n_rays, n_samples) bytes, 4-byte aligned.  Three launches. */
size_t nerf_sample_compact_ordered_scratch_bytes(int64_t n_rays, int n_samples);
int nerf_sample_compact_ordered(const float* rays_o, const float* rays_d, const float* u, int draw, uint64_t seed, uint64_t counter,
                                int64_t first_ray, int64_t n_rays, int n_samples, float near_plane, float far_plane,
                                const uint8_t* binary_grid, int resolution, float bound, float* z_out, int* slot_of_sample,
                                float* pts_compact, float* dirs_compact, unsigned* active_count, void* scratch,
                                size_t scratch_bytes, nerf_stream_t stream);

/* ---- hierarchical (inverse-CDF) fine sampling, opt-in extension --------------------
 * No reference counterpart (the reference has one stratified pass only); follows Mildenhall et al.
 * 2020 sec. 5.2 as restated in oracle/nerf_oracle.py::sample_pdf (parity unpinned).
 *   z_coarse [R,S] sorted depths, weights [R,S] compositing weights of the coarse pass,
 *   u [R,n_fine] uniform draws or NULL (deterministic linspace(0,1,n_fine));
 *   z_out [R, S + n_fine]: coarse and fine depths merged and sorted. */
int nerf_sample_pdf(const float* z_coarse, const float* weights, const float* u, int64_t n_rays,
                    int n_coarse, int n_fine, float* z_out, nerf_stream_t stream);

/* ---- a12: occupancy-grid refresh -----------------------------------------------
 * replaces the lattice construction and the grid / binary_grid update of
 * DensityGrid.update (src/renderer.py:49-54, 118-132); the sigma query in between goes through
 * the field kernels.
 *   nerf_grid_lattice: pts_out [res^3,3], nodes of linspace(-bound, bound, res), 'ij' order.
 *   nerf_grid_update : sigma [n_cells] freshly queried; grid [n_cells] in/out (overwritten, or
 *                      max(grid*decay, sigma) when dynamic != 0); binary_grid [n_cells] bytes =
 *                      grid > threshold; *active_count (device u64) = number of set cells. */
int nerf_grid_lattice(float bound, int resolution, float* pts_out, nerf_stream_t stream);
int nerf_grid_update(const float* sigma, float* grid, uint8_t* binary_grid, int64_t n_cells,
                     float decay, int dynamic, float threshold, unsigned long long* active_count,
                     nerf_stream_t stream);

/* ---- a5: Fourier features ---------------------------------------------------
 * replaces FourierRepresentation.forward (src/embeddings.py:22-32).
 *   x [N,dim] -> out [N, dim + 2*dim*n_freq] laid out [x | sin f0 | cos f0 | sin f1 | ...]. */
int nerf_fourier_encode(const float* x, int64_t n, int dim, int n_freq, float* out,
                        nerf_stream_t stream);

/* ---- a9: alpha compositing (wave-level transmittance scan) -----------------
 * replaces volume_render (src/renderer.py:204-237) and the second weights pass
 * for mean_delta_x (src/renderer.py:367-380).
 *   rgb [R,S,3], sigma [R,S], z [R,S], rays_d [R,3];
 *   bg: NULL, [3] (bg_rows = 1) or [R,3] (bg_rows = R);
 *   extra [R,S,3] or NULL -> extra_map [R,3] = sum_s w * extra  (delta_x mean);
 *   out_rgb [R,3], out_depth [R], out_acc [R]; weights_out [R,S] or NULL. */
int nerf_composite_fwd(const float* rgb, const float* sigma, const float* z, const float* rays_d,
                       const float* bg, int64_t bg_rows, const float* extra,
                       int64_t n_rays, int n_samples,
                       float* out_rgb, float* out_depth, float* out_acc,
                       float* extra_map, float* weights_out, nerf_stream_t stream);

/* backward of the above w.r.t. rgb, sigma (and extra when given).
 *   g_rgb [R,3], g_depth [R] or NULL, g_acc [R] or NULL, g_extra [R,3] or NULL;
 *   d_rgb [R,S,3], d_sigma [R,S], d_extra [R,S,3] or NULL. */
int nerf_composite_bwd(const float* rgb, const float* sigma, const float* z, const float* rays_d,
                       const float* bg, int64_t bg_rows, const float* extra,
                       const float* g_rgb, const float* g_depth, const float* g_acc,
                       const float* g_extra, int64_t n_rays, int n_samples,
                       float* d_rgb, float* d_sigma, float* d_extra, nerf_stream_t stream);

/* compositing straight from compact field outputs: sample (r,s) reads row slot_of_sample[r*S+s] of
 * rgb_compact [n_active,3] / sigma_compact [n_active]; skipped samples count as sigma = 0, rgb = 0
 * (src/renderer.py:328-333).  The backward writes d_rgb_compact / d_sigma_compact rows only. */
int nerf_composite_fwd_indexed(const float* rgb_compact, const float* sigma_compact, const int* slot_of_sample,
                               const float* z, const float* rays_d, const float* bg, int64_t bg_rows,
                               int64_t n_rays, int n_samples, float* out_rgb, float* out_depth,
                               float* out_acc, nerf_stream_t stream);
int nerf_composite_bwd_indexed(const float* rgb_compact, const float* sigma_compact, const int* slot_of_sample,
                               const float* z, const float* rays_d, const float* bg, int64_t bg_rows,
                               const float* g_rgb, const float* g_depth, const float* g_acc, int64_t n_rays,
                               int n_samples, float* d_rgb_compact, float* d_sigma_compact,
                               nerf_stream_t stream);

/* ---- a6: fused Fourier-encode + 8x256 density/colour decoder (bf16 MFMA) ----
 * replaces NeuralField.forward for mode part2_nerf (src/core.py:354-359) =
 * FourierRepresentation x2 (src/embeddings.py:22-32) + NeRFDecoder.forward
 * (src/decoders.py:68-87).  Architecture is fixed to the reference defaults the
 * kernels are specialised for: L_embed 10, L_embed_dir 4, hidden 256, 8 layers,
 * skip at 4, view_dim 128 (configs/part2.yaml.example); anything else returns
 * NERF_ENOSYS.
 *
 * Weights live in ONE flat fp32 buffer in reference state_dict order
 * (decoder.pts_layers.{0..7}.{weight,bias}, sigma_layer, feature_layer,
 * view_layer, rgb_layer; nn.Linear [out,in] row-major): 595,844 floats.
 * nerf_mlp_pack() converts it into the MFMA-fragment-ordered bf16 streams the
 * kernels read (forward stream, transposed stream for dgrad, fp32 biases).    */
#define NERF_MLP_PARAM_COUNT 595844

/* sizes (bytes) of the packed buffers the caller must allocate */
size_t nerf_mlp_packed_bytes(void);
/* params_f32 [595844] -> packed (nerf_mlp_packed_bytes() bytes, 256-B aligned) */
int nerf_mlp_pack(const float* params_f32, void* packed, nerf_stream_t stream);
/* the same for a subset of the streams: which = 1 the training streams (forward + transposed, 32x32x16
 * fragments) and the bias table; 2 the inference stream (16x16x32 fragments); 3 both (= nerf_mlp_pack).
 * A training loop repacks 1 after every optimiser step and 2 only before it renders. */
int nerf_mlp_pack_streams(const float* params_f32, void* packed, int which, nerf_stream_t stream);

/* Inputs, one of:
 *   ray mode   : rays_o/rays_d [R,3] + z [R,S]  (n = R*S samples, sample i -> ray i / S)
 *   point mode : pts/dirs [n,3] (rays_o = pts, rays_d = dirs, z = NULL, n_samples = 0);
 *                dirs are used as given (NeuralField.forward takes unit view dirs).
 * Outputs rgb [n,3], sigma [n] fp32.
 * stash: NULL for inference; for training a workspace of nerf_mlp_stash_bytes(n)
 * that nerf_mlp_bwd consumes: an image of every layer input + relu bitmasks.  Images are bf16 (5.1 KB + 0.3 KB
 * of mask words per sample): every MFMA of a training step contracts bf16 operands.  Option stash_fp8 = 1 makes
 * the asm-stream kernels write 8-bit images instead (e4m3, 2.5 KB per sample; weight gradients then come from
 * e4m3 x e5m2 operands -- narrower than the reference's precision, opt-in).  Forward and backward of one
 * step must run under the same options. */
size_t nerf_mlp_stash_bytes(int64_t n);
int nerf_mlp_fwd(const void* packed, const float* rays_o, const float* rays_d, const float* z,
                 int64_t n, int n_samples, float* rgb, float* sigma, void* stash,
                 nerf_stream_t stream);

/* The decoder as a stand-alone operator on ALREADY ENCODED inputs -- BaseDecoder.forward(x_enc, d_enc) of
 * src/decoders.py:68-87: x_enc [n,63] = FourierRepresentation(L=10) of the positions, d_enc [n,27] =
 * FourierRepresentation(L=4) of the view directions, fp32 row-major.  stash as in nerf_mlp_fwd. */
int nerf_mlp_fwd_encoded(const void* packed, const float* x_enc, const float* d_enc, int64_t n, float* rgb,
                         float* sigma, void* stash, nerf_stream_t stream);

/* Backward of nerf_mlp_fwd (autograd of src/decoders.py:68-87; the loss.backward() of
 * run.py:337 for the decoder).  rgb/sigma are the forward outputs, d_rgb [n,3] / d_sigma [n]
 * the upstream gradients; grads_f32 [595844] (same layout as the parameter vector) is
 * OVERWRITTEN.  Two kernels: a dgrad chain (transposed weight stream; pre-activation gradients kept
 * in `workspace` as bf16 images -- with option stash_fp8 as e5m2 images divided by a power of two derived
 * from the launch's largest output-layer derivative) and a split-K weight-gradient
 * pass over the stash (v_mfma_f32_32x32x16_bf16; stash_fp8: v_mfma_scale_f32_32x32x64_f8f6f4).
 * workspace: nerf_mlp_bwd_workspace_bytes(n), 256-byte aligned. */
size_t nerf_mlp_bwd_workspace_bytes(int64_t n);
int nerf_mlp_bwd(const void* packed, const void* stash, const float* rgb, const float* sigma,
                 const float* d_rgb, const float* d_sigma, int64_t n, float* grads_f32,
                 void* workspace, nerf_stream_t stream);
/* the two halves of nerf_mlp_bwd, separately launchable (profiling, overlap with comms) */
int nerf_mlp_bwd_dgrad(const void* packed, const void* stash, const float* rgb, const float* sigma,
                       const float* d_rgb, const float* d_sigma, int64_t n, void* workspace,
                       nerf_stream_t stream);
/* nerf_mlp_bwd_dgrad with the largest output-layer derivative supplied by the caller (one device fp32:
 * max over samples of |d_rgb rgb (1 - rgb)| and of |d_sigma| where sigma > 0, e.g. from
 * nerf_composite_mse_bwd) instead of a pass of its own; amax_dev NULL = nerf_mlp_bwd_dgrad. */
int nerf_mlp_bwd_dgrad_ex(const void* packed, const void* stash, const float* rgb, const float* sigma,
                          const float* d_rgb, const float* d_sigma, int64_t n, void* workspace,
                          const float* amax_dev, nerf_stream_t stream);
/* nerf_mlp_bwd_wgrad_part: the weight-gradient pass in two launches for a data-parallel caller that
 * all-reduces one parameter range while the other is still being computed.  part 1 writes
 * grads_f32[split, 595844) (pts_layers.4 .. rgb_layer), part 2 writes grads_f32[0, split)
 * (pts_layers.0 .. 3), split = nerf_mlp_wgrad_part_split(); part 0 = nerf_mlp_bwd_wgrad.  Each
 * part overwrites exactly its own range. */
int64_t nerf_mlp_wgrad_part_split(void);
int nerf_mlp_bwd_wgrad_part(const void* stash, const void* workspace, int64_t n, float* grads_f32, int part,
                            nerf_stream_t stream);
int nerf_mlp_bwd_wgrad(const void* stash, const void* workspace, int64_t n, float* grads_f32,
                       nerf_stream_t stream);

/* ---- a8: multiresolution hash grid ------------------------------------------------
 * replaces tinycudann's Encoding("HashGrid") behind HashRepresentation.forward
 * (src/embeddings.py:60-89), incl. its (x + bound) / (2 bound) clamp pre-step.  The library is
 * third party, unpinned and absent: the level table is this build's definition (see
 * oracle/nerf_oracle.py::hash_grid_levels) and is passed in by the host, one entry per level:
 *   scale (fp32), res, size (entries), offset (first entry), dense (1: x + y*res + z*res^2,
 *   0: xor-prime spatial hash), all HOST arrays of n_levels (<= 16) elements.
 * table: fp32 [entries, 2] (the flat `encoding.params` vector).
 * fwd: pts [n,3] world coordinates -> out_f32 [n, 2*n_levels] and/or out_nat_bf16, the bf16
 *      operand image nerf_imlp_fwd consumes (n rounded up to 128 rows); idx_out [n, n_levels, 8]
 *      absolute entry indices or NULL.
 * bwd: d_feat [n, 2*n_levels] -> d_table [entries, 2] += trilinear scatter (float atomics;
 *      the caller zeroes d_table). */
int nerf_hash_encode_fwd(const float* pts, int64_t n, const float* table, int n_levels,
                         const float* scale_host, const unsigned* res_host, const unsigned* size_host,
                         const unsigned* offset_host, const unsigned* dense_host, float bound,
                         float* out_f32, void* out_nat_bf16, unsigned* idx_out, nerf_stream_t stream);
/* the forward from an fp16 copy of the table ([entries, 2] halves; kept current by nerf_adamw_clip_step_shadow or
 * made with nerf_f32_to_f16): half the bytes per gather.  tinycudann evaluates its grid from fp16 parameters next
 * to the fp32 master copy the optimiser updates (reference src/embeddings.py:57-73 leaves its default). */
int nerf_hash_encode_fwd_f16(const float* pts, int64_t n, const void* table_f16, int n_levels,
                         const float* scale_host, const unsigned* res_host, const unsigned* size_host,
                         const unsigned* offset_host, const unsigned* dense_host, float bound,
                         float* out_f32, void* out_nat_bf16, nerf_stream_t stream);
/* operand image only, from either table, as bf16 (nat_dtype 0: what nerf_imlp_fwd reads) or fp16 (nat_dtype 1: what the
 * Part 4 forward chains read, nerf_p4_deform_fwd / nerf_p4_canon_fwd); exactly one of table_f32 / table_f16 is given */
int nerf_hash_encode_fwd_nat(const float* pts, int64_t n, const float* table_f32, const void* table_f16, int n_levels,
                             const float* scale_host, const unsigned* res_host, const unsigned* size_host,
                             const unsigned* offset_host, const unsigned* dense_host, float bound,
                             void* out_nat, int nat_dtype, nerf_stream_t stream);
/* n_tables grids of ONE level structure evaluated at the same points in one launch (Part 4's three deformation grids):
 * table g starts table_stride entries (of two fp16) after table g - 1, its operand image out_stride_bytes after the previous one */
int nerf_hash_encode_fwd_nat_tables(const float* pts, int64_t n, const void* tables_f16, int n_tables, int64_t table_stride,
                                    int n_levels, const float* scale_host, const unsigned* res_host, const unsigned* size_host,
                                    const unsigned* offset_host, const unsigned* dense_host, float bound, void* out_nat,
                                    int64_t out_stride_bytes, int nat_dtype, nerf_stream_t stream);
int nerf_f32_to_f16(const float* src, void* dst_f16, int64_t n, nerf_stream_t stream);
int nerf_hash_encode_bwd(const float* pts, int64_t n, int n_levels, const float* scale_host,
                         const unsigned* res_host, const unsigned* size_host, const unsigned* offset_host,
                         const unsigned* dense_host, float bound, const float* d_feat, float* d_table,
                         nerf_stream_t stream);
/* the same for levels [first_level, end_level) only: a data-parallel caller all-reduces one level
 * range of the table gradient (a contiguous slice of d_table) while the next range is computed. */
int nerf_hash_encode_bwd_levels(const float* pts, int64_t n, int n_levels, const float* scale_host,
                         const unsigned* res_host, const unsigned* size_host, const unsigned* offset_host,
                         const unsigned* dense_host, float bound, const float* d_feat, float* d_table, int first_level, int end_level,
                         nerf_stream_t stream);
/* the same with a workspace (nerf_hash_encode_bwd_workspace_bytes(n, n_levels), 256-B aligned; NULL = the
 * atomic form above): the levels whose table exceeds the LDS pass are scattered as a partial SORT --
 * contributions binned by 4096-entry table slice (8-byte records in the workspace), each slice summed in LDS
 * by the workgroup that owns it and added to d_table with plain coalesced read-modify-writes.  No global
 * float atomics on the hashed levels (they retire per line request, ~20 G/s, wherever they land).  d_table
 * must not be updated by another stream during the call.  Same sums as the atomic form up to fp32 order. */
size_t nerf_hash_encode_bwd_workspace_bytes(int64_t n, int n_levels);
int nerf_hash_encode_bwd_ws(const float* pts, int64_t n, int n_levels, const float* scale_host,
                         const unsigned* res_host, const unsigned* size_host, const unsigned* offset_host,
                         const unsigned* dense_host, float bound, const float* d_feat, float* d_table, int first_level, int end_level,
                         void* workspace, size_t workspace_bytes, nerf_stream_t stream);
/* the OVERWRITE form of the same pass: the table gradient of levels [first_level, end_level) is STORED, not accumulated --
 * the caller neither zeroes d_table nor is it read back (the 52 MB memset and the 52 MB read of a 13 M-entry table per
 * step).  Every slice of those levels is written exactly once: by the one work item that owns it, or -- slices cut into
 * several items, the coarse dense levels -- zeroed by the scatter launch and then added to with atomics. */
int nerf_hash_encode_bwd_ws_store(const float* pts, int64_t n, int n_levels, const float* scale_host,
                                  const unsigned* res_host, const unsigned* size_host, const unsigned* offset_host,
                                  const unsigned* dense_host, float bound, const float* d_feat, float* d_table,
                                  int first_level, int end_level, void* workspace, size_t workspace_bytes,
                                  nerf_stream_t stream);
/* The overwrite form for n_tables tables of ONE level structure, scattered to from the same points, in one pass of launches
 * (Part 4's three deformation grids): table t starts table_stride ENTRIES after table t - 1 in d_table, its feature gradients
 * [n, 2 n_levels] dfeat_stride FLOATS after the previous table's in d_feat; n_levels * n_tables <= 48;
 * workspace >= nerf_hash_encode_bwd_tables_workspace_bytes(n, n_levels, n_tables). */
size_t nerf_hash_encode_bwd_tables_workspace_bytes(int64_t n, int n_levels, int n_tables);
int nerf_hash_encode_bwd_ws_store_tables(const float* pts, int64_t n, int n_tables, int64_t table_stride, int n_levels,
                                         const float* scale_host, const unsigned* res_host, const unsigned* size_host,
                                         const unsigned* offset_host, const unsigned* dense_host, float bound,
                                         const float* d_feat, int64_t dfeat_stride, float* d_table, void* workspace,
                                         size_t workspace_bytes, nerf_stream_t stream);
/* ... its speculative form (protocol of nerf_hash_encode_bwd_ws_store_spec below: no count pass, capacities from the true counts the
 * previous call on this workspace left; the producer -- nerf_p4_deform_bwd -- max-accumulates the largest |d feature| into the slot
 * nerf_hash_encode_bwd_ws_slots(workspace, n, n_levels * n_tables, ...) names; d_feat row-major, or NULL: the producer wrote the
 * level-major copy into the workspace's grad_lm slot) */
int nerf_hash_encode_bwd_ws_store_tables_spec(const float* pts, int64_t n, int n_tables, int64_t table_stride, int n_levels,
                                              const float* scale_host, const unsigned* res_host, const unsigned* size_host,
                                              const unsigned* offset_host, const unsigned* dense_host, float bound,
                                              const float* d_feat, int64_t dfeat_stride, float* d_table, void* workspace,
                                              size_t workspace_bytes, void* status_host, nerf_stream_t stream);
/* The count pass of the binned backward done where the information already exists (Instant-NGP step, all levels):
 *   nerf_hash_encode_fwd_f16_hist  the forward also counts the corners per (level, table slice) into `bwd_workspace`
 *                                  (>= nerf_hash_encode_bwd_workspace_bytes(n, L); its header and counts are zeroed here);
 *   nerf_imlp_bwd_lm               the decoder's backward writes d features level-major and their largest magnitude straight
 *                                  into that workspace (slots from nerf_hash_encode_bwd_ws_slots) instead of d_feat [n,32];
 *   nerf_hash_encode_bwd_ws_store_precounted   plan + scatter + reduce only, overwrite form.
 * The three calls must see the same points, level table and workspace, in this order, on one stream. */
int nerf_hash_encode_fwd_f16_hist(const float* pts, int64_t n, const void* table_f16, int n_levels,
                                  const float* scale_host, const unsigned* res_host, const unsigned* size_host,
                                  const unsigned* offset_host, const unsigned* dense_host, float bound,
                                  void* out_nat_bf16, void* bwd_workspace, size_t bwd_workspace_bytes, nerf_stream_t stream);
/* (amax_bits_out: NERF_AMAX_WORDS u32 words; a producer max-accumulates the fp32 bits of its largest |d feature| into ANY of them --
 * its workgroups spread over all, same-address atomics retire one after the other -- the consumer takes the maximum of all and
 * clears them; grad_lm_out: float2 [n_levels][n], level-major) */
#define NERF_AMAX_WORDS 32
int nerf_hash_encode_bwd_ws_slots(void* workspace, int64_t n, int n_levels, void** amax_bits_out, void** grad_lm_out);
int nerf_hash_encode_bwd_ws_store_precounted(const float* pts, int64_t n, int n_levels, const float* scale_host,
                                             const unsigned* res_host, const unsigned* size_host,
                                             const unsigned* offset_host, const unsigned* dense_host, float bound,
                                             float* d_table, void* workspace, size_t workspace_bytes, nerf_stream_t stream);
/* The speculative form of the binned scatter: NO count pass.  Bin capacities come from the TRUE record counts that the previous
 * all-level call on this workspace (counted, precounted or speculative) left in it, + 1/8 + 64 records; a record that does not fit
 * its bin is added with a float atomic by a last small launch (steady-state training batches fill their bins within a few percent
 * from step to step: a handful of records).  Protocol: nerf_hash_encode_bwd_spec_begin (clears the header; needed only when the
 * PREVIOUS call on the workspace was a counted one -- a speculative call's last launch leaves the header clean) -> the decoder's
 * backward max-accumulates the largest |gradient| into the workspace's slot (nerf_hash_encode_bwd_ws_slots): nerf_imlp_bwd_amax
 * with row-major d_feat [n, 2L], or nerf_imlp_bwd_lm with the level-major copy in the workspace (then d_feat NULL here)
 * -> nerf_hash_encode_bwd_ws_store_spec.  The caller decides WHEN the estimates are good (same
 * occupancy grid, point count within ~10 % of the last call's) and reads the status block afterwards: 8 x u32, published by the
 * call's last launch at nerf_hash_encode_bwd_spec_status(workspace) (device) and, when status_host is given, in that host-mapped
 * pinned block as well (no copy launch; word [7] is written LAST, as 1, after a system-scope fence: a caller that cleared it before
 * the call may poll it instead of recording an event) -- [3] records that
 * overflowed, [4] != 0: records were LOST (overflow list full or estimates larger than the workspace): the gradient of that call is
 * incomplete, fall back to the counted form.  Levels of at most 256 slices (tables up to 2^20 entries per level); not with option "deterministic". */
int nerf_hash_encode_bwd_spec_begin(void* workspace, nerf_stream_t stream);
int nerf_hash_encode_bwd_ws_store_spec(const float* pts, int64_t n, int n_levels, const float* scale_host,
                                       const unsigned* res_host, const unsigned* size_host, const unsigned* offset_host,
                                       const unsigned* dense_host, float bound, const float* d_feat, float* d_table, void* workspace,
                                       size_t workspace_bytes, void* status_host, nerf_stream_t stream);
int nerf_imlp_bwd_amax(const void* packed, void* workspace, const float* rgb, const float* sigma, const float* d_rgb,
                       const float* d_sigma, int64_t n, float* grads_f32, float* d_feat, void* amax_bits, nerf_stream_t stream);
const void* nerf_hash_encode_bwd_spec_status(const void* workspace);

/* gradient with respect to the encoded positions (dynamic fields encode x + delta_x: reference
 * src/core.py:268-271, 341-344): d_pts [n,3] = d_feat . d features / d x, zero along an axis on which
 * HashRepresentation's clamp is active; d_pts is OVERWRITTEN. */
int nerf_hash_encode_bwd_input(const float* pts, int64_t n, const float* table, int n_levels,
                               const float* scale_host, const unsigned* res_host, const unsigned* size_host,
                               const unsigned* offset_host, const unsigned* dense_host, float bound,
                               const float* d_feat, float* d_pts, nerf_stream_t stream);
/* the same gradient from the fp16 copy of the table the forward evaluated (nerf_adamw_clip_step_shadow keeps it): the
 * derivative of the function the forward computed, at half the bytes per gather */
int nerf_hash_encode_bwd_input_f16(const float* pts, int64_t n, const void* table_f16, int n_levels,
                                   const float* scale_host, const unsigned* res_host, const unsigned* size_host,
                                   const unsigned* offset_host, const unsigned* dense_host, float bound,
                                   const float* d_feat, float* d_pts, nerf_stream_t stream);
/* ... ADDED to d_pts instead of overwriting it: a gradient that reaches the positions by another path as well (Part 4: the
 * displacement regulariser's, reference run.py:1852-1858) is already there -- no zeroing launch, no separate add */
int nerf_hash_encode_bwd_input_f16_accum(const float* pts, int64_t n, const void* table_f16, int n_levels,
                                         const float* scale_host, const unsigned* res_host, const unsigned* size_host,
                                         const unsigned* offset_host, const unsigned* dense_host, float bound,
                                         const float* d_feat, float* d_pts, nerf_stream_t stream);
/* ... from LEVEL-MAJOR feature gradients float2 [n_levels][n] (the layout a producer leaves in the hash backward's workspace,
 * nerf_hash_encode_bwd_ws_slots: this launch walks one level per workgroup, so its reads are coalesced); accumulate != 0: added to
 * d_pts.  Not with option "deterministic". */
int nerf_hash_encode_bwd_input_lm_f16(const float* pts, int64_t n, const void* table_f16, int n_levels,
                                      const float* scale_host, const unsigned* res_host, const unsigned* size_host,
                                      const unsigned* offset_host, const unsigned* dense_host, float bound,
                                      const void* grad_lm, float* d_pts, int accumulate, nerf_stream_t stream);


/* ---- a7: Instant decoder (two bias-free tiny MLPs, bf16 MFMA) -----------------------
 * replaces the two tinycudann FullyFusedMLP networks of InstantNeRFDecoder
 * (src/decoders.py:111-134, forward 136-162).  Parameter vector (fp32, [out,in] row-major):
 *   sigma_net W1 [64,32] | W2 [16,64] | color_net W1 [64,48] | W2 [64,64] | W3 [16,64] = 11264
 * (input 16 geometry channels + 27 direction-code channels, padded to 48; 3 rgb rows of 16).
 * workspace (nerf_imlp_workspace_bytes(n), 256-B aligned) starts with the hash operand image
 * (offset nerf_imlp_hash_operand_offset(n) = 0) and holds the training stash.
 * fwd: dirs [n,3] unit view directions -> rgb [n,3], sigma [n] = softplus(h0 - 5).
 * bwd: grads_f32 [11264] OVERWRITTEN; d_feat [n,32] = gradient w.r.t. the hash features. */
size_t nerf_imlp_packed_bytes(void);
size_t nerf_imlp_workspace_bytes(int64_t n);
size_t nerf_imlp_hash_operand_offset(int64_t n);
int nerf_imlp_pack(const float* params_f32, void* packed, nerf_stream_t stream);
int nerf_imlp_fwd(const void* packed, void* workspace, const float* dirs, int64_t n, float* rgb,
                  float* sigma, int train, nerf_stream_t stream);
/* InstantNeRFDecoder.forward(x_enc, d_enc) (src/decoders.py:136-162) on already encoded inputs: x_enc [n,32]
 * hash features, d_enc [n,27] direction codes, fp32 row-major; same workspace and backward
 * (nerf_imlp_bwd: d_feat is then the gradient with respect to x_enc). */
int nerf_imlp_fwd_encoded(const void* packed, void* workspace, const float* x_enc, const float* d_enc, int64_t n,
                          float* rgb, float* sigma, int train, nerf_stream_t stream);
int nerf_imlp_bwd(const void* packed, void* workspace, const float* rgb, const float* sigma,
                  const float* d_rgb, const float* d_sigma, int64_t n, float* grads_f32,
                  float* d_feat, nerf_stream_t stream);
/* nerf_imlp_bwd with the feature gradients written level-major [16][n] float2 (grad_lm) and their largest magnitude
 * max-accumulated into amax_bits (NERF_AMAX_WORDS words of fp32 bits): the inputs of nerf_hash_encode_bwd_ws_store_precounted */
int nerf_imlp_bwd_lm(const void* packed, void* workspace, const float* rgb, const float* sigma,
                     const float* d_rgb, const float* d_sigma, int64_t n, float* grads_f32,
                     void* grad_lm, void* amax_bits, nerf_stream_t stream);

/* ---- a10 + a11: evaluation of the vanilla field as one launch chain ----------------------------
 * replaces render_rays(perturb=False) / render_image's chunk loop (src/renderer.py:240-384, 387-418) for
 * mode part2_nerf: stratified depths -> fused Fourier + decoder -> compositing for every chunk of
 * `chunk_rays` rays, on the caller's stream, in ONE caller-provided workspace of
 * nerf_render_rays_workspace_bytes(chunk_rays, n_samples) bytes (256-byte aligned): no allocation and no
 * host synchronisation, so the whole frame can be captured in a hipGraph.  bg as in nerf_composite_fwd.
 * n_samples <= 1024. */
size_t nerf_render_rays_workspace_bytes(int64_t chunk_rays, int n_samples);
int nerf_render_rays_fwd(const void* packed, const float* rays_o, const float* rays_d, int64_t n_rays,
                         int n_samples, float near_plane, float far_plane, const float* bg, int64_t bg_rows,
                         int64_t chunk_rays, void* workspace, float* out_rgb, float* out_depth, float* out_acc,
                         nerf_stream_t stream);

/* ---- a9 + a14: compositing fused with the MSE loss and its backward --------------------------
 * replaces, for one training step, volume_render (src/renderer.py:204-237), nn.MSELoss
 * (run.py:308,334; run.py:598) and the compositing part of loss.backward() (run.py:337,619):
 *   pixel = composite(rgb, sigma, z) (+ background), loss += loss_weight * sum (pixel - target)^2,
 *   d_rgb / d_sigma = gradients of loss_weight * sum (pixel - target)^2.
 * loss_weight = 1 / (3 n_rays) gives the reference's mean.  slot_of_sample NULL: dense [R,S] inputs;
 * otherwise compact inputs as in nerf_composite_fwd_indexed.  loss_accum (device fp32) is ADDED to;
 * amax_accum (optional device fp32, max-accumulated; caller zeroes both) receives the vanilla decoder's
 * largest output-layer derivative for nerf_mlp_bwd_dgrad_ex.  pred_out [R,3] optional.
 * sum_ws: NULL (float atomics: two or three same-address atomics per workgroup, which retire one after the other) or
 * NERF_SUM_WS_FLOATS floats of scratch: every workgroup stores its partial sums there and a one-workgroup launch adds them in
 * workgroup order (no atomics: faster, and the same bits every run). */
#define NERF_SUM_WS_FLOATS 12288
int nerf_composite_mse_bwd(const float* rgb, const float* sigma, const int* slot_of_sample, const float* z,
                           const float* rays_d, const float* bg, int64_t bg_rows, const float* target,
                           float loss_weight, int64_t n_rays, int n_samples, float* pred_out, float* loss_accum,
                           float* d_rgb, float* d_sigma, float* amax_accum, float* sum_ws, nerf_stream_t stream);
/* ... with the displacement regulariser of the dynamic fields (Part 3 / 4): `extra` [n,3] (compact, like rgb) is
 * composited with the same weights into extra_map [R,3] (optional) = render_rays' extras['mean_delta_x']
 * (src/renderer.py:363-380), reg_accum += reg_weight * sum_rays |mean_delta_x|^2 (run.py:1838 with reg_weight =
 * deformation_reg_weight / (3 n_rays)), and its gradient is added: d_extra [n,3] and, through the weights, d_sigma. */
int nerf_composite_mse_reg_bwd(const float* rgb, const float* sigma, const int* slot_of_sample, const float* z,
                               const float* rays_d, const float* bg, int64_t bg_rows, const float* target,
                               float loss_weight, const float* extra, float reg_weight, int64_t n_rays, int n_samples,
                               float* pred_out, float* extra_map, float* loss_accum, float* reg_accum, float* d_rgb,
                               float* d_sigma, float* d_extra, float* sum_ws, nerf_stream_t stream);

/* ---- a14: optimiser ---------------------------------------------------------
 * replaces torch.optim.Adam / AdamW .step() (run.py:307,338; run.py:546,629) for
 * one flat fp32 parameter vector.  step counts from 1.  weight_decay is the
 * decoupled (AdamW) form; 0 gives plain Adam.  grad_scale multiplies the
 * gradient first (used for clip_grad_norm_, run.py:624-627, and 1/world). */
int nerf_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                   int64_t n, int step, float lr, float beta1, float beta2, float eps,
                   float weight_decay, const float* grad_scale_dev, nerf_stream_t stream);

/* Fused regulariser + global-norm clip + AdamW for one flat parameter group:
 * replaces, per group, the TV-L1 term on `representation.encoding.params` and its backward
 * (run.py:611-618), clip_grad_norm_(group, max_norm) (run.py:624-627) and AdamW.step() (run.py:629).
 *   nerf_tv_normsq      : grads = grads * grad_scale + tv_weight * d/dp mean|p[1:] - p[:-1]|  (tv_weight 0:
 *                         no TV term; grad_scale = 1/world after a summing all-reduce: it scales the data
 *                         gradient only, the regulariser is a function of the replicated parameters),
 *                         then normsq_dev[0] = sum(grads^2).  normsq_dev: NERF_NORMSQ_WS_FLOATS device floats -- [0] the
 *                         squared norm, [1..34) tickets (zero before the first call ever; every call leaves them zero), then one
 *                         partial per workgroup: the partials are added in workgroup
 *                         order by the last workgroup to finish, so every run and every data-parallel replica (identical
 *                         gradients after the all-reduce) gets the same bits, hence the same clip coefficient
 *   nerf_adamw_clip_step: AdamW with grads scaled by grad_scale * min(1, max_norm / (norm + 1e-6)),
 *                         norm = grad_scale * sqrt(normsq_dev[0]); normsq_dev NULL or max_norm <= 0: no clip. */
#define NERF_NORMSQ_WS_FLOATS 4136
int nerf_tv_normsq(const float* params, float* grads, int64_t n, float tv_weight, float grad_scale,
                   float* normsq_dev, nerf_stream_t stream);
/* the same pass WITHOUT zeroing normsq_dev[0] first: several parameter groups accumulate ONE global squared norm
 * (torch.nn.utils.clip_grad_norm_(model.parameters()) of the Part 3 / 4 loops, run.py:1172, 1943) in call order; the caller
 * zeroes normsq_dev[0] and [1] once per step and hands the workspace to every group's nerf_adamw_clip_step */
int nerf_tv_normsq_accum(const float* params, float* grads, int64_t n, float tv_weight, float grad_scale,
                         float* normsq_dev, nerf_stream_t stream);
/* the same pass over 1..4 equally long tables stored back to back (n elements in all; Part 4's three deformation grids in one
 * launch): every table has its own total variation (tv_weight / (n / n_tables - 1) per neighbour pair, no pair across a seam) */
int nerf_tv_normsq_accum_tables(const float* params, float* grads, int64_t n, int n_tables, float tv_weight, float grad_scale,
                                float* normsq_dev, nerf_stream_t stream);
int nerf_adamw_clip_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                         int step, float lr, float beta1, float beta2, float eps, float weight_decay,
                         const float* normsq_dev, float max_norm, float grad_scale, nerf_stream_t stream);
/* the same step, also writing params_f16_out[n] = fp16(updated params) for nerf_hash_encode_fwd_f16 */
int nerf_adamw_clip_step_shadow(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                         int step, float lr, float beta1, float beta2, float eps, float weight_decay,
                         const float* normsq_dev, float max_norm, float grad_scale, void* params_f16_out, nerf_stream_t stream);

/* The same two passes WITHOUT rewriting the gradient (38.5 instead of 42 bytes per parameter; what the engines call):
 *   nerf_tv_normsq_codes   : normsq_dev[0] (accumulate != 0: +)= sum (grads * grad_scale + TV term)^2 over n_tables equally long tables
 *                            stored back to back -- the first call of a step STORES (accumulate 0: no zeroing launch), the
 *                            groups that share the norm add (accumulate 1); normsq_dev[1..34) (the ordered sum's tickets) are zero before
 *                            the first call ever and left zero by every call; grads is NOT modified; tv_codes
 *                            (nerf_tv_codes_bytes(n) bytes; may be NULL when tv_weight == 0) receives the two-bit signs
 *                            1 + sign(p[i+1] - p[i]) the TV term is made of (0 across a table seam)
 *   nerf_adamw_clip_step_tv: AdamW with (grads * grad_scale + TV term rebuilt from tv_codes) * min(1, max_norm / (sqrt(normsq_dev[0])
 *                            + 1e-6)).  Elements [0, tv_split) carry TV weight tv_weight_lo over tables of seg_lo elements, the rest
 *                            tv_weight_hi / seg_hi (Part 4: the three deformation grids | the canonical grid in ONE launch);
 *                            elements from lr_split on (0: none) step with lr_hi (networks | displacement_scale);
 *                            params_f16_out optional.  tv_codes NULL: no TV term. */
size_t nerf_tv_codes_bytes(int64_t n);
int nerf_tv_normsq_codes(const float* params, const float* grads, int64_t n, int n_tables, float tv_weight, float grad_scale,
                         float* normsq_dev, int accumulate, void* tv_codes, nerf_stream_t stream);
int nerf_adamw_clip_step_tv(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int step, float lr,
                            float beta1, float beta2, float eps, float weight_decay, const float* normsq_dev, float max_norm,
                            float grad_scale, const void* tv_codes, int64_t tv_split, float tv_weight_lo, int64_t seg_lo,
                            float tv_weight_hi, int64_t seg_hi, int64_t lr_split, float lr_hi, void* params_f16_out,
                            nerf_stream_t stream);

/* ... and on a PIECE [params, params + n) of ONE table of table_elems elements: the sharded optimiser of the data-parallel engines
 * (reduce-scatter the table gradient, every rank steps its 1/N slice, all-gather the fp16 copy the forward reads).
 * halo bit 0: params[-1] belongs to the same table and holds its current value (one element exchanged per seam and step), bit 1:
 * params[n] does -- the TV terms at the piece's ends then equal the whole-table pass.  tv_codes points at the piece's first code
 * byte inside a buffer with at least one byte before it.  n a multiple of 4, 16-byte aligned pointers.  normsq_dev accumulates the
 * piece's part: the ranks' parts are summed by ONE scalar all-reduce before the AdamW launches. */
int nerf_tv_normsq_codes_piece(const float* params, const float* grads, int64_t n, int64_t table_elems, int halo, float tv_weight,
                               float grad_scale, float* normsq_dev, int accumulate, void* tv_codes, nerf_stream_t stream);
int nerf_adamw_clip_step_tv_piece(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int step, float lr,
                                  float beta1, float beta2, float eps, float weight_decay, const float* normsq_dev, float max_norm,
                                  float grad_scale, const void* tv_codes, float tv_weight, int64_t table_elems, int halo_lo,
                                  void* params_f16_out, nerf_stream_t stream);

/* One SMALL group (n <= 65536: a tiny MLP's weights) in one launch (every workgroup sums the whole gradient itself, in a fixed order,
 * then steps its own 1024 elements): squared norm of grads * grad_scale, clip_grad_norm_(max_norm) and AdamW.step() (reference run.py:624-629, the decoder group) -- instead of a zeroing
 * launch, a norm launch and an AdamW launch of ~4.5 us each.  normsq_out (nullable) receives the squared norm; zero_grads != 0:
 * grads is left zeroed for the next step's accumulating backward. */
int nerf_clip_adamw_small(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int step, float lr, float beta1,
                          float beta2, float eps, float weight_decay, float max_norm, float grad_scale, float* normsq_out,
                          int zero_grads, nerf_stream_t stream);

/* ---- f3: Part 4 dual-hash dynamic field (csrc/p4mlp.hip) ------------------------------------------------------
 * replaces, for NeuralField(mode part4).forward (src/core.py:282-352), the tinycudann FullyFusedMLP networks
 * HashDeformationDecoder.deform_net (src/decoders.py:285-295, 313-316) and InstantNeRFDecoder at pos_dim 32 + 21
 * (src/decoders.py:111-134, 149-160 with src/core.py:222), the nn.Linear TimeModulationNetwork
 * (src/decoders.py:321-371), the time Fourier code and the tri-grid blend (src/core.py:300-336).  Compiled for the
 * reference's example shapes: time code L = 10 (21 columns), time modulation 21 -> 64 -> 64, deformation grids of at
 * most 16 levels x 2 features, deformation decoder 88 -> 64 -> 64 -> 3, hidden 64, direction code L = 4.
 * params_f32 [nerf_p4_param_count()]: T1W [64,21] T1b [64] T2W [64,64] T2b [64] | D1 [64,96] D2 [64,64] D3 [16,64] |
 * S1 [64,64] S2 [16,64] | C1 [64,48] C2 [64,64] C3 [16,64] | displacement_scale [1] (module state-dict layouts).
 * One workspace of nerf_p4_workspace_bytes(n) carries the hash-grid operand images and every training image:
 * nerf_p4_workspace_offset(n, which): 0..2 nat images of the three deformation grids (write them with
 * nerf_hash_encode_fwd_nat(..., nat_dtype 1): the forward chains contract FP16 operands, as tinycudann's FullyFusedMLP
 * does; the backward chains and training images are bf16), 3 the canonical grid's; 4..6 d features [n,24] of the three deformation grids,
 * 7 d features [n,32] of the canonical grid (read them with nerf_hash_encode_bwd*). */
int64_t nerf_p4_param_count(void);
size_t nerf_p4_packed_bytes(void);
size_t nerf_p4_workspace_bytes(int64_t n);
size_t nerf_p4_workspace_offset(int64_t n, int which);
int nerf_p4_pack(const float* params_f32, void* packed, nerf_stream_t stream);
/* per-sample inputs of the field (src/core.py:289-297): t' [n] = time of the sample's ray, x' [n,3] = its position; with
 * coord_noise_std / time_noise_std > 0 (training, use_coord_noise) Gaussian noise from the counter-based generator keyed by
 * (seed, counter) and the sample's index in the global batch (first_ray * n_samples + g), t' clamped to [0,1].
 * n_samples > 0: slot_of_sample [n_rays * n_samples] maps samples to compact rows (-1 = skipped), ray_times [n_rays];
 * n_samples == 0: point mode, n_rays points with ray_times per point.  x_deform may be NULL (no coordinate noise). */
int nerf_p4_sample_inputs(const int* slot_of_sample, const float* pts_compact, const float* ray_times, int64_t n_rays,
                          int n_samples, float coord_noise_std, float time_noise_std, uint64_t seed, uint64_t counter,
                          int64_t first_ray, float* x_deform, float* t_deform, nerf_stream_t stream);
/* deformation chain: delta_x [n,3] and x_canonical = pts + delta_x.  blend [n,3] or NULL: explicit weights of the
 * three grids (the regulariser probes evaluate single grids) instead of the triangle weights of t'. */
int nerf_p4_deform_fwd(const void* packed, const float* params_f32, void* workspace, const float* pts, const float* t_deform,
                       const float* blend, int64_t n, float* delta_x, float* x_canonical, int train, nerf_stream_t stream);
/* canonical chain on the canonical grid's features (workspace slot 3), t' and unit view directions */
int nerf_p4_canon_fwd(const void* packed, void* workspace, const float* t_deform, const float* dirs, int64_t n, float* rgb,
                      float* sigma, int train, nerf_stream_t stream);
/* backward passes; parameter gradients are ACCUMULATED into grads_f32 [nerf_p4_param_count()] (zero it once per step).
 * amax_bits (nullable): NERF_AMAX_WORDS device words that max-accumulate the largest |d feature| the pass wrote, as fp32 bits;
 * grad_lm (nullable): INSTEAD of the row-major d features of workspace slots 4..7, the level-major copy float2 [levels][n]
 * (deformation chain: [3 x 12][n], virtual level = grid * 12 + level) -- both are the slots nerf_hash_encode_bwd_ws_slots names in
 * the hash backward's workspace, for the speculative scatters (nerf_hash_encode_bwd_ws_store_spec / _tables_spec with d_feat NULL)
 * and nerf_hash_encode_bwd_input_lm_f16. */
int nerf_p4_canon_bwd(const void* packed, void* workspace, const float* rgb, const float* sigma, const float* d_rgb,
                      const float* d_sigma, int64_t n, float* grads_f32, void* amax_bits, void* grad_lm, nerf_stream_t stream);
int nerf_p4_deform_bwd(const void* packed, const float* params_f32, void* workspace, const float* d_delta_x, int64_t n,
                       float* grads_f32, void* amax_bits, void* grad_lm, nerf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NERF_HIP_H */
