"""Training / rendering engine for the vanilla NeRF field (mode part2_nerf).

Host-side counterpart of the reference's per-step work in run_part2 (run.py:312-338) and of
render_image (src/renderer.py:387-418): it owns the flat fp32 parameter vector (reference
state_dict order), Adam moments, the packed bf16 weight streams and every workspace, and issues
the kernel sequence of one step on the current HIP stream:

    sample -> decoder fwd (+stash) -> composite fwd -> MSE grad -> composite bwd
           -> decoder dgrad chain -> wgrad -> [RCCL all-reduce] -> Adam -> repack

No arithmetic happens in PyTorch except the uniform jitter draw and the 3-element-per-ray MSE
residual (the reference's nn.MSELoss, run.py:308,334).
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import torch

from . import ops

Tensor = torch.Tensor


def param_shapes():
    """(name, shape) of NeRFDecoder parameters in reference registration order
    (src/decoders.py:49-66; keys as in state_dict under the ``decoder.`` prefix)."""
    out = []
    for i in range(8):
        k = 63 if i == 0 else (319 if i == 4 else 256)
        out += [(f"pts_layers.{i}.weight", (256, k)), (f"pts_layers.{i}.bias", (256,))]
    out += [("sigma_layer.weight", (1, 256)), ("sigma_layer.bias", (1,)),
            ("feature_layer.weight", (256, 256)), ("feature_layer.bias", (256,)),
            ("view_layer.weight", (128, 283)), ("view_layer.bias", (128,)),
            ("rgb_layer.weight", (3, 128)), ("rgb_layer.bias", (3,))]
    return out


def default_init(seed: int = 0) -> Tensor:
    """nn.Linear default init, U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weights and biases, drawn on
    the host from one seeded generator; returns the flat fp32 vector."""
    g = torch.Generator().manual_seed(seed)
    parts, fan_in = [], 1
    for name, shape in param_shapes():
        if name.endswith("weight"):
            fan_in = shape[1]
        parts.append(((torch.rand(shape, generator=g) * 2 - 1) / fan_in ** 0.5).reshape(-1))
    return torch.cat(parts)


def flatten_state_dict(sd: Dict[str, Tensor], prefix: str = "decoder.") -> Tensor:
    return torch.cat([sd[prefix + k].reshape(-1).float() for k, _ in param_shapes()])


def unflatten(flat: Tensor, prefix: str = "decoder.") -> Dict[str, Tensor]:
    out, off = {}, 0
    for k, shape in param_shapes():
        n = 1
        for s in shape:
            n *= s
        out[prefix + k] = flat[off:off + n].view(shape)
        off += n
    return out


class VanillaNerfEngine:
    def __init__(self, params: Optional[Tensor] = None, device: str = "cuda", lr: float = 5e-4,
                 near: float = 2.0, far: float = 6.0, white_bkgd: bool = True, seed: int = 0,
                 world_size: int = 1):
        self.device = torch.device(device)
        flat = default_init(seed) if params is None else params.detach().float().reshape(-1)
        assert flat.numel() == ops.MLP_PARAM_COUNT
        self.params = flat.to(self.device).contiguous()
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.grads = torch.empty_like(self.params)
        self.packed = torch.empty(ops.mlp_packed_bytes(), dtype=torch.uint8, device=self.device)
        self.lr, self.near, self.far = lr, near, far
        self.bg = (torch.ones(3) if white_bkgd else torch.zeros(3)).to(self.device)
        self.step_count = 0
        self.world_size = world_size
        self.grad_scale = torch.full((1,), 1.0 / world_size, device=self.device)
        self._ws: Dict[Tuple[str, int], Tensor] = {}
        # per-step device scalars (loss, largest output-layer derivative): one fresh zeroed slot per step
        # out of a ring that is cleared once per lap, instead of a memset launch per step
        self._scalars = torch.zeros(2, 1024, device=self.device)
        self.repack()

    # -- weights ---------------------------------------------------------------
    def repack(self, training_only: bool = False) -> None:
        """bf16 fragment streams of the current weights; a training step refreshes only the streams it uses,
        the inference stream (16x16x32 fragments) is refreshed when something is rendered next."""
        ops.mlp_pack(self.params, self.packed, which=1 if training_only else 3)
        self._infer_stale = training_only

    def _inference_weights(self) -> Tensor:
        if getattr(self, "_infer_stale", False):
            ops.mlp_pack(self.params, self.packed, which=2)
            self._infer_stale = False
        return self.packed

    def state_dict(self, prefix: str = "decoder.") -> Dict[str, Tensor]:
        return {k: v.clone() for k, v in unflatten(self.params, prefix).items()}

    def load_state_dict(self, sd: Dict[str, Tensor], prefix: str = "decoder.") -> None:
        self.params.copy_(flatten_state_dict(sd, prefix).to(self.device))
        self.repack()

    def _buf(self, kind: str, nbytes: int) -> Tensor:
        key = (kind, nbytes)
        if key not in self._ws:
            self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws[key]

    # -- training step (reference run.py:314-338) ------------------------------
    def compute_gradients(self, rays_o: Tensor, rays_d: Tensor, target: Tensor, n_samples: int = 64,
                   u: Optional[Tensor] = None, sync_grads=None, sync_grads_async=None, mark=None,
                   z: Optional[Tensor] = None) -> Tensor:
        """``z`` [R, n_samples]: jittered depths the caller already has (BlenderDataset.train_batch draws them
        together with the rays); otherwise they are drawn here (``u`` or torch.rand).
        ``mark(name)`` (optional) is called after each phase has been enqueued -- bench.py records a HIP
        event there, which times the kernels inside the step without serialising anything."""
        mark = mark or (lambda name: None)
        R = rays_o.shape[0]
        n = R * n_samples
        if z is None:
            if u is None:
                u = torch.rand(R, n_samples, device=self.device)
            z = ops.sample_rays(rays_o, rays_d, self.near, self.far, n_samples, u=u)
            mark("sample")
        stash = self._buf("stash", ops.mlp_stash_bytes(n))
        rgb, sigma = ops.mlp_fwd(self.packed, rays_o, rays_d, z, stash)
        mark("fwd")
        self._grad_calls = getattr(self, "_grad_calls", -1) + 1
        slot = self._grad_calls % self._scalars.shape[1]
        if slot == 0:
            self._scalars.zero_()
        loss, amax = self._scalars[0, slot:slot + 1], self._scalars[1, slot:slot + 1]
        # compositing + MSE + their backward in one kernel (run.py:324-337); it also hands the dgrad chain the
        # largest output-layer derivative, from which the e5m2 gradient images take their scale
        d_rgb, d_sigma, _ = ops.composite_mse_bwd(rgb.view(R, n_samples, 3), sigma.view(R, n_samples), z, rays_d, self.bg,
                                                  target, loss, amax_accum=amax)
        mark("loss")
        if sync_grads_async is not None:
            # weight gradients in two launches: the late layers' range is all-reduced (RCCL) while the
            # early layers' is still being computed; both are averaged by grad_scale below
            ops.mlp_bwd_overlapped(self.packed, stash, rgb, sigma, d_rgb.view(n, 3), d_sigma.view(n), self.grads,
                                   self._buf("bwd", ops.mlp_bwd_workspace_bytes(n)), sync_grads_async, amax=amax, mark=mark)
        else:
            ops.mlp_bwd(self.packed, stash, rgb, sigma, d_rgb.view(n, 3), d_sigma.view(n), self.grads,
                        self._buf("bwd", ops.mlp_bwd_workspace_bytes(n)), amax=amax, mark=mark)
        mark("wgrad")
        if sync_grads is not None:
            sync_grads(self.grads)            # RCCL all-reduce (sum); averaged by grad_scale in apply_gradients
        # a copy, not a view of the scalar ring: the ring is cleared every 1024 steps, a caller that keeps loss
        # tensors (to average them later) must not see them zeroed or overwritten
        return loss[0].clone()

    def apply_gradients(self) -> None:
        self.step_count += 1
        ops.adam_step(self.params, self.grads, self.exp_avg, self.exp_avg_sq, self.step_count, self.lr,
                      grad_scale=self.grad_scale if self.world_size > 1 else None)
        self.repack(training_only=True)

    def train_step(self, rays_o: Tensor, rays_d: Tensor, target: Tensor, n_samples: int = 64,
                   u: Optional[Tensor] = None, sync_grads=None, sync_grads_async=None, mark=None,
                   z: Optional[Tensor] = None) -> Tensor:
        """One step of reference run.py:314-338: gradients of the local batch (``self.grads``; summed over
        ranks by the ``sync_grads*`` callbacks), then Adam and the weight repack."""
        loss = self.compute_gradients(rays_o, rays_d, target, n_samples, u=u, sync_grads=sync_grads,
                                      sync_grads_async=sync_grads_async, mark=mark, z=z)
        self.apply_gradients()
        if mark is not None:
            mark("adam+pack")
        return loss

    # -- inference (reference render_image, src/renderer.py:387-418) ------------
    @torch.no_grad()
    def render_rays(self, rays_o: Tensor, rays_d: Tensor, n_samples: int, u: Optional[Tensor] = None):
        z = ops.sample_rays(rays_o, rays_d, self.near, self.far, n_samples, u=u)
        rgb, sigma = ops.mlp_fwd(self._inference_weights(), rays_o, rays_d, z)
        R = rays_o.shape[0]
        c, depth, acc, _, _ = ops.composite_fwd(rgb.view(R, n_samples, 3), sigma.view(R, n_samples), z, rays_d, self.bg)
        return c, depth, acc

    @torch.no_grad()
    def render_rays_hierarchical(self, rays_o: Tensor, rays_d: Tensor, n_coarse: int, n_fine: int):
        """Opt-in coarse -> inverse-CDF fine rendering with the single field ("64 coarse + 128 fine" of
        BASELINE.json; the reference itself has one stratified pass only)."""
        R = rays_o.shape[0]
        z = ops.sample_rays(rays_o, rays_d, self.near, self.far, n_coarse)
        rgb, sigma = ops.mlp_fwd(self._inference_weights(), rays_o, rays_d, z)
        w = ops.composite_fwd(rgb.view(R, n_coarse, 3), sigma.view(R, n_coarse), z, rays_d, self.bg, want_weights=True)[4]
        z_all = ops.sample_pdf(z, w, n_fine)
        S = n_coarse + n_fine
        rgb, sigma = ops.mlp_fwd(self._inference_weights(), rays_o, rays_d, z_all)
        c, depth, acc, _, _ = ops.composite_fwd(rgb.view(R, S, 3), sigma.view(R, S), z_all, rays_d, self.bg)
        return c, depth, acc

    @torch.no_grad()
    def render_image(self, rays_o: Tensor, rays_d: Tensor, n_samples: int, chunk: int = 65536, n_fine: int = 0) -> Tensor:
        shape = rays_o.shape[:-1]
        o, d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
        if n_fine == 0:
            # one launch chain over all chunks in one reused workspace (nerf_render_rays_fwd): chunking is a
            # property of the launch chain, not a Python loop with per-chunk allocations
            chunk = max(1, min(chunk, o.shape[0]))
            ws = self._buf("render", ops._lib.load().nerf_render_rays_workspace_bytes(chunk, n_samples))
            return ops.render_rays_fwd(self._inference_weights(), o, d, n_samples, self.near, self.far, self.bg, chunk, ws)[0].view(*shape, 3)
        out = torch.empty(o.shape[0], 3, device=self.device)
        for i in range(0, o.shape[0], chunk):
            out[i:i + chunk] = self.render_rays_hierarchical(o[i:i + chunk], d[i:i + chunk], n_samples, n_fine)[0]
        return out.view(*shape, 3)


class InstantNgpEngine:
    """Flat-parameter training / rendering engine for mode part2_instant: the per-step work of
    reference run_part2_instant (run.py:579-646) as one kernel sequence with no torch autograd or
    optimiser objects in the loop:

        sample+mask+compact -> hash encode -> tiny MLPs (+stash) -> indexed composite -> MSE grad
        -> indexed composite bwd -> tiny-MLP dgrad/wgrad -> hash scatter -> [all-reduce]
        -> TV + clip + AdamW (table) , clip + AdamW (nets) -> repack ; cosine LR on the host.
    """

    def __init__(self, cfg: Optional[dict] = None, device: str = "cuda", seed: int = 0, world_size: int = 1):
        cfg = dict(cfg or {})
        self.device = torch.device(device)
        self.seed = int(seed)
        self.bound = float(cfg.get("scene_bound", 1.5))
        if cfg.get("n_levels", 16) != 16 or cfg.get("n_features_per_level", 2) != 2 or cfg.get("hidden_dim", 64) != 64:
            raise NotImplementedError("libnerf_hip's tiny-MLP kernels are compiled for 16 levels x 2 features (32 hash "
                                      "channels) and 64 hidden units")
        self.levels = ops.HashLevelTable(cfg.get("n_levels", 16), cfg.get("log2_hashmap_size", 19),
                                         cfg.get("base_resolution", 16), cfg.get("per_level_scale", 1.5))
        g = torch.Generator().manual_seed(seed)
        # the table's buffers are allocated padded to world equal slices of whole 1024-element blocks (sharded optimiser:
        # project-nerf_amd/sharded.py); the views below are the table itself
        from .sharded import padded_length
        n_tab = self.levels.entries * 2
        n_pad = padded_length(n_tab, max(int(world_size), 1))
        self._table_buf = torch.zeros(n_pad, device=self.device)
        self._table_buf[:n_tab] = ((torch.rand(n_tab, generator=g) * 2 - 1) * 1e-4).to(self.device)
        self.table = self._table_buf[:n_tab]
        self.shard = None

        def xavier(rows, cols, fi, fo):
            return (torch.rand(rows, cols, generator=g) * 2 - 1) * (6.0 / (fi + fo)) ** 0.5
        w1 = xavier(64, 48, 43, 64)
        w1[:, 43:] = 0
        w3 = xavier(16, 64, 64, 3)
        w3[3:] = 0
        self.net = torch.cat([xavier(64, 32, 32, 64).reshape(-1), xavier(16, 64, 64, 16).reshape(-1), w1.reshape(-1),
                              xavier(64, 64, 64, 64).reshape(-1), w3.reshape(-1)]).to(self.device)
        self._m_buf, self._v_buf = torch.zeros(n_pad, device=self.device), torch.zeros(n_pad, device=self.device)
        self.state = {"table": (self._m_buf[:n_tab], self._v_buf[:n_tab]), "net": (torch.zeros_like(self.net), torch.zeros_like(self.net))}
        self._g_table_buf = torch.zeros(n_pad, device=self.device)
        self.g_table = self._g_table_buf[:n_tab]
        self.g_net = torch.empty_like(self.net)
        self._hash_ws = None
        # fp16 copy of the table for the forward gathers (tinycudann evaluates its grid from fp16 parameters next to
        # the fp32 master copy too): written by the optimiser kernel, refreshed here whenever torch code has
        # touched ``self.table`` in place (tensor version counter); cfg half_table: false keeps fp32 gathers
        self.half_table = bool(cfg.get("half_table", True))
        # precount: the hash forward counts the scatter's bins and the decoder's backward writes level-major gradients, so the
        # hash backward starts at its plan pass.  Built as the round-2 review asked and measured SLOWER on the same box
        # (tools/ab_instant_precount.py: 0.699 against 0.651 ms per step -- eight LDS atomics per point and level in the
        # latency-bound forward cost more than the 0.05 ms count pass they replace): off by default
        self.precount = bool(cfg.get("precount", False))
        # speculative hash backward (default on, single rank): NO count pass -- the bins' capacities come from the true counts of the
        # previous step's call; records that do not fit are added with atomics by a last small launch, a lost record (never seen in
        # training; the status block is read back one step late) switches the form off.  Used when the occupancy grid is the one
        # the estimates were taken on and the active-sample count is within 10 % of that call's
        from .specbwd import SpeculativeScatter
        self.spec = SpeculativeScatter(bool(cfg.get("speculative_hash_backward", True)) and not os.environ.get("NERF_NO_SPECULATIVE_BWD"))  # env: A/B aid
        self._table_h_buf = torch.zeros(n_pad, device=self.device, dtype=torch.float16) if self.half_table else None
        self.table_h = self._table_h_buf[:n_tab] if self.half_table else None
        self._table_version = None
        self.packed = ops.imlp_pack(self.net)
        self.near, self.far = float(cfg.get("near", 2.0)), float(cfg.get("far", 6.0))
        self.lr0, self.eta_min = float(cfg.get("learning_rate", 1e-2)), float(cfg.get("eta_min", 1e-4))
        self.t_max = int(cfg.get("train_iters", 2000))
        self.wd = float(cfg.get("weight_decay", 1e-5))
        self.tv_weight = float(cfg.get("tv_loss_weight", 1e-6)) if cfg.get("use_tv_loss", True) else 0.0
        self.bg = (torch.ones(3) if cfg.get("white_bkgd", True) else torch.zeros(3)).to(self.device)
        res = int(cfg.get("grid_resolution", 128))
        self.grid_threshold = float(cfg.get("grid_threshold", 0.12))
        self.grid = torch.zeros(res, res, res, device=self.device)
        self.binary_grid = torch.ones(res, res, res, dtype=torch.bool, device=self.device)
        self.step_count, self.world_size = 0, world_size
        self._scratch = ops.normsq_ws(self.device)
        self._net_scratch = ops.normsq_ws(self.device)
        self._loss_ring = torch.zeros(65536, device=self.device)      # a step's loss is a VIEW of its slot: valid for the next 32768 steps

    def lr(self) -> float:
        import math
        return self.eta_min + (self.lr0 - self.eta_min) * (1 + math.cos(math.pi * self.step_count / self.t_max)) / 2

    def _gather_table(self) -> Tensor:
        """the table the forward gathers from: the fp16 copy, brought up to date if torch code wrote the fp32 one"""
        if not self.half_table:
            return self.table.view(-1, 2)
        if self._table_version != self.table._version:
            ops.f32_to_f16(self.table, self.table_h)
            self._table_version = self.table._version
        return self.table_h.view(-1, 2)

    def _field(self, pts: Tensor, dirs: Tensor, train: bool, hist_ws: Optional[Tensor] = None):
        """``hist_ws`` (training, fp16 table): the forward also counts the corners per (level, table slice) into the binned
        hash backward's workspace, so that the backward can skip its count pass"""
        lib = ops._lib.load()
        n = pts.shape[0]
        ws = torch.empty(lib.nerf_imlp_workspace_bytes(n), device=self.device, dtype=torch.uint8)
        if hist_ws is not None:
            ops._lib.check(lib.nerf_hash_encode_fwd_f16_hist(pts.data_ptr(), n, self._gather_table().data_ptr(), self.levels.n_levels,
                                                             *self.levels.host_args(), float(self.bound), ws.data_ptr(), hist_ws.data_ptr(),
                                                             hist_ws.numel(), ops._stream()), "nerf_hash_encode_fwd_f16_hist")
        else:
            ops.hash_encode_fwd(pts, self._gather_table(), self.levels, self.bound, want_f32=False, out_nat=ws)
        rgb, sigma = torch.empty(n, 3, device=self.device), torch.empty(n, device=self.device)
        ops._lib.check(lib.nerf_imlp_fwd(self.packed.data_ptr(), ws.data_ptr(), dirs.data_ptr(), n, rgb.data_ptr(),
                                         sigma.data_ptr(), 1 if train else 0, ops._stream()), "nerf_imlp_fwd")
        return rgb, sigma, ws

    def level_groups(self, n_groups: int = 4):
        """Level ranges of roughly equal table bytes: the hash backward runs group by group so that a
        data-parallel caller can all-reduce one group's slice of the gradient while the next is computed."""
        off = [int(o) for o in self.levels.offset] + [self.levels.entries]
        cuts, target = [0], self.levels.entries / n_groups
        for l in range(1, self.levels.n_levels):
            if off[l] >= target * len(cuts) and len(cuts) < n_groups:
                cuts.append(l)
        cuts.append(self.levels.n_levels)
        return [(cuts[k], cuts[k + 1]) for k in range(len(cuts) - 1) if cuts[k] < cuts[k + 1]]

    def prepare_batch(self, rays_o: Tensor, rays_d: Tensor, n_samples: int = 128, u: Optional[Tensor] = None, first_ray: int = 0):
        """Queues the data-only front of a step -- stratified depths, occupancy mask, compaction (rows a1-a4) --
        and the read-back of the active count WITHOUT waiting for it.  A loop that prepares batch i+1 before it
        runs step i never stalls on the count: it arrives while step i computes (the reference, and
        ``compute_gradients`` without ``prepared``, wait at this point of every step).  The batch is compacted
        against the occupancy grid as it is now; prepare again after ``update_grid`` to use the new one."""
        jitter = None
        if u is None:                 # the jitter is drawn in the compaction kernel: no [R, S] tensor of uniforms
            self._jitter_counter = getattr(self, "_jitter_counter", -1) + 1
            jitter = (self.seed, self._jitter_counter)
        return ops.sample_compact_async(rays_o, rays_d, self.near, self.far, n_samples, self.binary_grid, self.bound, u=u,
                                        jitter=jitter, first_ray=first_ray)

    def compute_gradients(self, rays_o: Tensor, rays_d: Tensor, target: Tensor, n_samples: int = 128,
                          u: Optional[Tensor] = None, sync_grads_async=None, reduce_dtype=None, prepared=None,
                          shard_grads: bool = False) -> Tensor:
        """``shard_grads`` (after enable_sharded_optimizer): the local gradients as on one GPU, then ONE reduce-scatter of the table
        gradient (this rank ends with the summed gradient of its slice) and the tiny MLPs' small all-reduce -- the same two
        collectives on every rank whatever its shard held.  Otherwise see _compute_gradients."""
        if shard_grads and self.shard is not None:
            import torch.distributed as dist
            loss = self._compute_gradients(rays_o, rays_d, target, n_samples, u=u, prepared=prepared)
            self.shard.reduce_scatter_grads()
            if self.world_size > 1:
                dist.all_reduce(self.g_net, op=dist.ReduceOp.SUM)
            return loss
        return self._compute_gradients(rays_o, rays_d, target, n_samples, u=u, sync_grads_async=sync_grads_async, reduce_dtype=reduce_dtype,
                                       prepared=prepared)

    def _compute_gradients(self, rays_o: Tensor, rays_d: Tensor, target: Tensor, n_samples: int = 128,
                           u: Optional[Tensor] = None, sync_grads_async=None, reduce_dtype=None, prepared=None) -> Tensor:
        """Forward + backward of one batch (reference run.py:579-619): fills ``g_table`` / ``g_net`` with the
        gradients of the LOCAL mean-squared error and returns the loss.  ``sync_grads_async(view)`` (data
        parallel) starts the all-reduce of a finished gradient range and returns a handle: the tiny-MLP
        gradients go first, then the table gradient level group by level group, each on the wire while the
        next group's scatter runs; ``reduce_dtype=torch.bfloat16`` halves the bytes on the wire (the table
        gradient is 52 MB in fp32: a ring all-reduce over xGMI costs about half a step)."""
        lib = ops._lib.load()
        R = rays_o.shape[0]
        if prepared is not None:
            z, slots, pts, dirs = prepared.get()
        else:
            if u is None:
                u = torch.rand(R, n_samples, device=self.device)
            z, slots, pts, dirs = ops.sample_compact(rays_o, rays_d, self.near, self.far, n_samples, self.binary_grid,
                                                     self.bound, u=u)
        n = pts.shape[0]
        handles = []

        def reduce(view):
            if sync_grads_async is None:
                return
            if reduce_dtype is None or view.dtype == reduce_dtype:
                handles.append((sync_grads_async(view), None, None))
            else:
                wire = view.to(reduce_dtype)                      # comm plumbing: narrow, sum, widen
                handles.append((sync_grads_async(wire), wire, view))

        def table_slice(lo, hi):
            e0 = 2 * int(self.levels.offset[lo])
            e1 = 2 * (int(self.levels.offset[hi]) if hi < self.levels.n_levels else self.levels.entries)
            return self.g_table[e0:e1]

        if n == 0:
            # no active sample on THIS rank (its rays miss the occupied cells): zero gradients, but exactly the
            # collectives its peers issue -- the tiny-MLP gradient, then one all-reduce per level group with the same
            # element counts (mismatched collectives across ranks hang or corrupt memory in RCCL)
            self.g_net.zero_()
            self.g_table.zero_()
            pred = self.bg.expand(R, 3)
            loss = ((pred - target) ** 2).mean()
            reduce(self.g_net)
            if sync_grads_async is not None:
                for lo, hi in self.level_groups():
                    reduce(table_slice(lo, hi))
        else:
            # opt-in (cfg precount: true; single rank, fp16 table): forward counts the scatter's bins, the decoder's backward
            # hands its feature gradients over level-major -- the hash backward starts at its plan pass
            precount = sync_grads_async is None and self.half_table and self.precount
            hws = self._hash_bwd_workspace(n)
            grid_id = (self.binary_grid.data_ptr(), self.binary_grid._version)
            spec = sync_grads_async is None and not precount and self.half_table and self.spec.ok(hws.data_ptr(), grid_id, n)
            if spec:
                self.spec.begin(hws)               # (a launch only after a counted call: a speculative call leaves the header clean)
            rgb, sigma, ws = self._field(pts, dirs, True, hist_ws=hws if precount else None)
            P = lambda t: t.data_ptr()
            # a fresh zeroed loss slot per step out of a ring cleared half a lap ahead (no fill launch, no copy launch per step: the
            # returned loss is a view of the slot)
            self._grad_calls = getattr(self, "_grad_calls", -1) + 1
            ring = self._loss_ring.numel()
            slot = self._grad_calls % ring
            if slot % (ring // 2) == 0 and self._grad_calls > 0:
                self._loss_ring[slot:slot + ring // 2].zero_()
            loss = self._loss_ring[slot:slot + 1]
            d_rgb, d_sigma, _ = ops.composite_mse_bwd(rgb, sigma, z, rays_d, self.bg, target, loss, slots=slots)
            if precount or spec:
                import ctypes
                amax_p, lm_p = ctypes.c_void_p(), ctypes.c_void_p()
                ops._lib.check(lib.nerf_hash_encode_bwd_ws_slots(P(hws), n, self.levels.n_levels, ctypes.byref(amax_p), ctypes.byref(lm_p)),
                               "nerf_hash_encode_bwd_ws_slots")
                # the decoder's backward hands its feature gradients over level-major and max-accumulates their largest magnitude
                ops._lib.check(lib.nerf_imlp_bwd_lm(P(self.packed), P(ws), P(rgb), P(sigma), P(d_rgb), P(d_sigma), n, P(self.g_net),
                                                    lm_p, amax_p, ops._stream()), "nerf_imlp_bwd_lm")
                if spec:                           # no count pass: capacities from the last call's true counts
                    status = self.spec.status_block()
                    ops._lib.check(lib.nerf_hash_encode_bwd_ws_store_spec(P(pts), n, self.levels.n_levels, *self.levels.host_args(),
                                                                          float(self.bound), None, P(self.g_table), P(hws), hws.numel(),
                                                                          status.data_ptr(), ops._stream()), "nerf_hash_encode_bwd_ws_store_spec")
                    self.spec.issued(hws, grid_id, n, status)
                else:
                    ops._lib.check(lib.nerf_hash_encode_bwd_ws_store_precounted(P(pts), n, self.levels.n_levels, *self.levels.host_args(),
                                                                                float(self.bound), P(self.g_table), P(hws), hws.numel(),
                                                                                ops._stream()), "nerf_hash_encode_bwd_ws_store_precounted")
                    self.spec.counted(hws, grid_id, n)
                return loss[0]
            d_feat = torch.empty(n, 2 * self.levels.n_levels, device=self.device)
            ops._lib.check(lib.nerf_imlp_bwd(P(self.packed), P(ws), P(rgb), P(sigma), P(d_rgb), P(d_sigma), n,
                                             P(self.g_net), P(d_feat), ops._stream()), "nerf_imlp_bwd")
            reduce(self.g_net)
            # overwrite form: the table gradient is stored slice by slice -- no 52 MB memset, no read-back
            if sync_grads_async is None:
                ops.hash_encode_bwd(pts, self.levels, self.bound, d_feat, self.g_table, workspace=hws, overwrite=True)
                self.spec.counted(hws, grid_id, n)                  # the counted call left the bins' true counts in the workspace
            else:
                for lo, hi in self.level_groups():
                    ops.hash_encode_bwd(pts, self.levels, self.bound, d_feat, self.g_table, level_range=(lo, hi), workspace=hws, overwrite=True)
                    reduce(table_slice(lo, hi))
                self.spec.invalidate()                              # level-range calls: no estimates for all levels
            loss = loss[0]
        for h, wire, view in handles:
            if h is not None:
                h.wait()
            if wire is not None:
                view.copy_(wire)
        return loss

    def _hash_bwd_workspace(self, n: int) -> Tensor:
        """Workspace of the binned hash-gradient scatter, grown to the largest point count seen (8 B per corner
        record + 8 B per point and level: 0.23 GB for the steady-state 200 k points, 2.3 GB for an unpruned 16384 x 128 batch)."""
        need = ops.hash_encode_bwd_workspace_bytes(n, self.levels.n_levels)
        if self._hash_ws is None or self._hash_ws.numel() < need:
            self._hash_ws = None                                  # release before growing
            self._hash_ws = torch.empty(int(need * 1.25), dtype=torch.uint8, device=self.device)
        return self._hash_ws

    def enable_sharded_optimizer(self, rank: int) -> None:
        """Data parallelism as SURVEY 8(e) specifies for the table (project-nerf_amd/sharded.py): reduce-scatter of the table
        gradient, this rank steps its 1/world slice, the fp16 copy the forward reads is all-gathered.  Call once, after the replicas'
        parameters were made equal; compute_gradients(shard_grads=True) + apply_gradients() then take this path."""
        from .sharded import ShardedTableOptimizer
        if not self.half_table:
            raise ValueError("the sharded optimiser all-gathers the fp16 copy of the table: half_table must stay on")
        self._gather_table()
        n = self.table.numel()
        self.shard = ShardedTableOptimizer([(0, n, self.tv_weight)], n, rank, self.world_size, self._table_buf, self._g_table_buf,
                                           self._m_buf, self._v_buf, self._table_h_buf)

    def gather_master(self) -> None:
        """sharded optimiser: bring the fp32 master copy of every slice up to date on this rank (checkpoints, validation)"""
        if self.shard is not None:
            self.shard.gather_master()
            self._table_version = self.table._version

    def apply_gradients(self) -> None:
        """TV-L1 + global-norm clip + AdamW on the table, clip + AdamW on the tiny MLPs, cosine LR
        (reference run.py:611-630).  After a summing all-reduce the DATA gradient is averaged (1/world);
        the TV term is added unscaled."""
        lr = self.lr()
        self.step_count += 1
        scale = 1.0 / self.world_size
        if self.shard is not None:
            # the table group sharded over the ranks: this rank's slice of TV + norm and of clip + AdamW, ONE scalar all-reduce
            # for the table's squared norm; the tiny MLPs (their own clip, reference run.py:624-627) stepped on every rank
            import torch.distributed as dist
            normsq = self._scratch
            self.shard.accumulate_normsq(normsq, scale, first=True)
            if self.world_size > 1:
                dist.all_reduce(normsq[0:1], op=dist.ReduceOp.SUM)
            self.shard.adamw(normsq, self.step_count, lr, self.wd, 1.0, scale)
            ops.tv_clip_adamw_step(self.net, self.g_net, *self.state["net"], self.step_count, lr, max_norm=1.0,
                                   weight_decay=self.wd, grad_scale=scale, scratch=self._net_scratch)
            self.shard.exchange()
            self._table_version = self.table._version      # the fp16 copy is current (all-gathered); the fp32 master only in this slice
            ops.imlp_pack(self.net, self.packed)
            return
        if getattr(self, "_tv_codes", None) is None:
            self._tv_codes = torch.empty((self.table.numel() + 3) // 4, dtype=torch.uint8, device=self.device)
        ops.tv_clip_adamw_step(self.table, self.g_table, *self.state["table"], self.step_count, lr, tv_weight=self.tv_weight,
                               max_norm=1.0, weight_decay=self.wd, grad_scale=scale, scratch=self._scratch, tv_codes=self._tv_codes,
                               shadow_f16=self.table_h if self._table_version == self.table._version else None)
        ops.tv_clip_adamw_step(self.net, self.g_net, *self.state["net"], self.step_count, lr, max_norm=1.0,
                               weight_decay=self.wd, grad_scale=scale, scratch=self._scratch)
        ops.imlp_pack(self.net, self.packed)

    def train_step(self, rays_o: Tensor, rays_d: Tensor, target: Tensor, n_samples: int = 128,
                   u: Optional[Tensor] = None, sync_grads=None, sync_grads_async=None, reduce_dtype=None, prepared=None) -> Tensor:
        loss = self.compute_gradients(rays_o, rays_d, target, n_samples, u=u, sync_grads_async=sync_grads_async,
                                      reduce_dtype=reduce_dtype, prepared=prepared, shard_grads=self.shard is not None)
        if sync_grads is not None and self.shard is None:                # blocking form: two collectives after the backward pass
            sync_grads(self.g_table)
            sync_grads(self.g_net)
        self.apply_gradients()
        return loss

    @torch.no_grad()
    def update_grid(self) -> float:
        """DensityGrid.update for a static field (reference src/renderer.py:35-132)."""
        res = self.grid.shape[0]
        pts = ops.grid_lattice(self.bound, res, self.device)
        sig = torch.empty(res ** 3, device=self.device)
        zeros = torch.zeros(2 ** 18, 3, device=self.device)
        for i in range(0, pts.shape[0], 2 ** 18):
            p = pts[i:i + 2 ** 18]
            sig[i:i + 2 ** 18] = self._field(p, zeros[:p.shape[0]], False)[1]
        self.grid = sig.view(res, res, res)
        self.binary_grid, ratio = ops.grid_threshold(self.grid, self.grid_threshold)
        return ratio

    @torch.no_grad()
    def render_rays(self, rays_o: Tensor, rays_d: Tensor, n_samples: int):
        z, slots, pts, dirs = ops.sample_compact(rays_o, rays_d, self.near, self.far, n_samples, self.binary_grid, self.bound)
        R = rays_o.shape[0]
        if pts.shape[0] == 0:
            return self.bg.expand(R, 3).clone(), torch.zeros(R, device=self.device), torch.zeros(R, device=self.device)
        rgb, sigma, _ = self._field(pts, dirs, False)
        return ops.composite_indexed(rgb, sigma, slots, z, rays_d, self.bg)

    @torch.no_grad()
    def render_image(self, rays_o: Tensor, rays_d: Tensor, n_samples: int, chunk: int = 200000) -> Tensor:
        shape = rays_o.shape[:-1]
        o, d = rays_o.reshape(-1, 3).contiguous(), rays_d.reshape(-1, 3).contiguous()
        out = torch.empty(o.shape[0], 3, device=self.device)
        for i in range(0, o.shape[0], chunk):
            out[i:i + chunk] = self.render_rays(o[i:i + chunk], d[i:i + chunk], n_samples)[0]
        return out.view(*shape, 3)
