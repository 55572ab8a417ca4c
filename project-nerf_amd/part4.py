"""Part 4 dual-hash dynamic field on the fused HIP chains (csrc/p4mlp.hip): the operator behind
``NeuralField('part4').forward`` (reference src/core.py:282-352) and the flat-parameter training engine of the loop
body of reference run_part4 (run.py:1808-1990).

Two entry points over the same kernels:
  * ``field(...)``: differentiable operator for the module path (NeuralField + torch.optim): the parameters stay the
    module's own nn.Parameters (reference state-dict keys), autograd receives their gradients from the HIP backward;
  * ``DualHashEngine``: owns flat parameter / gradient / Adam vectors and issues the whole step -- compaction, the four
    hash encodings, the two fused MLP chains, compositing + MSE + displacement regulariser + their backward, the hash
    scatters, TV terms, ONE global-norm clip and AdamW with the reference's per-group learning rates -- with no torch
    autograd, no torch.optim and no library GEMM in the loop.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional

import torch

from . import _lib, ops

Tensor = torch.Tensor
P = lambda t: None if t is None else t.data_ptr()

# offsets into the flat parameter vector (csrc/p4mlp.hip)
T1W, T1B, T2W, T2B, D1, D2, D3, S1, S2, C1, C2, C3, SCALE, N_PARAMS = (0, 1344, 1408, 5504, 5568, 11712, 15808, 16832, 20928, 21952,
                                                                          25024, 29120, 30144, 30145)
# (state-dict key, offset, element count) of the module parameters inside the flat vector
MODULE_SLICES = (
    ("time_modulation.net.0.weight", T1W, 64 * 21), ("time_modulation.net.0.bias", T1B, 64),
    ("time_modulation.net.2.weight", T2W, 64 * 64), ("time_modulation.net.2.bias", T2B, 64),
    ("deform_decoder.deform_net.params", D1, 64 * 96 + 64 * 64 + 16 * 64),
    ("decoder.sigma_net.params", S1, 64 * 64 + 16 * 64), ("decoder.color_net.params", C1, 64 * 48 + 64 * 64 + 16 * 64),
    ("deform_decoder.displacement_scale", SCALE, 1),
)
GRIDS = ("deform_grid_start", "deform_grid_mid", "deform_grid_end", "canonical_repr")


def supported(cfg: dict) -> Optional[str]:
    """None if the fused chains are compiled for this configuration, else the reason they are not."""
    want = {"L_embed_time": (10, 10), "L_embed_dir": (4, 4), "time_modulation_dim": (64, 64), "time_modulation_layers": (2, 2),
            "deform_hidden_dim": (64, 64), "hidden_dim": (64, 64), "deform_n_levels": (12, 14), "n_levels": (16, 16),
            "deform_n_features_per_level": (2, 2), "n_features_per_level": (2, 2)}
    for key, (compiled, default) in want.items():
        if cfg.get(key, default) != compiled:
            return f"{key}={cfg.get(key, default)} (compiled: {compiled})"
    return None


def _check_count():
    n = _lib.load().nerf_p4_param_count()
    if n != N_PARAMS:
        raise _lib.NerfHipError(f"libnerf_hip.so reports {n} Part 4 parameters, this binding expects {N_PARAMS}")


class Workspace:
    """hash operand images, every training image, the d-feature arrays of ``n`` points (the kernels derive the layout from n);
    ``buf``: a caller-owned buffer of at least ``Workspace.bytes(n)`` bytes to lay it out in (the engine's grow-only buffer)"""

    @staticmethod
    def bytes(n: int) -> int:
        return max(_lib.load().nerf_p4_workspace_bytes(n), 256)

    def __init__(self, n: int, device, buf: Optional[Tensor] = None):
        lib = _lib.load()
        self.n = n
        need = Workspace.bytes(n)
        if buf is not None and buf.numel() < need:
            raise ValueError(f"Part 4 workspace: {need} bytes needed for {n} points, {buf.numel()} given")
        self.buf = buf if buf is not None else torch.empty(need, dtype=torch.uint8, device=device)
        self._off = [lib.nerf_p4_workspace_offset(n, k) for k in range(8)]

    def nat(self, k: int) -> Tensor:                     # 0..2 deformation grids, 3 canonical
        return self.buf[self._off[k]:]

    def d_feat(self, k: int) -> Tensor:                  # 0..2 [n,24], 3 [n,32]
        width = 32 if k == 3 else 24
        return self.buf[self._off[4 + k]:self._off[4 + k] + self.n * width * 4].view(torch.float32).view(self.n, width)


def pack(params: Tensor, packed: Optional[Tensor] = None) -> Tensor:
    lib = _lib.load()
    _check_count()
    params = ops._dev(params, "params")
    if params.numel() != N_PARAMS:
        raise ValueError(f"Part 4 networks: {N_PARAMS} parameters expected, got {params.numel()}")
    if packed is None:
        packed = torch.empty(lib.nerf_p4_packed_bytes(), dtype=torch.uint8, device=params.device)
    _lib.check(lib.nerf_p4_pack(P(params), P(packed), ops._stream()), "nerf_p4_pack")
    return packed


def sample_inputs(slots: Optional[Tensor], pts: Tensor, times: Tensor, n_rays: int, n_samples: int, std_x: float = 0.0,
                  std_t: float = 0.0, seed: int = 0, counter: int = 0, first_ray: int = 0):
    """(x' [n,3] or None, t' [n]) of the compacted samples (``n_samples`` > 0: ``times`` per ray) or of plain points
    (``n_samples`` == 0: ``times`` per point)."""
    lib = _lib.load()
    n = pts.shape[0]
    t_def = torch.empty(n, device=pts.device)
    x_def = torch.empty(n, 3, device=pts.device) if std_x > 0.0 else None
    _lib.check(lib.nerf_p4_sample_inputs(P(slots), P(pts), P(ops._dev(times.reshape(-1), "times")), n_rays, n_samples, float(std_x),
                                         float(std_t), int(seed), int(counter) & 0xFFFFFF, int(first_ray), P(x_def), P(t_def),
                                         ops._stream()), "nerf_p4_sample_inputs")
    return x_def, t_def


def forward_chain(packed, params, tables, levels_d, levels_c, bound, pts, x_def, t_def, dirs, ws: Workspace, train: bool,
                  blend: Optional[Tensor] = None):
    """hash encodings + both fused chains -> (rgb [n,3], sigma [n], delta_x [n,3], x_canonical [n,3]).
    ``tables``: [start, mid, end, canonical], each fp32 [E,2] or an fp16 copy."""
    lib = _lib.load()
    n = pts.shape[0]
    dev = pts.device
    x_in = pts if x_def is None else x_def
    # the engine's three deformation tables are views of one flat fp16 buffer: one launch; separate tensors: one launch each
    if not ops.hash_encode_fwd_nat_tables(x_in, tables[:3], levels_d, bound, [ws.nat(k) for k in range(3)], fp16=True):
        for k in range(3):
            ops.hash_encode_fwd_nat(x_in, tables[k], levels_d, bound, ws.nat(k), fp16=True)
    dx, xc = torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev)
    _lib.check(lib.nerf_p4_deform_fwd(P(packed), P(params), P(ws.buf), P(pts), P(t_def), P(blend), n, P(dx), P(xc), 1 if train else 0,
                                      ops._stream()), "nerf_p4_deform_fwd")
    if dirs is None:
        return None, None, dx, xc
    ops.hash_encode_fwd_nat(xc, tables[3], levels_c, bound, ws.nat(3), fp16=True)
    rgb, sigma = torch.empty(n, 3, device=dev), torch.empty(n, device=dev)
    _lib.check(lib.nerf_p4_canon_fwd(P(packed), P(ws.buf), P(t_def), P(dirs), n, P(rgb), P(sigma), 1 if train else 0, ops._stream()),
               "nerf_p4_canon_fwd")
    return rgb, sigma, dx, xc


def backward_chain(packed, params, table_c, levels_d, levels_c, bound, x_in, xc, ws: Workspace, rgb, sigma, d_rgb, d_sigma,
                   d_dx_extra, g_net, g_tables, hash_ws=None, after_canonical=None, after_grid=None, overwrite=False, tables_ws=None,
                   extra_in_place=False, spec=None):
    """Adds the gradients of one batch into ``g_net`` [30145] and ``g_tables`` (4 tensors [E*2]); ``overwrite`` (needs
    ``hash_ws``): the table gradients are STORED instead (no zeroing by the caller, no read-back -- the engine's data batch);
    ``tables_ws(n, n_levels, n_tables)``: workspace for scattering to the three deformation grids in one pass.  ``d_dx_extra`` [n,3] or
    None: gradient reaching delta_x directly (displacement regulariser, a caller's loss on delta_x); rgb None: the
    deformation chain alone (regulariser probes).  ``after_*`` callbacks: data-parallel hooks (ranges that are final).
    ``extra_in_place`` (fp16 canonical table): d_dx_extra is the caller's scratch -- the gradient through x_canonical is added INTO it
    (no zeroing launch, no separate add).  ``spec`` = (SpeculativeScatter of the canonical grid's workspace, of the deformation grids',
    occupancy-grid identity): the engine's steady state -- both scatters skip their count passes when the estimates hold."""
    lib = _lib.load()
    n = x_in.shape[0]
    d_dx = d_dx_extra
    # the engine's steady state: the scatters skip their count passes (specbwd.py) -- capacities from the last call's true counts
    spec_c = spec_d = None
    use_c = use_d = False
    hws_c = hws_d = None
    if spec is not None and overwrite and tables_ws is not None and hash_ws is not None and rgb is not None:
        spec_c, spec_d, grid_id = spec
        hws_c, hws_d = hash_ws(n, levels_c.n_levels), tables_ws(n, levels_d.n_levels, 3)
        use_c = table_c.dtype == torch.float16 and spec_c.ok(hws_c.data_ptr(), grid_id, n)
        use_d = spec_d.ok(hws_d.data_ptr(), grid_id, n)
    if rgb is not None:
        amax_c = lm_c = None
        if use_c:                                # the chain hands its d features over level-major, inside the scatter's workspace
            spec_c.begin(hws_c)
            amax_c, lm_c = ops.hash_bwd_slots(hws_c, n, levels_c.n_levels)
        _lib.check(lib.nerf_p4_canon_bwd(P(packed), P(ws.buf), P(rgb), P(sigma), P(d_rgb), P(d_sigma), n, P(g_net), amax_c, lm_c, ops._stream()),
                   "nerf_p4_canon_bwd")
        d_feat_c = ws.d_feat(3)
        in_place = extra_in_place and d_dx_extra is not None and table_c.dtype == torch.float16 and d_dx_extra.is_contiguous()
        d_xc = ops.hash_encode_bwd_input(xc, table_c.view(-1, 2), levels_c, bound, None if use_c else d_feat_c,
                                         add_to=d_dx_extra if in_place else None, grad_lm=lm_c)
        if use_c:
            status = spec_c.status_block()
            _lib.check(lib.nerf_hash_encode_bwd_ws_store_spec(P(xc), n, levels_c.n_levels, *levels_c.host_args(), float(bound), None,
                                                              P(g_tables[3]), P(hws_c), hws_c.numel(), status.data_ptr(), ops._stream()),
                       "nerf_hash_encode_bwd_ws_store_spec")
            spec_c.issued(hws_c, grid_id, n, status)
        else:
            ops.hash_encode_bwd(xc, levels_c, bound, d_feat_c, g_tables[3], workspace=hash_ws(n, levels_c.n_levels) if hash_ws else None,
                                overwrite=overwrite)
            if spec_c is not None:
                spec_c.counted(hws_c, grid_id, n)
        if after_grid is not None:
            after_grid(3)
        d_dx = d_xc if (d_dx_extra is None or in_place) else d_xc.add_(d_dx_extra)
    amax_d = lm_d = None
    if use_d:
        spec_d.begin(hws_d)
        amax_d, lm_d = ops.hash_bwd_slots(hws_d, n, 3 * levels_d.n_levels)
    _lib.check(lib.nerf_p4_deform_bwd(P(packed), P(params), P(ws.buf), P(d_dx.contiguous()), n, P(g_net), amax_d, lm_d, ops._stream()),
               "nerf_p4_deform_bwd")
    if after_canonical is not None:
        after_canonical()
    # the engine's overwrite form: the three deformation grids (views of one flat gradient buffer) in one pass of launches
    status = spec_d.status_block() if use_d else None
    if overwrite and tables_ws is not None and ops.hash_encode_bwd_tables(x_in, levels_d, bound, [ws.d_feat(k) for k in range(3)], g_tables[:3],
                                                                          tables_ws, spec_status=status, spec_lm=use_d):
        if use_d:
            spec_d.issued(hws_d, grid_id, n, status)
        elif spec_d is not None:
            spec_d.counted(hws_d, grid_id, n)
        if after_grid is not None:
            for k in range(3):
                after_grid(k)
        return
    if use_d:         # (cannot happen: the estimates come from a call that took the one-pass form on the same buffers)
        raise RuntimeError("Part 4 backward: the deformation grids' gradients were handed over level-major but their one-pass scatter is unavailable")
    for k in range(3):
        ops.hash_encode_bwd(x_in, levels_d, bound, ws.d_feat(k), g_tables[k], workspace=hash_ws(n, levels_d.n_levels) if hash_ws else None,
                            overwrite=overwrite)
        if after_grid is not None:
            after_grid(k)
    if spec_c is not None:                       # the per-grid calls went through the canonical grid's workspace
        spec_c.invalidate()


# --------------------------------------------------------------------------------------------------- module path
def flat_from_modules(model) -> Tensor:
    """the flat parameter vector of the fused chains, assembled (differentiably) from the module's own parameters"""
    sd = dict(model.named_parameters())
    return torch.cat([sd[key].reshape(-1) for key, _, _ in MODULE_SLICES])


class _Part4Field(torch.autograd.Function):
    """NeuralField('part4').forward on the fused chains; gradients w.r.t. the flat network parameters and the four
    hash tables (incl. the deformation grids, reached through d features / d x_canonical)."""

    @staticmethod
    def forward(ctx, flat, t0, t1, t2, tc, pts, x_def, t_def, dirs, levels_d, levels_c, bound, train):
        packed = pack(flat.detach())
        ws = Workspace(pts.shape[0], pts.device)
        tables = [t.detach().view(-1, 2) for t in (t0, t1, t2, tc)]
        rgb, sigma, dx, xc = forward_chain(packed, flat.detach(), tables, levels_d, levels_c, bound, pts, x_def, t_def, dirs, ws, train)
        if train:
            ctx.save_for_backward(flat, tc, pts if x_def is None else x_def, xc, rgb, sigma)
            ctx.packed, ctx.ws, ctx.meta = packed, ws, (levels_d, levels_c, bound, [t.numel() for t in (t0, t1, t2, tc)])
        return rgb, sigma, dx

    @staticmethod
    def backward(ctx, d_rgb, d_sigma, d_dx):
        flat, tc, x_in, xc, rgb, sigma = ctx.saved_tensors
        levels_d, levels_c, bound, sizes = ctx.meta
        g_net = torch.zeros(N_PARAMS, device=flat.device)
        g_tables = [torch.zeros(s, device=flat.device) for s in sizes]
        scratch = lambda n, L: ops._hash_bwd_scratch(x_in, levels_c if L == levels_c.n_levels else levels_d)
        backward_chain(ctx.packed, flat.detach(), tc.detach(), levels_d, levels_c, bound, x_in, xc, ctx.ws, rgb, sigma,
                       d_rgb.contiguous(), d_sigma.contiguous(), d_dx.contiguous(), g_net, g_tables, hash_ws=scratch)
        return (g_net, g_tables[0], g_tables[1], g_tables[2], g_tables[3]) + (None,) * 8


def field(model, x: Tensor, d: Tensor, t: Tensor, x_deform: Optional[Tensor] = None):
    """(rgb [n,3], sigma [n,1], delta_x [n,3]) of NeuralField('part4') for positions x, unit view directions d and per-sample
    times t [n,1] (already noised by the caller if it wants noise); ``x_deform``: the (noised) positions the deformation
    grids are queried at (default x)."""
    x, d = ops._dev(x, "x"), ops._dev(d, "d")
    t_def = ops._dev(t.reshape(-1), "t")
    if x.shape[0] == 0:
        return x.new_zeros(0, 3), x.new_zeros(0, 1), x.new_zeros(0, 3)
    grids = [getattr(model, g) for g in GRIDS]
    train = torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters())
    rgb, sigma, dx = _Part4Field.apply(flat_from_modules(model), *[g.encoding.params for g in grids], x,
                                       None if x_deform is None else ops._dev(x_deform, "x_deform"), t_def, d,
                                       grids[0].levels, grids[3].levels, float(grids[3].bound), train)
    return rgb, sigma.unsqueeze(-1), dx


# --------------------------------------------------------------------------------------------------- engine
class DualHashEngine:
    """Flat-parameter training / rendering engine of mode part4 (loop body of reference run_part4, run.py:1808-1990):

        batch -> compaction -> t', x' (+ noise) -> 3 deformation hash encodings -> deformation chain -> canonical hash
        encoding -> canonical chain -> compositing + MSE + displacement regulariser + backward (one kernel)
        -> canonical chain bwd + wgrad -> hash input gradient + scatter -> deformation chain bwd + wgrad -> 3 scatters
        -> [regulariser probes through the same kernels] -> [all-reduce] -> TV terms + ONE global-norm clip + AdamW
        (2x lr for the four grids, 5x for displacement_scale, cosine schedule) -> repack.
    """

    def __init__(self, cfg: dict, device: str = "cuda", seed: int = 0, world_size: int = 1):
        why = supported(cfg)
        if why is not None:
            raise NotImplementedError(f"the fused Part 4 chains are not compiled for {why}")
        _check_count()
        self.cfg = dict(cfg)
        self.device = torch.device(device)
        self.seed, self.world_size = int(seed), int(world_size)
        self.bound = float(cfg.get("scene_bound", 1.5))
        self.levels_d = ops.HashLevelTable(cfg.get("deform_n_levels", 14), cfg.get("deform_log2_hashmap_size", 19),
                                           cfg.get("deform_base_resolution", 16), cfg.get("deform_per_level_scale", 1.5))
        self.levels_c = ops.HashLevelTable(cfg.get("n_levels", 16), cfg.get("log2_hashmap_size", 19), cfg.get("base_resolution", 16),
                                           cfg.get("per_level_scale", 1.5))
        nd, nc = self.levels_d.entries * 2, self.levels_c.entries * 2
        self.table_sizes = [nd, nd, nd, nc]
        self.table_offsets = [0, nd, 2 * nd, 3 * nd]
        total = 3 * nd + nc
        g = torch.Generator().manual_seed(seed)
        # ONE flat buffer for the four grids (start | mid | end | canonical): one memset, one AdamW launch.  The buffers are allocated
        # padded to world equal slices of whole 1024-element blocks (the sharded optimiser's reduce-scatter / all-gather); the views
        # below are the tables themselves
        from .sharded import padded_length
        n_pad = padded_length(total, max(self.world_size, 1))
        self._tables_buf = torch.zeros(n_pad, device=self.device)
        self._tables_buf[:total] = ((torch.rand(total, generator=g) * 2 - 1) * 1e-4).to(self.device)
        self._tables_h_buf = torch.zeros(n_pad, dtype=torch.float16, device=self.device)
        self._g_tables_buf = torch.zeros(n_pad, device=self.device)
        self.tables, self.tables_h, self.g_tables = self._tables_buf[:total], self._tables_h_buf[:total], self._g_tables_buf[:total]
        self.shard = None                         # ShardedTableOptimizer once enable_sharded_optimizer() was called
        self.net = torch.zeros(N_PARAMS, device=self.device)
        # network gradients and the step's four scalars (loss, regulariser, squared gradient norm, spare) in one buffer: one fill per step
        self._g_net_scalars = torch.zeros(N_PARAMS + 4, device=self.device)
        self.g_net = self._g_net_scalars[:N_PARAMS]
        self._m_buf, self._v_buf = torch.zeros(n_pad, device=self.device), torch.zeros(n_pad, device=self.device)
        self.state = {"tables": (self._m_buf[:total], self._v_buf[:total]), "net": (torch.zeros_like(self.net), torch.zeros_like(self.net))}
        self.packed = torch.empty(_lib.load().nerf_p4_packed_bytes(), dtype=torch.uint8, device=self.device)
        self.near, self.far = float(cfg.get("near", 2.0)), float(cfg.get("far", 6.0))
        self.lr0, self.eta_min = float(cfg.get("learning_rate", 5e-4)), float(cfg.get("eta_min", 1e-4))
        self.t_max = int(cfg.get("train_iters", 20000))
        self.wd = float(cfg.get("weight_decay", 1e-5))
        self.max_norm = float(cfg.get("max_grad_norm", 1.0))
        self.reg_weight = float(cfg.get("deformation_reg_weight", 0.01))
        self.tv_disp = float(cfg.get("tv_displacement_weight", 0.001)) / 3.0 if cfg.get("use_tv_displacement", True) else 0.0
        self.tv_canon = float(cfg.get("tv_loss_weight", 1e-5))
        noisy = bool(cfg.get("use_coord_noise", False))
        self.std_x = float(cfg.get("coord_noise_std", 0.005)) if noisy else 0.0
        self.std_t = float(cfg.get("time_noise_std", 0.02)) if noisy else 0.0
        self.bg = (torch.ones(3) if cfg.get("white_bkgd", True) else torch.zeros(3)).to(self.device)
        res = int(cfg.get("grid_resolution", 128))
        self.grid_threshold = float(cfg.get("grid_threshold", 0.01))
        self.grid = torch.zeros(res, res, res, device=self.device)
        self.binary_grid = torch.ones(res, res, res, dtype=torch.bool, device=self.device)
        self.step_count = 0
        self._scalars = self._g_net_scalars[N_PARAMS:]
        self._loss_ring, self._grad_calls = torch.zeros(32768, 2, device=self.device), -1
        self._normsq_ws = ops.normsq_ws(self.device)
        assert (3 * nd) % 4 == 0
        self._tv_codes = torch.empty((total + 3) // 4, dtype=torch.uint8, device=self.device)   # two-bit signs of the TV terms
        self._ws: Dict[str, Tensor] = {}
        self._hash_ws = None
        self._hash_ws_tables = None
        self._hash_ws_probes = None
        from .specbwd import SpeculativeScatter
        on = bool(cfg.get("speculative_hash_backward", True)) and not os.environ.get("NERF_NO_SPECULATIVE_BWD")      # env: A/B aid
        self.spec_c, self.spec_d = SpeculativeScatter(on), SpeculativeScatter(on)
        self._counter = 0
        self.repack()

    # -- parameters ------------------------------------------------------------------------------------------
    def table(self, k: int, half: bool = False) -> Tensor:
        src = self.tables_h if half else self.tables
        return src[self.table_offsets[k]:self.table_offsets[k] + self.table_sizes[k]]

    def g_table(self, k: int) -> Tensor:
        return self.g_tables[self.table_offsets[k]:self.table_offsets[k] + self.table_sizes[k]]

    def repack(self) -> None:
        pack(self.net, self.packed)
        ops.f32_to_f16(self.tables, self.tables_h)

    def load_from_model(self, model) -> None:
        sd = dict(model.named_parameters())
        with torch.no_grad():
            for key, off, cnt in MODULE_SLICES:
                self.net[off:off + cnt].copy_(sd[key].reshape(-1))
            for k, name in enumerate(GRIDS):
                self.table(k).copy_(getattr(model, name).encoding.params)
        self.repack()

    def copy_to_model(self, model) -> None:
        sd = dict(model.named_parameters())
        with torch.no_grad():
            for key, off, cnt in MODULE_SLICES:
                sd[key].copy_(self.net[off:off + cnt].view(sd[key].shape))
            for k, name in enumerate(GRIDS):
                getattr(model, name).encoding.params.copy_(self.table(k))

    def lr(self, mult: float = 1.0) -> float:
        """CosineAnnealingLR of a group whose initial rate is mult * learning_rate (run.py:1684-1743)"""
        base = self.lr0 * mult
        return self.eta_min + (base - self.eta_min) * (1 + math.cos(math.pi * self.step_count / self.t_max)) / 2

    def _workspace(self, n: int, which: str = "batch") -> Workspace:
        """the step's workspace laid out in ONE grow-only buffer per use (data batch / regulariser probes): the active-point
        count changes almost every step, a buffer per count would churn hundreds of MB through the allocator"""
        need = Workspace.bytes(n)
        buf = self._ws.get(which)
        if buf is None or buf.numel() < need:
            self._ws.pop(which, None)                        # release before growing
            buf = self._ws[which] = torch.empty(int(need * 1.25), dtype=torch.uint8, device=self.device)
        return Workspace(n, self.device, buf=buf)

    def _hash_scratch_tables(self, n: int, n_levels: int, n_tables: int) -> Tensor:
        need = _lib.load().nerf_hash_encode_bwd_tables_workspace_bytes(n, n_levels, n_tables)
        if self._hash_ws_tables is None or self._hash_ws_tables.numel() < need:
            self._hash_ws_tables = None
            self._hash_ws_tables = torch.empty(int(need * 1.25), dtype=torch.uint8, device=self.device)
        return self._hash_ws_tables

    def _hash_scratch_probes(self, n: int, n_levels: int) -> Tensor:
        """the probes' (accumulating, counted) scatters: their own workspace -- the data batch's keep their bin estimates"""
        need = ops.hash_encode_bwd_workspace_bytes(n, n_levels)
        if self._hash_ws_probes is None or self._hash_ws_probes.numel() < need:
            self._hash_ws_probes = None
            self._hash_ws_probes = torch.empty(int(need * 1.25), dtype=torch.uint8, device=self.device)
        return self._hash_ws_probes

    def _hash_scratch(self, n: int, n_levels: int) -> Tensor:
        need = ops.hash_encode_bwd_workspace_bytes(n, n_levels)
        if self._hash_ws is None or self._hash_ws.numel() < need:
            self._hash_ws = None
            self._hash_ws = torch.empty(int(need * 1.25), dtype=torch.uint8, device=self.device)
        return self._hash_ws

    def _tables_for_forward(self):
        return [self.table(k, half=True).view(-1, 2) for k in range(4)]

    # -- field -----------------------------------------------------------------------------------------------
    def prepare_batch(self, rays_o: Tensor, rays_d: Tensor, n_samples: int, first_ray: int = 0):
        self._counter += 1
        return ops.sample_compact_async(rays_o, rays_d, self.near, self.far, n_samples, self.binary_grid, self.bound,
                                        jitter=(self.seed, self._counter), first_ray=first_ray), self._counter

    def compute_gradients(self, rays_o: Tensor, rays_d: Tensor, target: Tensor, times: Tensor, n_samples: int, prepared=None,
                          first_ray: int = 0, bg: Optional[Tensor] = None, sync_grads_async=None, probes=None,
                          shard_grads: bool = False) -> Tensor:
        """Forward + backward of one batch: fills g_net / g_tables with the gradient of
        MSE + deformation_reg_weight * mean(mean_delta_x^2) (+ the probe regularisers) of the LOCAL rays; returns the RGB loss.
        ``sync_grads_async(view)``: data-parallel hook, called with gradient ranges as they become final."""
        R = rays_o.shape[0]
        prepared, counter = prepared if prepared is not None else self.prepare_batch(rays_o, rays_d, n_samples, first_ray)
        z, slots, pts, dirs = prepared.get()
        n = pts.shape[0]
        bg = self.bg if bg is None else bg
        self._g_net_scalars.zero_()
        # a fresh zeroed (loss, regulariser) slot per step out of a ring cleared half a lap ahead: the returned loss is a VIEW of its
        # slot (valid for the next 16384 steps) -- no copy launch per step
        self._grad_calls += 1
        ring = self._loss_ring.shape[0]
        slot = self._grad_calls % ring
        if slot % (ring // 2) == 0 and self._grad_calls > 0:
            self._loss_ring[slot:slot + ring // 2].zero_()
        loss, reg = self._loss_ring[slot, 0:1], self._loss_ring[slot, 1:2]
        self.last_reg = reg[0]                 # displacement regulariser of this batch (before its weight), a view like the loss
        handles = []
        reduce = (lambda view: handles.append(sync_grads_async(view))) if sync_grads_async is not None else (lambda view: None)
        if n == 0:
            self.g_tables.zero_()
            loss = ((bg.expand(R, 3) - target) ** 2).mean().reshape(1)
        else:
            lib = _lib.load()
            ws = self._workspace(n)
            x_def, t_def = sample_inputs(slots, pts, times, R, n_samples, self.std_x, self.std_t, self.seed, counter, first_ray)
            rgb, sigma, dx, xc = forward_chain(self.packed, self.net, self._tables_for_forward(), self.levels_d, self.levels_c, self.bound,
                                               pts, x_def, t_def, dirs, ws, True)
            d_rgb, d_sigma, d_extra = torch.empty_like(rgb), torch.empty_like(sigma), torch.empty_like(dx)
            _lib.check(lib.nerf_composite_mse_reg_bwd(P(rgb), P(sigma), P(slots), P(z), P(rays_d), P(bg), 1, P(target), 1.0 / (3 * R),
                                                      P(dx), self.reg_weight / (3 * R), R, n_samples, None, None, P(loss), P(reg),
                                                      P(d_rgb), P(d_sigma), P(d_extra), P(ops.sum_ws(self.device)), ops._stream()),
                       "nerf_composite_mse_reg_bwd")
            g_tabs = [self.g_table(k) for k in range(4)]
            backward_chain(self.packed, self.net, self.table(3, half=True), self.levels_d, self.levels_c, self.bound, pts if x_def is None else x_def,
                           xc, ws, rgb, sigma, d_rgb, d_sigma, d_extra, self.g_net, g_tabs, hash_ws=self._hash_scratch, overwrite=True, tables_ws=self._hash_scratch_tables, extra_in_place=True,
                           spec=(self.spec_c, self.spec_d, (self.binary_grid.data_ptr(), self.binary_grid._version)),
                           after_grid=(lambda k: reduce(g_tabs[k])) if (sync_grads_async is not None and probes is None and not shard_grads) else None)
        if probes is not None:
            self._probe_regularisers(probes)
        if shard_grads and self.shard is not None:
            # sharded optimiser: ONE reduce-scatter of the flat table gradient (every rank ends with the summed gradient of its
            # slice) + the networks' small all-reduce; the same two collectives on every rank whatever its shard held
            import torch.distributed as dist
            self.shard.reduce_scatter_grads()
            if self.world_size > 1:
                dist.all_reduce(self.g_net, op=dist.ReduceOp.SUM)
        elif sync_grads_async is not None:
            if n == 0 or probes is not None:
                for k in (3, 0, 1, 2):                       # the order and sizes of the busy ranks' collectives
                    reduce(self.g_table(k))
            reduce(self.g_net)
            for h in handles:
                if h is not None:
                    h.wait()
        return loss[0]

    def _probe_regularisers(self, probes) -> None:
        """Temporal smoothness, unsupervised consistency and tri-grid anchor terms of the reference's loop (run.py:1861-1938)
        on their random probe points, through the SAME kernels as the data batch: one small batch of (x, t, grid weights)
        rows, deformation chain forward with stash, the terms' gradients w.r.t. delta_x as elementwise arithmetic, deformation
        chain backward accumulating into g_net / g_tables.  ``probes``: dict with any of
        temporal = (x [m,3], t [m,1], eps, weight), unsup = (x, t, weight), anchor = (x, weight)."""
        dev = self.device
        rows_x, rows_t, rows_w = [], [], []
        e = lambda k, m: torch.eye(3, device=dev)[k].expand(m, 3)
        spans = {}

        def add(name, x, t, grid):
            spans[name] = (sum(r.shape[0] for r in rows_x), x.shape[0])
            rows_x.append(x); rows_t.append(t.reshape(-1)); rows_w.append(e(grid, x.shape[0]))
        if "temporal" in probes:
            x, t, eps, _ = probes["temporal"]
            add("tmp0", x, t, 0); add("tmp1", x, t + eps, 0)
        if "unsup" in probes:
            x, t, _ = probes["unsup"]
            add("unsup", x, t, 0)
        if "anchor" in probes:
            x, _ = probes["anchor"]
            m = x.shape[0]
            add("at0", x, torch.zeros(m, device=dev), 0)
            add("a_start", x, torch.full((m,), 1.0 / 6.0, device=dev), 0)
            add("a_mid", x, torch.full((m,), 1.0 / 6.0, device=dev), 1)
        if not rows_x:
            return
        X, Tm, W = torch.cat(rows_x).contiguous(), torch.cat(rows_t).contiguous(), torch.cat(rows_w).contiguous()
        n = X.shape[0]
        ws = self._workspace(n, "probes")
        _, _, dx, _ = forward_chain(self.packed, self.net, self._tables_for_forward(), self.levels_d, self.levels_c, self.bound,
                                    X, None, Tm, None, ws, True, blend=W)
        g = torch.zeros_like(dx)
        sl = lambda name: slice(spans[name][0], spans[name][0] + spans[name][1])
        if "temporal" in probes:
            w = probes["temporal"][3] * 16
            diff = dx[sl("tmp0")] - dx[sl("tmp1")]
            g[sl("tmp0")] += 2 * w * diff / diff.numel()
            g[sl("tmp1")] -= 2 * w * diff / diff.numel()
        if "unsup" in probes:
            w = probes["unsup"][2] * 32
            d = dx[sl("unsup")]
            g[sl("unsup")] += (w / 3.0) * torch.sign(d.mean(dim=0, keepdim=True)).expand_as(d) / d.shape[0]
        if "anchor" in probes:
            w = probes["anchor"][1] * 16
            a0, a1, a2 = dx[sl("at0")], dx[sl("a_start")], dx[sl("a_mid")]
            g[sl("at0")] += 2 * w * a0 / a0.numel()
            g[sl("a_start")] += 2 * w * 0.1 * (a1 - a2) / a1.numel()
            g[sl("a_mid")] -= 2 * w * 0.1 * (a1 - a2) / a1.numel()
        backward_chain(self.packed, self.net, self.table(3), self.levels_d, self.levels_c, self.bound, X, None, ws, None, None, None, None,
                       g, self.g_net, [self.g_table(k) for k in range(4)], hash_ws=self._hash_scratch_probes)

    def enable_sharded_optimizer(self, rank: int) -> None:
        """Data parallelism as SURVEY 8(e) specifies for the big tables (project-nerf_amd/sharded.py): the table gradient is
        reduce-scattered, this rank steps its 1/world slice of the flat table buffer, the fp16 copy is all-gathered.  Call once,
        after the replicas' parameters were made equal; compute_gradients(shard_grads=True) + apply_gradients() then take this path."""
        from .sharded import ShardedTableOptimizer
        nd, nc = self.table_sizes[0], self.table_sizes[3]
        tabs = [(k * nd, nd, self.tv_disp) for k in range(3)] + [(3 * nd, nc, self.tv_canon)]
        self.shard = ShardedTableOptimizer(tabs, self.tables.numel(), rank, self.world_size, self._tables_buf, self._g_tables_buf,
                                           self._m_buf, self._v_buf, self._tables_h_buf)

    def gather_master(self) -> None:
        """sharded optimiser: bring the fp32 master copy of every slice up to date on this rank (checkpoints, validation)"""
        if self.shard is not None:
            self.shard.gather_master()

    def _apply_gradients_sharded(self) -> None:
        """apply_gradients with the table groups sharded over the ranks: this rank's slice of TV + norm and of clip + AdamW, ONE
        scalar all-reduce for the squared norm (the networks' part is added by rank 0), the networks stepped on every rank"""
        import torch.distributed as dist
        lib, st, sh = _lib.load(), ops._stream(), self.shard
        scale = 1.0 / self.world_size
        normsq = self._normsq_ws
        sh.accumulate_normsq(normsq, scale, first=True)         # the first piece stores the norm: no zeroing launch
        if sh.rank == 0:
            _lib.check(lib.nerf_tv_normsq_codes(P(self.net), P(self.g_net), N_PARAMS, 1, 0.0, scale, P(normsq), 1 if sh.pieces else 0, None, st),
                       "nerf_tv_normsq_codes")
        if self.world_size > 1:
            dist.all_reduce(normsq[0:1], op=dist.ReduceOp.SUM)      # every rank: the same bits, hence the same clip coefficient
        lr_t, lr_n, lr_s = self.lr(2.0), self.lr(1.0), self.lr(5.0)
        self.step_count += 1
        step = self.step_count
        sh.adamw(normsq, step, lr_t, self.wd, self.max_norm, scale)
        m, v = self.state["net"]
        _lib.check(lib.nerf_adamw_clip_step_tv(P(self.net), P(self.g_net), P(m), P(v), N_PARAMS, step, lr_n, 0.9, 0.999, 1e-8, self.wd,
                                               P(normsq), self.max_norm, scale, None, 0, 0.0, 0, 0.0, 0, SCALE, lr_s, None, st),
                   "nerf_adamw_clip_step_tv")
        sh.exchange()                             # neighbours' edge elements (next step's TV terms), fp16 copy of every slice
        pack(self.net, self.packed)

    def apply_gradients(self) -> None:
        """TV-L1 on the four grids, ONE global-norm clip over every parameter (clip_grad_norm_(model.parameters()),
        run.py:1943), AdamW with the reference's group rates and cosine schedule; after a summing all-reduce the data
        gradient is averaged (1/world), the TV terms are added unscaled."""
        if self.shard is not None:
            return self._apply_gradients_sharded()
        lib = _lib.load()
        st = ops._stream()
        scale = 1.0 / self.world_size
        normsq = self._normsq_ws                 # [0] the squared norm of ALL groups, [1] ticket, [2:] partials (include/nerf_hip.h)
        # pass 1 (three launches): ONE squared norm over all groups; the TV terms' signs go to a two-bit code per table entry instead
        # of into the gradient (38.5 instead of 42 bytes per parameter).  The three deformation grids (equal sizes, back to back in
        # the flat buffer) in one launch, each with its own total variation
        n_def, n_can, total = self.table_sizes[0], self.table_sizes[3], self.tables.numel()
        codes = self._tv_codes
        _lib.check(lib.nerf_tv_normsq_codes(P(self.table(0)), P(self.g_table(0)), 3 * n_def, 3, self.tv_disp, scale, P(normsq), 0, P(codes), st),
                   "nerf_tv_normsq_codes")          # the first group STORES the norm (no zeroing launch), the others add
        _lib.check(lib.nerf_tv_normsq_codes(P(self.table(3)), P(self.g_table(3)), n_can, 1, self.tv_canon, scale, P(normsq), 1,
                                            P(codes[3 * n_def // 4:]), st), "nerf_tv_normsq_codes")
        _lib.check(lib.nerf_tv_normsq_codes(P(self.net), P(self.g_net), N_PARAMS, 1, 0.0, scale, P(normsq), 1, None, st), "nerf_tv_normsq_codes")
        lr_t, lr_n, lr_s = self.lr(2.0), self.lr(1.0), self.lr(5.0)      # the rates of THIS step: scheduler.step() follows optimizer.step()
        self.step_count += 1
        step = self.step_count
        # pass 2 (two launches): the four grids (TV weights per range, fp16 copy written), the networks + displacement_scale (its own rate)
        m, v = self.state["tables"]
        _lib.check(lib.nerf_adamw_clip_step_tv(P(self.tables), P(self.g_tables), P(m), P(v), total, step, lr_t, 0.9, 0.999, 1e-8, self.wd,
                                               P(normsq), self.max_norm, scale, P(codes), 3 * n_def, self.tv_disp, n_def, self.tv_canon, n_can,
                                               0, 0.0, P(self.tables_h), st), "nerf_adamw_clip_step_tv")
        m, v = self.state["net"]
        _lib.check(lib.nerf_adamw_clip_step_tv(P(self.net), P(self.g_net), P(m), P(v), N_PARAMS, step, lr_n, 0.9, 0.999, 1e-8, self.wd,
                                               P(normsq), self.max_norm, scale, None, 0, 0.0, 0, 0.0, 0, SCALE, lr_s, None, st),
                   "nerf_adamw_clip_step_tv")
        pack(self.net, self.packed)

    def train_step(self, rays_o, rays_d, target, times, n_samples, prepared=None, first_ray: int = 0, bg=None, sync_grads_async=None,
                   probes=None) -> Tensor:
        loss = self.compute_gradients(rays_o, rays_d, target, times, n_samples, prepared=prepared, first_ray=first_ray, bg=bg,
                                      sync_grads_async=sync_grads_async, probes=probes, shard_grads=self.shard is not None)
        self.apply_gradients()
        return loss

    # -- occupancy grid / rendering ----------------------------------------------------------------------------
    @torch.no_grad()
    def field(self, pts: Tensor, dirs: Tensor, t: Tensor):
        """(rgb, sigma, delta_x) at points with per-point times, evaluation mode (no noise)"""
        n = pts.shape[0]
        ws = self._workspace(n)
        rgb, sigma, dx, _ = forward_chain(self.packed, self.net, self._tables_for_forward(), self.levels_d, self.levels_c, self.bound,
                                          pts.contiguous(), None, t.reshape(-1).contiguous(), dirs.contiguous(), ws, False)
        return rgb, sigma, dx

    @torch.no_grad()
    def update_grid(self, decay: float = 0.95) -> float:
        """DensityGrid.update of mode part4 (reference src/renderer.py:65-86, 122-125): density on the lattice at the time
        anchors 0, 0.5, 1, element-wise maximum, running maximum against the decayed history."""
        res = self.grid.shape[0]
        pts = ops.grid_lattice(self.bound, res, self.device)
        sig = torch.zeros(res ** 3, device=self.device)
        zeros = torch.zeros(2 ** 18, 3, device=self.device)
        for anchor in (0.0, 0.5, 1.0):
            for i in range(0, pts.shape[0], 2 ** 18):
                p = pts[i:i + 2 ** 18]
                s = self.field(p, zeros[:p.shape[0]], torch.full((p.shape[0],), anchor, device=self.device))[1]
                torch.maximum(sig[i:i + 2 ** 18], s, out=sig[i:i + 2 ** 18])
        self.binary_grid, ratio = ops.grid_threshold(sig.view(res, res, res), self.grid_threshold, prev=self.grid, decay=decay)
        return ratio

    @torch.no_grad()
    def render_rays(self, rays_o: Tensor, rays_d: Tensor, times: Tensor, n_samples: int, bg: Optional[Tensor] = None):
        z, slots, pts, dirs = ops.sample_compact(rays_o, rays_d, self.near, self.far, n_samples, self.binary_grid, self.bound)
        R = rays_o.shape[0]
        bg = self.bg if bg is None else bg
        if pts.shape[0] == 0:
            return bg.expand(R, 3).clone(), torch.zeros(R, device=self.device), torch.zeros(R, device=self.device)
        _, t_def = sample_inputs(slots, pts, times.expand(R, 1) if times.numel() == 1 else times, R, n_samples)
        ws = self._workspace(pts.shape[0])
        rgb, sigma, _, _ = forward_chain(self.packed, self.net, self._tables_for_forward(), self.levels_d, self.levels_c, self.bound, pts,
                                         None, t_def, dirs, ws, False)
        return ops.composite_indexed(rgb, sigma, slots, z, rays_d, bg)

    @torch.no_grad()
    def render_image(self, rays_o: Tensor, rays_d: Tensor, time: Tensor, n_samples: int, chunk: int = 65536) -> Tensor:
        shape = rays_o.shape[:-1]
        o, d = rays_o.reshape(-1, 3).contiguous(), rays_d.reshape(-1, 3).contiguous()
        out = torch.empty(o.shape[0], 3, device=self.device)
        for i in range(0, o.shape[0], chunk):
            out[i:i + chunk] = self.render_rays(o[i:i + chunk], d[i:i + chunk], time.reshape(1, 1).to(self.device), n_samples)[0]
        return out.view(*shape, 3)
