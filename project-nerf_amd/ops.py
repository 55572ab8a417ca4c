"""Tensor-level entry points of the HIP hot path.

PyTorch is used for device memory, streams and autograd bookkeeping only; all arithmetic
happens in libnerf_hip.so through the C ABI (include/nerf_hip.h).  Every function insists on
contiguous fp32 tensors that live on a HIP device and raises otherwise -- there is no CPU
or eager fallback.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib

Tensor = torch.Tensor


def _stream() -> int:
    """raw HIP stream of torch's current stream on the current device.  torch.cuda.current_stream() costs ~8 us of Python per
    call (device-index plumbing, availability checks) and a step makes dozens: the C accessor torch itself uses instead"""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _dev(t: Tensor, name: str, dtype=torch.float32) -> Tensor:
    if not isinstance(t, Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if t.device.type != "cuda":
        raise _lib.NerfHipError(f"{name} is on {t.device}; the hot path only runs on a HIP device (no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _p(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# --------------------------------------------------------------------------- a1-a3
def sample_rays(rays_o: Tensor, rays_d: Tensor, near: float, far: float, n_samples: int,
                u: Optional[Tensor] = None, want_points: bool = False):
    """z[R,S] (+ pts[R*S,3], dirs[R*S,3]); ``u`` [R,S] is the jitter draw or None."""
    lib = _lib.load()
    rays_o, rays_d = _dev(rays_o, "rays_o"), _dev(rays_d, "rays_d")
    R = rays_o.shape[0]
    if u is not None:
        u = _dev(u, "u")
        if tuple(u.shape) != (R, n_samples):
            raise ValueError(f"u must be [{R},{n_samples}]")
    z = torch.empty(R, n_samples, device=rays_o.device, dtype=torch.float32)
    pts = dirs = None
    if want_points:
        pts = torch.empty(R * n_samples, 3, device=rays_o.device, dtype=torch.float32)
        dirs = torch.empty_like(pts)
    _lib.check(lib.nerf_sample_rays(_p(rays_o), _p(rays_d), _p(u), R, n_samples, near, far,
                                    _p(z), _p(pts), _p(dirs), _stream()), "nerf_sample_rays")
    return (z, pts, dirs) if want_points else z


def active_mask(pts: Tensor, binary_grid: Tensor, bound: float, want_index: bool = False):
    lib = _lib.load()
    pts = _dev(pts, "pts")
    grid = _dev(binary_grid, "binary_grid", torch.bool)
    res = grid.shape[0]
    n = pts.shape[0]
    mask = torch.empty(n, device=pts.device, dtype=torch.bool)
    idx = torch.empty(n, 3, device=pts.device, dtype=torch.int64) if want_index else None
    _lib.check(lib.nerf_active_mask(_p(pts), n, _p(grid), res, float(bound), _p(mask), _p(idx), _stream()),
               "nerf_active_mask")
    return (mask, idx) if want_index else mask


# --------------------------------------------------------------------------- f1 batch sampling
def gather_rays(images: Tensor, poses: Tensor, img_idx: Tensor, pix_y: Tensor, pix_x: Tensor, focal: float,
                scene_scale: float = 1.0) -> Tuple[Tensor, Tensor, Tensor]:
    """Rays and RGBA targets of the drawn pixels from device-resident frames (one kernel)."""
    lib = _lib.load()
    images, poses = _dev(images, "images"), _dev(poses, "poses")
    n_img, H, W, _ = images.shape
    B = img_idx.numel()
    idx = [_dev(t.to(torch.int64), "index", torch.int64) for t in (img_idx, pix_y, pix_x)]
    o = torch.empty(B, 3, device=images.device)
    d = torch.empty(B, 3, device=images.device)
    rgba = torch.empty(B, 4, device=images.device)
    _lib.check(lib.nerf_gather_rays(_p(images), _p(poses), _p(idx[0]), _p(idx[1]), _p(idx[2]), B, n_img, H, W, float(focal),
                                    float(scene_scale), _p(o), _p(d), _p(rgba), _stream()), "nerf_gather_rays")
    return o, d, rgba


def gather_batch(images: Tensor, poses: Tensor, flat_idx: Tensor, focal: float, scene_scale: float = 1.0,
                 bg: Optional[Tensor] = None, want_rgba: bool = False):
    """One draw per ray over all pixels of all frames -> (rays_o, rays_d, target or None, rgba or None);
    ``bg`` [3] asks for the composited training target rgb * a + bg * (1 - a)."""
    lib = _lib.load()
    images, poses = _dev(images, "images"), _dev(poses, "poses")
    n_img, H, W, _ = images.shape
    flat_idx = _dev(flat_idx, "flat_idx", torch.int64)
    B = flat_idx.numel()
    o, d = torch.empty(B, 3, device=images.device), torch.empty(B, 3, device=images.device)
    rgba = torch.empty(B, 4, device=images.device) if (want_rgba or bg is None) else None
    target = torch.empty(B, 3, device=images.device) if bg is not None else None
    _lib.check(lib.nerf_gather_batch(_p(images), _p(poses), _p(flat_idx), B, n_img, H, W, float(focal), float(scene_scale),
                                     _p(None if bg is None else _dev(bg, "bg")), _p(o), _p(d), _p(rgba), _p(target), _stream()),
               "nerf_gather_batch")
    return o, d, target, rgba


def train_batch(images: Tensor, poses: Tensor, focal: float, batch: int, n_samples: int, near: float, far: float,
                seed: int, counter: int, bg: Optional[Tensor] = None, scene_scale: float = 1.0, perturb: bool = True,
                want_rgba: bool = False, first_ray: int = 0):
    """The data side of a training step as ONE kernel: (rays_o, rays_d, target or rgba, z).  Pixels and
    stratified jitter come from a counter-based generator keyed by (seed, counter).  ``first_ray``: this call forms
    rays [first_ray, first_ray + batch) of a larger batch (data-parallel shards of one global batch)."""
    lib = _lib.load()
    images, poses = _dev(images, "images"), _dev(poses, "poses")
    n_img, H, W, _ = images.shape
    dev = images.device
    o, d = torch.empty(batch, 3, device=dev), torch.empty(batch, 3, device=dev)
    z = torch.empty(batch, n_samples, device=dev)
    rgba = torch.empty(batch, 4, device=dev) if (want_rgba or bg is None) else None
    target = torch.empty(batch, 3, device=dev) if bg is not None else None
    _lib.check(lib.nerf_train_batch_shard(_p(images), _p(poses), n_img, H, W, float(focal), float(scene_scale),
                                          _p(None if bg is None else _dev(bg, "bg")), int(seed), int(counter), int(first_ray), batch,
                                          n_samples, float(near), float(far), 1 if perturb else 0, _p(o), _p(d), _p(rgba), _p(target),
                                          _p(z), _stream()), "nerf_train_batch")
    return o, d, (target if bg is not None else rgba), z


# --------------------------------------------------------------------------- deterministic mode
_DETERMINISTIC = None
_SUM_WS = {}


def set_deterministic(on: bool = True) -> None:
    """Library option "deterministic" (include/nerf_hip.h): every sum whose order would depend on scheduling takes an
    ordered form, so two runs of the same training step give the same bits (YAML key ``deterministic: true`` in run.py;
    environment NERF_DETERMINISTIC=1).  The wrappers below route compaction through nerf_sample_compact_ordered, give the
    hash scatter a workspace and the fused compositing kernels their ordered-sum scratch."""
    global _DETERMINISTIC
    _lib.set_option("deterministic", 1 if on else 0)
    _DETERMINISTIC = bool(on)


def deterministic() -> bool:
    global _DETERMINISTIC
    if _DETERMINISTIC is None:
        _DETERMINISTIC = bool(_lib.get_option("deterministic"))
    return _DETERMINISTIC


def sum_ws(device) -> Optional[Tensor]:
    """scratch of the loss / regulariser partial sums of nerf_composite_mse*_bwd (summed in workgroup order by a one-workgroup
    launch: no atomics, the same bits every run); one per device: launches that share it must be stream-ordered"""
    key = torch.device(device)
    if key not in _SUM_WS:
        _SUM_WS[key] = torch.zeros(_lib.SUM_WS_FLOATS, device=key)
    return _SUM_WS[key]


def normsq_ws(device) -> Tensor:
    """workspace of nerf_tv_normsq*: [0] the squared norm, [1] ticket, [2:] one partial per workgroup"""
    return torch.zeros(_lib.NORMSQ_WS_FLOATS, device=device)


_COMPACT_CHAIN = {}
_NO_CHAIN = bool(__import__("os").environ.get("NERF_NO_COMPACT_CHAIN"))      # A/B aid: the copy + event form instead


class _CompactChain:
    """per (device, stream): the chained compaction's device state (two counters used alternately + tickets) and a ring of
    host-mapped pinned (count, seq) pairs the kernels publish their counts into"""
    RING = 16

    def __init__(self, device):
        self.state = torch.zeros(_lib.COMPACT_CHAIN_WORDS, device=device, dtype=torch.int32)
        self.host = torch.zeros(self.RING, 2, dtype=torch.int32).pin_memory()
        self.host_np = self.host.numpy()          # polled without tensor overhead
        self.turn, self.seq = 0, 0
        self.owner = [None] * self.RING           # weak references to the results whose counts sit in the ring's slots

    def next(self):
        self.seq += 1
        turn, self.turn = self.turn, self.turn ^ 1
        slot = self.seq % self.RING
        prev = self.owner[slot]() if self.owner[slot] is not None else None
        if prev is not None and not prev._read:
            raise _lib.NerfHipError(f"sample_compact_async: more than {self.RING} batches prepared ahead without reading their counts")
        self.host_np[slot, 1] = 0                 # (a seq is never 0)
        return turn, slot, self.seq & 0x7FFFFFFF or 1


def _compact_launch(lib, rays_o, rays_d, u, jitter, first_ray, R, n_samples, near, far, grid, bound, z, slots, pts, dirs, count):
    """the compaction launch of sample_compact / sample_compact_async: ordered slots when deterministic.  ``count`` None (the
    asynchronous path with in-kernel jitter): nerf_sample_compact_jitter_chain -- no fill launch, no copy launch, no event: returns
    (host block as numpy [2], seq) to poll; otherwise returns the device counter the call used."""
    draw = u is None and jitter is not None
    seed, counter = (int(jitter[0]), int(jitter[1]) & 0xFFFFFF) if draw else (0, 0)
    if count is None:
        if draw and not deterministic() and not _NO_CHAIN:
            key = (rays_o.device, _stream())
            chain = _COMPACT_CHAIN.get(key)
            if chain is None:
                chain = _COMPACT_CHAIN[key] = _CompactChain(rays_o.device)
            turn, slot, seq = chain.next()
            _lib.check(lib.nerf_sample_compact_jitter_chain(_p(rays_o), _p(rays_d), seed, counter, int(first_ray), R, n_samples, near, far,
                                                            _p(grid), grid.shape[0], float(bound), _p(z), _p(slots), _p(pts), _p(dirs),
                                                            _p(chain.state), turn, chain.host[slot].data_ptr(), seq, _stream()),
                       "nerf_sample_compact_jitter_chain")
            return chain.host_np[slot], seq, chain, slot
        count = torch.empty(1, device=rays_o.device, dtype=torch.int32)          # cleared by the library call itself
    if deterministic():
        scratch = torch.empty(max(lib.nerf_sample_compact_ordered_scratch_bytes(R, n_samples), 4), dtype=torch.uint8, device=rays_o.device)
        _lib.check(lib.nerf_sample_compact_ordered(_p(rays_o), _p(rays_d), _p(u), 1 if draw else 0, seed, counter, int(first_ray), R, n_samples,
                                                   near, far, _p(grid), grid.shape[0], float(bound), _p(z), _p(slots), _p(pts), _p(dirs),
                                                   _p(count), _p(scratch), scratch.numel(), _stream()), "nerf_sample_compact_ordered")
    elif draw:
        _lib.check(lib.nerf_sample_compact_jitter_shard(_p(rays_o), _p(rays_d), seed, counter, int(first_ray), R, n_samples, near, far,
                                                        _p(grid), grid.shape[0], float(bound), _p(z), _p(slots), _p(pts), _p(dirs), _p(count),
                                                        _stream()), "nerf_sample_compact_jitter")
    else:
        _lib.check(lib.nerf_sample_compact(_p(rays_o), _p(rays_d), _p(u), R, n_samples, near, far, _p(grid), grid.shape[0],
                                           float(bound), _p(z), _p(slots), _p(pts), _p(dirs), _p(count), _stream()),
                   "nerf_sample_compact")
    return count


# --------------------------------------------------------------------------- a1-a4 fused
def sample_compact(rays_o: Tensor, rays_d: Tensor, near: float, far: float, n_samples: int,
                   binary_grid: Tensor, bound: float, u: Optional[Tensor] = None):
    """z [R,S], slot_of_sample [R*S] (int32, -1 = skipped), pts [n_act,3], dirs [n_act,3].
    One host read of the active count (the reference syncs at the same place: ``.any()`` and the
    boolean-index gathers, renderer.py:309-323)."""
    lib = _lib.load()
    rays_o, rays_d = _dev(rays_o, "rays_o"), _dev(rays_d, "rays_d")
    grid = _dev(binary_grid, "binary_grid", torch.bool)
    R = rays_o.shape[0]
    n = R * n_samples
    if u is not None:
        u = _dev(u, "u")
    z = torch.empty(R, n_samples, device=rays_o.device)
    slots = torch.empty(n, device=rays_o.device, dtype=torch.int32)
    pts = torch.empty(max(n, 1), 3, device=rays_o.device)
    dirs = torch.empty(max(n, 1), 3, device=rays_o.device)
    count = torch.empty(1, device=rays_o.device, dtype=torch.int32)          # the call clears it
    _compact_launch(lib, rays_o, rays_d, u, None, 0, R, n_samples, near, far, grid, bound, z, slots, pts, dirs, count)
    n_act = int(count.item())
    return z, slots, pts[:n_act], dirs[:n_act]


class CompactedSamples:
    """Result of ``sample_compact_async``: the kernel is queued, nothing has been waited for.  ``get()`` waits for the active count
    alone and returns (z, slots, pts[:n], dirs[:n]) -- by then usually long there, because the caller queued this batch's
    compaction ahead of the previous step's kernels.  The count arrives either in a host-mapped block the kernel's last workgroup
    writes (chained form: polled) or through a 4-byte copy + event queued behind the kernel."""

    def __init__(self, z, slots, pts, dirs, count_host, event, seq=None):
        self.z, self.slots, self._pts, self._dirs, self._count, self._event, self._seq = z, slots, pts, dirs, count_host, event, seq
        self._read, self._n = False, None

    def get(self):
        if self._n is not None:
            return self.z, self.slots, self._pts[:self._n], self._dirs[:self._n]
        if self._seq is not None:
            block, spins = self._count, 0
            while int(block[1]) != self._seq:         # host-mapped (count, seq): seq is written last, behind a system-scope fence
                spins += 1
                if spins > 20_000_000:
                    raise _lib.NerfHipError("sample_compact_async: the compaction kernel never published its count")
            n = int(block[0])
        else:
            self._event.synchronize()
            n = int(self._count[0])
        self._read, self._n = True, n             # (the ring slot may be reused from here on)
        return self.z, self.slots, self._pts[:n], self._dirs[:n]


def sample_compact_async(rays_o: Tensor, rays_d: Tensor, near: float, far: float, n_samples: int,
                         binary_grid: Tensor, bound: float, u: Optional[Tensor] = None,
                         jitter: Optional[Tuple[int, int]] = None, first_ray: int = 0) -> CompactedSamples:
    """``sample_compact`` without the host wait: same kernel, same outputs; the count comes back asynchronously.
    ``jitter=(seed, counter)`` (with ``u`` None): the stratified jitter is drawn inside the kernel (nerf_sample_compact_jitter)
    instead of being read from a [R, S] tensor of uniforms; ``first_ray``: these rays are rays [first_ray, first_ray + R)
    of a larger batch (data-parallel shards of one global batch draw the jitter one GPU would draw)."""
    lib = _lib.load()
    rays_o, rays_d = _dev(rays_o, "rays_o"), _dev(rays_d, "rays_d")
    grid = _dev(binary_grid, "binary_grid", torch.bool)
    R = rays_o.shape[0]
    n = R * n_samples
    if u is not None:
        u = _dev(u, "u")
    dev = rays_o.device
    z = torch.empty(R, n_samples, device=dev)
    slots = torch.empty(n, device=dev, dtype=torch.int32)
    pts, dirs = torch.empty(max(n, 1), 3, device=dev), torch.empty(max(n, 1), 3, device=dev)
    count = _compact_launch(lib, rays_o, rays_d, u, jitter, first_ray, R, n_samples, near, far, grid, bound, z, slots, pts, dirs, None)
    if isinstance(count, tuple):                                       # chained form: (host block, seq) to poll
        import weakref
        out = CompactedSamples(z, slots, pts, dirs, count[0], None, seq=count[1])
        count[2].owner[count[3]] = weakref.ref(out)
        return out
    count_host = torch.empty(1, dtype=torch.int32, pin_memory=True)
    count_host.copy_(count, non_blocking=True)
    event = torch.cuda.Event()
    event.record()
    return CompactedSamples(z, slots, pts, dirs, count_host, event)


class _CompositeIndexed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb_c, sigma_c, slots, z, rays_d, bg):
        lib = _lib.load()
        R, S = z.shape
        rgb_c, sigma_c = _dev(rgb_c, "rgb_c"), _dev(sigma_c, "sigma_c")
        out_rgb = torch.empty(R, 3, device=z.device)
        depth = torch.empty(R, device=z.device)
        acc = torch.empty(R, device=z.device)
        bg_rows = 0 if bg is None else (1 if bg.dim() == 1 else bg.shape[0])
        _lib.check(lib.nerf_composite_fwd_indexed(_p(rgb_c), _p(sigma_c), _p(slots), _p(z), _p(rays_d), _p(bg), bg_rows,
                                                  R, S, _p(out_rgb), _p(depth), _p(acc), _stream()),
                   "nerf_composite_fwd_indexed")
        ctx.save_for_backward(rgb_c, sigma_c, slots, z, rays_d, bg)
        return out_rgb, depth, acc

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_acc):
        lib = _lib.load()
        rgb_c, sigma_c, slots, z, rays_d, bg = ctx.saved_tensors
        R, S = z.shape
        d_rgb, d_sigma = torch.empty_like(rgb_c), torch.empty_like(sigma_c)
        bg_rows = 0 if bg is None else (1 if bg.dim() == 1 else bg.shape[0])
        _lib.check(lib.nerf_composite_bwd_indexed(_p(rgb_c), _p(sigma_c), _p(slots), _p(z), _p(rays_d), _p(bg), bg_rows,
                                                  _p(g_rgb.contiguous()), _p(g_depth.contiguous()), _p(g_acc.contiguous()),
                                                  R, S, _p(d_rgb), _p(d_sigma), _stream()), "nerf_composite_bwd_indexed")
        return d_rgb, d_sigma, None, None, None, None


def composite_indexed(rgb_c: Tensor, sigma_c: Tensor, slots: Tensor, z: Tensor, rays_d: Tensor,
                      bg: Optional[Tensor] = None):
    """volume_render over compact field outputs; skipped samples contribute sigma = 0."""
    return _CompositeIndexed.apply(rgb_c, sigma_c.reshape(-1), slots, z, rays_d, bg)


def sample_pdf(z_coarse: Tensor, weights: Tensor, n_fine: int, u: Optional[Tensor] = None) -> Tensor:
    """Inverse-CDF fine depths merged with the coarse ones: [R, S + n_fine], sorted (opt-in extension)."""
    lib = _lib.load()
    z_coarse, weights = _dev(z_coarse, "z_coarse"), _dev(weights, "weights")
    R, S = z_coarse.shape
    if u is not None:
        u = _dev(u, "u")
    out = torch.empty(R, S + n_fine, device=z_coarse.device)
    _lib.check(lib.nerf_sample_pdf(_p(z_coarse), _p(weights), _p(u), R, S, n_fine, _p(out), _stream()), "nerf_sample_pdf")
    return out


# --------------------------------------------------------------------------- a12
def grid_lattice(bound: float, resolution: int, device) -> Tensor:
    lib = _lib.load()
    pts = torch.empty(resolution ** 3, 3, device=device, dtype=torch.float32)
    if pts.device.type != "cuda":
        raise _lib.NerfHipError("grid_lattice: the hot path only runs on a HIP device")
    _lib.check(lib.nerf_grid_lattice(float(bound), resolution, _p(pts), _stream()), "nerf_grid_lattice")
    return pts


def grid_threshold(sigma_grid: Tensor, threshold: float, prev: Optional[Tensor] = None, decay: float = 1.0):
    """binary = grid > threshold (+ running max against ``prev`` for dynamic fields); returns
    (binary_grid bool, active ratio as a Python float -- the one host sync the reference also has)."""
    lib = _lib.load()
    cur = _dev(sigma_grid, "sigma_grid")
    grid = cur if prev is None else _dev(prev, "prev")
    binary = torch.empty(cur.shape, device=cur.device, dtype=torch.bool)
    count = torch.zeros(1, device=cur.device, dtype=torch.int64)
    _lib.check(lib.nerf_grid_update(_p(cur), _p(grid), _p(binary), cur.numel(), float(decay),
                                    0 if prev is None else 1, float(threshold), _p(count), _stream()),
               "nerf_grid_update")
    return binary, float(count.item()) / cur.numel()


# --------------------------------------------------------------------------- a5
def fourier_encode(x: Tensor, n_freq: int) -> Tensor:
    lib = _lib.load()
    x = _dev(x, "x")
    n, dim = x.shape
    out = torch.empty(n, dim + 2 * dim * n_freq, device=x.device, dtype=torch.float32)
    _lib.check(lib.nerf_fourier_encode(_p(x), n, dim, n_freq, _p(out), _stream()), "nerf_fourier_encode")
    return out


# --------------------------------------------------------------------------- a9
def composite_fwd(rgb: Tensor, sigma: Tensor, z: Tensor, rays_d: Tensor, bg: Optional[Tensor] = None,
                  extra: Optional[Tensor] = None, want_weights: bool = False):
    lib = _lib.load()
    rgb, sigma, z, rays_d = _dev(rgb, "rgb"), _dev(sigma, "sigma"), _dev(z, "z"), _dev(rays_d, "rays_d")
    R, S = z.shape
    bg_rows = 0
    if bg is not None:
        bg = _dev(bg, "bg")
        bg_rows = 1 if bg.dim() == 1 else bg.shape[0]
    out_rgb = torch.empty(R, 3, device=z.device, dtype=torch.float32)
    depth = torch.empty(R, device=z.device, dtype=torch.float32)
    acc = torch.empty(R, device=z.device, dtype=torch.float32)
    extra_map = None
    if extra is not None:
        extra = _dev(extra, "extra")
        extra_map = torch.empty(R, 3, device=z.device, dtype=torch.float32)
    w = torch.empty(R, S, device=z.device, dtype=torch.float32) if want_weights else None
    _lib.check(lib.nerf_composite_fwd(_p(rgb), _p(sigma), _p(z), _p(rays_d), _p(bg), bg_rows, _p(extra),
                                      R, S, _p(out_rgb), _p(depth), _p(acc), _p(extra_map), _p(w), _stream()),
               "nerf_composite_fwd")
    return out_rgb, depth, acc, extra_map, w


def composite_bwd(rgb, sigma, z, rays_d, bg, extra, g_rgb, g_depth, g_acc, g_extra):
    lib = _lib.load()
    rgb, sigma, z, rays_d = _dev(rgb, "rgb"), _dev(sigma, "sigma"), _dev(z, "z"), _dev(rays_d, "rays_d")
    R, S = z.shape
    bg_rows = 0
    if bg is not None:
        bg = _dev(bg, "bg")
        bg_rows = 1 if bg.dim() == 1 else bg.shape[0]
    g_rgb = _dev(g_rgb, "g_rgb")
    g_depth = None if g_depth is None else _dev(g_depth, "g_depth")
    g_acc = None if g_acc is None else _dev(g_acc, "g_acc")
    d_rgb = torch.empty(R, S, 3, device=z.device, dtype=torch.float32)
    d_sigma = torch.empty(R, S, device=z.device, dtype=torch.float32)
    d_extra = None
    if extra is not None and g_extra is not None:
        extra, g_extra = _dev(extra, "extra"), _dev(g_extra, "g_extra")
        d_extra = torch.empty(R, S, 3, device=z.device, dtype=torch.float32)
    else:
        extra = g_extra = None
    _lib.check(lib.nerf_composite_bwd(_p(rgb), _p(sigma), _p(z), _p(rays_d), _p(bg), bg_rows, _p(extra),
                                      _p(g_rgb), _p(g_depth), _p(g_acc), _p(g_extra), R, S,
                                      _p(d_rgb), _p(d_sigma), _p(d_extra), _stream()), "nerf_composite_bwd")
    return d_rgb, d_sigma, d_extra


def composite_mse_bwd(rgb: Tensor, sigma: Tensor, z: Tensor, rays_d: Tensor, bg: Optional[Tensor], target: Tensor,
                      loss_accum: Tensor, slots: Optional[Tensor] = None, amax_accum: Optional[Tensor] = None,
                      loss_weight: Optional[float] = None, want_pred: bool = False):
    """One kernel for compositing + MSE + their backward (reference run.py:324-337): returns
    (d_rgb, d_sigma, pred or None); the loss is ADDED to ``loss_accum`` (a zeroed device scalar)."""
    lib = _lib.load()
    rgb, sigma, z, rays_d, target = _dev(rgb, "rgb"), _dev(sigma, "sigma"), _dev(z, "z"), _dev(rays_d, "rays_d"), _dev(target, "target")
    R, S = z.shape
    bg_rows = 0
    if bg is not None:
        bg = _dev(bg, "bg")
        bg_rows = 1 if bg.dim() == 1 else bg.shape[0]
    d_rgb, d_sigma = torch.empty_like(rgb), torch.empty_like(sigma)
    pred = torch.empty(R, 3, device=z.device) if want_pred else None
    w = 1.0 / (3 * R) if loss_weight is None else loss_weight
    _lib.check(lib.nerf_composite_mse_bwd(_p(rgb), _p(sigma), _p(slots), _p(z), _p(rays_d), _p(bg), bg_rows, _p(target), w, R, S,
                                          _p(pred), _p(loss_accum), _p(d_rgb), _p(d_sigma), _p(amax_accum), _p(sum_ws(z.device)), _stream()),
               "nerf_composite_mse_bwd")
    return d_rgb, d_sigma, pred


class _Composite(torch.autograd.Function):
    """volume_render (reference src/renderer.py:204-237) as one fused kernel per direction."""

    @staticmethod
    def forward(ctx, rgb, sigma, z, rays_d, bg, extra):
        out_rgb, depth, acc, extra_map, _ = composite_fwd(rgb, sigma, z, rays_d, bg, extra)
        ctx.save_for_backward(rgb, sigma, z, rays_d, bg, extra)
        if extra_map is None:
            extra_map = out_rgb.new_zeros(0)
        return out_rgb, depth, acc, extra_map

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_acc, g_extra):
        rgb, sigma, z, rays_d, bg, extra = ctx.saved_tensors
        d_rgb, d_sigma, d_extra = composite_bwd(rgb, sigma, z, rays_d, bg, extra, g_rgb.contiguous(),
                                                g_depth.contiguous(), g_acc.contiguous(),
                                                None if extra is None else g_extra.contiguous())
        return d_rgb, d_sigma, None, None, None, d_extra


def composite(rgb: Tensor, sigma: Tensor, z: Tensor, rays_d: Tensor, bg: Optional[Tensor] = None,
              extra: Optional[Tensor] = None):
    """Differentiable w.r.t. rgb [R,S,3], sigma [R,S] (and extra [R,S,3])."""
    out_rgb, depth, acc, extra_map = _Composite.apply(rgb, sigma, z, rays_d, bg, extra)
    return out_rgb, depth, acc, (extra_map if extra is not None else None)


# --------------------------------------------------------------------------- a6
MLP_PARAM_COUNT = 595844


def mlp_packed_bytes() -> int:
    return _lib.load().nerf_mlp_packed_bytes()


def mlp_pack(params: Tensor, packed: Optional[Tensor] = None, which: int = 3) -> Tensor:
    """flat fp32 [595844] (reference state_dict order) -> fragment-ordered bf16 streams.  ``which``: 1 the
    training streams + biases, 2 the inference stream, 3 both."""
    lib = _lib.load()
    params = _dev(params, "params")
    if params.numel() != MLP_PARAM_COUNT:
        raise ValueError(f"params must have {MLP_PARAM_COUNT} elements, got {params.numel()}")
    if packed is None:
        packed = torch.empty(lib.nerf_mlp_packed_bytes(), device=params.device, dtype=torch.uint8)
    _lib.check(lib.nerf_mlp_pack_streams(_p(params), _p(packed), which, _stream()), "nerf_mlp_pack")
    return packed


def mlp_stash_bytes(n: int) -> int:
    return _lib.load().nerf_mlp_stash_bytes(n)


def mlp_fwd(packed: Tensor, rays_o: Tensor, rays_d: Tensor, z: Optional[Tensor],
            stash: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """ray mode (z [R,S] given): rgb [R*S,3], sigma [R*S]; point mode (z None): per point."""
    lib = _lib.load()
    rays_o, rays_d = _dev(rays_o, "rays_o"), _dev(rays_d, "rays_d")
    if z is not None:
        z = _dev(z, "z")
        n, S = z.numel(), z.shape[1]
    else:
        n, S = rays_o.shape[0], 0
    rgb = torch.empty(n, 3, device=rays_o.device, dtype=torch.float32)
    sigma = torch.empty(n, device=rays_o.device, dtype=torch.float32)
    _lib.check(lib.nerf_mlp_fwd(_p(packed), _p(rays_o), _p(rays_d), _p(z), n, S, _p(rgb), _p(sigma),
                                _p(stash), _stream()), "nerf_mlp_fwd")
    return rgb, sigma


def render_rays_fwd(packed: Tensor, rays_o: Tensor, rays_d: Tensor, n_samples: int, near: float, far: float,
                    bg: Optional[Tensor] = None, chunk: int = 65536, workspace: Optional[Tensor] = None):
    """(rgb [R,3], depth [R], acc [R]) of the vanilla field for any number of rays: one launch chain over
    chunks of ``chunk`` rays in one reused workspace (nerf_render_rays_fwd)."""
    lib = _lib.load()
    rays_o, rays_d = _dev(rays_o, "rays_o"), _dev(rays_d, "rays_d")
    R = rays_o.shape[0]
    chunk = max(1, min(chunk, max(R, 1)))
    need = lib.nerf_render_rays_workspace_bytes(chunk, n_samples)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, device=rays_o.device, dtype=torch.uint8)
    bg_rows = 0
    if bg is not None:
        bg = _dev(bg, "bg")
        bg_rows = 1 if bg.dim() == 1 else bg.shape[0]
    out = torch.empty(R, 3, device=rays_o.device)
    depth, acc = torch.empty(R, device=rays_o.device), torch.empty(R, device=rays_o.device)
    _lib.check(lib.nerf_render_rays_fwd(_p(packed), _p(rays_o), _p(rays_d), R, n_samples, float(near), float(far), _p(bg), bg_rows,
                                        chunk, _p(workspace), _p(out), _p(depth), _p(acc), _stream()), "nerf_render_rays_fwd")
    return out, depth, acc


def mlp_bwd_workspace_bytes(n: int) -> int:
    return _lib.load().nerf_mlp_bwd_workspace_bytes(n)


def mlp_bwd(packed: Tensor, stash: Tensor, rgb: Tensor, sigma: Tensor, d_rgb: Tensor, d_sigma: Tensor,
            grads: Optional[Tensor] = None, workspace: Optional[Tensor] = None, amax: Optional[Tensor] = None,
            mark=None) -> Tensor:
    """Parameter gradients [595844] (reference state_dict order) of the fused decoder.  ``amax``: the
    largest output-layer derivative as a device scalar (from composite_mse_bwd), else computed here."""
    lib = _lib.load()
    rgb, sigma = _dev(rgb, "rgb"), _dev(sigma, "sigma")
    d_rgb, d_sigma = _dev(d_rgb, "d_rgb"), _dev(d_sigma, "d_sigma")
    n = sigma.numel()
    if grads is None:
        grads = torch.empty(MLP_PARAM_COUNT, device=rgb.device, dtype=torch.float32)
    if workspace is None:
        workspace = torch.empty(lib.nerf_mlp_bwd_workspace_bytes(n), device=rgb.device, dtype=torch.uint8)
    if amax is None:
        _lib.check(lib.nerf_mlp_bwd(_p(packed), _p(stash), _p(rgb), _p(sigma), _p(d_rgb), _p(d_sigma), n,
                                    _p(grads), _p(workspace), _stream()), "nerf_mlp_bwd")
    else:
        _lib.check(lib.nerf_mlp_bwd_dgrad_ex(_p(packed), _p(stash), _p(rgb), _p(sigma), _p(d_rgb), _p(d_sigma), n,
                                             _p(workspace), _p(amax), _stream()), "nerf_mlp_bwd_dgrad_ex")
        if mark is not None:
            mark("dgrad")
        _lib.check(lib.nerf_mlp_bwd_wgrad(_p(stash), _p(workspace), n, _p(grads), _stream()), "nerf_mlp_bwd_wgrad")
    return grads


def mlp_bwd_overlapped(packed: Tensor, stash: Tensor, rgb: Tensor, sigma: Tensor, d_rgb: Tensor, d_sigma: Tensor,
                       grads: Tensor, workspace: Tensor, reduce_async, amax: Optional[Tensor] = None, mark=None) -> None:
    """mlp_bwd for data-parallel training: dgrad, then the weight gradients in two launches;
    ``reduce_async(view)`` starts the all-reduce of a finished parameter range and returns a
    handle (``.wait()``) or None -- the first range is on the wire while the second is computed."""
    lib = _lib.load()
    n = sigma.numel()
    _lib.check(lib.nerf_mlp_bwd_dgrad_ex(_p(packed), _p(stash), _p(rgb), _p(sigma), _p(d_rgb), _p(d_sigma), n,
                                         _p(workspace), _p(amax), _stream()), "nerf_mlp_bwd_dgrad")
    if mark is not None:
        mark("dgrad")
    split = lib.nerf_mlp_wgrad_part_split()
    handles = []
    for part, view in ((1, grads[split:]), (2, grads[:split])):
        _lib.check(lib.nerf_mlp_bwd_wgrad_part(_p(stash), _p(workspace), n, _p(grads), part, _stream()),
                   "nerf_mlp_bwd_wgrad_part")
        handles.append(reduce_async(view))
    for h in handles:
        if h is not None:
            h.wait()


class _Decoder(torch.autograd.Function):
    """Fused Fourier-encode + 8x256 decoder: forward stashes, backward = dgrad chain + wgrad."""

    @staticmethod
    def forward(ctx, params, packed, rays_o, rays_d, z):
        n = z.numel() if z is not None else rays_o.shape[0]
        stash = torch.empty(mlp_stash_bytes(n), device=params.device, dtype=torch.uint8)
        rgb, sigma = mlp_fwd(packed, rays_o, rays_d, z, stash)
        ctx.save_for_backward(packed, stash, rgb, sigma)
        return rgb, sigma

    @staticmethod
    def backward(ctx, d_rgb, d_sigma):
        packed, stash, rgb, sigma = ctx.saved_tensors
        grads = mlp_bwd(packed, stash, rgb, sigma, d_rgb.contiguous(), d_sigma.contiguous())
        return grads, None, None, None, None


class _DecoderEncoded(torch.autograd.Function):
    """NeRFDecoder.forward(x_enc, d_enc) (reference src/decoders.py:68-87): the same chain kernels fed with the
    caller's encodings; differentiable w.r.t. the flat parameter vector."""

    @staticmethod
    def forward(ctx, params, packed, x_enc, d_enc, train):
        lib = _lib.load()
        n = x_enc.shape[0]
        rgb = torch.empty(n, 3, device=x_enc.device)
        sigma = torch.empty(n, device=x_enc.device)
        stash = torch.empty(mlp_stash_bytes(n), device=x_enc.device, dtype=torch.uint8) if train else None
        _lib.check(lib.nerf_mlp_fwd_encoded(_p(packed), _p(x_enc), _p(d_enc), n, _p(rgb), _p(sigma), _p(stash), _stream()),
                   "nerf_mlp_fwd_encoded")
        if train:
            ctx.save_for_backward(packed, stash, rgb, sigma)
        return rgb, sigma

    @staticmethod
    def backward(ctx, d_rgb, d_sigma):
        packed, stash, rgb, sigma = ctx.saved_tensors
        return mlp_bwd(packed, stash, rgb, sigma, d_rgb.contiguous(), d_sigma.contiguous()), None, None, None, None


def decoder_encoded(params: Tensor, packed: Tensor, x_enc: Tensor, d_enc: Tensor):
    x_enc, d_enc = _dev(x_enc, "x_enc"), _dev(d_enc, "d_enc")
    if x_enc.shape[1] != 63 or d_enc.shape[1] != 27 or x_enc.shape[0] != d_enc.shape[0]:
        raise ValueError(f"expected x_enc [n,63] and d_enc [n,27], got {tuple(x_enc.shape)} and {tuple(d_enc.shape)}")
    if x_enc.requires_grad or d_enc.requires_grad:
        raise NotImplementedError("gradients w.r.t. the Fourier codes are not produced (positions are not trainable)")
    if x_enc.shape[0] == 0:
        return x_enc.new_zeros(0, 3), x_enc.new_zeros(0)
    train = torch.is_grad_enabled() and params.requires_grad
    return _DecoderEncoded.apply(params, packed, x_enc, d_enc, train)


def decoder(params: Tensor, packed: Tensor, rays_o: Tensor, rays_d: Tensor, z: Optional[Tensor]):
    """rgb [n,3], sigma [n]; differentiable w.r.t. the flat parameter vector when it requires grad."""
    if torch.is_grad_enabled() and params.requires_grad:
        return _Decoder.apply(params, packed, rays_o, rays_d, z)
    return mlp_fwd(packed, rays_o, rays_d, z)


# --------------------------------------------------------------------------- a7 / a8
import ctypes as _ct
import math as _math

import numpy as _np


class HashLevelTable:
    """Per-level table of the multiresolution grid (this build's definition; the third-party
    library is absent): scale_l = base * s^l - 1 evaluated in float64 and rounded once to fp32,
    res_l = ceil(scale_l) + 1, dense storage when res^3 fits the hash-map budget (sizes padded to
    8 entries).  Identical to oracle/nerf_oracle.py::hash_grid_levels."""

    def __init__(self, n_levels=16, log2_hashmap_size=19, base_resolution=16, per_level_scale=1.5):
        budget = 1 << log2_hashmap_size
        scale, res, size, offset, dense, off = [], [], [], [], [], 0
        for l in range(n_levels):
            s64 = base_resolution * (per_level_scale ** l) - 1.0
            r = int(_math.ceil(round(s64, 9))) + 1
            is_dense = r ** 3 <= budget
            sz = ((r ** 3 + 7) // 8) * 8 if is_dense else budget
            scale.append(s64); res.append(r); size.append(sz); offset.append(off); dense.append(int(is_dense))
            off += sz
        self.n_levels, self.entries = n_levels, off
        self.scale = _np.asarray(scale, dtype=_np.float32)
        self.res = _np.asarray(res, dtype=_np.uint32)
        self.size = _np.asarray(size, dtype=_np.uint32)
        self.offset = _np.asarray(offset, dtype=_np.uint32)
        self.dense = _np.asarray(dense, dtype=_np.uint32)

    def host_args(self):
        """the five level arrays as ctypes pointers (built once: the arrays are never reallocated)"""
        args = getattr(self, "_host_args", None)
        if args is None:
            args = self._host_args = tuple(a.ctypes.data_as(_ct.c_void_p) for a in (self.scale, self.res, self.size, self.offset, self.dense))
        return args


def hash_encode_fwd(pts: Tensor, table: Tensor, levels: HashLevelTable, bound: float,
                    want_f32: bool = True, out_nat: Optional[Tensor] = None, want_index: bool = False):
    """``table``: fp32 [entries, 2] (the parameters) or an fp16 copy of them (torch.float16: half the bytes per
    gather; see nerf_hash_encode_fwd_f16)."""
    lib = _lib.load()
    pts = _dev(pts, "pts")
    n = pts.shape[0]
    out = torch.empty(n, 2 * levels.n_levels, device=pts.device) if want_f32 else None
    if table.dtype == torch.float16:
        if want_index:
            raise ValueError("hash_encode_fwd: corner indices come with the fp32 table only")
        table = _dev(table, "table", torch.float16)
        _lib.check(lib.nerf_hash_encode_fwd_f16(_p(pts), n, _p(table), levels.n_levels, *levels.host_args(), float(bound),
                                                _p(out), _p(out_nat), _stream()), "nerf_hash_encode_fwd_f16")
        return out, None
    table = _dev(table, "table")
    idx = torch.empty(n, levels.n_levels, 8, device=pts.device, dtype=torch.int32) if want_index else None
    _lib.check(lib.nerf_hash_encode_fwd(_p(pts), n, _p(table), levels.n_levels, *levels.host_args(), float(bound),
                                        _p(out), _p(out_nat), _p(idx), _stream()), "nerf_hash_encode_fwd")
    return out, idx


def hash_encode_fwd_nat(pts: Tensor, table: Tensor, levels: HashLevelTable, bound: float, out_nat: Tensor, fp16: bool) -> None:
    """operand image of the hash features (blocks of 32 points x 16 features) as bf16 or fp16, from the fp32 parameters or
    their fp16 copy"""
    lib = _lib.load()
    pts = _dev(pts, "pts")
    half = table.dtype == torch.float16
    table = _dev(table, "table", torch.float16 if half else torch.float32)
    _lib.check(lib.nerf_hash_encode_fwd_nat(_p(pts), pts.shape[0], None if half else _p(table), _p(table) if half else None, levels.n_levels,
                                            *levels.host_args(), float(bound), _p(out_nat), 1 if fp16 else 0, _stream()),
               "nerf_hash_encode_fwd_nat")


def hash_encode_fwd_nat_tables(pts: Tensor, tables, levels: HashLevelTable, bound: float, outs, fp16: bool) -> bool:
    """The operand images of several fp16 tables of ONE level structure at the same points in one launch, when the tables
    and the images are equally spaced in memory (views of one flat buffer); False: not the case, nothing was launched."""
    if len(tables) < 2 or any(t.dtype != torch.float16 or not t.is_contiguous() for t in tables):
        return False
    t_step = tables[1].data_ptr() - tables[0].data_ptr()
    o_step = outs[1].data_ptr() - outs[0].data_ptr()
    if t_step <= 0 or o_step <= 0 or t_step % 4 or o_step % 2:
        return False
    for k in range(1, len(tables)):
        if tables[k].data_ptr() - tables[0].data_ptr() != k * t_step or outs[k].data_ptr() - outs[0].data_ptr() != k * o_step:
            return False
    pts = _dev(pts, "pts")
    _lib.check(_lib.load().nerf_hash_encode_fwd_nat_tables(_p(pts), pts.shape[0], _p(tables[0]), len(tables), t_step // 4, levels.n_levels,
                                                           *levels.host_args(), float(bound), _p(outs[0]), o_step, 1 if fp16 else 0,
                                                           _stream()), "nerf_hash_encode_fwd_nat_tables")
    return True


def f32_to_f16(src: Tensor, dst: Optional[Tensor] = None) -> Tensor:
    """fp16 copy of a flat fp32 parameter vector (the shadow table of the hash forward)"""
    src = _dev(src, "src")
    if dst is None:
        dst = torch.empty(src.shape, device=src.device, dtype=torch.float16)
    _lib.check(_lib.load().nerf_f32_to_f16(_p(src), _p(dst), src.numel(), _stream()), "nerf_f32_to_f16")
    return dst


def _hash_bwd_scratch(pts: Tensor, levels: "HashLevelTable") -> Tensor:
    """workspace of the binned scatter for the autograd paths (torch's caching allocator hands the block back)"""
    return torch.empty(_lib.load().nerf_hash_encode_bwd_workspace_bytes(pts.shape[0], levels.n_levels), dtype=torch.uint8,
                       device=pts.device)


def hash_encode_bwd_workspace_bytes(n: int, n_levels: int) -> int:
    return _lib.load().nerf_hash_encode_bwd_workspace_bytes(n, n_levels)


def hash_encode_bwd(pts: Tensor, levels: HashLevelTable, bound: float, d_feat: Tensor, d_table: Tensor,
                    level_range: Optional[Tuple[int, int]] = None, workspace: Optional[Tensor] = None, overwrite: bool = False) -> None:
    """Scatter-add into ``d_table`` (caller zeroes it); ``level_range`` (lo, hi) restricts the pass to those levels.
    ``workspace`` (uint8, >= hash_encode_bwd_workspace_bytes(n, L)) selects the binned form: partial sort by
    table slice + LDS sums instead of global float atomics.  ``overwrite`` (workspace form): the levels' gradient is
    STORED -- no zeroing by the caller, no read-back (nerf_hash_encode_bwd_ws_store)."""
    lib = _lib.load()
    pts, d_feat = _dev(pts, "pts"), _dev(d_feat, "d_feat")
    lo, hi = level_range if level_range is not None else (0, levels.n_levels)
    if overwrite and workspace is None:
        raise ValueError("hash_encode_bwd: overwrite needs the workspace form")
    if workspace is None and deterministic():
        workspace = _hash_bwd_scratch(pts, levels)           # the forms without a workspace end in float atomics
    if workspace is None:
        _lib.check(lib.nerf_hash_encode_bwd_levels(_p(pts), pts.shape[0], levels.n_levels, *levels.host_args(), float(bound),
                                                   _p(d_feat), _p(d_table), lo, hi, _stream()), "nerf_hash_encode_bwd")
    elif overwrite:
        _lib.check(lib.nerf_hash_encode_bwd_ws_store(_p(pts), pts.shape[0], levels.n_levels, *levels.host_args(), float(bound),
                                                     _p(d_feat), _p(d_table), lo, hi, _p(workspace), workspace.numel(), _stream()),
                   "nerf_hash_encode_bwd_ws_store")
    else:
        _lib.check(lib.nerf_hash_encode_bwd_ws(_p(pts), pts.shape[0], levels.n_levels, *levels.host_args(), float(bound),
                                               _p(d_feat), _p(d_table), lo, hi, _p(workspace), workspace.numel(), _stream()),
                   "nerf_hash_encode_bwd_ws")


def hash_encode_bwd_tables(pts: Tensor, levels: HashLevelTable, bound: float, d_feats, d_tables, workspace_of,
                           spec_status: Optional[Tensor] = None, spec_lm: bool = False) -> bool:
    """The overwrite-form scatter of several tables of ONE level structure from the same points in one pass of launches, when
    the tables' gradients and their feature gradients are equally spaced in memory (views of flat buffers).  ``workspace_of(n,
    n_levels, n_tables)`` returns the uint8 workspace.  False: not the case, nothing was launched.  ``spec_status`` (a host-mapped
    int32 [8] block): the speculative form (nerf_hash_encode_bwd_ws_store_tables_spec; project-nerf_amd/specbwd.py); ``spec_lm``: the
    producer wrote the level-major gradients into the workspace (d_feats only give the layout check its strides)."""
    k = len(d_tables)
    if k < 2 or any(t.dtype != torch.float32 or not t.is_contiguous() for t in list(d_tables) + list(d_feats)):
        return False
    t_step = d_tables[1].data_ptr() - d_tables[0].data_ptr()
    f_step = d_feats[1].data_ptr() - d_feats[0].data_ptr()
    if t_step <= 0 or f_step <= 0 or t_step % 8 or f_step % 4 or t_step // 8 < levels.entries:
        return False
    for i in range(1, k):
        if d_tables[i].data_ptr() - d_tables[0].data_ptr() != i * t_step or d_feats[i].data_ptr() - d_feats[0].data_ptr() != i * f_step:
            return False
    lib = _lib.load()
    if k * levels.n_levels > 48:
        return False
    pts = _dev(pts, "pts")
    n = pts.shape[0]
    ws = workspace_of(n, levels.n_levels, k)
    if spec_status is not None:
        _lib.check(lib.nerf_hash_encode_bwd_ws_store_tables_spec(_p(pts), n, k, t_step // 8, levels.n_levels, *levels.host_args(), float(bound),
                                                                 None if spec_lm else _p(d_feats[0]), f_step // 4, _p(d_tables[0]), _p(ws), ws.numel(),
                                                                 spec_status.data_ptr(), _stream()), "nerf_hash_encode_bwd_ws_store_tables_spec")
        return True
    _lib.check(lib.nerf_hash_encode_bwd_ws_store_tables(_p(pts), n, k, t_step // 8, levels.n_levels, *levels.host_args(), float(bound),
                                                        _p(d_feats[0]), f_step // 4, _p(d_tables[0]), _p(ws), ws.numel(), _stream()),
               "nerf_hash_encode_bwd_ws_store_tables")
    return True


def hash_bwd_slots(workspace: Tensor, n: int, n_levels: int) -> Tuple[int, int]:
    """device addresses of the hash backward workspace's slots a producer fills for the forms that do not count: (the
    NERF_AMAX_WORDS words for the largest |d feature|, the level-major gradients float2 [n_levels][n])"""
    import ctypes
    amax_p, lm_p = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.check(_lib.load().nerf_hash_encode_bwd_ws_slots(workspace.data_ptr(), n, n_levels, ctypes.byref(amax_p), ctypes.byref(lm_p)),
               "nerf_hash_encode_bwd_ws_slots")
    return amax_p.value, lm_p.value


IMLP_PARAM_COUNT = 11264
IMLP_SIGMA_PARAMS = 3072


def imlp_pack(params: Tensor, packed: Optional[Tensor] = None) -> Tensor:
    lib = _lib.load()
    params = _dev(params, "params")
    if params.numel() != IMLP_PARAM_COUNT:
        raise ValueError(f"instant decoder expects {IMLP_PARAM_COUNT} parameters, got {params.numel()}")
    if packed is None:
        packed = torch.empty(lib.nerf_imlp_packed_bytes(), device=params.device, dtype=torch.uint8)
    _lib.check(lib.nerf_imlp_pack(_p(params), _p(packed), _stream()), "nerf_imlp_pack")
    return packed


class _InstantField(torch.autograd.Function):
    """hash encode -> tiny MLPs (reference core.py:354-359 for part2_instant), fwd and bwd in HIP."""

    @staticmethod
    def forward(ctx, table, net_params, packed, pts, dirs, levels, bound, train):
        lib = _lib.load()
        n = pts.shape[0]
        ws = torch.empty(lib.nerf_imlp_workspace_bytes(n), device=pts.device, dtype=torch.uint8)
        hash_encode_fwd(pts, table, levels, bound, want_f32=False, out_nat=ws)
        rgb = torch.empty(n, 3, device=pts.device)
        sigma = torch.empty(n, device=pts.device)
        _lib.check(lib.nerf_imlp_fwd(_p(packed), _p(ws), _p(dirs), n, _p(rgb), _p(sigma), 1 if train else 0, _stream()),
                   "nerf_imlp_fwd")
        if train:
            ctx.save_for_backward(packed, ws, rgb, sigma, pts)
            ctx.levels, ctx.bound, ctx.table_shape = levels, bound, table.shape
        return rgb, sigma

    @staticmethod
    def backward(ctx, d_rgb, d_sigma):
        lib = _lib.load()
        packed, ws, rgb, sigma, pts = ctx.saved_tensors
        n = pts.shape[0]
        g_net = torch.empty(IMLP_PARAM_COUNT, device=pts.device)
        d_feat = torch.empty(n, 2 * ctx.levels.n_levels, device=pts.device)
        _lib.check(lib.nerf_imlp_bwd(_p(packed), _p(ws), _p(rgb), _p(sigma), _p(d_rgb.contiguous()),
                                     _p(d_sigma.contiguous()), n, _p(g_net), _p(d_feat), _stream()), "nerf_imlp_bwd")
        g_table = torch.zeros(ctx.table_shape, device=pts.device)
        hash_encode_bwd(pts, ctx.levels, ctx.bound, d_feat, g_table, workspace=_hash_bwd_scratch(pts, ctx.levels))
        return g_table, g_net, None, None, None, None, None, None


class _HashEncode(torch.autograd.Function):
    """HashRepresentation.forward as a stand-alone differentiable operator (table gradient only)."""

    @staticmethod
    def forward(ctx, table, pts, levels, bound):
        out, _ = hash_encode_fwd(pts, table, levels, bound)
        ctx.save_for_backward(pts, table)
        ctx.levels, ctx.bound = levels, bound
        return out

    @staticmethod
    def backward(ctx, d_out):
        pts, table = ctx.saved_tensors
        d_out = d_out.contiguous()
        g = d_pts = None
        if ctx.needs_input_grad[0]:
            g = torch.zeros(table.shape, device=pts.device)
            hash_encode_bwd(pts, ctx.levels, ctx.bound, d_out, g, workspace=_hash_bwd_scratch(pts, ctx.levels))
        if ctx.needs_input_grad[1]:
            d_pts = hash_encode_bwd_input(pts, table, ctx.levels, ctx.bound, d_out)
        return g, d_pts, None, None


def hash_encode_bwd_input(pts: Tensor, table: Tensor, levels: HashLevelTable, bound: float, d_feat: Optional[Tensor],
                          add_to: Optional[Tensor] = None, grad_lm: Optional[int] = None) -> Tensor:
    """d_pts [n,3]: gradient of the features w.r.t. the encoded positions (dynamic fields); ``table`` fp32 [E,2] or the fp16
    copy the forward evaluated.  ``add_to`` (fp16 table): a contiguous fp32 [n,3] tensor the gradient is ADDED to in place (and
    which is returned) -- a gradient reaching the positions by another path is already there.  ``grad_lm`` (fp16 table; instead of
    ``d_feat``): device address of the level-major gradients float2 [L][n] a producer left in the hash backward's workspace."""
    lib = _lib.load()
    if not isinstance(table, Tensor) or table.dtype not in (torch.float32, torch.float16):
        raise TypeError("hash_encode_bwd_input: table must be an fp32 or fp16 tensor")
    pts, table = _dev(pts, "pts"), _dev(table, "table", table.dtype)
    if grad_lm is not None:
        if table.dtype != torch.float16 or (add_to is not None and (add_to.shape != pts.shape or _dev(add_to, "add_to") is not add_to)):
            raise ValueError("hash_encode_bwd_input(grad_lm=...): fp16 table; add_to a contiguous fp32 tensor of the points' shape")
        out = add_to if add_to is not None else torch.empty_like(pts)
        _lib.check(lib.nerf_hash_encode_bwd_input_lm_f16(_p(pts), pts.shape[0], _p(table), levels.n_levels, *levels.host_args(), float(bound),
                                                         grad_lm, _p(out), 1 if add_to is not None else 0, _stream()),
                   "nerf_hash_encode_bwd_input_lm_f16")
        return out
    d_feat = _dev(d_feat, "d_feat")
    if add_to is not None:
        if table.dtype != torch.float16 or add_to.shape != pts.shape or _dev(add_to, "add_to") is not add_to:
            raise ValueError("hash_encode_bwd_input(add_to=...): fp16 table and a contiguous fp32 tensor of the points' shape")
        _lib.check(lib.nerf_hash_encode_bwd_input_f16_accum(_p(pts), pts.shape[0], _p(table), levels.n_levels, *levels.host_args(), float(bound),
                                                            _p(d_feat), _p(add_to), _stream()), "nerf_hash_encode_bwd_input_f16_accum")
        return add_to
    d_pts = torch.empty_like(pts)
    fn = lib.nerf_hash_encode_bwd_input_f16 if table.dtype == torch.float16 else lib.nerf_hash_encode_bwd_input
    _lib.check(fn(_p(pts), pts.shape[0], _p(table), levels.n_levels, *levels.host_args(), float(bound), _p(d_feat), _p(d_pts), _stream()),
               "nerf_hash_encode_bwd_input")
    return d_pts


def hash_encode(table: Tensor, pts: Tensor, levels: HashLevelTable, bound: float) -> Tensor:
    """features [n, 2 L] fp32, differentiable w.r.t. ``table`` [entries, 2] and w.r.t. the positions."""
    pts = _dev(pts, "pts")
    if pts.shape[0] == 0:
        return pts.new_zeros(0, 2 * levels.n_levels)
    if torch.is_grad_enabled() and (table.requires_grad or pts.requires_grad):
        return _HashEncode.apply(table, pts, levels, bound)
    return hash_encode_fwd(pts, table, levels, bound)[0]


class _InstantDecoderEncoded(torch.autograd.Function):
    """InstantNeRFDecoder.forward(x_enc, d_enc) (reference src/decoders.py:136-162): differentiable w.r.t. the
    flat tiny-MLP parameters and the hash features."""

    @staticmethod
    def forward(ctx, net_params, packed, x_enc, d_enc, train):
        lib = _lib.load()
        n = x_enc.shape[0]
        ws = torch.empty(lib.nerf_imlp_workspace_bytes(n), device=x_enc.device, dtype=torch.uint8)
        rgb, sigma = torch.empty(n, 3, device=x_enc.device), torch.empty(n, device=x_enc.device)
        _lib.check(lib.nerf_imlp_fwd_encoded(_p(packed), _p(ws), _p(x_enc), _p(d_enc), n, _p(rgb), _p(sigma), 1 if train else 0,
                                             _stream()), "nerf_imlp_fwd_encoded")
        if train:
            ctx.save_for_backward(packed, ws, rgb, sigma)
        return rgb, sigma

    @staticmethod
    def backward(ctx, d_rgb, d_sigma):
        lib = _lib.load()
        packed, ws, rgb, sigma = ctx.saved_tensors
        n = rgb.shape[0]
        g_net = torch.empty(IMLP_PARAM_COUNT, device=rgb.device)
        d_feat = torch.empty(n, 32, device=rgb.device)
        _lib.check(lib.nerf_imlp_bwd(_p(packed), _p(ws), _p(rgb), _p(sigma), _p(d_rgb.contiguous()), _p(d_sigma.contiguous()), n,
                                     _p(g_net), _p(d_feat), _stream()), "nerf_imlp_bwd")
        return g_net, None, d_feat, None, None


def instant_decoder_encoded(net_params: Tensor, packed: Tensor, x_enc: Tensor, d_enc: Tensor):
    x_enc, d_enc = _dev(x_enc.float(), "x_enc"), _dev(d_enc, "d_enc")
    if x_enc.shape[1] != 32 or d_enc.shape[1] != 27 or x_enc.shape[0] != d_enc.shape[0]:
        raise ValueError(f"expected x_enc [n,32] and d_enc [n,27], got {tuple(x_enc.shape)} and {tuple(d_enc.shape)}")
    if x_enc.shape[0] == 0:
        return x_enc.new_zeros(0, 3), x_enc.new_zeros(0)
    train = torch.is_grad_enabled() and (net_params.requires_grad or x_enc.requires_grad)
    return _InstantDecoderEncoded.apply(net_params, packed, x_enc, d_enc, train)


def instant_field(table: Tensor, net_params: Tensor, packed: Tensor, pts: Tensor, dirs: Tensor,
                  levels: HashLevelTable, bound: float):
    """rgb [n,3], sigma [n] for world-space points and unit view directions."""
    pts, dirs = _dev(pts, "pts"), _dev(dirs, "dirs")
    if pts.shape[0] == 0:
        return pts.new_zeros(0, 3), pts.new_zeros(0)
    train = torch.is_grad_enabled() and (table.requires_grad or net_params.requires_grad)
    return _InstantField.apply(table, net_params, packed, pts, dirs, levels, bound, train)


# --------------------------------------------------------------------------- a14
def adam_step(params: Tensor, grads: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, step: int, lr: float,
              beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 0.0,
              grad_scale: Optional[Tensor] = None) -> None:
    lib = _lib.load()
    for t, nm in ((params, "params"), (grads, "grads"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        if _dev(t, nm) is not t:
            raise ValueError(f"{nm} must be contiguous")
    _lib.check(lib.nerf_adam_step(_p(params), _p(grads), _p(exp_avg), _p(exp_avg_sq), params.numel(), step,
                                  lr, beta1, beta2, eps, weight_decay, _p(grad_scale), _stream()),
               "nerf_adam_step")


SMALL_GROUP = 65536        # nerf_clip_adamw_small's limit


def tv_clip_adamw_step(params: Tensor, grads: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, step: int, lr: float,
                       tv_weight: float = 0.0, max_norm: float = 0.0, weight_decay: float = 0.0,
                       beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, grad_scale: float = 1.0,
                       scratch: Optional[Tensor] = None, shadow_f16: Optional[Tensor] = None, tv_codes: Optional[Tensor] = None) -> None:
    """TV-L1 gradient (optional) + global-norm clip + AdamW on one flat group, two streaming passes.
    ``tv_codes``: a uint8 buffer of (n + 3) // 4 bytes for the TV term's signs (allocated per call otherwise).
    ``shadow_f16``: a torch.float16 tensor of the same size that receives the updated parameters as well.
    ``grad_scale`` (1/world after a summing all-reduce) scales the data gradient BEFORE the TV term is
    added, so the regulariser keeps its weight on any number of ranks (reference run.py:611-629)."""
    lib = _lib.load()
    for t, nm in ((params, "params"), (grads, "grads"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        if _dev(t, nm) is not t:
            raise ValueError(f"{nm} must be contiguous")
    # the squared-norm workspace (NERF_NORMSQ_WS_FLOATS: value, ticket, one partial per workgroup)
    normsq = scratch if (scratch is not None and scratch.numel() >= _lib.NORMSQ_WS_FLOATS) else normsq_ws(params.device)
    n = params.numel()
    if shadow_f16 is not None and (shadow_f16.dtype != torch.float16 or shadow_f16.numel() != n or not shadow_f16.is_contiguous()):
        raise ValueError("shadow_f16 must be a contiguous torch.float16 tensor of the parameters' size")
    # pass 1 leaves the TV term's signs (two bits per parameter) instead of rewriting the gradient; pass 2 rebuilds
    # g * grad_scale + TV term from them (38.5 instead of 42 bytes per parameter; the gradient buffer is left as it was)
    if tv_weight != 0.0:
        if tv_codes is None or tv_codes.numel() < (n + 3) // 4:
            tv_codes = torch.empty((n + 3) // 4, dtype=torch.uint8, device=params.device)
    else:
        tv_codes = None
    if tv_codes is None and shadow_f16 is None and n <= SMALL_GROUP:
        # a tiny MLP's weights: norm + clip + AdamW in one launch of one workgroup (normsq[0] receives the squared norm)
        _lib.check(lib.nerf_clip_adamw_small(_p(params), _p(grads), _p(exp_avg), _p(exp_avg_sq), n, step, lr, beta1, beta2, eps, weight_decay,
                                             max_norm, grad_scale, _p(normsq), 0, _stream()), "nerf_clip_adamw_small")
        return
    # the call STORES the squared norm (accumulate 0): no zeroing launch
    _lib.check(lib.nerf_tv_normsq_codes(_p(params), _p(grads), n, 1, tv_weight, grad_scale, _p(normsq), 0, _p(tv_codes), _stream()),
               "nerf_tv_normsq_codes")
    _lib.check(lib.nerf_adamw_clip_step_tv(_p(params), _p(grads), _p(exp_avg), _p(exp_avg_sq), n, step, lr, beta1, beta2, eps, weight_decay,
                                           _p(normsq), max_norm, grad_scale, _p(tv_codes), n, tv_weight, n, 0.0, 0, 0, 0.0, _p(shadow_f16),
                                           _stream()), "nerf_adamw_clip_step_tv")
