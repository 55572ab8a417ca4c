"""Renderer (reference src/renderer.py): stratified sampling, alpha compositing, render_rays /
render_image and the occupancy grid, with the reference's signatures and return arities.
Every arithmetic step runs in libnerf_hip.so."""
import torch
import torch.nn as nn

from . import ops


class DensityGrid(nn.Module):
    """Occupancy bitfield over [-bound, bound]^3 (reference src/renderer.py:5-183).  Buffers
    ``grid`` / ``binary_grid`` are part of the checkpoint format."""

    def __init__(self, resolution=128, bound=1.0, threshold=0.01):
        super().__init__()
        self.resolution, self.bound, self.threshold = resolution, bound, threshold
        self.register_buffer("grid", torch.zeros(resolution, resolution, resolution))
        self.register_buffer("binary_grid", torch.ones(resolution, resolution, resolution, dtype=torch.bool))
        self.scale = resolution / (2 * bound)
        self.offset = bound

    @torch.no_grad()
    def update(self, model, n_samples=128 ** 3, device="cuda", time=None, decay=1.0):
        """Re-query sigma on the res^3 lattice of linspace(-b, b, res) nodes in 2^18-point batches;
        static fields overwrite, dynamic ones keep max(grid*decay, current) (renderer.py:35-132)."""
        res = self.resolution
        mode = getattr(model, "mode", "unknown")
        if mode == "part3" and time is None:
            raise ValueError("Part 3 density grid update requires a time parameter")
        pts = ops.grid_lattice(self.bound, res, self.grid.device)
        batch = 2 ** 18

        def query(t_anchor=None):
            sig = torch.empty(res ** 3, device=self.grid.device)
            for i in range(0, pts.shape[0], batch):
                p = pts[i:i + batch]
                if t_anchor is None:
                    _, s = model(p, torch.zeros_like(p))
                else:
                    _, s, _ = model(p, torch.zeros_like(p), t=torch.full((p.shape[0], 1), t_anchor, device=p.device))
                sig[i:i + batch] = s.reshape(-1).float()
            return sig

        if mode == "part4":
            # density at the three time anchors 0, 0.5, 1 (`time` is ignored), element-wise maximum, then the
            # running maximum against the decayed history (reference src/renderer.py:65-86, 122-125)
            sig = torch.stack([query(a) for a in (0.0, 0.5, 1.0)], 0).max(dim=0)[0]
            self.binary_grid, ratio = ops.grid_threshold(sig.view(res, res, res).contiguous(), self.threshold, prev=self.grid, decay=decay)
            return ratio
        if mode == "part3":
            # density at the given time, running maximum against the decayed history (src/renderer.py:87-101, 122-125)
            sig = query(float(torch.as_tensor(time).reshape(-1)[0]))
            self.binary_grid, ratio = ops.grid_threshold(sig.view(res, res, res).contiguous(), self.threshold, prev=self.grid, decay=decay)
            return ratio
        self.grid = query().view(res, res, res)
        self.binary_grid, ratio = ops.grid_threshold(self.grid, self.threshold)
        return ratio

    def get_active_mask(self, pts):
        return ops.active_mask(pts, self.binary_grid, self.bound)

    def should_update(self, step, update_interval=16, warmup_iters=0):
        return step >= warmup_iters and step % update_interval == 0


def sample_stratified(near, far, n_samples, n_rays, device, perturb):
    """z [n_rays, n_samples] (reference src/renderer.py:186-201); jitter drawn with torch.rand."""
    u = torch.rand(n_rays, n_samples, device=device) if perturb else None
    dummy = torch.zeros(n_rays, 3, device=device)
    return ops.sample_rays(dummy, dummy, near, far, n_samples, u=u)


def volume_render(rgb, sigma, z_vals, rays_d, bg_color=None):
    """reference src/renderer.py:204-237, differentiable w.r.t. rgb and sigma."""
    out_rgb, depth, acc, _ = ops.composite(rgb.contiguous(), sigma.contiguous(), z_vals, rays_d, bg_color)
    return out_rgb, depth, acc


def render_rays(model, rays_o, rays_d, near, far, n_samples, perturb, density_grid=None, times=None,
                white_bkgd=True, bg_color=None):
    """reference src/renderer.py:240-384: 3-tuple (rgb, depth, acc) for static fields; for the dynamic (part3 / part4)
    fields a 4-tuple with ``extras['mean_delta_x']`` = sum_s w_s delta_x_s (missing ``times`` mean t = 0)."""
    device = rays_o.device
    n_rays = rays_o.shape[0]
    mode = getattr(model, "mode", "unknown")
    if mode in ("part3", "part4"):
        return _render_rays_dynamic(model, rays_o, rays_d, near, far, n_samples, perturb, density_grid, times, white_bkgd, bg_color)
    if bg_color is None:
        bg_color = torch.ones(3, device=device) if white_bkgd else torch.zeros(3, device=device)
    rays_o, rays_d = rays_o.contiguous(), rays_d.contiguous()
    u = torch.rand(n_rays, n_samples, device=device) if perturb else None

    if density_grid is None and hasattr(model, "field_from_rays") and mode == "part2_nerf":
        # fused path: sample points are formed inside the decoder kernel, never written to HBM
        z = ops.sample_rays(rays_o, rays_d, near, far, n_samples, u=u)
        rgb, sigma = model.field_from_rays(rays_o, rays_d, z)
    else:
        if density_grid is not None:
            # a1-a4 in one kernel: depths, points, occupancy test, compaction of the active samples
            z, slots, pts, dirs = ops.sample_compact(rays_o, rays_d, near, far, n_samples, density_grid.binary_grid,
                                                     density_grid.bound, u=u)
            if pts.shape[0] == 0:
                # nothing active: query sample 0 anyway so the autograd graph stays connected
                # (reference renderer.py:309-311)
                _, p_all, d_all = ops.sample_rays(rays_o[:1], rays_d[:1], near, far, n_samples,
                                                  u=None if u is None else u[:1].contiguous(), want_points=True)
                pts, dirs = p_all[:1].contiguous(), d_all[:1].contiguous()
                slots = slots.clone()
                slots[0] = 0
            c_rgb, c_sigma = model(pts, dirs)
            return ops.composite_indexed(c_rgb.float(), c_sigma.float(), slots, z, rays_d, bg_color)
        z, pts, dirs = ops.sample_rays(rays_o, rays_d, near, far, n_samples, u=u, want_points=True)
        rgb, sigma = model(pts, dirs)
    rgb = rgb.float().view(n_rays, n_samples, 3)
    sigma = sigma.float().view(n_rays, n_samples)
    return volume_render(rgb, sigma, z, rays_d, bg_color=bg_color)


def _render_rays_dynamic(model, rays_o, rays_d, near, far, n_samples, perturb, density_grid, times, white_bkgd, bg_color):
    """Dynamic branch of render_rays (reference src/renderer.py:277-384): per-ray times broadcast to the samples
    (t = 0 if none were given), occupancy-masked query with zero-filled scatter of rgb / sigma / delta_x, compositing
    with the displacement as an extra channel (one kernel yields rgb, depth, acc and mean_delta_x)."""
    device = rays_o.device
    n_rays = rays_o.shape[0]
    if bg_color is None:
        bg_color = torch.ones(3, device=device) if white_bkgd else torch.zeros(3, device=device)
    # the reference substitutes t = 0 for missing times BEFORE it decides on the return arity
    # (src/renderer.py:279-284, 363), so a dynamic field always yields the 4-tuple
    want_extras = True
    if times is None:
        times = torch.zeros((n_rays, 1), device=device)
    rays_o, rays_d = rays_o.contiguous(), rays_d.contiguous()
    u = torch.rand(n_rays, n_samples, device=device) if perturb else None
    t_flat = times.expand(-1, n_samples).reshape(-1, 1)
    if density_grid is not None:
        z, slots, pts, dirs = ops.sample_compact(rays_o, rays_d, near, far, n_samples, density_grid.binary_grid, density_grid.bound, u=u)
        active = slots >= 0
        if pts.shape[0] == 0:                                   # keep the graph connected (renderer.py:309-311)
            _, p_all, d_all = ops.sample_rays(rays_o[:1], rays_d[:1], near, far, n_samples,
                                              u=None if u is None else u[:1].contiguous(), want_points=True)
            pts, dirs = p_all[:1].contiguous(), d_all[:1].contiguous()
            active = torch.zeros_like(active)
            active[0] = True
            order = torch.zeros(1, dtype=torch.long, device=device)
        else:
            order = slots[active].long()                        # compact row of every active sample
        t_c = torch.empty(pts.shape[0], 1, device=device)
        t_c[order] = t_flat[active]
        c_rgb, c_sigma, c_delta = model(pts, dirs, t=t_c)
        n = n_rays * n_samples
        rgb = c_rgb.new_zeros(n, 3, dtype=torch.float32)
        sigma = c_sigma.new_zeros(n, 1, dtype=torch.float32)
        delta = c_delta.new_zeros(n, 3, dtype=torch.float32)
        rgb[active] = c_rgb.float()[order]
        sigma[active] = c_sigma.float()[order]
        delta[active] = c_delta.float()[order]
    else:
        z, pts, dirs = ops.sample_rays(rays_o, rays_d, near, far, n_samples, u=u, want_points=True)
        rgb, sigma, delta = model(pts, dirs, t=t_flat.contiguous())
    rgb = rgb.float().view(n_rays, n_samples, 3)
    sigma = sigma.float().view(n_rays, n_samples)
    extra = delta.float().view(n_rays, n_samples, 3).contiguous() if want_extras else None
    out_rgb, depth, acc, mean_delta = ops.composite(rgb.contiguous(), sigma.contiguous(), z, rays_d, bg_color, extra)
    if want_extras:
        return out_rgb, depth, acc, {"mean_delta_x": mean_delta}
    return out_rgb, depth, acc


def render_image(model, rays_o, rays_d, near, far, n_samples, chunk, white_bkgd):
    """reference src/renderer.py:387-418."""
    h, w = rays_o.shape[:2]
    o, d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
    if getattr(model, "mode", None) == "part2_nerf" and not torch.is_grad_enabled() and getattr(model.decoder, "fused", True):
        # the whole image as one launch chain (nerf_render_rays_fwd): same kernels, one reused workspace
        bg = torch.ones(3, device=o.device) if white_bkgd else torch.zeros(3, device=o.device)
        return ops.render_rays_fwd(model.decoder.packed_weights(), o.contiguous(), d.contiguous(), n_samples, near, far, bg,
                                   chunk)[0].view(h, w, 3)
    out = []
    for i in range(0, o.shape[0], chunk):
        out.append(render_rays(model=model, rays_o=o[i:i + chunk], rays_d=d[i:i + chunk], near=near, far=far,
                               n_samples=n_samples, perturb=False, white_bkgd=white_bkgd)[0])
    return torch.cat(out, dim=0).view(h, w, 3)
