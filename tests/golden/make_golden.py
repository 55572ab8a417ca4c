"""Generate golden vectors from the REFERENCE ITSELF (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Imports the read-only reference at /root/reference (pure-PyTorch CPU half; the
tinycudann half is not importable offline -- g13 runs the reference's instant GLUE around
the stand-in tests/golden/tinycudann_shim.py) and stores inputs + expected outputs
as small ``.npz`` files next to this script.  The reference never travels to
the GPU box; these vectors do.  Only data is stored here (SURVEY.md section 8c,
G1-G8, G11).
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
# only the reference may provide `src`: this repository ships a regular package of the same name
# (src/__init__.py), which would shadow the reference's namespace package if it were importable
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:] = [p for p in sys.path if os.path.abspath(p or os.getcwd()) not in (ROOT, HERE)]
sys.path.insert(0, REF)
os.chdir(REF)

import yaml  # noqa: E402
from src.core import NeuralField  # noqa: E402
from src.dataset import BlenderDataset  # noqa: E402
from src.embeddings import FourierRepresentation  # noqa: E402
from src.renderer import (DensityGrid, render_image, render_rays,  # noqa: E402
                          sample_stratified, volume_render)
from src.utils import compute_psnr  # noqa: E402


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"  {name}.npz  " + ", ".join(f"{k}{tuple(np.shape(v))}" for k, v in out.items()))


def synth_rays(n, gen):
    """Cameras on a radius-4.03 sphere looking roughly at the origin."""
    o = torch.randn(n, 3, generator=gen)
    o = o / o.norm(dim=-1, keepdim=True) * 4.0311
    tgt = (torch.rand(n, 3, generator=gen) - 0.5) * 1.6
    d = tgt - o
    d = d / d.norm(dim=-1, keepdim=True)
    return o, d


def g1_fourier():
    g = torch.Generator().manual_seed(101)
    for dim in (1, 2, 3):
        for L in (4, 6, 10, 15):
            x = (torch.rand(257, dim, generator=g) - 0.5) * 3.0
            y = FourierRepresentation(input_dim=dim, L=L)(x)
            save(f"g1_fourier_d{dim}_L{L}", x=x, y=y)


def g2_sampling():
    g = torch.Generator().manual_seed(102)
    for S in (64, 128):
        z0 = sample_stratified(2.0, 6.0, S, 5, "cpu", False)
        # the reference draws its jitter from the global RNG: seed, record, re-seed
        torch.manual_seed(7)
        u_ref = torch.rand(5, S)
        torch.manual_seed(7)
        z1 = sample_stratified(2.0, 6.0, S, 5, "cpu", True)
        save(f"g2_sampling_S{S}", z_plain=z0, u=u_ref, z_jitter=z1)


def g3_mask():
    g = torch.Generator().manual_seed(103)
    for res in (64, 128):
        grid = DensityGrid(resolution=res, bound=1.5, threshold=0.01)
        bits = torch.rand(res, res, res, generator=g) < 0.3
        grid.binary_grid = bits
        pts = (torch.rand(4096, 3, generator=g) - 0.5) * 3.4
        edge = torch.tensor([[-1.505, 0.0, 0.0], [-1.5, -1.5, -1.5], [1.4999, 1.4999, 1.4999],
                             [1.5, 0.0, 0.0], [0.0, -1.52, 0.0], [0.0, 0.0, 1.49999],
                             [-1.5 - 1.0 / 64, 0.1, 0.1], [0.3, 0.3, -1.5234]])
        pts = torch.cat([edge, pts], dim=0)
        idx = ((pts + grid.offset) * grid.scale).long()
        mask = grid.get_active_mask(pts)
        save(f"g3_mask_res{res}", pts=pts, bits=bits, idx=idx, mask=mask)


def g4_decoder():
    cfg = yaml.safe_load(open(os.path.join(REF, "configs", "part2.yaml.example")))
    torch.manual_seed(0)
    model = NeuralField(cfg)
    g = torch.Generator().manual_seed(104)
    pts = (torch.rand(512, 3, generator=g) - 0.5) * 3.0
    dirs = torch.randn(512, 3, generator=g)
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    with torch.no_grad():
        rgb, sigma = model(pts, dirs)
    sd = {k.replace("decoder.", ""): v for k, v in model.state_dict().items() if k.startswith("decoder.")}
    save("g4_decoder", pts=pts, dirs=dirs, rgb=rgb, sigma=sigma, **{"w:" + k: v for k, v in sd.items()})
    return model, sd


def g5_composite():
    g = torch.Generator().manual_seed(105)
    for S in (64, 128):
        R = 48
        z = sample_stratified(2.0, 6.0, S, R, "cpu", True)
        sig = torch.rand(R, S, generator=g) * 4.0
        sig[torch.rand(R, S, generator=g) < 0.4] = 0.0
        sig[torch.rand(R, S, generator=g) < 0.03] = 1e3
        sig[:4] = 0.0                                      # empty rays
        rgb = torch.rand(R, S, 3, generator=g)
        _, d = synth_rays(R, g)
        d = d * (0.5 + torch.rand(R, 1, generator=g))      # non-unit directions
        for tag, bg in (("none", None), ("vec", torch.tensor([0.2, 0.5, 0.9])),
                        ("ray", torch.rand(R, 3, generator=g))):
            rgb_ = rgb.clone().requires_grad_(True)
            sig_ = sig.clone().requires_grad_(True)
            c, dep, acc = volume_render(rgb_, sig_, z, d, bg_color=bg)
            gc = torch.rand(R, 3, generator=g)
            gd = torch.rand(R, generator=g) * 0.1
            ga = torch.rand(R, generator=g) * 0.1
            ((c * gc).sum() + (dep * gd).sum() + (acc * ga).sum()).backward()
            save(f"g5_composite_S{S}_{tag}", z=z, sigma=sig, rgb=rgb, rays_d=d,
                 bg=(np.zeros(0, np.float32) if bg is None else bg),
                 out_rgb=c, out_depth=dep, out_acc=acc, g_rgb_map=gc, g_depth=gd, g_acc=ga,
                 d_sigma=sig_.grad, d_rgb=rgb_.grad)


def g6_render(model):
    g = torch.Generator().manual_seed(106)
    o, d = synth_rays(96, g)
    with torch.no_grad():
        c0, dep0, acc0 = render_rays(model, o, d, 2.0, 6.0, 64, False)
    torch.manual_seed(11)
    u = torch.rand(96, 64)
    torch.manual_seed(11)
    target = torch.rand(96, 3, generator=g)
    model.zero_grad()
    c1, dep1, acc1 = render_rays(model, o, d, 2.0, 6.0, 64, True)
    loss = torch.nn.functional.mse_loss(c1, target)
    loss.backward()
    grads = {"dw:" + k.replace("decoder.", ""): p.grad for k, p in model.named_parameters()}
    save("g6_render", rays_o=o, rays_d=d, rgb_plain=c0, depth_plain=dep0, acc_plain=acc0,
         u=u, target=target, rgb_jitter=c1, depth_jitter=dep1, acc_jitter=acc1,
         loss=loss.detach(), **grads)
    # occupancy-masked path with a synthetic bitfield
    grid = DensityGrid(resolution=128, bound=1.5, threshold=0.01)
    ax = torch.linspace(-1.5, 1.5, 128)
    gx, gy, gz = torch.meshgrid(ax, ax, ax, indexing="ij")
    grid.binary_grid = (gx ** 2 + gy ** 2 + gz ** 2) < 0.8 ** 2
    with torch.no_grad():
        c2, dep2, acc2 = render_rays(model, o, d, 2.0, 6.0, 64, False, density_grid=grid,
                                     bg_color=torch.tensor([0.1, 0.2, 0.3]))
        img = render_image(model, o[:64].reshape(8, 8, 3), d[:64].reshape(8, 8, 3), 2.0, 6.0, 64, 24, True)
    save("g6_render_masked", rays_o=o, rays_d=d, radius=np.float32(0.8),
         bg=np.array([0.1, 0.2, 0.3], np.float32), rgb=c2, depth=dep2, acc=acc2, image8x8=img)


class _Blob(torch.nn.Module):
    mode = "part2_nerf"

    def forward(self, x, d):
        r2 = ((x - torch.tensor([0.2, -0.1, 0.3])) ** 2).sum(-1, keepdim=True)
        return torch.zeros(x.shape[0], 3), 5.0 * torch.exp(-r2 / 0.18)


class _BlobDyn(_Blob):
    mode = "part3"

    def forward(self, x, d, t=None):
        rgb, s = super().forward(x + t * 0.3, d)
        return rgb, s, torch.zeros_like(x)


def g7_grid():
    for res in (32, 64):
        grid = DensityGrid(resolution=res, bound=1.5, threshold=0.12)
        ratio = grid.update(_Blob(), device="cpu")
        save(f"g7_grid_static_res{res}", grid=grid.grid, binary=grid.binary_grid, ratio=np.float64(ratio))
    grid = DensityGrid(resolution=32, bound=1.5, threshold=0.12)
    r1 = grid.update(_BlobDyn(), device="cpu", time=torch.tensor([[0.0]]), decay=0.95)
    r2 = grid.update(_BlobDyn(), device="cpu", time=torch.tensor([[1.0]]), decay=0.95)
    save("g7_grid_dynamic_res32", grid=grid.grid, binary=grid.binary_grid,
         ratios=np.array([r1, r2]), decay=np.float32(0.95))
    sched = [(s, i, w, bool(grid.should_update(s, i, w))) for s in (0, 15, 16, 255, 256, 288, 300)
             for i, w in ((16, 0), (32, 256))]
    save("g7_should_update", table=np.array(sched, dtype=np.int64))


def g8_rays():
    ds = BlenderDataset.__new__(BlenderDataset)
    ds.H, ds.W = 20, 24
    ds.camera_angle_x = 0.6911112070083618
    ds.focal = 0.5 * ds.W / np.tan(0.5 * ds.camera_angle_x)
    ds.scene_scale = 0.8
    ds._directions = ds._build_directions()
    th, ph = 0.7, 0.4
    rot = torch.tensor([[np.cos(th), -np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph)],
                        [np.sin(th), np.cos(th) * np.cos(ph), -np.cos(th) * np.sin(ph)],
                        [0.0, np.sin(ph), np.cos(ph)]], dtype=torch.float32)
    c2w = torch.eye(4)
    c2w[:3, :3] = rot
    c2w[:3, 3] = torch.tensor([1.0, -2.0, 3.0])
    ro, rd = ds.get_rays(c2w)
    save("g8_rays", c2w=c2w, H=np.int64(ds.H), W=np.int64(ds.W), focal=np.float64(ds.focal),
         scene_scale=np.float32(ds.scene_scale), rays_o=ro.contiguous(), rays_d=rd)


def g10_optim():
    torch.manual_seed(5)
    ps = [torch.nn.Parameter(torch.randn(7, 5)), torch.nn.Parameter(torch.randn(11)), torch.nn.Parameter(torch.randn(3, 2))]
    init = [p.detach().clone() for p in ps]
    gs = [[torch.randn_like(p) for p in ps] for _ in range(5)]
    out = {}
    for name, make in (("adam", lambda: torch.optim.Adam(ps, lr=5e-4)),
                       ("adamw", lambda: torch.optim.AdamW(ps, lr=1e-2, weight_decay=1e-5))):
        for p, i in zip(ps, init):
            p.data.copy_(i)
        opt = make()
        sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=2000, eta_min=1e-4) if name == "adamw" else None
        lrs = []
        for step in range(5):
            for p, g in zip(ps, gs[step]):
                p.grad = g.clone()
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            if sch is not None:
                sch.step()
        for k, p in enumerate(ps):
            out[f"{name}_p{k}"] = p.detach().clone()
        out[f"{name}_lrs"] = np.array(lrs)
    for k in range(3):
        out[f"init_p{k}"] = init[k]
        out[f"grads_p{k}"] = torch.stack([gs[s][k] for s in range(5)])
    save("g10_optim", **out)


def g12_part1():
    """configs[0]: 2-D image fit field (Fourier L=15 + StandardMLP), the reference's CPU-runnable case."""
    cfg = {"mode": "part1_fourier", "use_positional_encoding": True, "L_embed": 15, "hidden_dim": 64,
           "num_layers": 3, "output_dim": 3}
    torch.manual_seed(12)
    model = NeuralField(cfg)
    coords = torch.stack(torch.meshgrid(torch.linspace(0, 1, 17), torch.linspace(0, 1, 19), indexing="ij"), -1).reshape(-1, 2)
    with torch.no_grad():
        rgb = model(coords)
    save("g12_part1", coords=coords, rgb=rgb, **{"w:" + k: v for k, v in model.state_dict().items()})


def instant_test_params(n_table):
    """Deterministic parameters of the instant goldens (the same formulas rebuild them in the tests, so the
    131,072-float table need not be stored): table = 0.5 sin(0.37 i + 0.11 (i mod 7)); nets: seeded uniform."""
    i = torch.arange(n_table, dtype=torch.float64)
    return (0.5 * torch.sin(0.37 * i + 0.11 * (i % 7))).float()


def g13_instant_glue():
    """The reference's own instant GLUE around a stand-in tinycudann (tests/golden/tinycudann_shim.py):
    NeuralField('part2_instant').forward, masked render_rays (+ autograd of an MSE loss), DensityGrid.update."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("tinycudann", os.path.join(HERE, "tinycudann_shim.py"))
    shim = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shim)
    sys.modules["tinycudann"] = shim
    cfg = {"mode": "part2_instant", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 12,
           "base_resolution": 16, "per_level_scale": 1.5, "scene_bound": 1.5, "L_embed_dir": 4, "hidden_dim": 64}
    torch.manual_seed(13)
    model = NeuralField(cfg)
    assert sorted(model.state_dict()) == ["decoder.color_net.params", "decoder.sigma_net.params",
                                          "dir_representation.freq_bands", "representation.encoding.params"]
    with torch.no_grad():
        t = model.representation.encoding.params
        t.copy_(instant_test_params(t.numel()))
        # visible densities: only the density row of the sigma head is enlarged (h0 - 5 reaches positive
        # values); the other 15 geometry channels and the colour net keep their initial scale
        model.decoder.sigma_net.params[:2048].mul_(1.5)
        model.decoder.sigma_net.params[2048:2048 + 64].mul_(24.0)
    gen = torch.Generator().manual_seed(31)
    # ---- field forward: points inside, on and outside the box (clamp path), unit directions
    pts = (torch.rand(700, 3, generator=gen) - 0.5) * 3.3
    pts[:4] = torch.tensor([[1.5, -1.5, 0.0], [-1.5, 1.5, 1.5], [1.6, 0.0, -1.7], [0.0, 0.0, 0.0]])
    dirs = torch.nn.functional.normalize(torch.randn(700, 3, generator=gen), dim=-1)
    with torch.no_grad():
        rgb, sigma = model(pts, dirs)
        x_enc = model.representation(pts)
    # ---- masked render_rays + MSE backward through the reference's scatter
    o, d = synth_rays(160, gen)
    u = torch.rand(160, 64, generator=gen)
    target = torch.rand(160, 3, generator=gen)
    grid = DensityGrid(resolution=64, bound=1.5, threshold=0.01)
    ax = torch.linspace(-1.5, 1.5, 64)
    gx, gy, gz = torch.meshgrid(ax, ax, ax, indexing="ij")
    grid.binary_grid = (gx ** 2 + gy ** 2 + gz ** 2) < 1.1 ** 2
    bg = torch.tensor([0.2, 0.4, 0.6])
    with torch.no_grad():
        c0, dep0, acc0 = render_rays(model, o, d, 2.0, 6.0, 64, False, density_grid=grid, bg_color=bg)
    import src.renderer as R
    orig = torch.rand
    torch.rand = lambda *a, **k: u.clone()                # the reference draws its jitter with torch.rand
    try:
        c1, dep1, acc1 = render_rays(model, o, d, 2.0, 6.0, 64, True, density_grid=grid, bg_color=bg)
    finally:
        torch.rand = orig
    loss = torch.nn.functional.mse_loss(c1, target)
    model.zero_grad()
    loss.backward()
    g_table = model.representation.encoding.params.grad
    nz = torch.nonzero(g_table).flatten()
    # ---- DensityGrid.update around the instant field (eval mode, zero view directions)
    model.eval()
    dg = DensityGrid(resolution=32, bound=1.5, threshold=0.05)
    ratio = dg.update(model, device="cpu")
    save("g13_instant_glue", sigma_net=model.decoder.sigma_net.params, color_net=model.decoder.color_net.params,
         table_probe=model.representation.encoding.params[::4099], pts=pts, dirs=dirs, rgb=rgb, sigma=sigma, x_enc=x_enc,
         rays_o=o, rays_d=d, u=u, target=target, bg=bg, radius=np.float32(1.1),
         rgb_plain=c0, depth_plain=dep0, acc_plain=acc0, rgb_jitter=c1, depth_jitter=dep1, acc_jitter=acc1, loss=loss,
         g_sigma_net=model.decoder.sigma_net.params.grad, g_color_net=model.decoder.color_net.params.grad,
         g_table_index=nz[::7], g_table_value=g_table[nz[::7]], g_table_norm=g_table.norm(), g_table_nonzero=np.int64(nz.numel()),
         grid=dg.grid, binary=dg.binary_grid, ratio=np.float64(ratio))
    del sys.modules["tinycudann"]


PART4_CFG = {"mode": "part4", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 11, "base_resolution": 16,
             "per_level_scale": 1.5, "scene_bound": 1.5, "L_embed_dir": 4, "L_embed_time": 10, "hidden_dim": 64,
             "time_modulation_dim": 64, "time_modulation_layers": 2, "deform_n_levels": 12, "deform_n_features_per_level": 2,
             "deform_log2_hashmap_size": 10, "deform_base_resolution": 16, "deform_per_level_scale": 1.5, "deform_hidden_dim": 64}


def part4_table(n, phase):
    i = torch.arange(n, dtype=torch.float64)
    return (0.5 * torch.sin(0.37 * i + phase + 0.11 * (i % 7))).float()


def g14_part4():
    """The reference's Part 4 dual-hash field (src/core.py:148-225, 282-352) around the stand-in tinycudann:
    forward (rgb, sigma, delta_x), its autograd (incl. the path through d features / d x of the canonical grid),
    render_rays with times (4-tuple, mean_delta_x) behind an occupancy grid, DensityGrid.update at the three anchors."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("tinycudann", os.path.join(HERE, "tinycudann_shim.py"))
    shim = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shim)
    sys.modules["tinycudann"] = shim
    torch.manual_seed(14)
    model = NeuralField(dict(PART4_CFG))
    with torch.no_grad():
        for k, (name, ph) in enumerate((("canonical_repr", 0.0), ("deform_grid_start", 1.0), ("deform_grid_mid", 2.0), ("deform_grid_end", 3.0))):
            t = getattr(model, name).encoding.params
            t.copy_(part4_table(t.numel(), ph))
        model.decoder.sigma_net.params[:64 * 64].mul_(1.5)
        model.decoder.sigma_net.params[64 * 64:64 * 64 + 64].mul_(24.0)     # the density row: visible densities
        model.deform_decoder.deform_net.params.mul_(2.0)
    sd = {k: v.clone() for k, v in model.state_dict().items() if "encoding.params" not in k and "freq_bands" not in k}
    gen = torch.Generator().manual_seed(41)
    n = 400
    pts = (torch.rand(n, 3, generator=gen) - 0.5) * 3.2
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)
    times = torch.rand(n, 1, generator=gen)
    times[:3] = torch.tensor([[0.0], [0.5], [1.0]])
    model.eval()
    rgb, sigma, delta = model(pts, dirs, t=times)
    w_rgb, w_dx = torch.randn(n, 3, generator=gen), torch.randn(n, 3, generator=gen)
    model.zero_grad()
    ((rgb * w_rgb).sum() + sigma.sum() + (delta * w_dx).sum()).backward()
    grads = {}
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        if "encoding.params" in k:
            nz = torch.nonzero(p.grad).flatten()[::5]
            grads["gi:" + k], grads["gv:" + k], grads["gn:" + k] = nz, p.grad[nz], p.grad.norm()
        else:
            grads["g:" + k] = p.grad.clone()
    # ---- render_rays with times behind an occupancy grid
    o, d = synth_rays(96, gen)
    ray_t = torch.rand(96, 1, generator=gen)
    grid = DensityGrid(resolution=64, bound=1.5, threshold=0.01)
    ax = torch.linspace(-1.5, 1.5, 64)
    gx, gy, gz = torch.meshgrid(ax, ax, ax, indexing="ij")
    grid.binary_grid = (gx ** 2 + gy ** 2 + gz ** 2) < 1.1 ** 2
    with torch.no_grad():
        c, dep, acc, extras = render_rays(model, o, d, 2.0, 6.0, 48, False, density_grid=grid, times=ray_t,
                                          bg_color=torch.tensor([0.2, 0.4, 0.6]))
        c3 = render_rays(model, o[:8], d[:8], 2.0, 6.0, 48, False)                  # no times: 3-tuple at t = 0
        dg = DensityGrid(resolution=24, bound=1.5, threshold=0.05)
        r1 = dg.update(model, device="cpu", decay=0.95)
        r2 = dg.update(model, device="cpu", decay=0.95)
    # ---- the loss terms of run_part4 (run.py:1835-1938) composed from the REFERENCE's operators on stored probes:
    # the loop draws them with torch.rand on the device; the terms themselves are deterministic functions of them
    reg_gen = torch.Generator().manual_seed(77)
    eps = 0.02
    probes = {"temporal_x": (torch.rand(64, 3, generator=reg_gen) * 2 - 1) * 1.5, "temporal_t": torch.rand(64, 1, generator=reg_gen) * (1 - eps),
              "unsup_x": (torch.rand(128, 3, generator=reg_gen) * 2 - 1) * 1.5, "unsup_t": torch.rand(128, 1, generator=reg_gen),
              "anchor_x": (torch.rand(128, 3, generator=reg_gen) * 2 - 1) * 1.5}
    disp = lambda grid_, x_, t_: model.deform_decoder(grid_(x_), model.time_modulation(model.time_encoder(t_)))
    model.zero_grad()
    tv = lambda p_: torch.mean(torch.abs(p_[1:] - p_[:-1]))
    reg = {
        "reg": torch.mean(extras["mean_delta_x"] ** 2) * 0.01,
        "tv_disp": sum(tv(getattr(model, n_).encoding.params) for n_ in ("deform_grid_start", "deform_grid_mid", "deform_grid_end")) * 0.001 / 3.0,
        "tv_canon": tv(model.canonical_repr.encoding.params) * 1e-5,
    }
    feat_ = model.deformation_grid(probes["temporal_x"])
    d0_ = model.deform_decoder(feat_, model.time_modulation(model.time_encoder(probes["temporal_t"])))
    d1_ = model.deform_decoder(feat_, model.time_modulation(model.time_encoder(probes["temporal_t"] + eps)))
    reg["temporal"] = torch.mean((d0_ - d1_) ** 2) * 1e-4 * 16
    reg["unsup"] = torch.mean(torch.abs(disp(model.deformation_grid, probes["unsup_x"], probes["unsup_t"]).mean(dim=0))) * 0.001 * 32
    at0_ = disp(model.deform_grid_start, probes["anchor_x"], torch.zeros(128, 1))
    t6_ = torch.full((128, 1), 1.0 / 6.0)
    cons_ = torch.mean((disp(model.deform_grid_start, probes["anchor_x"], t6_) - disp(model.deform_grid_mid, probes["anchor_x"], t6_)) ** 2) * 0.1
    reg["anchor"] = (torch.mean(at0_ ** 2) + cons_) * 0.01 * 16
    (reg["temporal"] + reg["unsup"] + reg["anchor"]).backward()
    reg_grads = {"rg:" + k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None and "encoding.params" not in k}
    reg_grads["rgn:deform_grid_start"] = model.deform_grid_start.encoding.params.grad.norm()
    reg_grads["rgn:deform_grid_mid"] = model.deform_grid_mid.encoding.params.grad.norm()
    save("g14_part4", pts=pts, dirs=dirs, times=times, rgb=rgb, sigma=sigma, delta=delta, w_rgb=w_rgb, w_dx=w_dx,
         **{"probe:" + k: v for k, v in probes.items()}, **{"reg:" + k: v.detach() for k, v in reg.items()}, **reg_grads,
         rays_o=o, rays_d=d, ray_t=ray_t, r_rgb=c, r_depth=dep, r_acc=acc, r_mean_delta=extras["mean_delta_x"],
         r3_rgb=c3[0], n_tuple3=np.int64(len(c3)), grid=dg.grid, binary=dg.binary_grid, ratios=np.array([r1, r2]),
         **{"w:" + k: v for k, v in sd.items()}, **grads)
    del sys.modules["tinycudann"]


def trained_spectrum_table(levels, seed):
    """Hash-table values with the spectrum of a trained field: amplitude 8 / resolution per level (uniform), i.e. every level
    contributes a bounded d feature / d x -- unlike part4_table's amplitude 0.5 at every level, where a 1e-3 shift of
    x_canonical lands in unrelated cells of the fine levels and any implementation's rounding is amplified without bound."""
    g = torch.Generator().manual_seed(seed)
    return torch.cat([(torch.rand(lv.size * 2, generator=g) - 0.5) * (8.0 / lv.res) for lv in levels])


def g14b_part4_trained():
    """g14's forward + autograd of the reference's NeuralField('part4') (src/core.py:282-352) on tables with a trained
    spectrum and the reference's DEFAULT displacement scale (0.1, src/decoders.py:296): the vector the fused HIP chains --
    the path bench.py times for configs[4] -- are compared with directly (rgb, sigma, delta_x and every parameter gradient,
    the four tables in full)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("tinycudann", os.path.join(HERE, "tinycudann_shim.py"))
    shim = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shim)
    sys.modules["tinycudann"] = shim
    torch.manual_seed(14)
    model = NeuralField(dict(PART4_CFG))
    names = ("canonical_repr", "deform_grid_start", "deform_grid_mid", "deform_grid_end")
    tables = {}
    with torch.no_grad():
        for k, name in enumerate(names):
            enc = getattr(model, name).encoding
            enc.params.copy_(trained_spectrum_table(enc.levels, 140 + k))
            tables["t:" + name] = enc.params.detach().clone()
        model.decoder.sigma_net.params[:64 * 64].mul_(1.5)
        model.decoder.sigma_net.params[64 * 64:64 * 64 + 64].mul_(24.0)     # the density row: visible densities
        model.deform_decoder.deform_net.params.mul_(2.0)
    assert abs(float(model.deform_decoder.displacement_scale) - 0.1) < 1e-7   # the reference's default
    sd = {k: v.clone() for k, v in model.state_dict().items() if "encoding.params" not in k and "freq_bands" not in k}
    gen = torch.Generator().manual_seed(43)
    n = 512
    pts = (torch.rand(n, 3, generator=gen) - 0.5) * 3.0
    pts[:16] = (torch.rand(16, 3, generator=gen) - 0.5) * 3.3              # a few outside the box (clamped lookups)
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)
    times = torch.rand(n, 1, generator=gen)
    times[:3] = torch.tensor([[0.0], [0.5], [1.0]])
    model.eval()
    rgb, sigma, delta = model(pts, dirs, t=times)
    w_rgb, w_dx = torch.randn(n, 3, generator=gen), torch.randn(n, 3, generator=gen)
    model.zero_grad()
    ((rgb * w_rgb).sum() + sigma.sum() + (delta * w_dx).sum()).backward()
    keep = lambda k: not k.startswith("deformation_grid.")
    grads = {"g:" + k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None and keep(k)}
    # the same forward + autograd with the stand-in's outputs rounded to fp16, tinycudann's output precision
    shim.FP16_OUTPUTS = True
    try:
        rgb16, sigma16, delta16 = model(pts, dirs, t=times)
        model.zero_grad()
        ((rgb16 * w_rgb).sum() + sigma16.sum() + (delta16 * w_dx).sum()).backward()
        grads16 = {"g16:" + k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None and keep(k)}
    finally:
        shim.FP16_OUTPUTS = False
    rel = lambda a, b: float((a - b).norm() / b.norm())
    print(f"  g14b: max |delta_x| {float(delta.abs().max()):.3e}, sigma range {float(sigma.min()):.3e} .. {float(sigma.max()):.3e}, "
          f"{int((sigma > 0.5).sum())} of {n} with sigma > 0.5")
    print(f"  g14b, fp16 operator outputs vs fp32: delta_x {float((delta16 - delta).abs().max() / delta.abs().max()):.2e} of max, "
          f"|d rgb| {float((rgb16 - rgb).abs().max()):.2e}; gradients rel: " +
          ", ".join(f"{k[2:].split('.')[0]}.{k.split('.')[-2]} {rel(grads16['g16:' + k[2:]], v):.3f}" for k, v in grads.items()))
    save("g14b_part4_trained", pts=pts, dirs=dirs, times=times, rgb=rgb, sigma=sigma, delta=delta, w_rgb=w_rgb, w_dx=w_dx,
         rgb16=rgb16, sigma16=sigma16, delta16=delta16,
         **tables, **{"w:" + k: v for k, v in sd.items()}, **grads, **grads16)
    del sys.modules["tinycudann"]


PART3_CFGS = {
    "nerf": {"mode": "part3", "canonical_type": "nerf", "L_embed": 6, "L_embed_canon": 8, "L_embed_dir": 4, "L_embed_time": 6,
             "hidden_dim": 64, "num_layers": 5, "skip_layer": 3, "view_dim": 32, "deform_hidden_dim": 48, "deform_num_layers": 3},
    "dtc": {"mode": "part3", "canonical_type": "nerf", "direct_time_conditioning": True, "L_embed": 6, "L_embed_canon": 8,
            "L_embed_dir": 4, "L_embed_time": 6, "hidden_dim": 64, "num_layers": 5, "skip_layer": 3, "view_dim": 32,
            "deform_hidden_dim": 48, "deform_num_layers": 3},
    "instant": {"mode": "part3", "canonical_type": "instant", "L_embed": 6, "L_embed_dir": 4, "L_embed_time": 10, "hidden_dim": 64,
                "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 11, "base_resolution": 16, "per_level_scale": 1.5,
                "scene_bound": 1.5, "deform_hidden_dim": 48, "deform_num_layers": 3},
}


def g15_part3():
    """The reference's Part 3 field (src/core.py:79-146, 233-281): MLP deformation + canonical NeRF MLP, direct time
    conditioning, and the hash-grid canonical variant around the stand-in tinycudann -- forward (rgb, sigma, delta_x),
    gradients of a weighted sum (the deformation MLP is reached only through d code / d x of the canonical encoding),
    render_rays with times and DensityGrid.update(time=...) with the running maximum."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("tinycudann", os.path.join(HERE, "tinycudann_shim.py"))
    shim = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shim)
    sys.modules["tinycudann"] = shim
    out = {}
    gen = torch.Generator().manual_seed(53)
    n = 300
    pts = (torch.rand(n, 3, generator=gen) - 0.5) * 2.6
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)
    times = torch.rand(n, 1, generator=gen)
    w_rgb, w_dx = torch.randn(n, 3, generator=gen), torch.randn(n, 3, generator=gen)
    o, d = synth_rays(64, gen)
    ray_t = torch.rand(64, 1, generator=gen)
    out.update(pts=pts, dirs=dirs, times=times, w_rgb=w_rgb, w_dx=w_dx, rays_o=o, rays_d=d, ray_t=ray_t)
    for tag, cfg in PART3_CFGS.items():
        torch.manual_seed(15)
        model = NeuralField(dict(cfg))
        with torch.no_grad():
            model.deform_net.net[-1].weight.mul_(3000.0)                  # visible displacements (init is 1e-4)
            model.deform_net.net[-1].bias.add_(0.02)
            if tag == "instant":
                t = model.canonical_repr.encoding.params
                t.copy_(part4_table(t.numel(), 0.5))
                model.decoder.sigma_net.params[:64 * 64].mul_(1.5)
                model.decoder.sigma_net.params[64 * 64:64 * 64 + 64].mul_(24.0)
            else:
                dec = model.decoder_direct if tag == "dtc" else model.decoder
                dec.sigma_layer.bias.add_(0.3)
                dec.sigma_layer.weight.mul_(4.0)
        model.eval()
        sd = {k: v.clone() for k, v in model.state_dict().items() if "encoding.params" not in k and "freq_bands" not in k}
        rgb, sigma, delta = model(pts, dirs, t=times)
        model.zero_grad()
        ((rgb * w_rgb).sum() + sigma.sum() + (delta * w_dx).sum()).backward()
        for k, p in model.named_parameters():
            if p.grad is None:
                continue
            if "encoding.params" in k:
                out[f"{tag}:gn:{k}"] = p.grad.norm()
            else:
                out[f"{tag}:g:{k}"] = p.grad.clone()
        grid = DensityGrid(resolution=64, bound=1.5, threshold=0.01)
        ax = torch.linspace(-1.5, 1.5, 64)
        gx, gy, gz = torch.meshgrid(ax, ax, ax, indexing="ij")
        grid.binary_grid = (gx ** 2 + gy ** 2 + gz ** 2) < 1.1 ** 2
        with torch.no_grad():
            c, dep, acc, extras = render_rays(model, o, d, 2.0, 6.0, 40, False, density_grid=grid, times=ray_t,
                                              bg_color=torch.tensor([0.2, 0.4, 0.6]))
            dg = DensityGrid(resolution=20, bound=1.5, threshold=0.05)
            r1 = dg.update(model, device="cpu", time=torch.tensor([[0.25]]), decay=0.9)
            r2 = dg.update(model, device="cpu", time=torch.tensor([[0.75]]), decay=0.9)
        out.update({f"{tag}:rgb": rgb, f"{tag}:sigma": sigma, f"{tag}:delta": delta, f"{tag}:r_rgb": c, f"{tag}:r_depth": dep,
                    f"{tag}:r_acc": acc, f"{tag}:r_mean_delta": extras["mean_delta_x"], f"{tag}:grid": dg.grid,
                    f"{tag}:binary": dg.binary_grid, f"{tag}:ratios": np.array([r1, r2]),
                    f"{tag}:n_params": np.int64(sum(p.numel() for p in model.parameters()))})
        out.update({f"{tag}:w:{k}": v for k, v in sd.items()})
    save("g15_part3", **out)
    del sys.modules["tinycudann"]


def g11_psnr():
    mse = np.array([1e-4, 3.3e-3, 0.02, 0.25])
    save("g11_psnr", mse=mse, psnr=np.array([compute_psnr(m) for m in mse]))


if __name__ == "__main__":
    torch.set_num_threads(8)
    import src.core as _probe
    assert _probe.__file__.startswith(REF), _probe.__file__
    print("writing golden vectors from", REF)
    if len(sys.argv) > 1:            # regenerate selected groups only, e.g. `make_golden.py g12_part1`
        for name in sys.argv[1:]:
            globals()[name]()
        raise SystemExit(0)
    g1_fourier()
    g2_sampling()
    g3_mask()
    model, _ = g4_decoder()
    g5_composite()
    g6_render(model)
    g7_grid()
    g8_rays()
    g10_optim()
    g11_psnr()
    g12_part1()
    g13_instant_glue()
    g14_part4()
    g14b_part4_trained()
    g15_part3()
