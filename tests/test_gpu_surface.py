"""The reference's module surface (src/*) on the MI355X kernels: checkpoint keys, render_rays /
render_image / DensityGrid against the reference's golden outputs, a short training run."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import yaml

from conftest import ROOT, golden
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.fixture(scope="module")
def field():
    assert torch.cuda.is_available()
    from src.core import NeuralField
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "part2.yaml.example")))
    model = NeuralField(cfg).cuda()
    g = golden("g4_decoder")
    sd = model.state_dict()
    for k, v in g.items():
        if k.startswith("w:"):
            assert "decoder." + k[2:] in sd, k          # reference checkpoint keys load unchanged
            sd["decoder." + k[2:]] = T(v)
    model.load_state_dict(sd)
    return model


def test_state_dict_keys_match_reference(field):
    keys = [k for k in field.state_dict() if k.startswith("decoder.")]
    assert keys == ["decoder." + k for k, _ in O.nerf_param_shapes()]
    assert "representation.freq_bands" in field.state_dict() and "dir_representation.freq_bands" in field.state_dict()


def test_neural_field_forward_and_errors(field):
    g = golden("g4_decoder")
    with torch.no_grad():
        rgb, sigma = field(T(g["pts"]).cuda(), T(g["dirs"]).cuda())
    assert rgb.shape == (512, 3) and sigma.shape == (512, 1)
    np.testing.assert_allclose(rgb.cpu().numpy(), g["rgb"], atol=2e-2)
    with pytest.raises(ValueError):
        field(T(g["pts"]).cuda())


def test_render_rays_and_image_vs_reference_golden(field):
    from src.renderer import render_image, render_rays
    g = golden("g6_render")
    o, d = T(g["rays_o"]).cuda(), T(g["rays_d"]).cuda()
    with torch.no_grad():
        c, dep, acc = render_rays(field, o, d, 2.0, 6.0, 64, False)
    # stated tolerance for the bf16 decoder vs the reference's fp32 renderer
    np.testing.assert_allclose(c.cpu().numpy(), g["rgb_plain"], atol=1e-2)
    np.testing.assert_allclose(dep.cpu().numpy(), g["depth_plain"], rtol=2e-2)
    np.testing.assert_allclose(acc.cpu().numpy(), g["acc_plain"], atol=1e-2)
    gm = golden("g6_render_masked")
    with torch.no_grad():
        img = render_image(field, o[:64].reshape(8, 8, 3), d[:64].reshape(8, 8, 3), 2.0, 6.0, 64, 24, True)
    np.testing.assert_allclose(img.cpu().numpy(), gm["image8x8"], atol=1e-2)


def test_render_rays_with_density_grid_vs_reference_golden(field):
    from src.renderer import DensityGrid, render_rays
    gm = golden("g6_render_masked")
    grid = DensityGrid(resolution=128, bound=1.5, threshold=0.01).cuda()
    ax = torch.linspace(-1.5, 1.5, 128)
    gx, gy, gz = torch.meshgrid(ax, ax, ax, indexing="ij")
    grid.binary_grid = ((gx ** 2 + gy ** 2 + gz ** 2) < float(gm["radius"]) ** 2).cuda()
    o, d = T(gm["rays_o"]).cuda(), T(gm["rays_d"]).cuda()
    with torch.no_grad():
        c, dep, acc = render_rays(field, o, d, 2.0, 6.0, 64, False, density_grid=grid, bg_color=T(gm["bg"]).cuda())
    np.testing.assert_allclose(c.cpu().numpy(), gm["rgb"], atol=1e-2)
    np.testing.assert_allclose(acc.cpu().numpy(), gm["acc"], atol=1e-2)


class _Blob(torch.nn.Module):
    mode = "part2_nerf"

    def forward(self, x, d):
        c = torch.tensor([0.2, -0.1, 0.3], device=x.device)
        return torch.zeros(x.shape[0], 3, device=x.device), 5.0 * torch.exp(-((x - c) ** 2).sum(-1, keepdim=True) / 0.18)


@pytest.mark.parametrize("res", [32, 64])
def test_density_grid_update_vs_reference_golden(res):
    from src.renderer import DensityGrid
    g = golden(f"g7_grid_static_res{res}")
    grid = DensityGrid(resolution=res, bound=1.5, threshold=0.12).cuda()
    ratio = grid.update(_Blob(), device="cuda")
    np.testing.assert_allclose(grid.grid.cpu().numpy(), g["grid"], rtol=1e-4, atol=1e-6)
    flips = (grid.binary_grid.cpu().numpy() != g["binary"]).sum()
    assert flips <= 2                                   # cells within 1 ulp of the threshold
    assert abs(ratio - float(g["ratio"])) < 3.0 / res ** 3
    table = golden("g7_should_update")["table"]
    assert all(grid.should_update(int(s), int(i), int(w)) == bool(want) for s, i, w, want in table)


def test_torch_optimizer_trains_through_module_surface(field):
    """Drop-in: nn.Parameters + torch.optim.Adam + render_rays + MSE, as in the reference loop."""
    import copy
    from src.renderer import render_rays
    model = copy.deepcopy(field)
    opt = torch.optim.Adam(model.parameters(), lr=5e-4)
    g = golden("g6_render")
    o, d, tgt = T(g["rays_o"]).cuda(), T(g["rays_d"]).cuda(), T(g["target"]).cuda()
    losses = []
    for _ in range(12):
        pred, _, _ = render_rays(model, o, d, 2.0, 6.0, 64, True)
        loss = torch.nn.functional.mse_loss(pred, tgt)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0]
    assert all(p.grad is not None for p in model.parameters())


def test_engine_convergence_on_synthetic_scene(tmp_path):
    """Flat-parameter fast path: PSNR on an analytic scene rises within a few hundred steps."""
    from src.dataset import BlenderDataset, write_synthetic_scene
    from project_nerf_amd.engine import VanillaNerfEngine
    root = write_synthetic_scene(str(tmp_path / "scene"), n_train=12, n_test=2, size=64)
    ds = BlenderDataset(root, "train", 1, True, 1.0).to("cuda")
    # a bare-ReLU density head can die in the first steps on a mostly-white scene (all gradients exactly zero
    # from then on; rounding-level noise decides, see tests/test_gpu_trained_parity.py): restart from the next seed
    for seed in range(8):
        eng = VanillaNerfEngine(seed=seed, lr=5e-4)
        torch.manual_seed(seed)
        first = last = None
        for step in range(500):
            o, d, rgba = ds.sample_random_rays(4096, "cuda")
            target = rgba[:, :3] * rgba[:, 3:4] + (1 - rgba[:, 3:4])
            loss = eng.train_step(o, d, target, 64)
            if step == 0:
                first = loss.item()
            if step == 150 and loss.item() > 0.1:
                break
        else:
            break
    last = loss.item()
    assert last < 0.5 * first, (first, last)
    o, d, tgt = BlenderDataset(root, "test", 1, True, 1.0).get_image_rays(0, "cuda")
    img = eng.render_image(o, d, 64, chunk=4096)
    psnr = -10 * np.log10(float(((img - tgt) ** 2).mean()))
    assert psnr > 17.0, psnr      # ~19-22 dB after 500 steps; 25+ dB after 1000


@pytest.mark.parametrize("engine", [True, False])
def test_run_py_cli_trains_and_evaluates(tmp_path, engine):
    """run.py part2: the default decoder shape trains on the flat-parameter engine (weights copied into the NeuralField
    for the checkpoint and the evaluation), `engine: false` on NeuralField + torch.optim.Adam.  Some initialisations
    start with every density negative (bare-ReLU head: no gradient at all): the test moves on
    to the next seed then."""
    from src.dataset import write_synthetic_scene
    root = write_synthetic_scene(str(tmp_path / "scene"), n_train=6, n_test=1, size=32)
    moved = 0.0
    for seed in range(4):
        out = tmp_path / f"out{seed}"
        cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "part2.yaml.example")))
        cfg.update(train_iters=20, batch_size=512, log_every=10, save_every=10, downscale=1, log_dir=str(out), engine=engine, seed=seed)
        cfg_path = tmp_path / f"part2_{seed}.yaml"
        cfg_path.write_text(yaml.safe_dump(cfg))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "run.py"), "--config", str(cfg_path), "--data_dir", root,
                            "--render_n", "1"], capture_output=True, text=True, cwd=ROOT, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "Test PSNR" in r.stdout
        ckpt = torch.load(out / "checkpoints" / "model_final.pth", map_location="cpu")
        assert set(ckpt) == {"model_state_dict", "config"}
        assert "decoder.pts_layers.4.weight" in ckpt["model_state_dict"]
        mid = torch.load(out / "checkpoints" / "model_step_000010.pth", map_location="cpu")["model_state_dict"]
        moved = float((mid["decoder.pts_layers.4.weight"] - ckpt["model_state_dict"]["decoder.pts_layers.4.weight"]).abs().max())
        if moved > 0.0:
            break
    assert moved > 0.0            # the weights in the checkpoints follow the training (engine weights are synced)


def test_part1_image_fit_field_vs_reference_golden():
    """BASELINE configs[0] (2-D fit): HIP Fourier features + library-GEMM MLP vs the reference's forward."""
    from src.core import NeuralField
    g = golden("g12_part1")
    cfg = {"mode": "part1_fourier", "use_positional_encoding": True, "L_embed": 15, "hidden_dim": 64,
           "num_layers": 3, "output_dim": 3}
    model = NeuralField(cfg)
    sd = {k[2:]: T(v) for k, v in g.items() if k.startswith("w:")}
    assert set(sd) == set(model.state_dict())
    model.load_state_dict(sd)
    model = model.cuda()
    with torch.no_grad():
        rgb = model(T(g["coords"]).cuda())
    np.testing.assert_allclose(rgb.cpu().numpy(), g["rgb"], atol=2e-5)
    # and it trains: fit a tiny procedural image for a few steps
    img = (0.5 + 0.5 * torch.sin(T(g["coords"]) * 12).repeat(1, 2)[:, :3]).cuda()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    first = None
    for _ in range(60):
        loss = torch.nn.functional.mse_loss(model(T(g["coords"]).cuda()), img)
        opt.zero_grad(); loss.backward(); opt.step()
        first = first if first is not None else loss.item()
    assert loss.item() < first


def test_hierarchical_render_matches_oracle_composition():
    """coarse pass -> sample_pdf -> fine pass (opt-in extension) against the same pipeline built from oracle pieces."""
    from project_nerf_amd.engine import VanillaNerfEngine
    g = golden("g6_render")
    params = {k[2:]: T(v) for k, v in golden("g4_decoder").items() if k.startswith("w:")}
    eng = VanillaNerfEngine(params=torch.cat([params[k].reshape(-1) for k, _ in O.nerf_param_shapes()]))
    o, d = T(g["rays_o"]), T(g["rays_d"])
    c, dep, acc = eng.render_rays_hierarchical(o.cuda(), d.cuda(), 64, 128)
    field = lambda p, v: O.nerf_field(params, p, v)
    z = O.stratified_depths(2.0, 6.0, 64, 96, False)
    pts, dirs = O.ray_points(o, d, z)
    with torch.no_grad():
        rgb, sig = field(pts, dirs)
        w = O.composite(rgb.view(96, 64, 3), sig.view(96, 64), z, d, torch.ones(3), return_weights=True)[3]
        z_all = O.sample_pdf(z, w, 128)
        pts, dirs = O.ray_points(o, d, z_all)
        rgb, sig = field(pts, dirs)
        c_ref, dep_ref, acc_ref = O.composite(rgb.view(96, 192, 3), sig.view(96, 192), z_all, d, torch.ones(3))
    np.testing.assert_allclose(c.cpu().numpy(), c_ref.numpy(), atol=1.5e-2)       # bf16 field, resampled depths
    np.testing.assert_allclose(acc.cpu().numpy(), acc_ref.numpy(), atol=1.5e-2)


@pytest.mark.gpu
def test_gather_rays_matches_host_formula(tmp_path):
    """nerf_gather_rays (GPU-resident batch sampling) vs the reference formula evaluated on the host
    (src/dataset.py:150-171): origins and pixels exact, directions to 1 ulp-level (the reference's bmm
    and this kernel sum three products in possibly different order)."""
    import project_nerf_amd  # noqa: F401
    from project_nerf_amd import ops
    from src.dataset import BlenderDataset, write_synthetic_scene
    root = write_synthetic_scene(str(tmp_path / "scene"), n_train=5, n_test=1, size=48)
    for scale in (1.0, 0.5):
        cpu = BlenderDataset(root, "train", 1, True, scale)
        g = torch.Generator().manual_seed(3)
        B = 3000
        img = torch.randint(0, len(cpu), (B,), generator=g)
        py = torch.randint(0, cpu.H, (B,), generator=g)
        px = torch.randint(0, cpu.W, (B,), generator=g)
        c2w = cpu.poses[img]
        dirs = torch.stack([(px - cpu.W * 0.5) / cpu.focal, -(py - cpu.H * 0.5) / cpu.focal, -torch.ones_like(px)], dim=-1)
        d_ref = torch.bmm(c2w[:, :3, :3], dirs.unsqueeze(-1)).squeeze(-1)
        d_ref = d_ref / torch.norm(d_ref, dim=-1, keepdim=True)
        o_ref = c2w[:, :3, 3] * scale if scale != 1.0 else c2w[:, :3, 3]
        o, d, rgba = ops.gather_rays(cpu.images.cuda(), cpu.poses.cuda(), img.cuda(), py.cuda(), px.cuda(), cpu.focal, scale)
        assert torch.equal(o.cpu(), o_ref) and torch.equal(rgba.cpu(), cpu.images[img, py, px])
        np.testing.assert_allclose(d.cpu().numpy(), d_ref.numpy(), rtol=0, atol=3e-7)
    # the dataset takes the kernel path when its frames live on the GPU
    gpu = BlenderDataset(root, "train", 1, True, 1.0).to("cuda")
    o, d, rgba = gpu.sample_random_rays(257, "cuda")
    assert o.shape == (257, 3) and d.shape == (257, 3) and rgba.shape == (257, 4) and o.is_cuda
    np.testing.assert_allclose(d.norm(dim=-1).cpu().numpy(), 1.0, atol=1e-6)


@pytest.mark.gpu
def test_train_batch_kernel_statistics_and_formulas(tmp_path):
    """nerf_train_batch: rays / targets equal the gather kernel's for the pixels it drew, depths obey the
    stratified-jitter formula, and both draws are uniform (chi-square over frames / rows / columns, moments of u)."""
    import project_nerf_amd  # noqa: F401
    from project_nerf_amd import ops
    from src.dataset import BlenderDataset, write_synthetic_scene
    root = write_synthetic_scene(str(tmp_path / "scene"), n_train=6, n_test=1, size=40)
    ds = BlenderDataset(root, "train", 1, True, 1.0).to("cuda")
    bg = torch.tensor([1.0, 0.5, 0.25], device="cuda")
    B, S = 60000, 64
    o, d, target, z = ds.train_batch(B, S, 2.0, 6.0, bg, seed=7, counter=0)
    o2, d2, t2, z2 = ds.train_batch(B, S, 2.0, 6.0, bg, seed=7, counter=0)
    assert torch.equal(o, o2) and torch.equal(z, z2)                       # counter-based: reproducible
    o3, _, _, z3 = ds.train_batch(B, S, 2.0, 6.0, bg, seed=7, counter=1)
    assert not torch.equal(z, z3) and not torch.equal(o, o3)               # a new counter is a new batch
    # data-parallel shards: three ranks with the same seed / counter and first_ray = their offset draw, between them,
    # exactly the batch one rank draws with the global size
    parts = [ds.train_batch(n, S, 2.0, 6.0, bg, seed=7, counter=0, first_ray=f) for f, n in ((0, 25000), (25000, 20000), (45000, 15000))]
    for k, whole in enumerate((o, d, target, z)):
        assert torch.equal(torch.cat([p[k] for p in parts], 0), whole), k
    # depths: inside their stratum, and u = (z - lower) / (upper - lower) is uniform
    plain = O.stratified_depths(2.0, 6.0, S, 1, False)[0]
    mids = 0.5 * (plain[1:] + plain[:-1])
    lower, upper = torch.cat([plain[:1], mids]).cuda(), torch.cat([mids, plain[-1:]]).cuda()
    assert bool(((z >= lower) & (z <= upper)).all())
    u = ((z - lower) / (upper - lower)).flatten().double()
    assert abs(float(u.mean()) - 0.5) < 2e-3 and abs(float(u.var()) - 1 / 12) < 2e-3
    assert abs(float(torch.corrcoef(torch.stack([u[:-1], u[1:]]))[0, 1])) < 5e-3
    # rays: recover the pixel from the origin (frame) and the target from the frames; uniform over frames/rows/cols
    frame = (o[:, None, :] - ds.poses[None, :, :3, 3]).abs().sum(-1).argmin(1)
    counts = torch.bincount(frame, minlength=len(ds)).double()
    chi2 = float(((counts - B / len(ds)) ** 2 / (B / len(ds))).sum())
    assert chi2 < 30.0, (chi2, counts)                                      # 5 degrees of freedom
    np.testing.assert_allclose(d.norm(dim=-1).cpu().numpy(), 1.0, atol=1e-6)
    # every (ray, target) pair must exist in the frames: compare against the dense per-frame table
    for f in range(len(ds)):
        sel = torch.nonzero(frame == f).flatten()[:400]
        oo, dd, tgt = ds.get_image_rays(f, "cuda")
        dots = d[sel] @ dd.reshape(-1, 3).T
        pix = dots.argmax(1)
        assert float((dots.max(1).values - 1).abs().max()) < 1e-6
        a = ds.images[f].reshape(-1, 4)[pix]
        want = a[:, :3] * a[:, 3:4] + bg * (1 - a[:, 3:4])
        assert torch.equal(target[sel], want)                               # same roundings as run.py:317-322
        rows = pix // ds.W
        assert 0 <= int(rows.min()) and int(rows.max()) < ds.H


# ------------------------------------------------------------------ operator surface
def test_nerf_decoder_forward_on_encoded_inputs(field):
    """BaseDecoder.forward(x_enc, d_enc) of reference src/decoders.py:68-87: the separable chain
    decoder(representation(x), dir_representation(d)) -- what reference core.py:357-359 evaluates -- against the
    fused entry and the reference's golden, values and parameter gradients."""
    import copy
    g = golden("g4_decoder")
    model = copy.deepcopy(field)
    pts, dirs = T(g["pts"]).cuda(), T(g["dirs"]).cuda()
    with torch.no_grad():
        x_enc, d_enc = model.representation(pts), model.dir_representation(dirs)
        assert x_enc.shape == (512, 63) and d_enc.shape == (512, 27)
        rgb, sigma = model.decoder(x_enc, d_enc)
        rgb_f, sigma_f = model(pts, dirs)
    assert rgb.shape == (512, 3) and sigma.shape == (512, 1)
    np.testing.assert_allclose(rgb.cpu().numpy(), g["rgb"], atol=2e-2)
    np.testing.assert_allclose(rgb.cpu().numpy(), rgb_f.cpu().numpy(), atol=6e-3)     # sinf codes vs in-register v_sin codes
    np.testing.assert_allclose(sigma.cpu().numpy(), sigma_f.cpu().numpy(), atol=6e-3 * max(1.0, float(sigma_f.max())))
    w = torch.randn(512, 3, generator=torch.Generator().manual_seed(1)).cuda()
    grads = []
    for fused in (False, True):
        model.zero_grad()
        out = model(pts, dirs) if fused else model.decoder(model.representation(pts), model.dir_representation(dirs))
        ((out[0] * w).sum() + out[1].sum()).backward()
        grads.append(torch.cat([p.grad.reshape(-1) for p in model.decoder.parameters()]))
    rel = float((grads[0] - grads[1]).norm() / grads[1].norm())
    assert rel < 2e-2, rel


@pytest.mark.parametrize("L,L_dir,use_dirs", [(6, 2, True), (10, 0, True), (0, 4, True), (4, 4, False)])
def test_narrower_fourier_codes_from_yaml(L, L_dir, use_dirs):
    """L_embed / L_embed_dir / use_viewdirs / use_positional_encoding are YAML keys of the reference
    (src/core.py:36-55): any code of up to 10 / 4 bands runs on the compiled kernels (zero-padded weight
    columns), checked against the oracle with the same parameters."""
    from src.core import NeuralField
    cfg = {"mode": "part2_nerf", "L_embed": L, "L_embed_dir": L_dir, "use_viewdirs": use_dirs,
           "use_positional_encoding": L > 0, "hidden_dim": 256, "num_layers": 8, "skip_layer": 4, "view_dim": 128}
    torch.manual_seed(L * 10 + L_dir)
    model = NeuralField(cfg).cuda()
    l_dir = L_dir if use_dirs else 0
    assert model.decoder.pts_layers[0].weight.shape == (256, 3 + 6 * L)
    assert model.decoder.view_layer.weight.shape == (128, 256 + 3 + 6 * l_dir)
    params = {k[len("decoder."):]: v.detach().cpu() * (1.5 if k.endswith("weight") else 1.0)
              for k, v in model.state_dict().items() if k.startswith("decoder.")}
    model.load_state_dict({**model.state_dict(), **{"decoder." + k: v for k, v in params.items()}})
    gen = torch.Generator().manual_seed(3)
    pts = (torch.rand(300, 3, generator=gen) - 0.5) * 2.4
    dirs = torch.nn.functional.normalize(torch.randn(300, 3, generator=gen), dim=-1)
    rgb, sigma = model(pts.cuda(), dirs.cuda())
    ref_rgb, ref_sigma = O.nerf_field(params, pts, dirs, l_pos=L, l_dir=l_dir)
    np.testing.assert_allclose(rgb.detach().cpu().numpy(), ref_rgb.numpy(), atol=2e-2)
    np.testing.assert_allclose(sigma.detach().cpu().numpy(), ref_sigma.numpy(), atol=3e-2 * max(1.0, float(ref_sigma.max())))
    # gradients reach every (unpadded) parameter with the reference's shapes
    (rgb.sum() + sigma.sum()).backward()
    for name, p in model.decoder.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape and bool(torch.isfinite(p.grad).all()), name
    ps = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    r2, s2 = O.nerf_field(ps, pts, dirs, l_pos=L, l_dir=l_dir)
    (r2.sum() + s2.sum()).backward()
    for name in ("pts_layers.0.weight", "pts_layers.4.weight", "view_layer.weight", "rgb_layer.weight"):
        got, want = dict(model.decoder.named_parameters())[name].grad.cpu(), ps[name].grad
        cos = float((got * want).sum() / (got.norm() * want.norm() + 1e-20))
        assert cos > 0.97, (name, cos)

    assert model.decoder.fused and not NeuralField({**cfg, "hidden_dim": 128}).decoder.fused     # library-GEMM path, next test


def test_decoder_shapes_other_than_the_compiled_one_run_as_library_gemms(tmp_path):
    """hidden_dim / num_layers / skip_layer / view_dim / L_embed from the YAML (reference src/core.py:36-55): any value
    builds and trains.  Shapes the chain kernels are not compiled for run the same layers as library GEMMs on the
    GPU around the HIP Fourier codes, against the oracle's fp32 restatement of src/decoders.py:68-87."""
    from src.core import NeuralField
    from src.renderer import render_image, render_rays
    cfg = {"mode": "part2_nerf", "L_embed": 12, "L_embed_dir": 5, "hidden_dim": 64, "num_layers": 5, "skip_layer": 2, "view_dim": 32}
    torch.manual_seed(3)
    model = NeuralField(cfg).cuda()
    assert not model.decoder.fused and NeuralField({"mode": "part2_nerf", "L_embed": 10}).decoder.fused
    keys = set(model.state_dict())
    assert {"decoder.pts_layers.2.weight", "decoder.sigma_layer.bias", "decoder.view_layer.weight", "decoder.rgb_layer.weight"} <= keys
    assert model.state_dict()["decoder.pts_layers.2.weight"].shape == (64, 64 + 75)       # skip layer: [h | x_enc]
    g = torch.Generator().manual_seed(1)
    pts, dirs = (torch.rand(500, 3, generator=g) - 0.5) * 3, torch.nn.functional.normalize(torch.randn(500, 3, generator=g), dim=-1)
    rgb, sigma = model(pts.cuda(), dirs.cuda())
    assert rgb.shape == (500, 3) and sigma.shape == (500, 1)
    sd = {k[len("decoder."):]: v.detach().cpu() for k, v in model.state_dict().items() if k.startswith("decoder.")}
    x_enc, d_enc = O.fourier_encode(pts, 12), O.fourier_encode(dirs, 5)
    h = x_enc
    for i in range(5):
        if i == 2:
            h = torch.cat([h, x_enc], -1)
        h = torch.relu(torch.nn.functional.linear(h, sd[f"pts_layers.{i}.weight"], sd[f"pts_layers.{i}.bias"]))
    ref_sigma = torch.relu(torch.nn.functional.linear(h, sd["sigma_layer.weight"], sd["sigma_layer.bias"]))
    feat = torch.nn.functional.linear(h, sd["feature_layer.weight"], sd["feature_layer.bias"])
    hv = torch.relu(torch.nn.functional.linear(torch.cat([feat, d_enc], -1), sd["view_layer.weight"], sd["view_layer.bias"]))
    ref_rgb = torch.sigmoid(torch.nn.functional.linear(hv, sd["rgb_layer.weight"], sd["rgb_layer.bias"]))
    np.testing.assert_allclose(rgb.detach().cpu().numpy(), ref_rgb.numpy(), atol=2e-4)         # sin/cos of 2^11 x in fp32
    np.testing.assert_allclose(sigma.detach().cpu().numpy(), ref_sigma.numpy(), atol=2e-4)
    # the operators on their own, the ray-mode renderer, gradients to every layer, the image loop
    rgb2, _ = model.decoder(model.representation(pts.cuda()), model.dir_representation(dirs.cuda()))
    assert torch.allclose(rgb2, rgb)
    gen = golden("g6_render")
    o, d = T(gen["rays_o"]).cuda(), T(gen["rays_d"]).cuda()
    c, _, _ = render_rays(model, o, d, 2.0, 6.0, 32, True)
    c.square().mean().backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in model.parameters())
    with torch.no_grad():
        img = render_image(model, o.view(8, 12, 3), d.view(8, 12, 3), 2.0, 6.0, 32, 40, True)
    assert img.shape == (8, 12, 3) and bool(torch.isfinite(img).all())
    with pytest.raises(ValueError):
        NeuralField({"mode": "part2_nerf", "L_embed": 10, "num_layers": 4, "skip_layer": 9})


def test_render_rays_fwd_launch_chain_equals_chunked_kernels(field):
    """nerf_render_rays_fwd (one launch chain, one workspace) == the three kernels called chunk by chunk, for a
    ragged last chunk, a per-ray background and 300 samples per ray (beyond the former 256-sample limit)."""
    import project_nerf_amd  # noqa: F401
    from project_nerf_amd import ops
    g = golden("g6_render")
    o, d = T(g["rays_o"]).cuda(), T(g["rays_d"]).cuda()
    packed = field.decoder.packed_weights()
    bg = torch.rand(96, 3, generator=torch.Generator().manual_seed(1)).cuda()
    for S, chunk in ((64, 40), (300, 96)):
        c, dep, acc = ops.render_rays_fwd(packed, o, d, S, 2.0, 6.0, bg, chunk)
        z = ops.sample_rays(o, d, 2.0, 6.0, S)
        rgb, sigma = ops.mlp_fwd(packed, o, d, z)
        if S <= 256:
            c2, dep2, acc2, _, _ = ops.composite_fwd(rgb.view(96, S, 3), sigma.view(96, S), z, d, bg)
            assert torch.equal(c, c2) and torch.equal(dep, dep2) and torch.equal(acc, acc2)
        rc, rd, ra = O.composite(rgb.view(96, S, 3).cpu(), sigma.view(96, S).cpu(), z.cpu(), d.cpu(), bg.cpu())
        np.testing.assert_allclose(c.cpu().numpy(), rc.numpy(), rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(dep.cpu().numpy(), rd.numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(ops.render_rays_fwd(packed, o, d, 64, 2.0, 6.0, torch.ones(3).cuda(), 17)[0].cpu().numpy(),
                               g["rgb_plain"], atol=1e-2)


@pytest.mark.parametrize("S", [257, 384, 700, 1024])
def test_composite_beyond_256_samples(S):
    import project_nerf_amd  # noqa: F401
    from project_nerf_amd import ops
    gen = torch.Generator().manual_seed(S)
    R = 9
    z = torch.sort(torch.rand(R, S, generator=gen) * 4 + 2, dim=-1).values
    sig = torch.rand(R, S, generator=gen) * 2
    rgb = torch.rand(R, S, 3, generator=gen)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen), dim=-1)
    r1, s1 = rgb.cuda().requires_grad_(True), sig.cuda().requires_grad_(True)
    c, dep, acc, _ = ops.composite(r1, s1, z.cuda(), d.cuda(), torch.ones(3).cuda())
    r2, s2 = rgb.clone().requires_grad_(True), sig.clone().requires_grad_(True)
    rc, rd, ra = O.composite(r2, s2, z, d, torch.ones(3))
    np.testing.assert_allclose(c.detach().cpu().numpy(), rc.detach().numpy(), rtol=3e-5, atol=3e-6)
    w = torch.randn(R, 3, generator=gen)
    (c * w.cuda()).sum().backward()
    (rc * w).sum().backward()
    ref = r2.grad.numpy()                       # weights deep in the ray are ~1e-7 of the row's largest: compare per row
    assert np.max(np.abs(r1.grad.cpu().numpy() - ref) / (np.abs(ref).max(axis=(1, 2), keepdims=True) + 1e-12)) < 1e-4
    ref = s2.grad.numpy()
    assert np.max(np.abs(s1.grad.cpu().numpy() - ref) / (np.abs(ref).max(axis=1, keepdims=True) + 1e-12)) < 1e-4
