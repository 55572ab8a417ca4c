"""Step time of the DROP-IN path -- NeuralField + render_rays + torch.optim.Adam, as run.py trains -- next to the flat-parameter
engine bench.py times (development aid)."""
import os, sys, time, tempfile, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from src.core import NeuralField
from src.dataset import BlenderDataset, write_synthetic_scene
from src.renderer import render_rays
from project_nerf_amd.engine import VanillaNerfEngine
root = write_synthetic_scene(tempfile.mkdtemp() + "/scene", n_train=8, n_test=1, size=64)
ds = BlenderDataset(root, "train", 1, True, 1.0).to("cuda")
bg = torch.ones(3, device="cuda")
model = NeuralField({"mode": "part2_nerf", "L_embed": 10, "L_embed_dir": 4}).cuda()
opt = torch.optim.Adam(model.parameters(), lr=5e-4)
def step():
    o, d, rgba = ds.sample_random_rays(4096, "cuda")
    target = rgba[:, :3] * rgba[:, 3:4] + bg * (1 - rgba[:, 3:4])
    pred, _, _ = render_rays(model, o, d, 2.0, 6.0, 64, True)
    loss = torch.nn.functional.mse_loss(pred, target)
    opt.zero_grad(); loss.backward(); opt.step()
for _ in range(20): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize(); print(f"drop-in path (NeuralField + torch.optim.Adam): {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms/step")
eng = VanillaNerfEngine(seed=0)
def estep():
    o, d, rgba = ds.sample_random_rays(4096, "cuda")
    eng.train_step(o, d, rgba[:, :3] * rgba[:, 3:4] + bg * (1 - rgba[:, 3:4]), 64)
for _ in range(20): estep()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): estep()
torch.cuda.synchronize(); print(f"engine (flat parameters, fused Adam + repack):   {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms/step")

# ---- Instant-NGP: module path (NeuralField + DensityGrid + torch.optim.AdamW + torch TV / clipping) against the engine
import yaml
from src.renderer import DensityGrid
from project_nerf_amd.engine import InstantNgpEngine
cfg = yaml.safe_load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "configs", "part2_instant.yaml.example")))
model = NeuralField(cfg).cuda()
grid = DensityGrid(128, 1.5, 0.12).cuda()
ax = torch.linspace(-1.5, 1.5, 128)
gx, gy, gz = torch.meshgrid(ax, ax, ax, indexing="ij")
grid.binary_grid = ((gx ** 2 + gy ** 2 + gz ** 2) < 0.75 ** 2).cuda()          # ~13 % of the volume active
opt = torch.optim.AdamW(model.parameters(), lr=1e-2, weight_decay=1e-5)
def istep():
    o, d, rgba = ds.sample_random_rays(16384, "cuda")
    target = rgba[:, :3] * rgba[:, 3:4] + bg * (1 - rgba[:, 3:4])
    pred, _, _ = render_rays(model, o, d, 2.0, 6.0, 128, True, density_grid=grid, bg_color=bg)
    p = model.representation.encoding.params
    loss = torch.nn.functional.mse_loss(pred, target) + torch.mean(torch.abs(p[1:] - p[:-1])) * 1e-6
    opt.zero_grad(); loss.backward()
    torch.nn.utils.clip_grad_norm_(model.representation.parameters(), max_norm=1.0)
    torch.nn.utils.clip_grad_norm_(model.decoder.parameters(), max_norm=1.0)
    opt.step()
for _ in range(10): istep()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): istep()
torch.cuda.synchronize(); print(f"Instant, module path (NeuralField + torch.optim.AdamW): {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms/step")
eng = InstantNgpEngine(cfg, seed=0)
eng.binary_grid.copy_(grid.binary_grid)
def iestep():
    o, d, target = ds.sample_batch(16384, eng.bg)
    eng.train_step(o, d, target, 128)
for _ in range(10): iestep()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): iestep()
torch.cuda.synchronize(); print(f"Instant, engine:                                        {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms/step")
