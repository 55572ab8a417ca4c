"""Step time of the Part 4 (dual-hash dynamic field) training loop body on synthetic rays (development aid):
configs/part4.yaml.example (8192 rays x 64 samples, 64^3 occupancy grid at ~12 % active), forward through render_rays with
times, the loop's regularisers, backward, clip, AdamW -- the body of project-nerf_amd/dynamic.py::run_dynamic.  Prints
ms per step and the kernels that take the most GPU time (torch profiler)."""
import os, sys, time
import torch, yaml
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from src.core import NeuralField
from src.renderer import DensityGrid, render_rays
from project_nerf_amd.dynamic import part4_regularisers

dev = "cuda"
cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "configs", "part4.yaml.example")))
torch.manual_seed(0)
model = NeuralField(cfg).to(dev)
grid = DensityGrid(cfg["grid_resolution"], cfg["scene_bound"], cfg["grid_threshold"]).to(dev)
grid.binary_grid.copy_(torch.rand_like(grid.grid) < 0.12)
R, S = cfg["batch_size"], cfg["n_samples"]
o = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1) * 4.03
d = torch.nn.functional.normalize(-o + 0.3 * torch.randn(R, 3, device=dev), dim=-1)
t = torch.rand(R, 1, device=dev)
target = torch.rand(R, 3, device=dev)
bg = torch.ones(3, device=dev)
opt = torch.optim.AdamW(model.parameters(), lr=cfg["learning_rate"], weight_decay=cfg["weight_decay"])
step_no = [0]


def step():
    step_no[0] += 1
    pred, _, _, extras = render_rays(model, o, d, cfg["near"], cfg["far"], S, True, density_grid=grid, times=t, bg_color=bg)
    loss = torch.nn.functional.mse_loss(pred, target) + sum(part4_regularisers(model, cfg, step_no[0], extras["mean_delta_x"]).values())
    opt.zero_grad()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
    opt.step()
    return loss


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 30 * 1e3
n_param = sum(p.numel() for p in model.parameters())
print(f"part4 step: {ms:.2f} ms = {R / ms * 1e-3:.2f} M rays/s  ({R} rays x {S} samples, {n_param / 1e6:.1f} M parameters)", flush=True)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(5):
        step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:14]
tot = sum(e.device_time_total for e in prof.key_averages()) / 5
print(f"GPU time per step (sum of kernels): {tot / 1e3:.2f} ms")
for e in rows:
    print(f"  {e.device_time_total / 5 / 1e3:7.3f} ms  x{e.count // 5:<4d} {e.key[:110]}")
