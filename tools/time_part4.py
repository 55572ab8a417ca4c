"""Step time of the Part 4 (dual-hash dynamic field) training step on part4.DualHashEngine at configs/part4.yaml.example
(8192 rays x 64 samples, 64^3 occupancy grid at ~12 % active, 28.5 M parameters): whole steps (with the every-16th-step
regulariser probes), the step's kernels by GPU time (torch profiler: no aten::mm / Cijk_* may appear), and the same loop
body on the module path (NeuralField + render_rays + torch.optim.AdamW) for comparison.
    python tools/time_part4.py [--module]"""
import os, sys, time
import torch, yaml
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from src.core import NeuralField
from project_nerf_amd import ops
from project_nerf_amd.dynamic import part4_probe_draws
from project_nerf_amd.part4 import DualHashEngine

dev = "cuda"
cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "configs", "part4.yaml.example")))
torch.manual_seed(0)
model = NeuralField(cfg).to(dev)
R, S = cfg["batch_size"], cfg["n_samples"]
o = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1) * 4.03
d = torch.nn.functional.normalize(-o + 0.3 * torch.randn(R, 3, device=dev), dim=-1)
t = torch.rand(R, 1, device=dev)
target = torch.rand(R, 3, device=dev)
eng = DualHashEngine(cfg, device=dev, seed=0)
eng.load_from_model(model)
eng.binary_grid = torch.rand_like(eng.grid) < 0.12
step_no = [300]
ahead = []


def step():
    # the product loop's order (dynamic.py): the next batch's compaction is queued ahead of this step's kernels
    step_no[0] += 1
    if not ahead:
        ahead.append(eng.prepare_batch(o, d, S))
    prepared = ahead.pop()
    ahead.append(eng.prepare_batch(o, d, S))
    return eng.train_step(o, d, target, t, S, prepared=prepared, probes=part4_probe_draws(cfg, step_no[0], dev))


for _ in range(20):
    step()
torch.cuda.synchronize()
n_steps = 160
t0 = time.perf_counter()
for _ in range(n_steps):
    loss = step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n_steps * 1e3
n_param = eng.tables.numel() + eng.net.numel()
z, slots, pts, dirs = ops.sample_compact(o, d, eng.near, eng.far, S, eng.binary_grid, eng.bound)
print(f"part4 engine step: {ms:.3f} ms = {R / ms * 1e-3:.2f} M rays/s  ({R} rays x {S} samples, {pts.shape[0]} active samples, "
      f"{n_param / 1e6:.1f} M parameters, probes every 16th step; loss {float(loss):.4f})", flush=True)
from torch.profiler import profile, ProfilerActivity
step_no[0] = 320          # 5 profiled steps without a probe step, then one with
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        step()
    torch.cuda.synchronize()
ev = prof.key_averages()
tot = sum(e.device_time_total for e in ev) / 5
print(f"GPU time per step (sum of kernels, no probe step): {tot / 1e3:.3f} ms in {sum(e.count for e in ev) // 5} launches")
for e in sorted(ev, key=lambda e: -e.device_time_total)[:24]:
    print(f"  {e.device_time_total / 5 / 1e3:7.4f} ms  x{e.count / 5:<5.1f} {e.key[:120]}")
bad = [e.key for e in ev if "Cijk" in e.key or "aten::mm" in e.key or "gemm" in e.key.lower()]
print("library GEMMs in the step:", bad or "none")
if "--module" in sys.argv:
    from src.renderer import DensityGrid, render_rays
    from project_nerf_amd.dynamic import part4_param_groups, part4_regularisers
    grid = DensityGrid(cfg["grid_resolution"], cfg["scene_bound"], cfg["grid_threshold"]).to(dev)
    grid.binary_grid = eng.binary_grid
    opt = torch.optim.AdamW(part4_param_groups(model, cfg["learning_rate"]), weight_decay=cfg["weight_decay"])
    bg = torch.ones(3, device=dev)

    def mstep():
        step_no[0] += 1
        pred, _, _, extras = render_rays(model, o, d, cfg["near"], cfg["far"], S, True, density_grid=grid, times=t, bg_color=bg)
        loss = torch.nn.functional.mse_loss(pred, target) + sum(part4_regularisers(model, cfg, step_no[0], extras["mean_delta_x"]).values())
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
        opt.step()
    for _ in range(5):
        mstep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        mstep()
    torch.cuda.synchronize()
    print(f"module path (fused operator under torch autograd + torch.optim): {(time.perf_counter() - t0) / 30 * 1e3:.2f} ms per step")
