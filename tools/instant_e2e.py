"""Instant-NGP path end to end on the synthetic scene: wall time to PSNR, train rays/s, 800x800 FPS."""
import os
import sys, time, tempfile, numpy as np, torch, yaml
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from src.core import NeuralField
from src.dataset import BlenderDataset, write_synthetic_scene, look_at_pose
from src.renderer import DensityGrid, render_rays
cfg = yaml.safe_load(open("configs/part2_instant.yaml.example"))
root = write_synthetic_scene(tempfile.mkdtemp() + "/scene", n_train=40, n_test=4, size=200)
ds = BlenderDataset(root, "train", 1, True, 1.0).to("cuda")
test = BlenderDataset(root, "test", 1, True, 1.0)
torch.manual_seed(0)
model = NeuralField(cfg).cuda()
grid = DensityGrid(128, 1.5, 0.12).cuda()
iters, batch, S = 1500, 16384, 128
opt = torch.optim.AdamW(model.parameters(), lr=1e-2, weight_decay=1e-5)
sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=iters, eta_min=1e-4)
bg = torch.ones(3, device="cuda")
def evaluate():
    model.eval(); ps = []
    with torch.no_grad():
        for i in range(len(test)):
            o, d, tgt = test.get_image_rays(i, "cuda")
            img = render_rays(model, o.reshape(-1, 3), d.reshape(-1, 3), 2.0, 6.0, S, False, density_grid=grid)[0]
            ps.append(-10 * np.log10(float(((img.clamp(0, 1) - tgt.reshape(-1, 3)) ** 2).mean())))
    model.train(); return float(np.mean(ps))
torch.cuda.synchronize(); t0 = time.time(); active = 1.0
for step in range(1, iters + 1):
    o, d, rgba = ds.sample_random_rays(batch, "cuda")
    target = rgba[:, :3] * rgba[:, 3:4] + (1 - rgba[:, 3:4])
    pred, _, _ = render_rays(model, o, d, 2.0, 6.0, S, True, density_grid=grid, bg_color=bg)
    loss = torch.nn.functional.mse_loss(pred, target)
    p = model.representation.encoding.params
    loss = loss + torch.mean(torch.abs(p[1:] - p[:-1])) * 1e-6
    opt.zero_grad(); loss.backward()
    torch.nn.utils.clip_grad_norm_(model.representation.parameters(), 1.0)
    torch.nn.utils.clip_grad_norm_(model.decoder.parameters(), 1.0)
    opt.step(); sched.step()
    interval = 32 if step < iters * 0.1 else (128 if step < iters * 0.5 else 512)
    if step < iters * 0.9 and grid.should_update(step, interval, 256):
        model.eval(); active = grid.update(model, device="cuda"); model.train()
    if step in (100, 300, 600, 1000, 1500):
        torch.cuda.synchronize(); el = time.time() - t0
        print(f"step {step}: {el:.1f}s  {step*batch/el/1e6:.2f} Mrays/s  loss {loss.item():.5f}  active {active:.3f}  test psnr {evaluate():.2f} dB", flush=True)
        t0 += 0  # evaluation time is included in wall; fine for an upper bound
# 800x800 render FPS
H = W = 800
focal = 0.5 * W / np.tan(0.5 * 0.6911112070083618)
c2w = torch.tensor(look_at_pose(4.0311 * np.array([0.6, 0.5, 0.62])), dtype=torch.float32)
j, i = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
dd = torch.stack([(i - W * .5) / focal, -(j - H * .5) / focal, -torch.ones_like(i)], -1).reshape(-1, 3).float() @ c2w[:3, :3].T
dd = (dd / dd.norm(dim=-1, keepdim=True)).cuda(); oo = c2w[:3, 3].expand_as(dd).contiguous().cuda()
model.eval()
with torch.no_grad():
    for chunk in (200000, 640000):
        def frame():
            return torch.cat([render_rays(model, oo[k:k + chunk], dd[k:k + chunk], 2.0, 6.0, S, False, density_grid=grid)[0] for k in range(0, H * W, chunk)])
        frame(); torch.cuda.synchronize(); t0 = time.time()
        for _ in range(5): frame()
        torch.cuda.synchronize(); dt = (time.time() - t0) / 5
        print(f"render 800x800 S={S} chunk={chunk}: {dt*1e3:.1f} ms/frame = {1/dt:.1f} FPS", flush=True)
