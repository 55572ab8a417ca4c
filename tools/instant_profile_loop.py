import os
import sys, time, tempfile, numpy as np, torch, yaml
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from src.core import NeuralField
from src.dataset import BlenderDataset, write_synthetic_scene
from src.renderer import DensityGrid, render_rays
cfg = yaml.safe_load(open("configs/part2_instant.yaml.example"))
root = write_synthetic_scene(tempfile.mkdtemp() + "/scene", n_train=20, n_test=1, size=100)
ds = BlenderDataset(root, "train", 1, True, 1.0).to("cuda")
torch.manual_seed(0)
model = NeuralField(cfg).cuda(); grid = DensityGrid(128, 1.5, 0.12).cuda()
opt = torch.optim.AdamW(model.parameters(), lr=1e-2, weight_decay=1e-5)
bg = torch.ones(3, device="cuda")
def step(batch=16384, S=128):
    o, d, rgba = ds.sample_random_rays(batch, "cuda")
    target = rgba[:, :3] * rgba[:, 3:4] + (1 - rgba[:, 3:4])
    pred, _, _ = render_rays(model, o, d, 2.0, 6.0, S, True, density_grid=grid, bg_color=bg)
    loss = torch.nn.functional.mse_loss(pred, target)
    p = model.representation.encoding.params
    loss = loss + torch.mean(torch.abs(p[1:] - p[:-1])) * 1e-6
    opt.zero_grad(); loss.backward()
    torch.nn.utils.clip_grad_norm_(model.representation.parameters(), 1.0)
    torch.nn.utils.clip_grad_norm_(model.decoder.parameters(), 1.0)
    opt.step()
for s in range(1, 401):
    step()
    if s in (256, 320, 384):
        model.eval(); r = grid.update(model, device="cuda"); model.train()
torch.cuda.synchronize(); t0 = time.time()
for s in range(50): step()
torch.cuda.synchronize(); print("ms/step", (time.time() - t0) / 50 * 1e3, "active", r)
