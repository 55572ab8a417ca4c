"""Repeat a short vanilla training run and report loss / gradient health per step (development aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, tempfile
from src.dataset import BlenderDataset, write_synthetic_scene
from project_nerf_amd.engine import VanillaNerfEngine
from project_nerf_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] == "bf16":
    _lib.set_option("chain_legacy", 1)
root = write_synthetic_scene(tempfile.mkdtemp() + "/scene", n_train=16, n_test=2, size=64)
ds = BlenderDataset(root, "train", 1, True, 1.0).to("cuda")
for rep in range(int(os.environ.get("REPS", 4))):
    eng = VanillaNerfEngine(seed=0, lr=5e-4)
    torch.manual_seed(0)
    bad = None
    hist = []
    for step in range(1, int(os.environ.get('STEPS', 600)) + 1):
        o, d, rgba = ds.sample_random_rays(4096, "cuda")
        loss = eng.train_step(o, d, rgba[:, :3] * rgba[:, 3:4] + (1 - rgba[:, 3:4]), 64)
        if step <= 2 or step % 100 == 0:
            g = eng.grads
            slot = (eng.step_count - 1) % eng._scalars.shape[1]
            fin = bool(torch.isfinite(g).all())
            hist.append((step, float(loss), float(g.norm()), float(eng._scalars[1, slot]), fin))
            if not fin and bad is None:
                bad = step
    print(f"rep {rep}: first non-finite grad at {bad}; final loss {float(loss):.5f}")
    for h in hist:
        print("   step %4d loss %.5f |g| %.4e amax %.3e finite %s" % h)
