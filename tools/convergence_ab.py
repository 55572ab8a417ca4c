"""8-bit training images (default asm-stream kernels) against bf16 images (compiler-scheduled kernels): test PSNR after
3000 steps on the synthetic scene, mean over the two test views, several seeds each (development aid).  Runs that fall
into the dead-density plateau (loss stuck > 0.1 at step 150: a property of the reference's bare-ReLU density head, see
DESIGN.md section 2) are reported and left out of the means."""
import os, sys, tempfile, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from src.dataset import BlenderDataset, write_synthetic_scene
from project_nerf_amd import _lib
from project_nerf_amd.engine import VanillaNerfEngine
root = write_synthetic_scene(tempfile.mkdtemp() + "/scene", n_train=20, n_test=2, size=100)
ds = BlenderDataset(root, "train", 1, True, 1.0).to("cuda")
test = BlenderDataset(root, "test", 1, True, 1.0)
views = [test.get_image_rays(i, "cuda") for i in range(len(test))]
def run(seed, legacy):
    _lib.set_option("chain_legacy", 1 if legacy else 0)
    eng = VanillaNerfEngine(seed=seed, lr=5e-4)
    torch.manual_seed(seed)
    for step in range(1, 3001):
        o, d, rgba = ds.sample_random_rays(4096, "cuda")
        loss = eng.train_step(o, d, rgba[:, :3] * rgba[:, 3:4] + (1 - rgba[:, 3:4]), 64)
        if step == 150 and float(loss) > 0.1:
            return None
    ps = []
    for o_t, d_t, tgt in views:
        img = eng.render_image(o_t, d_t, 64, chunk=4096)
        ps.append(-10 * np.log10(float(((img.clamp(0, 1) - tgt) ** 2).mean())))
    return float(np.mean(ps))
for name, legacy in (("8-bit images (asm stream, K = 64 wgrad)", False), ("bf16 images (compiler-scheduled)", True)):
    res = [run(seed, legacy) for seed in range(6)]
    ok = [r for r in res if r is not None]
    print(f"{name}: " + " ".join("dead" if r is None else f"{r:.2f}" for r in res) + f"  -> mean {np.mean(ok):.2f} dB over {len(ok)} runs", flush=True)
_lib.set_option("chain_legacy", 0)
