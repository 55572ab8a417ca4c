"""PSNR-vs-steps of the vanilla engine on the synthetic scene (development aid).  Default: the asm-stream kernels with bf16
training images (the bench headline); `fp8`: the same kernels with 8-bit images (option stash_fp8); `legacy`: the
compiler-scheduled kernels with bf16 images."""
import os
import sys, numpy as np, torch, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from src.dataset import BlenderDataset, write_synthetic_scene
from project_nerf_amd.engine import VanillaNerfEngine
root = write_synthetic_scene(tempfile.mkdtemp() + "/scene", n_train=20, n_test=2, size=100)
ds = BlenderDataset(root, "train", 1, True, 1.0).to("cuda")
test = BlenderDataset(root, "test", 1, True, 1.0)
from project_nerf_amd import _lib
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
if mode == "legacy":
    _lib.set_option("chain_legacy", 1)
if mode == "fp8":
    _lib.set_option("stash_fp8", 1)
print("training images:", {"legacy": "bf16 (compiler-scheduled kernels)", "fp8": "8-bit (asm-stream kernels)"}.get(mode, "bf16 (asm-stream kernels)"))
eng = VanillaNerfEngine(seed=0, lr=5e-4)
torch.manual_seed(0)
o_t, d_t, tgt = test.get_image_rays(0, "cuda")
t0 = time.time()
for step in range(1, 4001):
    o, d, rgba = ds.sample_random_rays(4096, "cuda")
    target = rgba[:, :3] * rgba[:, 3:4] + (1 - rgba[:, 3:4])
    loss = eng.train_step(o, d, target, 64)
    if step in (1, 100, 300, 1000, 2000, 3000, 4000):
        img = eng.render_image(o_t, d_t, 64, chunk=4096)
        psnr = -10 * np.log10(float(((img - tgt) ** 2).mean()))
        print(step, f"loss {loss.item():.5f} test psnr {psnr:.2f} dB  t={time.time()-t0:.1f}s", flush=True)
