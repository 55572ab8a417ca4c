"""Same-box A/B of the Instant-NGP step: precounted hash backward (default) against the separate count pass.
    python tools/ab_instant_precount.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, yaml
from src.dataset import BlenderDataset, SYNTHETIC_CAMERA_ANGLE, synthetic_frames
from project_nerf_amd.engine import InstantNgpEngine

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "part2_instant.yaml.example")))
images, poses = synthetic_frames(12, 400, "cuda", n_samples=128)
ds = BlenderDataset.from_tensors(images, poses, SYNTHETIC_CAMERA_ANGLE)
batch, S = 16384, 128
res = {}
for rep in range(2):
    for precount in (True, False):
        torch.manual_seed(0)
        eng = InstantNgpEngine(dict(cfg, train_iters=600, precount=precount), seed=0)
        ahead = []

        def draw():
            o, d, target = ds.sample_batch(batch, eng.bg)
            return o, d, target, eng.prepare_batch(o, d, S)

        def step():
            if not ahead:
                ahead.append(draw())
            o, d, target, prepared = ahead.pop()
            ahead.append(draw())
            return eng.train_step(o, d, target, S, prepared=prepared)
        for it in range(1, 501):
            step()
            if it >= 256 and it % 64 == 0:
                eng.update_grid()
                ahead.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            loss = step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 200 * 1e3
        res.setdefault(precount, []).append(ms)
        print(f"precount={precount}: {ms:.4f} ms per step, loss {float(loss):.5f}", flush=True)
print({k: min(v) for k, v in res.items()})
