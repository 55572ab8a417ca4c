"""Host side of the Instant-NGP engine step (one batch ahead, as bench.py runs it): time to queue a step against GPU time, cProfile.
    python tools/instant_host_profile.py [n_steps]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import yaml  # noqa: E402

from src.dataset import BlenderDataset, SYNTHETIC_CAMERA_ANGLE, synthetic_frames  # noqa: E402
from project_nerf_amd.engine import InstantNgpEngine  # noqa: E402

dev = torch.device("cuda")
cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "configs", "part2_instant.yaml.example")))
images, poses = synthetic_frames(12, 400, dev, n_samples=128)
ds = BlenderDataset.from_tensors(images, poses, SYNTHETIC_CAMERA_ANGLE)
torch.manual_seed(0)
eng = InstantNgpEngine(cfg, device="cuda", seed=0)
batch, S = 16384, 128
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


for it in range(1, 601):
    step()
    if it >= 256 and it % 64 == 0:
        eng.update_grid()
        ahead.clear()
n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n_steps):
    step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"{n_steps} steps: host queued them in {t_host / n_steps * 1e3:.3f} ms per step, GPU done after {t_all / n_steps * 1e3:.3f} ms per step")
pr = cProfile.Profile()
pr.enable()
for _ in range(n_steps):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
