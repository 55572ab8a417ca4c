"""Steady-state Instant-NGP engine step (one batch ahead, as bench.py runs it): step time and its kernels by GPU time (torch profiler).
    python tools/time_instant.py"""
import collections
import os
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
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step()
torch.cuda.synchronize()
print(f"instant engine step: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms ({batch} rays x {S} samples)")
n_prof = 20
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
    for _ in range(n_prof):
        step()
    torch.cuda.synchronize()
tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA:
        tot[ev.name] += ev.device_time
        cnt[ev.name] += 1
print(f"GPU time per step (sum of kernels): {sum(tot.values()) / n_prof / 1e3:.3f} ms in {sum(cnt.values()) // n_prof} launches")
for name, us in sorted(tot.items(), key=lambda kv: -kv[1])[:32]:
    print(f"   {us / n_prof / 1e3:.4f} ms  x{cnt[name] / n_prof:.1f}   {name[:120]}")
