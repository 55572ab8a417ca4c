"""Host side of the Part 4 engine step: cProfile over steady-state steps (top cumulative entries), wall time against GPU time.
    python tools/part4_host_profile.py [n_steps]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import yaml  # noqa: E402

from src.core import NeuralField  # noqa: E402
from project_nerf_amd.part4 import DualHashEngine  # noqa: E402

dev = "cuda"
cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "configs", "part4.yaml.example")))
torch.manual_seed(0)
R, S = cfg["batch_size"], cfg["n_samples"]
o = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1) * 4.03
d = torch.nn.functional.normalize(-o + 0.3 * torch.randn(R, 3, device=dev), dim=-1)
t, target = torch.rand(R, 1, device=dev), torch.rand(R, 3, device=dev)
eng = DualHashEngine(cfg, device=dev, seed=0)
eng.load_from_model(NeuralField(cfg).to(dev))
eng.binary_grid = torch.rand_like(eng.grid) < 0.12
n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300


def loop(n, first=301):
    for step in range(first, first + n):
        if step % 16 == 0:
            continue                       # probe steps aside: the steady-state data step
        eng.train_step(o, d, target, t, S, probes=None)


loop(50)
torch.cuda.synchronize()
t0 = time.perf_counter()
loop(n_steps)
t_host = time.perf_counter() - t0          # the host has queued everything
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
steps = sum(1 for s in range(301, 301 + n_steps) if s % 16)
print(f"{steps} steps: host queued them in {t_host / steps * 1e3:.3f} ms per step, GPU done after {t_all / steps * 1e3:.3f} ms per step")
pr = cProfile.Profile()
pr.enable()
loop(n_steps)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
