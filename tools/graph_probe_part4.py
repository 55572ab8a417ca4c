"""How much of the Part 4 engine step is launch gaps?  ONE steady-state step (configs/part4.yaml.example sizes) captured in a
HIP graph with its host-side arguments baked in -- same batch, same active-sample count, same optimiser step number on every
replay: TIMING ONLY -- against the same step as ~75 stream launches.  An upper bound for what graph capture of the step could
buy: a real capture needs launch parameters that do not depend on the active-sample count (it changes every step).
    python tools/graph_probe_part4.py"""
import os
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


def step(prepared):
    return eng.train_step(o, d, target, t, S, prepared=prepared)


prep = eng.prepare_batch(o, d, S)
prep[0].get()
for _ in range(20):
    step(prep)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step(prep)
torch.cuda.synchronize()
print(f"stream launches: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms/step ({prep[0].get()[2].shape[0]} active samples)")
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step(prep)
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = step(prep)
torch.cuda.synchronize()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    g.replay()
torch.cuda.synchronize()
print(f"graph replay:    {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms/step   loss {float(loss):.5f}")
