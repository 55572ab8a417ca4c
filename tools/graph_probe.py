"""How much of the vanilla training step is launch gaps?  Captures ONE step (batch draw + forward + loss + backward + Adam + repack)
in a HIP graph with its host-side arguments baked in (same batch, same Adam step number on every replay: timing only) and replays
it (development aid)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import project_nerf_amd
from project_nerf_amd.engine import VanillaNerfEngine
from src.dataset import BlenderDataset, SYNTHETIC_CAMERA_ANGLE
from bench import hemisphere_poses
dev = torch.device("cuda")
ds = BlenderDataset.from_tensors(torch.rand(20, 800, 800, 4, device=dev), hemisphere_poses(20, 7).to(dev), SYNTHETIC_CAMERA_ANGLE)
eng = VanillaNerfEngine(seed=0)
R, S = 4096, 64
def step(c):
    o, d, target, z = ds.train_batch(R, S, eng.near, eng.far, eng.bg, seed=100, counter=c)
    return eng.train_step(o, d, target, S, z=z)
for c in range(20): step(c)
torch.cuda.synchronize()
t0 = time.perf_counter()
for c in range(200): step(c)
torch.cuda.synchronize()
print(f"stream launches: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms/step")
eng._grad_calls = 0            # slot 1 at capture: no conditional zero_ inside the graph
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for c in range(3): step(c)
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = step(7)
torch.cuda.synchronize()
for _ in range(20): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): g.replay()
torch.cuda.synchronize()
print(f"graph replay:    {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms/step   loss {float(loss):.5f}")
