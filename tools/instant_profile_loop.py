"""Steady-state Instant-NGP engine steps for the profilers (rocprofv3 --stats / --pmc), apart from the training that
leads there -- so that a kernel's average in the profile IS its steady-state time (the first 256 steps of a run work on
2 M samples per batch, the pruned steady state on ~200 k: averaged together they describe neither).

    python tools/instant_profile_loop.py train /tmp/instant_state.pt [iters]   # bench.py's Instant run: 800 x 800 synthetic
                                                                                # frames, 16384 rays x 128, grid updates; saves
                                                                                # the trained engine + eight drawn batches
    python tools/instant_profile_loop.py loop  /tmp/instant_state.pt [steps]   # ONLY steady-state steps on the saved state:
                                                                                # what runs under the profiler
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import yaml  # noqa: E402

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
BATCH, S = 16384, 128


def engine(iters):
    from project_nerf_amd.engine import InstantNgpEngine
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "part2_instant.yaml.example")))
    cfg["train_iters"] = iters
    return InstantNgpEngine(cfg, device="cuda", seed=0)


def train(path, iters):
    from src.dataset import BlenderDataset, SYNTHETIC_CAMERA_ANGLE, synthetic_frames
    images, poses = synthetic_frames(30, 800, "cuda", n_samples=192)
    ds = BlenderDataset.from_tensors(images, poses, SYNTHETIC_CAMERA_ANGLE)
    torch.manual_seed(0)
    eng = engine(iters)
    active = 1.0
    for it in range(1, iters + 1):
        o, d, target = ds.sample_batch(BATCH, eng.bg)
        eng.train_step(o, d, target, S)
        interval = 32 if it < iters * 0.1 else (128 if it < iters * 0.5 else 512)      # run.py:636-641
        if it < iters * 0.9 and it >= 256 and it % interval == 0:
            active = eng.update_grid()
    batches = [tuple(t.cpu() for t in ds.sample_batch(BATCH, eng.bg)) for _ in range(8)]
    torch.save({"table": eng.table.cpu(), "net": eng.net.cpu(), "state": {k: tuple(t.cpu() for t in v) for k, v in eng.state.items()},
                "grid": eng.grid.cpu(), "binary_grid": eng.binary_grid.cpu(), "step_count": eng.step_count, "iters": iters,
                "batches": batches, "active": active}, path)
    print(f"trained {iters} steps, active ratio {active:.4f}; state -> {path}")


def loop(path, steps):
    from project_nerf_amd import ops
    st = torch.load(path)
    eng = engine(st["iters"])
    with torch.no_grad():
        eng.table.copy_(st["table"])
        eng.net.copy_(st["net"])
        for k, (m, v) in st["state"].items():
            eng.state[k][0].copy_(m)
            eng.state[k][1].copy_(v)
        eng.grid.copy_(st["grid"])
        eng.binary_grid = st["binary_grid"].cuda()
    eng.step_count = st["step_count"] - steps - 8          # the cosine schedule stays inside its range
    ops.imlp_pack(eng.net, eng.packed)
    batches = [tuple(t.cuda() for t in b) for b in st["batches"]]
    ahead = []

    def draw(i):
        o, d, target = batches[i % len(batches)]
        return o, d, target, eng.prepare_batch(o, d, S)
    counts = []
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for i in range(steps + 4):
        if i == 4:
            ev[0].record()                                   # four untimed steps first (allocator, workspaces)
        if not ahead:
            ahead.append(draw(i))
        o, d, target, prepared = ahead.pop()
        ahead.append(draw(i + 1))
        counts.append(prepared.get()[2].shape[0])
        eng.train_step(o, d, target, S, prepared=prepared)
    ev[1].record()
    torch.cuda.synchronize()
    print(f"done {steps} steady-state steps: {ev[0].elapsed_time(ev[1]) / steps:.4f} ms/step, "
          f"{sum(counts) / len(counts):.0f} active samples per batch of {BATCH * S}")


if __name__ == "__main__":
    mode, path = sys.argv[1], sys.argv[2]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else (1000 if mode == "train" else 96)
    (train if mode == "train" else loop)(path, n)
