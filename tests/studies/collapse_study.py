"""Same-seed collapse study: is the "dead density" plateau a property of the MODEL or of the
kernels' arithmetic?

The reference's density head is a bare ReLU (src/decoders.py:78).  On a mostly-white scene the first ~100 steps can
push every density below zero; from then on every gradient is exactly zero and the run never recovers.  This script
runs the SAME ten seeds -- same initial weights, same ray batches, same stratified jitter, same Adam -- through

  * fp32      : the oracle's fp32 field + compositing under torch autograd on the GPU (the reference's arithmetic),
  * hip-bf16  : VanillaNerfEngine, asm-stream kernels, bf16 training images (the default),
  * hip-fp8   : the same with the opt-in 8-bit training images (option stash_fp8),

and reports, per seed and trainer, whether the run is dead at step 150 (loss > 0.1: the fixture's criterion in
tests/test_gpu_trained_parity.py) and its loss at the last step.  Lives under tests/ because it calls the oracle.

    python tests/studies/collapse_study.py [n_seeds] [steps]      ->  profiles/r03_collapse_study.txt (stdout)
"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import project_nerf_amd  # noqa: E402,F401
from oracle import nerf_oracle as O  # noqa: E402
from project_nerf_amd import _lib  # noqa: E402
from project_nerf_amd.engine import VanillaNerfEngine, flatten_state_dict  # noqa: E402
from src.dataset import BlenderDataset, write_synthetic_scene  # noqa: E402

N_SEEDS = int(sys.argv[1]) if len(sys.argv) > 1 else 10
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 300
R, S = 4096, 64


def batches(ds, seed):
    """the same (rays, target, jitter) sequence for every trainer of a seed"""
    g = torch.Generator(device="cuda").manual_seed(1000 + seed)
    torch.manual_seed(seed)
    for _ in range(STEPS):
        o, d, rgba = ds.sample_random_rays(R, "cuda")
        yield o, d, rgba[:, :3] * rgba[:, 3:4] + (1 - rgba[:, 3:4]), torch.rand(R, S, device="cuda", generator=g)


def run_fp32(ds, seed):
    params = {k: v.cuda().requires_grad_(True) for k, v in O.nerf_init_params(seed=seed).items()}
    opt = torch.optim.Adam(list(params.values()), lr=5e-4)
    hist = []
    with torch.device("cuda"):
        for o, d, target, u in batches(ds, seed):
            pred, _, _ = O.render_rays(lambda p, v: O.nerf_field(params, p, v), o, d, 2.0, 6.0, S, True, u=u)
            loss = torch.nn.functional.mse_loss(pred, target)
            opt.zero_grad()
            loss.backward()
            opt.step()
            hist.append(loss.detach())
    return [float(h) for h in hist]


def run_hip(ds, seed, fp8):
    _lib.set_option("stash_fp8", 1 if fp8 else 0)
    try:
        init = flatten_state_dict({"decoder." + k: v for k, v in O.nerf_init_params(seed=seed).items()})
        eng = VanillaNerfEngine(params=init, lr=5e-4)
        hist = [eng.train_step(o, d, target, S, u=u) for o, d, target, u in batches(ds, seed)]
        return [float(h) for h in hist]
    finally:
        _lib.set_option("stash_fp8", 0)


def main():
    root = write_synthetic_scene(tempfile.mkdtemp() + "/scene", n_train=16, n_test=2, size=64)
    ds = BlenderDataset(root, "train", 1, True, 1.0).to("cuda")
    trainers = [("fp32", lambda s: run_fp32(ds, s)), ("hip-bf16", lambda s: run_hip(ds, s, False)), ("hip-fp8", lambda s: run_hip(ds, s, True))]
    probe = min(150, STEPS) - 1
    dead = {name: 0 for name, _ in trainers}
    print(f"# {N_SEEDS} seeds x {STEPS} steps of {R} rays x {S} samples; dead = loss > 0.1 at step {probe + 1}")
    print("seed  " + "  ".join(f"{name:>28s}" for name, _ in trainers))
    for seed in range(N_SEEDS):
        cells = []
        for name, fn in trainers:
            h = fn(seed)
            is_dead = h[probe] > 0.1
            dead[name] += int(is_dead)
            cells.append(f"{'DEAD' if is_dead else 'ok  '} l150 {h[probe]:.4f} last {h[-1]:.4f}")
        print(f"{seed:4d}  " + "  ".join(f"{c:>28s}" for c in cells), flush=True)
    print("dead runs: " + ", ".join(f"{name} {dead[name]}/{N_SEEDS}" for name, _ in trainers))


if __name__ == "__main__":
    main()
