"""Data parallelism through the PRODUCT entry point (pytest -m gpu): `run.py --config ...` launched as the driver
launches multi-GPU work -- `python -m torch.distributed.run --nproc-per-node 2 run.py ...`, one process per rank --
against the same command as a single process.  Both ranks share the box's one GPU (NERF_SINGLE_DEVICE=1) and talk
over gloo instead of RCCL (RCCL refuses two ranks on one device), so this covers everything but the transport:
the process group is up before the first GPU call, every rank forms its shard of ONE global batch, the flat
gradients are summed and averaged (clip after the all-reduce), evaluation renders row bands and gathers them, rank
0 alone prints.  Done-criterion of VERDICT r2 item 5: the loss trajectory of the two-rank run equals the single-rank
run on the concatenated batch."""
import os
import re
import subprocess
import sys

import pytest
import yaml

from conftest import ROOT

pytestmark = pytest.mark.gpu

CONFIGS = {
    "part2_nerf": dict(mode="part2_nerf", L_embed=10, L_embed_dir=4, hidden_dim=256, num_layers=8, skip_layer=4, view_dim=128,
                       n_samples=64, render_n_samples=64, batch_size=2048, train_iters=40, learning_rate=5e-4, log_every=5,
                       save_every=0, chunk=4096, downscale=1, seed=0),
    "part2_instant": dict(mode="part2_instant", n_levels=16, n_features_per_level=2, log2_hashmap_size=19, base_resolution=16,
                          per_level_scale=1.5, scene_bound=1.5, hidden_dim=64, L_embed_dir=4, n_samples=64, render_n_samples=64,
                          batch_size=4096, train_iters=60, learning_rate=1e-2, log_every=5, val_every=10000, chunk=8192, downscale=1,
                          use_density_grid=True, grid_resolution=128, grid_threshold=0.01, grid_warmup_iters=16, seed=0,
                          dp_gradient_wire="fp32"),
}


# the replicated optimiser (all-reduce of the whole table gradient level group by level group) with the DEFAULT bf16 wire
CONFIGS["part2_instant_replicated_bf16_wire"] = dict(CONFIGS["part2_instant"], dp_sharded_optimizer=False)
del CONFIGS["part2_instant_replicated_bf16_wire"]["dp_gradient_wire"]


CONFIGS["part4"] = dict(
    mode="part4", downscale=1, white_bkgd=True, near=2.0, far=6.0, n_samples=32, render_n_samples=32, batch_size=2048, chunk=4096,
    train_iters=48, learning_rate=1e-2, weight_decay=1e-5, eta_min=1e-4, max_grad_norm=1.0, log_every=2, val_every=10000,
    use_tv_displacement=True, tv_displacement_weight=1e-4, tv_loss_weight=1e-6, deformation_reg_weight=1e-4,
    use_temporal_smooth=True, use_static_anchor=True, use_unsupervised_consistency=True, use_coord_noise=True, coord_noise_std=1e-3,
    time_noise_std=1e-2, use_random_bg=True, random_bg_start=24, deform_n_levels=12, deform_n_features_per_level=2,
    deform_log2_hashmap_size=12, deform_base_resolution=16, deform_per_level_scale=1.5, deform_hidden_dim=64, L_embed_time=10,
    time_modulation_dim=64, time_modulation_layers=2, n_levels=16, n_features_per_level=2, log2_hashmap_size=14, base_resolution=16,
    per_level_scale=1.5, scene_bound=1.5, hidden_dim=64, use_density_grid=True, grid_resolution=32, grid_threshold=0.01,
    grid_warmup_iters=8, grid_stop_ratio=0.9, seed=0)


@pytest.fixture(scope="module")
def scene(tmp_path_factory):
    """static frames for Part 2, the same frames with time stamps (a D-NeRF style root) for Part 4"""
    import json
    from src.dataset import write_synthetic_scene
    root = write_synthetic_scene(str(tmp_path_factory.mktemp("dp_scene") / "s"), n_train=8, n_test=2, size=48)
    for split in ("train", "test"):
        path = os.path.join(root, f"transforms_{split}.json")
        meta = json.load(open(path))
        n = len(meta["frames"])
        for k, frame in enumerate(meta["frames"]):
            frame["time"] = k / max(n - 1, 1)
        json.dump(meta, open(path, "w"))
    return root


def _run(cfg_path, scene, log_dir, world, port):
    env = dict(os.environ, NERF_SINGLE_DEVICE="1", NERF_DIST_BACKEND="gloo")
    tail = [os.path.join(ROOT, "run.py"), "--config", cfg_path, "--data_dir", scene, "--render_n", "1"]
    if world == 1:
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + tail
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=log_dir, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    losses = [float(m.group(2)) for m in re.finditer(r">>> Step (\d+)/\d+ \| Loss ([0-9.eE+-]+)", r.stdout)]
    psnr = [float(m.group(1)) for m in re.finditer(r">>> Test PSNR: ([0-9.]+) dB", r.stdout)]
    return losses, psnr, r.stdout


@pytest.mark.parametrize("mode", sorted(CONFIGS))
def test_two_rank_cli_run_follows_the_single_rank_trajectory(mode, scene, tmp_path):
    cfg = dict(CONFIGS[mode])
    out = {}
    for world in (1, 2):
        work = tmp_path / f"w{world}"
        work.mkdir()
        cfg["log_dir"] = str(work / "out")
        cfg_path = str(work / "cfg.yaml")
        with open(cfg_path, "w") as f:
            yaml.safe_dump(cfg, f)
        out[world] = _run(cfg_path, scene, str(work), world, 29700 + sorted(CONFIGS).index(mode))
    (l1, p1, s1), (l2, p2, s2) = out[1], out[2]
    n_logs = cfg["train_iters"] // cfg["log_every"]
    assert len(l1) == len(l2) == n_logs, (s1[-800:], s2[-800:])          # rank 0 alone prints: one line per logged step
    assert "data parallel: 2 ranks" in s2 and len(p1) == len(p2) == 1
    if mode != "part2_nerf" and cfg.get("dp_sharded_optimizer", True):
        assert "sharded optimiser: every rank steps" in s2              # reduce-scatter + sliced TV / AdamW + all-gather (sharded.py)
    if mode != "part2_nerf":
        # the replicas end BIT-EQUAL (tables, networks, occupancy grid): identical summed gradients, ONE squared norm that is the
        # same bits on every rank, replicated grid updates -- whatever summation-order noise did to the trajectory itself
        assert "replica divergence after" in s2 and "steps: 0.000e+00" in s2, s2[-1500:]
    # same global batch, same jitter, gradients summed and averaged: the first logged steps agree with the single-rank run up to
    # summation order (measured: 3e-5).  Later steps are a SMOKE test only for the hash-grid modes: their training map amplifies a
    # 1e-5 relative perturbation of the tables to a 3 % change of the next step's gradients and to O(1) after two more
    # (tests/studies/part4_mode_diff.py, profiles/r04_part4_sensitivity.txt: the same growth with the ordered sums on, i.e. without
    # any summation-order noise) -- a two-rank run is another partition of the same sums, so its trajectory leaves the single-rank
    # one after a few steps by construction.  What IS held tightly: the shards' summed gradients equal the full batch's
    # (tests/test_gpu_data_parallel.py), the sharded optimiser equals the replicated one from the same state
    # (tests/test_gpu_sharded_optimizer.py), the replicas stay bit-equal (above).
    early = 3 if mode == "part4" else (len(l1) if mode == "part2_nerf" else 2)
    for k, (a, b) in enumerate(zip(l1, l2)):
        if k >= early:
            break
        bound = 2e-3 if k < 1 else 2e-2
        assert abs(a - b) <= bound * max(a, 1e-3), (k, l1, l2)
    assert l1[-1] < l1[0] and l2[-1] < l2[0]                              # both train
    assert all(v == v and v < 1e3 for v in l2)
    # row-band evaluation = whole-frame evaluation (vanilla: same weights up to summation order); the hash-grid modes end at
    # different weights (above): their PSNR is a smoke test
    assert abs(p1[0] - p2[0]) < (0.5 if mode == "part2_nerf" else 4.0), (p1, p2)
