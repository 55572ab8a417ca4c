"""Option "deterministic" (include/nerf_hip.h; pytest -m gpu): the reference's gradients are plain sums
(loss.backward(), run.py:1941-1944); with the option on, every sum of the Instant / Part 4 steps whose order would depend
on scheduling takes an ordered form, and two runs of the same steps must give the SAME BITS -- losses, gradients,
parameters, occupancy grids.  Also: the ordered forms equal the default ones up to summation order, replicas that apply
the same all-reduced gradient stay bit-equal (the squared norm behind the clip coefficient is an ordered sum in every
mode), and a Part 4 rank without active samples issues its peers' collectives."""
import os

import numpy as np
import pytest
import torch
import yaml

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture
def det():
    import project_nerf_amd  # noqa: F401
    from project_nerf_amd import ops
    ops.set_deterministic(True)
    try:
        yield ops
    finally:
        ops.set_deterministic(False)


def _rays(n, seed):
    g = torch.Generator().manual_seed(seed)
    o = torch.randn(n, 3, generator=g)
    o = o / o.norm(dim=-1, keepdim=True) * 4.0311
    tgt = (torch.rand(n, 3, generator=g) - 0.5) * 1.6
    d = tgt - o
    return o.cuda(), (d / d.norm(dim=-1, keepdim=True)).cuda(), torch.rand(n, 3, generator=g).cuda(), g


def _sphere(res, radius=1.2):
    ax = torch.linspace(-1.5, 1.5, res)
    gx, gy, gz = torch.meshgrid(ax, ax, ax, indexing="ij")
    return ((gx ** 2 + gy ** 2 + gz ** 2) < radius ** 2).cuda()


def test_ordered_compaction_puts_the_slots_in_sample_order(det):
    ops = det
    R, S = 777, 48                                                # 37,296 samples: ten passes, the last one ragged
    o, d, _, g = _rays(R, 2)
    u = torch.rand(R, S, generator=g).cuda()
    grid = _sphere(64)
    z1, s1, p1, v1 = ops.sample_compact(o, d, 2.0, 6.0, S, grid, 1.5, u=u)
    ops.set_deterministic(False)
    z0, s0, p0, v0 = ops.sample_compact(o, d, 2.0, 6.0, S, grid, 1.5, u=u)
    ops.set_deterministic(True)
    assert torch.equal(z0, z1) and p0.shape == p1.shape and p1.shape[0] > 0
    assert torch.equal(s0 >= 0, s1 >= 0)
    act = s1 >= 0
    assert torch.equal(s1[act].long(), torch.arange(int(act.sum()), device="cuda"))          # sample order
    assert torch.equal(p1, p0[s0[act].long()]) and torch.equal(v1, v0[s0[act].long()])       # the same points and directions, reordered
    # the jitter drawn in the kernel, a shard of a larger batch, and the plain depths
    a = ops.sample_compact_async(o, d, 2.0, 6.0, S, grid, 1.5, jitter=(5, 9), first_ray=100).get()
    ops.set_deterministic(False)
    b = ops.sample_compact_async(o, d, 2.0, 6.0, S, grid, 1.5, jitter=(5, 9), first_ray=100).get()
    c = ops.sample_compact(o, d, 2.0, 6.0, S, grid, 1.5)
    ops.set_deterministic(True)
    e = ops.sample_compact(o, d, 2.0, 6.0, S, grid, 1.5)
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2][b[1][b[1] >= 0].long()])
    assert torch.equal(c[0], e[0]) and torch.equal(e[2], c[2][c[1][c[1] >= 0].long()])
    # nothing active, and an empty batch
    none = ops.sample_compact(o, d, 2.0, 6.0, S, torch.zeros(8, 8, 8, dtype=torch.bool, device="cuda"), 1.5, u=u)
    assert none[2].shape[0] == 0 and bool((none[1] == -1).all())


def test_squared_norm_is_an_ordered_sum():
    """nerf_tv_normsq*: one partial per workgroup, added in workgroup order (every mode): the same bits on every launch"""
    import project_nerf_amd  # noqa: F401
    from project_nerf_amd import _lib, ops
    lib = _lib.load()
    gen = torch.Generator().manual_seed(3)
    n = 3 * 1000003 + 1
    p, g0 = torch.randn(n, generator=gen).cuda(), torch.randn(n, generator=gen).cuda()
    st = torch.cuda.current_stream().cuda_stream
    vals = []
    for _ in range(6):
        ws, g = ops.normsq_ws("cuda"), g0.clone()
        ws[1] = 7.0                                                # a stale ticket: the zeroing form clears it
        _lib.check(lib.nerf_tv_normsq(p.data_ptr(), g.data_ptr(), n, 0.3, 0.5, ws.data_ptr(), st), "nerf_tv_normsq")
        _lib.check(lib.nerf_tv_normsq_accum(p.data_ptr(), g.data_ptr(), 4099, 0.0, 1.0, ws.data_ptr(), st), "nerf_tv_normsq_accum")
        vals.append((float(ws[0]), float((g.double() ** 2).sum() + (g[:4099].double() ** 2).sum())))
        assert float(ws[1]) == 0.0                                 # the ticket is back at zero
    assert len({v for v, _ in vals}) == 1, vals
    assert abs(vals[0][0] - vals[0][1]) <= 2e-6 * vals[0][1]


def _instant_run(steps, seed=0):
    from project_nerf_amd import ops
    from project_nerf_amd.engine import InstantNgpEngine
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "part2_instant.yaml.example")))
    cfg.update(train_iters=steps, grid_resolution=64, grid_threshold=0.01)
    eng = InstantNgpEngine(cfg, seed=seed)
    g = torch.Generator().manual_seed(11)
    eng.table.copy_(((torch.rand(eng.table.numel(), generator=g) - 0.5) * 0.5).cuda())
    eng.net[2048:2048 + 64] *= 20.0
    ops.imlp_pack(eng.net, eng.packed)
    eng.binary_grid = _sphere(64)
    R, S = 2048, 64
    o, d, target, _ = _rays(R, 4)
    losses, grads = [], None
    prepared = eng.prepare_batch(o, d, S)
    for step in range(1, steps + 1):
        nxt = eng.prepare_batch(o, d, S)                           # one batch ahead, as the training loop does
        losses.append(float(eng.train_step(o, d, target, S, prepared=prepared)))
        prepared = nxt
        if step == 1:
            grads = (eng.g_table.clone(), eng.g_net.clone())
        if step == steps // 2:
            eng.update_grid()
            prepared = eng.prepare_batch(o, d, S)
    return losses, grads, eng.table.clone(), eng.net.clone(), eng.binary_grid.clone()


def test_instant_training_runs_are_bit_equal(det):
    a, b = _instant_run(12), _instant_run(12)
    assert a[0] == b[0], (a[0], b[0])                              # every logged loss
    for x, y in zip(a[1], b[1]):
        assert torch.equal(x, y)
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    assert a[0][-1] < a[0][0] and float(a[1][0].abs().max()) > 0.0 and float(a[1][1].abs().max()) > 0.0


def test_instant_ordered_gradients_equal_the_default_gradients(det):
    ops = det
    d1 = _instant_run(1)
    ops.set_deterministic(False)
    d0 = _instant_run(1)
    ops.set_deterministic(True)
    assert abs(d1[0][0] - d0[0][0]) <= 1e-6 * d0[0][0]
    for x, y, tol in ((d1[1][0], d0[1][0], 1e-5), (d1[1][1], d0[1][1], 1e-4)):      # table: integer sums in both; nets: bf16 MFMA sums
        assert float((x - y).norm() / y.norm()) < tol                               # regrouped over other spans of samples


PART4 = dict(mode="part4", n_levels=16, n_features_per_level=2, log2_hashmap_size=14, base_resolution=16, per_level_scale=1.5,
             scene_bound=1.5, L_embed_dir=4, L_embed_time=10, hidden_dim=64, time_modulation_dim=64, time_modulation_layers=2,
             deform_n_levels=12, deform_n_features_per_level=2, deform_log2_hashmap_size=12, deform_base_resolution=16,
             deform_per_level_scale=1.5, deform_hidden_dim=64, grid_resolution=32, learning_rate=1e-2, train_iters=40,
             deformation_reg_weight=0.05, use_coord_noise=True, coord_noise_std=1e-3, time_noise_std=1e-2)


def _part4_engine(seed=0, world=1):
    from project_nerf_amd.part4 import DualHashEngine
    eng = DualHashEngine(PART4, seed=seed, world_size=world)
    g = torch.Generator().manual_seed(21)
    with torch.no_grad():
        for k in range(4):
            lv = eng.levels_c if k == 3 else eng.levels_d
            t = torch.cat([(torch.rand(int(lv.size[l]) * 2, generator=g) - 0.5) * (8.0 / float(lv.res[l])) for l in range(lv.n_levels)])
            eng.table(k).copy_(t.cuda())
        eng.net.copy_((torch.randn(eng.net.numel(), generator=g) * 0.15).cuda())
        eng.net[16832 + 64 * 64:16832 + 64 * 64 + 64] *= 8.0         # the density row: visible densities
        eng.net[30144] = 0.1
    eng.repack()
    eng.binary_grid = _sphere(32)
    return eng


def _probes(step):
    g = torch.Generator(device="cuda").manual_seed(100 + step)
    rnd = lambda *s: torch.rand(*s, device="cuda", generator=g)
    return {"temporal": ((rnd(64, 3) * 2 - 1) * 1.5, rnd(64, 1) * 0.98, 0.02, 1e-4), "unsup": ((rnd(128, 3) * 2 - 1) * 1.5, rnd(128, 1), 1e-3),
            "anchor": ((rnd(128, 3) * 2 - 1) * 1.5, 0.01)}


def _part4_run(steps):
    eng = _part4_engine()
    R, S = 1024, 32
    o, d, target, g = _rays(R, 6)
    t = torch.rand(R, 1, generator=g).cuda()
    losses, grads = [], None
    for step in range(1, steps + 1):
        bg = torch.rand(3, generator=g).cuda() if step > 4 else None                   # random-background augmentation
        losses.append(float(eng.train_step(o, d, target, t, S, bg=bg, probes=_probes(step) if step % 3 == 0 else None)))
        if step in (1, 3):                                         # step 1: the data batch alone; step 3: with the regulariser probes
            grads = (grads or ()) + (eng.g_tables.clone(), eng.g_net.clone())
        if step == steps // 2:
            eng.update_grid(decay=0.95)
    return losses, grads, eng.tables.clone(), eng.net.clone(), eng.binary_grid.clone()


def test_part4_training_runs_are_bit_equal(det):
    a, b = _part4_run(10), _part4_run(10)
    assert a[0] == b[0], (a[0], b[0])
    for x, y in zip(a[1], b[1]):
        assert torch.equal(x, y)
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    assert float(a[1][2].abs().max()) > 0.0 and float(a[1][3][30144].abs()) > 0.0      # incl. the displacement-scale gradient
    assert all(np.isfinite(a[0]))


def test_part4_ordered_gradients_equal_the_default_gradients(det):
    ops = det
    d1 = _part4_run(3)
    ops.set_deterministic(False)
    d0 = _part4_run(3)
    ops.set_deterministic(True)
    # the first step's gradients: the same sums in another order (tables: integer sums in both modes; networks: bf16 MFMA sums
    # regrouped over other spans of samples)
    assert abs(d1[0][0] - d0[0][0]) <= 1e-6 * d0[0][0]
    rel = [float((x - y).norm() / y.norm()) for x, y in zip(d1[1], d0[1])]
    print(f"[part4 ordered vs default sums] step 1: tables {rel[0]:.2e}, networks {rel[1]:.2e}; step 3 (two optimiser steps later): "
          f"tables {rel[2]:.2e}, networks {rel[3]:.2e}; losses {d1[0]} vs {d0[0]}")
    assert rel[0] < 1e-4 and rel[1] < 2e-2, rel
    # two AdamW steps later the runs are NOT close any more: AdamW turns every gradient entry, however small, into a step of
    # +-lr (2x lr for the tables), so an entry whose sum cancels to rounding noise moves by +-lr with the sign of that noise --
    # the training map amplifies summation-order differences (tests/studies/part4_sensitivity.py measures the same growth from a
    # one-ulp change of one weight with the ordered sums on).  Hence no bound here.


def test_hash_input_gradient_ordered_equals_atomic(det):
    ops = det
    t = ops.HashLevelTable(16, 14, 16, 1.5)
    gen = torch.Generator().manual_seed(2)
    table = (torch.rand(t.entries, 2, generator=gen) - 0.5).cuda()
    pts = ((torch.rand(5000, 3, generator=gen) - 0.5) * 3.2).cuda()
    d_feat = torch.randn(5000, 32, generator=gen).cuda()
    a1, a2 = ops.hash_encode_bwd_input(pts, table, t, 1.5, d_feat), ops.hash_encode_bwd_input(pts, table, t, 1.5, d_feat)
    h1 = ops.hash_encode_bwd_input(pts, table.half(), t, 1.5, d_feat)
    ops.set_deterministic(False)
    b, hb = ops.hash_encode_bwd_input(pts, table, t, 1.5, d_feat), ops.hash_encode_bwd_input(pts, table.half(), t, 1.5, d_feat)
    ops.set_deterministic(True)
    assert torch.equal(a1, a2)
    assert float((a1 - b).abs().max()) <= 1e-5 * float(b.abs().max()) and float((h1 - hb).abs().max()) <= 1e-5 * float(hb.abs().max())
    assert float(a1[(pts.abs() > 1.5)].abs().max()) == 0.0            # outside the box: the clamp passes no gradient


def test_replicas_that_apply_the_same_summed_gradient_stay_bit_equal():
    """Two data-parallel replicas in one process: each computes its shard's gradient, both receive the SAME sum (what the
    all-reduce leaves on every rank) and step -- with the clip active -- for several steps: tables, networks, Adam state and
    occupancy grids stay bit-equal.  (The clip coefficient comes from a squared norm that is summed in a fixed order in every
    mode; with one float atomic per workgroup it differed in the last bit between ranks and the replicas drifted apart.)"""
    import project_nerf_amd  # noqa: F401
    R, S, world = 1024, 32, 2
    o, d, target, g = _rays(R, 8)
    t = torch.rand(R, 1, generator=g).cuda()
    reps = [_part4_engine(world=world) for _ in range(world)]
    for e in reps:
        e.max_norm = 1e-3                                              # the clip is active on every step
    half = R // world
    for step in range(1, 7):
        for r, e in enumerate(reps):
            sl = slice(r * half, (r + 1) * half)
            e.compute_gradients(o[sl].contiguous(), d[sl].contiguous(), target[sl].contiguous(), t[sl].contiguous(), S, first_ray=r * half,
                                probes=_probes(step) if step % 2 == 0 else None)
        sum_t, sum_n = reps[0].g_tables + reps[1].g_tables, reps[0].g_net + reps[1].g_net
        for e in reps:
            e.g_tables.copy_(sum_t)
            e.g_net.copy_(sum_n)
            assert float(e._normsq_ws[0]) >= 0.0
            e.apply_gradients()
        assert float(reps[0]._normsq_ws[0]) > e.max_norm ** 2          # clipping happened
        if step == 3:
            for e in reps:
                e.update_grid(decay=0.95)
        a, b = reps
        assert torch.equal(a.tables, b.tables) and torch.equal(a.net, b.net) and torch.equal(a.tables_h, b.tables_h), step
        assert torch.equal(a.state["tables"][1], b.state["tables"][1]) and torch.equal(a.binary_grid, b.binary_grid), step


def test_part4_rank_without_active_samples_issues_the_same_collectives():
    """DualHashEngine.compute_gradients on a shard that misses every occupied cell must enter exactly the collectives of a
    busy rank -- same order, same element counts -- with and without the regulariser probes (mismatched collectives hang or
    corrupt memory in RCCL)."""
    import project_nerf_amd  # noqa: F401
    eng = _part4_engine()
    R, S = 256, 32
    o, d, target, g = _rays(R, 12)
    t = torch.rand(R, 1, generator=g).cuda()
    busy_grid, idle_grid = _sphere(32), torch.zeros(32, 32, 32, dtype=torch.bool, device="cuda")

    def schedule(grid, probes):
        calls = []
        eng.binary_grid = grid
        loss = eng.compute_gradients(o, d, target, t, S, sync_grads_async=lambda v: calls.append((v.data_ptr(), v.numel())), probes=probes)
        assert bool(torch.isfinite(loss))
        return calls
    for probes in (None, _probes(1)):
        busy, idle = schedule(busy_grid, probes), schedule(idle_grid, probes)
        assert busy == idle and len(busy) == 5, (busy, idle)
        assert [nel for _, nel in busy] == [eng.table_sizes[3], eng.table_sizes[0], eng.table_sizes[1], eng.table_sizes[2], eng.net.numel()]
    # the idle rank without probes contributes zeros; its loss is the background's
    schedule(idle_grid, None)
    assert float(eng.g_tables.abs().max()) == 0.0 and float(eng.g_net.abs().max()) == 0.0
