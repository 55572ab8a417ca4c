#!/usr/bin/env python
"""Headline benchmark: vanilla NeRF (BASELINE.json configs[1]) on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one training step of the hot path (reference run.py:312-338) on one batch of synthetic
rays: 4096 rays x 64 stratified samples (configs/part2.yaml.example: batch_size 4096, n_samples 64):
batch sampling from 100 GPU-resident 800x800 frames + jittered stratified depths (nerf_train_batch: pixel
draws, rays, composited targets and depths in one kernel) -> fused bf16-MFMA decoder fwd (+ bf16 training images)
-> composite + MSE + backward (one kernel) -> dgrad chain -> wgrad (bf16 MFMA) -> (RCCL all-reduce) -> Adam ->
weight repack, inputs resident in HBM.  ``value`` / ``ms_per_step`` / ``dtype`` are THIS step: every MFMA in it
contracts bf16 operands (BASELINE configs[1]).  The same step on 8-bit training images (library option stash_fp8:
e4m3 x e5m2 weight-gradient operands, narrower than the config's precision) is timed afterwards and reported
beside it as ``value_fp8_images`` -- a labelled secondary figure, never the headline.  Rays shard across ranks (weak scaling: every rank owns its own
4096-ray batch; one gradient all-reduce per step).  ``value`` = rays/s summed over ranks.  The same run
also reports: the 800x800, 128-samples/ray render (``render_fps``); every phase of the step timed INSIDE
the step with HIP events on the launch stream (``kernels_in_step``, what rocprofv3's per-kernel averages
in profiles/ show) and each kernel launched back to back on its own (``kernels``); the roofline of the
dominant kernel from its in-step time; the step against the MFMA roof SURVEY 8(d) binds configs[1] to
(``step_frac_of_binding_roofline``); on rank 0 at N=1 the CPU oracle on a bounded sample of the same step
(``cpu_baseline``) and a shortened Instant-NGP run (configs[2]: ``instant``).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_FLOP = 1186816           # per sample, forward  (2 x 593,408 MAC)
TRAIN_FLOP = 3489024         # per sample, fwd + dgrad + wgrad
WGRAD_ELEMS = (64 + 8 * 256 + 256 + 128 + 32) + (16 + 128 + 256 + 8 * 256)   # image elements per sample read by wgrad: 4976
DGRAD_FLOP = 2 * (593408 - 35712 - 256 * 63)   # transposed chain, code columns and layer 0 skipped
WGRAD_FLOP = 2 * 593408
MFMA_PEAK_TFLOPS = 2500.0    # gfx950 dense bf16 (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0


def kernel_sources_sha16():
    """fingerprint of the HIP sources the library is built from (the generated headers follow from gen_*.py)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "project-nerf_amd", "csrc", "*"))):
        name = os.path.basename(path)
        if os.path.isfile(path) and name.endswith((".hip", ".h", ".cpp", ".py")) and name not in ("mlp_stream_asm.h", "mlp_mtile_asm.h"):
            h.update(name.encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def pmc_summary(name):
    """HBM bytes per launch from a committed rocprofv3 --pmc summary (profiles/<name>; tools/pmc_r04.sh + tools/pmc_summarize.py:
    FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE, separate passes).  The counters cannot be read inside this
    process, so the numbers are quoted from that file -- and ONLY while the kernels are the ones they were taken on: the
    summary records the fingerprint of csrc/ at collection time; if the tree's differs (or the file has none), every
    `traffic` of the line is null and `traffic_source.stale` says why.  Returns (per-kernel dict or {}, source record)."""
    path = os.path.join(ROOT, "profiles", name)
    src = {"file": "profiles/" + name, "kernel_sources_sha16_now": kernel_sources_sha16()}
    try:
        with open(path) as f:
            data = json.load(f)
    except OSError:
        return {}, dict(src, stale="summary file not found")
    meta = data.pop("_meta", {})
    src.update(kernel_sources_sha16_at_collection=meta.get("kernel_sources_sha16"), git_head_at_collection=meta.get("git_head"))
    if meta.get("kernel_sources_sha16") != src["kernel_sources_sha16_now"]:
        return {}, dict(src, stale="the kernels' sources have changed since the counters were collected: traffic withheld")
    return data, dict(src, stale=False)


def synth_rays(n, seed, device):
    """Cameras on the NeRF-Synthetic hemisphere (radius 4.0311) looking at the scene."""
    g = torch.Generator().manual_seed(seed)
    o = torch.randn(n, 3, generator=g)
    o[:, 2] = o[:, 2].abs()
    o = o / o.norm(dim=-1, keepdim=True) * 4.0311
    tgt = (torch.rand(n, 3, generator=g) - 0.5) * 2.0
    d = tgt - o
    d = d / d.norm(dim=-1, keepdim=True)
    target = torch.rand(n, 3, generator=g)
    return o.to(device), d.to(device), target.to(device)


def event_ms(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def cpu_baseline(rays, samples, threads):
    """The oracle (CPU restatement of the reference, fp32 PyTorch) on a bounded sample of the step."""
    from oracle import nerf_oracle as O
    torch.set_num_threads(threads)
    params = {k: v.clone().requires_grad_(True) for k, v in O.nerf_init_params(seed=0).items()}
    opt = torch.optim.Adam(list(params.values()), lr=5e-4)
    o, d, target = synth_rays(rays, 1, "cpu")
    field = lambda p, v: O.nerf_field(params, p, v)
    times = []
    for it in range(3):
        t0 = time.perf_counter()
        pred, _, _ = O.render_rays(field, o, d, 2.0, 6.0, samples, True)
        loss = torch.nn.functional.mse_loss(pred, target)
        opt.zero_grad()
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
    step_s = min(times[1:])
    return {"value": rays / step_s, "unit": "rays/s", "cores": threads, "kind": "port", "seconds_of_cpu_work": sum(times),
            "sample": f"train step on {rays} rays x {samples} samples (1 warm-up + 2 timed), oracle/nerf_oracle.py fp32"}


def bench_instant(args, device, iters=1000, size=800, n_train=30, standalone=False):
    """Instant-NGP variant (BASELINE.json configs[2]) on the flat-parameter engine (same kernels as
    NeuralField + DensityGrid + render_rays + AdamW): analytic scene rendered at 800 x 800 on the GPU
    (ground truth only), wall time to PSNR, steady-state train rays/s with the occupancy grid active,
    per-kernel times and rooflines, 800 x 800 render FPS at the trained field."""
    import numpy as np
    import yaml
    from src.dataset import BlenderDataset, SYNTHETIC_CAMERA_ANGLE, look_at_pose, synthetic_frames
    from project_nerf_amd import ops
    from project_nerf_amd.engine import InstantNgpEngine
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "part2_instant.yaml.example")))
    batch, S = 16384, 128
    cfg["train_iters"] = iters
    t_gen = time.perf_counter()
    images, poses = synthetic_frames(n_train + 2, size, device, n_samples=192)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    ds = BlenderDataset.from_tensors(images[:n_train], poses[:n_train], SYNTHETIC_CAMERA_ANGLE)
    test = BlenderDataset.from_tensors(images[n_train:], poses[n_train:], SYNTHETIC_CAMERA_ANGLE)
    torch.manual_seed(0)
    eng = InstantNgpEngine(cfg, device=str(device), seed=0)
    bg = eng.bg

    def psnr():
        vals = []
        for i in range(len(test)):
            o, d, tgt = test.get_image_rays(i, device)
            img = eng.render_image(o, d, S)
            vals.append(-10 * np.log10(float(((img.clamp(0, 1) - tgt) ** 2).mean())))
        return float(np.mean(vals))

    # one batch ahead: the draw, the rays and the compaction of batch i+1 are queued BEFORE step i's kernels, so
    # the active count of batch i+1 is on the host by the time step i ends and no step waits for its own count
    # (NERF_BENCH_NO_PREFETCH=1: draw and compact in line, one host wait per step as in the reference)
    pipelined = os.environ.get("NERF_BENCH_NO_PREFETCH") is None
    ahead = []

    def draw():
        o, d, target = ds.sample_batch(batch, bg)
        return o, d, target, eng.prepare_batch(o, d, S)

    def step():
        if not pipelined:
            o, d, target = ds.sample_batch(batch, bg)
            return eng.train_step(o, d, target, S)
        if not ahead:
            ahead.append(draw())
        o, d, target, prepared = ahead.pop()
        ahead.append(draw())
        return eng.train_step(o, d, target, S, prepared=prepared)

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    curve, active = [], 1.0
    for it in range(1, iters + 1):
        step()
        interval = 32 if it < iters * 0.1 else (128 if it < iters * 0.5 else 512)      # run.py:636-641
        if it < iters * 0.9 and it >= 256 and it % interval == 0:
            active = eng.update_grid()
            ahead.clear()                      # the waiting batch was compacted against the previous grid
        if it in (300, 600, iters):
            torch.cuda.synchronize()
            curve.append({"step": it, "train_seconds": time.perf_counter() - t0, "test_psnr_db": psnr()})
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # ---- per-kernel timings of one steady-state batch (rows a7 / a8 of SURVEY 8) ----
    lib = ops._lib.load()
    P = lambda t: t.data_ptr()
    o, d, _ = ds.sample_batch(batch, bg)
    uu = torch.rand(batch, S, device=device)
    z, slots, pts, dirs = ops.sample_compact(o, d, eng.near, eng.far, S, eng.binary_grid, eng.bound, u=uu)
    n = pts.shape[0]
    L = eng.levels.n_levels
    ws = torch.empty(lib.nerf_imlp_workspace_bytes(n), device=device, dtype=torch.uint8)
    rgb, sigma = torch.empty(n, 3, device=device), torch.empty(n, device=device)
    d_rgb, d_sigma, d_feat = torch.randn_like(rgb), torch.randn_like(sigma), torch.empty(n, 2 * L, device=device)
    stv = ops._stream()
    k = {
        "batch_sampling": event_ms(lambda: ds.sample_batch(batch, bg), 20),
        "sample_compact": event_ms(lambda: ops.sample_compact(o, d, eng.near, eng.far, S, eng.binary_grid, eng.bound, u=uu), 20),
        "hash_fwd": event_ms(lambda: ops.hash_encode_fwd(pts, eng._gather_table(), eng.levels, eng.bound, want_f32=False, out_nat=ws), 20),
        "hash_fwd_fp32_table": event_ms(lambda: ops.hash_encode_fwd(pts, eng.table.view(-1, 2), eng.levels, eng.bound, want_f32=False, out_nat=ws), 20),
        "imlp_fwd": event_ms(lambda: lib.nerf_imlp_fwd(P(eng.packed), P(ws), P(dirs), n, P(rgb), P(sigma), 1, stv), 20),
        "imlp_bwd": event_ms(lambda: lib.nerf_imlp_bwd(P(eng.packed), P(ws), P(rgb), P(sigma), P(d_rgb), P(d_sigma), n,
                                                       P(eng.g_net), P(d_feat), stv), 20),
        "hash_bwd": event_ms(lambda: ops.hash_encode_bwd(pts, eng.levels, eng.bound, d_feat, eng.g_table,
                                                         workspace=eng._hash_bwd_workspace(n)), 20),
        "tv_clip_adamw(table)": event_ms(lambda: ops.tv_clip_adamw_step(eng.table, eng.g_table, *eng.state["table"], 1, 0.0,
                                                                         tv_weight=eng.tv_weight, max_norm=1.0, weight_decay=eng.wd,
                                                                         grad_scale=1.0, scratch=eng._scratch), 20),
    }
    # algorithmic bytes: 8 corners x 8 B per level and point (+ 12 B in, 4 B per feature out); atomics likewise;
    # the fused regulariser + optimiser streams params, grads and both moments (read) and params + moments (write)
    gather = n * L * 8 * (4 if eng.half_table else 8)
    n_tab = eng.table.numel()
    roof = {
        "hash_fwd": {"bound": "hbm", "kernel": "hash_fwd_kernel", "achieved": (gather + n * (12 + 2 * L * 2)) / k["hash_fwd"] * 1e-6,
                     "note": "table gathers from the fp16 copy of the table (mostly L2 / Infinity Cache hits: 26 MB re-read by every batch)"},
        "hash_bwd": {"bound": "hbm", "kernel": "hash_bin_count + _plan + _scatter + _reduce kernels",
                     "achieved": (n * L * 8 * 8 + n * (12 + 2 * L * 4)) / k["hash_bwd"] * 1e-6,
                     "note": "binned scatter: 8-byte corner records written once and read once (16 B per corner against the 8 B "
                             "counted here), slice sums in LDS (64-bit fixed point), plain read-modify-write of the table"},
        "tv_clip_adamw(table)": {"bound": "hbm", "kernel": "tv_normsq_kernel + adamw_clip_kernel",
                                 "achieved": n_tab * 4 * 9 / k["tv_clip_adamw(table)"] * 1e-6},
        "imlp_fwd": {"bound": "hbm", "kernel": "imlp_fwd_kernel<true>", "achieved": n * (64 + 12 + 16 + 2 * (64 + 16 + 64 + 64 + 48)) / k["imlp_fwd"] * 1e-6,
                     "note": "operand image in, rgb/sigma out, bf16 stash of every layer input"},
    }
    # measured HBM bytes per launch from the committed rocprofv3 --pmc summary of `bench.py --workload instant`
    # (tools/pmc_r03.sh: FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE, separate passes); multi-kernel entries sum
    pmc, pmc_src = pmc_summary("r04_instant_pmc_summary.json")

    def traffic(*names):
        vals = [pmc.get(nm, {}).get("hbm_bytes_per_launch_corrected") for nm in names]
        return sum(vals) if vals and all(v is not None for v in vals) else None
    traffic_of = {"hash_fwd": ("hash_fwd_kernel<fp16 table>",),
                  "hash_bwd": ("hash_bin_count_pm_kernel", "hash_bin_plan_kernel", "hash_bin_scatter_kernel<true>", "hash_bin_reduce_kernel"),
                  "tv_clip_adamw(table)": ("tv_normsq_kernel<true>", "adamw_clip_kernel<true>"), "imlp_fwd": ("imlp_fwd_kernel<true>",)}
    for name, v in roof.items():
        v.update({"peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": v["achieved"] / HBM_PEAK_GBS, "traffic": traffic(*traffic_of[name])})
    H = W = 800
    focal = 0.5 * W / np.tan(0.5 * SYNTHETIC_CAMERA_ANGLE)
    c2w = torch.tensor(look_at_pose(4.0311 * np.array([0.6, 0.5, 0.62])), dtype=torch.float32)
    j, i = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    dd = torch.stack([(i - W * .5) / focal, -(j - H * .5) / focal, -torch.ones_like(i)], -1).reshape(-1, 3).float() @ c2w[:3, :3].T
    dd = (dd / dd.norm(dim=-1, keepdim=True)).to(device)
    oo = c2w[:3, 3].expand_as(dd).contiguous().to(device)
    eng.render_image(oo, dd, S)
    torch.cuda.synchronize()
    # frames timed one by one, median reported (a frame is 5 ms and waits on the host four times -- once per chunk for
    # the active count: a descheduled host thread on a shared box showed as a 6x slower frame in 2 of ~10 runs)
    frame_s = []
    for _ in range(max(args.render_frames, 7)):
        t1 = time.perf_counter()
        eng.render_image(oo, dd, S)
        torch.cuda.synchronize()
        frame_s.append(time.perf_counter() - t1)
    rt = float(np.median(frame_s))
    out = {
        "metric": "train rays/sec + 800x800 render FPS, NeRF-Synthetic Lego; PSNR parity", "value": batch * args.steps / dt,
        "unit": "rays/s", "n_gpus": 1, "steps": args.steps, "warmup": iters, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "Part 2 Instant-NGP (L16 F2 T2^19 hash grid + tiny MLPs, 128^3 occupancy grid), steady-state train step",
                   "rays_per_gpu": batch, "samples_per_ray": S, "active_ratio": active,
                   "scene": f"analytic scene, {n_train} training frames of {size}x{size} rendered on the GPU ({t_gen:.1f} s, not timed)"},
        "kernels": {kk: {"ms": v} for kk, v in k.items()}, "active_samples": n, "rooflines": roof, "traffic_source": pmc_src,
        "render_fps": 1.0 / rt, "render_ms_per_frame": rt * 1e3, "render_ms_per_frame_max": max(frame_s) * 1e3, "psnr_curve": curve,
        "reference_headline": "26+ dB in 5 min, 10+ FPS (RTX 4060 Laptop, Lego; README.md:12,136)"}
    if standalone:
        print(json.dumps(out))
    return out


def bench_part4(args, device, steps=200):
    """Part 4 dual-hash dynamic field (BASELINE.json configs[4], one GPU's share) on part4.DualHashEngine at
    configs/part4.yaml.example: 8192 rays x 64 samples, 64^3 occupancy grid at ~12 % active cells (what training reaches
    after pruning), 28.5 M parameters, the every-16th-step regulariser probes included.  HBM-bound (SURVEY 8(d) cfg C's
    reasoning: hash gathers / scatters and parameter streaming); per-kernel-group rooflines with PMC traffic."""
    import yaml
    from src.core import NeuralField
    from project_nerf_amd import ops
    from project_nerf_amd import part4 as p4
    from project_nerf_amd.dynamic import part4_probe_draws
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "part4.yaml.example")))
    torch.manual_seed(0)
    model = NeuralField(cfg).to(device)
    R, S = cfg["batch_size"], cfg["n_samples"]
    o = torch.nn.functional.normalize(torch.randn(R, 3, device=device), dim=-1) * 4.03
    d = torch.nn.functional.normalize(-o + 0.3 * torch.randn(R, 3, device=device), dim=-1)
    t, target = torch.rand(R, 1, device=device), torch.rand(R, 3, device=device)
    eng = p4.DualHashEngine(cfg, device=str(device), seed=0)
    eng.load_from_model(model)
    eng.binary_grid = torch.rand_like(eng.grid) < 0.12
    step_no = [300]
    ahead = []

    def step():
        # the product loop's order (project-nerf_amd/dynamic.py::run_dynamic): the NEXT batch's compaction is queued ahead of this step's
        # kernels, so the read-back of its active count never stalls the host (the reference waits at that point of every step)
        step_no[0] += 1
        if not ahead:
            ahead.append(eng.prepare_batch(o, d, S))
        prepared = ahead.pop()
        ahead.append(eng.prepare_batch(o, d, S))
        return eng.train_step(o, d, target, t, S, prepared=prepared, probes=part4_probe_draws(cfg, step_no[0], device))
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # kernel groups launched back to back on the steady-state batch (HIP events on the launch stream)
    z, slots, pts, dirs = ops.sample_compact(o, d, eng.near, eng.far, S, eng.binary_grid, eng.bound)
    n = pts.shape[0]
    ws = p4.Workspace(n, device)
    _, t_def = p4.sample_inputs(slots, pts, t, R, S)
    tabs = eng._tables_for_forward()
    fwd = lambda: p4.forward_chain(eng.packed, eng.net, tabs, eng.levels_d, eng.levels_c, eng.bound, pts, None, t_def, dirs, ws, True)
    rgb, sigma, dx, xc = fwd()
    d_rgb, d_sigma, d_dx = torch.randn_like(rgb) * 1e-3, torch.randn_like(sigma) * 1e-3, torch.randn_like(dx) * 1e-3
    g_tabs = [eng.g_table(i) for i in range(4)]
    Ld, Lc = eng.levels_d.n_levels, eng.levels_c.n_levels
    k = {
        "hash_fwd (3 deformation grids + canonical)": event_ms(lambda: (
            ops.hash_encode_fwd_nat_tables(pts, tabs[:3], eng.levels_d, eng.bound, [ws.nat(i) for i in range(3)], fp16=True),      # as the step calls it
            ops.hash_encode_fwd_nat(pts, tabs[3], eng.levels_c, eng.bound, ws.nat(3), fp16=True)), 20),
        "hash_fwd + fused chains fwd": event_ms(fwd, 20),
        "chains bwd + tiny-MLP wgrad + hash input gradient + 4 scatters": event_ms(
            lambda: p4.backward_chain(eng.packed, eng.net, eng.table(3, half=True), eng.levels_d, eng.levels_c, eng.bound, pts, xc, ws, rgb, sigma,
                                      d_rgb, d_sigma, d_dx.clone(), eng.g_net, g_tabs, hash_ws=eng._hash_scratch, overwrite=True,
                                      tables_ws=eng._hash_scratch_tables), 20),   # as the engine's step calls it
        "tv + clip + adamw (28.5 M parameters)": event_ms(eng.apply_gradients, 20),
    }
    n_par = eng.tables.numel() + eng.net.numel()
    pmc, pmc_src = pmc_summary("r04_part4_pmc_summary.json")

    def traffic(names):
        # kernels launched at several sizes per step (four grids, three optimiser groups): mean bytes per launch x launches per step
        vals = [pmc.get(nm, {}).get("hbm_bytes_per_launch_mean") for nm in names]
        return sum(v * c for v, c in zip(vals, names.values())) if vals and all(v is not None for v in vals) else None
    # algorithmic bytes: forward gathers 8 corners x 4 B (fp16 pairs) per level and point; the scatter's read-modify-write 8 corners
    # x 16 B (+ the canonical grid's input gradient: 8 corners x 8 B); the optimiser streams params / grads / moments
    # (TV + norm: 12 B, AdamW + fp16 copy: 30 B per parameter)
    gather = n * (3 * Ld + Lc) * 8 * 4
    scatter = n * (3 * Ld + Lc) * 8 * 16 + n * Lc * 8 * 8
    roof = {
        "hash_fwd": {"kernel": "hash_fwd_kernel<fp16 table> x4", "work_per_launch": gather + n * (12 + 4 * 64),
                     "ms": k["hash_fwd (3 deformation grids + canonical)"], "traffic": traffic({"hash_fwd_kernel<fp16 table>": 2})},      # the three deformation grids in one launch + the canonical grid
        "backward": {"kernel": "p4 chains bwd + mlp_wgrad_small_kernel<true> x2 + hash_bwd_input_kernel + hash_bin_* x2 (deformation grids in one pass, canonical grid)", "work_per_launch": scatter,
                     "ms": k["chains bwd + tiny-MLP wgrad + hash input gradient + 4 scatters"],
                     "traffic": traffic({"hash_bin_count_pm_kernel": 2, "hash_bin_plan_kernel": 2, "hash_bin_scatter_kernel<true>": 2,
                                         "hash_bin_reduce_kernel": 2, "hash_bwd_input_kernel<fp16 table>": 1, "p4::canon_bwd_kernel": 1,
                                         "p4::deform_bwd_kernel": 1, "mlp_wgrad_small_kernel<true>": 2})},
        "tv_clip_adamw": {"kernel": "tv_normsq_kernel<true> x5 + adamw_clip_kernel<true> x3", "work_per_launch": n_par * 42,
                          "ms": k["tv + clip + adamw (28.5 M parameters)"],
                          "traffic": traffic({"tv_normsq_kernel<true>": 3, "adamw_clip_kernel<true>": 3})},
    }
    for v in roof.values():
        v.update({"bound": "hbm", "achieved": v["work_per_launch"] / v["ms"] * 1e-6, "peak": HBM_PEAK_GBS, "unit": "GB/s"})
        v["frac"] = v["achieved"] / HBM_PEAK_GBS
    step_bytes = gather + scatter + n_par * 42 + n * 1200          # + ~1.2 kB per sample of tiny-MLP training images
    return {"value": R * steps / dt, "unit": "rays/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "dtype": "bf16",
            "config": {"workload": "Part 4 dual-hash dynamic field train step (configs/part4.yaml.example), DualHashEngine", "rays_per_gpu": R,
                       "samples_per_ray": S, "active_samples": n, "parameters": n_par,
                       "regulariser_probes": "every 16th / 32nd step, through the same kernels"},
            "kernels": {kk: {"ms": v} for kk, v in k.items()}, "rooflines": roof, "traffic_source": pmc_src,
            "step_roofline": {"bound": "hbm", "achieved": step_bytes / (dt / steps) * 1e-9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": step_bytes / (dt / steps) * 1e-9 / HBM_PEAK_GBS, "work_per_step": step_bytes},
            "round2_module_path_ms_per_step": 9.39}


def bench_instant_dp(args, device, rank, world, dist):
    """BASELINE.json configs[3]: Instant-NGP training with the ray batch data-parallel over the ranks (weak scaling:
    16,384 rays per rank), gradients summed over RCCL/xGMI.  The tiny-MLP gradients go on the wire first, the 52 MB
    table gradient follows level group by level group while the next group's scatter runs (bf16 on the wire unless
    NERF_BENCH_REDUCE_FP32 is set); TV + clip + AdamW run replicated.  800 x 800 evaluation in row bands, gathered
    on rank 0."""
    import numpy as np
    import yaml
    from src.dataset import BlenderDataset, SYNTHETIC_CAMERA_ANGLE, synthetic_frames
    from project_nerf_amd import parallel
    from project_nerf_amd.engine import InstantNgpEngine
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "part2_instant.yaml.example")))
    batch, S, iters = 16384, 128, int(os.environ.get("NERF_BENCH_DP_ITERS", 600))      # the override shortens the rehearsal test
    cfg["train_iters"] = iters
    images, poses = synthetic_frames(12, 400, device, n_samples=128)          # every rank renders the same frames (same seed)
    ds = BlenderDataset.from_tensors(images, poses, SYNTHETIC_CAMERA_ANGLE)
    torch.manual_seed(1000 + rank)                                          # ... and draws its own rays
    eng = InstantNgpEngine(cfg, device=str(device), seed=0, world_size=world)
    wire = None if os.environ.get("NERF_BENCH_REDUCE_FP32") else torch.bfloat16

    ahead = []

    def draw():
        o, d, target = ds.sample_batch(batch, eng.bg)
        return o, d, target, eng.prepare_batch(o, d, S)

    def step():                                                             # one batch ahead, as in bench_instant
        if not ahead:
            ahead.append(draw())
        o, d, target, prepared = ahead.pop()
        ahead.append(draw())
        return eng.train_step(o, d, target, S, sync_grads_async=parallel.allreduce_sum_async, reduce_dtype=wire, prepared=prepared)

    for it in range(1, iters + 1):
        step()
        if it >= 256 and it % 64 == 0 and it < iters * 0.9:
            eng.update_grid()                                               # replicated: a pure function of the replicated weights
            ahead.clear()
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    dist.barrier(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    # replicas must still agree after hundreds of steps of summed gradients
    probe = torch.stack([eng.table[::65537].double().sum(), eng.net.double().sum()])
    lo, hi = probe.clone(), probe.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    # evaluation: row bands of one 800 x 800 view
    H = W = 800
    o, d, _ = BlenderDataset.from_tensors(images[:1, :1, :1].expand(1, H, W, 4).contiguous(), poses[:1], SYNTHETIC_CAMERA_ANGLE).get_image_rays(0, device)
    r0, r1 = parallel.shard_range(H, rank, world)
    eng.render_image(o[r0:r1].contiguous(), d[r0:r1].contiguous(), S)
    dist.barrier(); torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.render_frames):
        band = eng.render_image(o[r0:r1].contiguous(), d[r0:r1].contiguous(), S)
        parallel.gather_row_bands(band, H, dst=0)
    dist.barrier(); torch.cuda.synchronize()
    rt = (time.perf_counter() - t1) / args.render_frames
    if rank == 0:
        print(json.dumps({
            "metric": "train rays/sec + 800x800 render FPS, NeRF-Synthetic Lego; PSNR parity", "value": batch * world * args.steps / dt,
            "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": iters, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "Part 2 Instant-NGP, ray batches data-parallel (BASELINE configs[3])", "rays_per_gpu": batch,
                       "samples_per_ray": S, "parallelism": f"ray-dp{world}", "gradient_wire_dtype": "fp32" if wire is None else "bf16",
                       "table_gradient_bytes_fp32": int(eng.g_table.numel() * 4), "level_groups": eng.level_groups()},
            "final_loss": float(loss), "replica_divergence": float((hi - lo).abs().max()), "render_fps": 1.0 / rt}))
    dist.destroy_process_group()


def hemisphere_poses(n, seed):
    import numpy as np
    from src.dataset import look_at_pose
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        th, ph = rng.uniform(0, 2 * np.pi), rng.uniform(0.15, 1.2)
        out.append(torch.tensor(look_at_pose(4.0311 * np.array([np.cos(th) * np.cos(ph), np.sin(th) * np.cos(ph), np.sin(ph)])),
                                dtype=torch.float32))
    return torch.stack(out, 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--render-samples", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-render", action="store_true")
    ap.add_argument("--no-instant", action="store_true", help="skip the shortened Instant-NGP block of the default run")
    ap.add_argument("--render-frames", type=int, default=10)
    ap.add_argument("--frames", type=int, default=100, help="GPU-resident 800x800 training frames the batches are drawn from")
    ap.add_argument("--no-part4", action="store_true", help="skip the Part 4 block of the default run")
    ap.add_argument("--workload", choices=["vanilla", "instant", "part4"], default="vanilla",
                    help="vanilla = BASELINE.json configs[1] (default, the judged line); instant = configs[2] / configs[3]; "
                         "part4 = configs[4] on one GPU")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("NERF_BENCH_SINGLE_DEVICE"):      # rehearsal of the N>1 control flow on a 1-GPU box
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import project_nerf_amd  # noqa: F401
    from project_nerf_amd import ops, parallel
    from project_nerf_amd.engine import VanillaNerfEngine
    from src.dataset import BlenderDataset, SYNTHETIC_CAMERA_ANGLE
    if args.workload == "instant" and world == 1:
        return bench_instant(args, device, standalone=True)
    if args.workload == "part4":
        if world != 1:
            raise SystemExit("--workload part4 is a one-GPU measurement (its data-parallel path: run.py, tests/test_gpu_cli_dp.py)")
        print(json.dumps(bench_part4(args, device, steps=max(args.steps, 200))))
        return
    dist = None
    if world > 1:
        import torch.distributed as dist
        parallel.init_distributed("cuda")       # backend nccl == RCCL over xGMI
    if args.workload == "instant":
        return bench_instant_dp(args, device, rank, world, dist)

    eng = VanillaNerfEngine(seed=0, world_size=world, device=str(device))
    R, S = args.rays, args.samples
    # synthetic data of the benchmark's shape: `frames` RGBA frames of 800 x 800 resident in HBM (uniform noise:
    # the step's cost does not depend on the pixel values), cameras on the NeRF-Synthetic hemisphere
    torch.manual_seed(100 + rank)
    ds = BlenderDataset.from_tensors(torch.rand(args.frames, 800, 800, 4, device=device), hemisphere_poses(args.frames, 7).to(device),
                                     SYNTHETIC_CAMERA_ANGLE)
    # the flat 2.38 MB gradient is all-reduced in two ranges, the first while the second is still being
    # computed (NERF_BENCH_SYNC_ALLREDUCE=1: one blocking collective after the backward pass instead)
    blocking = os.environ.get("NERF_BENCH_SYNC_ALLREDUCE") is not None
    sync = parallel.allreduce_sum_ if world > 1 and blocking else None
    sync_async = parallel.allreduce_sum_async if world > 1 and not blocking else None
    if world > 1 and os.environ.get("NERF_BENCH_NATIVE_COMM") is not None:
        # A/B: the same two-range overlapped all-reduce through libnerf_comm.so (include/nerf_comm.h) instead of
        # torch.distributed's process group; the bootstrap token travels through the process group
        from project_nerf_amd._comm import NativeComm
        sync, sync_async = None, parallel.native_allreduce_sum_async(NativeComm.from_torch_distributed())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    counter = [0]

    def step(mark=None):
        # the data side of the step (run.py:314-322 + the jitter of render_rays): one kernel
        # every rank draws ITS shard of one global batch of world x R rays (same seed and counter, first_ray = rank x R):
        # the union over the ranks is the batch one GPU would draw with the global size
        o, d, target, z = ds.train_batch(R, S, eng.near, eng.far, eng.bg, seed=100, counter=counter[0], first_ray=rank * R)
        counter[0] += 1
        if mark is not None:
            mark("batch_sampling")
        return eng.train_step(o, d, target, S, sync_grads=sync, sync_grads_async=sync_async, mark=mark, z=z)

    def timed(n_steps):
        """exactly n_steps steps between barrier + synchronize on both sides; MAX over ranks"""
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            loss = step()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, loss

    ops._lib.set_option("stash_fp8", 0)          # the headline: bf16 training images, whatever the environment says
    for _ in range(args.warmup):
        step()
    dt, loss = timed(args.steps)
    ms_per_step = dt / args.steps * 1e3
    rays_per_s = R * world * args.steps / dt

    out = {
        "metric": "train rays/sec + 800x800 render FPS, NeRF-Synthetic Lego; PSNR parity",
        "value": rays_per_s, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "Part 2 vanilla NeRF train step (L10/L4 Fourier, 8x256 skip-4 MLP, view 128), batch sampling included",
                   "rays_per_gpu": R, "samples_per_ray": S, "global_rays": R * world,
                   "parallelism": f"ray-dp{world}", "weights": "random init (seed 0)",
                   "frames": f"{args.frames} x 800x800 RGBA resident in HBM (uniform noise)",
                   "training_images": "bf16 (layer inputs and pre-activation gradients); forward, dgrad and wgrad all on "
                                      "v_mfma_f32_32x32x16_bf16, fp32 accumulate everywhere",
                   "render": f"800x800 x {args.render_samples} samples/ray"},
        "final_loss": float(loss.item()),
    }

    # ---- every phase timed INSIDE the step: a HIP event on the launch stream after each phase.  EVERY rank runs
    # these steps (they contain the gradient all-reduce: a rank-0-only loop would wait for its peers for ever)
    phases = ["batch_sampling", "fwd", "loss", "dgrad", "wgrad", "adam+pack"]
    acc = {p: 0.0 for p in phases}
    reps = max(args.steps, 200)              # also keeps the GPU busy long enough for an outside utilisation sampler
    for _ in range(reps):
        evs = [("start", torch.cuda.Event(enable_timing=True))]
        evs[0][1].record()

        def mark(name):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append((name, e))
        step(mark)
        torch.cuda.synchronize()
        for (_, e0), (name, e1) in zip(evs[:-1], evs[1:]):
            acc[name] = acc.get(name, 0.0) + e0.elapsed_time(e1)
    barrier()

    # ---- labelled secondary figure: the same step on 8-bit training images (opt-in library option stash_fp8) ----
    ops._lib.set_option("stash_fp8", 1)
    for _ in range(max(args.warmup, 3)):
        step()
    n8 = max(args.steps, 200)
    dt8, _ = timed(n8)
    ops._lib.set_option("stash_fp8", 0)
    out["value_fp8_images"] = R * world * n8 / dt8
    out["ms_per_step_fp8_images"] = dt8 / n8 * 1e3
    out["fp8_images_note"] = {"dtype_wgrad": "e4m3 x e5m2 (v_mfma_scale_f32_32x32x64_f8f6f4, unit block scales)", "dtype_fwd_dgrad": "bf16",
                              "steps": n8, "option": "stash_fp8 (NERF_STASH_FP8=1); narrower than BASELINE configs[1]'s bf16: "
                                                     "reported beside the headline, not as it"}

    if rank == 0:
        n = R * S
        in_step = {p: acc[p] / reps for p in acc}
        out["kernels_in_step"] = {p: {"ms": v} for p, v in in_step.items()}
        out["kernels_in_step_note"] = ("HIP events on the launch stream between the phases of the timed step; "
                                       "fwd = mlp_fwd_stream_kernel<true, true>, loss = composite_mse_bwd_kernel, dgrad = mlp_bwd_stream_kernel<true>, "
                                       "wgrad = mlp_wgrad_kernel<false> + wgrad_reduce_kernel, batch_sampling = train_batch_kernel (pixel draws, rays, targets, jittered depths)")
        # ---- the same kernels launched back to back on their own (warm caches) ----
        o, d, target = ds.sample_batch(R, eng.bg)
        u = torch.rand(R, S, device=device)
        z = ops.sample_rays(o, d, 2.0, 6.0, S, u=u)
        stash = torch.empty(ops.mlp_stash_bytes(n), dtype=torch.uint8, device=device)
        rgb, sigma = ops.mlp_fwd(eng.packed, o, d, z, stash)
        scal = torch.zeros(2, device=device)
        d_rgb, d_sigma, _ = ops.composite_mse_bwd(rgb.view(R, S, 3), sigma.view(R, S), z, d, eng.bg, target, scal[0:1], amax_accum=scal[1:2])
        ws = torch.empty(ops.mlp_bwd_workspace_bytes(n), dtype=torch.uint8, device=device)
        grads = torch.empty_like(eng.params)
        lib = ops._lib.load()
        st = torch.cuda.current_stream().cuda_stream
        P = lambda t: t.data_ptr()
        k = {}
        k["mlp_fwd_train"] = event_ms(lambda: ops.mlp_fwd(eng.packed, o, d, z, stash), 20)
        k["mlp_fwd_infer"] = event_ms(lambda: ops.mlp_fwd(eng.packed, o, d, z), 20)
        k["mlp_bwd_dgrad"] = event_ms(lambda: lib.nerf_mlp_bwd_dgrad_ex(P(eng.packed), P(stash), P(rgb), P(sigma), P(d_rgb), P(d_sigma), n, P(ws), P(scal[1:2]), st), 20)
        k["mlp_bwd_wgrad"] = event_ms(lambda: lib.nerf_mlp_bwd_wgrad(P(stash), P(ws), n, P(grads), st), 20)
        k["composite_mse_bwd"] = event_ms(lambda: ops.composite_mse_bwd(rgb.view(R, S, 3), sigma.view(R, S), z, d, eng.bg, target, scal[0:1]), 20)
        tp, tm, tv = eng.params.clone(), torch.zeros_like(grads), torch.zeros_like(grads)
        k["adam+pack"] = event_ms(lambda: (ops.adam_step(tp, grads, tm, tv, 1, 5e-4), ops.mlp_pack(eng.params, eng.packed)), 20)
        k["batch_sampling"] = event_ms(lambda: ds.train_batch(R, S, eng.near, eng.far, eng.bg, seed=1, counter=3), 20)
        stash_b = ops.mlp_stash_bytes(n)
        image_bytes = 1 if stash_b < 4000 * n else 2      # 8-bit images (option stash_fp8) or bf16 (the default, what was timed)
        assert image_bytes == 2, "the headline step must run on bf16 training images"
        n_pad = (n + 255) // 256 * 256
        wgrad_bytes = WGRAD_ELEMS * image_bytes * n_pad
        kern = {
            "mlp_fwd_train": {"ms": k["mlp_fwd_train"], "tflops": n * FWD_FLOP / k["mlp_fwd_train"] * 1e-9},
            "mlp_fwd_infer": {"ms": k["mlp_fwd_infer"], "tflops": n * FWD_FLOP / k["mlp_fwd_infer"] * 1e-9},
            "mlp_bwd_dgrad": {"ms": k["mlp_bwd_dgrad"], "tflops": n * DGRAD_FLOP / k["mlp_bwd_dgrad"] * 1e-9},
            "mlp_bwd_wgrad": {"ms": k["mlp_bwd_wgrad"], "tflops": n * WGRAD_FLOP / k["mlp_bwd_wgrad"] * 1e-9,
                              "gbs": wgrad_bytes / k["mlp_bwd_wgrad"] * 1e-6},
            "composite_mse_bwd": {"ms": k["composite_mse_bwd"], "gbs": (n * 36 + R * 44) / k["composite_mse_bwd"] * 1e-6},
            "adam+pack": {"ms": k["adam+pack"]}, "batch_sampling": {"ms": k["batch_sampling"]},
        }
        out["kernels"] = kern
        # ---- rooflines: algorithmic work per launch / launch time.  `achieved` and `frac` come from the IN-STEP time
        # (what the rocprofv3 averages of profiles/ reproduce); the back-to-back figure is kept beside it.  PMC traffic
        # from the committed rocprofv3 --pmc summary of this same command (FETCH_SIZE doubled, separate passes).
        pmc, pmc_src = pmc_summary("r04_pmc_summary.json")
        out["traffic_source"] = pmc_src
        traffic = lambda name: (pmc.get(name, {}).get("hbm_bytes_per_launch_corrected") if (R, S) == (4096, 64) else None)

        def mfma_roof(kernel, phase, iso, work):
            a, b = work / in_step[phase] * 1e-9, work / k[iso] * 1e-9
            return {"bound": "mfma", "kernel": kernel, "achieved": a, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": a / MFMA_PEAK_TFLOPS,
                    "traffic": traffic(kernel), "work_per_launch": work, "launch_ms": in_step[phase],
                    "achieved_back_to_back": b, "launch_ms_back_to_back": k[iso]}
        # configs[1] is MFMA-bound (compulsory HBM bytes are negligible; the training images are design
        # traffic), so every decoder kernel is held against the dense bf16 MFMA peak.  The weight-gradient kernel's
        # HBM view (it reads every training image once) is kept beside it under its own key.
        roofs = {
            "mlp_fwd_train": mfma_roof("mlp_fwd_stream_kernel<true, true>", "fwd", "mlp_fwd_train", n * FWD_FLOP),
            "mlp_bwd_dgrad": mfma_roof("mlp_bwd_stream_kernel<true>", "dgrad", "mlp_bwd_dgrad", n * DGRAD_FLOP),
            "mlp_bwd_wgrad": mfma_roof("mlp_wgrad_kernel<false>", "wgrad", "mlp_bwd_wgrad", n * WGRAD_FLOP),
            "mlp_bwd_wgrad_hbm_view": {"bound": "hbm", "kernel": "mlp_wgrad_kernel<false>", "achieved": wgrad_bytes / in_step["wgrad"] * 1e-6,
                                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": wgrad_bytes / in_step["wgrad"] * 1e-6 / HBM_PEAK_GBS,
                                       "traffic": traffic("mlp_wgrad_kernel<false>"), "work_per_launch": wgrad_bytes, "launch_ms": in_step["wgrad"],
                                       "achieved_back_to_back": wgrad_bytes / k["mlp_bwd_wgrad"] * 1e-6, "launch_ms_back_to_back": k["mlp_bwd_wgrad"],
                                       "note": f"design traffic, not compulsory bytes: reads every bf16 training image once ({WGRAD_ELEMS * 2} B/sample)"},
        }
        # `roofline` = the dominant kernel of the step (longest in-step time) against SURVEY 8(d)'s bound for this config
        dom = max((name for name in roofs if roofs[name]["bound"] == "mfma"), key=lambda name: roofs[name]["launch_ms"])
        out["roofline"] = dict(roofs[dom], note="dominant kernel of the timed step by in-step time (HIP events on the launch stream); "
                               "algorithmic FLOPs of SURVEY 8(d) / launch time against the dense bf16 MFMA peak")
        out["rooflines"] = roofs
        step_ach = R * S * TRAIN_FLOP / (ms_per_step * 1e-3) * 1e-12
        out["step_tflops"] = step_ach
        out["step_frac_of_mfma_peak"] = step_ach / MFMA_PEAK_TFLOPS
        # configs[1] is bound by the MFMA roof (negligible compulsory bytes): the step against it
        out["step_frac_of_binding_roofline"] = {"bound": "mfma", "achieved": step_ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                                "frac": step_ach / MFMA_PEAK_TFLOPS, "work_per_step": R * S * TRAIN_FLOP}
        step_traffic = [traffic(roofs[r]["kernel"]) for r in ("mlp_fwd_train", "mlp_bwd_dgrad", "mlp_bwd_wgrad")]
        # the small kernels of the step, where the committed PMC summary has them (the partial-tile reduce of wgrad,
        # compositing + loss, batch sampling, Adam, weight repack)
        extras = [traffic(k_) for k_ in ("wgrad_reduce_kernel", "composite_mse_bwd_kernel<1>", "train_batch_kernel", "adam_kernel", "pack_kernel")]
        out["step_hbm_traffic_bytes"] = (sum(step_traffic) + sum(t for t in extras if t is not None)
                                         if all(t is not None for t in step_traffic) else None)

    # ---- render: 800x800 rays, 128 samples, rays split into row bands across ranks ----
    if not args.no_render:
        H = W = 800
        rows = H // world
        ro, rd, _ = synth_rays(rows * W, 7 + rank, device)
        eng.render_image(ro, rd, args.render_samples)     # warm-up
        barrier()
        t0 = time.perf_counter()
        frames = args.render_frames
        for _ in range(frames):
            img = eng.render_image(ro, rd, args.render_samples)
            if world > 1:
                bands = [torch.empty_like(img) for _ in range(world)] if rank == 0 else None
                dist.gather(img, bands, dst=0)
        barrier()
        rt = (time.perf_counter() - t0) / frames
        if world > 1:
            t = torch.tensor([rt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            rt = float(t.item())
        if world == 1:
            # opt-in extension: "64 coarse + 128 fine" of BASELINE.json (no reference counterpart):
            # 64-sample coarse pass, inverse-CDF resampling, 192-sample fine pass with the same field
            eng.render_image(ro, rd, 64, n_fine=128)
            torch.cuda.synchronize()
            th = time.perf_counter()
            eng.render_image(ro, rd, 64, n_fine=128)
            torch.cuda.synchronize()
            out["render_fps_64coarse_128fine"] = 1.0 / (time.perf_counter() - th)
        out["render_fps"] = 1.0 / rt
        out["render_ms_per_frame"] = rt * 1e3
        out["render_tflops"] = H * W * args.render_samples * FWD_FLOP / rt * 1e-12
        out["render_frac_of_mfma_peak"] = out["render_tflops"] / MFMA_PEAK_TFLOPS

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        # the full 4096-ray batch of the workload: ~10-20 s of CPU work on the box's 16-core share for one GPU
        out["cpu_baseline"] = cpu_baseline(R, S, min(cores, 16))
        out["gpu_over_cpu"] = rays_per_s / out["cpu_baseline"]["value"]

    if rank == 0 and world == 1 and not args.no_instant:
        del ds
        torch.cuda.empty_cache()
        try:                      # the side blocks must never cost the run its headline line
            inst = bench_instant(args, device)
            out["instant"] = {key: inst[key] for key in ("value", "unit", "ms_per_step", "render_fps", "render_ms_per_frame", "psnr_curve",
                                                         "kernels", "rooflines", "traffic_source", "active_samples", "config", "reference_headline")}
        except Exception as e:    # noqa: BLE001
            out["instant"] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0 and world == 1 and not args.no_part4:
        torch.cuda.empty_cache()
        try:
            out["part4"] = bench_part4(args, device)
        except Exception as e:    # noqa: BLE001
            out["part4"] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
