"""Part 3 / Part 4 training and evaluation loop (modes part3, part4; reference run.py:903-1559 run_part3 and
run.py:1562-2331 run_part4): DynamicDataset,
NeuralField('part4'), render_rays with per-ray times, AdamW + cosine LR, occupancy grid refreshed at the three
time anchors, best-on-validation checkpoints, and the loss terms of the reference's loop
(``part4_regularisers``: displacement magnitude, total variation on the deformation grids and on the canonical
grid, temporal smoothness, unsupervised consistency, tri-grid anchor; run.py:1832-1938) with its YAML keys,
defaults and every-n-steps schedule.  Part 4 at the reference's example shapes trains on part4.DualHashEngine (fused HIP
chains, no torch autograd / optimiser / library GEMM in the loop); other shapes and Part 3 compose the field from the
stand-alone operators (hash encodings and Fourier codes in HIP, the small MLPs as library GEMMs) under torch autograd."""
import os

import numpy as np
import torch


def _total_variation(params):
    """mean |p[i+1] - p[i]| over the flat parameter vector (run.py:1849-1851, 1857-1859)"""
    return torch.mean(torch.abs(params[1:] - params[:-1]))


def part4_regularisers(model, cfg, step, mean_delta_x, generator=None, probes=None):
    """The non-photometric loss terms of the reference's Part 4 loop (run.py:1835-1938) as a dict of scalars; the
    caller adds them to the RGB loss.  Same YAML keys, defaults, sample counts and compensation factors (a term
    evaluated every k-th step is weighted by k).  ``generator``: optional torch.Generator for the random probes;
    ``probes``: the probe points themselves (temporal_x/_t, unsup_x/_t, anchor_x) instead of fresh draws."""
    device = mean_delta_x.device
    zero = torch.zeros((), device=device)
    bound = float(cfg.get("scene_bound", 1.5))
    warm = cfg.get("grid_warmup_iters", 256)
    rand = lambda *shape: torch.rand(*shape, device=device, generator=generator)
    terms = {"reg": torch.mean(mean_delta_x ** 2) * float(cfg.get("deformation_reg_weight", 0.01))}
    # total variation: the three displacement grids (averaged), then the canonical grid
    terms["tv_disp"] = zero
    if cfg.get("use_tv_displacement", True):
        grids = [getattr(model, n) for n in ("deform_grid_start", "deform_grid_mid", "deform_grid_end") if hasattr(model, n)]
        tv = sum(_total_variation(g.encoding.params) for g in grids)
        terms["tv_disp"] = tv * float(cfg.get("tv_displacement_weight", 0.001)) / 3.0
    terms["tv_canon"] = zero
    tv_w = float(cfg.get("tv_loss_weight", 1e-5))
    if tv_w > 0 and hasattr(model, "canonical_repr"):
        terms["tv_canon"] = _total_variation(model.canonical_repr.encoding.params) * tv_w

    def displacement(grid, x, t):
        return model.deform_decoder(grid(x), model.time_modulation(model.time_encoder(t)))

    # temporal smoothness: the displacement of a point changes little over eps (every 16th step, 64 probes)
    terms["temporal"] = zero
    if cfg.get("use_temporal_smooth", True) and step > warm and step % 16 == 0:
        eps = float(cfg.get("temporal_epsilon", 0.02))
        x = probes["temporal_x"] if probes else (rand(64, 3) * 2 - 1) * bound
        t = probes["temporal_t"] if probes else rand(64, 1) * (1.0 - eps)
        feat = model.deformation_grid(x)
        d0 = model.deform_decoder(feat, model.time_modulation(model.time_encoder(t)))
        d1 = model.deform_decoder(feat, model.time_modulation(model.time_encoder(t + eps)))
        terms["temporal"] = torch.mean((d0 - d1) ** 2) * float(cfg.get("temporal_smooth_weight", 1e-4)) * 16
    # unsupervised consistency: no net drift of the whole volume (every 32nd step, 128 probes)
    terms["unsup"] = zero
    if cfg.get("use_unsupervised_consistency", False) and step > warm and step % 32 == 0:
        t = probes["unsup_t"] if probes else rand(128, 1)
        x = probes["unsup_x"] if probes else (rand(128, 3) * 2 - 1) * bound
        d = displacement(model.deformation_grid, x, t)
        terms["unsup"] = torch.mean(torch.abs(d.mean(dim=0))) * float(cfg.get("unsup_consistency_weight", 0.001)) * 32
    # tri-grid anchor: zero displacement at t = 0 on the start grid, start and mid grids agree at t = 1/6
    terms["anchor"] = zero
    if cfg.get("use_static_anchor", True) and step > warm and step % 16 == 0:
        x = probes["anchor_x"] if probes else (rand(128, 3) * 2 - 1) * bound
        at0 = displacement(model.deform_grid_start, x, torch.zeros(128, 1, device=device))
        t6 = torch.full((128, 1), 1.0 / 6.0, device=device)
        d_start, d_mid = displacement(model.deform_grid_start, x, t6), displacement(model.deform_grid_mid, x, t6)
        consistency = torch.mean((d_start - d_mid) ** 2) * 0.1
        terms["anchor"] = (torch.mean(at0 ** 2) + consistency) * float(cfg.get("static_anchor_weight", 0.01)) * 16
    return terms


def part3_regularisers(model, cfg, step, mean_delta_x, generator=None, probes=None):
    """The non-photometric loss terms of the reference's Part 3 loop (run.py:1107-1165): displacement magnitude,
    total variation on a hash-grid canonical field, temporal smoothness (every 2nd step), unsupervised consistency
    (every 4th step) -- same YAML keys, defaults and compensation factors."""
    device = mean_delta_x.device
    zero = torch.zeros((), device=device)
    bound = float(cfg.get("scene_bound", 1.2))
    warm = cfg.get("grid_warmup_iters", 256)
    rand = lambda *shape: torch.rand(*shape, device=device, generator=generator)
    terms = {"reg": torch.mean(mean_delta_x ** 2) * float(cfg.get("deformation_reg_weight", 1e-4)), "tv": zero, "temporal": zero,
             "unsup": zero}
    instant = cfg.get("canonical_type", "nerf") == "instant"
    if cfg.get("use_tv_loss", True) and instant and hasattr(model, "canonical_repr") and hasattr(model.canonical_repr, "encoding"):
        terms["tv"] = _total_variation(model.canonical_repr.encoding.params) * float(cfg.get("tv_loss_weight", 1e-6))
    if getattr(model, "direct_time_conditioning", False):
        return terms                                       # no deformation field to regularise
    displacement = lambda x, t: model.deform_net(model.pos_encoder_for_deform(x), model.time_encoder(t))
    if cfg.get("use_temporal_smooth", True) and step > warm and step % 2 == 0:
        eps, n = float(cfg.get("temporal_epsilon", 0.02)), int(cfg.get("temporal_n_samples", 256))
        x = probes["temporal_x"] if probes else (rand(n, 3) * 2 - 1) * bound
        t = probes["temporal_t"] if probes else rand(n, 1) * (1.0 - eps)
        feat = model.pos_encoder_for_deform(x)
        d0, d1 = model.deform_net(feat, model.time_encoder(t)), model.deform_net(feat, model.time_encoder(t + eps))
        terms["temporal"] = torch.mean((d0 - d1) ** 2) * float(cfg.get("temporal_smooth_weight", 1e-4)) * 2
    if cfg.get("use_unsupervised_consistency", False) and step > warm and step % 4 == 0:
        n = min(int(cfg.get("unsup_n_samples", 512)), 512)
        t = probes["unsup_t"] if probes else rand(n, 1)
        x = probes["unsup_x"] if probes else (rand(n, 3) * 2 - 1) * bound
        terms["unsup"] = torch.mean(torch.abs(displacement(x, t).mean(dim=0))) * float(cfg.get("unsup_consistency_weight", 0.001)) * 4
    return terms


def part4_param_groups(model, lr):
    """AdamW parameter groups of the reference's Part 4 loop (run.py:1684-1738): 2x the base rate for the three
    deformation grids and the canonical grid, 5x for ``deform_decoder.displacement_scale``, 1x for the deformation MLP
    and everything else (time modulation, canonical decoder)."""
    groups = []
    for name in ("deform_grid_start", "deform_grid_mid", "deform_grid_end"):
        if hasattr(model, name):
            groups.append({"params": list(getattr(model, name).parameters()), "lr": lr * 2.0, "name": name})
    if not hasattr(model, "deform_grid_start") and hasattr(model, "deformation_grid"):
        groups.append({"params": list(model.deformation_grid.parameters()), "lr": lr * 2.0, "name": "deformation_grid"})
    if hasattr(model, "canonical_repr"):
        groups.append({"params": list(model.canonical_repr.parameters()), "lr": lr * 2.0, "name": "canonical_repr"})
    if hasattr(model, "deform_decoder"):
        groups.append({"params": [model.deform_decoder.displacement_scale], "lr": lr * 5.0, "name": "displacement_scale"})
        groups.append({"params": [p for n, p in model.deform_decoder.named_parameters() if "displacement_scale" not in n], "lr": lr,
                       "name": "deform_decoder"})
    excluded = ("deform_grid_start", "deform_grid_mid", "deform_grid_end", "deformation_grid", "canonical_repr", "deform_decoder")
    others = [p for n, p in model.named_parameters() if not any(ex in n for ex in excluded)]
    if others:
        groups.append({"params": others, "lr": lr, "name": "others"})
    return groups


def part4_probe_draws(cfg, step, device, generator=None):
    """The random probe points of the reference's every-n-steps regularisers (run.py:1861-1938) in DualHashEngine's form, or
    None on the steps that evaluate none of them: 64 temporal probes every 16th step, 128 anchor probes every 16th step,
    128 consistency probes every 32nd step, all after the occupancy-grid warm-up."""
    warm = cfg.get("grid_warmup_iters", 256)
    if step <= warm or step % 16:
        return None
    bound = float(cfg.get("scene_bound", 1.5))
    rand = lambda *shape: torch.rand(*shape, device=device, generator=generator)
    probes = {}
    if cfg.get("use_temporal_smooth", True):
        eps = float(cfg.get("temporal_epsilon", 0.02))
        probes["temporal"] = ((rand(64, 3) * 2 - 1) * bound, rand(64, 1) * (1.0 - eps), eps, float(cfg.get("temporal_smooth_weight", 1e-4)))
    if cfg.get("use_unsupervised_consistency", False) and step % 32 == 0:
        t = rand(128, 1)
        probes["unsup"] = ((rand(128, 3) * 2 - 1) * bound, t, float(cfg.get("unsup_consistency_weight", 0.001)))
    if cfg.get("use_static_anchor", True):
        probes["anchor"] = ((rand(128, 3) * 2 - 1) * bound, float(cfg.get("static_anchor_weight", 0.01)))
    return probes or None


def run_dynamic(cfg, args):
    """Part 3 / Part 4 loop.  Data parallelism as in run.py's module docstring: every rank forms its shard of one global
    batch (same torch seed on every rank), gradients are summed over RCCL and averaged, the global-norm clip follows the
    all-reduce, evaluation renders row bands gathered on rank 0."""
    from . import parallel
    from .core import NeuralField
    from .dataset import DynamicDataset
    from .renderer import DensityGrid, render_rays
    from .utils import compute_psnr, compute_psnr_torch
    if not args.data_dir:
        raise ValueError("Part 3 / Part 4 require --data_dir pointing to a D-NeRF dataset root.")
    if not torch.cuda.is_available():
        raise RuntimeError("the NeRF hot path runs on a HIP device only (no CPU fallback)")
    rank, world = parallel.rank_world()
    main_rank = rank == 0
    say = print if main_rank else (lambda *a, **k: None)
    device = torch.device("cuda", torch.cuda.current_device())
    downscale, white_bkgd = cfg.get("downscale", 2), cfg.get("white_bkgd", True)
    near, far = float(cfg.get("near", 2.0)), float(cfg.get("far", 6.0))
    n_samples = cfg.get("n_samples", 64)
    render_n = cfg.get("render_n_samples", n_samples)
    batch, iters, lr = cfg.get("batch_size", 4096), cfg.get("train_iters", 5000), cfg.get("learning_rate", 0.01)
    chunk = args.render_chunk or cfg.get("chunk", 16384)
    log_every = cfg.get("log_every", 50)
    part3 = cfg.get("mode") == "part3"
    log_dir = os.path.join(cfg.get("log_dir", "output/part3" if part3 else "output/part4"), os.path.basename(args.data_dir.rstrip("/")))
    os.makedirs(log_dir, exist_ok=True)
    train_set = DynamicDataset(args.data_dir, "train", downscale, white_bkgd, cfg.get("scene_scale", 1.0)).to(device)
    split = "test" if os.path.exists(os.path.join(args.data_dir, "transforms_test.json")) else "train"
    test_set = DynamicDataset(args.data_dir, split, downscale, white_bkgd, cfg.get("scene_scale", 1.0))
    model = NeuralField(cfg).to(device)
    grid = None
    # Part 3 builds the occupancy grid only around a hash-grid canonical field (run.py:985-1000), Part 4 always
    # (run.py:1648-1661); resolution default 128 in both
    if cfg.get("use_density_grid", True) and (not part3 or cfg.get("canonical_type", "nerf") == "instant"):
        grid = DensityGrid(cfg.get("grid_resolution", 128), cfg.get("scene_bound", 1.5), cfg.get("grid_threshold", 0.01)).to(device)
    if args.checkpoint:
        ckpt = torch.load(args.checkpoint, map_location=device)
        model.load_state_dict(ckpt["model_state_dict"])
        if grid is not None and "density_grid" in ckpt:
            grid.load_state_dict(ckpt["density_grid"])
    bg = torch.ones(3, device=device) if white_bkgd else torch.zeros(3, device=device)
    eval_bg = bg                                    # validation / test always composite onto the dataset's background

    def render_band(o, d, t):
        rows, width = o.shape[0], o.shape[1]
        o, d = o.reshape(-1, 3), d.reshape(-1, 3)
        if o.shape[0] == 0:
            return o.new_zeros(0, width, 3)
        pred = torch.cat([render_rays(model, o[i:i + chunk], d[i:i + chunk], near, far, render_n, False, density_grid=grid,
                                      times=t.expand(min(chunk, o.shape[0] - i), 1), bg_color=eval_bg)[0]
                          for i in range(0, o.shape[0], chunk)], 0)
        return pred.view(rows, width, 3)

    def evaluate(indices):
        """every rank renders a row band of each view, rank 0 gathers and scores (valid on rank 0)"""
        model.eval()
        vals = []
        with torch.no_grad():
            for idx in indices:
                o, d, tgt, t = test_set.get_image_rays(idx, device)
                pred = parallel.render_row_bands(lambda ob, db: render_band(ob, db, t), o, d)
                if main_rank:
                    vals.append(compute_psnr_torch(pred.reshape(-1, 3).clamp(0, 1), tgt.reshape(-1, 3)))
        model.train()
        return float(np.mean(vals)) if vals else 0.0

    def save_best(step, value):
        save = {"model_state_dict": model.state_dict(), "config": cfg, "step": step, "val_psnr": value}
        if grid is not None:
            save["density_grid"] = grid.state_dict()
        torch.save(save, os.path.join(log_dir, "best_model.pth"))

    best = 0.0
    local = parallel.check_global_batch(batch, world)       # this rank's shard [lo, hi) of the step's global batch
    lo = rank * local
    warm, stop, decay = cfg.get("grid_warmup_iters", 256), cfg.get("grid_stop_ratio", 0.9), cfg.get("grid_decay", 0.95)
    # random-background augmentation (run.py:1043-1044, 1771-1772): a fresh colour per step for target AND render
    random_bg_start = cfg.get("random_bg_start", 0) if cfg.get("use_random_bg", False) else float("inf")
    from . import part4 as p4
    use_engine = (not args.eval_only and not part3 and cfg.get("engine", True) and grid is not None
                  and getattr(model, "_p4_fused", False) and p4.supported(cfg) is None)
    if world > 1 and not args.eval_only:
        say(f">>> data parallel: {world} ranks x {local} rays (global batch {local * world}); clip after the all-reduce")
    if use_engine:
        # The example shapes train on the flat-parameter engine (part4.DualHashEngine: fused chains, fused compositing + loss
        # + regulariser + backward, one global-norm clip + AdamW with the reference's group rates, no torch autograd / optimiser
        # / library GEMM in the loop); weights and occupancy grid are copied into the NeuralField / DensityGrid for
        # validation, checkpoints and evaluation.  `engine: false` in the YAML or another shape: the module path below.
        from . import ops
        eng = p4.DualHashEngine({**cfg, "train_iters": iters, "learning_rate": lr, "grid_resolution": grid.resolution,
                                 "grid_threshold": grid.threshold, "scene_bound": grid.bound}, device=str(device),
                                seed=int(cfg.get("seed", 0) or 0), world_size=world)
        eng.load_from_model(model)
        with torch.no_grad():
            eng.grid.copy_(grid.grid)
            eng.binary_grid.copy_(grid.binary_grid)
        # replicas start from rank 0's values whatever the seeds did (then stay equal: identical all-reduced gradients, ONE
        # squared norm summed in a fixed order, replicated occupancy-grid updates)
        parallel.broadcast_([eng.tables, eng.net, eng.grid, eng.binary_grid])
        eng.repack()
        sync_async = parallel.allreduce_sum_async if world > 1 else None
        if world > 1 and cfg.get("dp_sharded_optimizer", True):
            # reduce-scatter of the flat table gradient, every rank steps its 1/N slice of the four grids (TV + ONE
            # squared norm + AdamW), all-gather of the fp16 copies the forward reads (project-nerf_amd/sharded.py)
            eng.enable_sharded_optimizer(rank)
            sync_async = None
            say(f">>> sharded optimiser: every rank steps {eng.shard.per} of {eng.tables.numel()} table parameters")
        pixels = train_set.H * train_set.W

        def sync():
            eng.gather_master()                             # sharded optimiser: the other ranks' slices of the fp32 master
            eng.copy_to_model(model)
            grid.grid, grid.binary_grid = eng.grid.clone(), eng.binary_grid.clone()

        def draw(step):
            # one uniform draw over all pixels of all frames (the same draw on every rank), this rank's shard of it; the time
            # stamp of a ray is its frame's
            idx = torch.randint(0, len(train_set) * pixels, (local * world,), device=device)[lo:lo + local].contiguous()
            step_bg = torch.rand(3, device=device) if step >= random_bg_start else eval_bg
            o, d, target, _ = ops.gather_batch(train_set.rgba, train_set.poses, idx, train_set.focal, train_set.scene_scale, bg=step_bg)
            t = train_set.times[idx // pixels].view(-1, 1)
            return o, d, target, t, step_bg, eng.prepare_batch(o, d, n_samples, first_ray=lo)

        active, ahead = 1.0, []
        for step in range(1, iters + 1):
            if not ahead:
                ahead.append(draw(step))
            o, d, target, t, step_bg, prepared = ahead.pop()
            if step < iters:
                ahead.append(draw(step + 1))                # compaction of the next batch queued ahead of this step's kernels
            loss_rgb = eng.train_step(o, d, target, t, n_samples, prepared=prepared, first_ray=lo, bg=step_bg,
                                      sync_grads_async=sync_async, probes=part4_probe_draws(cfg, step, device))
            if step < iters * stop:
                interval = 32 if step < iters * 0.1 else (128 if step < iters * 0.5 else 512)
                if step >= warm and step % interval == 0:
                    active = eng.update_grid(decay=decay)   # three time anchors, running maximum (replicated)
                    ahead.clear()                           # the waiting batch was compacted against the previous grid
            if step % log_every == 0:
                loss_val = parallel.mean_over_ranks(loss_rgb).item()
                say(f">>> Step {step}/{iters} | Loss {loss_val:.6f} | PSNR {compute_psnr(loss_val):.2f} dB | LR {eng.lr():.6f}"
                    f" | Skip: {(1 - active) * 100:.1f}%")
            if step % cfg.get("val_every", 500) == 0 or step == iters:
                sync()
                v = evaluate(range(min(len(test_set), cfg.get("val_views", 4))))
                say(f"    [Validation] PSNR: {v:.2f} dB")
                if v > best and main_rank:
                    best = v
                    save_best(step, best)
        if world > 1:
            eng.gather_master()
            say(f">>> replica divergence after {iters} steps: {parallel.replica_divergence([eng.tables, eng.net, eng.binary_grid]):.3e}")
        sync()
    elif not args.eval_only:
        # Part 3: one group (run.py:1016); Part 4: the reference's per-group learning rates (run.py:1684-1738)
        parallel.broadcast_([p.data for p in model.parameters()] + ([grid.grid, grid.binary_grid] if grid is not None else []))
        opt = torch.optim.AdamW(model.parameters() if part3 else part4_param_groups(model, lr), lr=lr,
                                weight_decay=cfg.get("weight_decay", 1e-5))
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=iters, eta_min=cfg.get("eta_min", 1e-4))
        active = 1.0
        model.train()
        for step in range(1, iters + 1):
            o, d, rgba, t = train_set.sample_random_rays(local * world, device)         # the same draw on every rank
            o, d, rgba, t = (x[lo:lo + local].contiguous() for x in (o, d, rgba, t))
            bg = torch.rand(3, device=device) if step >= random_bg_start else eval_bg
            target = rgba[:, :3] * rgba[:, 3:4] + bg * (1 - rgba[:, 3:4])
            pred, _, _, extras = render_rays(model, o, d, near, far, n_samples, True, density_grid=grid, times=t, bg_color=bg)
            loss_rgb = torch.nn.functional.mse_loss(pred, target)
            regs = part3_regularisers if part3 else part4_regularisers
            loss = loss_rgb + sum(regs(model, cfg, step, extras["mean_delta_x"]).values())
            opt.zero_grad()
            loss.backward()
            parallel.allreduce_mean_grads_(list(model.parameters()))                    # clip AFTER the all-reduce
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=float(cfg.get("max_grad_norm", 1.0)))
            opt.step()
            sched.step()
            if grid is not None and part3:
                # run.py:1191-1222: every 16 / 64 / 256 steps the union over 16 (later 8) times across the sequence, no decay
                interval = 16 if step < iters * 0.1 else (64 if step < iters * 0.5 else 256)
                if grid.should_update(step, interval, warm):
                    model.eval()
                    for t_val in torch.linspace(float(train_set.times.min()), float(train_set.times.max()), 16 if step < 1000 else 8):
                        active = grid.update(model, device=device, time=t_val.view(1, 1), decay=1.0)
                    model.train()
            elif grid is not None and step < iters * stop:
                interval = 32 if step < iters * 0.1 else (128 if step < iters * 0.5 else 512)
                if grid.should_update(step, interval, warm):
                    model.eval()
                    active = grid.update(model, device=device, decay=decay)      # three time anchors, running maximum
                    model.train()
            if step % log_every == 0:
                say(f">>> Step {step}/{iters} | Loss {parallel.mean_over_ranks(loss).item():.6f} | PSNR "
                    f"{compute_psnr(parallel.mean_over_ranks(loss_rgb).item()):.2f} dB | Skip: {(1 - active) * 100:.1f}%")
            if step % cfg.get("val_every", 500) == 0 or step == iters:
                v = evaluate(range(min(len(test_set), cfg.get("val_views", 4))))
                say(f"    [Validation] PSNR: {v:.2f} dB")
                if v > best and main_rank:
                    best = v
                    save_best(step, best)
    n_eval = len(test_set) if args.render_n in (None, -1) else min(args.render_n, len(test_set))
    avg = evaluate(range(n_eval))
    say(f">>> Test PSNR: {avg:.2f} dB (best validation {best:.2f} dB)")
    return avg
