"""Instant-NGP variant (mode part2_instant): hash-grid representation, tiny-MLP decoder and the
training loop of reference run.py:396-900, on the HIP kernels.

The reference delegates both operators to tinycudann (absent, unpinned); the classes below keep
its attribute surface -- ``representation.encoding.params`` (flat fp32, read by the TV
regulariser, run.py:614), ``decoder.sigma_net.params``, ``decoder.color_net.params`` -- so that
state_dict keys match, while sizes/layouts are this build's definition (see include/nerf_hip.h)."""
import os

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .decoders import InstantNeRFDecoder
from .embeddings import FourierRepresentation, HashRepresentation


def build_instant_field(field, config):
    """NeuralField.__init__ for mode part2_instant (reference src/core.py:57-77)."""
    field.representation = HashRepresentation(
        n_levels=config.get("n_levels", 16), n_features_per_level=config.get("n_features_per_level", 2),
        log2_hashmap_size=config.get("log2_hashmap_size", 19), base_resolution=config.get("base_resolution", 16),
        per_level_scale=config.get("per_level_scale", 1.5), bound=config.get("scene_bound", 1.0))
    field.dir_representation = FourierRepresentation(input_dim=3, L=config.get("L_embed_dir", 4), use_encoding=True)
    field.decoder = InstantNeRFDecoder(pos_dim=field.representation.out_dim, dir_dim=field.dir_representation.out_dim,
                                       hidden_dim=config.get("hidden_dim", 64))

    def _instant_forward(x, d):
        rep, dec = field.representation, field.decoder
        rgb, sigma = ops.instant_field(rep.table(), dec.flat_parameters(), dec.packed_weights(), x, d, rep.levels, rep.bound)
        return rgb, sigma.unsqueeze(-1)
    field._instant_forward = _instant_forward


def run_instant(cfg, args):
    """Training / evaluation loop of reference run_part2_instant (run.py:396-900): AdamW + cosine LR,
    TV-L1 on the flat hash table, per-group grad clipping, occupancy grid refreshed every
    32 / 128 / 512 steps after the warm-up, best-on-validation checkpoints."""
    import random
    from .core import NeuralField
    from .dataset import BlenderDataset
    from .renderer import DensityGrid, render_rays
    from .utils import compute_psnr, compute_psnr_torch
    if not args.data_dir:
        raise ValueError("Part 2 Instant requires --data_dir pointing to a NeRF dataset root.")
    if not torch.cuda.is_available():
        raise RuntimeError("the NeRF hot path runs on a HIP device only (no CPU fallback)")
    from . import parallel
    rank, world = parallel.rank_world()              # data parallelism: see run.py's module docstring
    main_rank = rank == 0
    say = print if main_rank else (lambda *a, **k: None)
    device = torch.device("cuda", torch.cuda.current_device())
    downscale, white_bkgd = cfg.get("downscale", 2), cfg.get("white_bkgd", True)
    near, far = float(cfg.get("near", 2.0)), float(cfg.get("far", 6.0))
    n_samples = cfg.get("n_samples", 32)
    render_n = cfg.get("render_n_samples", n_samples)
    batch, iters, lr = cfg.get("batch_size", 8192), cfg.get("train_iters", 5000), cfg.get("learning_rate", 0.01)
    log_every, chunk = cfg.get("log_every", 50), args.render_chunk or cfg.get("chunk", 16384)
    log_dir = os.path.join(cfg.get("log_dir", "output/part2_instant"), os.path.basename(args.data_dir.rstrip("/")))
    os.makedirs(log_dir, exist_ok=True)
    train_set = BlenderDataset(args.data_dir, "train", downscale, white_bkgd, cfg.get("scene_scale", 1.0)).to(device)
    split = "test" if os.path.exists(os.path.join(args.data_dir, "transforms_test.json")) else "val"
    test_set = BlenderDataset(args.data_dir, split, downscale, white_bkgd, cfg.get("scene_scale", 1.0))
    model = NeuralField(cfg).to(device)
    grid = None
    if cfg.get("use_density_grid", True):
        grid = DensityGrid(cfg.get("grid_resolution", 128), cfg.get("scene_bound", 1.5), cfg.get("grid_threshold", 0.01)).to(device)
    if args.checkpoint:
        ckpt = torch.load(args.checkpoint, map_location=device)
        model.load_state_dict(ckpt["model_state_dict"])
        if grid is not None and "density_grid" in ckpt:
            grid.load_state_dict(ckpt["density_grid"])

    def render_band(o, d):
        rows, width = o.shape[0], o.shape[1]
        o, d = o.reshape(-1, 3), d.reshape(-1, 3)
        if o.shape[0] == 0:
            return o.new_zeros(0, width, 3)
        pred = torch.cat([render_rays(model, o[i:i + chunk], d[i:i + chunk], near, far, render_n, False,
                                      white_bkgd=white_bkgd, density_grid=grid)[0]
                          for i in range(0, o.shape[0], chunk)], 0)
        return pred.view(rows, width, 3)

    def evaluate(ds, indices):
        """PSNR over the given views; under data parallelism every rank renders a row band of each view, rank 0 gathers
        (every rank must call this with the same indices; the value is valid on rank 0)"""
        model.eval()
        out = []
        with torch.no_grad():
            for idx in indices:
                o, d, tgt = ds.get_image_rays(idx, device)
                pred = parallel.render_row_bands(render_band, o, d)
                if main_rank:
                    out.append(compute_psnr_torch(pred.reshape(-1, 3).clamp(0, 1), tgt.reshape(-1, 3)))
        model.train()
        return float(np.mean(out)) if out else 0.0

    best = 0.0
    # The reference's default shape (16 levels x 2 features, 64 hidden units, occupancy grid) trains on the flat-parameter
    # engine bench.py times (fused compositing + loss + backward, binned hash backward, fused TV + clip + AdamW, compaction
    # one batch ahead); weights and occupancy grid are copied into the NeuralField / DensityGrid for validation, checkpoints
    # and evaluation.  `engine: false` in the YAML or another shape: NeuralField + torch.optim.AdamW below.
    use_engine = (not args.eval_only and cfg.get("engine", True) and grid is not None and model.decoder.fused
                  and cfg.get("n_levels", 16) == 16 and cfg.get("n_features_per_level", 2) == 2)
    if use_engine:
        from .engine import InstantNgpEngine
        eng = InstantNgpEngine({**cfg, "scene_bound": cfg.get("scene_bound", 1.5), "grid_threshold": grid.threshold,
                                "grid_resolution": grid.resolution, "train_iters": iters, "learning_rate": lr}, device=str(device),
                               seed=int(cfg.get("seed", 0) or 0), world_size=world)
        local = parallel.check_global_batch(batch, world)    # this rank's shard [lo, hi) of the step's global batch
        lo, hi = rank * local, (rank + 1) * local
        sync_async = parallel.allreduce_sum_async if world > 1 else None
        wire = torch.bfloat16 if (world > 1 and cfg.get("dp_gradient_wire", "bf16") == "bf16") else None
        if world > 1:
            say(f">>> data parallel: {world} ranks x {local} rays (global batch {local * world}); table gradient all-reduced level "
                f"group by level group ({'bf16' if wire is not None else 'fp32'} on the wire), clip after the all-reduce")
        with torch.no_grad():                               # start from the NeuralField's weights (its init or the checkpoint)
            eng.table.copy_(model.representation.encoding.params)
            eng.net.copy_(model.decoder.flat_parameters())
            eng.packed = ops.imlp_pack(eng.net)
            eng.grid.copy_(grid.grid)
            eng.binary_grid.copy_(grid.binary_grid)
        # replicas start from rank 0's values whatever the seeds did (then stay equal: identical all-reduced gradients, a
        # squared norm summed in a fixed order, replicated occupancy-grid updates)
        parallel.broadcast_([eng.table, eng.net, eng.grid, eng.binary_grid])
        eng.packed = ops.imlp_pack(eng.net)
        sharded = world > 1 and cfg.get("dp_sharded_optimizer", True) and eng.half_table
        if sharded:
            # reduce-scatter of the table gradient, every rank steps its 1/N slice of the table (TV + norm + AdamW),
            # all-gather of the fp16 copy the forward reads (project-nerf_amd/sharded.py); `dp_sharded_optimizer: false`: the
            # replicated optimiser (all-reduce of the whole table gradient, level group by level group)
            eng.enable_sharded_optimizer(rank)
            say(f">>> sharded optimiser: every rank steps {eng.shard.per} of {eng.table.numel()} table parameters")

        def sync():
            eng.gather_master()                              # sharded optimiser: the other ranks' slices of the fp32 master
            with torch.no_grad():
                model.representation.encoding.params.copy_(eng.table)
                model.decoder.sigma_net.params.copy_(eng.net[:model.decoder.sigma_net.params.numel()])
                model.decoder.color_net.params.copy_(eng.net[model.decoder.sigma_net.params.numel():])
                grid.grid = eng.grid.clone()
                grid.binary_grid = eng.binary_grid.clone()

        warm, stop = cfg.get("grid_warmup_iters", 256), cfg.get("grid_stop_ratio", 0.9)
        val_idx = random.sample(range(len(test_set)), max(1, int(len(test_set) * 0.3)))
        active, ahead = 1.0, []

        def draw():
            # every rank draws the same global batch of pixels (same torch seed) and forms rays [lo, hi) of it; the jitter of
            # sample g of the shard is draw (lo * n_samples + g) of the step: the union over the ranks is one GPU's batch
            o, d, target = train_set.sample_batch(local * world, eng.bg, shard=(lo, hi) if world > 1 else None)
            return o, d, target, eng.prepare_batch(o, d, n_samples, first_ray=lo)

        for step in range(1, iters + 1):
            if not ahead:
                ahead.append(draw())
            o, d, target, prepared = ahead.pop()
            ahead.append(draw())
            loss_rgb = eng.train_step(o, d, target, n_samples, prepared=prepared, sync_grads_async=None if sharded else sync_async,
                                      reduce_dtype=wire)
            if step < iters * stop:
                interval = 32 if step < iters * 0.1 else (128 if step < iters * 0.5 else 512)
                if step >= warm and step % interval == 0:
                    active = eng.update_grid()
                    ahead.clear()                           # the waiting batch was compacted against the previous grid
            if step % log_every == 0:
                loss_val = parallel.mean_over_ranks(loss_rgb).item()
                say(f">>> Step {step}/{iters} | Loss {loss_val:.6f} | PSNR {compute_psnr(loss_val):.2f} dB"
                    f" | Skip: {(1 - active) * 100:.1f}%")
            if step % cfg.get("val_every", 500) == 0:
                sync()
                v = evaluate(test_set, val_idx)
                say(f"    [Validation] PSNR: {v:.2f} dB")
                if v > best and main_rank:
                    best = v
                    torch.save({"model_state_dict": model.state_dict(), "config": cfg, "step": step, "val_psnr": best,
                                "density_grid": grid.state_dict()}, os.path.join(log_dir, "best_model.pth"))
        if world > 1:
            eng.gather_master()
            say(f">>> replica divergence after {iters} steps: {parallel.replica_divergence([eng.table, eng.net, eng.binary_grid]):.3e}")
        sync()
    elif not args.eval_only:
        parallel.broadcast_([p.data for p in model.parameters()] + ([grid.grid, grid.binary_grid] if grid is not None else []))
        opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=cfg.get("weight_decay", 1e-5))
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=iters, eta_min=cfg.get("eta_min", 1e-4))
        use_tv, tv_w = cfg.get("use_tv_loss", True), float(cfg.get("tv_loss_weight", 1e-6))
        warm, stop = cfg.get("grid_warmup_iters", 256), cfg.get("grid_stop_ratio", 0.9)
        n_val = max(1, int(len(test_set) * 0.3))
        val_idx = random.sample(range(len(test_set)), n_val)
        active = 1.0
        model.train()
        local = parallel.check_global_batch(batch, world)
        for step in range(1, iters + 1):
            o, d, rgba = train_set.sample_random_rays(local * world, device)       # same draw on every rank, own shard kept
            o, d, rgba = (t[rank * local:(rank + 1) * local].contiguous() for t in (o, d, rgba))
            bg = torch.ones(3, device=device) if white_bkgd else torch.zeros(3, device=device)
            target = rgba[:, :3] * rgba[:, 3:4] + bg * (1 - rgba[:, 3:4])
            pred, _, _ = render_rays(model, o, d, near, far, n_samples, True, white_bkgd=white_bkgd,
                                     density_grid=grid, bg_color=bg)
            loss_rgb = torch.nn.functional.mse_loss(pred, target)
            loss = loss_rgb
            if use_tv:
                p = model.representation.encoding.params
                loss = loss + torch.mean(torch.abs(p[1:] - p[:-1])) * tv_w
            opt.zero_grad()
            loss.backward()
            parallel.allreduce_mean_grads_(list(model.parameters()))           # clip AFTER the all-reduce: the global norm
            torch.nn.utils.clip_grad_norm_(model.representation.parameters(), max_norm=1.0)
            torch.nn.utils.clip_grad_norm_(model.decoder.parameters(), max_norm=1.0)
            opt.step()
            sched.step()
            if grid is not None and step < iters * stop:
                interval = 32 if step < iters * 0.1 else (128 if step < iters * 0.5 else 512)
                if grid.should_update(step, interval, warm):
                    model.eval()
                    active = grid.update(model, device=device, time=None)
                    model.train()
            if step % log_every == 0:
                say(f">>> Step {step}/{iters} | Loss {parallel.mean_over_ranks(loss).item():.6f} | PSNR "
                    f"{compute_psnr(parallel.mean_over_ranks(loss_rgb).item()):.2f} dB | Skip: {(1 - active) * 100:.1f}%")
            if step % cfg.get("val_every", 500) == 0:
                v = evaluate(test_set, val_idx)
                say(f"    [Validation] PSNR: {v:.2f} dB")
                if v > best and main_rank:
                    best = v
                    save = {"model_state_dict": model.state_dict(), "config": cfg, "step": step, "val_psnr": best}
                    if grid is not None:
                        save["density_grid"] = grid.state_dict()
                    torch.save(save, os.path.join(log_dir, "best_model.pth"))
    n_eval = len(test_set) if args.render_n in (None, -1) else min(args.render_n, len(test_set))
    avg = evaluate(test_set, range(n_eval))
    say(f">>> Test PSNR: {avg:.2f} dB (best validation {best:.2f} dB)")
    return avg
