"""Two data-parallel ranks on ONE GPU (gloo; launched by tests/test_gpu_sharded_optimizer.py through torch.distributed.run):
the sharded optimiser of project-nerf_amd/sharded.py -- reduce-scatter of the table gradient, every rank steps its slice (TV + ONE
squared norm + clip + AdamW), all-gather of the fp16 copy -- against the replicated optimiser (all-reduce of the whole gradient,
every rank steps everything) from the same state on the same shards.  Prints one line per check; exit code 0 = all held."""
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, HERE)
import project_nerf_amd  # noqa: F401,E402
from project_nerf_amd import parallel  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


def part4(rank, world):
    from test_gpu_deterministic import PART4, _probes, _rays, _part4_engine
    R, S = 1024, 32
    o, d, target, g = _rays(R, 6)
    t = torch.rand(R, 1, generator=g).cuda()
    lo, hi = rank * R // world, (rank + 1) * R // world
    sl = lambda x: x[lo:hi].contiguous()
    rep, shd = _part4_engine(world=world), _part4_engine(world=world)
    for e in (rep, shd):
        e.max_norm = 0.05                                   # the clip is active
    shd.enable_sharded_optimizer(rank)
    assert len(shd.shard.pieces) >= (2 if rank == 0 else 1)   # rank 0: its slice spans table seams
    ok = True
    for step in (1, 2):
        probes = _probes(step) if step == 2 else None
        l_rep = rep.compute_gradients(sl(o), sl(d), sl(target), sl(t), S, first_ray=lo, probes=probes, sync_grads_async=parallel.allreduce_sum_async)
        l_shd = shd.compute_gradients(sl(o), sl(d), sl(target), sl(t), S, first_ray=lo, probes=probes, shard_grads=True)
        # the reduce-scattered slice equals the all-reduced gradient's slice (gloo sums two ranks in one order: bit-equal)
        a, b = shd.shard.lo, shd.shard.hi
        g_ok = torch.equal(shd._g_tables_buf[a:b], rep._g_tables_buf[a:b]) and torch.equal(shd.g_net, rep.g_net)
        rep.apply_gradients()
        shd.apply_gradients()
        shd.gather_master()
        r_t, r_n, r_h = rel(shd.tables, rep.tables), rel(shd.net, rep.net), rel(shd.tables_h.float(), rep.tables_h.float())
        n_rep, n_shd = float(rep._normsq_ws[0]), float(shd._normsq_ws[0])
        # the only difference: ONE squared norm summed over slices and ranks instead of over the whole vector in workgroup order
        tol = 5e-6 if step == 1 else 5e-2                   # step 2 starts from states that differ by the first step's round-off
        good = g_ok and r_t < tol and r_n < tol and r_h < max(tol, 1e-4) and abs(n_rep - n_shd) <= 1e-5 * n_rep and n_rep > rep.max_norm ** 2
        ok = ok and (good if step == 1 else (r_t < tol and r_n < tol))
        if rank == 0:
            print(f"[part4 step {step}] grads equal {g_ok}; tables {r_t:.2e} networks {r_n:.2e} fp16 copy {r_h:.2e}; normsq {n_rep:.6e} / {n_shd:.6e}; "
                  f"loss {float(l_rep):.7f} / {float(l_shd):.7f}", flush=True)
    # every rank's fp16 copy is the same (all-gathered), the fp32 master too once gathered
    div = parallel.replica_divergence([shd.tables_h, shd.tables, shd.net])
    ok = ok and div == 0.0
    if rank == 0:
        print(f"[part4] replica divergence of the sharded engine: {div:.3e}", flush=True)
    return ok


def instant(rank, world):
    import yaml
    from project_nerf_amd import ops
    from project_nerf_amd.engine import InstantNgpEngine
    from test_gpu_deterministic import _rays, _sphere
    cfg = yaml.safe_load(open(os.path.join(HERE, "..", "configs", "part2_instant.yaml.example")))
    cfg.update(train_iters=10, tv_loss_weight=1e-3, log2_hashmap_size=15)
    R, S = 1024, 48
    o, d, target, g = _rays(R, 4)
    lo, hi = rank * R // world, (rank + 1) * R // world
    sl = lambda x: x[lo:hi].contiguous()
    engs = []
    for _ in range(2):
        e = InstantNgpEngine(cfg, seed=0, world_size=world)
        e.table.copy_(((torch.rand(e.table.numel(), generator=torch.Generator().manual_seed(11)) - 0.5) * 0.5).cuda())
        e.net[2048:2048 + 64] *= 20.0
        ops.imlp_pack(e.net, e.packed)
        e.binary_grid = _sphere(128)
        engs.append(e)
    rep, shd = engs
    shd.enable_sharded_optimizer(rank)
    u = torch.rand(R, S, generator=g).cuda()
    rep.compute_gradients(sl(o), sl(d), sl(target), S, u=sl(u), sync_grads_async=parallel.allreduce_sum_async)
    shd.compute_gradients(sl(o), sl(d), sl(target), S, u=sl(u), shard_grads=True)
    a, b = shd.shard.lo, shd.shard.hi
    g_ok = float((shd._g_table_buf[a:b] - rep._g_table_buf[a:b]).abs().max()) <= 1e-6 * float(rep.g_table.abs().max())
    rep.apply_gradients()
    shd.apply_gradients()
    shd.gather_master()
    r_t, r_n, r_h = rel(shd.table, rep.table), rel(shd.net, rep.net), rel(shd._gather_table().float(), rep._gather_table().float())
    ok = g_ok and r_t < 5e-6 and r_n < 5e-6 and r_h < 1e-4
    if rank == 0:
        print(f"[instant step 1] grads equal {g_ok}; table {r_t:.2e} networks {r_n:.2e} fp16 copy {r_h:.2e}", flush=True)
    return ok


if __name__ == "__main__":
    rank, _, world = parallel.init_distributed("cuda")
    # ordered sums everywhere: the two engines of a rank then compute bit-equal LOCAL gradients, and what is left between the
    # replicated and the sharded step is the one thing that differs by construction -- the order the squared norm is summed in
    from project_nerf_amd import ops
    ops.set_deterministic(True)
    good = part4(rank, world) and instant(rank, world)
    flag = torch.tensor([1.0 if good else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("SHARDED OPTIMISER OK" if float(flag) == 1.0 else "SHARDED OPTIMISER MISMATCH", flush=True)
    sys.exit(0 if float(flag) == 1.0 else 1)
