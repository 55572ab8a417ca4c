"""Ray data-parallel path with world_size 2 on CPU (gloo): shards, the single gradient all-reduce
and the row-band gather.  The HIP kernels are not involved (no GPU here); the collectives and the
bookkeeping are exactly what bench.py / the engine use with backend nccl (= RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import project_nerf_amd  # noqa: F401
    from project_nerf_amd import parallel as P
    r, _, w = P.init_distributed("cpu")
    assert (r, w) == (rank, world)
    # a "model": loss = mean over rays of (a . x)^2; each rank owns a contiguous ray shard
    g = torch.Generator().manual_seed(0)
    rays = torch.randn(101, 7, generator=g)
    a = torch.randn(7, generator=g)
    lo, hi = P.shard_range(rays.shape[0], rank, world)
    local = rays[lo:hi]
    # local gradient of the SUM over the shard, then one all-reduce and a 1/N scale = global mean gradient
    grad = (2 * (local @ a)[:, None] * local).sum(0)
    P.allreduce_sum_(grad)
    grad /= rays.shape[0]
    full = (2 * (rays @ a)[:, None] * rays).mean(0)
    ok_grad = torch.allclose(grad, full, rtol=1e-5, atol=1e-6)
    # the overlapped form: two ranges of one flat buffer reduced by two asynchronous collectives
    flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
    handles = [P.allreduce_sum_async(flat[6:]), P.allreduce_sum_async(flat[:6])]
    for h in handles:
        h.wait()
    ok_grad = ok_grad and torch.equal(flat, torch.arange(10, dtype=torch.float32) * sum(range(1, world + 1)))
    # row-band gather of a rendered image
    H = 13
    img = torch.arange(H * 5 * 3, dtype=torch.float32).view(H, 5, 3)
    blo, bhi = P.shard_range(H, rank, world)
    got = P.gather_row_bands(img[blo:bhi].clone(), H, dst=0)
    ok_img = (got is None) if rank != 0 else torch.equal(got, img)
    out[rank] = bool(ok_grad and ok_img)
    dist.barrier()
    dist.destroy_process_group()


def _overlap_worker(rank, world, port, out):
    """ops.mlp_bwd_overlapped and InstantNgpEngine.compute_gradients drive their reduce callback with REAL
    asynchronous gloo collectives; the kernels are replaced by a recording stand-in that fills the gradient
    ranges (no GPU here): order of launches and collectives, ranges, waits."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import project_nerf_amd  # noqa: F401
    from project_nerf_amd import ops, parallel as P
    P.init_distributed("cpu")
    n_params = ops.MLP_PARAM_COUNT
    split = 300000
    grads = torch.zeros(n_params)
    log = []

    class FakeLib:
        def nerf_mlp_bwd_dgrad_ex(self, *a):
            log.append("dgrad")
            return 0

        def nerf_mlp_wgrad_part_split(self):
            return split

        def nerf_mlp_bwd_wgrad_part(self, stash, ws, n, g, part, stream):
            log.append(f"wgrad{part}")
            (grads[split:] if part == 1 else grads[:split]).fill_(float(rank + 1) * part)
            return 0

    real_load, real_stream = ops._lib.load, ops._stream
    ops._lib.load, ops._stream = (lambda: FakeLib()), (lambda: 0)
    try:
        class T:                                            # stands in for device tensors: only data_ptr / numel are used
            def data_ptr(self): return 0
            def numel(self): return 64

        def reduce_async(view):
            log.append(f"reduce{view.numel()}")
            return P.allreduce_sum_async(view)
        ops.mlp_bwd_overlapped(T(), T(), T(), T(), T(), T(), grads, T(), reduce_async)
    finally:
        ops._lib.load, ops._stream = real_load, real_stream
    want_order = ["dgrad", "wgrad1", f"reduce{n_params - split}", "wgrad2", f"reduce{split}"]
    tot = sum(range(1, world + 1))
    ok = log == want_order and bool((grads[split:] == tot * 1.0).all()) and bool((grads[:split] == tot * 2.0).all())
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_world_size_2_overlapped_backward_callback_protocol():
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_overlap_worker, args=(2, port, out), nprocs=2, join=True)
        assert dict(out) == {0: True, 1: True}


def test_shard_range_partitions():
    import project_nerf_amd  # noqa: F401
    from project_nerf_amd.parallel import shard_range
    for n in (0, 1, 7, 8, 800, 4097):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


@pytest.mark.timeout(120)
def test_world_size_2_allreduce_and_gather():
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
        assert dict(out) == {0: True, 1: True}


def _loop_helpers_worker(rank, world, port, out):
    """the helpers run.py's data-parallel loops are built from (parallel.rank_world / mean_over_ranks /
    allreduce_mean_grads_ / render_row_bands) and the Part 4 engine's collective schedule, with real gloo collectives"""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import project_nerf_amd  # noqa: F401
    from project_nerf_amd import parallel as P
    P.init_distributed("cpu")
    ok = P.rank_world() == (rank, world)
    ok = ok and abs(float(P.mean_over_ranks(torch.tensor(float(rank + 1)))) - (world + 1) / 2) < 1e-6
    # module path: every rank's gradients of ITS shard's mean loss, averaged = gradient of the global batch's mean loss
    g = torch.Generator().manual_seed(3)
    x, w0 = torch.randn(64, 5, generator=g), torch.randn(5, 2, generator=g)
    lin = torch.nn.Linear(5, 2)
    with torch.no_grad():
        lin.weight.copy_(w0.t())
        lin.bias.zero_()
    lo, hi = P.shard_range(64, rank, world)
    lin(x[lo:hi]).pow(2).mean().backward()
    P.allreduce_mean_grads_(list(lin.parameters()))
    ref = torch.nn.Linear(5, 2)
    with torch.no_grad():
        ref.weight.copy_(w0.t())
        ref.bias.zero_()
    ref(x).pow(2).mean().backward()
    ok = ok and torch.allclose(lin.weight.grad, ref.weight.grad, rtol=1e-5, atol=1e-6) and torch.allclose(lin.bias.grad, ref.bias.grad, rtol=1e-5, atol=1e-6)
    # a parameter that has NO gradient on one rank (its shard has no active sample; a branch unused on this step): the flat
    # buffer still covers every parameter -- same collective length on all ranks -- and every rank ends with the same gradient
    two = torch.nn.ModuleList([torch.nn.Linear(3, 2), torch.nn.Linear(3, 2)])
    with torch.no_grad():
        for k, m in enumerate(two):
            m.weight.fill_(0.5 + k)
            m.bias.zero_()
    inp = torch.ones(4, 3)
    (two[0](inp).sum() + (two[1](inp).sum() if rank == 0 else 0.0)).backward()
    assert (two[1].weight.grad is None) == (rank != 0)
    P.allreduce_mean_grads_(list(two.parameters()))
    ok = ok and torch.allclose(two[0].weight.grad, torch.full((2, 3), 4.0)) and torch.allclose(two[1].weight.grad, torch.full((2, 3), 4.0 / world))
    # start-up synchronisation: every replica takes rank 0's values (bool tensors: the occupancy grid)
    t_f, t_b = torch.full((5,), float(rank + 7)), torch.tensor([rank == 0, True, rank != 0])
    P.broadcast_([t_f, t_b])
    ok = ok and bool((t_f == 7.0).all()) and t_b.tolist() == [True, True, False]
    ok = ok and P.check_global_batch(4096, world) == 4096 // world
    try:
        P.check_global_batch(4097, 2)
        ok = False
    except ValueError:
        pass
    # evaluation: row bands rendered per rank, gathered on rank 0 -- also when there are fewer rows than ranks would like
    for H in (13, 1):
        o = torch.arange(H * 4 * 3, dtype=torch.float32).view(H, 4, 3)
        img = P.render_row_bands(lambda ob, db: ob * 2 + db, o, o * 0 + 1)
        ok = ok and ((img is None) if rank else torch.equal(img, o * 2 + 1))
    # the Part 4 engine's reduce schedule on a rank WITHOUT active samples equals the busy ranks' (canonical grid, the three
    # deformation grids, then the networks): here only the order / sizes of the collectives matter
    sizes = [40, 12, 12, 12, 9]
    bufs = [torch.full((s,), float(rank + 1)) for s in sizes]
    handles = [P.allreduce_sum_async(b) for b in bufs]
    for h in handles:
        h.wait()
    ok = ok and all(bool((b == sum(range(1, world + 1))).all()) for b in bufs)
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_world_size_2_training_loop_helpers():
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_loop_helpers_worker, args=(2, port, out), nprocs=2, join=True)
        assert dict(out) == {0: True, 1: True}


def _sharded_worker(rank, world, port, out):
    """project-nerf_amd/sharded.py's collectives and slice bookkeeping with real gloo collectives (no kernels: no GPU here)"""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import project_nerf_amd  # noqa: F401
    from project_nerf_amd import parallel as P, sharded as S
    P.init_distributed("cpu")
    n = 5 * 1024 + 512                                       # not a multiple of world x 1024
    n_pad = S.padded_length(n, world)
    per = n_pad // world
    ok = n_pad % (world * 1024) == 0 and n_pad >= n and per * world == n_pad
    flat = torch.zeros(n_pad)
    flat[:n] = torch.arange(n, dtype=torch.float32) * (rank + 1)
    S.reduce_scatter_sum_(flat, per, rank, world)
    want = torch.zeros(n_pad)
    want[:n] = torch.arange(n, dtype=torch.float32) * sum(range(1, world + 1))
    ok = ok and torch.equal(flat[rank * per:(rank + 1) * per], want[rank * per:(rank + 1) * per])
    half = torch.zeros(n_pad, dtype=torch.float16)
    half[rank * per:(rank + 1) * per] = float(rank + 1)
    S.all_gather_slices_(half, per, rank, world)
    ok = ok and all(bool((half[r * per:(r + 1) * per] == float(r + 1)).all()) for r in range(world))
    # slices, pieces and halos of a four-table layout (Part 4's: three equal deformation grids + a larger canonical grid)
    nd, nc = 1536, n - 3 * 1536
    tabs = [(k * nd, nd, 0.1) for k in range(3)] + [(3 * nd, nc, 0.2)]
    bufs = [torch.zeros(n_pad) for _ in range(4)] + [torch.zeros(n_pad, dtype=torch.float16)]
    opt = S.ShardedTableOptimizer(tabs, n, rank, world, *bufs)
    covered = torch.zeros(n, dtype=torch.int32)
    for a, cnt, table_elems, tv_w, halo in opt.pieces:
        covered[a:a + cnt] += 1
        off = next(o for o, c, _ in tabs if o <= a < o + c)
        ok = ok and (halo & 1) == (1 if a > off else 0) and table_elems in (nd, nc) and a % 4 == 0 and cnt % 4 == 0
    total = covered.clone()
    dist.all_reduce(total, op=dist.ReduceOp.SUM)
    ok = ok and bool((total == 1).all())                      # every table element belongs to exactly one rank's pieces
    # the neighbours' edge elements after a step
    opt.params[:n] = torch.arange(n, dtype=torch.float32) + 1000.0 * rank
    opt.exchange()
    if rank > 0:
        ok = ok and float(opt.params[opt.lo - 1]) == float(opt.lo - 1) + 1000.0 * (rank - 1)
    if rank + 1 < world and opt.hi < n:
        ok = ok and float(opt.params[opt.hi]) == float(opt.hi) + 1000.0 * (rank + 1)
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_world_size_2_sharded_optimizer_collectives():
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_sharded_worker, args=(2, port, out), nprocs=2, join=True)
        assert dict(out) == {0: True, 1: True}
