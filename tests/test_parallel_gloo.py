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
