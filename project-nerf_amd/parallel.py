"""Ray data-parallelism over the GPUs of one node.

One process per GPU; rays are independent, so the only exchange in a training step is ONE
all-reduce (sum) of the flat fp32 gradient vector -- 2.38 MB for the vanilla decoder -- over
RCCL/xGMI (``torch.distributed`` backend "nccl" on ROCm; "gloo" on CPU for the tests).  The mean
over the global batch is taken by scaling with 1/world inside the Adam kernel.  Evaluation
splits an image into contiguous row bands and gathers them on rank 0.  The reference has no
distributed code at all (grep: no ``distributed`` / ``all_reduce``), this is the build's own.
"""
import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def init_distributed(device_type: Optional[str] = None) -> Tuple[int, int, int]:
    """Reads RANK / LOCAL_RANK / WORLD_SIZE (torchrun contract); returns (rank, local_rank, world)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("NERF_SINGLE_DEVICE") or os.environ.get("NERF_BENCH_SINGLE_DEVICE"):
        local_rank = 0            # rehearsal of the N > 1 control flow on a one-GPU box (tests only; use with NERF_DIST_BACKEND=gloo)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        use_gpu = (device_type or ("cuda" if torch.cuda.is_available() else "cpu")) == "cuda"
        backend = os.environ.get("NERF_DIST_BACKEND", "nccl" if use_gpu else "gloo")   # override: tests only
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            if use_gpu:
                torch.cuda.set_device(local_rank)
            dist.init_process_group(backend)
    return rank, local_rank, world


def rank_world() -> Tuple[int, int]:
    """(rank, world) of the initialised process group, (0, 1) without one"""
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def mean_over_ranks(value: torch.Tensor) -> torch.Tensor:
    """mean of a (scalar) tensor over the ranks -- logging only; every rank must call it"""
    if dist.is_initialized() and dist.get_world_size() > 1:
        value = value.detach().clone()
        dist.all_reduce(value, op=dist.ReduceOp.SUM)
        value /= dist.get_world_size()
    return value


def allreduce_mean_grads_(params) -> None:
    """module path (torch.optim on nn.Parameters): one summing all-reduce of the flattened gradients, averaged.  The flat
    buffer covers EVERY parameter in the given order -- zeros where a rank has no gradient (a shard without active samples, a
    branch unused on this step) -- so that all ranks issue a collective of the same length, and the result is written back
    to every parameter so that all ranks step the same set."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= dist.get_world_size()
    off = 0
    for p in params:
        g = flat[off:off + p.numel()].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += p.numel()


def broadcast_(tensors, src: int = 0) -> None:
    """Start-up synchronisation of the replicas: every tensor (parameters, optimiser state, occupancy grid) takes rank
    ``src``'s values.  Replicas that start equal and apply identical updates (the all-reduced gradient, a squared norm summed
    in a fixed order) stay bit-equal; this makes the first condition hold whatever the seeds did."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    for t in tensors:
        if t.dtype == torch.bool:                         # gloo / RCCL have no bool collectives
            u = t.to(torch.uint8)
            dist.broadcast(u, src=src)
            t.copy_(u.to(torch.bool))
        else:
            dist.broadcast(t, src=src)


def replica_divergence(tensors) -> float:
    """largest |value on one rank - value on another| over the given replicated tensors (0.0: the replicas are bit-equal);
    every rank must call it.  Two collectives (MAX, MIN) per tensor: an end-of-run check, not a per-step one."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return 0.0
    worst = 0.0
    for t in tensors:
        hi = t.detach().to(torch.float32 if t.dtype in (torch.bool, torch.uint8, torch.float16) else t.dtype).clone()
        lo = hi.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        worst = max(worst, float((hi - lo).abs().max()) if hi.numel() else 0.0)
    return worst


def check_global_batch(batch: int, world: int) -> int:
    """rays per rank of a global batch; the split must be exact (a remainder would silently drop rays)"""
    if batch % world:
        raise ValueError(f"batch_size {batch} is not divisible by the {world} data-parallel ranks")
    return batch // world


def render_row_bands(render_fn, rays_o: torch.Tensor, rays_d: torch.Tensor, dst: int = 0) -> Optional[torch.Tensor]:
    """Evaluation under data parallelism: rays_o / rays_d [H, W, 3]; every rank renders rows shard_range(H, rank, world)
    with ``render_fn(o_band, d_band) -> [rows, W, C]`` and the bands are gathered on ``dst`` (None elsewhere).  A band
    rendered alone is bit-identical to the same rows of the full frame (rays are independent)."""
    rank, world = rank_world()
    H = rays_o.shape[0]
    lo, hi = shard_range(H, rank, world)
    band = render_fn(rays_o[lo:hi].contiguous(), rays_d[lo:hi].contiguous())
    return gather_row_bands(band, H, dst=dst)


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n items; the union over ranks is exactly range(n)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_sum_(flat: torch.Tensor) -> torch.Tensor:
    """In-place sum over ranks of one flat buffer (a single collective per step)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def allreduce_sum_async(flat: torch.Tensor):
    """Starts the in-place sum over ranks of a (contiguous view of a) flat buffer; returns the work
    handle, or None when there is nothing to do.  With the nccl (= RCCL) backend the collective runs
    on the communicator's own stream behind the work already queued on the current stream, so
    kernels launched afterwards overlap with it; ``handle.wait()`` orders the current stream after it."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
    return None


def gather_row_bands(band: torch.Tensor, rows_total: int, dst: int = 0) -> Optional[torch.Tensor]:
    """Each rank holds rows shard_range(rows_total, rank, world) of an image [rows, W, C]; returns
    the assembled image on ``dst`` (None elsewhere)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return band
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(rows_total, r, world) for r in range(world)]
    max_rows = max(hi - lo for lo, hi in sizes)
    pad = band.new_zeros((max_rows,) + tuple(band.shape[1:]))
    pad[: band.shape[0]] = band
    bufs: Optional[List[torch.Tensor]] = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0)


def native_allreduce_sum_async(comm):
    """The engines' ``sync_grads_async`` callback over libnerf_comm.so (``_comm.NativeComm``) instead of
    torch.distributed: the collective is enqueued on a side stream behind the work already queued on the
    current one; ``handle.wait()`` orders the current stream after it."""
    side = torch.cuda.Stream()

    class _Handle:
        def __init__(self, event):
            self.event = event

        def wait(self):
            torch.cuda.current_stream().wait_event(self.event)

    def start(flat: torch.Tensor):
        side.wait_stream(torch.cuda.current_stream())
        flat.record_stream(side)          # the caching allocator must not recycle the buffer while the side stream uses it
        with torch.cuda.stream(side):
            comm.allreduce_sum_(flat)
            event = torch.cuda.Event()
            event.record(side)
        return _Handle(event)

    return start
