"""ctypes binding of libnerf_comm.so (include/nerf_comm.h): the RCCL exchange step as a C ABI.  torch.distributed
stays the default transport of parallel.py (the driver launches ranks with torch.distributed.run); NativeComm is
the same step for hosts that are not torch.distributed programs, and an A/B for the ones that are."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnerf_comm.so")
c_ptr, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64

PROTOTYPES = {
    "nerf_comm_abi_version": (i32, []),
    "nerf_comm_last_error": (ctypes.c_char_p, []),
    "nerf_comm_unique_id_bytes": (i32, []),
    "nerf_comm_get_unique_id": (i32, [c_ptr]),
    "nerf_comm_init": (i32, [c_ptr, i32, i32, ctypes.POINTER(c_ptr)]),
    "nerf_comm_rank": (i32, [c_ptr]),
    "nerf_comm_world": (i32, [c_ptr]),
    "nerf_comm_allreduce_sum": (i32, [c_ptr, c_ptr, i64, i32, c_ptr]),
    "nerf_comm_reduce_scatter_sum": (i32, [c_ptr, c_ptr, i64, i32, c_ptr]),
    "nerf_comm_all_gather": (i32, [c_ptr, c_ptr, i64, i32, c_ptr]),
    "nerf_comm_gather_tiles": (i32, [c_ptr, c_ptr, ctypes.POINTER(i64), c_ptr, i32, c_ptr]),
    "nerf_comm_destroy": (i32, [c_ptr]),
}
_lib = None


class NerfCommError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NerfCommError(f"{LIB_PATH} not found: build it with `python project-nerf_amd/build.py`")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _check(rc, what):
    if rc != 0:
        raise NerfCommError(f"{what} failed ({rc}): {load().nerf_comm_last_error().decode()}")


class NativeComm:
    """One RCCL communicator for this process's GPU.  ``unique_id``: the 128-byte token of rank 0
    (``NativeComm.unique_id()``), shipped to the other ranks by the caller; ``from_torch_distributed`` ships it
    through an initialised torch.distributed group (any backend)."""

    def __init__(self, unique_id: bytes, rank: int, world: int):
        lib = load()
        if len(unique_id) != lib.nerf_comm_unique_id_bytes():
            raise ValueError(f"unique id must be {lib.nerf_comm_unique_id_bytes()} bytes")
        self._h = c_ptr()
        buf = ctypes.create_string_buffer(unique_id, len(unique_id))
        _check(lib.nerf_comm_init(buf, rank, world, ctypes.byref(self._h)), "nerf_comm_init")
        self.rank, self.world = rank, world

    @staticmethod
    def unique_id() -> bytes:
        lib = load()
        buf = ctypes.create_string_buffer(lib.nerf_comm_unique_id_bytes())
        _check(lib.nerf_comm_get_unique_id(buf), "nerf_comm_get_unique_id")
        return buf.raw

    @classmethod
    def from_torch_distributed(cls):
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls(box[0], rank, world)

    def allreduce_sum_(self, flat: torch.Tensor) -> torch.Tensor:
        """in-place sum over the ranks, enqueued on the current stream (fp32 or bf16, contiguous, on the GPU)"""
        if flat.device.type != "cuda" or not flat.is_contiguous():
            raise NerfCommError("allreduce_sum_: a contiguous tensor on the HIP device is required")
        dtype = {torch.float32: 0, torch.bfloat16: 1}.get(flat.dtype)
        if dtype is None:
            raise TypeError(f"allreduce_sum_: fp32 or bf16, got {flat.dtype}")
        _check(load().nerf_comm_allreduce_sum(self._h, flat.data_ptr(), flat.numel(), dtype, torch.cuda.current_stream().cuda_stream),
               "nerf_comm_allreduce_sum")
        return flat

    def reduce_scatter_sum_(self, flat: torch.Tensor, per: int) -> torch.Tensor:
        """sharded optimiser: flat = world equal slices of ``per`` elements; slice ``rank`` becomes the sum over the ranks (in place)"""
        if flat.device.type != "cuda" or not flat.is_contiguous() or flat.numel() != per * self.world:
            raise NerfCommError("reduce_scatter_sum_: a contiguous HIP tensor of world x per elements is required")
        dtype = {torch.float32: 0, torch.bfloat16: 1}.get(flat.dtype)
        if dtype is None:
            raise TypeError(f"reduce_scatter_sum_: fp32 or bf16, got {flat.dtype}")
        _check(load().nerf_comm_reduce_scatter_sum(self._h, flat.data_ptr(), per, dtype, torch.cuda.current_stream().cuda_stream),
               "nerf_comm_reduce_scatter_sum")
        return flat

    def all_gather_(self, flat: torch.Tensor, per: int) -> torch.Tensor:
        """sharded optimiser: every rank's slice ``rank`` of flat (fp16 copy or fp32 master) on every rank, in place"""
        if flat.device.type != "cuda" or not flat.is_contiguous() or flat.numel() != per * self.world or flat.element_size() not in (2, 4):
            raise NerfCommError("all_gather_: a contiguous HIP tensor of world x per elements of 2 or 4 bytes is required")
        _check(load().nerf_comm_all_gather(self._h, flat.data_ptr(), per, flat.element_size(), torch.cuda.current_stream().cuda_stream),
               "nerf_comm_all_gather")
        return flat

    def gather_row_bands(self, band: torch.Tensor, rows_total: int, dst: int = 0):
        """row bands of an image (parallel.shard_range split of ``rows_total``) -> the whole image on ``dst``"""
        from .parallel import shard_range
        if band.device.type != "cuda":
            raise NerfCommError("gather_row_bands: the band must live on the HIP device")
        if band.dtype != torch.float32:                    # the wire format is fp32: a narrower band would be read past its end
            raise TypeError(f"gather_row_bands: fp32 bands only, got {band.dtype}")
        band = band.contiguous()
        # the row shape comes from the tensor's trailing dimensions, so a rank whose band is EMPTY (rows_total < world)
        # still joins the grouped send/recv with count 0 instead of raising on its own while its peers wait
        per_row = 1
        for d in band.shape[1:]:
            per_row *= int(d)
        rows = [shard_range(rows_total, r, self.world) for r in range(self.world)]
        if band.shape[0] != rows[self.rank][1] - rows[self.rank][0]:
            raise ValueError(f"gather_row_bands: rank {self.rank} holds {band.shape[0]} rows, its shard of {rows_total} has "
                             f"{rows[self.rank][1] - rows[self.rank][0]}")
        counts = (i64 * self.world)(*[(b - a) * per_row for a, b in rows])
        out = torch.empty((rows_total,) + tuple(band.shape[1:]), device=band.device, dtype=torch.float32) if self.rank == dst else None
        _check(load().nerf_comm_gather_tiles(self._h, band.data_ptr(), counts, None if out is None else out.data_ptr(), dst,
                                             torch.cuda.current_stream().cuda_stream), "nerf_comm_gather_tiles")
        return out

    def close(self):
        if self._h:
            _check(load().nerf_comm_destroy(self._h), "nerf_comm_destroy")
            self._h = c_ptr()
