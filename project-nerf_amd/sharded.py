"""Sharded optimiser of the big hash-table groups under data parallelism ("one-shot reduce-scatter +
all-gather ... the TV + clip + AdamW pass sharded"; the reference has no distributed code, its optimiser step is
run.py:611-630 / 1941-1944).

Instead of all-reducing the whole table gradient and stepping the whole table on every rank (replicated optimiser: 2 x 114 MB on
the wire and 1.2 GB of optimiser traffic per rank and step for Part 4), every rank

  1. receives the SUM of the ranks' gradients for ITS 1/N slice of the flat table buffer (reduce-scatter),
  2. runs TV + squared norm on the slice (the TV terms at the slice's ends reach one element into the neighbours' slices: that
     element is exchanged once per step), adds its part of the ONE squared norm, and all ranks sum the parts (a scalar all-reduce:
     the clip coefficient is the same bits everywhere),
  3. runs clip + AdamW on the slice (fp32 master, both moments: only the slice is ever touched) writing the slice of the fp16 copy,
  4. all-gathers the fp16 copy, which is what the forward and the hash input gradient read; the fp32 master of the OTHER slices
     goes stale and is refreshed only for checkpoints / validation (``gather_master``).

Wire bytes per step and rank: (N-1)/N x (4 B reduce-scatter + 2 B all-gather) per table parameter instead of 2 x (N-1)/N x 4 B.
The small networks stay replicated (their gradient is all-reduced as before).  Layout: slices are equal, multiples of 1024 elements
(the buffers are allocated padded to world x slice); a slice may span several tables: it is stepped piece by piece (table ∩ slice)
through nerf_tv_normsq_codes_piece / nerf_adamw_clip_step_tv_piece."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist

from . import _lib, ops

Tensor = torch.Tensor
P = lambda t: None if t is None else t.data_ptr()
ALIGN = 1024


def padded_length(n: int, world: int) -> int:
    """length of a flat buffer of n elements padded so that it splits into `world` equal slices of whole 1024-element blocks"""
    per = (n + world * ALIGN - 1) // (world * ALIGN) * ALIGN
    return per * world


class ShardedTableOptimizer:
    """``tables``: (offset, elements, tv_weight) of every table inside the flat buffers (offsets and sizes multiples of 4).
    ``params`` / ``grads`` / ``exp_avg`` / ``exp_avg_sq`` / ``shadow`` (fp16): flat buffers of padded_length(n, world) elements
    (the padding is zero and never stepped)."""

    def __init__(self, tables: Sequence[Tuple[int, int, float]], n: int, rank: int, world: int, params: Tensor, grads: Tensor,
                 exp_avg: Tensor, exp_avg_sq: Tensor, shadow: Tensor):
        self.n, self.rank, self.world = n, rank, world
        n_pad = padded_length(n, world)
        for name, t in (("params", params), ("grads", grads), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq), ("shadow", shadow)):
            if t.numel() != n_pad or not t.is_contiguous():
                raise ValueError(f"sharded optimiser: {name} must be a contiguous buffer of padded_length(n, world) = {n_pad} elements")
        self.per = n_pad // world
        self.lo = rank * self.per
        self.hi = max(self.lo, min((rank + 1) * self.per, n))          # a rank past the end owns padding only
        self.params, self.grads, self.m, self.v, self.shadow = params, grads, exp_avg, exp_avg_sq, shadow
        # TV sign codes: one byte per four elements of the slice, + the byte before the first piece (nerf_hip.h)
        self.codes = torch.zeros(self.per // 4 + 16, dtype=torch.uint8, device=params.device)
        self.pieces: List[Tuple[int, int, int, float, int]] = []     # (lo, n, table_elems, tv_weight, halo)
        for off, cnt, tv_w in tables:
            a, b = max(off, self.lo), min(off + cnt, self.hi)
            if a < b:
                if (a % 4) or ((b - a) % 4):
                    raise ValueError("sharded optimiser: table offsets and sizes must be multiples of 4")
                self.pieces.append((a, b - a, cnt, float(tv_w), (1 if a > off else 0) | (2 if b < off + cnt else 0)))
        self._edge = torch.zeros(2, device=params.device)

    # -- step --------------------------------------------------------------------------------------------------------
    def reduce_scatter_grads(self) -> None:
        """grads[lo:hi] <- sum over ranks of grads[lo:hi] (the other slices are left as they are: never read afterwards)"""
        reduce_scatter_sum_(self.grads, self.per, self.rank, self.world)

    def accumulate_normsq(self, normsq_ws: Tensor, grad_scale: float, first: bool = False) -> None:
        """adds this rank's part of the squared norm of (grads * grad_scale + TV terms) to normsq_ws[0] (``first``: the first piece
        STORES it, a rank without pieces stores zero -- no zeroing launch); leaves the TV sign codes"""
        lib, st = _lib.load(), ops._stream()
        if first and not self.pieces:
            normsq_ws[:1].zero_()
        for i, (a, cnt, table_elems, tv_w, halo) in enumerate(self.pieces):
            _lib.check(lib.nerf_tv_normsq_codes_piece(P(self.params[a:]), P(self.grads[a:]), cnt, table_elems, halo, tv_w, grad_scale,
                                                      P(normsq_ws), 0 if (first and i == 0) else 1, P(self.codes[16 + (a - self.lo) // 4:]), st),
                       "nerf_tv_normsq_codes_piece")

    def adamw(self, normsq_ws: Tensor, step: int, lr: float, weight_decay: float, max_norm: float, grad_scale: float,
              beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8) -> None:
        lib, st = _lib.load(), ops._stream()
        for a, cnt, table_elems, tv_w, halo in self.pieces:
            _lib.check(lib.nerf_adamw_clip_step_tv_piece(P(self.params[a:]), P(self.grads[a:]), P(self.m[a:]), P(self.v[a:]), cnt, step, lr,
                                                         beta1, beta2, eps, weight_decay, P(normsq_ws), max_norm, grad_scale,
                                                         P(self.codes[16 + (a - self.lo) // 4:]) if tv_w != 0.0 else None, tv_w, table_elems,
                                                         halo & 1, P(self.shadow[a:]), st), "nerf_adamw_clip_step_tv_piece")

    def exchange(self) -> None:
        """after the step: the neighbours' edge elements of the fp32 master (next step's TV terms) and the fp16 copy of every slice"""
        if self.world == 1:
            return
        if self.hi > self.lo:
            self._edge[0], self._edge[1] = self.params[self.lo], self.params[self.hi - 1]
        edges = [torch.empty_like(self._edge) for _ in range(self.world)]
        dist.all_gather(edges, self._edge)
        if self.rank > 0:
            self.params[self.lo - 1] = edges[self.rank - 1][1]
        if self.rank + 1 < self.world and self.hi < self.n:
            self.params[self.hi] = edges[self.rank + 1][0]
        all_gather_slices_(self.shadow, self.per, self.rank, self.world)

    def gather_master(self) -> None:
        """the fp32 master of every slice on every rank (checkpoints, validation through the module path)"""
        if self.world > 1:
            all_gather_slices_(self.params, self.per, self.rank, self.world)


def reduce_scatter_sum_(flat: Tensor, per: int, rank: int, world: int) -> None:
    """flat[rank * per:(rank + 1) * per] <- sum over ranks; one collective with RCCL, one reduce per slice with gloo (which has
    no reduce-scatter: tests only)"""
    if world == 1:
        return
    if dist.get_backend() == "nccl":
        dist.reduce_scatter_tensor(flat[rank * per:(rank + 1) * per], flat, op=dist.ReduceOp.SUM)
    else:
        for r in range(world):
            dist.reduce(flat[r * per:(r + 1) * per], dst=r, op=dist.ReduceOp.SUM)


def all_gather_slices_(flat: Tensor, per: int, rank: int, world: int) -> None:
    """every rank's slice of flat on every rank (in place)"""
    if world == 1:
        return
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(flat, flat[rank * per:(rank + 1) * per].clone())
    else:
        dist.all_gather([flat[r * per:(r + 1) * per] for r in range(world)], flat[rank * per:(rank + 1) * per].clone())
