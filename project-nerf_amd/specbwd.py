"""Host-side bookkeeping of the SPECULATIVE hash backward on one workspace (include/nerf_hip.h:
nerf_hash_encode_bwd_ws_store_spec / _tables_spec; the reference's counterpart is tcnn's scatter inside loss.backward(),
run.py:619 / 1941).

The speculative form skips the count pass: the bins' capacities come from the TRUE record counts the previous call on the same
workspace left behind.  That is sound while consecutive batches fill the bins alike -- same occupancy grid, about the same number of
active points -- which this class decides; a record that does not fit is added through an overflow list by the call's last launch,
and that launch publishes eight status words into a host-mapped block ([3] overflowed records, [4] != 0: records were LOST, [7] = 1
written last).  The block is read a step or two later without any event or copy launch; a lost record switches the engine back to
the counted form for good (and warns: one step's table gradient was incomplete)."""
from __future__ import annotations

import warnings
from typing import Optional

import torch

from . import _lib, ops


class SpeculativeScatter:
    def __init__(self, enabled: bool):
        self.enabled = bool(enabled)
        self._from = None                 # (workspace address, occupancy-grid identity, point count) of the call that left the estimates
        self._clean = None                # workspace whose header the last call (a speculative one) left clean: no spec_begin needed
        self._status: Optional[torch.Tensor] = None      # two host-mapped status blocks, used alternately
        self._turn = 0
        self._checks = {}                 # turn -> status block not yet read
        self.last_status: Optional[torch.Tensor] = None
        self.calls = 0

    # -- decision ------------------------------------------------------------------------------------------------------
    def ok(self, ws_ptr: int, grid_id, n: int) -> bool:
        """may this call trust the record counts the last call left in the workspace?"""
        if not self.enabled or ops.deterministic() or self._from is None:
            return False
        for turn in list(self._checks):                       # blocks of earlier calls that have been published
            if int(self._checks[turn][7]) != 0:
                self._read(turn)
        if not self.enabled:
            return False
        last_ws, last_grid, n_last = self._from
        return last_ws == ws_ptr and last_grid == grid_id and 0.5 * n_last <= n <= 1.1 * n_last

    def _read(self, turn: int) -> None:
        status = self._checks.pop(turn)
        if int(status[7]) == 0:                               # not published yet: wait for the stream (rare: the block is two calls old)
            torch.cuda.current_stream().synchronize()
        if int(status[4]) != 0:
            warnings.warn(f"speculative hash backward: records were lost (status {status.tolist()}): one step's table gradient was "
                          "incomplete; the counted form is used from here on")
            self.enabled = False

    # -- around a speculative call ----------------------------------------------------------------------------------------
    def begin(self, ws: torch.Tensor) -> None:
        """before the producer's backward: a clean header (a launch only when the previous call on the workspace was a counted one)"""
        if self._clean != ws.data_ptr():
            _lib.check(_lib.load().nerf_hash_encode_bwd_spec_begin(ws.data_ptr(), ops._stream()), "nerf_hash_encode_bwd_spec_begin")

    def status_block(self) -> torch.Tensor:
        """the host-mapped block this call's last launch publishes its status into"""
        if self._status is None:
            self._status = torch.zeros(2, 8, dtype=torch.int32).pin_memory()
        self._turn ^= 1
        if self._turn in self._checks:                        # the call that used this block two calls ago: read it before reuse
            self._read(self._turn)
        status = self._status[self._turn]
        status[7] = 0
        return status

    def issued(self, ws: torch.Tensor, grid_id, n: int, status: torch.Tensor) -> None:
        self._from = (ws.data_ptr(), grid_id, n)
        self._clean = ws.data_ptr()
        self._checks[self._turn] = status
        self.last_status = status
        self.calls += 1

    # -- other calls on the workspace ---------------------------------------------------------------------------------------
    def counted(self, ws: torch.Tensor, grid_id, n: int) -> None:
        """a counted all-level call left the bins' true counts (and a used header) in the workspace"""
        self._from = (ws.data_ptr(), grid_id, n)
        self._clean = None

    def invalidate(self) -> None:
        """something else used the workspace (a level-range call, another point set): no estimates"""
        self._from = None
        self._clean = None
