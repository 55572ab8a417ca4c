"""TV + squared-norm pass (nerf_tv_normsq_codes) alone: launch time against the workgroup count (option tv_blocks; 1024 threads per
workgroup) on a table of Part 4's canonical grid size and of the Instant table size.   python tools/time_tv.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from project_nerf_amd import _lib, ops  # noqa: E402

lib = _lib.load()
for n in (24_000_000, 12_600_000, 4_500_000):
    p, g = torch.randn(n, device="cuda"), torch.randn(n, device="cuda") * 1e-3
    codes = torch.empty((n + 3) // 4, dtype=torch.uint8, device="cuda")
    ws = ops.normsq_ws("cuda")
    for blocks in (0, 128, 256, 512, 768, 1024, 2048):
        _lib.set_option("tv_blocks", blocks)
        st = ops._stream()
        for _ in range(5):
            _lib.check(lib.nerf_tv_normsq_codes(p.data_ptr(), g.data_ptr(), n, 1, 1e-4, 1.0, ws.data_ptr(), 0, codes.data_ptr(), st), "tv")
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(50):
            _lib.check(lib.nerf_tv_normsq_codes(p.data_ptr(), g.data_ptr(), n, 1, 1e-4, 1.0, ws.data_ptr(), 0, codes.data_ptr(), st), "tv")
        ev[1].record()
        torch.cuda.synchronize()
        us = ev[0].elapsed_time(ev[1]) * 1e3 / 50
        print(f"n {n:>10d}  tv_blocks {blocks:>5d}: {us:7.1f} us  {n * 8.25 / us / 1e6:6.2f} TB/s  normsq {float(ws[0]):.6e}")
    _lib.set_option("tv_blocks", 0)
