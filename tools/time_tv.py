"""TV + squared-norm pass (nerf_tv_normsq_codes) alone: launch time against the workgroup count (option tv_blocks; 1024 threads per
workgroup) on a table of Part 4's canonical grid size, of the Instant table size and of the three deformation grids' -- WARM (the
same buffers every launch: 192 MB sit in the 256-MB Infinity Cache) and COLD (six buffer pairs in turn: 1.1 GB, every launch reads
from HBM, as in a training step).   python tools/time_tv.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from project_nerf_amd import _lib, ops  # noqa: E402

lib = _lib.load()
for n in (24_000_000, 12_600_000, 4_500_000):
    pairs = [(torch.randn(n, device="cuda"), torch.randn(n, device="cuda") * 1e-3) for _ in range(6)]
    codes = torch.empty((n + 3) // 4, dtype=torch.uint8, device="cuda")
    ws = ops.normsq_ws("cuda")
    for blocks in (0, 128, 512, 1024):
        _lib.set_option("tv_blocks", blocks)
        st = ops._stream()
        for cold in (False, True):
            def launch(i):
                p, g = pairs[i % 6 if cold else 0]
                _lib.check(lib.nerf_tv_normsq_codes(p.data_ptr(), g.data_ptr(), n, 1, 1e-4, 1.0, ws.data_ptr(), 0, codes.data_ptr(), st), "tv")
            for i in range(6):
                launch(i)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for i in range(48):
                launch(i)
            ev[1].record()
            torch.cuda.synchronize()
            us = ev[0].elapsed_time(ev[1]) * 1e3 / 48
            print(f"n {n:>10d}  tv_blocks {blocks:>5d}  {'cold' if cold else 'warm'}: {us:7.1f} us  {n * 8.25 / us / 1e6:6.2f} TB/s")
    _lib.set_option("tv_blocks", 0)
    del pairs
