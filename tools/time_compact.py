"""single-pass compaction (returning atomic per 4096 samples) against the ordered three-launch form, on an Instant-sized and a Part 4-sized batch"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import project_nerf_amd  # noqa
from project_nerf_amd import ops


def event_ms(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


torch.manual_seed(0)
for R, S, res in ((16384, 128, 128), (8192, 64, 64)):
    o = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda"), dim=-1) * 4.03
    d = torch.nn.functional.normalize(-o + 0.3 * torch.randn(R, 3, device="cuda"), dim=-1)
    ax = torch.linspace(-1.5, 1.5, res)
    gx, gy, gz = torch.meshgrid(ax, ax, ax, indexing="ij")
    grid = ((gx ** 2 + gy ** 2 + gz ** 2) < 0.75 ** 2).cuda()
    out = {}
    for det in (False, True):
        ops.set_deterministic(det)
        run = lambda: ops.sample_compact_async(o, d, 2.0, 6.0, S, grid, 1.5, jitter=(0, 3))
        n = run().get()[2].shape[0]
        out[det] = event_ms(run)
    ops.set_deterministic(False)
    print(f"{R} x {S}, {n} active: single pass {out[False]:.1f} us, ordered (three launches) {out[True]:.1f} us   (includes the pinned read-back queueing)")
