"""Does the time of the binned hash backward depend on WHERE its workspace lies?  (round-3 finding: the scatter + reduce
launches of the Part 4 step take 0.05 or 0.09 ms per step from one process to the next -- same box, code and data.)
Eight workspaces allocated one after the other (all kept alive: eight different placements), the canonical grid's backward
(T = 2^20, 54 k points) timed on each, round-robin, several rounds; then the same workspace at shifted offsets.
    python tools/bimodal_probe.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import project_nerf_amd  # noqa: F401,E402
from project_nerf_amd import ops  # noqa: E402


def event_ms(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


if __name__ == "__main__":
    torch.manual_seed(0)
    lv = ops.HashLevelTable(16, 20, 16, 1.5)
    n = 54000
    # points along rays through a sphere (coherent like a real batch: consecutive samples share coarse cells)
    o = torch.nn.functional.normalize(torch.randn(n // 8, 1, 3, device="cuda"), dim=-1) * 1.2
    pts = (o * torch.linspace(-1, 1, 8, device="cuda").view(1, 8, 1) * 0.9 + 0.05 * torch.randn(n // 8, 8, 3, device="cuda")).reshape(-1, 3).contiguous()
    n = pts.shape[0]
    d_feat = torch.randn(n, 32, device="cuda") * 1e-4
    g_table = torch.empty(lv.entries * 2, device="cuda")
    need = ops.hash_encode_bwd_workspace_bytes(n, 16)
    print(f"{n} points, workspace {need / 1e6:.1f} MB, gradient table {g_table.numel() * 4 / 1e6:.1f} MB at {g_table.data_ptr():#x}")
    keep, bufs = [], []
    for k in range(8):
        keep.append(torch.empty(int((3 + 7 * k) * 1e6), dtype=torch.uint8, device="cuda"))      # odd-sized spacers: different placements
        bufs.append(torch.empty(int(need * 1.25), dtype=torch.uint8, device="cuda"))
    run = lambda ws: ops.hash_encode_bwd(pts, lv, 1.5, d_feat, g_table, workspace=ws, overwrite=True)
    for rnd in range(3):
        print(f"round {rnd}: " + "  ".join(f"{b.data_ptr() % (1 << 32):#011x}:{event_ms(lambda: run(b)) * 1e3:6.1f}us" for b in bufs))
    big = torch.empty(int(need * 1.25) + (4 << 20), dtype=torch.uint8, device="cuda")
    print("one buffer, shifted start: " + "  ".join(f"+{off >> 10}K:{event_ms(lambda: run(big[off:])) * 1e3:6.1f}us"
                                                     for off in (0, 256, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096)))
    gts = [torch.empty(lv.entries * 2, device="cuda") for _ in range(4)]
    print("four gradient tables, one workspace: " + "  ".join(f"{t.data_ptr() % (1 << 32):#011x}:" +
          f"{event_ms(lambda: ops.hash_encode_bwd(pts, lv, 1.5, d_feat, t, workspace=bufs[0], overwrite=True)) * 1e3:6.1f}us" for t in gts))
