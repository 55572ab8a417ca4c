"""Cost of the hash-grid backward scatter (development aid): atomic form vs the binned (workspace) form, whole
pass and per level (library option hash_bwd_only_level)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import project_nerf_amd
from project_nerf_amd import ops
t = ops.HashLevelTable(16, 19, 16, 1.5)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 198000
torch.manual_seed(0)
# samples along rays through a 12 % occupied region, as in the steady-state training batch
o = torch.randn(n // 128 + 1, 1, 3) * 0.3
d = torch.nn.functional.normalize(torch.randn(n // 128 + 1, 1, 3), dim=-1)
pts = (o + d * torch.linspace(-0.8, 0.8, 128).view(1, 128, 1)).reshape(-1, 3)[:n].contiguous().cuda()
d_feat = torch.randn(n, 32, device="cuda")
g = torch.zeros(t.entries, 2, device="cuda")
ws = torch.empty(ops.hash_encode_bwd_workspace_bytes(n, 16), dtype=torch.uint8, device="cuda")
print(f"{n} points, workspace {ws.numel() / 1e6:.1f} MB")
def tm(workspace=None, it=10):
    for _ in range(2): ops.hash_encode_bwd(pts, t, 1.5, d_feat, g, workspace=workspace)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): ops.hash_encode_bwd(pts, t, 1.5, d_feat, g, workspace=workspace)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
print(f"all levels, atomic form: {tm():.3f} ms")
print(f"all levels, binned form: {tm(ws):.3f} ms")
a, b = torch.zeros_like(g), torch.zeros_like(g)
ops.hash_encode_bwd(pts, t, 1.5, d_feat, a)
ops.hash_encode_bwd(pts, t, 1.5, d_feat, b, workspace=ws)
print(f"binned vs atomic: max |diff| {float((a - b).abs().max()):.3e} of max {float(a.abs().max()):.3e}; norm ratio {float(b.norm() / a.norm()):.7f}")
if "--levels" in sys.argv:
    for lvl in range(16):
        ops._lib.set_option("hash_bwd_only_level", lvl)
        print(f"LDS levels + level {lvl:2d}: {tm():.3f} ms", flush=True)
    ops._lib.set_option("hash_bwd_only_level", -1)
