"""Per-level cost of the hash-grid backward scatter (development aid; NERF_HASH_BWD_ONLY_LEVEL)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import project_nerf_amd
from project_nerf_amd import ops
t = ops.HashLevelTable(16, 19, 16, 1.5)
n = 198000
torch.manual_seed(0)
# samples along rays through a 12 % occupied region, as in the steady-state training batch
o = torch.randn(n // 128 + 1, 1, 3) * 0.3
d = torch.nn.functional.normalize(torch.randn(n // 128 + 1, 1, 3), dim=-1)
pts = (o + d * torch.linspace(-0.8, 0.8, 128).view(1, 128, 1)).reshape(-1, 3)[:n].contiguous().cuda()
d_feat = torch.randn(n, 32, device="cuda")
g = torch.zeros(t.entries, 2, device="cuda")
def tm(it=10):
    for _ in range(2): ops.hash_encode_bwd(pts, t, 1.5, d_feat, g)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): ops.hash_encode_bwd(pts, t, 1.5, d_feat, g)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
print(f"all levels: {tm():.3f} ms")
for lvl in range(16):
    ops._lib.set_option("hash_bwd_only_level", lvl)
    print(f"LDS levels + level {lvl:2d} (res {t.res[lvl] if hasattr(t, 'res') else '?'}): {tm():.3f} ms", flush=True)
