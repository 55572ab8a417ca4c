"""Binned hash backward of ONE level (or a range) on the steady-state Instant batch shape, for rocprofv3 --stats:
    python tools/hash_bwd_level.py <first_level> <end_level> [n_points]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import project_nerf_amd  # noqa: E402,F401
from project_nerf_amd import ops  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 198000
t = ops.HashLevelTable(16, 19, 16, 1.5)
torch.manual_seed(0)
# samples along rays through a 12 % occupied region, as in the steady-state training batch
o = torch.randn(n // 128 + 1, 1, 3) * 0.3
d = torch.nn.functional.normalize(torch.randn(n // 128 + 1, 1, 3), dim=-1)
pts = (o + d * torch.linspace(-0.8, 0.8, 128).view(1, 128, 1)).reshape(-1, 3)[:n].contiguous().cuda()
d_feat = torch.randn(n, 32, device="cuda")
g = torch.zeros(t.entries, 2, device="cuda")
ws = torch.empty(ops.hash_encode_bwd_workspace_bytes(n, 16), dtype=torch.uint8, device="cuda")
for _ in range(30):
    ops.hash_encode_bwd(pts, t, 1.5, d_feat, g, level_range=(lo, hi), workspace=ws, overwrite=True)
torch.cuda.synchronize()
