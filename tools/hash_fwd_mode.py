"""Hash-grid forward (fp16 table -> operand image) in one launch mode, for the counter passes:
    python tools/hash_fwd_mode.py <hash_xcd 0|1> <lds_kb>   (timing by HIP events, three repetitions)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import project_nerf_amd  # noqa: E402,F401
from project_nerf_amd import ops  # noqa: E402

xcd, kb = int(sys.argv[1]), int(sys.argv[2])
torch.manual_seed(0)
L = ops.HashLevelTable()
R, S = 1560, 128
o = torch.nn.functional.normalize(torch.randn(R, 3), dim=-1) * 1.45
d = torch.nn.functional.normalize(-o + 0.5 * torch.randn(R, 3), dim=-1)
t = torch.linspace(0.0, 2.9, S)
pts = (o[:, None] + d[:, None] * t[None, :, None]).reshape(-1, 3).clamp(-1.5, 1.5).cuda().contiguous()
n = pts.shape[0]
table_h = (torch.rand(L.entries, 2) * 2e-4 - 1e-4).cuda().half()
lib = ops._lib.load()
ws = torch.empty(lib.nerf_imlp_workspace_bytes(n), device="cuda", dtype=torch.uint8)
lib.nerf_set_option(b"hash_xcd", xcd)
lib.nerf_set_option(b"hash_fwd_lds_kb", kb)
f = lambda: ops.hash_encode_fwd(pts, table_h, L, 1.5, want_f32=False, out_nat=ws)
for rep in range(3):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        f()
    e1.record()
    torch.cuda.synchronize()
    print(f"hash_xcd {xcd} lds {kb} KB n {n}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us", flush=True)
