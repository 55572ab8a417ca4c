"""Hash-grid forward alone on fixed points (development aid): 1560 rays x 128 samples inside the box (the locality of a
training batch's active samples), fp16 table, operand image out -- the call the Instant step makes."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import project_nerf_amd
from project_nerf_amd import ops
torch.manual_seed(0)
L = ops.HashLevelTable()
R, S = 1560, 128
o = torch.nn.functional.normalize(torch.randn(R, 3), dim=-1) * 1.45
d = torch.nn.functional.normalize(-o + 0.5 * torch.randn(R, 3), dim=-1)
t = torch.linspace(0.0, 2.9, S)
pts = (o[:, None] + d[:, None] * t[None, :, None]).reshape(-1, 3).clamp(-1.5, 1.5).cuda().contiguous()
n = pts.shape[0]
table = (torch.rand(L.entries, 2) * 2e-4 - 1e-4).cuda()
table_h = table.half()
lib = ops._lib.load()
ws = torch.empty(lib.nerf_imlp_workspace_bytes(n), device="cuda", dtype=torch.uint8)
def tm(f, it=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
for xcd in (1, 0):
 lib.nerf_set_option(b"hash_xcd", xcd)
 for kb in [int(x) for x in os.environ.get("LDS_KB", "0,36").split(",")]:
  lib.nerf_set_option(b"hash_fwd_lds_kb", kb)
  a = tm(lambda: ops.hash_encode_fwd(pts, table_h, L, 1.5, want_f32=False, out_nat=ws))
  b = tm(lambda: ops.hash_encode_fwd(pts, table, L, 1.5, want_f32=False, out_nat=ws))
  c = tm(lambda: ops.hash_encode_fwd(pts, table_h, L, 1.5, want_f32=True, out_nat=None))
  print(f"xcd-aware {xcd} lds {kb:3d} KB  n = {n}: fp16 table -> operand image {a:.1f} us | fp32 table -> operand image {b:.1f} us | fp16 table -> fp32 rows {c:.1f} us", flush=True)
