import os
import sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import project_nerf_amd
from project_nerf_amd import ops
from project_nerf_amd.engine import default_init
mode = sys.argv[1] if len(sys.argv) > 1 else "infer"
R, S = 65536, 128
packed = ops.mlp_pack(default_init(0).cuda())
o = torch.randn(R, 3, device="cuda"); d = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda"), dim=-1)
z = ops.sample_rays(o, d, 2.0, 6.0, S)
for _ in range(3):
    ops.mlp_fwd(packed, o, d, z)
torch.cuda.synchronize()
