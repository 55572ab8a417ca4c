"""One debug mode of the decoder's weight-gradient launch, for the kernel trace:  python tools/wgrad_mode.py <debug> [rays]
(rocprofv3 --kernel-trace --stats -- python3 tools/wgrad_mode.py 3 gives the GPU duration of the skeleton without host time)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import project_nerf_amd  # noqa: E402,F401
from project_nerf_amd import ops  # noqa: E402
from project_nerf_amd.engine import default_init  # noqa: E402

dbg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
S = 64
n = R * S
packed = ops.mlp_pack(default_init(0).cuda())
o = torch.randn(R, 3, device="cuda")
d = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda"), dim=-1)
z = ops.sample_rays(o, d, 2.0, 6.0, S)
stash = torch.empty(ops.mlp_stash_bytes(n), dtype=torch.uint8, device="cuda")
rgb, sigma = ops.mlp_fwd(packed, o, d, z, stash)
ws = torch.empty(ops.mlp_bwd_workspace_bytes(n), dtype=torch.uint8, device="cuda")
grads = torch.empty(ops.MLP_PARAM_COUNT, device="cuda")
lib = ops._lib.load()
st = torch.cuda.current_stream().cuda_stream
lib.nerf_mlp_bwd_dgrad(packed.data_ptr(), stash.data_ptr(), rgb.data_ptr(), sigma.data_ptr(), torch.randn_like(rgb).data_ptr(),
                       torch.randn_like(sigma).data_ptr(), n, ws.data_ptr(), st)
ops._lib.set_option("wgrad_debug", dbg)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(2):
    e0.record()
    for _ in range(40):
        lib.nerf_mlp_bwd_wgrad(stash.data_ptr(), ws.data_ptr(), n, grads.data_ptr(), st)
    e1.record()
    torch.cuda.synchronize()
print(f"debug={dbg} rays={R}: {e0.elapsed_time(e1) / 40:.4f} ms per call (events)")
