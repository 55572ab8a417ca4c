import os
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import project_nerf_amd
from project_nerf_amd import ops
from project_nerf_amd.engine import default_init
R, S = 4096, 64; n = R * S
packed = ops.mlp_pack(default_init(0).cuda())
o = torch.randn(R, 3, device="cuda"); d = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda"), dim=-1)
z = ops.sample_rays(o, d, 2.0, 6.0, S)
stash = torch.empty(ops.mlp_stash_bytes(n), dtype=torch.uint8, device="cuda")
rgb, sigma = ops.mlp_fwd(packed, o, d, z, stash)
ws = torch.empty(ops.mlp_bwd_workspace_bytes(n), dtype=torch.uint8, device="cuda")
grads = torch.empty(ops.MLP_PARAM_COUNT, device="cuda")
lib = ops._lib.load(); st = torch.cuda.current_stream().cuda_stream
lib.nerf_mlp_bwd_dgrad(packed.data_ptr(), stash.data_ptr(), rgb.data_ptr(), sigma.data_ptr(), torch.randn_like(rgb).data_ptr(), torch.randn_like(sigma).data_ptr(), n, ws.data_ptr(), st)
def t(it=20):
    for _ in range(3): ops._lib.check(lib.nerf_mlp_bwd_wgrad(stash.data_ptr(), ws.data_ptr(), n, grads.data_ptr(), st), "wgrad")
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): lib.nerf_mlp_bwd_wgrad(stash.data_ptr(), ws.data_ptr(), n, grads.data_ptr(), st)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
gb = (stash.numel() + ws.numel()) * 1e-9
for dbg in (0, 1, 2, 3, 7):
    ops._lib.set_option("wgrad_debug", dbg)
    ms = t()
    print(f"debug={dbg}: {ms:.3f} ms  {gb/ms*1e3:.0f} GB/s", flush=True)
ops._lib.set_option("wgrad_debug", 0)
if stash.numel() > 4000 * n:          # bf16 images (the default): the byte-based span cost model's fixed share per ring stage
    for ovh in (16384, 49152, 98304, 196608, 393216):
        ops._lib.set_option("wgrad_overhead", ovh)
        print(f"wgrad_overhead {ovh}: {t():.3f} ms", flush=True)
    ops._lib.set_option("wgrad_overhead", 98304)
    ops._lib.set_option("wgrad_atomic", 1)
    print(f"wgrad_atomic=1: {t():.3f} ms", flush=True)
    ops._lib.set_option("wgrad_atomic", 0)
    sys.exit(0)
for bw, fixed in ((192, 2000), (192, 400), (192, 1000), (192, 3000), (128, 2000), (256, 2000), (384, 2000), (160, 2000), (224, 1500), (100000, 2000)):
    ops._lib.set_option("wgrad_bw_x16", bw); ops._lib.set_option("wgrad_fixed", fixed)
    print(f"bw_x16 {bw} fixed {fixed}: {t():.3f} ms", flush=True)
ops._lib.set_option("wgrad_bw_x16", 192); ops._lib.set_option("wgrad_fixed", 2000)
for name in ("wgrad_k16", "wgrad_atomic"):
    ops._lib.set_option(name, 1)
    print(f"{name}=1: {t():.3f} ms", flush=True)
    ops._lib.set_option(name, 0)
