"""Chain kernels on a fraction of the chip (development aid): per-pass time of inference, forward + training
images and dgrad with the workgroup count capped (option chain_grid), 8 passes per workgroup.  With a few CUs
busy the chip is far below its power limit and holds its top clock, so a cost that is still there is issue /
stall time and a cost that is gone was clock (power)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import project_nerf_amd
from project_nerf_amd import ops
from project_nerf_amd.engine import default_init
packed = ops.mlp_pack(default_init(0).cuda())
lib = ops._lib.load(); st = torch.cuda.current_stream().cuda_stream
def tm(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
PASSES = 8
for grid in (8, 32, 64, 128, 256):
    assert lib.nerf_set_option(b"chain_grid", grid) == 0
    R, S = 4 * grid * PASSES, 64
    n = R * S
    o = torch.randn(R, 3, device="cuda"); d = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda"), dim=-1)
    z = ops.sample_rays(o, d, 2.0, 6.0, S)
    stash = torch.empty(ops.mlp_stash_bytes(n), dtype=torch.uint8, device="cuda")
    rgb, sigma = ops.mlp_fwd(packed, o, d, z, stash)
    ws = torch.empty(ops.mlp_bwd_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    d_rgb, d_sigma = torch.randn_like(rgb), torch.randn_like(sigma)
    amax = torch.full((1,), 4.0, device="cuda")
    P = lambda t: t.data_ptr()
    t_inf = tm(lambda: ops.mlp_fwd(packed, o, d, z))
    lib.nerf_set_option(b"infer_shape32", 1)
    t_inf32 = tm(lambda: ops.mlp_fwd(packed, o, d, z))
    lib.nerf_set_option(b"infer_shape32", 0)
    t_trn = tm(lambda: ops.mlp_fwd(packed, o, d, z, stash))
    t_bwd = tm(lambda: lib.nerf_mlp_bwd_dgrad_ex(P(packed), P(stash), P(rgb), P(sigma), P(d_rgb), P(d_sigma), n, P(ws), P(amax), st))
    print(f"workgroups {grid:3d} x {PASSES} passes: infer16 {t_inf / PASSES:6.1f}  infer32 {t_inf32 / PASSES:6.1f}  fwd+stash {t_trn / PASSES:6.1f}  "
          f"dgrad {t_bwd / PASSES:6.1f} us per pass", flush=True)
