"""Launch time of the decoder chain kernels against the number of 256-sample tiles per workgroup (development aid):
T(passes) = fixed + per_pass * passes.  256 CUs x 256 samples = 65,536 samples per pass."""
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
rows = []
for passes in (1, 2, 3, 4, 6, 8, 16, 32):
    R, S = 1024 * passes, 64
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
    t_trn = tm(lambda: ops.mlp_fwd(packed, o, d, z, stash))
    t_bwd = tm(lambda: lib.nerf_mlp_bwd_dgrad_ex(P(packed), P(stash), P(rgb), P(sigma), P(d_rgb), P(d_sigma), n, P(ws), P(amax), st))
    rows.append((passes, t_inf, t_trn, t_bwd))
    print(f"passes {passes:3d}: infer {t_inf:8.1f} us   fwd+stash {t_trn:8.1f} us   dgrad {t_bwd:8.1f} us", flush=True)
import numpy as np
a = np.array(rows)
for col, name in ((1, "infer"), (2, "fwd+stash"), (3, "dgrad")):
    slope, icpt = np.polyfit(a[:, 0], a[:, col], 1)
    print(f"{name}: {icpt:.1f} us fixed + {slope:.1f} us per pass")
