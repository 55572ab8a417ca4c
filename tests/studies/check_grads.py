"""Per-tensor gradient errors of the decoder backward: every kernel family against the fp32 oracle and against
the oracle with the family's own rounding points.  Lives under tests/ because it calls the oracle (test
infrastructure: nothing outside tests/, smoke() and bench.py's cpu_baseline may).
    python tests/studies/check_grads.py [R S]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import project_nerf_amd  # noqa
from project_nerf_amd import ops
from oracle import nerf_oracle as O
import test_gpu_parity as T

R, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (40, 64)
params = O.nerf_init_params(seed=3)
o, d = T.synth_rays(R, 17)
z = O.stratified_depths(2.0, 6.0, S, R, True, u=torch.rand(R, S, generator=torch.Generator().manual_seed(2))).contiguous()
n = R * S
gen = torch.Generator().manual_seed(5)
d_rgb, d_sigma = torch.randn(n, 3, generator=gen), torch.randn(n, generator=gen)
pts, dirs = O.ray_points(o, d, z)
ref32 = T.oracle_param_grads(params, pts, dirs, d_rgb, d_sigma)
packed = ops.mlp_pack(T.dev(T.flat_params(params)))
for fam in T.FAMILIES:
    T.select_family(fam)
    stash = torch.empty(ops.mlp_stash_bytes(n), dtype=torch.uint8, device="cuda")
    rgb, sigma = ops.mlp_fwd(packed, T.dev(o), T.dev(d), T.dev(z), stash)
    grads = ops.mlp_bwd(packed, stash, rgb, sigma, T.dev(d_rgb), T.dev(d_sigma)).cpu()
    ref16 = T.bf16_param_grads(params, pts, dirs, d_rgb, d_sigma, fp8_images=fam == "asm-stream-fp8")
    print(f"== {fam}: stash {stash.numel() / n:.0f} B/sample")
    off = 0
    for name, shape in O.nerf_param_shapes():
        cnt = int(np.prod(shape)); g = grads[off:off + cnt].reshape(shape); off += cnt
        rel16 = float((g - ref16[name]).norm() / (ref16[name].norm() + 1e-12))
        rel32 = float((g - ref32[name]).norm() / (ref32[name].norm() + 1e-12))
        cos32 = float((g * ref32[name]).sum() / (g.norm() * ref32[name].norm() + 1e-20))
        print(f"{name:28s} matched {rel16:.4f}  fp32 {rel32:.4f} cos {cos32:.5f}  |g| {float(g.norm()):.3e}")
T.select_family("asm-stream")
