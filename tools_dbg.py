import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import project_nerf_amd
from project_nerf_amd import ops
from oracle import nerf_oracle as O
from test_gpu_parity import synth_rays, flat_params, oracle_param_grads, bf16_decoder
for scale in (1.0, 2.0):
    R, S = 40, 64
    params = O.nerf_init_params(seed=3)
    params = {k: (v * scale if k.endswith("weight") else v) for k, v in params.items()}
    o, d = synth_rays(R, 17)
    u = torch.rand(R, S, generator=torch.Generator().manual_seed(2))
    z = O.stratified_depths(2.0, 6.0, S, R, True, u=u).contiguous()
    n = R * S
    gen = torch.Generator().manual_seed(5)
    d_rgb = torch.randn(n, 3, generator=gen); d_sigma = torch.randn(n, generator=gen)
    packed = ops.mlp_pack(flat_params(params).cuda())
    stash = torch.empty(ops.mlp_stash_bytes(n), dtype=torch.uint8, device="cuda")
    rgb, sigma = ops.mlp_fwd(packed, o.cuda(), d.cuda(), z.cuda(), stash)
    grads = ops.mlp_bwd(packed, stash, rgb, sigma, d_rgb.cuda(), d_sigma.cuda()).cpu()
    pts, dirs = O.ray_points(o, d, z)
    ref = oracle_param_grads(params, pts, dirs, d_rgb, d_sigma)
    off = 0
    print("scale", scale)
    for name, shape in O.nerf_param_shapes():
        cnt = int(np.prod(shape)); g = grads[off:off + cnt].reshape(shape); off += cnt
        rel = float((g - ref[name]).norm() / (ref[name].norm() + 1e-12))
        cos = float((g * ref[name]).sum() / (g.norm() * ref[name].norm() + 1e-20))
        print(f"  {name:24s} rel {rel:.4f} cos {cos:.5f} |ref| {float(ref[name].norm()):.3e} |g| {float(g.norm()):.3e}")
