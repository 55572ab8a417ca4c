"""debug: why does a 1e-5 perturbation of the deformation grids blow the next step's gradients up?"""
import os, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..")); sys.path.insert(0, os.path.join(HERE, ".."))
import project_nerf_amd  # noqa
from project_nerf_amd import ops
from test_gpu_deterministic import _part4_engine, _rays

ops.set_deterministic(True)
for noise in (0.0, 1e-5, 1e-7, "repack-only"):
    eng = _part4_engine()
    R, S = 1024, 32
    o, d, target, g = _rays(R, 6)
    t = torch.rand(R, 1, generator=g).cuda()
    for step in (1, 2):
        loss = float(eng.compute_gradients(o, d, target, t, S))
        gt = [eng.g_table(k) for k in range(4)]
        print(f"noise {noise} step {step}: loss {loss:.8f} scale {float(eng.net[30144]):.6f} g_scale {float(eng.g_net[30144]):.4e} "
              f"|g_net| {float(eng.g_net.norm()):.4e} max {float(eng.g_net.abs().max()):.3e} "
              + " ".join(f"|g{k}| {float(x.norm()):.3e}/{float(x.abs().max()):.2e}" for k, x in enumerate(gt)), flush=True)
        eng.apply_gradients()
        print(f"    after step: normsq {float(eng._normsq_ws[0]):.5e} tables_h==f16(tables): {bool(torch.equal(eng.tables_h, eng.tables.half()))} "
              f"scale {float(eng.net[30144]):.6f}")
        if step == 1 and noise:
            if noise != "repack-only":
                gen = torch.Generator(device="cuda").manual_seed(1234)
                for k in range(3):
                    tt = eng.table(k)
                    tt.mul_(1.0 + noise * torch.randn(tt.shape, device="cuda", generator=gen))
            eng.repack()
