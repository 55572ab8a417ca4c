"""debug aid: canonical chain of the fused Part 4 field against the fp32 module path, with parts of S1 zeroed"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_part4_engine as E

torch.manual_seed(0)
for variant in ("full", "no_time_cols", "no_hash_cols", "zero_delta"):
    outs = []
    for fused in (False, True):
        m, g = E.build_model(fused)
        with torch.no_grad():
            s1 = m.decoder.sigma_net.params[:4096].view(64, 64)
            if variant == "no_time_cols":
                s1[:, 32:] = 0
            if variant == "no_hash_cols":
                s1[:, :32] = 0
            if variant == "zero_delta":
                m.deform_decoder.displacement_scale.zero_()
            rgb, sigma, delta = m.eval()(E.T(g["pts"]).cuda(), E.T(g["dirs"]).cuda(), t=E.T(g["times"]).cuda())
        outs.append((rgb.cpu(), sigma.cpu(), delta.cpu()))
    (r0, s0, d0), (r1, s1_, d1) = outs
    print(f"{variant:14s} rgb {float((r0 - r1).abs().max()):.3e} sigma abs {float((s0 - s1_).abs().max()):.3e} (max {float(s0.abs().max()):.3e}) delta {float((d0 - d1).abs().max()):.3e}")
    if variant == "full":
        i = int((s0 - s1_).abs().argmax())
        print("  worst sigma row", i, float(s0.view(-1)[i]), float(s1_.view(-1)[i]), "t", float(g["times"].reshape(-1)[i]))
