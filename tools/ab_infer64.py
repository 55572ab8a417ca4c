"""Inference decoder: 64 samples per wave / four waves per workgroup (option infer64) against the 32-sample / eight-wave stream on the
same inputs: outputs bit for bit, launch time on a render chunk (65,536 rays x 128 samples), interleaved.   python tools/ab_infer64.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from project_nerf_amd import _lib, ops  # noqa: E402
from project_nerf_amd.engine import default_init  # noqa: E402

R, S = 65536, 128
g = torch.Generator().manual_seed(1)
o = (torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1) * 4.0).cuda()
d = torch.nn.functional.normalize(-o.cpu() + 0.4 * torch.randn(R, 3, generator=g), dim=-1).cuda()
z = ops.sample_rays(o, d, 2.0, 6.0, S)
params = default_init(0).cuda()
packed = ops.mlp_pack(params)
outs = {}
for mode in (0, 1):
    _lib.set_option("infer64", mode)
    rgb, sigma = ops.mlp_fwd(packed, o, d, z)[:2]
    torch.cuda.synchronize()
    outs[mode] = (rgb.clone(), sigma.clone())
print("rgb equal:", torch.equal(outs[0][0], outs[1][0]), " sigma equal:", torch.equal(outs[0][1], outs[1][1]),
      " max |d rgb|", float((outs[0][0] - outs[1][0]).abs().max()), " max |d sigma|", float((outs[0][1] - outs[1][1]).abs().max()))
# a ragged size (not a multiple of 256 samples)
n_r = 1000
for mode in (0, 1):
    _lib.set_option("infer64", mode)
    outs[mode] = ops.mlp_fwd(packed, o[:n_r], d[:n_r], z[:n_r].contiguous())[:2]
print("ragged equal:", torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]))
for rep in range(3):
    for mode in (0, 1):
        _lib.set_option("infer64", mode)
        for _ in range(3):
            ops.mlp_fwd(packed, o, d, z)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(20):
            ops.mlp_fwd(packed, o, d, z)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / 20
        print(f"infer64={mode}: {ms:.3f} ms per {R} x {S} samples  -> {640000 * 128 / (R * S) * ms:.2f} ms per 800x800x128 frame = "
              f"{1e3 / (640000 * 128 / (R * S) * ms):.2f} FPS")
_lib.set_option("infer64", 0)
