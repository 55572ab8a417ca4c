"""Does training still converge when the weight gradients come from narrower training images?  (development aid, numerics
only: a pure-torch fp32 trainer on the GPU whose Linear layers compute dW from QUANTISED copies of their input and of the
pre-activation gradient -- forward and dgrad stay exact, as in the product, where only the wgrad operands are narrow.)

Same scene, batch and optimiser as tools/convergence_ab.py (synthetic scene, 4096 rays x 64 samples, Adam 5e-4); test PSNR
after STEPS steps, mean over the two test views, a few seeds; runs in the dead-density plateau are left out.
Schemes: see tools/quant_wgrad_study.py (fp32 = exact gradient, cur = today's images, mx6 / mx4 = 6 / 4-bit block-scaled).

usage: [SEED0=0] [SCHEMES=fp32,cur,mx6,mx4] python tools/quant_convergence_study.py [STEPS=2000] [SEEDS=3]"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from src.dataset import BlenderDataset, write_synthetic_scene
from project_nerf_amd.engine import default_init, unflatten

dev = "cuda"
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
SEEDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
root = write_synthetic_scene(tempfile.mkdtemp() + "/scene", n_train=20, n_test=2, size=100)
ds = BlenderDataset(root, "train", 1, True, 1.0).to(dev)
test = BlenderDataset(root, "test", 1, True, 1.0)
views = [test.get_image_rays(i, dev) for i in range(len(test))]

FMT = {"e4m3": (4, 3, 448.0), "e5m2": (5, 2, 57344.0), "e3m2": (3, 2, 28.0), "e2m3": (2, 3, 7.5), "e2m1": (2, 1, 6.0)}


def minifloat(x, fmt):
    E, M, max_val = FMT[fmt]
    bias = (1 << (E - 1)) - 1
    ax = x.abs().clamp_min(1e-45)
    e = torch.floor(torch.log2(ax)).clamp(1 - bias, (1 << E) - 1 - bias)
    step = torch.exp2(e - M)
    return torch.sign(x) * (torch.round(ax / step) * step).clamp_max(max_val)


def q_global(x, fmt, amax_to=None):
    if amax_to is None:
        return minifloat(x, fmt)
    s = 2.0 ** torch.ceil(torch.log2(x.abs().max().clamp_min(1e-30) / amax_to))
    return minifloat(x / s, fmt) * s


def q_block(x, fmt, block=32):
    n, f = x.shape
    xb = x.view(n // block, block, f)
    amax = xb.abs().amax(1, keepdim=True).clamp_min(2.0 ** -126)
    s = torch.exp2(torch.floor(torch.log2(amax)) - float(np.floor(np.log2(FMT[fmt][2]))))
    return (minifloat(xb / s, fmt) * s).view(n, f)


SCHEMES = {
    "fp32": None,
    "cur": (lambda a: q_global(a, "e4m3"), lambda g: q_global(g, "e5m2", 128.0)),
    "mx6": (lambda a: q_block(a, "e2m3"), lambda g: q_block(g, "e3m2")),
    "mx4": (lambda a: q_block(a, "e2m1"), lambda g: q_block(g, "e2m1")),
}


class QLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, scheme):
        ctx.save_for_backward(x, w)
        ctx.scheme = scheme
        return x @ w.T + b

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        dx = g @ w
        if ctx.scheme is None:
            return dx, g.T @ x, g.sum(0), None
        qa, qg = ctx.scheme
        gq = qg(g)
        return dx, gq.T @ qa(x), gq.sum(0), None      # the kernel's bias sums come from the same narrow gradient image


def fourier(x, L):
    out = [x]
    for k in range(L):
        out += [torch.sin(x * (2.0 ** k) * np.pi), torch.cos(x * (2.0 ** k) * np.pi)]
    return torch.cat(out, -1)


def field(p, pts, dirs, scheme):
    lin = lambda name, x: QLinear.apply(x, p[name + ".weight"], p[name + ".bias"], scheme)
    x_enc, d_enc = fourier(pts, 10), fourier(dirs, 4)
    h = x_enc
    for i in range(8):
        if i == 4:
            h = torch.cat([h, x_enc], -1)
        h = torch.relu(lin(f"pts_layers.{i}", h))
    sigma = torch.relu(lin("sigma_layer", h))[:, 0]
    hv = torch.relu(lin("view_layer", torch.cat([lin("feature_layer", h), d_enc], -1)))
    return torch.sigmoid(lin("rgb_layer", hv)), sigma


def render(p, o, d, S, scheme, perturb):
    R = o.shape[0]
    t = torch.linspace(0, 1, S, device=dev)
    z = (2.0 * (1 - t) + 6.0 * t).expand(R, S)
    if perturb:
        mids = 0.5 * (z[:, 1:] + z[:, :-1])
        lo, hi = torch.cat([z[:, :1], mids], -1), torch.cat([mids, z[:, -1:]], -1)
        z = lo + (hi - lo) * torch.rand(R, S, device=dev)
    pts = (o[:, None] + d[:, None] * z[..., None]).reshape(-1, 3)
    dirs = (d / d.norm(dim=-1, keepdim=True))[:, None].expand(R, S, 3).reshape(-1, 3)
    rgb, sigma = field(p, pts, dirs, scheme)
    rgb, sigma = rgb.view(R, S, 3), sigma.view(R, S)
    delta = torch.cat([z[:, 1:] - z[:, :-1], torch.full((R, 1), 1e10, device=dev)], -1) * d.norm(dim=-1, keepdim=True)
    alpha = 1 - torch.exp(-sigma * delta)
    T = torch.cumprod(torch.cat([torch.ones(R, 1, device=dev), 1 - alpha + 1e-10], -1), -1)[:, :-1]
    w = alpha * T
    return (w[..., None] * rgb).sum(1) + (1 - w.sum(1, keepdim=True))


def run(seed, scheme):
    p = {k.replace("decoder.", ""): v.clone().to(dev).requires_grad_(True) for k, v in unflatten(default_init(seed)).items()}
    opt = torch.optim.Adam(list(p.values()), lr=5e-4)
    torch.manual_seed(seed)
    t0 = time.time()
    for step in range(1, STEPS + 1):
        o, d, rgba = ds.sample_random_rays(4096, dev)
        target = rgba[:, :3] * rgba[:, 3:4] + (1 - rgba[:, 3:4])
        loss = ((render(p, o, d, 64, scheme, True) - target) ** 2).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        if step == 150 and float(loss.detach()) > 0.1:
            return None
        if step % 500 == 0:
            print(f"    seed {seed} step {step} loss {float(loss):.5f} ({time.time() - t0:.0f} s)", flush=True)
    ps = []
    with torch.no_grad():
        for o_t, d_t, tgt in views:
            o_t, d_t = o_t.reshape(-1, 3), d_t.reshape(-1, 3)
            img = torch.cat([render(p, o_t[i:i + 4096], d_t[i:i + 4096], 64, None, False) for i in range(0, o_t.shape[0], 4096)])
            ps.append(-10 * np.log10(float(((img.clamp(0, 1) - tgt.reshape(-1, 3)) ** 2).mean())))
    return float(np.mean(ps))


SEED0 = int(os.environ.get("SEED0", 0))
ONLY = [k for k in os.environ.get("SCHEMES", ",".join(SCHEMES)).split(",") if k]
print(f"test PSNR after {STEPS} steps (dB), seeds {SEED0}..{SEED0 + SEEDS - 1}", flush=True)
for name in ONLY:
    scheme = SCHEMES[name]
    res = [run(seed, scheme) for seed in range(SEED0, SEED0 + SEEDS)]
    ok = [r for r in res if r is not None]
    mean = f"{np.mean(ok):.2f}" if ok else "-"
    print(f"{name:5s}: " + " ".join("dead " if r is None else f"{r:.2f}" for r in res) + f"  -> mean {mean} dB over {len(ok)} runs", flush=True)
