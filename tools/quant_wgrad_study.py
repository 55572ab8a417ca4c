"""How narrow can the training images get?  (development aid, numerics only -- no kernel uses these formats yet)

The training step is energy-bound on the 1400 W board cap and most of the avoidable energy is the 4.9 KB per sample of
8-bit images the chain kernels write for the weight-gradient kernel.  v_mfma_scale_f32_32x32x64_f8f6f4
also takes 6-bit and 4-bit operands with one E8M0 scale per 32 values along K -- for wgrad K is the sample axis.  This
script trains the vanilla field for a while with the product kernels, then recomputes ONE batch's weight gradients in fp32
torch from quantised copies of the layer inputs h_l and pre-activation gradients d_l and reports, per weight tensor, the
cosine and the relative error against the unquantised fp32 gradient:

  cur     : what the kernels do today -- h in e4m3 with unit scale, d in e5m2 with one power-of-two scale per launch
  mx8     : e4m3 / e5m2 with a block scale per (feature, 32 samples)
  mx6     : e2m3 activations / e3m2 gradients, block scales          (-25 % bytes)
  mx6b    : e3m2 both
  mx4     : e2m1 both, block scales                                   (-50 % bytes)
  mx4h    : e2m1 activations, e5m2 gradients (global scale)           (-25 % bytes)

usage: python tools/quant_wgrad_study.py [train_steps=1500]"""
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from src.dataset import BlenderDataset, write_synthetic_scene
from project_nerf_amd.engine import VanillaNerfEngine

dev = "cuda"
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
root = write_synthetic_scene(tempfile.mkdtemp() + "/scene", n_train=20, n_test=2, size=100)
ds = BlenderDataset(root, "train", 1, True, 1.0).to(dev)
eng = VanillaNerfEngine(seed=0, lr=5e-4)
torch.manual_seed(0)
for step in range(steps):
    o, d, rgba = ds.sample_random_rays(4096, dev)
    loss = eng.train_step(o, d, rgba[:, :3] * rgba[:, 3:4] + (1 - rgba[:, 3:4]), 64)
print(f"trained {steps} steps, loss {float(loss):.5f}", flush=True)
sd = {k.replace("decoder.", ""): v.float().requires_grad_(True) for k, v in eng.state_dict().items()}


# ---- one batch in fp32 torch: layer inputs h_l and pre-activation gradients d_l of every Linear ----
def fourier(x, L):
    out = [x]
    for k in range(L):
        out += [torch.sin(x * (2.0 ** k) * np.pi), torch.cos(x * (2.0 ** k) * np.pi)]
    return torch.cat(out, -1)


R, S = 4096, 64
o, d, rgba = ds.sample_random_rays(R, dev)
target = rgba[:, :3] * rgba[:, 3:4] + (1 - rgba[:, 3:4])
t = torch.linspace(0, 1, S, device=dev)
z = (2.0 * (1 - t) + 6.0 * t).expand(R, S)
pts = (o[:, None] + d[:, None] * z[..., None]).reshape(-1, 3)
dirs = (d / d.norm(dim=-1, keepdim=True))[:, None].expand(R, S, 3).reshape(-1, 3)
x_enc, d_enc = fourier(pts, 10), fourier(dirs, 4)
taps = {}          # name -> (input, pre-activation)


def linear(name, x):
    y = x @ sd[name + ".weight"].T + sd[name + ".bias"]
    y.retain_grad()
    taps[name] = (x.detach(), y)
    return y


h = x_enc
for i in range(8):
    if i == 4:
        h = torch.cat([h, x_enc], -1)
    h = torch.relu(linear(f"pts_layers.{i}", h))
sigma = torch.relu(linear("sigma_layer", h))[:, 0].view(R, S)
feat = linear("feature_layer", h)
hv = torch.relu(linear("view_layer", torch.cat([feat, d_enc], -1)))
rgb = torch.sigmoid(linear("rgb_layer", hv)).view(R, S, 3)
delta = torch.cat([z[:, 1:] - z[:, :-1], torch.full((R, 1), 1e10, device=dev)], -1) * d.norm(dim=-1, keepdim=True)
alpha = 1 - torch.exp(-sigma * delta)
T = torch.cumprod(torch.cat([torch.ones(R, 1, device=dev), 1 - alpha + 1e-10], -1), -1)[:, :-1]
w = alpha * T
pred = (w[..., None] * rgb).sum(1) + (1 - w.sum(1, keepdim=True))
((pred - target) ** 2).mean().backward()


# ---- minifloat quantisers (round to nearest even, saturating, subnormals kept) ----
def minifloat(x, E, M, bias=None, max_val=None):
    bias = (1 << (E - 1)) - 1 if bias is None else bias
    emin = 1 - bias
    emax = (1 << E) - 1 - bias                      # OCP formats below use every exponent code for finite values
    if max_val is None:
        max_val = 2.0 ** emax * (2 - 2.0 ** -M)
    ax = x.abs().clamp_min(1e-45)
    e = torch.floor(torch.log2(ax)).clamp(emin, emax)
    step = torch.exp2(e - M)
    q = torch.round(ax / step) * step
    return torch.sign(x) * q.clamp_max(max_val)


FMT = {"e4m3": dict(E=4, M=3, max_val=448.0), "e5m2": dict(E=5, M=2, max_val=57344.0),
       "e3m2": dict(E=3, M=2, max_val=28.0), "e2m3": dict(E=2, M=3, max_val=7.5), "e2m1": dict(E=2, M=1, max_val=6.0)}


def q_global(x, fmt, amax_to=None):
    """one power-of-two scale for the whole tensor: the largest magnitude lands in [amax_to/2, amax_to) (None: unit scale)"""
    if amax_to is None:
        return minifloat(x, **FMT[fmt])
    s = 2.0 ** torch.ceil(torch.log2(x.abs().max() / amax_to))
    return minifloat(x / s, **FMT[fmt]) * s


def q_block(x, fmt, block=32):
    """MX: one E8M0 scale per (feature, 32 consecutive samples); x is [samples, features]"""
    n, f = x.shape
    xb = x.view(n // block, block, f)
    amax = xb.abs().amax(1, keepdim=True).clamp_min(2.0 ** -126)
    emax_elem = np.floor(np.log2(FMT[fmt]["max_val"]))
    s = torch.exp2(torch.floor(torch.log2(amax)) - emax_elem)
    return (minifloat(xb / s, **FMT[fmt]) * s).view(n, f)


SCHEMES = {
    "cur": (lambda a: q_global(a, "e4m3"), lambda g: q_global(g, "e5m2", 128.0)),
    "mx8": (lambda a: q_block(a, "e4m3"), lambda g: q_block(g, "e5m2")),
    "mx6": (lambda a: q_block(a, "e2m3"), lambda g: q_block(g, "e3m2")),
    "mx6b": (lambda a: q_block(a, "e3m2"), lambda g: q_block(g, "e3m2")),
    "mx4": (lambda a: q_block(a, "e2m1"), lambda g: q_block(g, "e2m1")),
    "mx4h": (lambda a: q_block(a, "e2m1"), lambda g: q_global(g, "e5m2", 128.0)),
}
print(f"{'weight gradient':24s}" + "".join(f"{k:>18s}" for k in SCHEMES) + "      (cosine / relative L2 error against fp32)")
tot = {k: [0.0, 0.0, 0.0] for k in SCHEMES}
with torch.no_grad():
    for name, (x, y) in taps.items():
        g = y.grad
        ref = g.T @ x
        row = f"{name + '.weight':24s}"
        for k, (qa, qg) in SCHEMES.items():
            est = qg(g).T @ qa(x)
            cos = float((est * ref).sum() / (est.norm() * ref.norm()))
            rel = float((est - ref).norm() / ref.norm())
            row += f"   {cos:.5f}/{rel:.4f}"
            tot[k][0] += float((est * ref).sum()); tot[k][1] += float(est.norm() ** 2); tot[k][2] += float(ref.norm() ** 2)
        print(row, flush=True)
print(f"{'all weights':24s}" + "".join(f"   {t[0] / np.sqrt(t[1] * t[2]):.5f}        " for t in tot.values()))
