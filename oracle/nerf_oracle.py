"""CPU oracle for the NeRF volumetric-rendering hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain fp32 PyTorch-CPU restatement of the reference algorithm
(CV-Project2025/Project-NeRF).  It is the *checker* for the HIP kernels and the
`cpu_baseline` leg of bench.py.  Nothing under ``project-nerf_amd/`` may import
it: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg are allowed to.  It is never the thing that is shipped or
measured as the product.

Pinning
-------
* Rows a1-a6, a9-a12 of SURVEY.md §8 (sampling, ray points, occupancy mask,
  Fourier encoding, the 8x256 decoder, alpha compositing, render_rays, density
  grid update) are PINNED: ``tests/golden/make_golden.py`` imports the reference
  itself in the build container and stores its outputs under ``tests/golden/``;
  ``tests/test_oracle_golden.py`` checks this file against them (bit-exact for
  sampling / voxel indices, <=1e-6 for floating point).
* Rows a7-a8 (multiresolution hash grid, fully-fused tiny MLPs) live in the
  third-party ``tinycudann`` module, which is not vendored, not version-pinned
  and not installable offline (reference ``src/embeddings.py:57``,
  ``src/decoders.py:107``).  Those functions restate the published Instant-NGP
  algorithm (Mueller et al. 2022, section 3) and are **parity unpinned**.

Every function cites the reference file:line it follows.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# ----------------------------------------------------------------------------
# a1  stratified sampling                       reference src/renderer.py:186-201
# ----------------------------------------------------------------------------

def stratified_depths(near: float, far: float, n_samples: int, n_rays: int,
                      perturb: bool, u: Optional[Tensor] = None) -> Tensor:
    """z[R,S] along each ray.  ``u`` is the uniform jitter draw ([R,S], row
    major, same order as ``torch.rand(z.shape)`` in renderer.py:198); passing it
    in makes the perturbed path reproducible for the HIP side."""
    t = torch.linspace(0.0, 1.0, steps=n_samples)
    z = near * (1.0 - t) + far * t                       # renderer.py:190
    z = z.expand(n_rays, n_samples)
    if perturb:
        centre = 0.5 * (z[:, 1:] + z[:, :-1])            # renderer.py:195
        hi = torch.cat([centre, z[:, -1:]], dim=-1)
        lo = torch.cat([z[:, :1], centre], dim=-1)
        if u is None:
            u = torch.rand(z.shape)
        z = lo + (hi - lo) * u                           # renderer.py:199
    return z


# ----------------------------------------------------------------------------
# a2  ray -> sample points                      reference src/renderer.py:291-300
# ----------------------------------------------------------------------------

def ray_points(rays_o: Tensor, rays_d: Tensor, z: Tensor) -> Tuple[Tensor, Tensor]:
    """pts[R*S,3] = o + d*z (mul then add), dirs[R*S,3] = d/|d| per sample."""
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z[..., None]
    unit = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    unit = unit[:, None, :].expand(-1, z.shape[1], -1)
    return pts.reshape(-1, 3), unit.reshape(-1, 3)


# ----------------------------------------------------------------------------
# a3  occupancy lookup                          reference src/renderer.py:134-166
# ----------------------------------------------------------------------------

def voxel_index(pts: Tensor, bound: float, resolution: int) -> Tensor:
    """int64 voxel index per axis; the scale is a Python double that torch
    demotes to fp32 before the multiply; ``.long()`` truncates toward zero."""
    scale = resolution / (2 * bound)                     # renderer.py:32
    return ((pts + bound) * scale).long()                # renderer.py:145


def active_mask(pts: Tensor, binary_grid: Tensor, bound: float) -> Tensor:
    res = binary_grid.shape[0]
    idx = voxel_index(pts, bound, res)
    ok = (idx >= 0).all(dim=-1) & (idx < res).all(dim=-1)
    out = torch.zeros(pts.shape[0], dtype=torch.bool)
    sel = idx[ok]
    if sel.shape[0] > 0:
        out[ok] = binary_grid[sel[:, 0], sel[:, 1], sel[:, 2]]
    return out


# ----------------------------------------------------------------------------
# a5  Fourier features                          reference src/embeddings.py:22-32
# ----------------------------------------------------------------------------

def fourier_encode(x: Tensor, n_freq: int) -> Tensor:
    """[x | sin(x 2^0 pi) | cos(x 2^0 pi) | sin(x 2^1 pi) | ...]; the argument is
    rounded twice in fp32, (x*f) then (*pi as double demoted to fp32)."""
    if n_freq == 0:
        return x
    bands = 2.0 ** torch.linspace(0.0, n_freq - 1, steps=n_freq)   # embeddings.py:15
    parts = [x]
    for f in bands:
        arg = x * f * np.pi
        parts.append(torch.sin(arg))
        parts.append(torch.cos(arg))
    return torch.cat(parts, dim=-1)


# ----------------------------------------------------------------------------
# a6  8x256 density+colour decoder              reference src/decoders.py:37-87
# ----------------------------------------------------------------------------

def nerf_param_shapes(pos_dim: int = 63, dir_dim: int = 27, hidden: int = 256,
                      n_layers: int = 8, skip: int = 4, view_dim: int = 128
                      ) -> List[Tuple[str, Tuple[int, ...]]]:
    """Parameter names/shapes exactly as in the reference state_dict
    (decoder.* keys), in registration order."""
    out = []
    for i in range(n_layers):
        k = pos_dim if i == 0 else hidden
        if i == skip:
            k += pos_dim
        out.append((f"pts_layers.{i}.weight", (hidden, k)))
        out.append((f"pts_layers.{i}.bias", (hidden,)))
    out += [("sigma_layer.weight", (1, hidden)), ("sigma_layer.bias", (1,)),
            ("feature_layer.weight", (hidden, hidden)), ("feature_layer.bias", (hidden,)),
            ("view_layer.weight", (view_dim, hidden + dir_dim)), ("view_layer.bias", (view_dim,)),
            ("rgb_layer.weight", (3, view_dim)), ("rgb_layer.bias", (3,))]
    return out


def nerf_init_params(seed: int = 0, **kw) -> Dict[str, Tensor]:
    """nn.Linear default init (kaiming-uniform a=sqrt(5) -> U(-1/sqrt(k), 1/sqrt(k))
    for weight and bias) drawn from one seeded CPU generator."""
    g = torch.Generator().manual_seed(seed)
    params = {}
    shapes = nerf_param_shapes(**kw)
    fan_in = 1
    for name, shape in shapes:
        if name.endswith("weight"):
            fan_in = shape[1]
        b = 1.0 / math.sqrt(fan_in)
        params[name] = (torch.rand(shape, generator=g) * 2.0 - 1.0) * b
    return params


def nerf_decoder(params: Dict[str, Tensor], x_enc: Tensor, d_enc: Tensor,
                 n_layers: int = 8, skip: int = 4,
                 keep: Optional[dict] = None) -> Tuple[Tensor, Tensor]:
    """rgb[N,3] (sigmoid), sigma[N,1] (relu).  At the skip layer the input is
    cat([h, x]) -- hidden first, then the encoding (decoders.py:72-73)."""
    h = x_enc
    for i in range(n_layers):
        if i == skip:
            h = torch.cat([h, x_enc], dim=-1)
        h = F.relu(F.linear(h, params[f"pts_layers.{i}.weight"], params[f"pts_layers.{i}.bias"]))
        if keep is not None:
            keep[f"h{i}"] = h
    sigma = F.relu(F.linear(h, params["sigma_layer.weight"], params["sigma_layer.bias"]))
    feat = F.linear(h, params["feature_layer.weight"], params["feature_layer.bias"])
    hv = F.relu(F.linear(torch.cat([feat, d_enc], dim=-1),
                         params["view_layer.weight"], params["view_layer.bias"]))
    rgb = torch.sigmoid(F.linear(hv, params["rgb_layer.weight"], params["rgb_layer.bias"]))
    if keep is not None:
        keep["feat"] = feat
        keep["hv"] = hv
    return rgb, sigma


def nerf_field(params: Dict[str, Tensor], pts: Tensor, dirs: Tensor,
               l_pos: int = 10, l_dir: int = 4) -> Tuple[Tensor, Tensor]:
    """NeuralField.forward for mode part2_nerf (core.py:354-359)."""
    return nerf_decoder(params, fourier_encode(pts, l_pos), fourier_encode(dirs, l_dir))


# ----------------------------------------------------------------------------
# a9  alpha compositing                         reference src/renderer.py:204-237
# ----------------------------------------------------------------------------

def composite(rgb: Tensor, sigma: Tensor, z: Tensor, rays_d: Tensor,
              bg: Optional[Tensor] = None, return_weights: bool = False):
    """rgb[R,S,3], sigma[R,S], z[R,S], rays_d[R,3] -> rgb_map[R,3], depth[R], acc[R].
    T_i = prod_{j<i}(1 - alpha_j + 1e-10); last interval is 1e10 long."""
    step = z[:, 1:] - z[:, :-1]
    step = torch.cat([step, torch.full_like(step[:, :1], 1e10)], dim=-1)
    step = step * torch.norm(rays_d[:, None, :], dim=-1)
    alpha = 1.0 - torch.exp(-sigma * step)
    trans = torch.cumprod(
        torch.cat([torch.ones((alpha.shape[0], 1)), 1.0 - alpha + 1e-10], dim=-1), dim=-1)[:, :-1]
    w = alpha * trans
    rgb_map = torch.sum(w[..., None] * rgb, dim=-2)
    depth = torch.sum(w * z, dim=-1)
    acc = torch.sum(w, dim=-1)
    if bg is not None:
        if bg.dim() == 1:
            bg = bg.unsqueeze(0)
        rgb_map = rgb_map + (1.0 - acc)[..., None] * bg
    if return_weights:
        return rgb_map, depth, acc, w
    return rgb_map, depth, acc


# ----------------------------------------------------------------------------
# a10 render_rays (static field)                reference src/renderer.py:240-384
# ----------------------------------------------------------------------------

def render_rays(field, rays_o: Tensor, rays_d: Tensor, near: float, far: float,
                n_samples: int, perturb: bool, u: Optional[Tensor] = None,
                binary_grid: Optional[Tensor] = None, grid_bound: float = 1.5,
                white_bkgd: bool = True, bg: Optional[Tensor] = None):
    """``field(pts, dirs) -> (rgb[N,3], sigma[N,1])``.  With ``binary_grid`` only
    active samples are queried; the rest contribute sigma = 0, rgb = 0
    (renderer.py:303-343); if nothing is active sample 0 is forced on."""
    n_rays = rays_o.shape[0]
    if bg is None:
        bg = torch.ones(3) if white_bkgd else torch.zeros(3)
    z = stratified_depths(near, far, n_samples, n_rays, perturb, u)
    pts, dirs = ray_points(rays_o, rays_d, z)
    if binary_grid is not None:
        m = active_mask(pts, binary_grid, grid_bound)
        if not m.any():
            m = m.clone()
            m[0] = True
        c_rgb, c_sig = field(pts[m], dirs[m])
        rgb = c_rgb.new_zeros(pts.shape[0], 3, dtype=torch.float32)
        sig = c_sig.new_zeros(pts.shape[0], 1, dtype=torch.float32)
        rgb[m] = c_rgb.float()
        sig[m] = c_sig.float()
    else:
        rgb, sig = field(pts, dirs)
    rgb = rgb.float().view(n_rays, n_samples, 3)
    sig = sig.float().view(n_rays, n_samples)
    return composite(rgb, sig, z, rays_d, bg)


def render_image(field, rays_o: Tensor, rays_d: Tensor, near: float, far: float,
                 n_samples: int, chunk: int, white_bkgd: bool) -> Tensor:
    """Chunked full-image render, no jitter (renderer.py:387-418)."""
    h, w = rays_o.shape[:2]
    o = rays_o.reshape(-1, 3)
    d = rays_d.reshape(-1, 3)
    out = []
    for i in range(0, o.shape[0], chunk):
        out.append(render_rays(field, o[i:i + chunk], d[i:i + chunk], near, far,
                               n_samples, False, white_bkgd=white_bkgd)[0])
    return torch.cat(out, dim=0).view(h, w, 3)


# ----------------------------------------------------------------------------
# a12 occupancy-grid refresh                    reference src/renderer.py:35-132
# ----------------------------------------------------------------------------

def grid_lattice(bound: float, resolution: int) -> Tensor:
    """[res^3,3] lattice of linspace(-b,b,res) nodes, 'ij' order, x slowest."""
    ax = torch.linspace(-bound, bound, resolution)
    gx, gy, gz = torch.meshgrid(ax, ax, ax, indexing="ij")
    return torch.stack([gx, gy, gz], dim=-1).reshape(-1, 3)


def density_grid_update(field, bound: float, resolution: int, threshold: float,
                        prev_grid: Optional[Tensor] = None, decay: float = 1.0,
                        dynamic: bool = False, batch: int = 2 ** 18):
    """Static fields overwrite the grid; dynamic ones keep max(prev*decay, cur)
    (renderer.py:122-125).  Returns (grid, binary_grid, active_ratio)."""
    pts = grid_lattice(bound, resolution)
    vals = []
    for i in range(0, pts.shape[0], batch):
        p = pts[i:i + batch]
        _, s = field(p, torch.zeros_like(p))
        vals.append(s.squeeze(-1))
    cur = torch.cat(vals, dim=0).reshape(resolution, resolution, resolution)
    grid = torch.maximum(prev_grid * decay, cur) if dynamic else cur
    binary = grid > threshold
    return grid, binary, binary.float().mean().item()


def should_update(step: int, interval: int = 16, warmup: int = 0) -> bool:
    """renderer.py:168-183."""
    return step >= warmup and step % interval == 0


# ----------------------------------------------------------------------------
# hierarchical fine sampling -- NOT in the reference (no sample_pdf / searchsorted anywhere in the
# tree); opt-in extension named by BASELINE.json.  Restates Mildenhall et al. 2020, section 5.2.
# PARITY UNPINNED.
# ----------------------------------------------------------------------------

def sample_pdf(z: Tensor, weights: Tensor, n_fine: int, u: Optional[Tensor] = None) -> Tensor:
    """z [R,S] coarse depths, weights [R,S] -> merged, sorted depths [R, S+n_fine]."""
    bins = 0.5 * (z[:, 1:] + z[:, :-1])
    w = weights[:, 1:-1] + 1e-5
    pdf = w / w.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)          # [R, S-1]
    if u is None:
        u = torch.linspace(0.0, 1.0, n_fine).expand(z.shape[0], n_fine)
    u = u.contiguous()
    inds = torch.searchsorted(cdf.contiguous(), u, right=True)
    below = (inds - 1).clamp(min=0)
    above = inds.clamp(max=cdf.shape[-1] - 1)
    c0, c1 = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    b0, b1 = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = c1 - c0
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    fine = b0 + (u - c0) / denom * (b1 - b0)
    return torch.sort(torch.cat([z, fine], -1), -1).values


# ----------------------------------------------------------------------------
# a8  multiresolution hash grid  (tinycudann -- PARITY UNPINNED)
#     call site reference src/embeddings.py:60-89; algorithm Instant-NGP sec. 3
# ----------------------------------------------------------------------------

HASH_PRIMES = (1, 2654435761, 805459861)


@dataclass
class HashLevel:
    scale: float      # fp32 value; position on the level = x01*scale + 0.5
    res: int          # vertices per axis
    size: int         # table entries on this level
    offset: int       # first entry in the flat table
    dense: bool       # True -> x + y*res + z*res^2 ; False -> spatial hash mod size


def hash_grid_levels(n_levels: int = 16, log2_hashmap_size: int = 19,
                     base_resolution: int = 16, per_level_scale: float = 1.5
                     ) -> List[HashLevel]:
    """Per-level table.  scale_l = base*s^l - 1 is evaluated in float64 on the
    host and rounded once to fp32; res_l = ceil(scale_l)+1; a level is stored
    densely when res^3 fits in the hash-map budget; sizes are padded to a
    multiple of 8 entries.  This definition is the build's own choice (the
    third-party source is absent; SURVEY.md section 2.1 dagger note) and is
    shared verbatim by the HIP kernel so indices agree bit for bit."""
    budget = 1 << log2_hashmap_size
    out, off = [], 0
    for l in range(n_levels):
        s64 = base_resolution * (per_level_scale ** l) - 1.0
        res = int(math.ceil(round(s64, 9))) + 1
        n = res ** 3
        dense = n <= budget
        size = ((n + 7) // 8) * 8 if dense else budget
        out.append(HashLevel(float(np.float32(s64)), res, size, off, dense))
        off += size
    return out


def hash_grid_entries(levels: Sequence[HashLevel]) -> int:
    return levels[-1].offset + levels[-1].size


def hash_grid_index(levels: Sequence[HashLevel], x01: Tensor) -> Tuple[Tensor, Tensor]:
    """x01[N,3] in [0,1] -> (idx[N,L,8] int64 absolute entry, w[N,L,8] fp32 trilinear
    weights).  Corner c has bit0->x, bit1->y, bit2->z."""
    n = x01.shape[0]
    idx = torch.empty(n, len(levels), 8, dtype=torch.int64)
    wts = torch.empty(n, len(levels), 8, dtype=torch.float32)
    for li, lv in enumerate(levels):
        pos = x01 * np.float32(lv.scale) + np.float32(0.5)
        cell = torch.floor(pos)
        frac = pos - cell
        cell = cell.to(torch.int64)
        for c in range(8):
            off = torch.tensor([(c >> 0) & 1, (c >> 1) & 1, (c >> 2) & 1])
            g = cell + off
            wx = torch.where(off[0] == 1, frac[:, 0], 1.0 - frac[:, 0])
            wy = torch.where(off[1] == 1, frac[:, 1], 1.0 - frac[:, 1])
            wz = torch.where(off[2] == 1, frac[:, 2], 1.0 - frac[:, 2])
            wts[:, li, c] = wx * wy * wz
            if lv.dense:
                e = g[:, 0] + g[:, 1] * lv.res + g[:, 2] * lv.res * lv.res
                e = e % lv.size
            else:
                u = g & 0xFFFFFFFF
                hsh = (u[:, 0] * HASH_PRIMES[0]) & 0xFFFFFFFF
                hsh = hsh ^ ((u[:, 1] * HASH_PRIMES[1]) & 0xFFFFFFFF)
                hsh = hsh ^ ((u[:, 2] * HASH_PRIMES[2]) & 0xFFFFFFFF)
                e = hsh % lv.size
            idx[:, li, c] = e + lv.offset
    return idx, wts


def hash_normalise(x: Tensor, bound: float) -> Tensor:
    """HashRepresentation.forward pre-step (embeddings.py:86-87)."""
    return ((x + bound) / (2 * bound)).clamp(0.0, 1.0)


def hash_encode(levels: Sequence[HashLevel], table: Tensor, x01: Tensor) -> Tensor:
    """table[E,F] fp32 -> features[N, L*F] (fp32; the product rounds to bf16)."""
    idx, w = hash_grid_index(levels, x01)
    feat = table[idx]                                    # [N,L,8,F]
    return (feat * w[..., None]).sum(dim=2).reshape(x01.shape[0], -1)


# ----------------------------------------------------------------------------
# a7  fully-fused tiny MLPs  (tinycudann -- PARITY UNPINNED)
#     call sites reference src/decoders.py:111-134, 149-160
# ----------------------------------------------------------------------------

def tiny_mlp(weights: Sequence[Tensor], x: Tensor, out_act: Optional[str] = None) -> Tensor:
    """Bias-free MLP, ReLU between layers, weights[i] is [out_i, in_i]."""
    h = x
    for i, w in enumerate(weights):
        h = F.linear(h, w)
        if i + 1 < len(weights):
            h = F.relu(h)
    if out_act == "sigmoid":
        h = torch.sigmoid(h)
    return h


def instant_decoder(sigma_w: Sequence[Tensor], color_w: Sequence[Tensor],
                    x_enc: Tensor, d_enc: Tensor) -> Tuple[Tensor, Tensor]:
    """InstantNeRFDecoder.forward (decoders.py:136-162): sigma = softplus(h0-5);
    the colour net sees all 16 geometry channels followed by the direction code."""
    h = tiny_mlp(sigma_w, x_enc)
    sigma = F.softplus(h[..., 0:1] - 5.0)
    rgb = tiny_mlp(color_w, torch.cat([h, d_enc], dim=-1), out_act="sigmoid")
    return rgb, sigma


# ----------------------------------------------------------------------------
# a14 optimiser maths                           reference run.py:307, 546-550
# ----------------------------------------------------------------------------

def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
              beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
              weight_decay: float = 0.0) -> None:
    """torch.optim.Adam / AdamW (decoupled decay) single-tensor update, in place."""
    if weight_decay != 0.0:
        p.mul_(1.0 - lr * weight_decay)
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def cosine_lr(base_lr: float, eta_min: float, step: int, t_max: int) -> float:
    """Closed form of CosineAnnealingLR after ``step`` scheduler steps."""
    return eta_min + (base_lr - eta_min) * (1.0 + math.cos(math.pi * step / t_max)) / 2.0


def psnr_from_mse(mse: float) -> float:
    """utils.py:12-22."""
    return 10.0 * np.log10(1.0 / mse)


# ----------------------------------------------------------------------------
# camera rays (next row f1)                     reference src/dataset.py:78-122
# ----------------------------------------------------------------------------

def camera_rays(c2w: Tensor, h: int, w: int, focal: float, scene_scale: float = 1.0):
    """Pixel-centre-free pinhole model: x=(i-W/2)/f, y=-(j-H/2)/f, z=-1, rotated by
    c2w[:3,:3] and normalised; origin = c2w[:3,3]*scene_scale."""
    jj, ii = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    d = torch.stack([(ii - w * 0.5) / focal, -(jj - h * 0.5) / focal,
                     -torch.ones_like(ii)], dim=-1).reshape(-1, 3)
    rd = torch.matmul(d, c2w[:3, :3].T).reshape(h, w, 3)
    rd = rd / torch.norm(rd, dim=-1, keepdim=True)
    ro = c2w[:3, 3].expand_as(rd)
    if scene_scale != 1.0:
        ro = ro * scene_scale
    return ro, rd
