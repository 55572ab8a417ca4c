"""Training / rendering engine for the vanilla NeRF field (mode part2_nerf).

Host-side counterpart of the reference's per-step work in run_part2 (run.py:312-338) and of
render_image (src/renderer.py:387-418): it owns the flat fp32 parameter vector (reference
state_dict order), Adam moments, the packed bf16 weight streams and every workspace, and issues
the kernel sequence of one step on the current HIP stream:

    sample -> decoder fwd (+stash) -> composite fwd -> MSE grad -> composite bwd
           -> decoder dgrad chain -> wgrad -> [RCCL all-reduce] -> Adam -> repack

No arithmetic happens in PyTorch except the uniform jitter draw and the 3-element-per-ray MSE
residual (the reference's nn.MSELoss, run.py:308,334).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import ops

Tensor = torch.Tensor


def param_shapes():
    """(name, shape) of NeRFDecoder parameters in reference registration order
    (src/decoders.py:49-66; keys as in state_dict under the ``decoder.`` prefix)."""
    out = []
    for i in range(8):
        k = 63 if i == 0 else (319 if i == 4 else 256)
        out += [(f"pts_layers.{i}.weight", (256, k)), (f"pts_layers.{i}.bias", (256,))]
    out += [("sigma_layer.weight", (1, 256)), ("sigma_layer.bias", (1,)),
            ("feature_layer.weight", (256, 256)), ("feature_layer.bias", (256,)),
            ("view_layer.weight", (128, 283)), ("view_layer.bias", (128,)),
            ("rgb_layer.weight", (3, 128)), ("rgb_layer.bias", (3,))]
    return out


def default_init(seed: int = 0) -> Tensor:
    """nn.Linear default init, U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weights and biases, drawn on
    the host from one seeded generator; returns the flat fp32 vector."""
    g = torch.Generator().manual_seed(seed)
    parts, fan_in = [], 1
    for name, shape in param_shapes():
        if name.endswith("weight"):
            fan_in = shape[1]
        parts.append(((torch.rand(shape, generator=g) * 2 - 1) / fan_in ** 0.5).reshape(-1))
    return torch.cat(parts)


def flatten_state_dict(sd: Dict[str, Tensor], prefix: str = "decoder.") -> Tensor:
    return torch.cat([sd[prefix + k].reshape(-1).float() for k, _ in param_shapes()])


def unflatten(flat: Tensor, prefix: str = "decoder.") -> Dict[str, Tensor]:
    out, off = {}, 0
    for k, shape in param_shapes():
        n = 1
        for s in shape:
            n *= s
        out[prefix + k] = flat[off:off + n].view(shape)
        off += n
    return out


class VanillaNerfEngine:
    def __init__(self, params: Optional[Tensor] = None, device: str = "cuda", lr: float = 5e-4,
                 near: float = 2.0, far: float = 6.0, white_bkgd: bool = True, seed: int = 0,
                 world_size: int = 1):
        self.device = torch.device(device)
        flat = default_init(seed) if params is None else params.detach().float().reshape(-1)
        assert flat.numel() == ops.MLP_PARAM_COUNT
        self.params = flat.to(self.device).contiguous()
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.grads = torch.empty_like(self.params)
        self.packed = torch.empty(ops.mlp_packed_bytes(), dtype=torch.uint8, device=self.device)
        self.lr, self.near, self.far = lr, near, far
        self.bg = (torch.ones(3) if white_bkgd else torch.zeros(3)).to(self.device)
        self.step_count = 0
        self.world_size = world_size
        self.grad_scale = torch.full((1,), 1.0 / world_size, device=self.device)
        self._ws: Dict[Tuple[str, int], Tensor] = {}
        self.repack()

    # -- weights ---------------------------------------------------------------
    def repack(self) -> None:
        ops.mlp_pack(self.params, self.packed)

    def state_dict(self, prefix: str = "decoder.") -> Dict[str, Tensor]:
        return {k: v.clone() for k, v in unflatten(self.params, prefix).items()}

    def load_state_dict(self, sd: Dict[str, Tensor], prefix: str = "decoder.") -> None:
        self.params.copy_(flatten_state_dict(sd, prefix).to(self.device))
        self.repack()

    def _buf(self, kind: str, nbytes: int) -> Tensor:
        key = (kind, nbytes)
        if key not in self._ws:
            self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws[key]

    # -- training step (reference run.py:314-338) ------------------------------
    def train_step(self, rays_o: Tensor, rays_d: Tensor, target: Tensor, n_samples: int = 64,
                   u: Optional[Tensor] = None, sync_grads=None) -> Tensor:
        R = rays_o.shape[0]
        n = R * n_samples
        if u is None:
            u = torch.rand(R, n_samples, device=self.device)
        z = ops.sample_rays(rays_o, rays_d, self.near, self.far, n_samples, u=u)
        stash = self._buf("stash", ops.mlp_stash_bytes(n))
        rgb, sigma = ops.mlp_fwd(self.packed, rays_o, rays_d, z, stash)
        rgb3 = rgb.view(R, n_samples, 3)
        sig2 = sigma.view(R, n_samples)
        pred, _, _, _, _ = ops.composite_fwd(rgb3, sig2, z, rays_d, self.bg)
        diff = pred - target
        loss = (diff * diff).mean()
        g_pred = diff * (2.0 / diff.numel())
        d_rgb, d_sigma, _ = ops.composite_bwd(rgb3, sig2, z, rays_d, self.bg, None, g_pred, None, None, None)
        ops.mlp_bwd(self.packed, stash, rgb, sigma, d_rgb.view(n, 3), d_sigma.view(n), self.grads,
                    self._buf("bwd", ops.mlp_bwd_workspace_bytes(n)))
        if sync_grads is not None:
            sync_grads(self.grads)            # RCCL all-reduce (sum); averaged by grad_scale below
        self.step_count += 1
        ops.adam_step(self.params, self.grads, self.exp_avg, self.exp_avg_sq, self.step_count, self.lr,
                      grad_scale=self.grad_scale if self.world_size > 1 else None)
        self.repack()
        return loss

    # -- inference (reference render_image, src/renderer.py:387-418) ------------
    @torch.no_grad()
    def render_rays(self, rays_o: Tensor, rays_d: Tensor, n_samples: int, u: Optional[Tensor] = None):
        z = ops.sample_rays(rays_o, rays_d, self.near, self.far, n_samples, u=u)
        rgb, sigma = ops.mlp_fwd(self.packed, rays_o, rays_d, z)
        R = rays_o.shape[0]
        c, depth, acc, _, _ = ops.composite_fwd(rgb.view(R, n_samples, 3), sigma.view(R, n_samples), z, rays_d, self.bg)
        return c, depth, acc

    @torch.no_grad()
    def render_image(self, rays_o: Tensor, rays_d: Tensor, n_samples: int, chunk: int = 65536) -> Tensor:
        shape = rays_o.shape[:-1]
        o, d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
        out = torch.empty(o.shape[0], 3, device=self.device)
        for i in range(0, o.shape[0], chunk):
            out[i:i + chunk] = self.render_rays(o[i:i + chunk], d[i:i + chunk], n_samples)[0]
        return out.view(*shape, 3)
