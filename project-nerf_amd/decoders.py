"""Decoders (reference src/decoders.py).  NeRFDecoder keeps the reference's parameter names so
checkpoints interchange; its arithmetic is the fused bf16-MFMA chain (ops.decoder)."""
import torch
import torch.nn as nn

from . import ops
from .abstract import BaseDecoder
from .embeddings import ParamHolder


class StandardMLP(BaseDecoder):
    """Part 1 image-fit MLP (reference src/decoders.py:6-26): Linear/ReLU x num_layers, Linear, Sigmoid.
    Outside the volumetric hot path (SURVEY 8 scope): plain library GEMMs through torch.nn on the
    device; only its Fourier features go through the HIP kernel."""

    def __init__(self, input_dim, hidden_dim=256, output_dim=3, num_layers=3):
        super().__init__()
        layers = [nn.Linear(input_dim, hidden_dim), nn.ReLU()]
        for _ in range(num_layers - 1):
            layers += [nn.Linear(hidden_dim, hidden_dim), nn.ReLU()]
        layers += [nn.Linear(hidden_dim, output_dim), nn.Sigmoid()]
        self.net = nn.Sequential(*layers)

    def forward(self, x):
        return self.net(x)


class NeRFDecoder(BaseDecoder):
    """8x256 MLP with skip at layer 4, sigma / feature heads, 128-wide view branch
    (reference src/decoders.py:29-87).  The HIP kernels are specialised for the reference
    defaults (pos 63, dir 27, hidden 256, 8 layers, skip 4, view 128)."""

    def __init__(self, pos_dim, dir_dim, hidden_dim=256, num_layers=8, skip_layer=4, view_dim=128):
        super().__init__()
        if (pos_dim, dir_dim, hidden_dim, num_layers, skip_layer, view_dim) != (63, 27, 256, 8, 4, 128):
            raise NotImplementedError(
                "libnerf_hip is compiled for L_embed 10 / L_embed_dir 4 / hidden 256 / 8 layers / skip 4 / "
                f"view 128; got pos {pos_dim}, dir {dir_dim}, hidden {hidden_dim}, layers {num_layers}, "
                f"skip {skip_layer}, view {view_dim}")
        self.skip_layer = skip_layer
        layers = []
        for i in range(num_layers):
            k = pos_dim if i == 0 else hidden_dim
            if i == skip_layer:
                k += pos_dim
            layers.append(nn.Linear(k, hidden_dim))
        self.pts_layers = nn.ModuleList(layers)
        self.sigma_layer = nn.Linear(hidden_dim, 1)
        self.feature_layer = nn.Linear(hidden_dim, hidden_dim)
        self.view_layer = nn.Linear(hidden_dim + dir_dim, view_dim)
        self.rgb_layer = nn.Linear(view_dim, 3)
        self._packed = None
        self._packed_version = None

    def flat_parameters(self):
        """One fp32 vector in registration (= state_dict) order; autograd splits its gradient back."""
        return torch.cat([p.reshape(-1) for p in self.parameters()])

    def packed_weights(self):
        """bf16 fragment streams, rebuilt whenever any parameter was modified in place."""
        version = tuple(p._version for p in self.parameters())
        if self._packed is None or version != self._packed_version or self._packed.device != self.rgb_layer.bias.device:
            with torch.no_grad():
                self._packed = ops.mlp_pack(self.flat_parameters())
            self._packed_version = version
        return self._packed

    def field(self, pts, dirs, z=None):
        """Fused encode + decode.  Point mode: pts/dirs [N,3]; ray mode: rays_o/rays_d [R,3] + z [R,S]."""
        rgb, sigma = ops.decoder(self.flat_parameters(), self.packed_weights(), pts, dirs, z)
        return rgb, sigma.unsqueeze(-1)

    def forward(self, x, d):
        raise NotImplementedError(
            "NeRFDecoder consumes raw coordinates through the fused kernel: call NeuralField(x, d) or "
            "NeRFDecoder.field(pts, dirs); separately encoded inputs are not a supported entry point")


class InstantNeRFDecoder(BaseDecoder):
    """reference src/decoders.py:90-162: sigma-net 32->64->16, colour-net (16+27)->64->64->3."""

    def __init__(self, pos_dim, dir_dim, hidden_dim=64):
        super().__init__()
        if (pos_dim, dir_dim, hidden_dim) != (32, 27, 64):
            raise NotImplementedError("libnerf_hip is compiled for pos 32 / dir 27 / hidden 64")

        def xavier(rows, cols, fan_in, fan_out):
            return (torch.rand(rows, cols) * 2 - 1) * (6.0 / (fan_in + fan_out)) ** 0.5
        s = torch.cat([xavier(64, 32, 32, 64).reshape(-1), xavier(16, 64, 64, 16).reshape(-1)])
        w1 = xavier(64, 48, 43, 64)
        w1[:, 43:] = 0
        w3 = xavier(16, 64, 64, 3)
        w3[3:] = 0
        c = torch.cat([w1.reshape(-1), xavier(64, 64, 64, 64).reshape(-1), w3.reshape(-1)])
        self.sigma_net = ParamHolder(s)
        self.color_net = ParamHolder(c)
        self._packed, self._version = None, None

    def flat_parameters(self):
        return torch.cat([self.sigma_net.params, self.color_net.params])

    def packed_weights(self):
        v = (self.sigma_net.params._version, self.color_net.params._version)
        if self._packed is None or v != self._version or self._packed.device != self.sigma_net.params.device:
            with torch.no_grad():
                self._packed = ops.imlp_pack(self.flat_parameters())
            self._version = v
        return self._packed

    def forward(self, x_enc, d_enc):
        raise NotImplementedError("the Instant decoder runs fused with the hash encoding: call NeuralField(x, d)")
