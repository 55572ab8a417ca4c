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
    (reference src/decoders.py:29-87).  The HIP kernels are specialised for the reference's
    architecture (hidden 256, 8 layers, skip 4, view 128) with up to L_embed 10 / L_embed_dir 4
    frequency bands: a narrower code is a PREFIX of the 63 / 27-wide one (src/embeddings.py:28-32), so its
    weight matrices are zero-padded to the compiled width and the extra code columns contribute nothing."""

    def __init__(self, pos_dim, dir_dim, hidden_dim=256, num_layers=8, skip_layer=4, view_dim=128):
        super().__init__()
        ok_dims = lambda dim, max_l: dim in [3 + 6 * l for l in range(max_l + 1)]
        # ``fused``: the shape the HIP chain kernels are compiled for.  Any other hidden_dim / num_layers /
        # skip_layer / view_dim / code width from the YAML (reference src/core.py:36-55) still works: the same
        # layers then run as library GEMMs on the GPU (torch -> hipBLASLt) around the HIP Fourier codes -- slower,
        # same parameters and state-dict keys, fp32 like the reference.
        self.fused = (hidden_dim, num_layers, skip_layer, view_dim) == (256, 8, 4, 128) and ok_dims(pos_dim, 10) and ok_dims(dir_dim, 4)
        if pos_dim < 1 or dir_dim < 1 or not 0 < skip_layer < max(num_layers, 1) + 1 or num_layers < 1:
            raise ValueError(f"NeRFDecoder: pos {pos_dim} / dir {dir_dim} columns, {num_layers} layers, skip {skip_layer}")
        self.pos_dim, self.dir_dim = pos_dim, dir_dim
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
        """One fp32 vector in registration (= state_dict) order with the compiled 63 / 27-wide code columns
        (narrower codes: zero-padded columns); autograd splits / slices its gradient back."""
        pad = torch.nn.functional.pad
        parts = []
        for name, p in self.named_parameters():
            if name in ("pts_layers.0.weight", f"pts_layers.{self.skip_layer}.weight") and self.pos_dim < 63:
                p = pad(p, (0, 63 - self.pos_dim))
            elif name == "view_layer.weight" and self.dir_dim < 27:
                p = pad(p, (0, 27 - self.dir_dim))
            parts.append(p.reshape(-1))
        return torch.cat(parts)

    def packed_weights(self):
        """bf16 fragment streams, rebuilt whenever any parameter was modified in place."""
        version = tuple(p._version for p in self.parameters())
        if self._packed is None or version != self._packed_version or self._packed.device != self.rgb_layer.bias.device:
            with torch.no_grad():
                self._packed = ops.mlp_pack(self.flat_parameters())
            self._packed_version = version
        return self._packed

    def _layers(self, x_enc, d_enc):
        """the reference's forward (src/decoders.py:68-87) layer by layer: library GEMMs, fp32"""
        h = x_enc
        for i, layer in enumerate(self.pts_layers):
            if i == self.skip_layer:
                h = torch.cat([h, x_enc], dim=-1)
            h = torch.relu(layer(h))
        sigma = torch.relu(self.sigma_layer(h))
        hv = torch.relu(self.view_layer(torch.cat([self.feature_layer(h), d_enc], dim=-1)))
        return torch.sigmoid(self.rgb_layer(hv)), sigma

    def field(self, pts, dirs, z=None):
        """Fused encode + decode.  Point mode: pts/dirs [N,3]; ray mode: rays_o/rays_d [R,3] + z [R,S]."""
        if not self.fused:
            if z is not None:                                  # ray mode: sample points and per-sample directions
                n_s = z.shape[1]
                pts = (pts[:, None, :] + dirs[:, None, :] * z[..., None]).reshape(-1, 3)
                dirs = dirs[:, None, :].expand(-1, n_s, -1).reshape(-1, 3)
            if (self.pos_dim - 3) % 6 or (self.dir_dim - 3) % 6:
                raise ValueError("NeRFDecoder.field: the code widths are not 3 + 6 L; encode first and call forward(x_enc, d_enc)")
            code = lambda v, dim: ops.fourier_encode(v.contiguous(), (dim - 3) // 6) if dim > 3 else v
            return self._layers(code(pts, self.pos_dim), code(dirs, self.dir_dim))
        rgb, sigma = ops.decoder(self.flat_parameters(), self.packed_weights(), pts, dirs, z)
        return rgb, sigma.unsqueeze(-1)

    def forward(self, x_enc, d_enc):
        """reference src/decoders.py:68-87: rgb [N,3], sigma [N,1] from the encoded position / direction.
        (NeuralField takes the fused path ``field`` -- the codes are then formed in registers.)"""
        if x_enc.shape[-1] != self.pos_dim or d_enc.shape[-1] != self.dir_dim:
            raise ValueError(f"expected x_enc [N,{self.pos_dim}] and d_enc [N,{self.dir_dim}]")
        if not self.fused:
            if x_enc.device.type != "cuda":
                raise ops._lib.NerfHipError("NeRFDecoder runs on a HIP device only (no CPU fallback)")
            return self._layers(x_enc, d_enc)
        pad = torch.nn.functional.pad
        rgb, sigma = ops.decoder_encoded(self.flat_parameters(), self.packed_weights(), pad(x_enc, (0, 63 - self.pos_dim)),
                                         pad(d_enc, (0, 27 - self.dir_dim)))
        return rgb, sigma.unsqueeze(-1)


def _pad16(n):
    return (n + 15) // 16 * 16


def tiny_mlp_shapes(n_in, n_out, hidden, n_hidden_layers):
    """[out, in] matrix shapes of a bias-free fully fused MLP in this build's flat ``params`` layout
    (include/nerf_hip.h): in / out widths rounded up to multiples of 16, matrices concatenated row-major."""
    widths = [_pad16(n_in)] + [hidden] * n_hidden_layers + [_pad16(n_out)]
    return [(widths[i + 1], widths[i]) for i in range(len(widths) - 1)]


def tiny_mlp_init(n_in, n_out, hidden, n_hidden_layers):
    parts = []
    shapes = tiny_mlp_shapes(n_in, n_out, hidden, n_hidden_layers)
    for i, (o, k) in enumerate(shapes):
        fan_in = n_in if i == 0 else k
        fan_out = n_out if i == len(shapes) - 1 else o
        w = (torch.rand(o, k) * 2 - 1) * (6.0 / (fan_in + fan_out)) ** 0.5
        if i == 0:
            w[:, n_in:] = 0
        if i == len(shapes) - 1:
            w[n_out:] = 0
        parts.append(w.reshape(-1))
    return torch.cat(parts)


def tiny_mlp_apply(params, shapes, x, n_out, sigmoid=False):
    """Bias-free MLP, ReLU between layers, pad inputs zero -- the tiny networks whose shapes the fused HIP
    kernels are not compiled for (Part 4: 88 -> 64 -> 64 -> 3 and 53 -> 64 -> 16) run as library GEMMs."""
    pad = shapes[0][1] - x.shape[-1]
    h = torch.nn.functional.pad(x.float(), (0, pad)) if pad else x.float()
    off = 0
    for i, (o, k) in enumerate(shapes):
        h = torch.nn.functional.linear(h, params[off:off + o * k].view(o, k))
        off += o * k
        if i + 1 < len(shapes):
            h = torch.relu(h)
    h = h[:, :n_out]
    return torch.sigmoid(h) if sigmoid else h


class InstantNeRFDecoder(BaseDecoder):
    """reference src/decoders.py:90-162: sigma-net pos->64->16, colour-net (16+dir)->64->64->3.
    pos 32 / dir 27 / hidden 64 (Part 2 Instant) runs on the fused HIP tiny-MLP kernels; the time-conditioned
    shape of Part 4 (pos 32 + 21) runs the same arithmetic as library GEMMs."""

    def __init__(self, pos_dim, dir_dim, hidden_dim=64):
        super().__init__()
        self.pos_dim, self.dir_dim, self.hidden_dim = pos_dim, dir_dim, hidden_dim
        self.fused = (pos_dim, dir_dim, hidden_dim) == (32, 27, 64)
        if not self.fused:
            self._s_shapes = tiny_mlp_shapes(pos_dim, 16, hidden_dim, 1)
            self._c_shapes = tiny_mlp_shapes(16 + dir_dim, 3, hidden_dim, 2)
            self.sigma_net = ParamHolder(tiny_mlp_init(pos_dim, 16, hidden_dim, 1))
            self.color_net = ParamHolder(tiny_mlp_init(16 + dir_dim, 3, hidden_dim, 2))
            return

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
        """reference src/decoders.py:136-162: sigma = softplus(h0 - 5), rgb = colour-net(cat([h16, d_enc]))."""
        if not self.fused:
            h = tiny_mlp_apply(self.sigma_net.params, self._s_shapes, x_enc, 16)
            sigma = torch.nn.functional.softplus(h[..., 0:1] - 5.0)
            rgb = tiny_mlp_apply(self.color_net.params, self._c_shapes, torch.cat([h, d_enc], dim=-1), 3, sigmoid=True)
            return rgb, sigma
        rgb, sigma = ops.instant_decoder_encoded(self.flat_parameters(), self.packed_weights(), x_enc, d_enc)
        return rgb, sigma.unsqueeze(-1)


class HashDeformationDecoder(BaseDecoder):
    """reference src/decoders.py:264-318: [hash features | time modulation] -> 64 -> 64 -> 3 (bias-free, ReLU),
    times the learnable ``displacement_scale`` (0.1 at start)."""

    def __init__(self, hash_dim, time_mod_dim, hidden_dim=64):
        super().__init__()
        self._shapes = tiny_mlp_shapes(hash_dim + time_mod_dim, 3, hidden_dim, 2)
        self.deform_net = ParamHolder(tiny_mlp_init(hash_dim + time_mod_dim, 3, hidden_dim, 2))
        self.displacement_scale = nn.Parameter(torch.tensor(0.1))

    def forward(self, hash_feat, time_mod):
        delta = tiny_mlp_apply(self.deform_net.params, self._shapes, torch.cat([hash_feat, time_mod], dim=-1), 3)
        return delta * self.displacement_scale


class TimeModulationNetwork(BaseDecoder):
    """reference src/decoders.py:321-371: Linear/ReLU stack on the time code, sigmoid gate; last layer
    xavier-uniform weights and bias -1."""

    def __init__(self, time_dim, output_dim=64, hidden_dim=64, num_layers=2):
        super().__init__()
        self.output_dim = output_dim
        layers, in_dim = [], time_dim
        for i in range(num_layers):
            out_dim = hidden_dim if i < num_layers - 1 else output_dim
            layers.append(nn.Linear(in_dim, out_dim))
            if i < num_layers - 1:
                layers.append(nn.ReLU())
            in_dim = out_dim
        self.net = nn.Sequential(*layers)
        nn.init.xavier_uniform_(layers[-1].weight)
        nn.init.constant_(layers[-1].bias, -1.0)

    def forward(self, time_feat):
        return torch.sigmoid(self.net(time_feat))


class DeformationNetwork(BaseDecoder):
    """reference src/decoders.py:165-195 (Part 3): displacement delta_x [N,3] from the encoded position and time;
    a plain nn.Linear / ReLU stack (library GEMMs), last layer initialised to near-zero output."""

    def __init__(self, pos_dim, time_dim, hidden_dim=128, num_layers=4):
        super().__init__()
        self.num_layers = num_layers
        layers = [nn.Linear(pos_dim + time_dim, hidden_dim), nn.ReLU()]
        for _ in range(num_layers - 2):
            layers.extend([nn.Linear(hidden_dim, hidden_dim), nn.ReLU()])
        out = nn.Linear(hidden_dim, 3)
        nn.init.uniform_(out.weight, -1e-4, 1e-4)
        nn.init.zeros_(out.bias)
        layers.append(out)
        self.net = nn.Sequential(*layers)

    def forward(self, x_feat, t_feat):
        return self.net(torch.cat([x_feat, t_feat], dim=-1))
