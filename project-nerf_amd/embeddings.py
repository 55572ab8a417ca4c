"""Encodings (reference src/embeddings.py) backed by the HIP kernels."""
import torch
import torch.nn as nn

from . import ops
from .abstract import BaseRepresentation


class FourierRepresentation(BaseRepresentation):
    """[x | sin(2^k pi x) | cos(2^k pi x)]_k, k < L (reference src/embeddings.py:6-36).
    The buffer name ``freq_bands`` is part of the checkpoint format."""

    def __init__(self, input_dim=2, L=10, use_encoding=True):
        super().__init__()
        self.input_dim, self.L, self.use_encoding = input_dim, L, use_encoding
        if use_encoding and L > 0:
            self.register_buffer("freq_bands", 2.0 ** torch.linspace(0.0, L - 1, steps=L))
            self._out_dim = input_dim + 2 * input_dim * L
        else:
            self.register_buffer("freq_bands", torch.empty(0))
            self._out_dim = input_dim

    def forward(self, x):
        if not self.use_encoding or self.L == 0:
            return x
        if x.requires_grad and torch.is_grad_enabled():
            # dynamic fields encode x + delta_x (reference src/core.py:268-271): the code is differentiable
            return _FourierEncode.apply(x.contiguous(), self.L)
        return ops.fourier_encode(x, self.L)

    @property
    def out_dim(self):
        return self._out_dim


class _FourierEncode(torch.autograd.Function):
    """HIP Fourier code with its input gradient: d sin(a x) = a cos(a x) dx, d cos(a x) = -a sin(a x) dx with
    a = 2^k pi, evaluated from the code itself (which already holds every sin and cos)."""

    @staticmethod
    def forward(ctx, x, n_freq):
        out = ops.fourier_encode(x, n_freq)
        ctx.save_for_backward(out)
        ctx.n_freq = n_freq
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        dim, L = out.shape[1] // (1 + 2 * ctx.n_freq), ctx.n_freq
        g, out = g.contiguous(), out
        band = (2.0 ** torch.arange(L, device=g.device, dtype=g.dtype) * torch.pi).view(1, L, 1)
        sc, gsc = out[:, dim:].view(-1, L, 2, dim), g[:, dim:].view(-1, L, 2, dim)       # [n, band, (sin, cos), coord]
        dx = g[:, :dim] + ((gsc[:, :, 0] * sc[:, :, 1] - gsc[:, :, 1] * sc[:, :, 0]) * band).sum(dim=1)
        return dx, None


class ParamHolder(nn.Module):
    """Stands in for a tcnn module (tcnn.Encoding / tcnn.Network): one flat fp32 ``params`` vector,
    so that state_dict keys read ``...encoding.params`` / ``...sigma_net.params`` as in the reference."""

    def __init__(self, init, n_output_dims=None):
        super().__init__()
        self.params = nn.Parameter(init)
        self.n_output_dims = n_output_dims


class HashRepresentation(BaseRepresentation):
    """reference src/embeddings.py:39-93."""

    def __init__(self, n_levels=16, n_features_per_level=2, log2_hashmap_size=19, base_resolution=16,
                 per_level_scale=1.5, bound=1.0):
        super().__init__()
        if n_features_per_level != 2 or not 1 <= n_levels <= 16:
            raise NotImplementedError("libnerf_hip's hash-grid kernels are compiled for 2 features per level and up to 16 levels")
        self.bound = bound
        self.levels = ops.HashLevelTable(n_levels, log2_hashmap_size, base_resolution, per_level_scale)
        init = (torch.rand(self.levels.entries * 2) * 2 - 1) * 1e-4
        self.encoding = ParamHolder(init, n_output_dims=n_levels * n_features_per_level)
        self._out_dim = self.encoding.n_output_dims

    def table(self):
        return self.encoding.params.view(-1, 2)

    def forward(self, x):
        """reference src/embeddings.py:75-89 (normalise to [0,1], clamp, encode); differentiable w.r.t. the table."""
        return ops.hash_encode(self.table(), x, self.levels, self.bound)

    @property
    def out_dim(self):
        return self._out_dim
