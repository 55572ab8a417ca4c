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
        if x.requires_grad:
            raise NotImplementedError("gradients w.r.t. encoded coordinates are not part of the static path")
        return ops.fourier_encode(x, self.L)

    @property
    def out_dim(self):
        return self._out_dim


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
