"""Encodings (reference src/embeddings.py) backed by the HIP kernels."""
import torch

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
