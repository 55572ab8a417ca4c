"""NeuralField (reference src/core.py:9-363): mode dispatch around the HIP operators."""
import torch.nn as nn

from .decoders import NeRFDecoder, StandardMLP
from .embeddings import FourierRepresentation


class NeuralField(nn.Module):
    """coords -> encoding -> decoder.  Attribute names (``mode``, ``representation``,
    ``dir_representation``, ``decoder``) are read by the renderer and the training scripts."""

    def __init__(self, config):
        super().__init__()
        self.mode = config["mode"]
        use_pe = config.get("use_positional_encoding", True)
        L = config.get("L_embed", 0) if use_pe else 0
        if self.mode == "part1_fourier":
            # 2-D image fit (reference core.py:25-34): HIP Fourier features + a library-GEMM MLP
            self.representation = FourierRepresentation(input_dim=2, L=L, use_encoding=use_pe)
            self.decoder = StandardMLP(input_dim=self.representation.out_dim, hidden_dim=config["hidden_dim"],
                                       output_dim=config["output_dim"], num_layers=config.get("num_layers", 3))
        elif self.mode == "part2_nerf":
            self.representation = FourierRepresentation(input_dim=3, L=L, use_encoding=use_pe)
            use_dir = config.get("use_viewdirs", True)
            L_dir = config.get("L_embed_dir", 4) if use_dir else 0
            self.dir_representation = FourierRepresentation(input_dim=3, L=L_dir, use_encoding=use_dir)
            self.decoder = NeRFDecoder(
                pos_dim=self.representation.out_dim, dir_dim=self.dir_representation.out_dim,
                hidden_dim=config.get("hidden_dim", 256), num_layers=config.get("num_layers", 8),
                skip_layer=config.get("skip_layer", 4), view_dim=config.get("view_dim", 128))
        elif self.mode == "part2_instant":
            from .instant import build_instant_field
            build_instant_field(self, config)
        else:
            raise NotImplementedError(
                f"mode {self.mode!r}: only part1_fourier and the static hot path (part2_nerf, part2_instant) are "
                "built; part3/part4 are listed as next rows in DESIGN.md")

    def forward(self, x, d=None, t=None):
        if self.mode == "part1_fourier":
            return self.decoder(self.representation(x))
        if self.mode in ("part2_nerf", "part2_instant"):
            if d is None:
                raise ValueError(f"{self.mode} requires view directions.")
            if self.mode == "part2_nerf":
                return self.decoder.field(x, d)
            return self._instant_forward(x, d)
        raise NotImplementedError(self.mode)

    def field_from_rays(self, rays_o, rays_d, z):
        """Fused ray-mode entry used by render_rays: sample points are formed in registers."""
        if self.mode != "part2_nerf":
            raise NotImplementedError
        return self.decoder.field(rays_o, rays_d, z)
