"""NeuralField (reference src/core.py:9-363): mode dispatch around the HIP operators."""
import torch.nn as nn

import torch

from .decoders import (DeformationNetwork, HashDeformationDecoder, InstantNeRFDecoder, NeRFDecoder, StandardMLP,
                       TimeModulationNetwork)
from .embeddings import FourierRepresentation, HashRepresentation


class NeuralField(nn.Module):
    """coords -> encoding -> decoder.  Attribute names (``mode``, ``representation``,
    ``dir_representation``, ``decoder``) are read by the renderer and the training scripts."""

    def __init__(self, config):
        super().__init__()
        self.mode = config["mode"]
        # coordinate / time noise augmentation of the dynamic modes (reference src/core.py:15-18)
        self.use_coord_noise = config.get("use_coord_noise", False)
        self.coord_noise_std = config.get("coord_noise_std", 0.005)
        self.time_noise_std = config.get("time_noise_std", 0.02)
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
        elif self.mode == "part3":
            # MLP deformation field + canonical field (reference src/core.py:79-146): Fourier codes and the hash grid in
            # HIP (the codes differentiable in their input: the loss reaches the deformation through x + delta_x), the
            # deformation MLP and a canonical decoder of non-compiled width as library GEMMs
            self.dir_representation = FourierRepresentation(input_dim=3, L=config.get("L_embed_dir", 4), use_encoding=True)
            self.time_encoder = FourierRepresentation(input_dim=1, L=config.get("L_embed_time", 10), use_encoding=True)
            self.pos_encoder_for_deform = FourierRepresentation(input_dim=3, L=config.get("L_embed", 10), use_encoding=True)
            self.deform_net = DeformationNetwork(pos_dim=self.pos_encoder_for_deform.out_dim, time_dim=self.time_encoder.out_dim,
                                                 hidden_dim=config.get("deform_hidden_dim", 128),
                                                 num_layers=config.get("deform_num_layers", 4))
            mlp_kw = dict(hidden_dim=config.get("hidden_dim", 256), num_layers=config.get("num_layers", 8),
                          skip_layer=config.get("skip_layer", 4), view_dim=config.get("view_dim", 128))
            if config.get("canonical_type", "nerf") == "instant":
                self.canonical_repr = HashRepresentation(
                    n_levels=config.get("n_levels", 16), n_features_per_level=config.get("n_features_per_level", 2),
                    log2_hashmap_size=config.get("log2_hashmap_size", 19), base_resolution=config.get("base_resolution", 16),
                    per_level_scale=config.get("per_level_scale", 1.5), bound=config.get("scene_bound", 1.0))
                self.decoder = InstantNeRFDecoder(pos_dim=self.canonical_repr.out_dim + self.time_encoder.out_dim,
                                                  dir_dim=self.dir_representation.out_dim, hidden_dim=config.get("hidden_dim", 64))
            else:
                self.canonical_repr = FourierRepresentation(input_dim=3, L=config.get("L_embed_canon", 10), use_encoding=True)
                self.decoder = NeRFDecoder(pos_dim=self.canonical_repr.out_dim + self.time_encoder.out_dim,
                                           dir_dim=self.dir_representation.out_dim, **mlp_kw)
            self.direct_time_conditioning = config.get("direct_time_conditioning", False)
            if self.direct_time_conditioning:
                self.pos_encoder_direct = FourierRepresentation(input_dim=3, L=config.get("L_embed", 10), use_encoding=True)
                self.decoder_direct = NeRFDecoder(pos_dim=self.pos_encoder_direct.out_dim + self.time_encoder.out_dim,
                                                  dir_dim=self.dir_representation.out_dim, **mlp_kw)
        elif self.mode == "part4":
            # dual-hash dynamic field (reference src/core.py:148-225): three deformation hash grids anchored in
            # time, a shared displacement decoder gated by a time-modulation MLP, a canonical hash grid and a
            # time-conditioned Instant decoder.  Attribute names are read by run_part4's regularisers.
            self.dir_representation = FourierRepresentation(input_dim=3, L=config.get("L_embed_dir", 4), use_encoding=True)
            self.time_encoder = FourierRepresentation(input_dim=1, L=config.get("L_embed_time", 10), use_encoding=True)
            mod_dim = config.get("time_modulation_dim", 64)
            self.time_modulation = TimeModulationNetwork(time_dim=self.time_encoder.out_dim, output_dim=mod_dim, hidden_dim=mod_dim,
                                                         num_layers=config.get("time_modulation_layers", 2))
            grid_cfg = dict(n_levels=config.get("deform_n_levels", 14), n_features_per_level=config.get("deform_n_features_per_level", 2),
                            log2_hashmap_size=config.get("deform_log2_hashmap_size", 19),
                            base_resolution=config.get("deform_base_resolution", 16),
                            per_level_scale=config.get("deform_per_level_scale", 1.5), bound=config.get("scene_bound", 1.5))
            self.deform_grid_start = HashRepresentation(**grid_cfg)
            self.deform_grid_mid = HashRepresentation(**grid_cfg)
            self.deform_grid_end = HashRepresentation(**grid_cfg)
            with torch.no_grad():        # break the symmetry of the three grids (src/core.py:191-196)
                for grid in (self.deform_grid_mid, self.deform_grid_end):
                    grid.encoding.params.add_(torch.randn_like(grid.encoding.params) * 1e-4)
            self.deformation_grid = self.deform_grid_start       # alias kept for the training script (src/core.py:199)
            self.deform_decoder = HashDeformationDecoder(hash_dim=self.deform_grid_start.out_dim, time_mod_dim=mod_dim,
                                                         hidden_dim=config.get("deform_hidden_dim", 64))
            self.canonical_repr = HashRepresentation(
                n_levels=config.get("n_levels", 16), n_features_per_level=config.get("n_features_per_level", 2),
                log2_hashmap_size=config.get("log2_hashmap_size", 19), base_resolution=config.get("base_resolution", 16),
                per_level_scale=config.get("per_level_scale", 1.5), bound=config.get("scene_bound", 1.5))
            self.decoder = InstantNeRFDecoder(pos_dim=self.canonical_repr.out_dim + self.time_encoder.out_dim,
                                              dir_dim=self.dir_representation.out_dim, hidden_dim=config.get("hidden_dim", 64))
            # the reference's example shapes run on the fused HIP chains (csrc/p4mlp.hip); `fused_part4: false` in the YAML
            # or any other shape: the same arithmetic composed from the stand-alone operators (library GEMMs for the MLPs)
            from . import part4
            self._p4_fused = bool(config.get("fused_part4", True)) and part4.supported(config) is None
        else:
            raise NotImplementedError(f"mode {self.mode!r}: built are part1_fourier, part2_nerf, part2_instant, part3 and part4")

    def forward(self, x, d=None, t=None):
        if self.mode == "part1_fourier":
            return self.decoder(self.representation(x))
        if self.mode in ("part2_nerf", "part2_instant"):
            if d is None:
                raise ValueError(f"{self.mode} requires view directions.")
            if self.mode == "part2_nerf":
                return self.decoder.field(x, d)
            return self._instant_forward(x, d)
        if self.mode == "part3":
            return self._part3_forward(x, d, t)
        if self.mode == "part4":
            return self._part4_forward(x, d, t)
        raise NotImplementedError(self.mode)

    def _part3_forward(self, x, d, t):
        """reference src/core.py:233-281: (rgb, sigma, delta_x); delta_x = 0 under direct time conditioning."""
        if t is None:
            raise ValueError("Part 3 requires time input 't'.")
        if d is None:
            raise ValueError("part3 requires view directions.")
        x, d = x.contiguous(), d.contiguous()
        if self.direct_time_conditioning:
            h = torch.cat([self.pos_encoder_direct(x), self.time_encoder(t.contiguous())], dim=-1)
            rgb, sigma = self.decoder_direct(h, self.dir_representation(d))
            return rgb, sigma, torch.zeros_like(x)
        x_deform, t_deform = x, t
        if self.training and self.use_coord_noise:
            if self.coord_noise_std > 0:
                x_deform = x + torch.randn_like(x) * self.coord_noise_std
            if self.time_noise_std > 0:
                t_deform = torch.clamp(t + torch.randn_like(t) * self.time_noise_std, 0.0, 1.0)
        feat_t = self.time_encoder(t_deform.contiguous())
        delta_x = self.deform_net(self.pos_encoder_for_deform(x_deform.contiguous()), feat_t)
        feat_can = self.canonical_repr((x + delta_x).contiguous())
        rgb, sigma = self.decoder(torch.cat([feat_can, feat_t], dim=-1), self.dir_representation(d))
        return rgb, sigma, delta_x

    def _part4_forward(self, x, d, t):
        """reference src/core.py:282-352: (rgb, sigma, delta_x).  The four hash encodings (table gradients AND
        the gradient with respect to the canonical position, through which the loss reaches the deformation),
        the time / direction Fourier codes run in HIP; the nonstandard tiny MLPs as library GEMMs."""
        if t is None:
            raise ValueError("Part 4 requires time input 't'.")
        if d is None:
            raise ValueError("part4 requires view directions.")
        x_deform, t_deform = x, t
        if self.training and self.use_coord_noise:
            if self.coord_noise_std > 0:
                x_deform = x + torch.randn_like(x) * self.coord_noise_std
            if self.time_noise_std > 0:
                t_deform = torch.clamp(t + torch.randn_like(t) * self.time_noise_std, 0.0, 1.0)
        if getattr(self, "_p4_fused", False) and x.is_cuda:
            from . import part4
            return part4.field(self, x, d, t_deform, None if x_deform is x else x_deform)
        feat_t = self.time_encoder(t_deform.contiguous())
        time_mod = self.time_modulation(feat_t)
        x_deform = x_deform.contiguous()
        feats = [g(x_deform) for g in (self.deform_grid_start, self.deform_grid_mid, self.deform_grid_end)]
        # triangle weights around the anchors 0, 0.5, 1 (bandwidth 0.5), normalised (src/core.py:313-332)
        w = [torch.clamp(1.0 - torch.abs(t_deform - a) / 0.5, 0.0, 1.0) for a in (0.0, 0.5, 1.0)]
        w_sum = w[0] + w[1] + w[2] + 1e-8
        deform_feat = (w[0] / w_sum) * feats[0] + (w[1] / w_sum) * feats[1] + (w[2] / w_sum) * feats[2]
        delta_x = self.deform_decoder(deform_feat, time_mod)
        x_canonical = (x + delta_x).contiguous()
        feat_can = self.canonical_repr(x_canonical)
        feat_d = self.dir_representation(d.contiguous())
        rgb, sigma = self.decoder(torch.cat([feat_can, feat_t], dim=-1), feat_d)
        return rgb, sigma, delta_x

    def field_from_rays(self, rays_o, rays_d, z):
        """Fused ray-mode entry used by render_rays: sample points are formed in registers."""
        if self.mode != "part2_nerf":
            raise NotImplementedError
        return self.decoder.field(rays_o, rays_d, z)
