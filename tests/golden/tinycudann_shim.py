"""Stand-in for the third-party ``tinycudann`` module -- BUILD CONTAINER ONLY, TEST INFRASTRUCTURE.

The reference's Instant paths import tinycudann lazily (src/embeddings.py:57, src/decoders.py:107); the
library is not vendored, not version-pinned and not installable offline, so its INTERNALS stay unpinned.
This shim lets the reference's own GLUE around it run as the oracle:
``HashRepresentation.forward``'s normalise + clamp, ``InstantNeRFDecoder.forward``'s softplus(h0 - 5) and
cat([h16, d_enc]) wiring, ``NeuralField('part2_instant')``, masked ``render_rays`` and
``DensityGrid.update`` around an instant field.  ``tests/golden/make_golden.py::g13_instant_glue`` inserts
it into ``sys.modules`` as ``tinycudann`` before building the reference model; nothing else imports it
(it never runs on the GPU box and is not part of the product).

Semantics = this build's definition of the two operators (oracle/nerf_oracle.py::hash_encode, tiny_mlp;
layouts in include/nerf_hip.h): flat fp32 ``params``; hash grid = per-level [entries_l, F] tables
concatenated; networks = bias-free, matrices [out, in] row-major concatenated with in/out widths rounded
up to multiples of 16, pad inputs zero."""
import importlib.util
import math
import os
import sys

import torch
import torch.nn as nn

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_spec = importlib.util.spec_from_file_location("nerf_oracle_for_shim", os.path.join(_ROOT, "oracle", "nerf_oracle.py"))
O = importlib.util.module_from_spec(_spec)
sys.modules[_spec.name] = O          # dataclasses look their module up while the class body runs
_spec.loader.exec_module(O)


def _pad16(n):
    return (n + 15) // 16 * 16


# tinycudann returns fp16 from both operators ("[N, L*F] fp16", "[N, out] fp16").  False (default): the
# stand-in stays fp32 end to end (the goldens' main vectors).  True: outputs are rounded to fp16 and widened again -- what
# make_golden.py::g14b uses for its SECOND set of vectors, which shows how far the reference's own Part 4 field moves when
# only the operators' output precision is tinycudann's (displacements rounded to fp16 land in other fine hash cells).
FP16_OUTPUTS = False


def _out(y):
    return y.half().float() if FP16_OUTPUTS else y


class Encoding(nn.Module):
    def __init__(self, n_input_dims, encoding_config, dtype=None):
        super().__init__()
        assert n_input_dims == 3 and encoding_config["otype"] == "HashGrid"
        c = encoding_config
        self.levels = O.hash_grid_levels(c["n_levels"], c["log2_hashmap_size"], c["base_resolution"], c["per_level_scale"])
        self.n_features = c["n_features_per_level"]
        self.n_output_dims = c["n_levels"] * self.n_features
        self.params = nn.Parameter((torch.rand(O.hash_grid_entries(self.levels) * self.n_features) * 2 - 1) * 1e-4)

    def forward(self, x01):
        return _out(O.hash_encode(self.levels, self.params.view(-1, self.n_features), x01))


class Network(nn.Module):
    def __init__(self, n_input_dims, n_output_dims, network_config):
        super().__init__()
        c = network_config
        assert c["otype"] == "FullyFusedMLP" and c["activation"] == "ReLU"
        self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims
        self.out_act = None if c["output_activation"] in (None, "None") else c["output_activation"].lower()
        widths = [_pad16(n_input_dims)] + [c["n_neurons"]] * c["n_hidden_layers"] + [_pad16(n_output_dims)]
        self.shapes = [(widths[i + 1], widths[i]) for i in range(len(widths) - 1)]
        parts = []
        for i, (o, k) in enumerate(self.shapes):
            fan_in = n_input_dims if i == 0 else k
            fan_out = n_output_dims if i == len(self.shapes) - 1 else o
            w = (torch.rand(o, k) * 2 - 1) * math.sqrt(6.0 / (fan_in + fan_out))
            if i == 0:
                w[:, n_input_dims:] = 0
            if i == len(self.shapes) - 1:
                w[n_output_dims:] = 0
            parts.append(w.reshape(-1))
        self.params = nn.Parameter(torch.cat(parts))

    def weights(self):
        out, off = [], 0
        for o, k in self.shapes:
            out.append(self.params[off:off + o * k].view(o, k))
            off += o * k
        return out

    def forward(self, x):
        pad = self.shapes[0][1] - x.shape[-1]
        if pad:
            x = torch.cat([x, x.new_zeros(x.shape[0], pad)], dim=-1)
        y = O.tiny_mlp(self.weights(), x.float(), out_act=self.out_act)
        return _out(y[:, :self.n_output_dims])
