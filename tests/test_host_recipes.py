"""Host-side restatements of bit-exact recipes the kernels rely on (no GPU needed)."""
import numpy as np
import pytest
import torch


def linspace01_recipe(n):
    """csrc/sample.hip linspace01(): lower half step*i, upper half fma(-step, n-1-i, 1);
    emulated in float64 (products of fp32 values are exact there) then rounded once."""
    step = np.float32(1.0) / np.float32(n - 1)
    i = np.arange(n)
    lo = (np.float64(step) * i).astype(np.float32)
    hi = (1.0 - np.float64(step) * (n - 1 - i)).astype(np.float32)
    return np.where(i < n // 2, lo, hi)


@pytest.mark.parametrize("n", [2, 3, 5, 32, 64, 65, 96, 127, 128, 192, 256, 512])
def test_linspace_recipe_matches_torch_cpu(n):
    assert np.array_equal(linspace01_recipe(n), torch.linspace(0.0, 1.0, n).numpy())


def test_voxel_scale_is_fp32_demoted():
    """Python double 128/3 demoted to fp32 is what multiplies."""
    s = torch.tensor(1.0) * (128 / (2 * 1.5))
    assert s.item() == float(np.float32(128 / 3.0))


def test_part4_param_groups_follow_the_reference():
    """run.py:1684-1738: 2x lr for the three deformation grids and the canonical grid, 5x for displacement_scale, 1x for
    the deformation MLP and the rest; every parameter in exactly one group (round-2 advisor finding: one rate for all)."""
    import torch
    import project_nerf_amd  # noqa: F401
    from project_nerf_amd.core import NeuralField
    from project_nerf_amd.dynamic import part4_param_groups
    cfg = dict(mode="part4", deform_n_levels=4, deform_log2_hashmap_size=10, n_levels=4, log2_hashmap_size=10, scene_bound=1.5)
    model = NeuralField(cfg)
    groups = part4_param_groups(model, 0.01)
    lrs = {g["name"]: g["lr"] for g in groups}
    assert lrs == {"deform_grid_start": 0.02, "deform_grid_mid": 0.02, "deform_grid_end": 0.02, "canonical_repr": 0.02,
                   "displacement_scale": 0.05, "deform_decoder": 0.01, "others": 0.01}
    seen = [id(p) for g in groups for p in g["params"]]
    assert len(seen) == len(set(seen)) and set(seen) == {id(p) for p in model.parameters()}
    opt = torch.optim.AdamW(groups, lr=0.01, weight_decay=1e-5)          # the optimiser accepts them as they are
    assert [g["lr"] for g in opt.param_groups] == [g["lr"] for g in groups]
