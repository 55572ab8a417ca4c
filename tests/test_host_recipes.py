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
