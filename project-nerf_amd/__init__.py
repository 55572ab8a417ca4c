"""MI355X-native NeRF volumetric-rendering hot path.

Layout:
  csrc/            hand-written HIP kernels for gfx950 + the C ABI (include/nerf_hip.h)
  libnerf_hip.so   built in-tree by build.py (hipcc --offload-arch=gfx950)
  _lib.py          ctypes binding of the C ABI
  ops.py           tensor-level wrappers and autograd Functions (device memory + streams only)
  abstract.py, embeddings.py, decoders.py, core.py, renderer.py, dataset.py, utils.py
                   host-side mirror of the reference's module surface (src/*.py)

Nothing here falls back to PyTorch math: if the HIP library is missing or a tensor is not on a
HIP device the ops raise.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
