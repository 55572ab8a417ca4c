"""Drop-in module path of the reference (src/utils.py): re-exports the MI355X implementation."""
import project_nerf_amd  # noqa: F401  (import shim for the hyphenated package directory)
from project_nerf_amd.utils import *  # noqa: F401,F403
