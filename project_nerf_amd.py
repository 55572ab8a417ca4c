"""Import shim: the package directory is named ``project-nerf_amd`` (not a valid Python
identifier), so ``import project_nerf_amd`` loads it from there under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "project-nerf_amd")
_spec = importlib.util.spec_from_file_location(
    "project_nerf_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["project_nerf_amd"] = _mod
_spec.loader.exec_module(_mod)
