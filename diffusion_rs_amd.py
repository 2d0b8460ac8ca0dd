"""Import shim: the package directory is `diffusion-rs_amd/` (a hyphen is not importable), so
`import diffusion_rs_amd` executes this file, which loads that directory as the package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "diffusion-rs_amd")
_spec = importlib.util.spec_from_file_location("diffusion_rs_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["diffusion_rs_amd"] = _mod
_spec.loader.exec_module(_mod)
