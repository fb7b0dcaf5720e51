"""Import shim: exposes the package directory `clip-guided-diffusion_amd/` as the module `cgd_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "clip-guided-diffusion_amd")
_spec = importlib.util.spec_from_file_location("cgd_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["cgd_amd"] = _mod
_spec.loader.exec_module(_mod)
