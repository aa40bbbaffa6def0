"""Import alias: `import lrm_amd` loads the package directory
`legged-robot-movability-cuda_amd/` (its name is not a valid Python identifier)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "legged-robot-movability-cuda_amd")
_spec = importlib.util.spec_from_file_location(
    "lrm_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["lrm_amd"] = _mod
_spec.loader.exec_module(_mod)
