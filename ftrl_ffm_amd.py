"""Import shim: the package directory is `ftrl-ffm_amd/` (not a valid Python identifier), so
`import ftrl_ffm_amd` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ftrl-ffm_amd")
_spec = importlib.util.spec_from_file_location(
    "ftrl_ffm_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ftrl_ffm_amd"] = _mod
_spec.loader.exec_module(_mod)
