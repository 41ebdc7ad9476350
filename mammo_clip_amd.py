"""Import shim: makes the package directory ``mammo-clip_amd/`` importable as ``mammo_clip_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mammo-clip_amd")
_spec = importlib.util.spec_from_file_location(
    "mammo_clip_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mammo_clip_amd"] = _mod
_spec.loader.exec_module(_mod)
