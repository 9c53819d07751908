"""Import shim: the package directory is named `periodicschurdecompositions.jl_amd` (contains a dot, so it is
not importable with a plain `import`); `import psd_amd` loads it by path."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "periodicschurdecompositions.jl_amd")
_spec = importlib.util.spec_from_file_location("psd_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["psd_amd"] = _mod
_spec.loader.exec_module(_mod)
