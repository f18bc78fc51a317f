"""Import shim: makes the package directory `sgfhe.jl_amd/` (whose name is not a valid Python
identifier) importable as `sgfhe_jl_amd`."""

import importlib.util
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
_pkg = os.path.join(_root, "sgfhe.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "sgfhe_jl_amd", os.path.join(_pkg, "__init__.py"), submodule_search_locations=[_pkg])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sgfhe_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
