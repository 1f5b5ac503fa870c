"""Import shim: the package directory is ``t-deed_amd/`` (the name the build
contract fixes); a hyphen is not a legal Python identifier, so ``import
tdeed_amd`` lands here and this file swaps itself for the real package, whose
``__path__`` points into ``t-deed_amd/`` so that ``tdeed_amd.model`` etc. resolve.
"""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "t-deed_amd")
_spec = importlib.util.spec_from_file_location(
    "tdeed_amd", os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["tdeed_amd"] = _mod
_spec.loader.exec_module(_mod)
