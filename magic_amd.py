"""Import shim: the package directory is named ``vln-magic_amd`` (not a valid Python identifier),
so ``import magic_amd`` loads it under this importable name."""
import importlib.util
import os
import sys

_root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vln-magic_amd")
_spec = importlib.util.spec_from_file_location(
    "magic_amd", os.path.join(_root, "__init__.py"), submodule_search_locations=[_root])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["magic_amd"] = _mod
_spec.loader.exec_module(_mod)
