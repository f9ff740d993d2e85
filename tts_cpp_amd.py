"""Import shim: the package directory is literally ``tts.cpp_amd/`` (a dot is not importable),
so ``import tts_cpp_amd`` loads that directory as the package ``tts_cpp_amd``."""
import importlib.util
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_root, "tts.cpp_amd")
_spec = importlib.util.spec_from_file_location(
    "tts_cpp_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["tts_cpp_amd"] = _mod
_spec.loader.exec_module(_mod)
