"""Import shim: the package directory is ``rtfs-net_amd/`` (not a valid Python identifier), so
``import rtfs_net_amd`` loads it from there under this importable name."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rtfs-net_amd")
_spec = importlib.util.spec_from_file_location(
    "rtfs_net_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["rtfs_net_amd"] = _mod
_spec.loader.exec_module(_mod)
