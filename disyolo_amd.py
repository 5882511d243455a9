"""Import shim: the package directory is named ``dis-yolo_amd`` (hyphen, as the
project contract requires), which Python cannot import by name.  ``import
disyolo_amd`` loads that directory as a regular package under this name."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dis-yolo_amd")
_spec = importlib.util.spec_from_file_location(
    "disyolo_amd", os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["disyolo_amd"] = _mod
_spec.loader.exec_module(_mod)
