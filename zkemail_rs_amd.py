"""Import shim: the package directory is named ``zkemail.rs_amd`` (after the reference repo),
which Python's import statement cannot spell.  ``import zkemail_rs_amd`` loads it."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "zkemail.rs_amd")
_spec = importlib.util.spec_from_file_location(
    "zkemail_rs_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["zkemail_rs_amd"] = _mod
_spec.loader.exec_module(_mod)
