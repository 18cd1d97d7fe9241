"""Import shim: the product package lives in `sequence-alignment-tools_amd/` (not a valid module
name), this makes it importable as `sat_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sequence-alignment-tools_amd")
_spec = importlib.util.spec_from_file_location("sat_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sat_amd"] = _mod
_spec.loader.exec_module(_mod)
