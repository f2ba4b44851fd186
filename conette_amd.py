"""Import alias: the package directory is ``conette-audio-captioning_amd/`` (not a valid Python
identifier), this shim loads it under the importable name ``conette_amd``."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "conette-audio-captioning_amd")
_spec = _ilu.spec_from_file_location("conette_amd", _os.path.join(_dir, "__init__.py"),
                                     submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["conette_amd"] = _mod
_spec.loader.exec_module(_mod)
