"""Import shim: the package lives in `learning-adaptive-neighborhoods-for-gnns_amd/` (not a valid Python
identifier); `import dgg_amd` loads it from there under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "learning-adaptive-neighborhoods-for-gnns_amd")
_spec = importlib.util.spec_from_file_location("dgg_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dgg_amd"] = _mod
_spec.loader.exec_module(_mod)
