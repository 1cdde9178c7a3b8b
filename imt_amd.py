"""Import alias: `import imt_amd` loads the package in ./indexed-merkle-tree-halo2_amd/
(a directory name with dashes cannot be imported by name)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "indexed-merkle-tree-halo2_amd")
_spec = importlib.util.spec_from_file_location("imt_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["imt_amd"] = _mod
_spec.loader.exec_module(_mod)
