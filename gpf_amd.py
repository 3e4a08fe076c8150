"""Import shim: the package directory is `genparticlefilters.jl_amd/` (contains a dot, so it is not
importable by name).  `import gpf_amd` loads it under this module name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "genparticlefilters.jl_amd")
_spec = importlib.util.spec_from_file_location("gpf_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gpf_amd"] = _mod
_spec.loader.exec_module(_mod)
