"""Import alias: `import sgg_amd` loads the package in ./scene-graph-gan_amd/ (a directory name that is not a
valid Python identifier).  Sub-modules resolve as sgg_amd.<name>."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scene-graph-gan_amd")
_spec = importlib.util.spec_from_file_location("sgg_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sgg_amd"] = _mod
_spec.loader.exec_module(_mod)
