"""Import helper: the package directory `montecarlo-surfacer_amd/` has a hyphen
(it mirrors the reference's name), so it is loaded by path under the module
name `montecarlo_surfacer_amd`."""
import importlib.util
import os
import sys

_NAME = "montecarlo_surfacer_amd"


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    root = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(root, "montecarlo-surfacer_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(_NAME, path,
                                                  submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
