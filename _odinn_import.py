"""Import helper: the product package lives in the directory ``odinn.jl_amd/`` (the name
the build contract fixes), which is not a valid Python identifier.  ``load()`` registers it
as the module ``odinn_jl_amd``."""
import importlib.util
import os
import sys

_NAME = "odinn_jl_amd"


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "odinn.jl_amd")
    spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(root, "__init__.py"), submodule_search_locations=[root]
    )
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
