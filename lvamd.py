"""Loads the product package directory `limo-velo_amd/` under the importable name `limo_velo_amd`.

The directory name is fixed by the project layout and contains a hyphen, which Python cannot import
directly; `import lvamd; lv = lvamd.load()` (or `from lvamd import limo_velo_amd`) resolves it.
"""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_PKG_DIR = os.path.join(_ROOT, "limo-velo_amd")


def load():
    name = "limo_velo_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(_PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[_PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


limo_velo_amd = load()
