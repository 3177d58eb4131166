"""Registers the package directory `protgen-gan_amd/` under the importable name
`protgen_gan_amd` (a hyphen cannot appear in a Python identifier)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "protgen-gan_amd")
NAME = "protgen_gan_amd"


def load():
    if NAME in sys.modules:
        return sys.modules[NAME]
    spec = importlib.util.spec_from_file_location(NAME, os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[NAME] = mod
    try:
        spec.loader.exec_module(mod)
    except Exception:
        del sys.modules[NAME]
        raise
    return mod
