"""Loader for the package directory `deal-and-ceed-on-gpu_amd/` (its name is not a valid Python
identifier, so it is imported under the module name `deal_and_ceed_on_gpu_amd`)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "deal-and-ceed-on-gpu_amd")
MODULE_NAME = "deal_and_ceed_on_gpu_amd"


def load():
    if MODULE_NAME in sys.modules:
        return sys.modules[MODULE_NAME]
    spec = importlib.util.spec_from_file_location(
        MODULE_NAME, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[MODULE_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
