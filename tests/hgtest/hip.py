"""Loads homography.js_amd/hgwarp.py (ctypes binding of the C ABI) by path: the package directory has a dot in its name."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG_DIR = os.path.join(ROOT, "homography.js_amd")


def load():
    if "hgwarp" in sys.modules:
        return sys.modules["hgwarp"]
    spec = importlib.util.spec_from_file_location("hgwarp", os.path.join(PKG_DIR, "hgwarp.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["hgwarp"] = mod
    spec.loader.exec_module(mod)
    return mod
