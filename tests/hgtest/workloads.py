"""Re-exports homography.js_amd/workloads.py (loaded by path: the package directory has a dot in its name)."""
import importlib.util
import os
import sys

_p = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "homography.js_amd", "workloads.py")
_spec = importlib.util.spec_from_file_location("hg_workloads", _p)
_m = importlib.util.module_from_spec(_spec)
sys.modules["hg_workloads"] = _m
_spec.loader.exec_module(_m)
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("_")})
