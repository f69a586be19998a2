"""Alias: `import ltompc` == the package in `lap-time-optimization_amd/` (whose name is not a Python identifier)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
_pkg = importlib.import_module("lap-time-optimization_amd")
sys.modules[__name__] = _pkg
