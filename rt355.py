"""Alias for the renderer package, whose directory name (`ray-tracing_amd`) is not a
valid identifier:  `import rt355` == importlib.import_module("ray-tracing_amd")."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("ray-tracing_amd")
sys.modules[__name__] = _pkg
