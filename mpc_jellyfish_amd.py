"""Importable alias for the package directory `mpc-jellyfish_amd/` (the hyphen is not a valid
Python identifier):  `import mpc_jellyfish_amd as mj`."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("mpc-jellyfish_amd")
sys.modules[__name__] = _pkg
