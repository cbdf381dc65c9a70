"""Importable alias for the package directory (whose name, fixed by the repo contract, is not an identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module(
    "reinforcement-learning-for-playing-tetris-with-prescribed-initial-configuration-and-limited-moves_amd")
sys.modules[__name__] = _pkg
