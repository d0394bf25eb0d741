"""Makes the hyphen-named product package importable from the drop-in modules."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)


def package(sub=None):
    name = "deepq-decoding_amd" + ("." + sub if sub else "")
    return importlib.import_module(name)
