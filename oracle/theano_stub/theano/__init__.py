"""NumPy-evaluated stand-in for Theano -- oracle tooling only, see _lazy.py."""
from ._lazy import shared, function          # noqa: F401
from . import tensor, compile                # noqa: F401

__version__ = '0.0-numpy-stub'
