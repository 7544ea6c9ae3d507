"""plspy_amd -- MI355X-native engine for plspy's permutation / bootstrap /
split-half resampling path, behind plspy's own ``PLS()`` / ``methods[]``
surface.  See DESIGN.md for scope and INTEGRATION.md for the C ABI."""
from . import exceptions  # noqa: F401
from .pls import PLS, methods  # noqa: F401

__all__ = ["PLS", "methods", "exceptions"]
