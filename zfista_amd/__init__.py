"""zfista_amd - MI355X-native drop-in for zfista.minimize_proximal_gradient.

Public surface mirrors zfista/__init__.py:1-3 (one exported function) plus the
native operator objects of ``zfista_amd.problems``.
"""
from .proximal_gradient import minimize_proximal_gradient

__all__ = ["minimize_proximal_gradient"]
