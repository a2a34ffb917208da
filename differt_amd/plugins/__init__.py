"""Plugins mirrored from ``differt.plugins`` (only what sits downstream of the traced paths)."""

from . import deepmimo

__all__ = ["deepmimo"]
