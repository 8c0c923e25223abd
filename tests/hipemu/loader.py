"""Loads the host-interpreted TEST build of the kernels (never used by the product)."""
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import build_emu  # noqa: E402

from rainbow_amd import _lib as L  # noqa: E402

_emu = None


def load(strict=False):
    global _emu
    if _emu is None:
        path = build_emu.build()
        _emu = L.declare(C.CDLL(path), strict=strict)
    return _emu
