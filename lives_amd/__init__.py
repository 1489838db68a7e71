"""lives_amd -- MI355X (gfx950) per-frame effects engine for LiVES.

The product is ``liblivesgpu.so`` (C ABI: include/lives_gpu.h, built from lives_amd/csrc/ by
``lives_amd/csrc/build.sh``) plus the weed plugin ``livesgpu_fx.so``.  The Python in this package is a
thin ctypes binding used by tests and bench.py; it never computes pixels itself.
"""
from . import lib  # noqa: F401

__all__ = ["lib", "ops"]
