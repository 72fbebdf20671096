"""kvazaar_amd -- MI355X-native block kernels for Kvazaar's strategy API.

The product is kvazaar_amd/libkvzhip.so (hand-written HIP kernels for gfx950 +
the C ABI of include/kvz_hip.h).  This package is the thin Python host side used
by tests and bench.py; it has no CPU fallback and never imports oracle/."""
from . import _lib  # noqa: F401

__all__ = ["_lib", "api"]
