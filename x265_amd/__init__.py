"""x265_amd: MI355X (gfx950) implementation of the x265-Milestone pixel-primitive hot path.

The product is libx265_amd.so (HIP kernels behind a C ABI, see include/x265_amd.h). This package is the
thin Python binding used by the tests and the benchmark; it never computes anything itself and
raises if the native library is missing.
"""
from .capi import XmiError, lib, load  # noqa: F401

__version__ = "0.1"
