"""tensor-ops_amd: MI355X (gfx950) backend for the `Tensor`/`BLAS` typeclass
boundary of mstksg/tensor-ops.

  csrc/      hand-written HIP kernels + the C ABI (include/tensorops_hip.h)
  host/      C++ mirror of the reference's host side (TOp / Learn layers)
  capi.py    ctypes declarations of every C-ABI symbol (harness plumbing)
  hipt.py    `class Tensor` method names over device handles, for tests/bench

There is NO CPU fallback: importing works anywhere (so the build and the
symbol table can be checked without a GPU), but any compute call raises
unless the HIP library is built and an MI355X is visible.
"""
from . import capi  # noqa: F401
