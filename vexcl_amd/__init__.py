"""vexcl_amd -- MI355X-native (gfx950) implementation of VexCL's
vector-expression hot path: hand-written HIP kernels behind a C ABI
(include/vexhip.h, vexcl_amd/csrc) + the reference's host interface
(C++ headers under vexcl/, this Python mirror for the harness).
"""
from ._capi import Error, build, lib, LIB_PATH, EXPORTS  # noqa: F401

__all__ = ["Error", "build", "lib", "LIB_PATH", "EXPORTS"]


def __getattr__(name):
    # torch is only needed by the tensor-level API
    if name in ("ops", "distributed"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
