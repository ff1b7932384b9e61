"""proton_amd -- MI355X-native per-cell HHO local-operator assembly (ProtoN-compatible).

The product is proton_amd/lib/libproton_amd.so (hand-written HIP kernels for gfx950 behind
the C ABI of include/proton_amd.h) and the C++ host header proton_amd/host/hho.hpp that
mirrors the reference's make_hho_* / assembler API.  This Python package is test/bench
plumbing: a ctypes binding (capi) and a torch-tensor convenience layer (batch).
"""
from . import capi  # noqa: F401
from .capi import (Context, DegreeInfo, ProtonAmdError, degree_info, sizes_for,  # noqa: F401
                   QUAD_TENSOR, QUAD_FAN, STAB_NONE, STAB_NAIVE, STAB_FANCY)

__all__ = ["capi", "Context", "DegreeInfo", "ProtonAmdError", "degree_info", "sizes_for",
           "QUAD_TENSOR", "QUAD_FAN", "STAB_NONE", "STAB_NAIVE", "STAB_FANCY"]
