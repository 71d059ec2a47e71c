"""cuadmm_amd: MI355X-native SDP-ADMM iteration engine (drop-in for cuADMM's SDPSolver hot path).

The product is the C-ABI shared library built from cuadmm_amd/csrc (HIP kernels for gfx950 + host
engine); this package is a thin ctypes mirror of the reference's SDPSolver / Problem interface.
"""
from ._lib import CuadmmError, LIB_PATH, load  # noqa: F401
from .solver import Problem, SDPSolver  # noqa: F401

__all__ = ["SDPSolver", "Problem", "CuadmmError", "load", "LIB_PATH"]
