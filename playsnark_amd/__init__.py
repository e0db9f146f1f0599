"""playsnark_amd -- MI355X-native prover hot path for nikkolasg/playsnark.

The product is `libplaysnark_hip.so` (HIP kernels for gfx950 behind the C ABI declared in
include/playsnark_hip.h).  This package is the thin Python host mirror of the reference's Go
surface for that path (Poly.BlindEval, QAP.Quotient, Groth16Prove, PHGR13Prove): ctypes only,
no arithmetic, no CPU fallback -- importing `playsnark_amd.api` fails loudly when the HIP
library has not been built, and every compute call fails when no gfx950 device is visible.
"""
from .build import build_library, library_path  # noqa: F401

__all__ = ["build_library", "library_path"]
