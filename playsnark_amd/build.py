"""Build helper for libplaysnark_hip.so (hipcc, gfx950 only; cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG, "csrc")


def library_path() -> str:
    """PLAYSNARK_HIP_LIB selects an alternative build of the same library (A/B experiments)."""
    return os.environ.get("PLAYSNARK_HIP_LIB") or os.path.join(_PKG, "libplaysnark_hip.so")


def _stale() -> bool:
    so = library_path()
    if not os.path.exists(so):
        return True
    t = os.path.getmtime(so)
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC)]
    srcs.append(os.path.join(_PKG, "..", "include", "playsnark_hip.h"))
    return any(os.path.getmtime(s) > t for s in srcs if os.path.isfile(s))


def build_library(force: bool = False, quiet: bool = True) -> str:
    """Compile every HIP source for gfx950 into playsnark_amd/libplaysnark_hip.so."""
    if force or _stale():
        env = dict(os.environ)
        subprocess.check_call(
            ["make", "-C", _CSRC] + (["-B"] if force else []),
            stdout=subprocess.DEVNULL if quiet else None,
            stderr=subprocess.DEVNULL if quiet else None,
            env=env,
        )
    return library_path()
