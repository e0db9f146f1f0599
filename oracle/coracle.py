"""TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/liboracle.so (see oracle/oracle.h).

All values cross this binding as Python ints (Fr), None/(x, y) tuples (points, same
representation as oracle.pyref) or raw big-endian byte strings (bulk arrays).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

from . import pyref as pr

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OK, ERR_LENGTH, ERR_NOT_DIVISIBLE, ERR_ENCODING = 0, -1, -2, -3


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("oracle.c", "oracle.h", "curve_tmpl.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.or_now.restype = C.c_double
    return _LIB


def _buf(n):
    return C.create_string_buffer(n)


# ---- Fr ----
def fr_from_i64(v):
    o = _buf(32)
    lib().or_fr_from_i64(C.c_int64(v), o)
    return int.from_bytes(o.raw, "big")


def _fr2(fn, a, b):
    o = _buf(32)
    getattr(lib(), fn)(pr.fr_to_be32(a), pr.fr_to_be32(b), o)
    return int.from_bytes(o.raw, "big")


def fr_add(a, b):
    return _fr2("or_fr_add", a, b)


def fr_sub(a, b):
    return _fr2("or_fr_sub", a, b)


def fr_mul(a, b):
    return _fr2("or_fr_mul", a, b)


def fr_inv(a):
    o = _buf(32)
    lib().or_fr_inv(pr.fr_to_be32(a), o)
    return int.from_bytes(o.raw, "big")


def pack_fr(v):
    return b"".join(pr.fr_to_be32(x) for x in v)


def unpack_fr(b):
    return [int.from_bytes(b[i : i + 32], "big") for i in range(0, len(b), 32)]


# ---- Poly ----
def poly_mul(a, b):
    o = _buf(32 * (len(a) + len(b) - 1))
    lib().or_poly_mul(pack_fr(a), C.c_size_t(len(a)), pack_fr(b), C.c_size_t(len(b)), o)
    return unpack_fr(o.raw)


def poly_eval(p, x):
    o = _buf(32)
    lib().or_poly_eval(pack_fr(p), C.c_size_t(len(p)), pr.fr_to_be32(x), o)
    return int.from_bytes(o.raw, "big")


def poly_div2(p, d):
    q = _buf(32 * (len(p) - len(d) + 1))
    r = _buf(32 * max(len(d) - 1, 1))
    rc = lib().or_poly_div2(pack_fr(p), C.c_size_t(len(p)), pack_fr(d), C.c_size_t(len(d)), q, r)
    if rc:
        raise ValueError(rc)
    return unpack_fr(q.raw), unpack_fr(r.raw)[: len(d) - 1]


def interpolate(ys):
    o = _buf(32 * len(ys))
    lib().or_interpolate(pack_fr(ys), C.c_size_t(len(ys)), o)
    return unpack_fr(o.raw)


# ---- points ----
class _G:
    def __init__(self, name, nbytes, to_b, from_b):
        self.name, self.nb, self.to_b, self.from_b = name, nbytes, to_b, from_b

    def pack(self, pts):
        return b"".join(self.to_b(p) for p in pts)

    def unpack(self, b):
        return [self.from_b(b[i : i + self.nb]) for i in range(0, len(b), self.nb)]

    def mul(self, k, pt="gen"):
        """Point.Mul(s, p); pt == 'gen' is the reference's nil (generator)."""
        o = _buf(self.nb)
        arg = None if pt == "gen" else self.to_b(pt)
        rc = getattr(lib(), f"or_{self.name}_mul")(pr.fr_to_be32(k), arg, o)
        assert rc == 0, rc
        return self.from_b(o.raw)

    def add(self, a, b):
        o = _buf(self.nb)
        rc = getattr(lib(), f"or_{self.name}_add")(self.to_b(a), self.to_b(b), o)
        assert rc == 0, rc
        return self.from_b(o.raw)

    def on_curve(self, a):
        return bool(getattr(lib(), f"or_{self.name}_on_curve")(self.to_b(a)))

    def blind_eval(self, scalars, points_bytes, npoints=None):
        """Poly.BlindEval (algebra.go:348-359); raises on length mismatch like the panic."""
        npoints = len(points_bytes) // self.nb if npoints is None else npoints
        o = _buf(self.nb)
        rc = getattr(lib(), f"or_{self.name}_blind_eval")(
            pack_fr(scalars), C.c_size_t(len(scalars)), points_bytes, C.c_size_t(npoints), o
        )
        if rc == ERR_LENGTH:
            raise ValueError(
                "mismatch of length between poly %d and blinded eval points %d" % (len(scalars), npoints)
            )
        assert rc == 0, rc
        return self.from_b(o.raw)

    def blind_eval_i64(self, scalars, points_bytes):
        arr = (C.c_int64 * len(scalars))(*scalars)
        o = _buf(self.nb)
        rc = getattr(lib(), f"or_{self.name}_blind_eval_i64")(arr, points_bytes, C.c_size_t(len(scalars)), o)
        assert rc == 0, rc
        return self.from_b(o.raw)

    def msm_pippenger(self, scalars_bytes, points_bytes, n, threads=1):
        o = _buf(self.nb)
        rc = getattr(lib(), f"or_{self.name}_msm_pippenger")(
            scalars_bytes, points_bytes, C.c_size_t(n), C.c_int(threads), o
        )
        assert rc == 0, rc
        return self.from_b(o.raw)

    def powers_commit(self, e, shift, power):
        """GeneratePowersCommit (algebra.go:371-384) -> raw bytes of power+1 points."""
        o = _buf(self.nb * (power + 1))
        rc = getattr(lib(), f"or_{self.name}_powers_commit")(
            pr.fr_to_be32(e), pr.fr_to_be32(shift), C.c_size_t(power), o
        )
        assert rc == 0, rc
        return o.raw

    def gen_points(self, k0, q, n):
        """Synthetic vector out[i] = (k0 + i*q)*G as raw bytes."""
        o = _buf(self.nb * n)
        rc = getattr(lib(), f"or_{self.name}_gen_points")(pr.fr_to_be32(k0), pr.fr_to_be32(q), C.c_size_t(n), o)
        assert rc == 0, rc
        return o.raw


G1 = _G("g1", 96, pr.g1_to_bytes, pr.g1_from_bytes)
G2 = _G("g2", 192, pr.g2_to_bytes, pr.g2_from_bytes)


def g1_compress(pt):
    o = _buf(48)
    assert lib().or_g1_compress(pr.g1_to_bytes(pt), o) == 0
    return o.raw


def g1_decompress(b):
    o = _buf(96)
    rc = lib().or_g1_decompress(b, o)
    if rc:
        raise ValueError("bad encoding")
    return pr.g1_from_bytes(o.raw)


def g2_compress(pt):
    o = _buf(96)
    assert lib().or_g2_compress(pr.g2_to_bytes(pt), o) == 0
    return o.raw


def g2_decompress(b):
    o = _buf(192)
    rc = lib().or_g2_decompress(b, o)
    if rc:
        raise ValueError("bad encoding")
    return pr.g2_from_bytes(o.raw)


# ---- QAP ----
def to_qap_dense(L, Rm, O):
    """ToQAP (qap.go:35-65) on dense int matrices (rows = gates). Returns (left,right,out,z)."""
    n, m = len(L), len(L[0])
    flat = lambda M: (C.c_int64 * (n * m))(*[v for row in M for v in row])
    bufs = [_buf(32 * n * m) for _ in range(3)]
    z = _buf(32 * (n + 1))
    rc = lib().or_to_qap_dense(flat(L), flat(Rm), flat(O), C.c_size_t(n), C.c_size_t(m), *bufs, z)
    assert rc == 0
    polys = []
    for b in bufs:
        v = unpack_fr(b.raw)
        polys.append([v[i * n : (i + 1) * n] for i in range(m)])
    return polys[0], polys[1], polys[2], unpack_fr(z.raw)


def aggregate_poly(polys, sol_fr):
    n, m = len(polys[0]), len(polys)
    o = _buf(32 * n)
    lib().or_aggregate_poly(b"".join(pack_fr(p) for p in polys), C.c_size_t(n), C.c_size_t(m), pack_fr(sol_fr), o)
    return unpack_fr(o.raw)


def quotient_from_aggregates(A, B, Cc, z):
    n = len(A)
    h = _buf(32 * max(n - 1, 1))
    rc = lib().or_quotient_from_aggregates(pack_fr(A), pack_fr(B), pack_fr(Cc), pack_fr(z), C.c_size_t(n), h)
    if rc == ERR_NOT_DIVISIBLE:
        raise ArithmeticError("apocalypse")
    assert rc == 0, rc
    return unpack_fr(h.raw)[: n - 1]


def quotient_from_values(yA, yB, yC):
    """Aggregate polys by interpolating L.s, R.s, O.s on {1..n}, then the literal
    Mul/Sub/Div2 of qap.go:151-162.  Returns (A, B, C, h)."""
    n = len(yA)
    bufs = [_buf(32 * n) for _ in range(3)]
    h = _buf(32 * max(n - 1, 1))
    rc = lib().or_quotient_from_values(pack_fr(yA), pack_fr(yB), pack_fr(yC), C.c_size_t(n), *bufs, h)
    if rc == ERR_NOT_DIVISIBLE:
        raise ArithmeticError("apocalypse")
    assert rc == 0, rc
    return tuple(unpack_fr(b.raw) for b in bufs) + (unpack_fr(h.raw)[: n - 1],)


def fast_quotient_bytes(yA: bytes, yB: bytes, yC: bytes, n: int):
    """B1-h: (A, B, C, h) as big-endian byte strings by the quasi-linear CPU algorithm (n up to 2^20)."""
    bufs = [_buf(32 * n) for _ in range(3)]
    h = _buf(32 * max(n - 1, 1))
    rc = lib().or_fast_quotient(yA, yB, yC, C.c_size_t(n), *bufs, h)
    if rc == ERR_NOT_DIVISIBLE:
        raise ArithmeticError("apocalypse")
    assert rc == 0, rc
    return tuple(b.raw for b in bufs) + (h.raw[: 32 * (n - 1)],)


def fast_quotient(yA, yB, yC):
    n = len(yA)
    return tuple(unpack_fr(b) for b in fast_quotient_bytes(pack_fr(yA), pack_fr(yB), pack_fr(yC), n))
