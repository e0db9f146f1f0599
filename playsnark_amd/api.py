"""Python host mirror of the reference's Go surface for the prover hot path.

Same names, argument meaning and error behaviour as the reference, bound to the C ABI:

    Poly.BlindEval(zero, blindedPoint)   algebra.go:348-359  -> Poly.BlindEval(points)
    QAP.Quotient(sol)                    qap.go:151-162      -> QAP.Quotient(sol)
    Groth16Prove(tr, q, sol)             groth16.go:122-211  -> Groth16Prove(tr, q, sol, r, s)
    PHGR13Prove(ek, qap, solution)       pinochio.go:207-254 -> PHGR13Prove(ek, qap, solution)

The reference signals errors by panicking; here the same conditions raise
`LengthMismatch` (message of algebra.go:351) and `Apocalypse` ("apocalypse", qap.go:159).
No arithmetic happens in this file: everything is a call into libplaysnark_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Optional, Sequence

from . import _lib
from ._lib import PS_G1, PS_G2, lib

G1, G2 = PS_G1, PS_G2
R_ORDER = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001  # the scalar field: encodings only, no arithmetic here
FMT_AFFINE, FMT_COMPRESSED = _lib.PS_FMT_AFFINE, _lib.PS_FMT_COMPRESSED
_WIRE = {PS_G1: 96, PS_G2: 192}


class PlaysnarkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"playsnark_hip error {code}: {msg}")
        self.code = code


class LengthMismatch(ValueError):
    """panic(fmt.Sprintf("mismatch of length between poly %d and blinded eval points %d")) algebra.go:351"""


class Apocalypse(ArithmeticError):
    """panic("apocalypse") qap.go:159 / pinochio.go:215: the witness does not satisfy the QAP."""


def _check(rc: int):
    if rc == _lib.PS_OK:
        return
    msg = (lib.ps_last_error() or b"").decode()
    if rc == _lib.PS_ERR_LENGTH:
        raise LengthMismatch(msg)
    if rc == _lib.PS_ERR_NOT_DIVISIBLE:
        raise Apocalypse("apocalypse")
    raise PlaysnarkError(rc, msg)


def device_count() -> int:
    return lib.ps_device_count()


class Context:
    """One GPU + stream + workspace (ps_ctx)."""

    def __init__(self, device: int = 0):
        h = C.c_void_p()
        _check(lib.ps_ctx_create(device, C.byref(h)))
        self._h = h
        self.device = device

    def sync(self):
        _check(lib.ps_ctx_sync(self._h))

    @property
    def stream(self) -> int:
        return lib.ps_ctx_stream(self._h)

    def set_window(self, bits: int):
        _check(lib.ps_msm_set_window(self._h, bits))

    def set_tables(self, enable: bool):
        """Whether the provers build window tables for their CRS arrays (default: yes, for keys of >= 32 points)."""
        _check(lib.ps_ctx_set_tables(self._h, int(enable)))

    def microbench_mad(self) -> float:
        """Measured v_mad_u64_u32 issue rate in lane-operations per second (ps_microbench_mad, ~1 ms)."""
        v = C.c_double(0.0)
        _check(lib.ps_microbench_mad(self._h, C.byref(v)))
        return v.value

    def set_table_budget(self, nbytes: int):
        """Bytes a prover may spend on ONE window table (negative: automatic from the free device memory); a table that
        does not fit is skipped and the sums over that array take the plain plan."""
        _check(lib.ps_ctx_set_table_budget(self._h, nbytes))

    def set_slice(self, entries: int):
        _check(lib.ps_msm_set_slice(self._h, entries))

    def set_tail(self, mode: int):
        """0 automatic, 1 chains (the long sums' tail), 2 trees of lane-cooperative additions (the short sums' tail)."""
        _check(lib.ps_msm_set_tail(self._h, mode))

    STAGES = ("digits", "scan", "scatter", "queue", "accumulate", "fixup", "reduce")

    def set_timing(self, enable: bool):
        _check(lib.ps_ctx_set_timing(self._h, int(enable)))

    def last_stage_ms(self) -> dict:
        ms = (C.c_float * len(self.STAGES))()
        _check(lib.ps_msm_last_stage_ms(self._h, ms))
        return dict(zip(self.STAGES, list(ms)))

    def last_prove_phase_ms(self) -> dict:
        """Host wall-clock split of the last Groth16Prove / PHGR13Prove on this context."""
        ms = (C.c_float * 4)()
        _check(lib.ps_prove_last_phase_ms(self._h, ms))
        return dict(zip(("quotient", "prep_or_h_sum", "sums", "total"), list(ms)))

    def last_msm_info(self) -> dict:
        info = _lib.MsmInfo()
        _check(lib.ps_msm_last_info(self._h, C.byref(info)))
        return {k: getattr(info, k) for k, _ in info._fields_}

    def close(self):
        if getattr(self, "_h", None):
            lib.ps_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Points:
    """A device-resident []G1 / []G2 slice (ps_points), e.g. Groth16Setup.Xi (groth16.go:43)."""

    def __init__(self, ctx: Context, handle, owner=None):
        self.ctx, self._h, self._owner = ctx, handle, owner

    @classmethod
    def upload(cls, ctx: Context, group: int, raw: bytes, fmt: int = _lib.PS_FMT_AFFINE) -> "Points":
        wb = _WIRE[group] if fmt == _lib.PS_FMT_AFFINE else _WIRE[group] // 2
        assert len(raw) % wb == 0
        h = C.c_void_p()
        _check(lib.ps_points_upload(ctx._h, group, raw, len(raw) // wb, fmt, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_scalars(cls, ctx: Context, group: int, k: "Poly") -> "Points":
        """out[i] = k[i] * G: the commit loop of GeneratePowersCommit (algebra.go:371-384)."""
        h = C.c_void_p()
        _check(lib.ps_points_from_scalars(ctx._h, group, k._h, C.byref(h)))
        return cls(ctx, h)

    @property
    def group(self) -> int:
        return lib.ps_points_group(self._h)

    def __len__(self):
        return lib.ps_points_len(self._h)

    def download(self, first: int = 0, n: Optional[int] = None) -> bytes:
        n = len(self) - first if n is None else n
        buf = C.create_string_buffer(_WIRE[self.group] * max(n, 1))
        _check(lib.ps_points_download(self.ctx._h, self._h, first, n, buf))
        return buf.raw[: _WIRE[self.group] * n]

    def download_compressed(self, first: int = 0, n: Optional[int] = None) -> bytes:
        """kyber MarshalBinary form (ZCash compressed, 48 B G1 / 96 B G2), compressed on the GPU."""
        n = len(self) - first if n is None else n
        wb = _WIRE[self.group] // 2
        buf = C.create_string_buffer(wb * max(n, 1))
        _check(lib.ps_points_download_fmt(self.ctx._h, self._h, first, n, _lib.PS_FMT_COMPRESSED, buf))
        return buf.raw[: wb * n]

    # Flat key files (the reference has no persistence): a 16-byte header
    # b"PSNK" | u8 version | u8 group | u8 format | u8 0 | u64 count (little-endian), then the points.
    def save(self, path: str, compressed: bool = True):
        import struct

        fmt = _lib.PS_FMT_COMPRESSED if compressed else _lib.PS_FMT_AFFINE
        with open(path, "wb") as f:
            f.write(b"PSNK" + struct.pack("<BBBBQ", 1, self.group, fmt, 0, len(self)))
            f.write(self.download_compressed() if compressed else self.download())

    @classmethod
    def load(cls, ctx: Context, path: str) -> "Points":
        import struct

        with open(path, "rb") as f:
            head = f.read(16)
            if len(head) != 16 or head[:4] != b"PSNK":
                raise PlaysnarkError(-3, f"{path}: not a playsnark key file")
            version, group, fmt, _, count = struct.unpack("<BBBBQ", head[4:])
            if version != 1 or group not in (PS_G1, PS_G2) or fmt not in (_lib.PS_FMT_AFFINE, _lib.PS_FMT_COMPRESSED):
                raise PlaysnarkError(-3, f"{path}: unsupported key file header")
            wb = _WIRE[group] if fmt == _lib.PS_FMT_AFFINE else _WIRE[group] // 2
            raw = f.read()
        if len(raw) != wb * count:
            raise PlaysnarkError(-1, f"{path}: {len(raw)} bytes of points, header says {count} x {wb}")
        return cls.upload(ctx, group, raw, fmt)

    def slice(self, first: int, n: int) -> "Points":
        h = C.c_void_p()
        _check(lib.ps_points_slice(self._h, first, n, C.byref(h)))
        return Points(self.ctx, h, owner=self)

    def to_lagrange(self, qap: "QAP", nodes: int = 0) -> "Points":
        """This array read as {x^i P} (a monomial-form CRS array of the reference's setups: Xi, Xi2 with nodes = 0; XiT, gsi
        with nodes = 1) -> {l_j(x) P} on the QAP's interpolation nodes (1..n, or n+1..2n-1), without the secret point
        (ps_points_monomial_to_lagrange): the one-time conversion that puts a reference-made key on the prover's fast route."""
        h = C.c_void_p()
        _check(lib.ps_points_monomial_to_lagrange(qap.ctx._h, qap._h, self._h, nodes, C.byref(h)))
        return Points(qap.ctx, h)

    def precompute(self, window_bits: int = 0) -> "Points":
        """Build the window table 2^(c w) P of this resident array once (ps_points_precompute): later sums over it,
        or over slices of it, share one bucket set.  Returns self."""
        _check(lib.ps_points_precompute(self.ctx._h, self._h, window_bits))
        return self

    def drop_table(self) -> "Points":
        """Release the window table (ps_points_precompute with window_bits = -1); sums go back to the plain plan."""
        _check(lib.ps_points_precompute(self.ctx._h, self._h, -1))
        return self

    @property
    def table_window(self) -> int:
        return lib.ps_points_table_window(self._h)

    def in_subgroup(self) -> bool:
        """[r]P = O for every point (what kyber's UnmarshalBinary enforces on the Go side [upstream])."""
        ok = C.c_int(0)
        _check(lib.ps_points_check_subgroup(self.ctx._h, self._h, C.byref(ok)))
        return bool(ok.value)

    def free(self):
        if self._h:
            lib.ps_points_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _be32(v: int) -> bytes:
    return int(v).to_bytes(32, "big")


class Poly:
    """type Poly []Element (algebra.go:89), device-resident (ps_scalars).  Also stands in for
    Vector = []Value (algebra.go:13) through from_values (Value.ToFieldElement, curve.go:17-19)."""

    def __init__(self, ctx: Context, handle, owner=None):
        self.ctx, self._h, self._owner = ctx, handle, owner

    @classmethod
    def upload(cls, ctx: Context, coeffs) -> "Poly":
        raw = coeffs if isinstance(coeffs, (bytes, bytearray)) else b"".join(_be32(c) for c in coeffs)
        h = C.c_void_p()
        _check(lib.ps_scalars_upload(ctx._h, bytes(raw), len(raw) // 32, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_values(cls, ctx: Context, values: Sequence[int]) -> "Poly":
        arr = (C.c_int64 * len(values))(*values)
        h = C.c_void_p()
        _check(lib.ps_scalars_upload_i64(ctx._h, arr, len(values), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_device_be32(cls, ctx: Context, device_ptr: int, n: int) -> "Poly":
        h = C.c_void_p()
        _check(lib.ps_scalars_from_device_be32(ctx._h, C.c_void_p(device_ptr), n, C.byref(h)))
        return cls(ctx, h)

    def __len__(self):
        return lib.ps_scalars_len(self._h)

    def download_bytes(self, first: int = 0, n: Optional[int] = None) -> bytes:
        n = len(self) - first if n is None else n
        buf = C.create_string_buffer(32 * max(n, 1))
        _check(lib.ps_scalars_download(self.ctx._h, self._h, first, n, buf))
        return buf.raw[: 32 * n]

    def download(self, first: int = 0, n: Optional[int] = None):
        raw = self.download_bytes(first, n)
        return [int.from_bytes(raw[i : i + 32], "big") for i in range(0, len(raw), 32)]

    def slice(self, first: int, n: int) -> "Poly":
        h = C.c_void_p()
        _check(lib.ps_scalars_slice(self._h, first, n, C.byref(h)))
        return Poly(self.ctx, h, owner=self)

    def Mul(self, p2: "Poly") -> "Poly":
        """func (p Poly) Mul(p2 Poly) Poly (algebra.go:92-105)."""
        h = C.c_void_p()
        _check(lib.ps_poly_mul(self.ctx._h, self._h, p2._h, C.byref(h)))
        return Poly(self.ctx, h)

    def BlindEval(self, blindedPoint: Points) -> bytes:
        """func (p Poly) BlindEval(zero Commit, blindedPoint []Commit) Commit (algebra.go:348).
        The group is the dynamic type of the points, as in the reference;
        `zero` is implied.  Returns the affine big-endian point (96 B G1 / 192 B G2)."""
        out = C.create_string_buffer(_WIRE[blindedPoint.group])
        _check(lib.ps_msm(self.ctx._h, blindedPoint._h, self._h, out))
        return out.raw

    def free(self):
        if self._h:
            lib.ps_scalars_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def msm_launch(ctx: Context, points: Points, scalars: Poly):
    """Asynchronous BlindEval: enqueue on the context stream (ps_msm_launch)."""
    _check(lib.ps_msm_launch(ctx._h, points._h, scalars._h))


def msm_finish(ctx: Context, group: int) -> bytes:
    out = C.create_string_buffer(_WIRE[group])
    _check(lib.ps_msm_finish(ctx._h, out))
    return out.raw


def blind_eval_host(ctx: Context, points: Points, scalars) -> bytes:
    """Seam S1 as the cgo shim calls it (BlindEvalHIP -> ps_msm_be32 / ps_msm_i64): Poly.BlindEval (algebra.go:348-359) with
    the scalars still in HOST memory -- 32-byte big-endian rows as `bytes`, or a numpy int64 array of witness values
    (Value.ToFieldElement, curve.go:17-19).  The upload over PCIe is inside the call; nothing of the caller's is kept."""
    out = C.create_string_buffer(_WIRE[points.group])
    if isinstance(scalars, (bytes, bytearray)):
        _check(lib.ps_msm_be32(ctx._h, points._h, bytes(scalars), len(scalars) // 32, out))
    else:
        import numpy as np

        arr = np.ascontiguousarray(scalars, dtype=np.int64)
        _check(lib.ps_msm_i64(ctx._h, points._h, arr.ctypes.data_as(C.c_void_p), arr.size, out))
    return out.raw


def msm_multi(ctx: Context, points: list, scalars: Poly) -> list:
    """[scalars.BlindEval(p) for p in points] with one digit sort shared by all arrays (ps_msm_multi):
    the nine computeSolCommit calls of PHGR13Prove (pinochio.go:231-241) have this shape."""
    k = len(points)
    outs = [C.create_string_buffer(_WIRE[p.group]) for p in points]
    pa = (C.c_void_p * max(k, 1))(*[p._h for p in points])
    oa = (C.c_void_p * max(k, 1))(*[C.cast(o, C.c_void_p) for o in outs])
    _check(lib.ps_msm_multi(ctx._h, pa, k, scalars._h, oa))
    return [o.raw for o in outs]


def point_convert(group: int, raw: bytes, in_fmt: int, out_fmt: int) -> bytes:
    """One point between the ZCash uncompressed and compressed forms (kyber MarshalBinary)."""
    out = C.create_string_buffer(_WIRE[group] if out_fmt == _lib.PS_FMT_AFFINE else _WIRE[group] // 2)
    _check(lib.ps_point_convert(group, in_fmt, out_fmt, raw, out))
    return out.raw


def points_lincomb(group: int, points: bytes, scalars: Sequence[int]) -> bytes:
    """sum_i scalars[i] * points[i] for a handful of affine points, on the host (ps_points_lincomb)."""
    out = C.create_string_buffer(_WIRE[group])
    _check(lib.ps_points_lincomb(group, points, b"".join(_be32(k % R_ORDER) for k in scalars), len(scalars), out))
    return out.raw


def msm_multi_device(ctxs: Sequence[Context], shards: Sequence[Points], scalars: Sequence["Poly"]) -> bytes:
    """One sum whose index-range shards live on several devices of this process (ps_msm_multi_device)."""
    k = len(ctxs)
    ca = (C.c_void_p * k)(*[c._h for c in ctxs])
    pa = (C.c_void_p * k)(*[p._h for p in shards])
    sa = (C.c_void_p * k)(*[s._h for s in scalars])
    out = C.create_string_buffer(_WIRE[shards[0].group])
    _check(lib.ps_msm_multi_device(ca, pa, sa, k, out))
    return out.raw


def points_sum(group: int, raw: bytes) -> bytes:
    """Sum of affine points: folds the per-GPU partial sums after the RCCL gather."""
    out = C.create_string_buffer(_WIRE[group])
    _check(lib.ps_points_sum(group, raw, len(raw) // _WIRE[group], out))
    return out.raw


# -----------------------------------------------------------------------------------------
# QAP / provers
# -----------------------------------------------------------------------------------------
def _csr(rows: Sequence[Sequence[tuple]]):
    """rows[g] = [(col, int_value), ...] -> (Csr struct, keep-alive arrays)."""
    row_ptr = [0]
    cols, vals = [], []
    for r in rows:
        for col, v in r:
            if v != 0:
                cols.append(col)
                vals.append(v)
        row_ptr.append(len(cols))
    a = (C.c_uint32 * len(row_ptr))(*row_ptr)
    b = (C.c_uint32 * max(len(cols), 1))(*cols)
    c = (C.c_int64 * max(len(vals), 1))(*vals)
    s = _lib.Csr(C.cast(a, C.c_void_p), C.cast(b, C.c_void_p), C.cast(c, C.c_void_p))
    return s, (a, b, c)


def dense_to_rows(m: Sequence[Sequence[int]]):
    return [[(j, v) for j, v in enumerate(row) if v != 0] for row in m]


class QAP:
    """type QAP (qap.go:10-27), kept in the sparse evaluation form the hot path needs: the three
    R1CS matrices in CSR (rows = gates) on the reference's domain {1..n}."""

    def __init__(self, ctx: Context, nbVars: int, nbIO: int, left_rows, right_rows, out_rows):
        self.ctx = ctx
        self.nbVars, self.nbIO, self.nbGates = nbVars, nbIO, len(left_rows)
        keep = []
        structs = []
        for rows in (left_rows, right_rows, out_rows):
            s, k = _csr(rows)
            structs.append(s)
            keep.append(k)
        h = C.c_void_p()
        _check(lib.ps_qap_create(ctx._h, self.nbGates, nbVars, nbIO, C.byref(structs[0]), C.byref(structs[1]),
                                 C.byref(structs[2]), C.byref(h)))
        self._h = h

    @classmethod
    def from_csr(cls, ctx: Context, nbVars: int, nbIO: int, left, right, out) -> "QAP":
        """The three matrices as (row_ptr, col, val) triples of array-likes exposing the buffer protocol
        (numpy uint32 / uint32 / int64): no per-row Python lists, for circuits of millions of gates."""
        self = cls.__new__(cls)
        self.ctx, self.nbVars, self.nbIO = ctx, nbVars, nbIO
        self.nbGates = len(left[0]) - 1
        structs, keep = [], []
        for row_ptr, col, val in (left, right, out):
            arrs = []
            for a, ct in ((row_ptr, C.c_uint32), (col, C.c_uint32), (val, C.c_int64)):
                mv = memoryview(a).cast("B")
                if mv.nbytes % C.sizeof(ct):
                    raise ValueError("CSR array of the wrong element type")
                buf = (ct * max(mv.nbytes // C.sizeof(ct), 1))()
                C.memmove(buf, mv.tobytes(), mv.nbytes)
                arrs.append(buf)
            if len(arrs[0]) != self.nbGates + 1:
                raise LengthMismatch("row_ptr arrays of different lengths")
            keep.append(arrs)
            structs.append(_lib.Csr(*[C.cast(a, C.c_void_p) for a in arrs]))
        h = C.c_void_p()
        _check(lib.ps_qap_create(ctx._h, self.nbGates, nbVars, nbIO, C.byref(structs[0]), C.byref(structs[1]),
                                 C.byref(structs[2]), C.byref(h)))
        self._h = h
        return self

    @classmethod
    def from_dense(cls, ctx: Context, nbVars: int, nbIO: int, left, right, out) -> "QAP":
        """ToQAP(circuit R1CS) (qap.go:35) from the dense matrices of r1cs.go:99-101."""
        return cls(ctx, nbVars, nbIO, dense_to_rows(left), dense_to_rows(right), dense_to_rows(out))

    def computeAggregatePoly(self, sol: Poly):
        """(left, right, out Poly) of qap.go:164-175 plus h in one pass."""
        hs = [C.c_void_p() for _ in range(4)]
        _check(lib.ps_qap_quotient(self.ctx._h, self._h, sol._h, *[C.byref(h) for h in hs]))
        return tuple(Poly(self.ctx, h) for h in hs)

    def interpolate(self, sol: Poly, which: int) -> Poly:
        """One of the three aggregate polynomials of computeAggregatePoly (qap.go:164-175): 0 left, 1 right, 2 out."""
        h = C.c_void_p()
        _check(lib.ps_qap_interpolate(self.ctx._h, self._h, sol._h, which, C.byref(h)))
        return Poly(self.ctx, h)

    def computeAB(self, sol: Poly):
        """(left, right, h): the Groth16 route -- A and B as coefficient vectors, h = floor(A*B / z); C is never
        interpolated (what Groth16Prove needs, groth16.go:146-185)."""
        hs = [C.c_void_p() for _ in range(3)]
        _check(lib.ps_qap_quotient(self.ctx._h, self._h, sol._h, C.byref(hs[0]), C.byref(hs[1]), None, C.byref(hs[2])))
        return tuple(Poly(self.ctx, h) for h in hs)

    def IsValid(self, sol: Poly) -> bool:
        """func (q *QAP) IsValid(sol Vector) bool (qap.go:107): does z divide left*right - out?  (A wrong number of
        solution variables panics in the reference's sanityCheck, qap.go:177-189: PlaysnarkError here.)"""
        ok = C.c_int(0)
        _check(lib.ps_qap_is_valid(self.ctx._h, self._h, sol._h, C.byref(ok)))
        return bool(ok.value)

    def Quotient(self, sol: Poly) -> Poly:
        """func (q QAP) Quotient(sol Vector) Poly (qap.go:151): raises Apocalypse when the
        remainder is non-zero."""
        h = C.c_void_p()
        _check(lib.ps_qap_quotient(self.ctx._h, self._h, sol._h, None, None, None, C.byref(h)))
        return Poly(self.ctx, h)

    def free(self):
        if getattr(self, "_h", None):
            lib.ps_qap_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Groth16Setup:
    """The prover's part of type Groth16Setup (groth16.go:30-61)."""

    def __init__(self, Alpha: bytes, Beta: bytes, Delta: bytes, Beta2: bytes, Delta2: bytes, Xi: Points,
                 Xi2: Points, NioLP: Points, XiT: Points, LXi: Optional[Points] = None, LXi2: Optional[Points] = None,
                 LXiT: Optional[Points] = None):
        """LXi, LXi2, LXiT: the same CRS in Lagrange form (l_j(x) G1, l_j(x) G2, lambda_k(x) t(x)/delta G1), as the device
        setup emits it; with all three the prover needs no polynomial in coefficient form (ps_groth16_pk)."""
        self.Alpha, self.Beta, self.Delta, self.Beta2, self.Delta2 = Alpha, Beta, Delta, Beta2, Delta2
        self.Xi, self.Xi2, self.NioLP, self.XiT = Xi, Xi2, NioLP, XiT
        self.LXi, self.LXi2, self.LXiT = LXi, LXi2, LXiT

    def monomial_only(self) -> "Groth16Setup":
        """The key as the reference's NewGroth16TrustedSetup makes it: the monomial arrays alone."""
        return Groth16Setup(self.Alpha, self.Beta, self.Delta, self.Beta2, self.Delta2, self.Xi, self.Xi2, self.NioLP, self.XiT)

    def with_lagrange(self, qap: "QAP") -> "Groth16Setup":
        """The same key with its Lagrange-form arrays computed from the monomial ones ALONE (no toxic waste, groth16.go:13-14):
        what a key made by the reference's NewGroth16TrustedSetup needs once to prove on the fast route."""
        return Groth16Setup(self.Alpha, self.Beta, self.Delta, self.Beta2, self.Delta2, self.Xi, self.Xi2, self.NioLP, self.XiT,
                            self.Xi.to_lagrange(qap, 0), self.Xi2.to_lagrange(qap, 0), self.XiT.to_lagrange(qap, 1))

    def _struct(self):
        pk = _lib.Groth16Pk()
        for name, src in (("alpha", self.Alpha), ("beta", self.Beta), ("delta", self.Delta),
                          ("beta2", self.Beta2), ("delta2", self.Delta2)):
            C.memmove(getattr(pk, name), src, len(src))
        pk.xi, pk.xi2, pk.nio_lp, pk.xi_t = self.Xi._h, self.Xi2._h, self.NioLP._h, self.XiT._h
        if self.LXi is not None and self.LXi2 is not None and self.LXiT is not None:
            pk.lxi, pk.lxi2, pk.lxi_t = self.LXi._h, self.LXi2._h, self.LXiT._h
        return pk


def NewGroth16TrustedSetup(qap: "QAP", alpha: int, beta: int, delta: int, x: int, gamma: int):
    """func NewGroth16TrustedSetup(qap QAP) Groth16Setup (groth16.go:64) on the device; the toxic
    waste is drawn by the caller (the reference draws it at :67-84).  Returns (Groth16Setup for the
    prover, dict with Gamma / IoLP for the verifier)."""
    tw = _lib.Groth16Toxic()
    for name, v in (("alpha", alpha), ("beta", beta), ("delta", delta), ("x", x), ("gamma", gamma)):
        C.memmove(getattr(tw, name), _be32(v), 32)
    crs = _lib.Groth16Crs()
    _check(lib.ps_groth16_setup(qap.ctx._h, qap._h, C.byref(tw), C.byref(crs)))
    pts = {f: Points(qap.ctx, C.c_void_p(getattr(crs, f))) for f in ("xi", "xi2", "io_lp", "nio_lp", "xi_t", "lxi", "lxi2", "lxi_t")}
    tr = Groth16Setup(bytes(crs.alpha), bytes(crs.beta), bytes(crs.delta), bytes(crs.beta2), bytes(crs.delta2),
                      pts["xi"], pts["xi2"], pts["nio_lp"], pts["xi_t"], pts["lxi"], pts["lxi2"], pts["lxi_t"])
    return tr, {"Gamma": bytes(crs.gamma), "IoLP": pts["io_lp"]}


class Groth16Proof:
    """type Groth16Proof (groth16.go:106-118); tp = (R, S) as supplied."""

    def __init__(self, R, S, A, B, Cc):
        self.R, self.S, self.A, self.B, self.C = R, S, A, B, Cc


def Groth16Prove(tr: Groth16Setup, q: QAP, sol: Poly, r: int, s: int) -> Groth16Proof:
    """func Groth16Prove(tr Groth16Setup, q QAP, sol Vector) Groth16Proof (groth16.go:122).
    r and s are drawn by the caller (the reference draws them at :148 and :158)."""
    A = C.create_string_buffer(96)
    B = C.create_string_buffer(192)
    Cc = C.create_string_buffer(96)
    pk = tr._struct()
    _check(lib.ps_groth16_prove(q.ctx._h, C.byref(pk), q._h, sol._h, _be32(r), _be32(s), A, B, Cc))
    return Groth16Proof(r, s, A.raw, B.raw, Cc.raw)


def Groth16ProveMulti(devices: Sequence[tuple], r: int, s: int) -> Groth16Proof:
    """Groth16Prove over several devices of this process, each holding only its index range of the CRS arrays
    (ps_groth16_prove_multi).  devices[d] = (Groth16Setup with the d-th ranges, QAP on that device, solution on that device)."""
    arr = (_lib.Groth16Device * len(devices))()
    keep = []
    for d, (tr, q, sol) in enumerate(devices):
        pk = tr._struct()
        keep.append(pk)
        arr[d].ctx, arr[d].qap, arr[d].sol, arr[d].pk = q.ctx._h, q._h, sol._h, pk
    A, B, Cc = C.create_string_buffer(96), C.create_string_buffer(192), C.create_string_buffer(96)
    _check(lib.ps_groth16_prove_multi(arr, len(devices), _be32(r), _be32(s), A, B, Cc))
    return Groth16Proof(r, s, A.raw, B.raw, Cc.raw)


class PHGR13EvalKey:
    """type PHGR13EvalKey (pinochio.go:37-62)."""

    FIELDS = ("vs", "ws", "ys", "vas", "was", "yas", "gsi", "vbs", "wbs", "ybs")

    def __init__(self, **kw):
        for f in self.FIELDS:
            setattr(self, f, kw[f])
        self.lgsi = kw.get("lgsi")  # optional: gsi in Lagrange form on the nodes n+1..2n-1 (ps_phgr13_ek.lgsi)

    def monomial_only(self) -> "PHGR13EvalKey":
        return PHGR13EvalKey(**{f: getattr(self, f) for f in self.FIELDS})

    def with_lagrange(self, qap: "QAP") -> "PHGR13EvalKey":
        """The same key with gsi's Lagrange form on the nodes n+1..2n-1 computed from gsi alone (no toxic waste)."""
        return PHGR13EvalKey(lgsi=self.gsi.to_lagrange(qap, 1), **{f: getattr(self, f) for f in self.FIELDS})

    def _struct(self):
        ek = _lib.Phgr13Ek()
        for f in self.FIELDS:
            setattr(ek, f, getattr(self, f)._h)
        if self.lgsi is not None:
            ek.lgsi = self.lgsi._h
        return ek


class PHGR13VerifKey:
    """type PHGR13VerifKey (pinochio.go:64-91): fixed points as affine bytes, vs/ws/ys over all variables."""

    FIXED = ("av", "aw", "ay", "gamma", "bgamma", "bgamma2", "yts")

    def __init__(self, vs: Points, ws: Points, ys: Points, **fixed):
        self.vs, self.ws, self.ys = vs, ws, ys
        for f in self.FIXED:
            setattr(self, f, fixed[f])

    def fixed_points(self) -> dict:
        return {f: getattr(self, f) for f in self.FIXED}


def NewPHGR13TrustedSetup(qap: "QAP", s: int, av: int, aw: int, ay: int, rv: int, rw: int, beta: int, gamma: int):
    """func NewPHGR13TrustedSetup(qap QAP) PHGR13Setup (pinochio.go:93) on the device; the toxic waste
    is drawn by the caller, in the reference's draw order (:99-138).  Returns (EK, VK)."""
    tw = _lib.Phgr13Toxic()
    for name, v in (("s", s), ("av", av), ("aw", aw), ("ay", ay), ("rv", rv), ("rw", rw), ("beta", beta), ("gamma", gamma)):
        C.memmove(getattr(tw, name), _be32(v), 32)
    crs = _lib.Phgr13Crs()
    _check(lib.ps_phgr13_setup(qap.ctx._h, qap._h, C.byref(tw), C.byref(crs)))
    pts = {f: Points(qap.ctx, C.c_void_p(getattr(crs, f))) for f in PHGR13EvalKey.FIELDS + ("vk_vs", "vk_ws", "vk_ys", "lgsi")}
    ek = PHGR13EvalKey(lgsi=pts["lgsi"], **{f: pts[f] for f in PHGR13EvalKey.FIELDS})
    vk = PHGR13VerifKey(pts["vk_vs"], pts["vk_ws"], pts["vk_ys"], **{f: bytes(getattr(crs, f)) for f in PHGR13VerifKey.FIXED})
    return ek, vk


class PHGR13Proof:
    """type PHGR13Proof (pinochio.go:180-203)."""

    FIELDS = ("vss", "vass", "wss", "wass", "yss", "yass", "hs", "gz")

    def __init__(self, raw: _lib.Phgr13Proof):
        for f in self.FIELDS:
            setattr(self, f, bytes(getattr(raw, f)))


def PHGR13Prove(ek: PHGR13EvalKey, qap: QAP, solution: Poly) -> PHGR13Proof:
    """func PHGR13Prove(ek PHGR13EvalKey, qap QAP, solution Vector) PHGR13Proof (pinochio.go:207)."""
    out = _lib.Phgr13Proof()
    s = ek._struct()
    _check(lib.ps_phgr13_prove(qap.ctx._h, C.byref(s), qap._h, solution._h, C.byref(out)))
    return PHGR13Proof(out)


# -----------------------------------------------------------------------------------------
# verifiers
# -----------------------------------------------------------------------------------------
def pairing_equal(a1: bytes, b1: bytes, a2: bytes, b2: bytes) -> bool:
    """Pair(a1, b1).Equal(Pair(a2, b2)) (curve.go:36-38); host-only."""
    eq = C.c_int(0)
    _check(lib.ps_pairing_equal(a1, b1, a2, b2, C.byref(eq)))
    return bool(eq.value)


def Groth16Verify(ctx: Context, Alpha: bytes, Beta2: bytes, Gamma: bytes, Delta2: bytes, IoLP: Points, p: Groth16Proof,
                  io: Poly) -> bool:
    """func Groth16Verify(tr Groth16Setup, q QAP, p Groth16Proof, io Vector) bool (groth16.go:214)."""
    vk = _lib.Groth16Vk()
    for name, src in (("alpha", Alpha), ("beta2", Beta2), ("gamma", Gamma), ("delta2", Delta2)):
        C.memmove(getattr(vk, name), src, len(src))
    vk.io_lp = IoLP._h
    ok = C.c_int(0)
    _check(lib.ps_groth16_verify(ctx._h, C.byref(vk), io._h, p.A, p.B, p.C, C.byref(ok)))
    return bool(ok.value)


def PHGR13Verify(ctx: Context, vk_points: dict, vs_io: Points, ws_io: Points, ys_io: Points, p: "PHGR13Proof", io: Poly) -> bool:
    """func PHGR13Verify(vk PHGR13VerifKey, qap QAP, p PHGR13Proof, io Vector) bool (pinochio.go:281).
    vk_points: av, aw, ay, gamma, bgamma, bgamma2, yts as affine bytes."""
    vk = _lib.Phgr13Vk()
    for name in ("av", "aw", "ay", "gamma", "bgamma", "bgamma2", "yts"):
        C.memmove(getattr(vk, name), vk_points[name], len(vk_points[name]))
    vk.vs_io, vk.ws_io, vk.ys_io = vs_io._h, ws_io._h, ys_io._h
    raw = _lib.Phgr13Proof()
    for f in PHGR13Proof.FIELDS:
        C.memmove(getattr(raw, f), getattr(p, f), len(getattr(p, f)))
    ok = C.c_int(0)
    _check(lib.ps_phgr13_verify(ctx._h, C.byref(vk), io._h, C.byref(raw), C.byref(ok)))
    return bool(ok.value)
