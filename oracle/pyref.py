"""TEST INFRASTRUCTURE ONLY -- pure-Python big-int restatement of the playsnark hot path.

This file is part of the *oracle* (checker).  Only tests/, tests/golden/gen_golden.py,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.  The product
(playsnark_amd/) never does.

PARITY STATUS: "parity unpinned" at byte level.  The reference (nikkolasg/playsnark) is Go;
no Go toolchain and none of its curve dependencies (drand/kyber v1.1.3,
drand/kyber-bls12381 v0.2.1-0.20200920171356-02a6d1c7cc77,
kilic/bls12-381 v0.0.0-20200820230200-6b2c19996391; go.mod:5-13) exist in this image, and its
tests hold no fixed point/proof vectors (every prover test draws crypto/rand).  This
restatement is pinned by (i) the reference's fixed Fr cases (algebra_test.go:10-19,48-104,
qap_test.go:10-62, r1cs_test.go:10-30), (ii) the reference's algebraic identity tests
restated (algebra_test.go:21-35, groth16_test.go:32-107, pinocchio_test.go:23-278),
(iii) public BLS12-381 constants checked numerically (on-curve, r*G = O, r = z^4-z^2+1).

Every function cites the reference file:line it follows.  Python loops => small cases only.
"""
from __future__ import annotations

# --------------------------------------------------------------------------------------
# BLS12-381 public constants (the curve behind kyber-bls12381 / kilic, go.mod:6-8)
# --------------------------------------------------------------------------------------
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
BLS_Z = 0xD201000000010000  # |z|, the curve parameter (z is negative)

G1_X = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
G1_Y = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
G2_X0 = 0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8
G2_X1 = 0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E
G2_Y0 = 0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801
G2_Y1 = 0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE


# --------------------------------------------------------------------------------------
# Fr -- kyber.Scalar semantics used by the reference (curve.go:17-31)
# --------------------------------------------------------------------------------------
def fr(v: int) -> int:
    """Value.ToFieldElement = SetInt64 (curve.go:17-19): Euclidean reduction mod r."""
    return v % R


def fr_inv(a: int) -> int:
    return pow(a, R - 2, R)


def fr_div(a: int, b: int) -> int:
    return a * fr_inv(b) % R


# --------------------------------------------------------------------------------------
# Fp / Fp2
# --------------------------------------------------------------------------------------
def fp_inv(a: int) -> int:
    return pow(a, P - 2, P)


def fp_sqrt(a: int):
    """p = 3 mod 4 => sqrt = a^((p+1)/4); returns None if a is a non-residue."""
    s = pow(a, (P + 1) // 4, P)
    return s if s * s % P == a % P else None


def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_neg(a):
    return ((-a[0]) % P, (-a[1]) % P)


def f2_mul(a, b):
    # (a0 + a1 u)(b0 + b1 u), u^2 = -1
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_sqr(a):
    return f2_mul(a, a)


def f2_inv(a):
    d = fp_inv((a[0] * a[0] + a[1] * a[1]) % P)
    return (a[0] * d % P, (-a[1]) * d % P)


def f2_pow(a, e):
    out = (1, 0)
    while e:
        if e & 1:
            out = f2_mul(out, a)
        a = f2_sqr(a)
        e >>= 1
    return out


def f2_sqrt(a):
    """Square root in Fp2 = Fp[u]/(u^2+1) (p = 3 mod 4); None if non-residue."""
    if a == (0, 0):
        return (0, 0)
    # Algorithm 9 of Adj & Rodriguez-Henriquez (complex method)
    a1 = f2_pow(a, (P - 3) // 4)
    alpha = f2_mul(a1, f2_mul(a1, a))
    a0 = f2_mul(f2_pow(alpha, P), alpha)
    if a0 == (P - 1, 0):
        return None
    x0 = f2_mul(a1, a)
    if alpha == (P - 1, 0):
        x = f2_mul((0, 1), x0)
    else:
        b = f2_pow(f2_add((1, 0), alpha), (P - 1) // 2)
        x = f2_mul(b, x0)
    return x if f2_sqr(x) == a else None


# --------------------------------------------------------------------------------------
# Generic short-Weierstrass affine group law, y^2 = x^3 + b, over a field given by ops.
# Points are None (identity; kyber Point.Null(), curve.go:43-44) or (x, y).
# --------------------------------------------------------------------------------------
class _Grp:
    def __init__(self, add, sub, mul, inv, neg, zero, one, b, gen, name):
        self.fadd, self.fsub, self.fmul, self.finv, self.fneg = add, sub, mul, inv, neg
        self.zero, self.one, self.b, self.gen, self.name = zero, one, b, gen, name

    def on_curve(self, pt):
        if pt is None:
            return True
        x, y = pt
        return self.fmul(y, y) == self.fadd(self.fmul(self.fmul(x, x), x), self.b)

    def neg(self, pt):
        """kyber Point.Neg (groth16.go:200)."""
        return None if pt is None else (pt[0], self.fneg(pt[1]))

    def add(self, p, q):
        """kyber Point.Add (algebra.go:356, groth16.go:138)."""
        if p is None:
            return q
        if q is None:
            return p
        if p[0] == q[0]:
            if p[1] != q[1] or p[1] == self.zero:
                return None
            x2 = self.fmul(p[0], p[0])
            num = self.fadd(self.fadd(x2, x2), x2)
            lam = self.fmul(num, self.finv(self.fadd(p[1], p[1])))
        else:
            lam = self.fmul(self.fsub(q[1], p[1]), self.finv(self.fsub(q[0], p[0])))
        x3 = self.fsub(self.fsub(self.fmul(lam, lam), p[0]), q[0])
        y3 = self.fsub(self.fmul(lam, self.fsub(p[0], x3)), p[1])
        return (x3, y3)

    def mul(self, k: int, pt=None):
        """kyber Point.Mul(s, p): p == nil means the generator (curve.go:25-31 usage
        `NewG1().Mul(s, nil)`, algebra.go:373).  Scalar is reduced mod r first."""
        if pt is None:
            pt = self.gen
        k %= R
        acc = None
        for bit in bin(k)[2:] if k else "":
            acc = self.add(acc, acc)
            if bit == "1":
                acc = self.add(acc, pt)
        return acc

    def mul_pt(self, k: int, pt):
        """Scalar-mul of an explicit point which may be the identity."""
        if pt is None:
            return None
        return self.mul(k, pt)

    def msm(self, scalars, points):
        """Poly.BlindEval: serial sum of Mul + Add (algebra.go:348-359)."""
        if len(scalars) != len(points):
            raise ValueError(
                "mismatch of length between poly %d and blinded eval points %d"
                % (len(scalars), len(points))
            )
        acc = None
        for s, pt in zip(scalars, points):
            acc = self.add(acc, self.mul_pt(s, pt))
        return acc


G1 = _Grp(
    add=lambda a, b: (a + b) % P,
    sub=lambda a, b: (a - b) % P,
    mul=lambda a, b: a * b % P,
    inv=fp_inv,
    neg=lambda a: (-a) % P,
    zero=0,
    one=1,
    b=4,
    gen=(G1_X, G1_Y),
    name="G1",
)
G2 = _Grp(
    add=f2_add,
    sub=f2_sub,
    mul=f2_mul,
    inv=f2_inv,
    neg=f2_neg,
    zero=(0, 0),
    one=(1, 0),
    b=(4, 4),
    gen=((G2_X0, G2_X1), (G2_Y0, G2_Y1)),
    name="G2",
)


def check_constants():
    """Numerical self-check of the public constants."""
    z = BLS_Z
    assert R == z**4 - z**2 + 1
    assert P == ((z + 1) ** 2 * R) // 3 + (-z)  # p = (z-1)^2 r/3 + z with z negative
    assert G1.on_curve(G1.gen) and G2.on_curve(G2.gen)
    # r*G == O: (r-1)*G == -G   (mul() reduces mod r, so test through r-1)
    assert G1.mul(R - 1) == G1.neg(G1.gen)
    assert G2.mul(R - 1) == G2.neg(G2.gen)
    assert (R - 1) % (1 << 32) == 0 and (R - 1) % (1 << 33) != 0
    w = pow(7, (R - 1) >> 32, R)
    assert pow(w, 1 << 31, R) == R - 1
    return True


# --------------------------------------------------------------------------------------
# Wire formats.  kyber Point.MarshalBinary (pinochio.go:256-275) is the ZCash compressed
# form [upstream]; the C ABI also takes the ZCash uncompressed form.
# --------------------------------------------------------------------------------------
def fr_to_be32(a: int) -> bytes:
    return (a % R).to_bytes(32, "big")


def fr_from_be32(b: bytes) -> int:
    return int.from_bytes(b, "big")


def g1_to_bytes(pt) -> bytes:
    """ZCash uncompressed: x(48) || y(48) big-endian; identity = 0x40 then zeros."""
    if pt is None:
        return bytes([0x40]) + bytes(95)
    return pt[0].to_bytes(48, "big") + pt[1].to_bytes(48, "big")


def g1_from_bytes(b: bytes):
    if b[0] & 0x40:
        return None
    return (int.from_bytes(b[:48], "big"), int.from_bytes(b[48:96], "big"))


def g2_to_bytes(pt) -> bytes:
    """ZCash uncompressed: x_c1 || x_c0 || y_c1 || y_c0."""
    if pt is None:
        return bytes([0x40]) + bytes(191)
    (x0, x1), (y0, y1) = pt
    return b"".join(v.to_bytes(48, "big") for v in (x1, x0, y1, y0))


def g2_from_bytes(b: bytes):
    if b[0] & 0x40:
        return None
    v = [int.from_bytes(b[i * 48 : (i + 1) * 48], "big") for i in range(4)]
    return ((v[1], v[0]), (v[3], v[2]))


def g1_compress(pt) -> bytes:
    if pt is None:
        return bytes([0xC0]) + bytes(47)
    x, y = pt
    flag = 0x80 | (0x20 if y > (P - 1) // 2 else 0)
    out = bytearray(x.to_bytes(48, "big"))
    out[0] |= flag
    return bytes(out)


def g1_decompress(b: bytes):
    if b[0] & 0x40:
        return None
    x = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:48], "big")
    y = fp_sqrt((x * x * x + 4) % P)
    if y is None:
        raise ValueError("not on curve")
    if (y > (P - 1) // 2) != bool(b[0] & 0x20):
        y = P - y
    return (x, y)


def _f2_lex_larger(y):
    # ZCash: compare c1 first, then c0
    if y[1] != 0:
        return y[1] > (P - 1) // 2
    return y[0] > (P - 1) // 2


def g2_compress(pt) -> bytes:
    if pt is None:
        return bytes([0xC0]) + bytes(95)
    (x0, x1), y = pt
    flag = 0x80 | (0x20 if _f2_lex_larger(y) else 0)
    out = bytearray(x1.to_bytes(48, "big") + x0.to_bytes(48, "big"))
    out[0] |= flag
    return bytes(out)


def g2_decompress(b: bytes):
    if b[0] & 0x40:
        return None
    x1 = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:48], "big")
    x0 = int.from_bytes(b[48:96], "big")
    x = (x0, x1)
    y = f2_sqrt(f2_add(f2_mul(f2_sqr(x), x), (4, 4)))
    if y is None:
        raise ValueError("not on curve")
    if _f2_lex_larger(y) != bool(b[0] & 0x20):
        y = f2_neg(y)
    return (x, y)


# --------------------------------------------------------------------------------------
# algebra.go: Poly = []Element, lowest degree first
# --------------------------------------------------------------------------------------
def poly_mul(p, p2):
    """Poly.Mul, schoolbook (algebra.go:92-105)."""
    out = [0] * (len(p) + len(p2) - 1)
    for i, v1 in enumerate(p):
        for j, v2 in enumerate(p2):
            out[i + j] = (out[i + j] + v1 * v2) % R
    return out


def poly_eval(p, x):
    """Poly.Eval, Horner (algebra.go:107-115)."""
    v = 0
    for c in reversed(p):
        v = (v * x + c) % R
    return v


def poly_add(p, p2):
    """Poly.Add (algebra.go:161-178): result length = max of the two."""
    out = [0] * max(len(p), len(p2))
    for i, c in enumerate(p):
        out[i] = c % R
    for i, c in enumerate(p2):
        out[i] = (out[i] + c) % R
    return out


def poly_sub(p, p2):
    """Poly.Sub (algebra.go:180-197)."""
    out = [0] * max(len(p), len(p2))
    for i, c in enumerate(p):
        out[i] = c % R
    for i, c in enumerate(p2):
        out[i] = (out[i] - c) % R
    return out


def poly_normalize(p):
    """Poly.Normalize (algebra.go:230-239): strip high zero coefficients."""
    n = len(p)
    while n > 0 and p[n - 1] % R == 0:
        n -= 1
    return p[:n]


def poly_div2(p, p2):
    """(*Poly).Div2 long division (algebra.go:140-159).  Emits exactly one quotient
    coefficient per iteration regardless of leading zeros, so len(q) = len(p)-len(p2)+1."""
    r = list(p)
    q = []
    while len(r) > 0 and len(r) >= len(p2):
        t = fr_div(r[-1], p2[-1])
        deg_t = len(r) - len(p2)
        tpoly = [0] * (deg_t + 1)
        tpoly[-1] = t
        q = poly_add(q, tpoly)
        r = poly_sub(r, poly_mul(tpoly, p2))[: len(r) - 1]
    return q, r


def poly_div(p, p2):
    """Poly.Div synthetic division (algebra.go:119-137); operates highest-degree-first in
    the reference's own (test-only) usage."""
    out = [c % R for c in p]
    for i in range(len(p) - (len(p2) - 1)):
        out[i] = fr_div(out[i], p2[0])
        coef = out[i]
        if coef != 0:
            for j in range(1, len(p2)):
                out[i + j] = (out[i + j] + (-p2[j]) * coef) % R
    sep = len(out) - (len(p2) - 1)
    return out[:sep], out[sep:]


def interpolate(ys):
    """Interpolate (algebra.go:254-281) with lagrangeBasis (algebra.go:313-338):
    p(i+1) = ys[i]."""
    n = len(ys)
    acc = [0]
    for j in range(1, n + 1):
        basis = [1]
        den = 1
        for m in range(1, n + 1):
            if m == j:
                continue
            basis = poly_mul(basis, [(-m) % R, 1])
            den = den * fr_inv((j - m) % R) % R
        basis = [c * den % R * ys[j - 1] % R for c in basis]
        acc = poly_add(acc, basis)
    return acc


def generate_powers_commit(grp, e, shift, power):
    """GeneratePowersCommit (algebra.go:371-384): {(shift*e^i)*G} for i=0..power."""
    out = [grp.mul(shift)]
    si = 1
    for _ in range(power):
        si = si * e % R
        out.append(grp.mul(si * shift % R))
    return out


# --------------------------------------------------------------------------------------
# r1cs.go: toy circuit x^3 + x + 5 = 35
# --------------------------------------------------------------------------------------
class R1CS:
    """R1CS builder (r1cs.go:78-174): vars = [const, inputs.., outputs.., intermediates..]."""

    def __init__(self):
        self.inputs, self.outputs, self.intermediates = [], [], []
        self.left, self.right, self.out = [], [], []

    @property
    def vars(self):
        return ["const"] + self.inputs + self.outputs + self.intermediates

    def nb_io(self):
        return 1 + len(self.inputs) + len(self.outputs)

    def new_input(self, n):
        self.inputs.append(n)

    def new_output(self, n):
        self.outputs.append(n)

    def new_var(self, n):
        self.intermediates.append(n)

    def _on(self, *names):
        return [1 if v in names else 0 for v in self.vars]

    def mul(self, l, r, o):
        self.left.append(self._on(l))
        self.right.append(self._on(r))
        self.out.append(self._on(o))

    def add(self, a, b, o):
        self.left.append(self._on(a, b))
        self.right.append(self._on("const"))
        self.out.append(self._on(o))

    def add_const(self, a, k, o):
        row = self._on("const", a)
        row[0] *= k
        self.left.append(row)
        self.right.append(self._on("const"))
        self.out.append(self._on(o))


def create_r1cs():
    """createR1CS (r1cs.go:178-198)."""
    c = R1CS()
    c.new_input("x")
    c.new_output("out")
    c.new_var("u")
    c.new_var("v")
    c.new_var("w")
    c.mul("x", "x", "u")
    c.mul("u", "x", "v")
    c.add("v", "x", "w")
    c.add_const("w", 5, "out")
    return c


def create_witness(c):
    """createWitness (r1cs.go:67-76)."""
    vals = {"const": 1, "x": 3, "out": 35, "u": 9, "v": 27, "w": 30}
    return [vals[v] for v in c.vars]


# --------------------------------------------------------------------------------------
# qap.go
# --------------------------------------------------------------------------------------
class QAP:
    pass


def to_qap(c) -> QAP:
    """ToQAP (qap.go:35-65) + qapInterpolate (qap.go:67-93)."""

    def interp(m):
        cols = list(zip(*m))
        return [interpolate([fr(v) for v in col]) for col in cols]

    q = QAP()
    q.left, q.right, q.out = interp(c.left), interp(c.right), interp(c.out)
    q.nbVars, q.nbGates, q.nbIO = len(c.vars), len(c.left), c.nb_io()
    z = None
    for i in range(1, q.nbGates + 1):
        xi = [fr(-i), 1]
        z = xi if z is None else poly_mul(z, xi)
    q.z = z
    return q


def compute_aggregate_poly(q, sol):
    """computeAggregatePoly (qap.go:164-175)."""
    left, right, out = [], [], []
    for i, val in enumerate(sol):
        pv = [fr(val)]
        left = poly_add(left, poly_mul(q.left[i], pv))
        right = poly_add(right, poly_mul(q.right[i], pv))
        out = poly_add(out, poly_mul(q.out[i], pv))
    return left, right, out


def quotient(q, sol):
    """QAP.Quotient (qap.go:151-162)."""
    left, right, out = compute_aggregate_poly(q, sol)
    px = poly_sub(poly_mul(left, right), out)
    hx, rem = poly_div2(px, q.z)
    if len(poly_normalize(rem)) > 0:
        raise ArithmeticError("apocalypse")
    return hx


# --------------------------------------------------------------------------------------
# groth16.go
# --------------------------------------------------------------------------------------
class Obj:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def groth16_setup(q, alpha, beta, delta, x, gamma):
    """NewGroth16TrustedSetup (groth16.go:64-101) with the toxic waste supplied."""
    tr = Obj()
    tr.tw = Obj(Alpha=alpha, Beta=beta, Delta=delta, X=x, Gamma=gamma)
    tr.Alpha = G1.mul(alpha)
    tr.Beta, tr.Beta2 = G1.mul(beta), G2.mul(beta)
    tr.Delta, tr.Delta2 = G1.mul(delta), G2.mul(delta)
    tr.Xi = generate_powers_commit(G1, x, 1, q.nbGates - 1)
    tr.Xi2 = generate_powers_commit(G2, x, 1, q.nbGates - 1)
    tr.Gamma = G2.mul(gamma)
    diff = q.nbVars - q.nbIO

    def full_linear_poly(lo, hi, div):
        """fullLinearPoly / linearPolyForVar (groth16.go:238-264)."""
        lps, commits = [], []
        for i in range(lo, hi):
            ui, vi, wi = (poly_eval(pp[i], x) for pp in (q.left, q.right, q.out))
            lp = fr_div((wi + beta * ui + alpha * vi) % R, div)
            lps.append(lp)
            commits.append(G1.mul(lp))
        return lps, commits

    tr.tw.IoLP, tr.IoLP = full_linear_poly(0, diff, gamma)
    tr.tw.NioLP, tr.NioLP = full_linear_poly(diff, q.nbVars, delta)
    txd = fr_div(poly_eval(q.z, x), delta)
    tr.XiT = generate_powers_commit(G1, x, txd, q.nbGates - 2)
    return tr


def groth16_prove(tr, q, sol, r, s):
    """Groth16Prove (groth16.go:122-211) with r, s supplied instead of drawn (:148,158)."""

    def sum_blind(grp, polys, xi):
        total = None
        for i in range(q.nbVars):
            uix = grp.msm(polys[i], xi)
            total = grp.add(total, grp.mul_pt(fr(sol[i]), uix))
        return total

    A = sum_blind(G1, q.left, tr.Xi)
    A = G1.add(A, G1.mul_pt(r, tr.Delta))
    A = G1.add(tr.Alpha, A)
    B = sum_blind(G2, q.right, tr.Xi2)
    B = G2.add(B, G2.mul_pt(s, tr.Delta2))
    B = G2.add(tr.Beta2, B)
    diff = q.nbVars - q.nbIO
    nio = None
    for i, pt in enumerate(tr.NioLP):
        nio = G1.add(nio, G1.mul_pt(fr(sol[i + diff]), pt))
    C = nio
    h = quotient(q, sol)
    C = G1.add(C, G1.msm(h, tr.XiT))
    C = G1.add(C, G1.mul_pt(s, A))
    B1 = sum_blind(G1, q.right, tr.Xi)
    B1 = G1.add(B1, G1.mul_pt(s, tr.Delta))
    B1 = G1.add(B1, tr.Beta)
    C = G1.add(C, G1.mul_pt(r, B1))
    rsd = G1.mul_pt(r * s % R, tr.Delta)
    C = G1.add(C, G1.neg(rsd))
    return Obj(R=r, S=s, A=A, B=B, C=C)


# --------------------------------------------------------------------------------------
# pinochio.go
# --------------------------------------------------------------------------------------
def generate_eval_commit(grp, base, polys, x, shift):
    """generateEvalCommit (pinochio.go:381-388): {(shift*p_i(x))*base}."""
    return [grp.mul_pt(poly_eval(poly_normalize(pp), x) * shift % R, base) for pp in polys]


def phgr13_setup(q, s, av, aw, ay, rv, rw, beta, gamma):
    """NewPHGR13TrustedSetup (pinochio.go:93-176) with the randomness supplied in draw order."""
    ek, vk = Obj(), Obj()
    ek.gsi = generate_powers_commit(G1, s, 1, len(q.z) - 1 - 2)
    gv = G1.mul(rv)
    gw = G2.mul(rw)
    g1w = G1.mul(rw)
    ry = rv * rw % R
    gy, g2y = G1.mul(ry), G2.mul(ry)
    diff = q.nbVars - q.nbIO
    ek.vs = generate_eval_commit(G1, gv, q.left[diff:], s, 1)
    ek.ws = generate_eval_commit(G2, gw, q.right[diff:], s, 1)
    ek.ys = generate_eval_commit(G1, gy, q.out[diff:], s, 1)
    ek.vas = generate_eval_commit(G1, gv, q.left[diff:], s, av)
    ek.was = generate_eval_commit(G1, g1w, q.right[diff:], s, aw)
    ek.yas = generate_eval_commit(G1, gy, q.out[diff:], s, ay)
    ek.vbs = generate_eval_commit(G1, gv, q.left[diff:], s, beta)
    ek.wbs = generate_eval_commit(G1, g1w, q.right[diff:], s, beta)
    ek.ybs = generate_eval_commit(G1, gy, q.out[diff:], s, beta)
    bgamma = gamma * beta % R
    vk.g1 = G1.gen
    vk.av, vk.aw, vk.ay = G2.mul(av), G1.mul(aw), G2.mul(ay)
    vk.gamma, vk.bgamma, vk.bgamma2 = G2.mul(gamma), G1.mul(bgamma), G2.mul(bgamma)
    vk.yts = G2.mul_pt(poly_eval(q.z, s), g2y)
    vk.vs = generate_eval_commit(G1, gv, q.left, s, 1)
    vk.ws = generate_eval_commit(G2, gw, q.right, s, 1)
    vk.ys = generate_eval_commit(G1, gy, q.out, s, 1)
    t = Obj(beta=beta, s=s, gv=gv, gw=gw, gy=gy, ry=ry, rv=rv, rw=rw)
    return Obj(EK=ek, VK=vk, t=t)


def phgr13_prove(ek, q, sol):
    """PHGR13Prove (pinochio.go:207-254)."""
    hx = quotient(q, sol)  # inlined copy pinochio.go:209-216
    ghs = G1.msm(hx, ek.gsi)
    diff = q.nbVars - q.nbIO

    def sol_commit(grp, ec):
        acc = None
        for i, pt in enumerate(ec):
            acc = grp.add(acc, grp.mul_pt(fr(sol[diff + i]), pt))
        return acc

    out = Obj(
        hs=ghs,
        vss=sol_commit(G1, ek.vs),
        wss=sol_commit(G2, ek.ws),
        yss=sol_commit(G1, ek.ys),
        vass=sol_commit(G1, ek.vas),
        wass=sol_commit(G1, ek.was),
        yass=sol_commit(G1, ek.yas),
    )
    out.gz = G1.add(sol_commit(G1, ek.vbs), G1.add(sol_commit(G1, ek.wbs), sol_commit(G1, ek.ybs)))
    return out


# --------------------------------------------------------------------------------------
# deterministic test RNG shared by fixtures (SplitMix64; seed "playsnark")
# --------------------------------------------------------------------------------------
class SplitMix64:
    def __init__(self, seed=0x706C6179736E61726B & 0xFFFFFFFFFFFFFFFF):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def fr(self):
        v = 0
        for _ in range(4):
            v = (v << 64) | self.next()
        return v % R


if __name__ == "__main__":
    check_constants()
    print("constants ok")
