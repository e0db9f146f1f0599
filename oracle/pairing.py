"""TEST INFRASTRUCTURE ONLY: BLS12-381 ate pairing in pure Python, and Groth16Verify /
PHGR13Verify restated on it (groth16.go:214-233, pinochio.go:281-378).

The reference gets Pair() from kilic/bls12-381 through kyber (curve.go:33-38); that library is
absent here.  Verification only needs *a* non-degenerate bilinear map e: G1 x G2 -> GT on the same
curve, because every check is an equality of products of pairings; this file uses the plain ate
Miller loop over Fp12 = Fp[w]/(w^12 - 2 w^6 + 2) (the textbook construction, slow: ~1 s per
pairing).  GT "Add" in the reference is multiplication in Fp12 (groth16.go:231).

A proof produced by the HIP path that passes these equations is validated by mathematics alone,
not by the oracle's restatement of the prover.
"""
from __future__ import annotations

from . import pyref as pr

P = pr.P
ATE_LOOP = pr.BLS_Z  # |z|
# Fp12 as polynomials of degree < 12 in w with w^12 = 2 w^6 - 2
_ONE = [1] + [0] * 11


def _reduce(c):
    c = list(c)
    for i in range(len(c) - 1, 11, -1):
        t = c[i]
        if t:
            c[i - 6] = (c[i - 6] + 2 * t) % P
            c[i - 12] = (c[i - 12] - 2 * t) % P
    return [x % P for x in c[:12]]


def f12_mul(a, b):
    c = [0] * 23
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                if y:
                    c[i + j] += x * y
    return _reduce(c)


def f12_add(a, b):
    return [(x + y) % P for x, y in zip(a, b)]


def f12_sub(a, b):
    return [(x - y) % P for x, y in zip(a, b)]


def f12_scalar(a, k):
    return [x * k % P for x in a]


def _deg(p):
    d = len(p) - 1
    while d and p[d] == 0:
        d -= 1
    return d


def _poly_rounded_div(a, b):
    dega, degb = _deg(a), _deg(b)
    temp = list(a)
    o = [0] * len(a)
    inv = pow(b[degb], P - 2, P)
    for i in range(dega - degb, -1, -1):
        q = temp[degb + i] * inv % P
        o[i] = (o[i] + q) % P
        for c in range(degb + 1):
            temp[c + i] = (temp[c + i] - q * b[c]) % P
    return o[: _deg(o) + 1]


def f12_inv(a):
    """Extended Euclid in Fp[w] against the modulus w^12 - 2 w^6 + 2."""
    lm, hm = [1] + [0] * 12, [0] * 13
    low, high = list(a) + [0], [2, 0, 0, 0, 0, 0, P - 2, 0, 0, 0, 0, 0, 1]
    while _deg(low):
        r = _poly_rounded_div(high, low)
        r += [0] * (13 - len(r))
        nm, new = list(hm), list(high)
        for i in range(13):
            for j in range(13 - i):
                nm[i + j] = (nm[i + j] - lm[i] * r[j]) % P
                new[i + j] = (new[i + j] - low[i] * r[j]) % P
        lm, low, hm, high = nm, new, lm, low
    inv = pow(low[0], P - 2, P)
    return [x * inv % P for x in lm[:12]]


def f12_div(a, b):
    return f12_mul(a, f12_inv(b))


def f12_pow(a, e):
    out, base = list(_ONE), list(a)
    while e:
        if e & 1:
            out = f12_mul(out, base)
        base = f12_mul(base, base)
        e >>= 1
    return out


def _embed_fp(x):
    return [x % P] + [0] * 11


def _twist(q):
    """E'(Fp2) -> E(Fp12): untwist by w^2, w^3 with the isomorphism u -> w^6 - 1."""
    (x0, x1), (y0, y1) = q
    nx = [(x0 - x1) % P] + [0] * 5 + [x1] + [0] * 5
    ny = [(y0 - y1) % P] + [0] * 5 + [y1] + [0] * 5
    w2 = [0, 0, 1] + [0] * 9
    w3 = [0, 0, 0, 1] + [0] * 8
    return (f12_div(nx, w2), f12_div(ny, w3))


def _dbl(pt):
    x, y = pt
    m = f12_div(f12_scalar(f12_mul(x, x), 3), f12_scalar(y, 2))
    nx = f12_sub(f12_mul(m, m), f12_scalar(x, 2))
    ny = f12_sub(f12_mul(m, f12_sub(x, nx)), y)
    return (nx, ny)


def _add(p1, p2):
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2 and y1 == y2:
        return _dbl(p1)
    m = f12_div(f12_sub(y2, y1), f12_sub(x2, x1))
    nx = f12_sub(f12_sub(f12_mul(m, m), x1), x2)
    ny = f12_sub(f12_mul(m, f12_sub(x1, nx)), y1)
    return (nx, ny)


def _line(p1, p2, t):
    x1, y1 = p1
    x2, y2 = p2
    xt, yt = t
    if x1 != x2:
        m = f12_div(f12_sub(y2, y1), f12_sub(x2, x1))
    elif y1 == y2:
        m = f12_div(f12_scalar(f12_mul(x1, x1), 3), f12_scalar(y1, 2))
    else:
        return f12_sub(xt, x1)
    return f12_sub(f12_mul(m, f12_sub(xt, x1)), f12_sub(yt, y1))


_FINAL_EXP = (P**12 - 1) // pr.R


def pair(p1, q2):
    """Pair(a G1, b G2) (curve.go:36-38).  Identity in either slot gives one."""
    if p1 is None or q2 is None:
        return list(_ONE)
    Q = _twist(q2)
    Pt = (_embed_fp(p1[0]), _embed_fp(p1[1]))
    R = Q
    f = list(_ONE)
    for i in range(ATE_LOOP.bit_length() - 2, -1, -1):
        f = f12_mul(f12_mul(f, f), _line(R, R, Pt))
        R = _dbl(R)
        if (ATE_LOOP >> i) & 1:
            f = f12_mul(f, _line(R, Q, Pt))
            R = _add(R, Q)
    return f12_pow(f, _FINAL_EXP)


gt_mul = f12_mul  # kyber GT "Add"


# ---------------------------------------------------------------------------------------
# Verifiers, statement by statement
# ---------------------------------------------------------------------------------------
def groth16_verify(tr, proof_A, proof_B, proof_C, io):
    """Groth16Verify (groth16.go:214-233).  Points as oracle tuples; tr needs Alpha, Beta2, IoLP,
    Gamma, Delta2; io = sol[:diff] as field elements."""
    left = pair(proof_A, proof_B)                               # :218
    a = pair(tr.Alpha, tr.Beta2)                                # :224
    b1 = None
    for i, iolp in enumerate(tr.IoLP):                          # :225-228
        b1 = pr.G1.add(b1, pr.G1.mul_pt(io[i], iolp))
    b = pair(b1, tr.Gamma)                                      # :229
    c = pair(proof_C, tr.Delta2)                                # :230
    right = gt_mul(a, gt_mul(b, c))                             # :231
    return left == right


def phgr13_verify(vk, diff, proof, io):
    """PHGR13Verify (pinochio.go:281-378).  vk: g1? av aw ay gamma bgamma bgamma2 yts vs ws ys."""
    G1, G2 = pr.G1, pr.G2

    def io_commit(grp, polys):  # computeCommitIOSolution, :390-407
        acc = None
        for i, gs in enumerate(polys):
            acc = grp.add(acc, grp.mul_pt(io[i], gs))
        return acc

    gv = G1.add(io_commit(G1, vk.vs[:diff]), proof.vss)         # :293-306
    gw = G2.add(io_commit(G2, vk.ws[:diff]), proof.wss)
    gy = G1.add(io_commit(G1, vk.ys[:diff]), proof.yss)
    left = pair(gv, gw)                                         # :312
    right = gt_mul(pair(proof.hs, vk.yts), pair(gy, G2.gen))    # :315-318
    if left != right:
        return False
    g2 = G2.gen
    if pair(proof.vass, g2) != pair(proof.vss, vk.av):          # :333-337
        return False
    if pair(proof.wass, g2) != pair(vk.aw, proof.wss):          # :340-344
        return False
    if pair(proof.yass, g2) != pair(proof.yss, vk.ay):          # :346-350
        return False
    left = pair(proof.gz, vk.gamma)                             # :359
    t1 = pair(G1.add(proof.vss, proof.yss), vk.bgamma2)         # :365-366
    t2 = pair(vk.bgamma, proof.wss)                             # :367
    return gt_mul(t1, t2) == left                               # :368-372
