"""TEST INFRASTRUCTURE ONLY: Groth16 / PHGR13 setup and prove composed from the C oracle's
primitives, statement by statement after groth16.go / pinochio.go, for sparse circuits.

Where the reference nests per-variable polynomials (sumBlind, groth16.go:134-141) this file uses
the aggregate polynomials of computeAggregatePoly (qap.go:164-175) -- the same group element by
linearity; tests/test_oracle_*.py check the literal nested form (oracle/pyref.py) against this
one on the toy circuit.  Parity status: see oracle/oracle.h ("parity unpinned" at byte level).
"""
from __future__ import annotations

from . import coracle as co
from . import pyref as pr

R = pr.R


class SparseR1CS:
    """Rows = gates; each row a list of (variable index, integer coefficient).  Variable order
    [const, inputs.., outputs.., intermediates..] as r1cs.go:132-144."""

    def __init__(self, nb_vars, nb_io, left, right, out):
        self.nbVars, self.nbIO = nb_vars, nb_io
        self.left, self.right, self.out = left, right, out
        self.nbGates = len(left)

    def dense(self):
        def d(rows):
            m = [[0] * self.nbVars for _ in rows]
            for g, row in enumerate(rows):
                for c, v in row:
                    m[g][c] += v
            return m

        return d(self.left), d(self.right), d(self.out)

    def values(self, sol_fr):
        """(L.s, R.s, O.s) in Fr: what the aggregate polynomials interpolate on {1..n}."""
        def mv(rows):
            return [sum(v * sol_fr[c] for c, v in row) % R for row in rows]

        return mv(self.left), mv(self.right), mv(self.out)


def toy_circuit():
    """createR1CS + createWitness (r1cs.go:178-198, 67-76) in sparse form."""
    c = pr.create_r1cs()
    rows = lambda m: [[(j, v) for j, v in enumerate(r) if v] for r in m]
    return SparseR1CS(len(c.vars), c.nb_io(), rows(c.left), rows(c.right), rows(c.out)), pr.create_witness(c)


def synthetic_circuit(n_gates: int, x0: int = 3):
    """the toy's gate pattern (Mul, Mul, Add, AddConst) tiled, each block's result
    feeding the next block's x.  Variables: const, x (input), out (output), then intermediates;
    nbIO = 3.  Witness values are reduced mod r (they outgrow int64 after a few blocks)."""
    CONST, X, OUT = 0, 1, 2
    nvars = 3
    vals = {CONST: 1, X: x0 % R}
    left, right, out = [], [], []
    cur = X

    def new_var():
        nonlocal nvars
        nvars += 1
        return nvars - 1

    targets = []
    g = 0
    u = v = w = None
    while g < n_gates:
        kind = g % 4
        last = g == n_gates - 1
        o = OUT if last else new_var()
        if kind == 0:  # u = x*x
            left.append([(cur, 1)]); right.append([(cur, 1)]); out.append([(o, 1)])
            vals[o] = vals[cur] * vals[cur] % R
            u = o
        elif kind == 1:  # v = u*x
            left.append([(u, 1)]); right.append([(cur, 1)]); out.append([(o, 1)])
            vals[o] = vals[u] * vals[cur] % R
            v = o
        elif kind == 2:  # w = v + x
            left.append([(v, 1), (cur, 1)]); right.append([(CONST, 1)]); out.append([(o, 1)])
            vals[o] = (vals[v] + vals[cur]) % R
            w = o
        else:  # next x = w + 5
            left.append([(CONST, 5), (w, 1)]); right.append([(CONST, 1)]); out.append([(o, 1)])
            vals[o] = (vals[w] + 5) % R
            cur = o
        g += 1
    sol = [vals[i] for i in range(nvars)]
    return SparseR1CS(nvars, 3, left, right, out), sol


def lagrange_at(n: int, x: int):
    """l_j(x) for the nodes 1..n (the basis behind Interpolate, algebra.go:254-338), O(n)."""
    x %= R
    pre = [1] * (n + 1)
    for i in range(1, n + 1):
        pre[i] = pre[i - 1] * ((x - i) % R) % R
    suf = [1] * (n + 2)
    for i in range(n, 0, -1):
        suf[i] = suf[i + 1] * ((x - i) % R) % R
    fact = [1] * (n + 1)
    for i in range(1, n + 1):
        fact[i] = fact[i - 1] * i % R
    ifact = [1] * (n + 1)
    ifact[n] = pr.fr_inv(fact[n])
    for i in range(n, 0, -1):
        ifact[i - 1] = ifact[i] * i % R
    out = []
    for j in range(1, n + 1):
        iden = ifact[j - 1] * ifact[n - j] % R  # 1 / ((j-1)! (n-j)!), sign (-1)^(n-j)
        if (n - j) & 1:
            iden = R - iden
        out.append(pre[j - 1] * suf[j + 1] % R * iden % R)
    return out, pre[n]  # basis values, z(x)


def var_poly_evals(c: SparseR1CS, x: int):
    """u_i(x), v_i(x), w_i(x) for every variable i (= qap.left[i].Eval(x) etc.) and z(x)."""
    lj, zx = lagrange_at(c.nbGates, x)
    res = []
    for rows in (c.left, c.right, c.out):
        acc = [0] * c.nbVars
        for g, row in enumerate(rows):
            for col, v in row:
                acc[col] = (acc[col] + v * lj[g]) % R
        res.append(acc)
    return res[0], res[1], res[2], zx


def bit_circuit(n_gates: int, seed: int = 7):
    """n booleanity gates b_i * b_i = b_i over n + 1 variables [const, b_1 .. b_n] (nbIO = nbVars - 1 puts the
    reference's `diff` split right after the constant): the commonest gate of real circuits, and a witness
    that fits the reference's Vector = []int (algebra.go:13) -- random bits.  Returns (circuit, int witness)."""
    rng = pr.SplitMix64(seed)
    bits, word = [], 0
    for i in range(n_gates):
        if i % 64 == 0:
            word = rng.next()
        bits.append((word >> (i % 64)) & 1)
    rows = [[(1 + i, 1)] for i in range(n_gates)]
    return SparseR1CS(n_gates + 1, n_gates, rows, rows, rows), [1] + bits


class Bag:
    def __init__(self, **kw):
        self.__dict__.update(kw)


# ---------------------------------------------------------------------------------------
# Groth16 (groth16.go)
# ---------------------------------------------------------------------------------------
def groth16_setup(c: SparseR1CS, alpha, beta, delta, x, gamma):
    """NewGroth16TrustedSetup (groth16.go:64-101), toxic waste supplied.  Points as raw bytes."""
    n = c.nbGates
    u, v, w, zx = var_poly_evals(c, x)
    diff = c.nbVars - c.nbIO
    one = 1

    def full_linear_poly(lo, hi, div):  # groth16.go:254-264
        lps = [pr.fr_div((w[i] + beta * u[i] + alpha * v[i]) % R, div) for i in range(lo, hi)]
        return lps, b"".join(co.G1.to_b(co.G1.mul(lp)) for lp in lps)

    tw = Bag(Alpha=alpha, Beta=beta, Delta=delta, X=x, Gamma=gamma)
    tw.IoLP, iolp = full_linear_poly(0, diff, gamma)
    tw.NioLP, niolp = full_linear_poly(diff, c.nbVars, delta)
    txd = pr.fr_div(zx, delta)
    return Bag(
        tw=tw,
        Alpha=co.G1.to_b(co.G1.mul(alpha)),
        Beta=co.G1.to_b(co.G1.mul(beta)),
        Beta2=co.G2.to_b(co.G2.mul(beta)),
        Delta=co.G1.to_b(co.G1.mul(delta)),
        Delta2=co.G2.to_b(co.G2.mul(delta)),
        Gamma=co.G2.to_b(co.G2.mul(gamma)),
        Xi=co.G1.powers_commit(x, one, n - 1),
        Xi2=co.G2.powers_commit(x, one, n - 1),
        IoLP=iolp,
        NioLP=niolp,
        XiT=co.G1.powers_commit(x, txd, n - 2) if n >= 2 else b"",
    )


def groth16_prove(tr, c: SparseR1CS, sol_fr, r, s, fast=False):
    """Groth16Prove (groth16.go:122-211) with r, s supplied.  fast=True swaps the serial
    BlindEval loops for the oracle's Pippenger (same group elements) for larger circuits."""
    yA, yB, yC = c.values(sol_fr)
    A_c, B_c, C_c, h = co.quotient_from_values(yA, yB, yC)  # raises "apocalypse"

    def blind(group, coeffs, raw):
        if fast:
            return group.msm_pippenger(co.pack_fr(coeffs), raw, len(coeffs), 8)
        return group.blind_eval(coeffs, raw)

    G1, G2 = co.G1, co.G2
    A = blind(G1, A_c, tr.Xi)                                   # :146
    A = G1.add(A, G1.mul(r, G1.from_b(tr.Delta)))               # :149-151
    A = G1.add(G1.from_b(tr.Alpha), A)                          # :152
    B = blind(G2, B_c, tr.Xi2)                                  # :157
    B = G2.add(B, G2.mul(s, G2.from_b(tr.Delta2)))              # :159-160
    B = G2.add(G2.from_b(tr.Beta2), B)                          # :161
    diff = c.nbVars - c.nbIO
    n_nio = len(tr.NioLP) // 96
    nio = blind(G1, [sol_fr[i + diff] for i in range(n_nio)], tr.NioLP)   # :176-178
    Cp = nio
    htd = blind(G1, h, tr.XiT) if h else None                   # :184-185
    Cp = G1.add(Cp, htd)
    Cp = G1.add(Cp, _mulpt(G1, s, A))                           # :189-190
    B1 = blind(G1, B_c, tr.Xi)                                  # :192
    B1 = G1.add(B1, G1.mul(s, G1.from_b(tr.Delta)))             # :193-194
    B1 = G1.add(B1, G1.from_b(tr.Beta))                         # :195
    Cp = G1.add(Cp, _mulpt(G1, r, B1))                          # :196-197
    rsd = G1.mul(r * s % R, G1.from_b(tr.Delta))                # :199
    Cp = G1.add(Cp, _neg(rsd))                                  # :200
    return Bag(R=r, S=s, A=G1.to_b(A), B=G2.to_b(B), C=G1.to_b(Cp), coeffs=(A_c, B_c, C_c, h))


def _mulpt(group, k, pt):
    return None if pt is None else group.mul(k, pt)


def _neg(pt):
    if pt is None:
        return None
    x, y = pt
    return (x, (-y) % pr.P) if isinstance(y, int) else (x, ((-y[0]) % pr.P, (-y[1]) % pr.P))


def groth16_dlog_check(tr, c: SparseR1CS, sol_fr, proof):
    """TestGroth16ProofGen (groth16_test.go:32-107): recompute the discrete logs of A, B, C from
    the toxic waste and compare [dlog]G with the proof elements."""
    tw = tr.tw
    u, v, w, zx = var_poly_evals(c, tw.X)
    G1, G2 = co.G1, co.G2
    a = (sum(u[i] * sol_fr[i] for i in range(c.nbVars)) + proof.R * tw.Delta + tw.Alpha) % R
    b = (sum(v[i] * sol_fr[i] for i in range(c.nbVars)) + proof.S * tw.Delta + tw.Beta) % R
    okA = G1.to_b(G1.mul(a)) == proof.A
    okB = G2.to_b(G2.mul(b)) == proof.B
    diff = c.nbVars - c.nbIO
    res = 0
    for i in range(diff, c.nbVars):
        res += (w[i] + tw.Beta * u[i] + tw.Alpha * v[i]) % R * pr.fr_div(sol_fr[i], tw.Delta)
    h = proof.coeffs[3]
    res = (res + pr.fr_div(pr.poly_eval(h, tw.X) * zx % R, tw.Delta)) % R
    # + s*a + r*b - r*s*delta, all in the exponent
    cdlog = (res + proof.S * a + proof.R * b - proof.R * proof.S % R * tw.Delta) % R
    okC = G1.to_b(G1.mul(cdlog)) == proof.C
    return okA, okB, okC


# ---------------------------------------------------------------------------------------
# PHGR13 (pinochio.go)
# ---------------------------------------------------------------------------------------
def phgr13_setup(c: SparseR1CS, s, av, aw, ay, rv, rw, beta, gamma):
    """NewPHGR13TrustedSetup (pinochio.go:93-176), randomness supplied in draw order."""
    n = c.nbGates
    G1, G2 = co.G1, co.G2
    u, v, w, zs = var_poly_evals(c, s)
    gv, gw, g1w = G1.mul(rv), G2.mul(rw), G1.mul(rw)
    ry = rv * rw % R
    gy, g2y = G1.mul(ry), G2.mul(ry)
    diff = c.nbVars - c.nbIO

    def eval_commit(group, base, evals, shift):  # generateEvalCommit, pinochio.go:381-388
        return b"".join(group.to_b(_mulpt(group, e * shift % R, base)) for e in evals)

    ek = Bag(
        gsi=G1.powers_commit(s, 1, n - 2) if n >= 2 else b"",   # :101 (z.Degree()-2 = n-2)
        vs=eval_commit(G1, gv, u[diff:], 1),
        ws=eval_commit(G2, gw, v[diff:], 1),
        ys=eval_commit(G1, gy, w[diff:], 1),
        vas=eval_commit(G1, gv, u[diff:], av),
        was=eval_commit(G1, g1w, v[diff:], aw),
        yas=eval_commit(G1, gy, w[diff:], ay),
        vbs=eval_commit(G1, gv, u[diff:], beta),
        wbs=eval_commit(G1, g1w, v[diff:], beta),
        ybs=eval_commit(G1, gy, w[diff:], beta),
    )
    t = Bag(beta=beta, s=s, gv=gv, gw=gw, gy=gy, ry=ry, rv=rv, rw=rw, av=av, aw=aw, ay=ay, gamma=gamma,
            u=u, v=v, w=w, zs=zs)
    bgamma = gamma * beta % R
    vk = Bag(  # PHGR13VerifKey, pinochio.go:64-91,140-160 (points as tuples for oracle.pairing)
        g1=G1.mul(1), av=G2.mul(av), aw=G1.mul(aw), ay=G2.mul(ay), gamma=G2.mul(gamma),
        bgamma=G1.mul(bgamma), bgamma2=G2.mul(bgamma), yts=_mulpt(G2, zs, g2y),
        vs=[_mulpt(G1, e, gv) for e in u], ws=[_mulpt(G2, e, gw) for e in v], ys=[_mulpt(G1, e, gy) for e in w],
    )
    return Bag(EK=ek, VK=vk, t=t)


def phgr13_prove(ek, c: SparseR1CS, sol_fr, fast=False):
    """PHGR13Prove (pinochio.go:207-254)."""
    yA, yB, yC = c.values(sol_fr)
    _, _, _, h = co.quotient_from_values(yA, yB, yC)            # :209-216
    G1, G2 = co.G1, co.G2

    def blind(group, coeffs, raw):
        if not coeffs:
            return None
        if fast:
            return group.msm_pippenger(co.pack_fr(coeffs), raw, len(coeffs), 8)
        return group.blind_eval(coeffs, raw)

    diff = c.nbVars - c.nbIO

    def sol_commit(group, raw):  # computeSolCommit, :222-229
        k = len(raw) // group.nb
        return blind(group, [sol_fr[diff + i] for i in range(k)], raw)

    out = Bag(
        hs=G1.to_b(blind(G1, h, ek.gsi)),                       # :218
        vss=G1.to_b(sol_commit(G1, ek.vs)),                     # :231
        wss=G2.to_b(sol_commit(G2, ek.ws)),                     # :232
        yss=G1.to_b(sol_commit(G1, ek.ys)),                     # :233
        vass=G1.to_b(sol_commit(G1, ek.vas)),                   # :234
        wass=G1.to_b(sol_commit(G1, ek.was)),                   # :235
        yass=G1.to_b(sol_commit(G1, ek.yas)),                   # :236
    )
    gz = G1.add(sol_commit(G1, ek.vbs), G1.add(sol_commit(G1, ek.wbs), sol_commit(G1, ek.ybs)))  # :239-242
    out.gz = G1.to_b(gz)
    out.h = h
    return out
