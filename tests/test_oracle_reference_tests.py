"""CPU: the reference's own tests restated against the oracle.  These are the
algebraic pins of an oracle whose byte-level parity the reference cannot pin (no Go toolchain, no
fixed vectors in the reference's tests)."""
import pytest

SEED = 0x706C6179736E61726B & 0xFFFFFFFFFFFFFFFF


def test_TestAlgebraBlindEval(co, pr):
    """algebra_test.go:21-35: BlindEval(p, {shift*x^i*G}) == (shift*p(x))*G, degree 4."""
    rng = pr.SplitMix64(SEED + 1)
    p = [rng.fr() for _ in range(5)]
    x, shift = rng.fr(), rng.fr()
    blinded = co.G1.powers_commit(x, shift, 4)
    assert co.G1.blind_eval(p, blinded) == co.G1.mul(shift * co.poly_eval(p, x) % pr.R)


def test_TestPinocchioCombine(co, pr):
    """pinocchio_test.go:11-21: same with shift = 1; also on G2."""
    rng = pr.SplitMix64(SEED + 2)
    p = [rng.fr() for _ in range(5)]
    x = rng.fr()
    assert co.G1.blind_eval(p, co.G1.powers_commit(x, 1, 4)) == co.G1.mul(co.poly_eval(p, x))
    assert co.G2.blind_eval(p, co.G2.powers_commit(x, 1, 4)) == co.G2.mul(co.poly_eval(p, x))


def test_BlindEval_length_mismatch_panics(co):
    """algebra.go:350-352."""
    with pytest.raises(ValueError, match="mismatch of length between poly 2 and blinded eval points 3"):
        co.G1.blind_eval([1, 2], co.G1.gen_points(1, 1, 3))


def test_TestAlgebraInterpolate(co, pr):
    """algebra_test.go:37-45: Interpolate(ys)(i+1) == ys[i]  -- pins the domain {1..n}."""
    rng = pr.SplitMix64(SEED + 3)
    for n in (1, 2, 5, 9):
        ys = [rng.fr() for _ in range(n)]
        p = co.interpolate(ys)
        assert [co.poly_eval(p, i + 1) for i in range(n)] == ys
        assert p == pr.interpolate(ys)


def test_TestAlgebraPolyDiv(co, pr):
    """algebra_test.go:179-195: q*p2 + r == p1 and (p1*h)/h == p1 with zero remainder."""
    rng = pr.SplitMix64(SEED + 4)
    p1 = [rng.fr() for _ in range(6)]
    p2 = [rng.fr() for _ in range(5)]
    q, r = co.poly_div2(p1, p2)
    assert pr.poly_add(co.poly_mul(q, p2), r) == p1
    h = [2, 2]
    q2, rem = co.poly_div2(co.poly_mul(p1, h), h)
    assert q2 == p1 and rem == [0]


def test_TestR1CSEquation_and_TestQAPManual(co, pr):
    """r1cs_test.go:10-30, qap_test.go:10-62 on the toy circuit."""
    c = pr.create_r1cs()
    s = pr.create_witness(c)
    dot = lambda row: sum(a * b for a, b in zip(row, s))
    assert all(dot(l) * dot(r) - dot(o) == 0 for l, r, o in zip(c.left, c.right, c.out))
    left, right, out, z = co.to_qap_dense(c.left, c.right, c.out)
    assert [co.poly_eval(left[1], g) for g in (1, 2, 3, 4)] == [1, 0, 1, 0]
    for gate in range(1, 5):
        ev = lambda polys: sum(pr.fr(v) * co.poly_eval(p, gate) for v, p in zip(s, polys)) % pr.R
        assert (ev(out) - ev(left) * ev(right)) % pr.R == 0
    assert all(len(p) == 4 for p in left + right + out) and len(z) == 5


def test_TestGroth16TrustedSetup_degree_pin(co, pr):
    """groth16_test.go:9-20: deg h == nbGates - 2, for the toy and for synthetic circuits."""
    from oracle import restate as rs

    for n in (4, 9, 32):
        c, sol = rs.synthetic_circuit(n) if n != 4 else (rs.toy_circuit()[0], [pr.fr(v) for v in rs.toy_circuit()[1]])
        h = co.quotient_from_values(*c.values(sol))[3]
        assert len(h) - 1 == n - 2


def test_TestQAPValidity_and_apocalypse(co, pr):
    """qap_test.go:64-71 IsValid; a broken witness makes Quotient panic("apocalypse")."""
    from oracle import restate as rs

    c, sol = rs.synthetic_circuit(11)
    co.quotient_from_values(*c.values(sol))
    bad = list(sol)
    bad[4] = (bad[4] + 1) % pr.R
    with pytest.raises(ArithmeticError, match="apocalypse"):
        co.quotient_from_values(*c.values(bad))


def test_TestGroth16ProofGen_synthetic(co, pr):
    """groth16_test.go:32-107 on a 20-gate synthetic circuit: dlog(A), dlog(B), dlog(C)."""
    from oracle import restate as rs

    rng = pr.SplitMix64(SEED + 6)
    c, sol = rs.synthetic_circuit(20)
    tr = rs.groth16_setup(c, *[rng.fr() for _ in range(5)])
    pf = rs.groth16_prove(tr, c, sol, rng.fr(), rng.fr())
    assert rs.groth16_dlog_check(tr, c, sol, pf) == (True, True, True)
    fast = rs.groth16_prove(tr, c, sol, pf.R, pf.S, fast=True)
    assert (fast.A, fast.B, fast.C) == (pf.A, pf.B, pf.C)


def test_TestPinocchioProofValidDivision_identities(co, pr):
    """pinocchio_test.go:33-80 and :207-227 without the pairing: hs == h(s)G,
    vss/wss/yss == (sum s_k p_k(s)) g_{v,w,y}, gz == beta (r_v v + r_w w + r_y y) G."""
    from oracle import restate as rs

    rng = pr.SplitMix64(SEED + 7)
    c, wit = rs.toy_circuit()
    sol = [pr.fr(v) for v in wit]
    st = rs.phgr13_setup(c, *[rng.fr() for _ in range(8)])
    pp = rs.phgr13_prove(st.EK, c, sol)
    t = st.t
    diff = c.nbVars - c.nbIO
    assert pp.hs == co.G1.to_b(co.G1.mul(pr.poly_eval(pp.h, t.s)))
    vks = sum(t.u[i] * sol[i] for i in range(diff, c.nbVars)) % pr.R
    wks = sum(t.v[i] * sol[i] for i in range(diff, c.nbVars)) % pr.R
    yks = sum(t.w[i] * sol[i] for i in range(diff, c.nbVars)) % pr.R
    assert pp.vss == co.G1.to_b(co.G1.mul(vks, t.gv))
    assert pp.wss == co.G2.to_b(co.G2.mul(wks, t.gw))
    assert pp.yss == co.G1.to_b(co.G1.mul(yks, t.gy))
    assert pp.vass == co.G1.to_b(co.G1.mul(vks * t.av % pr.R, t.gv))
    gz = t.beta * (t.rv * vks + t.rw * wks + t.ry * yks) % pr.R
    assert pp.gz == co.G1.to_b(co.G1.mul(gz))


def test_pippenger_matches_reference_loop(co, pr):
    """The fast CPU baseline (B1) against the serial Mul+Add loop (B0), N <= 2^10, G1 and G2."""
    rng = pr.SplitMix64(SEED + 8)
    for grp, n in ((co.G1, 1), (co.G1, 37), (co.G1, 1024), (co.G2, 200)):
        sc = [rng.fr() for _ in range(n)]
        raw = grp.gen_points(rng.fr(), rng.fr(), n)
        assert grp.msm_pippenger(co.pack_fr(sc), raw, n, 4) == grp.blind_eval(sc, raw)
