"""GPU parity of the QAP quotient pipeline and the two provers (through the C ABI) against the
oracle's restatement of qap.go / groth16.go / pinochio.go.  Bit-exact on every coefficient and
every proof byte; the oracle computes the aggregate polynomials by Lagrange interpolation and h by
the reference's own schoolbook Mul + long division Div2, the GPU by SpMV + Newton-basis
interpolation + NTT products + power-series division -- agreement is the parity claim.
"""
import pytest

pytestmark = pytest.mark.gpu

SEED = 0x706C6179736E61726B & 0xFFFFFFFFFFFFFFFF


def _upload_circuit(ps_api, ctx, c):
    return ps_api.QAP(ctx, c.nbVars, c.nbIO, c.left, c.right, c.out)


@pytest.mark.parametrize("na,nb", [(1, 1), (2, 3), (5, 4), (64, 64), (100, 29), (1000, 1000), (4097, 33)])
def test_poly_mul_matches_schoolbook(ps_api, ctx, co, pr, na, nb):
    """Poly.Mul (algebra.go:92-105); includes TestAlgebraPolyMul's fixed case shape."""
    rng = pr.SplitMix64(SEED + na * 7 + nb)
    a = [rng.fr() for _ in range(na)]
    b = [rng.fr() for _ in range(nb)]
    got = ps_api.Poly.upload(ctx, a).Mul(ps_api.Poly.upload(ctx, b)).download()
    assert got == co.poly_mul(a, b)


def test_poly_mul_reference_fixed_case(ps_api, ctx):
    """TestAlgebraPolyMul (algebra_test.go:76-104): (1+2x)(3+x^2) = 3+6x+x^2+2x^3."""
    got = ps_api.Poly.upload(ctx, [1, 2]).Mul(ps_api.Poly.upload(ctx, [3, 0, 1])).download()
    assert got == [3, 6, 1, 2]


def test_quotient_toy_matches_survey_fixture(ps_api, ctx, pr):
    """Config #1 (x^3+x+5=35): h, A from SURVEY.md 8c / tests/golden; TestGroth16TrustedSetup's
    length pin deg h == n-2 (groth16_test.go:16-19)."""
    from oracle import restate as rs

    c, wit = rs.toy_circuit()
    q = _upload_circuit(ps_api, ctx, c)
    sol = ps_api.Poly.from_values(ctx, wit)
    A, B, Cc, h = q.computeAggregatePoly(sol)
    assert len(h) == q.nbGates - 1
    assert h.download() == [
        0x4D491A377113A8DACCD13AB0066BE558E27E6D5755543D54AAAAAAA9FFFFFFFD,
        0x46D8580827A75AC891152076B08D923C24F3E43AB8E28D8D9C71C71BD5555567,
        0x0CE1845E92D89C2477783472ABBCA6397B15123938E35F8E1C71C71C55555552,
    ]
    ref = pr.to_qap(pr.create_r1cs())
    assert (A.download(), B.download(), Cc.download()) == tuple(pr.compute_aggregate_poly(ref, wit))
    assert q.Quotient(sol).download() == pr.quotient(ref, wit)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 17, 63, 64, 65, 100, 128, 200, 256, 511, 512])
def test_quotient_matches_reference_algorithm(ps_api, ctx, co, pr, n):
    """A, B, C, h for the tiled synthetic circuit against interpolation + Mul + Div2."""
    from oracle import restate as rs

    c, sol = rs.synthetic_circuit(n, x0=3 + n)
    q = _upload_circuit(ps_api, ctx, c)
    A, B, Cc, h = q.computeAggregatePoly(ps_api.Poly.upload(ctx, sol))
    want = co.quotient_from_values(*c.values(sol))
    assert A.download() == want[0]
    assert B.download() == want[1]
    assert Cc.download() == want[2]
    assert h.download() == want[3]
    # QAP.Quotient alone takes the other route to the same polynomial: values of h on the nodes
    # n+1..2n-1 (one convolution per aggregate), then a single values -> monomial conversion
    if n >= 2:
        assert q.Quotient(ps_api.Poly.upload(ctx, sol)).download() == want[3]


def test_quotient_apocalypse_and_sanity(ps_api, ctx, pr):
    """panic("apocalypse") (qap.go:158-160) on a witness that violates one gate; sanityCheck
    (qap.go:177-189) on a wrong-length solution."""
    from oracle import restate as rs

    c, sol = rs.synthetic_circuit(37)
    q = _upload_circuit(ps_api, ctx, c)
    bad = list(sol)
    bad[5] = (bad[5] + 1) % pr.R
    with pytest.raises(ps_api.Apocalypse):
        q.Quotient(ps_api.Poly.upload(ctx, bad))
    with pytest.raises(ps_api.PlaysnarkError):
        q.Quotient(ps_api.Poly.upload(ctx, sol[:-1]))
    assert len(q.Quotient(ps_api.Poly.upload(ctx, sol))) == 36


def test_quotient_large_n_properties(ps_api, ctx, co, pr):
    """n = 2^14: size-independent checks.  A(j) = (L.s)_j at sampled gates (TestAlgebraInterpolate's
    property), and A(t)B(t) - C(t) = h(t) z(t) at a random point (the QAP identity the reference
    checks in the clear, pinocchio_test.go:147-155)."""
    from oracle import restate as rs

    n = 1 << 14
    c, sol = rs.synthetic_circuit(n)
    q = _upload_circuit(ps_api, ctx, c)
    A, B, Cc, h = (p.download() for p in q.computeAggregatePoly(ps_api.Poly.upload(ctx, sol)))
    yA, yB, yC = c.values(sol)
    for j in (1, 2, 3, 1000, n // 2, n - 1, n):
        assert co.poly_eval(A, j) == yA[j - 1]
        assert co.poly_eval(B, j) == yB[j - 1]
        assert co.poly_eval(Cc, j) == yC[j - 1]
    t = pr.SplitMix64(SEED + 5).fr()
    _, zt = rs.lagrange_at(n, t)
    lhs = (co.poly_eval(A, t) * co.poly_eval(B, t) - co.poly_eval(Cc, t)) % pr.R
    assert lhs == co.poly_eval(h, t) * zt % pr.R
    assert len(h) == n - 1
    assert q.Quotient(ps_api.Poly.upload(ctx, sol)).download() == h  # h-only route, same bytes


def _points(ps_api, ctx, group, raw):
    return ps_api.Points.upload(ctx, group, raw)


def _groth16_pk(ps_api, ctx, tr):
    return ps_api.Groth16Setup(
        tr.Alpha, tr.Beta, tr.Delta, tr.Beta2, tr.Delta2,
        _points(ps_api, ctx, ps_api.G1, tr.Xi), _points(ps_api, ctx, ps_api.G2, tr.Xi2),
        _points(ps_api, ctx, ps_api.G1, tr.NioLP), _points(ps_api, ctx, ps_api.G1, tr.XiT),
    )


def test_groth16_toy_proof_bit_identical(ps_api, ctx, co, pr):
    """Config #1 end to end: TestGroth16ProofGen's dlog identities (groth16_test.go:32-107) and
    byte equality with the literal nested-sumBlind restatement."""
    from oracle import restate as rs

    rng = pr.SplitMix64(SEED + 16)
    c, wit = rs.toy_circuit()
    tox = [rng.fr() for _ in range(5)]
    r, s = rng.fr(), rng.fr()
    tr = rs.groth16_setup(c, *tox)
    sol = [pr.fr(v) for v in wit]
    want = rs.groth16_prove(tr, c, sol, r, s)
    assert rs.groth16_dlog_check(tr, c, sol, want) == (True, True, True)
    q = _upload_circuit(ps_api, ctx, c)
    proof = ps_api.Groth16Prove(_groth16_pk(ps_api, ctx, tr), q, ps_api.Poly.from_values(ctx, wit), r, s)
    assert (proof.A, proof.B, proof.C) == (want.A, want.B, want.C)
    # literal reference form (per-variable polynomials, nested loops) from the Python twin
    ql = pr.to_qap(pr.create_r1cs())
    lit = pr.groth16_prove(pr.groth16_setup(ql, *tox), ql, wit, r, s)
    assert (proof.A, proof.B, proof.C) == (co.G1.to_b(lit.A), co.G2.to_b(lit.B), co.G1.to_b(lit.C))


@pytest.mark.parametrize("n", [7, 64, 300])
def test_groth16_synthetic_proof_bit_identical(ps_api, ctx, co, pr, n):
    from oracle import restate as rs

    rng = pr.SplitMix64(SEED + 160 + n)
    c, sol = rs.synthetic_circuit(n)
    tr = rs.groth16_setup(c, *[rng.fr() for _ in range(5)])
    r, s = rng.fr(), rng.fr()
    want = rs.groth16_prove(tr, c, sol, r, s, fast=True)
    assert rs.groth16_dlog_check(tr, c, sol, want) == (True, True, True)
    q = _upload_circuit(ps_api, ctx, c)
    proof = ps_api.Groth16Prove(_groth16_pk(ps_api, ctx, tr), q, ps_api.Poly.upload(ctx, sol), r, s)
    assert (proof.A, proof.B, proof.C) == (want.A, want.B, want.C)


def _phgr13_ek(ps_api, ctx, ek):
    kw = {}
    for f in ps_api.PHGR13EvalKey.FIELDS:
        kw[f] = _points(ps_api, ctx, ps_api.G2 if f == "ws" else ps_api.G1, getattr(ek, f))
    return ps_api.PHGR13EvalKey(**kw)


@pytest.mark.parametrize("n", [4, 50, 256])
def test_phgr13_proof_bit_identical(ps_api, ctx, co, pr, n):
    """PHGR13Prove is deterministic (no prover randomness): all 8 elements byte-equal.
    n = 4 is the toy circuit of TestPinocchioProofValidDivision (pinocchio_test.go:23-29)."""
    from oracle import restate as rs

    rng = pr.SplitMix64(SEED + 13 + n)
    if n == 4:
        c, wit = rs.toy_circuit()
        sol = [pr.fr(v) for v in wit]
        sol_dev = ps_api.Poly.from_values(ctx, wit)
    else:
        c, sol = rs.synthetic_circuit(n)
        sol_dev = ps_api.Poly.upload(ctx, sol)
    setup = rs.phgr13_setup(c, *[rng.fr() for _ in range(8)])
    want = rs.phgr13_prove(setup.EK, c, sol, fast=n > 16)
    q = _upload_circuit(ps_api, ctx, c)
    proof = ps_api.PHGR13Prove(_phgr13_ek(ps_api, ctx, setup.EK), q, sol_dev)
    for f in ps_api.PHGR13Proof.FIELDS:
        assert getattr(proof, f) == getattr(want, f), f
    # hs == h(s)*G  (pinocchio_test.go:33-44)
    assert proof.hs == co.G1.to_b(co.G1.mul(pr.poly_eval(want.h, setup.t.s)))


@pytest.mark.parametrize("n", [4, 37, 64, 65, 200, 1500])
def test_monomial_key_to_lagrange_form_without_toxic_waste(ps_api, ctx, co, pr, n):
    """ps_points_monomial_to_lagrange: the monomial arrays the reference's setups emit -- Xi, Xi2, XiT
    (groth16.go:79-97), gsi (pinochio.go:101) -- turned into their Lagrange form over the group elements alone (the toxic
    waste "must be delete[d]", groth16.go:13-14).  Byte-identical to the arrays the device setup computes FROM the toxic
    waste (themselves checked against the oracle in test_groth16_trusted_setup_on_device), for both node sets and both groups;
    a key converted this way proves the oracle's proof on the route without interpolation; wrong lengths are BlindEval's panic."""
    from oracle import restate as rs

    rng = pr.SplitMix64(SEED + 777 + n)
    if n == 4:
        c, wit = rs.toy_circuit()
        sol = [pr.fr(v) for v in wit]
    else:
        c, sol = rs.synthetic_circuit(n + (-n % 4))  # 40, 64, 68, 200, 1500 gates: n < np, n = np, a first tree level, several
    n = len(c.left)
    tox = [rng.fr() for _ in range(5)]
    q = _upload_circuit(ps_api, ctx, c)
    tr, _vk = ps_api.NewGroth16TrustedSetup(q, *tox)
    conv = tr.monomial_only().with_lagrange(q)
    assert conv.LXi.download() == tr.LXi.download()
    assert conv.LXi2.download() == tr.LXi2.download()
    assert conv.LXiT.download() == tr.LXiT.download()
    r, s = rng.fr(), rng.fr()
    dsol = ps_api.Poly.upload(ctx, sol)
    got = ps_api.Groth16Prove(conv, q, dsol, r, s)
    assert ctx.last_prove_phase_ms()["quotient"] >= 0
    if n <= 256:
        ref = rs.groth16_prove(rs.groth16_setup(c, *tox), c, sol, r, s, fast=n > 16)
        assert (got.A, got.B, got.C) == (ref.A, ref.B, ref.C)
    else:
        mono = ps_api.Groth16Prove(tr.monomial_only(), q, dsol, r, s)
        assert (got.A, got.B, got.C) == (mono.A, mono.B, mono.C)
    if n >= 4:
        tox8 = [rng.fr() for _ in range(8)]
        ek, _pvk = ps_api.NewPHGR13TrustedSetup(q, *tox8)
        conv_ek = ek.monomial_only().with_lagrange(q)
        assert conv_ek.lgsi.download() == ek.lgsi.download()
        a, b = ps_api.PHGR13Prove(conv_ek, q, dsol), ps_api.PHGR13Prove(ek.monomial_only(), q, dsol)
        for f in ps_api.PHGR13Proof.FIELDS:
            assert getattr(a, f) == getattr(b, f), f
    with pytest.raises(ps_api.LengthMismatch):
        tr.Xi.to_lagrange(q, 1)  # n points where the nodes n+1..2n-1 take n-1
    with pytest.raises(ps_api.LengthMismatch):
        tr.XiT.to_lagrange(q, 0)


def test_monomial_key_to_lagrange_form_at_2pow16(ps_api, ctx, co, pr):
    """The conversion at 2^16 constraints (VERDICT r3 item 6's spot check, here the whole arrays): Xi, Xi2 and XiT of a
    device-made key, converted without the toxic waste, are byte for byte the Lagrange-form arrays the setup computed from
    it; the converted key proves the bytes of the monomial key's proof."""
    import time

    from oracle import restate as rs

    n = 1 << 16
    c, sol = rs.synthetic_circuit(n)
    c.nbIO = c.nbVars - 3
    rng = pr.SplitMix64(SEED + 7777)
    q = _upload_circuit(ps_api, ctx, c)
    tr, _vk = ps_api.NewGroth16TrustedSetup(q, *[rng.fr() for _ in range(5)])
    t0 = time.time()
    conv = tr.monomial_only().with_lagrange(q)
    print("[perf] monomial -> Lagrange form of a 2^16-constraint Groth16 key (Xi, Xi2, XiT): %.1f s" % (time.time() - t0))
    assert conv.LXi.download() == tr.LXi.download()
    assert conv.LXi2.download() == tr.LXi2.download()
    assert conv.LXiT.download() == tr.LXiT.download()
    r, s = rng.fr(), rng.fr()
    dsol = ps_api.Poly.upload(ctx, sol)
    a, b = ps_api.Groth16Prove(conv, q, dsol, r, s), ps_api.Groth16Prove(tr.monomial_only(), q, dsol, r, s)
    assert (a.A, a.B, a.C) == (b.A, b.B, b.C)


@pytest.mark.parametrize("group", ["G1", "G2"])
def test_key_conversion_scalar_multiplication_edge_scalars(ps_api, ctx, co, pr, group):
    """The scalar multiplication inside ps_points_monomial_to_lagrange (csrc/lagrange.hpp, ec_mul_glv: k = k1 + k2 lambda with
    lambda P = (beta x, y), both halves in signed radix-16 digits over one table) against the oracle's Point.Mul, scalar by
    scalar, on the values a random twiddle never hits: 0, 1, the neighbours of lambda and of 2^128, r - 1, digit patterns of
    all -8 / all 7 / all 0, and the identity as the point (ps_debug_points_scale: a test hook, not in the header)."""
    import ctypes as C

    lam = pr.BLS_Z * pr.BLS_Z - 1
    assert (lam * lam + lam + 1) % pr.R == 0
    R = pr.R
    rng = pr.SplitMix64(SEED + 4242)
    nib = lambda d: sum(d << (4 * i) for i in range(32))  # the same nibble 32 times
    edge = [0, 1, 2, 7, 8, 9, 15, 16, lam - 1, lam, lam + 1, 2 * lam, lam * lam % R, R - 1, R - 2, R - lam, (1 << 128) - 1, 1 << 128,
            (1 << 128) + 1, (1 << 255) % R, nib(0x8), nib(0x7), nib(0xF), nib(0x8) * (lam + 1) % R, (nib(0x7) + nib(0x7) * lam) % R,
            (lam - 1) + (lam - 1) * lam, 8 * lam + 8, (R - 1) // 2]
    ks = [k % R for k in edge] + [rng.fr() for _ in range(36)]
    G = getattr(co, group)
    gid = ps_api.G1 if group == "G1" else ps_api.G2
    base = [G.mul(rng.fr()) for _ in range(4)]
    pts = [base[i % 4] for i in range(len(ks))]
    ident = G.to_b(None)
    raw = G.pack(pts) + ident  # ... and the identity as the last point
    ks.append(rng.fr())
    dpts = ps_api.Points.upload(ctx, gid, raw)
    dks = ps_api.Poly.upload(ctx, ks)
    lib = ps_api.lib
    lib.ps_debug_points_scale.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.ps_debug_points_scale.restype = C.c_int
    h = C.c_void_p()
    ps_api._check(lib.ps_debug_points_scale(ctx._h, dpts._h, dks._h, C.byref(h)))
    got = ps_api.Points(ctx, h).download()
    for i, k in enumerate(ks[:-1]):
        want = G.to_b(G.mul(k, pts[i])) if k else ident
        assert got[i * G.nb:(i + 1) * G.nb] == want, (group, i, hex(k))
    assert got[-G.nb:] == ident  # k * O = O


@pytest.mark.parametrize("n", [4, 37, 200])
def test_groth16_trusted_setup_on_device(ps_api, ctx, co, pr, n):
    """NewGroth16TrustedSetup (groth16.go:64-101) on the device against the oracle: every CRS array
    byte-identical (Xi, Xi2, IoLP, NioLP, XiT and the six fixed points), then a proof made with the
    device CRS equals the oracle's proof."""
    from oracle import restate as rs

    rng = pr.SplitMix64(SEED + 64 + n)
    if n == 4:
        c, wit = rs.toy_circuit()
        sol = [pr.fr(v) for v in wit]
    else:
        c, sol = rs.synthetic_circuit(n)
    tox = [rng.fr() for _ in range(5)]
    want = rs.groth16_setup(c, *tox)
    q = _upload_circuit(ps_api, ctx, c)
    tr, vk = ps_api.NewGroth16TrustedSetup(q, *tox)
    assert (tr.Alpha, tr.Beta, tr.Delta, tr.Beta2, tr.Delta2, vk["Gamma"]) == (
        want.Alpha, want.Beta, want.Delta, want.Beta2, want.Delta2, want.Gamma)
    assert tr.Xi.download() == want.Xi
    assert tr.Xi2.download() == want.Xi2
    assert tr.XiT.download() == want.XiT
    assert tr.NioLP.download() == want.NioLP
    assert vk["IoLP"].download() == want.IoLP
    # the same CRS in Lagrange form: l_j(x) G on {1..n} in both groups, lambda_k(x) t(x)/delta G on the nodes n+1..2n-1
    x, delta = tox[3], tox[2]
    lj, zx = rs.lagrange_at(n, x)
    lam, _ = rs.lagrange_at(n - 1, (x - n) % pr.R)  # equally spaced nodes: the basis of {n+1..2n-1} at x is that of {1..n-1} at x-n
    assert tr.LXi.download() == b"".join(co.G1.to_b(co.G1.mul(v)) for v in lj)
    assert tr.LXi2.download() == b"".join(co.G2.to_b(co.G2.mul(v)) for v in lj)
    assert tr.LXiT.download() == b"".join(co.G1.to_b(co.G1.mul(v * pr.fr_div(zx, delta) % pr.R)) for v in lam)
    r, s = rng.fr(), rng.fr()
    dsol = ps_api.Poly.upload(ctx, sol)
    proof = ps_api.Groth16Prove(tr, q, dsol, r, s)  # the Lagrange route: no interpolation, no division
    ref = rs.groth16_prove(want, c, sol, r, s, fast=n > 16)
    assert (proof.A, proof.B, proof.C) == (ref.A, ref.B, ref.C)
    mono = ps_api.Groth16Prove(tr.monomial_only(), q, dsol, r, s)  # the key as the reference makes it: coefficients
    assert (mono.A, mono.B, mono.C) == (ref.A, ref.B, ref.C)
    if n > 4:
        bad = list(sol)
        bad[4] = (bad[4] + 1) % pr.R
        for key in (tr, tr.monomial_only()):
            with pytest.raises(ps_api.Apocalypse):
                ps_api.Groth16Prove(key, q, ps_api.Poly.upload(ctx, bad), r, s)
    # x on the interpolation domain (or on the nodes n+1..2n-1 of the Lagrange form of XiT) is refused: a denominator would vanish
    with pytest.raises(ps_api.PlaysnarkError):
        ps_api.NewGroth16TrustedSetup(q, tox[0], tox[1], tox[2], 3, tox[4])
    with pytest.raises(ps_api.PlaysnarkError):
        ps_api.NewGroth16TrustedSetup(q, tox[0], tox[1], tox[2], n + 1, tox[4])


@pytest.mark.parametrize("n", [4, 37, 200])
def test_phgr13_trusted_setup_on_device(ps_api, ctx, co, pr, n):
    """NewPHGR13TrustedSetup (pinochio.go:93-176) on the device against the oracle: the ten
    evaluation-key arrays, the seven fixed verification-key points and vk.vs / vk.ws / vk.ys over all
    variables are byte-identical; a proof made with the device EK equals the oracle's proof and
    PHGR13Verify (pinochio.go:281-378) accepts it with the device VK."""
    from oracle import restate as rs

    rng = pr.SplitMix64(SEED + 93 + n)
    if n == 4:
        c, wit = rs.toy_circuit()
        sol = [pr.fr(v) for v in wit]
    else:
        c, sol = rs.synthetic_circuit(n)
    diff = c.nbVars - c.nbIO
    tox = [rng.fr() for _ in range(8)]
    want = rs.phgr13_setup(c, *tox)
    q = _upload_circuit(ps_api, ctx, c)
    ek, vk = ps_api.NewPHGR13TrustedSetup(q, *tox)
    for f in ps_api.PHGR13EvalKey.FIELDS:
        assert getattr(ek, f).download() == getattr(want.EK, f), f
    G1, G2 = co.G1, co.G2
    groups = {"av": G2, "aw": G1, "ay": G2, "gamma": G2, "bgamma": G1, "bgamma2": G2, "yts": G2}
    for f, grp in groups.items():
        assert getattr(vk, f) == grp.to_b(getattr(want.VK, f)), f
    assert vk.vs.download() == G1.pack(want.VK.vs)
    assert vk.ws.download() == G2.pack(want.VK.ws)
    assert vk.ys.download() == G1.pack(want.VK.ys)
    lam, _ = rs.lagrange_at(n - 1, (tox[0] - n) % pr.R)  # gsi in Lagrange form on the nodes n+1..2n-1
    assert ek.lgsi.download() == b"".join(G1.to_b(G1.mul(v)) for v in lam)
    sol_dev = ps_api.Poly.upload(ctx, sol)
    proof = ps_api.PHGR13Prove(ek, q, sol_dev)  # h by its values over lgsi
    ref = rs.phgr13_prove(want.EK, c, sol, fast=n > 16)
    mono = ps_api.PHGR13Prove(ek.monomial_only(), q, sol_dev)  # h by its coefficients over gsi, as the reference's key allows
    for f in ps_api.PHGR13Proof.FIELDS:
        assert getattr(proof, f) == getattr(ref, f), f
        assert getattr(mono, f) == getattr(ref, f), f
    io = ps_api.Poly.upload(ctx, sol[:diff])
    args = (vk.vs.slice(0, diff), vk.ws.slice(0, diff), vk.ys.slice(0, diff))
    assert ps_api.PHGR13Verify(ctx, vk.fixed_points(), *args, proof, io)
    bad_io = list(sol[:diff])
    bad_io[0] = (bad_io[0] + 1) % pr.R
    assert not ps_api.PHGR13Verify(ctx, vk.fixed_points(), *args, proof, ps_api.Poly.upload(ctx, bad_io))
    with pytest.raises(ps_api.PlaysnarkError):  # s on the interpolation domain
        ps_api.NewPHGR13TrustedSetup(q, 2, *tox[1:])


def test_phgr13_arrays_of_unequal_length(ps_api, ctx, co, pr):
    """computeSolCommit (pinochio.go:222-229) ranges over len(evalCommit): an evaluation key whose
    arrays differ in length (not what the setup produces) takes the one-sum-at-a-time path and still
    matches the oracle element by element."""
    from oracle import restate as rs

    rng = pr.SplitMix64(SEED + 777)
    c, sol = rs.synthetic_circuit(40)
    c = rs.SparseR1CS(c.nbVars, c.nbVars - 3, c.left, c.right, c.out)  # 3 public, the rest in the sums
    setup = rs.phgr13_setup(c, *[rng.fr() for _ in range(8)])
    ek = setup.EK
    ek.vs = ek.vs[:-96]      # one G1 point shorter
    ek.ws = ek.ws[:-2 * 192]  # two G2 points shorter
    want = rs.phgr13_prove(ek, c, sol, fast=True)
    q = _upload_circuit(ps_api, ctx, c)
    proof = ps_api.PHGR13Prove(_phgr13_ek(ps_api, ctx, ek), q, ps_api.Poly.upload(ctx, sol))
    for f in ps_api.PHGR13Proof.FIELDS:
        assert getattr(proof, f) == getattr(want, f), f


def test_groth16_int64_witness_with_negative_values(ps_api, ctx, co, pr):
    """The reference's Vector is []int (algebra.go:13): a witness uploaded as int64 -- negative values
    included -- keeps the short-scalar plan; Groth16Prove then sums its NioLP part separately and the
    three proof elements are still byte-identical to the oracle (which works with the values mod r)."""
    from oracle import restate as rs

    rng = pr.SplitMix64(SEED + 4321)
    c, sol = rs.synthetic_circuit(8, x0=pr.R - 3)  # x = -3: every wire is a small signed integer
    wit = [v if v < pr.R // 2 else v - pr.R for v in sol]
    assert min(wit) < 0 and max(abs(v) for v in wit) < 1 << 62
    for nb_io in (3, c.nbVars - 3):  # both sides of the reference's `diff` split carry witness values
        cc = rs.SparseR1CS(c.nbVars, nb_io, c.left, c.right, c.out)
        tox = [rng.fr() for _ in range(5)]
        r, s = rng.fr(), rng.fr()
        want = rs.groth16_prove(rs.groth16_setup(cc, *tox), cc, sol, r, s)
        q = _upload_circuit(ps_api, ctx, cc)
        tr, _ = ps_api.NewGroth16TrustedSetup(q, *tox)
        proof = ps_api.Groth16Prove(tr, q, ps_api.Poly.from_values(ctx, wit), r, s)
        assert (proof.A, proof.B, proof.C) == (want.A, want.B, want.C)
        # PHGR13 on the same witness: all seven sums take the short plan
        ptox = [rng.fr() for _ in range(8)]
        pwant = rs.phgr13_prove(rs.phgr13_setup(cc, *ptox).EK, cc, sol)
        ek, _ = ps_api.NewPHGR13TrustedSetup(q, *ptox)
        pp = ps_api.PHGR13Prove(ek, q, ps_api.Poly.from_values(ctx, wit))
        for f in ps_api.PHGR13Proof.FIELDS:
            assert getattr(pp, f) == getattr(pwant, f), f


@pytest.mark.parametrize("world", [1, 3, 8])
def test_sharded_phgr13_partials_fold_to_the_proof(ps_api, ctx, co, pr, world):
    """BASELINE config #5 shape on one GPU: the index-range shares that `world` ranks would compute
    (playsnark_amd.dist.ShardedPHGR13.partials), folded, are the unsharded proof byte for byte -- also
    when some ranks get an empty range (8 ranks, 5-element arrays)."""
    from oracle import restate as rs
    from playsnark_amd.dist import ShardedPHGR13

    rng = pr.SplitMix64(SEED + 555 + world)
    c, sol = rs.synthetic_circuit(41)
    setups = [(c, "few"), (rs.SparseR1CS(c.nbVars, c.nbVars - 3, c.left, c.right, c.out), "many")]
    if world == 8:
        setups.append((rs.SparseR1CS(c.nbVars, 5, c.left, c.right, c.out), "five"))
    for cc, _ in setups:
        q = _upload_circuit(ps_api, ctx, cc)
        ek, _vk = ps_api.NewPHGR13TrustedSetup(q, *[rng.fr() for _ in range(8)])
        sol_dev = ps_api.Poly.upload(ctx, sol)
        whole = ps_api.PHGR13Prove(ek, q, sol_dev)
        sh = ShardedPHGR13(ctx, None, world, 0)
        folded = sh.fold([sh.partials(ek, q, sol_dev, rank=g) for g in range(world)])
        for f in ps_api.PHGR13Proof.FIELDS:
            assert getattr(folded, f) == getattr(whole, f), f


def test_sharded_phgr13_refuses_keys_the_unsharded_prover_panics_on(ps_api, ctx, co, pr):
    """A gsi longer than h, or evaluation-key arrays of different lengths, must not be silently truncated by the
    index-range slices: BlindEval panics on them (algebra.go:350-352) and so does the sharded prover."""
    from oracle import restate as rs
    from playsnark_amd.dist import ShardedPHGR13

    rng = pr.SplitMix64(SEED + 556)
    c, sol = rs.synthetic_circuit(20)
    c = rs.SparseR1CS(c.nbVars, c.nbVars - 3, c.left, c.right, c.out)
    q = _upload_circuit(ps_api, ctx, c)
    ek, _vk = ps_api.NewPHGR13TrustedSetup(q, *[rng.fr() for _ in range(8)])
    sol_dev = ps_api.Poly.upload(ctx, sol)
    sh = ShardedPHGR13(ctx, None, 2, 0)
    sh.partials(ek, q, sol_dev)  # the key as the setup made it is fine
    good_gsi, good_vs = ek.gsi, ek.vs
    ek.gsi = ps_api.Points.upload(ctx, ps_api.G1, good_gsi.download() + co.G1.to_b(co.G1.mul(5)))  # one point too many
    with pytest.raises(ps_api.LengthMismatch):
        sh.partials(ek, q, sol_dev)
    with pytest.raises(ps_api.LengthMismatch):
        ps_api.PHGR13Prove(ek, q, sol_dev)
    ek.gsi = good_gsi
    ek.vs = good_vs.slice(0, len(good_vs) - 1)
    with pytest.raises(ps_api.LengthMismatch):
        sh.partials(ek, q, sol_dev)
    ek.vs = good_vs


@pytest.mark.parametrize("world", [1, 3, 8])
def test_sharded_groth16_partials_fold_to_the_proof(ps_api, ctx, co, pr, world):
    """The shares that `world` ranks compute with ps_groth16_prove_shard (index ranges of Xi, Xi2, NioLP,
    XiT; fixed points on rank 0), folded, are the unsharded proof byte for byte -- for both positions of
    the reference's `diff` split and with more ranks than some arrays have elements."""
    from oracle import restate as rs
    from playsnark_amd.dist import ShardedGroth16

    rng = pr.SplitMix64(SEED + 808 + world)
    c, sol = rs.synthetic_circuit(45)
    for cc in (c, rs.SparseR1CS(c.nbVars, c.nbVars - 3, c.left, c.right, c.out)):
        q = _upload_circuit(ps_api, ctx, cc)
        tr, _ = ps_api.NewGroth16TrustedSetup(q, *[rng.fr() for _ in range(5)])
        sol_dev = ps_api.Poly.upload(ctx, sol)
        r, s = rng.fr(), rng.fr()
        whole = ps_api.Groth16Prove(tr, q, sol_dev, r, s)
        sh = ShardedGroth16(ctx, None, world, 0)
        folded = sh.fold([sh.partials(tr, q, sol_dev, r, s, rank=g) for g in range(world)], r, s)
        assert (folded.A, folded.B, folded.C) == (whole.A, whole.B, whole.C)
    with pytest.raises(ps_api.PlaysnarkError):
        ShardedGroth16(ctx, None, 2, 0).partials(tr, q, sol_dev, r, s, rank=2)


def _local_groth16_key(ps_api, ctx, tr, n, nn, rank, world):
    """The d-th index ranges of an oracle-made Groth16 setup, uploaded on their own (a rank-local CRS)."""
    from playsnark_amd.dist import shard_range

    def part(raw, total, nb, group):
        first, cnt = shard_range(total, rank, world)
        return ps_api.Points.upload(ctx, group, raw[first * nb:(first + cnt) * nb])

    return ps_api.Groth16Setup(tr.Alpha, tr.Beta, tr.Delta, tr.Beta2, tr.Delta2, part(tr.Xi, n, 96, ps_api.G1),
                               part(tr.Xi2, n, 192, ps_api.G2), part(tr.NioLP, nn, 96, ps_api.G1), part(tr.XiT, n - 1, 96, ps_api.G1))


@pytest.mark.parametrize("ndev", [1, 2, 3, 5])
def test_groth16_prove_over_rank_local_keys(ps_api, ctx, co, pr, ndev):
    """Rank-local CRS (every rank / device holds only its index ranges): (i) ps_groth16_prove_multi, the in-process
    multi-device entry -- here `ndev` contexts on the one GPU, each with its own QAP, solution and local arrays; with three or
    more the quotient's parts A, B, h come from different contexts --; (ii) ShardedGroth16Local, the one-process-per-GPU form,
    with simulated ranks.  Both give the bytes of the oracle's (and the unsharded) proof; wrong ranges are refused."""
    from oracle import restate as rs
    from playsnark_amd.dist import ShardedGroth16, ShardedGroth16Local

    rng = pr.SplitMix64(SEED + 6000 + ndev)
    c, sol = rs.synthetic_circuit(53)
    c = rs.SparseR1CS(c.nbVars, c.nbVars - 3, c.left, c.right, c.out)
    tox = [rng.fr() for _ in range(5)]
    r, s = rng.fr(), rng.fr()
    tr = rs.groth16_setup(c, *tox)
    want = rs.groth16_prove(tr, c, sol, r, s)
    n, nn = c.nbGates, c.nbIO
    ctxs = [ctx] + [ps_api.Context(0) for _ in range(ndev - 1)]
    devices = []
    for d, cx in enumerate(ctxs):
        q = _upload_circuit(ps_api, cx, c)
        devices.append((_local_groth16_key(ps_api, cx, tr, n, nn, d, ndev), q, ps_api.Poly.upload(cx, sol)))
    proof = ps_api.Groth16ProveMulti(devices, r, s)
    assert (proof.A, proof.B, proof.C) == (want.A, want.B, want.C)
    # the same through the per-rank class, ranks simulated one after the other on their own contexts
    parts = [ShardedGroth16Local(cx, None, ndev, d).partials(devices[d][0], devices[d][1], devices[d][2], r, s) for d, cx in enumerate(ctxs)]
    folded = ShardedGroth16.fold(parts, r, s)
    assert (folded.A, folded.B, folded.C) == (want.A, want.B, want.C)
    if ndev > 1:  # a rank that holds someone else's range is told so (BlindEval would panic on the lengths, algebra.go:350-352)
        wrong = [devices[0]] + [(devices[0][0], devices[d][1], devices[d][2]) for d in range(1, ndev)]
        with pytest.raises(ps_api.LengthMismatch):
            ps_api.Groth16ProveMulti(wrong, r, s)
        with pytest.raises(ps_api.LengthMismatch):
            ShardedGroth16Local(ctxs[1], None, ndev, 1).partials(_local_groth16_key(ps_api, ctxs[1], tr, n, nn, 1, ndev + 1),
                                                                  devices[1][1], devices[1][2], r, s)
    bad = list(sol)
    bad[5] = (bad[5] + 1) % pr.R
    with pytest.raises(ps_api.Apocalypse):
        ps_api.Groth16ProveMulti([(k, q, ps_api.Poly.upload(q.ctx, bad)) for k, q, _ in devices], r, s)
    for cx in ctxs[1:]:
        cx.close()


def test_msm_over_shards_on_several_contexts(ps_api, ctx, co, pr):
    """ps_msm_multi_device: index-range shards on different contexts (devices) of one process, summed side by side."""
    from playsnark_amd.dist import shard_range

    rng = pr.SplitMix64(SEED + 6100)
    n, ndev = 1000, 3
    sc = [rng.fr() for _ in range(n)]
    ctxs = [ctx, ps_api.Context(0), ps_api.Context(0)]
    for gid, og in ((ps_api.G1, co.G1), (ps_api.G2, co.G2)):
        raw = og.gen_points(rng.fr(), rng.fr(), n)
        shards, scs = [], []
        for d, cx in enumerate(ctxs):
            first, cnt = shard_range(n, d, ndev)
            shards.append(ps_api.Points.upload(cx, gid, raw[first * og.nb:(first + cnt) * og.nb]))
            scs.append(ps_api.Poly.upload(cx, sc[first:first + cnt]))
        assert ps_api.msm_multi_device(ctxs, shards, scs) == og.to_b(og.msm_pippenger(co.pack_fr(sc), raw, n, 4))
    assert ps_api.points_lincomb(ps_api.G1, co.G1.to_b(co.G1.mul(5)) + co.G1.to_b(co.G1.mul(7)), [3, pr.R - 2]) == co.G1.to_b(co.G1.mul(1))
    for cx in ctxs[1:]:
        cx.close()


def test_sharded_phgr13_over_rank_local_keys(ps_api, ctx, co, pr):
    """ShardedPHGR13(local=True): the evaluation key of a rank holds only its index ranges."""
    from oracle import restate as rs
    from playsnark_amd.dist import ShardedPHGR13, shard_range

    rng = pr.SplitMix64(SEED + 6200)
    c, sol = rs.synthetic_circuit(41)
    c = rs.SparseR1CS(c.nbVars, c.nbVars - 3, c.left, c.right, c.out)
    setup = rs.phgr13_setup(c, *[rng.fr() for _ in range(8)])
    want = rs.phgr13_prove(setup.EK, c, sol, fast=True)
    q = _upload_circuit(ps_api, ctx, c)
    sol_dev = ps_api.Poly.upload(ctx, sol)
    world = 3
    parts = []
    for g in range(world):
        fields = {}
        for f in ps_api.PHGR13EvalKey.FIELDS:
            raw, nb = getattr(setup.EK, f), (192 if f == "ws" else 96)
            first, cnt = shard_range(len(raw) // nb, g, world)
            fields[f] = ps_api.Points.upload(ctx, ps_api.G2 if f == "ws" else ps_api.G1, raw[first * nb:(first + cnt) * nb])
        sh = ShardedPHGR13(ctx, None, world, g, local=True)
        parts.append(sh.partials(ps_api.PHGR13EvalKey(**fields), q, sol_dev))
        if g == 1:
            with pytest.raises(ps_api.LengthMismatch):
                ShardedPHGR13(ctx, None, world + 1, g, local=True).partials(ps_api.PHGR13EvalKey(**fields), q, sol_dev)
    folded = ShardedPHGR13.fold(parts)
    for f in ps_api.PHGR13Proof.FIELDS:
        assert getattr(folded, f) == getattr(want, f), f


def test_aggregate_polynomials_and_h_byte_exact_at_2pow16(ps_api, ctx, co, pr):
    """A, B, C and h at n = 2^16 -- far beyond the literal O(n^3) restatement -- byte for byte against the oracle's fast CPU
    algorithm (or_fast_quotient, itself pinned to the literal one at small n), both routes of the GPU quotient."""
    from oracle import restate as rs

    n = 1 << 16
    c, sol = rs.synthetic_circuit(n)
    dot = lambda rows: [sum(v * sol[j] for j, v in row) % pr.R for row in rows]
    want = co.fast_quotient_bytes(co.pack_fr(dot(c.left)), co.pack_fr(dot(c.right)), co.pack_fr(dot(c.out)), n)
    q = _upload_circuit(ps_api, ctx, c)
    dsol = ps_api.Poly.upload(ctx, sol)
    got = tuple(p.download_bytes() for p in q.computeAggregatePoly(dsol))
    for name, g, w in zip("ABCh", got, want):
        assert g == w, name
    assert q.Quotient(dsol).download_bytes() == want[3]      # the h-only route
    assert q.interpolate(dsol, 1).download_bytes() == want[1]  # one polynomial alone


def test_qap_is_valid_is_the_reference_divisibility_test(ps_api, ctx, pr):
    """(*QAP).IsValid (qap.go:107-148, TestQAPValidity qap_test.go): true for the toy witness and the tiled circuit's
    solution, false once any wire value is off by one; a wrong number of variables is sanityCheck's panic."""
    from oracle import restate as rs

    c, wit = rs.toy_circuit()
    q = _upload_circuit(ps_api, ctx, c)
    assert q.IsValid(ps_api.Poly.from_values(ctx, wit)) is True
    for i in range(1, len(wit)):
        bad = list(wit)
        bad[i] += 1
        assert q.IsValid(ps_api.Poly.from_values(ctx, bad)) is False, i
    c, sol = rs.synthetic_circuit(300)
    q = _upload_circuit(ps_api, ctx, c)
    assert q.IsValid(ps_api.Poly.upload(ctx, sol)) is True
    sol[17] = (sol[17] + 1) % pr.R
    assert q.IsValid(ps_api.Poly.upload(ctx, sol)) is False
    with pytest.raises(ps_api.PlaysnarkError):
        q.IsValid(ps_api.Poly.upload(ctx, sol[:-1]))


def test_groth16_split_form_of_c_gives_the_oracle_proof(ps_api, co, pr, monkeypatch):
    """With a Lagrange-form key and >= 2^19 constraints Groth16Prove sums B in G1 on its own (over Xi with the values b_j,
    which are wire values: small in most of a real circuit) and adds s A + r B1 to C on the host, instead of one sum with
    the full-width scalars s a_j + r b_j (prove.inc, groth16_prove_impl; groth16.go:180-205 is the statement restated).
    The environment moves the threshold so that the form runs at a size the oracle proves in seconds: the same bytes for
    the whole proof, for an int64 witness (NioLP as a sum of its own) and for the shares of 3 ranks folded."""
    from oracle import restate as rs
    from playsnark_amd.dist import ShardedGroth16

    rng = pr.SplitMix64(SEED + 1919)
    for kind in ("synthetic", "bits"):
        c, sol = rs.synthetic_circuit(300) if kind == "synthetic" else rs.bit_circuit(257, seed=5)
        c = rs.SparseR1CS(c.nbVars, c.nbVars - (3 if kind == "synthetic" else 1), c.left, c.right, c.out)
        tw = [rng.fr() for _ in range(5)]
        r, s = rng.fr(), rng.fr()
        want = rs.groth16_prove(rs.groth16_setup(c, *tw), c, sol, r, s, fast=True)
        for min_n in ("2", "1000000000"):
            monkeypatch.setenv("PS_G16_B1_MIN_N", min_n)
            cx = ps_api.Context(0)
            q = _upload_circuit(ps_api, cx, c)
            tr, _ = ps_api.NewGroth16TrustedSetup(q, *tw)
            for dsol in [ps_api.Poly.upload(cx, sol)] + ([ps_api.Poly.from_values(cx, sol)] if kind == "bits" else []):
                for key in (tr, tr.monomial_only()):
                    got = ps_api.Groth16Prove(key, q, dsol, r, s)
                    assert (got.A, got.B, got.C) == (want.A, want.B, want.C), (kind, min_n)
            sh = ShardedGroth16(cx, None, 3, 0)
            dsol = ps_api.Poly.upload(cx, sol)
            folded = sh.fold([sh.partials(tr, q, dsol, r, s, rank=g) for g in range(3)], r, s)
            assert (folded.A, folded.B, folded.C) == (want.A, want.B, want.C), (kind, min_n, "shares")
            cx.close()


def test_one_key_two_contexts_two_threads_same_proof(ps_api, co, pr):
    """A proving key uploaded once and used from two host threads, each with its own context and QAP, as the reference's
    pure Groth16Prove / PHGR13Prove allow (groth16.go:122, pinochio.go:207): the window tables the provers attach to the
    key's arrays on first use are built once, under the array's lock (prove.inc mutated the caller's key
    through const, unsynchronised).  Both threads race into the first proof; every proof must be the oracle's."""
    import threading

    from oracle import restate as rs

    n = 2048  # keys of >= 1024 points get tables
    c, sol = rs.synthetic_circuit(n)
    c.nbIO = c.nbVars - 3  # the reference's split diff = nbVars - nbIO after (const, x, out): ~n witness values in the prover's sums
    rng = pr.SplitMix64(SEED + 909)
    tw = [rng.fr() for _ in range(5)]
    tw8 = [rng.fr() for _ in range(8)]
    r, s = rng.fr(), rng.fr()
    ctx0 = ps_api.Context(0)
    q0 = _upload_circuit(ps_api, ctx0, c)
    tr, _vk = ps_api.NewGroth16TrustedSetup(q0, *tw)
    ek, _pvk = ps_api.NewPHGR13TrustedSetup(q0, *tw8)
    tr_mono = tr.monomial_only()
    ctx0.sync()
    want = rs.groth16_prove(rs.groth16_setup(c, *tw), c, sol, r, s, fast=True)
    want_p = rs.phgr13_prove(rs.phgr13_setup(c, *tw8).EK, c, sol, fast=True)
    errors, barrier = [], threading.Barrier(2)

    def worker(tid):
        try:
            cx = ps_api.Context(0)
            q = _upload_circuit(ps_api, cx, c)
            dsol = ps_api.Poly.upload(cx, sol)
            barrier.wait()
            for rep in range(4):
                for key in (tr, tr_mono):
                    got = ps_api.Groth16Prove(key, q, dsol, r, s)
                    assert (got.A, got.B, got.C) == (want.A, want.B, want.C), ("groth16", tid, rep)
                gp = ps_api.PHGR13Prove(ek, q, dsol)
                for f in ps_api.PHGR13Proof.FIELDS:
                    assert getattr(gp, f) == getattr(want_p, f), ("phgr13", tid, rep, f)
            cx.close()
        except BaseException as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((tid, repr(e)))
            try:
                barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert ek.vs.table_window > 0  # the shared evaluation key did get its tables, once
    ctx0.close()


def test_tables_that_do_not_fit_fall_back_to_the_plain_plan(ps_api, co, pr):
    """Memory policy of the provers' window tables (VERDICT r2 / ADVICE): with a budget too small for a table (here forced
    through ps_ctx_set_table_budget; by default what hipMemGetInfo leaves) the first proof does not fail with PS_ERR_HIP --
    the sums run the plain plan (ps_msm_info.window_table == 0), the arrays are not asked again, and the proof bytes are
    those of the table plan."""
    from oracle import restate as rs

    n = 2048
    c, sol = rs.synthetic_circuit(n)
    c.nbIO = c.nbVars - 3
    rng = pr.SplitMix64(SEED + 910)
    tw, tw8 = [rng.fr() for _ in range(5)], [rng.fr() for _ in range(8)]
    r, s = rng.fr(), rng.fr()
    want = rs.groth16_prove(rs.groth16_setup(c, *tw), c, sol, r, s, fast=True)
    want_p = rs.phgr13_prove(rs.phgr13_setup(c, *tw8).EK, c, sol, fast=True)
    for budget, expect_table in ((4096, 0), (-1, 1)):
        cx = ps_api.Context(0)
        cx.set_table_budget(budget)
        q = _upload_circuit(ps_api, cx, c)
        dsol = ps_api.Poly.upload(cx, sol)
        tr, _vk = ps_api.NewGroth16TrustedSetup(q, *tw)
        ek, _pvk = ps_api.NewPHGR13TrustedSetup(q, *tw8)
        for _ in range(2):
            got = ps_api.Groth16Prove(tr, q, dsol, r, s)
            assert (got.A, got.B, got.C) == (want.A, want.B, want.C)
            assert cx.last_msm_info()["window_table"] == expect_table
            gp = ps_api.PHGR13Prove(ek, q, dsol)
            for f in ps_api.PHGR13Proof.FIELDS:
                assert getattr(gp, f) == getattr(want_p, f), f
            assert cx.last_msm_info()["window_table"] == expect_table
        assert (ek.vs.table_window > 0) == bool(expect_table)
        if not expect_table:
            # The mark a context with a small budget leaves on a SHARED array is not for ever: it records what that
            # context had to offer, and a context with more -- here the automatic budget -- asks again and gets its tables.
            cx2 = ps_api.Context(0)
            q2 = _upload_circuit(ps_api, cx2, c)
            gp2 = ps_api.PHGR13Prove(ek, q2, ps_api.Poly.upload(cx2, sol))
            for f in ps_api.PHGR13Proof.FIELDS:
                assert getattr(gp2, f) == getattr(want_p, f), f
            assert cx2.last_msm_info()["window_table"] == 1 and ek.vs.table_window > 0
            cx2.close()
            for arr in (ek.vs, ek.ws, ek.ys, ek.vas, ek.was, ek.yas, ek.gsi, ek.vbs):
                arr.drop_table()
        # an explicit request is still honoured (and still an error when it cannot be)
        if not expect_table:
            ek.vs.precompute(-1)   # clears the "declined" mark
            ek.vs.precompute(0)
            assert ek.vs.table_window > 0
        cx.close()
