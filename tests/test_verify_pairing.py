"""Groth16Verify / PHGR13Verify (groth16.go:214-233, pinochio.go:281-378) on a pure-Python pairing:
TestGroth16Verify (groth16_test.go:22-30), TestPinocchioProofValidDivision's final Verify
(pinocchio_test.go:195) and TestPinocchioInvalidProof (pinocchio_test.go:198-278), first for the
oracle's proofs on the CPU, then (gpu-marked) for proofs made by the HIP path -- the latter is a
check of the GPU result by mathematics alone."""
import pytest

SEED = 0x706C6179736E61726B & 0xFFFFFFFFFFFFFFFF


def _groth16_material(pr, co, rs, circuit="toy"):
    rng = pr.SplitMix64(SEED + 404)
    c, sol = (rs.toy_circuit() if circuit == "toy" else rs.synthetic_circuit(circuit))
    if circuit == "toy":
        wit, sol = sol, [pr.fr(v) for v in sol]
    else:
        wit = None
    tr = rs.groth16_setup(c, *[rng.fr() for _ in range(5)])
    r, s = rng.fr(), rng.fr()
    diff = c.nbVars - c.nbIO
    trp = rs.Bag(Alpha=co.G1.from_b(tr.Alpha), Beta2=co.G2.from_b(tr.Beta2), IoLP=co.G1.unpack(tr.IoLP),
                 Gamma=co.G2.from_b(tr.Gamma), Delta2=co.G2.from_b(tr.Delta2))
    return c, wit, sol, tr, trp, r, s, diff


def test_pairing_is_bilinear_and_nondegenerate(pr):
    from oracle import pairing as pg

    e = pg.pair(pr.G1.gen, pr.G2.gen)
    assert e != pg._ONE and pg.f12_pow(e, pr.R) == pg._ONE
    assert pg.pair(pr.G1.mul(6), pr.G2.mul(35)) == pg.f12_pow(e, 210)
    assert pg.pair(None, pr.G2.gen) == pg._ONE


def test_TestGroth16Verify_on_oracle_proof(co, pr):
    from oracle import pairing as pg
    from oracle import restate as rs

    c, wit, sol, tr, trp, r, s, diff = _groth16_material(pr, co, rs)
    pf = rs.groth16_prove(tr, c, sol, r, s)
    A, B, Cc = co.G1.from_b(pf.A), co.G2.from_b(pf.B), co.G1.from_b(pf.C)
    assert pg.groth16_verify(trp, A, B, Cc, sol[:diff])
    assert not pg.groth16_verify(trp, A, B, pr.G1.add(Cc, pr.G1.gen), sol[:diff])
    assert not pg.groth16_verify(trp, A, B, Cc, [sol[0], (sol[1] + 1) % pr.R, sol[2]])


def test_TestPinocchio_verify_and_invalid_proofs_on_oracle_proof(co, pr):
    from oracle import pairing as pg
    from oracle import restate as rs

    rng = pr.SplitMix64(SEED + 505)
    c, wit = rs.toy_circuit()
    sol = [pr.fr(v) for v in wit]
    st = rs.phgr13_setup(c, *[rng.fr() for _ in range(8)])
    pp = rs.phgr13_prove(st.EK, c, sol)
    diff = c.nbVars - c.nbIO
    tup = lambda: rs.Bag(**{f: (co.G2 if f == "wss" else co.G1).from_b(getattr(pp, f))
                            for f in ("vss", "vass", "wss", "wass", "yss", "yass", "hs", "gz")})
    assert pg.phgr13_verify(st.VK, diff, tup(), sol[:diff])
    rnd = pr.G1.mul(rng.fr())
    for field in ("yss", "vss", "hs", "gz"):  # pinocchio_test.go:243-262
        bad = tup()
        setattr(bad, field, rnd)
        assert not pg.phgr13_verify(st.VK, diff, bad, sol[:diff]), field


@pytest.mark.gpu
def test_gpu_proofs_pass_the_reference_verifiers(ps_api, ctx, co, pr):
    from oracle import pairing as pg
    from oracle import restate as rs

    up = lambda g, b: ps_api.Points.upload(ctx, g, b)
    for circuit in ("toy", 21):
        c, wit, sol, tr, trp, r, s, diff = _groth16_material(pr, co, rs, circuit)
        q = ps_api.QAP(ctx, c.nbVars, c.nbIO, c.left, c.right, c.out)
        dsol = ps_api.Poly.from_values(ctx, wit) if wit else ps_api.Poly.upload(ctx, sol)
        pk = ps_api.Groth16Setup(tr.Alpha, tr.Beta, tr.Delta, tr.Beta2, tr.Delta2, up(ps_api.G1, tr.Xi),
                                 up(ps_api.G2, tr.Xi2), up(ps_api.G1, tr.NioLP), up(ps_api.G1, tr.XiT))
        proof = ps_api.Groth16Prove(pk, q, dsol, r, s)
        assert pg.groth16_verify(trp, co.G1.from_b(proof.A), co.G2.from_b(proof.B), co.G1.from_b(proof.C), sol[:diff])
        # the proof elements also survive the compressed wire form kyber uses
        Ac = ps_api.point_convert(ps_api.G1, proof.A, ps_api.FMT_AFFINE, ps_api.FMT_COMPRESSED)
        assert co.g1_decompress(Ac) == co.G1.from_b(proof.A)
    rng = pr.SplitMix64(SEED + 606)
    c, wit = rs.toy_circuit()
    sol = [pr.fr(v) for v in wit]
    st = rs.phgr13_setup(c, *[rng.fr() for _ in range(8)])
    ek = ps_api.PHGR13EvalKey(**{f: up(ps_api.G2 if f == "ws" else ps_api.G1, getattr(st.EK, f))
                                 for f in ps_api.PHGR13EvalKey.FIELDS})
    q = ps_api.QAP(ctx, c.nbVars, c.nbIO, c.left, c.right, c.out)
    pp = ps_api.PHGR13Prove(ek, q, ps_api.Poly.from_values(ctx, wit))
    tup = rs.Bag(**{f: (co.G2 if f == "wss" else co.G1).from_b(getattr(pp, f)) for f in ps_api.PHGR13Proof.FIELDS})
    assert pg.phgr13_verify(st.VK, c.nbVars - c.nbIO, tup, sol[: c.nbVars - c.nbIO])


def test_product_pairing_is_bilinear_host_only(co, pr):
    """ps_pairing_equal (the product's host-side Pair, curve.go:36-38) -- runs without a GPU.
    Long scalar walks exercise the lazy-limb discipline of the Miller loop (a value carried
    without a multiplication must be re-canonicalised, see pairing.inc)."""
    from playsnark_amd import api

    rng = pr.SplitMix64(SEED + 808)
    g1 = lambda k: co.G1.to_b(co.G1.mul(k))
    g2 = lambda k: co.G2.to_b(co.G2.mul(k))
    for _ in range(3):
        a, b = rng.fr(), rng.fr()
        assert api.pairing_equal(g1(a), g2(b), g1(a * b % pr.R), g2(1))
        assert api.pairing_equal(g1(a), g2(b), g1(b), g2(a))
        assert not api.pairing_equal(g1(a), g2(b), g1((a * b + 1) % pr.R), g2(1))
    assert api.pairing_equal(co.G1.to_b(None), g2(5), g1(7), co.G2.to_b(None))  # identity pairs to one


def test_product_pairing_refuses_points_outside_the_subgroup(co, pr, off_subgroup):
    """Untrusted points must be canonical, on the curve AND of order r (kyber's UnmarshalBinary rejects the
    rest [upstream]; the Miller loop and the signed-digit folding assume order r).  Host-only: runs without a GPU."""
    from playsnark_amd import api

    g1 = lambda k: co.G1.to_b(co.G1.mul(k))
    g2 = lambda k: co.G2.to_b(co.G2.mul(k))
    bad1, bad2 = co.G1.to_b(off_subgroup[0]), co.G2.to_b(off_subgroup[1])
    assert api.pairing_equal(g1(6), g2(35), g1(210), g2(1))
    for args in ((bad1, g2(3), g1(3), g2(1)), (g1(3), bad2, g1(3), g2(1)), (g1(3), g2(1), bad1, g2(1)), (g1(3), g2(1), g1(3), bad2)):
        with pytest.raises(api.PlaysnarkError) as e:
            api.pairing_equal(*args)
        assert e.value.code == -3 and "subgroup" in str(e.value)
    inf_junk = bytearray(co.G1.to_b(None))
    inf_junk[50] = 7  # "identity" with non-zero trailing bytes
    with pytest.raises(api.PlaysnarkError) as e:
        api.pairing_equal(bytes(inf_junk), g2(3), g1(3), g2(1))
    assert e.value.code == -3
    with pytest.raises(api.PlaysnarkError):
        api.points_sum(api.G1, bytes(inf_junk))


@pytest.mark.gpu
def test_product_verifiers_accept_and_reject(ps_api, ctx, co, pr, off_subgroup):
    """ps_groth16_verify / ps_phgr13_verify on GPU-made proofs: accept, and reject
    the tampered proofs / public inputs of TestPinocchioInvalidProof (pinocchio_test.go:243-276);
    the verdicts agree with the oracle's independent Python pairing."""
    from oracle import pairing as pg
    from oracle import restate as rs

    up = lambda g, b: ps_api.Points.upload(ctx, g, b)
    c, wit, sol, tr, trp, r, s, diff = _groth16_material(pr, co, rs, 33)
    q = ps_api.QAP(ctx, c.nbVars, c.nbIO, c.left, c.right, c.out)
    pk = ps_api.Groth16Setup(tr.Alpha, tr.Beta, tr.Delta, tr.Beta2, tr.Delta2, up(ps_api.G1, tr.Xi), up(ps_api.G2, tr.Xi2),
                             up(ps_api.G1, tr.NioLP), up(ps_api.G1, tr.XiT))
    proof = ps_api.Groth16Prove(pk, q, ps_api.Poly.upload(ctx, sol), r, s)
    iolp = up(ps_api.G1, tr.IoLP)
    io = ps_api.Poly.upload(ctx, sol[:diff])
    verify = lambda p, pub: ps_api.Groth16Verify(ctx, tr.Alpha, tr.Beta2, tr.Gamma, tr.Delta2, iolp, p, pub)
    assert verify(proof, io)
    assert pg.groth16_verify(trp, co.G1.from_b(proof.A), co.G2.from_b(proof.B), co.G1.from_b(proof.C), sol[:diff])
    bad = ps_api.Groth16Proof(r, s, proof.A, proof.B, co.G1.to_b(co.G1.mul(12345)))
    assert not verify(bad, io)
    bad_io = list(sol[:diff])
    bad_io[1] = (bad_io[1] + 1) % pr.R
    assert not verify(proof, ps_api.Poly.upload(ctx, bad_io))
    with pytest.raises(ps_api.LengthMismatch):
        verify(proof, ps_api.Poly.upload(ctx, sol[: diff - 1]))
    # on-curve points of the wrong order are refused, in the proof and in the key (the soundness gap of an on-curve-only test)
    bad1, bad2 = co.G1.to_b(off_subgroup[0]), co.G2.to_b(off_subgroup[1])
    for forged in (ps_api.Groth16Proof(r, s, bad1, proof.B, proof.C), ps_api.Groth16Proof(r, s, proof.A, bad2, proof.C),
                   ps_api.Groth16Proof(r, s, proof.A, proof.B, bad1)):
        with pytest.raises(ps_api.PlaysnarkError) as e:
            verify(forged, io)
        assert e.value.code == -3
    with pytest.raises(ps_api.PlaysnarkError):
        ps_api.Groth16Verify(ctx, tr.Alpha, tr.Beta2, bad2, tr.Delta2, iolp, proof, io)
    with pytest.raises(ps_api.PlaysnarkError):
        ps_api.Groth16Verify(ctx, tr.Alpha, tr.Beta2, tr.Gamma, tr.Delta2, up(ps_api.G1, tr.IoLP[:96] + bad1 + tr.IoLP[192:]), proof, io)

    rng = pr.SplitMix64(SEED + 909)
    c, wit = rs.toy_circuit()
    sol = [pr.fr(v) for v in wit]
    diff = c.nbVars - c.nbIO
    st = rs.phgr13_setup(c, *[rng.fr() for _ in range(8)])
    ek = ps_api.PHGR13EvalKey(**{f: up(ps_api.G2 if f == "ws" else ps_api.G1, getattr(st.EK, f)) for f in ps_api.PHGR13EvalKey.FIELDS})
    q = ps_api.QAP(ctx, c.nbVars, c.nbIO, c.left, c.right, c.out)
    pp = ps_api.PHGR13Prove(ek, q, ps_api.Poly.from_values(ctx, wit))
    vk = st.VK
    vkb = {"av": co.G2.to_b(vk.av), "aw": co.G1.to_b(vk.aw), "ay": co.G2.to_b(vk.ay), "gamma": co.G2.to_b(vk.gamma),
           "bgamma": co.G1.to_b(vk.bgamma), "bgamma2": co.G2.to_b(vk.bgamma2), "yts": co.G2.to_b(vk.yts)}
    vs_io = up(ps_api.G1, co.G1.pack(vk.vs[:diff]))
    ws_io = up(ps_api.G2, co.G2.pack(vk.ws[:diff]))
    ys_io = up(ps_api.G1, co.G1.pack(vk.ys[:diff]))
    io = ps_api.Poly.from_values(ctx, wit[:diff])
    assert ps_api.PHGR13Verify(ctx, vkb, vs_io, ws_io, ys_io, pp, io)
    rnd = co.G1.to_b(co.G1.mul(rng.fr()))
    for field in ("yss", "vss", "hs", "gz", "vass"):  # pinocchio_test.go:243-262
        saved = getattr(pp, field)
        setattr(pp, field, rnd)
        assert not ps_api.PHGR13Verify(ctx, vkb, vs_io, ws_io, ys_io, pp, io), field
        setattr(pp, field, saved)
    for field, forged in (("wss", co.G2.to_b(off_subgroup[1])), ("hs", co.G1.to_b(off_subgroup[0]))):
        saved = getattr(pp, field)
        setattr(pp, field, forged)
        with pytest.raises(ps_api.PlaysnarkError) as e:
            ps_api.PHGR13Verify(ctx, vkb, vs_io, ws_io, ys_io, pp, io)
        assert e.value.code == -3
        setattr(pp, field, saved)
    for key in ("bgamma2", "av", "ay"):  # pinocchio_test.go:265-276 (random VK element)
        tampered = dict(vkb)
        tampered[key] = co.G2.to_b(co.G2.mul(rng.fr()))
        assert not ps_api.PHGR13Verify(ctx, tampered, vs_io, ws_io, ys_io, pp, io), key
