"""CPU: the C oracle against the committed golden fixtures (written by the independent pure-Python
twin, tests/golden/gen_golden.py) and against the public BLS12-381 constants."""
import json
import os

import pytest

G = os.path.join(os.path.dirname(__file__), "golden")
load = lambda n: json.load(open(os.path.join(G, n)))
H = lambda s: int(s, 16)


def test_public_constants(pr):
    assert pr.check_constants()
    # ZCash/IETF compressed generator encodings (public known answers)
    assert pr.g1_compress(pr.G1.gen).hex() == (
        "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")
    # the full 96 bytes: x_c1 || x_c0 with the flag bits on the first byte -- this also pins the coordinate ORDER of G2
    assert pr.g2_compress(pr.G2.gen).hex() == (
        "93e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
        "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8")


def test_curve_known_answers(co, pr):
    kat = load("curve_kat.json")
    assert co.g1_compress(pr.G1.gen).hex() == kat["g1_generator_compressed"]
    assert co.g2_compress(pr.G2.gen).hex() == kat["g2_generator_compressed"]
    for row in kat["mul"]:
        k = H(row["k"])
        p1, p2 = co.G1.mul(k), co.G2.mul(k)
        assert pr.g1_to_bytes(p1).hex() == row["g1"] and pr.g2_to_bytes(p2).hex() == row["g2"]
        assert co.g1_compress(p1).hex() == row["g1c"] and co.g2_compress(p2).hex() == row["g2c"]
        assert co.g1_decompress(bytes.fromhex(row["g1c"])) == p1
        assert co.g2_decompress(bytes.fromhex(row["g2c"])) == p2
        assert co.G1.on_curve(p1) and co.G2.on_curve(p2)
    for name, grp, frm in (("g1", co.G1, pr.g1_from_bytes), ("g2", co.G2, pr.g2_from_bytes)):
        a, b, s, d = (frm(bytes.fromhex(x)) for x in kat["add"][name])
        assert grp.add(a, b) == s and grp.add(a, a) == d and grp.add(a, None) == a


def test_reference_fixed_fr_cases(co, pr):
    fc = load("fr_cases.json")
    c = fc["TestAlgebraEval"]
    assert co.poly_eval([pr.fr(v) for v in c["p"]], c["x"]) == c["want"]
    c = fc["TestAlgebraPolyMul"]
    assert co.poly_mul(c["p1"], c["p2"]) == c["want"]
    c = fc["TestAlgebraMinimal"]
    q, rem = co.poly_div2([pr.fr(v) for v in c["p"]], [pr.fr(v) for v in c["z"]])
    assert q == c["q"] and rem == c["rem"]
    c = fc["TestAlgebraPolyDivManual"]
    q, rem = co.poly_div2(c["p1"], c["p2"])
    assert q == [0, 2] and rem == [0]


def test_toy_qap_fixture(co, pr):
    t = load("toy_qap.json")
    wit = t["witness"]
    assert wit == [1, 3, 35, 9, 27, 30] and t["vars"] == ["const", "x", "out", "u", "v", "w"]
    assert (t["Ls"], t["Rs"], t["Os"]) == ([3, 9, 30, 35], [3, 3, 1, 1], [9, 27, 30, 35])
    assert [v if v < 2**200 else v - pr.R for v in map(H, t["z"])] == [24, -50, 35, -10, 1]
    left, right, out, z = co.to_qap_dense(t["left"], t["right"], t["out"])
    assert z == list(map(H, t["z"]))
    assert left == [list(map(H, p)) for p in t["left_polys"]]
    sol = [pr.fr(v) for v in wit]
    A, B, C = (co.aggregate_poly(p, sol) for p in (left, right, out))
    assert (A, B, C) == tuple(list(map(H, t[k])) for k in "ABC")
    h = co.quotient_from_aggregates(A, B, C, z)
    assert h == list(map(H, t["h"]))
    # the hex values of SURVEY.md 8c
    assert t["h"][0] == "4d491a377113a8daccd13ab0066be558e27e6d5755543d54aaaaaaa9fffffffd"
    assert t["A"][1] == "26a48d1bb889d46d66689d580335f2ac713f36abaaaa1eaa55555554ffffffb7" and H(t["A"][0]) == 0x2B
    # interpolating the value vectors gives the same polynomials (linearity of qap.go:168-173)
    assert co.quotient_from_values(t["Ls"], t["Rs"], t["Os"]) == (A, B, C, h)


def test_msm_small_fixture(co, pr):
    m = load("msm_small.json")
    sc = list(map(H, m["scalars"]))
    for name, grp in (("g1", co.G1), ("g2", co.G2)):
        raw = bytes.fromhex(m[f"{name}_points"])
        pts = grp.unpack(raw)
        for d, p in zip(map(H, m["dlogs"]), pts):
            assert grp.mul(d) == p
        want = bytes.fromhex(m[f"{name}_result"])
        assert grp.to_b(grp.blind_eval(sc, raw)) == want
        assert grp.to_b(grp.msm_pippenger(co.pack_fr(sc), raw, len(sc), 3)) == want
    assert co.G1.to_b(co.G1.blind_eval_i64(m["i64_scalars"], bytes.fromhex(m["g1_points"]))).hex() == m["g1_result_i64"]


def test_groth16_toy_fixture(co, pr):
    from oracle import restate as rs

    g = load("groth16_toy.json")
    c, wit = rs.toy_circuit()
    tox = [H(g["toxic"][k]) for k in ("alpha", "beta", "delta", "x", "gamma")]
    tr = rs.groth16_setup(c, *tox)
    for k in ("Alpha", "Beta", "Delta", "Beta2", "Delta2", "Xi", "Xi2", "NioLP", "IoLP", "XiT"):
        assert getattr(tr, k).hex() == g[k], k
    sol = [pr.fr(v) for v in wit]
    pf = rs.groth16_prove(tr, c, sol, H(g["r"]), H(g["s"]))
    assert (pf.A.hex(), pf.B.hex(), pf.C.hex()) == (g["A"], g["B"], g["C"])
    assert co.g1_compress(co.G1.from_b(pf.A)).hex() == g["A_compressed"]
    assert co.g2_compress(co.G2.from_b(pf.B)).hex() == g["B_compressed"]
    assert rs.groth16_dlog_check(tr, c, sol, pf) == (True, True, True)


def test_phgr13_toy_fixture(co, pr):
    from oracle import restate as rs

    g = load("phgr13_toy.json")
    c, wit = rs.toy_circuit()
    rnd = [H(g["randomness"][k]) for k in ("s", "av", "aw", "ay", "rv", "rw", "beta", "gamma")]
    st = rs.phgr13_setup(c, *rnd)
    for k, v in g["ek"].items():
        assert getattr(st.EK, k).hex() == v, k
    pp = rs.phgr13_prove(st.EK, c, [pr.fr(v) for v in wit])
    for k, v in g["proof"].items():
        assert getattr(pp, k).hex() == v, k


@pytest.mark.parametrize("n", [2, 3, 4, 7, 33, 64, 100, 257])
def test_fast_quotient_is_the_literal_algorithm(co, pr, n):
    """B1-h: the quasi-linear CPU oracle for A, B, C, h (NTT products, Newton basis, product tree, series
    division) against the literal restatement of computeAggregatePoly + Mul + Sub + Div2 (qap.go:151-175) -- which is what
    pins it; the GPU tests then use it at sizes the literal algorithm cannot reach (2^16)."""
    from oracle import restate as rs

    if n >= 4:
        c, sol = rs.synthetic_circuit(n)
        dot = lambda rows: [sum(v * sol[j] for j, v in row) % pr.R for row in rows]
        yA, yB, yC = dot(c.left), dot(c.right), dot(c.out)
    else:
        rng = pr.SplitMix64(n)
        yA, yB = [rng.fr() for _ in range(n)], [rng.fr() for _ in range(n)]
        yC = [a * b % pr.R for a, b in zip(yA, yB)]
    assert tuple(map(list, co.fast_quotient(yA, yB, yC))) == tuple(map(list, co.quotient_from_values(yA, yB, yC)))
    yC[n // 2] = (yC[n // 2] + 1) % pr.R
    with pytest.raises(ArithmeticError):
        co.fast_quotient(yA, yB, yC)


def test_final_exponentiation_chain_identity():
    """The x-chain of the host pairing's final exponentiation (csrc/pairing_math.inc::final_exp) rests on
    3 (p^4 - p^2 + 1) / r = (x - 1)^2 (x + p)(x^2 + p^2 - 1) + 3 for the BLS12-381 parameter x (Hayashida, Hayasaka,
    Teruya 2020), and on w^(p-1) = xi^((p-1)/6) needing 6 | p - 1; 3 is prime to r, so the cube decides "== 1" alike."""
    from oracle import pyref as pr

    p, r = pr.P, pr.R
    x = -0xD201000000010000
    assert (p**4 - p**2 + 1) % r == 0 and (p - 1) % 6 == 0 and r % 3 != 0
    assert 3 * ((p**4 - p**2 + 1) // r) == (x - 1) ** 2 * (x + p) * (x * x + p * p - 1) + 3
    assert (p**12 - 1) == (p**6 - 1) * (p**2 + 1) * (p**4 - p**2 + 1)
