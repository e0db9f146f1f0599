#!/usr/bin/env python3
"""Generates the committed golden fixtures (tests/golden/*.json) with the pure-Python big-int
twin oracle/pyref.py -- independent of the C oracle and of the HIP code, which are both checked
against these files.  The reference itself (Go) cannot run in this image and its tests hold no
fixed point/proof vectors, so these are restatement outputs, not reference outputs; the toy
polynomial values coincide with the ones derived in SURVEY.md section 8c.

    python tests/golden/gen_golden.py        (takes ~20 s)
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import pyref as pr  # noqa: E402

hx = lambda v: "%064x" % v


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1, sort_keys=True)
        f.write("\n")


def main():
    assert pr.check_constants()
    rng = pr.SplitMix64()

    # ---- curve known answers ----
    ks = [1, 2, 3, 0xDEADBEEF, pr.R - 1, (1 << 255) % pr.R] + [rng.fr() for _ in range(4)]
    kat = {"g1_generator_compressed": pr.g1_compress(pr.G1.gen).hex(),
           "g2_generator_compressed": pr.g2_compress(pr.G2.gen).hex(), "mul": []}
    for k in ks:
        p1, p2 = pr.G1.mul(k), pr.G2.mul(k)
        kat["mul"].append({"k": hx(k), "g1": pr.g1_to_bytes(p1).hex(), "g1c": pr.g1_compress(p1).hex(),
                           "g2": pr.g2_to_bytes(p2).hex(), "g2c": pr.g2_compress(p2).hex()})
    a, b = pr.G1.mul(ks[6]), pr.G1.mul(ks[7])
    a2, b2 = pr.G2.mul(ks[6]), pr.G2.mul(ks[7])
    kat["add"] = {"g1": [pr.g1_to_bytes(a).hex(), pr.g1_to_bytes(b).hex(), pr.g1_to_bytes(pr.G1.add(a, b)).hex(),
                         pr.g1_to_bytes(pr.G1.add(a, a)).hex()],
                  "g2": [pr.g2_to_bytes(a2).hex(), pr.g2_to_bytes(b2).hex(), pr.g2_to_bytes(pr.G2.add(a2, b2)).hex(),
                         pr.g2_to_bytes(pr.G2.add(a2, a2)).hex()]}
    dump("curve_kat.json", kat)

    # ---- the reference's fixed Fr cases ----
    fr_cases = {
        "TestAlgebraEval": {"p": [1, 1], "x": 1, "want": 2},                               # algebra_test.go:10-19
        "TestAlgebraPolyMul": {"p1": [1, 2], "p2": [3, 0, 1], "want": [3, 6, 1, 2]},       # :76-104
        "TestAlgebraMinimal": {"p": [0, 4, -6, 2], "z": [2, -3, 1], "q": [0, 2], "rem": [0, 0]},  # :48-74
        "TestAlgebraPolyDivManual": {"p1": [0, 2, 4], "p2": [1, 2]},                       # :155-177
    }
    dump("fr_cases.json", fr_cases)

    # ---- toy QAP (config #1) ----
    c = pr.create_r1cs()
    wit = pr.create_witness(c)
    q = pr.to_qap(c)
    A, B, C = pr.compute_aggregate_poly(q, wit)
    h = pr.quotient(q, wit)
    dump("toy_qap.json", {
        "vars": c.vars, "witness": wit, "nbVars": q.nbVars, "nbGates": q.nbGates, "nbIO": q.nbIO,
        "left": c.left, "right": c.right, "out": c.out,
        "Ls": [sum(x * y for x, y in zip(row, wit)) for row in c.left],
        "Rs": [sum(x * y for x, y in zip(row, wit)) for row in c.right],
        "Os": [sum(x * y for x, y in zip(row, wit)) for row in c.out],
        "z": [hx(v) for v in q.z], "A": [hx(v) for v in A], "B": [hx(v) for v in B], "C": [hx(v) for v in C],
        "h": [hx(v) for v in h],
        "left_polys": [[hx(v) for v in p] for p in q.left],
    })

    # ---- small MSMs with known discrete logs ----
    n = 12
    dl = [rng.fr() for _ in range(n)]
    sc = [rng.fr() for _ in range(n)]
    sc[3], sc[4], sc[5] = 0, 1, pr.R - 1
    p1 = [pr.G1.mul(d) for d in dl]
    p2 = [pr.G2.mul(d) for d in dl]
    dump("msm_small.json", {
        "dlogs": [hx(v) for v in dl], "scalars": [hx(v) for v in sc],
        "g1_points": b"".join(pr.g1_to_bytes(p) for p in p1).hex(),
        "g2_points": b"".join(pr.g2_to_bytes(p) for p in p2).hex(),
        "g1_result": pr.g1_to_bytes(pr.G1.msm(sc, p1)).hex(),
        "g2_result": pr.g2_to_bytes(pr.G2.msm(sc, p2)).hex(),
        "i64_scalars": [0, 1, 1, -1, 35, 9, 27, 30, -5, 1000003, 2, 3],
        "g1_result_i64": pr.g1_to_bytes(pr.G1.msm([pr.fr(v) for v in [0, 1, 1, -1, 35, 9, 27, 30, -5, 1000003, 2, 3]], p1)).hex(),
    })

    # ---- Groth16 on the toy, fixed toxic waste and (r, s) ----
    tox = [rng.fr() for _ in range(5)]
    r, s = rng.fr(), rng.fr()
    tr = pr.groth16_setup(q, *tox)
    pf = pr.groth16_prove(tr, q, wit, r, s)
    dump("groth16_toy.json", {
        "toxic": dict(zip(["alpha", "beta", "delta", "x", "gamma"], map(hx, tox))), "r": hx(r), "s": hx(s),
        "Alpha": pr.g1_to_bytes(tr.Alpha).hex(), "Beta": pr.g1_to_bytes(tr.Beta).hex(),
        "Delta": pr.g1_to_bytes(tr.Delta).hex(), "Beta2": pr.g2_to_bytes(tr.Beta2).hex(),
        "Delta2": pr.g2_to_bytes(tr.Delta2).hex(),
        "Xi": b"".join(map(pr.g1_to_bytes, tr.Xi)).hex(), "Xi2": b"".join(map(pr.g2_to_bytes, tr.Xi2)).hex(),
        "NioLP": b"".join(map(pr.g1_to_bytes, tr.NioLP)).hex(), "IoLP": b"".join(map(pr.g1_to_bytes, tr.IoLP)).hex(),
        "XiT": b"".join(map(pr.g1_to_bytes, tr.XiT)).hex(),
        "A": pr.g1_to_bytes(pf.A).hex(), "B": pr.g2_to_bytes(pf.B).hex(), "C": pr.g1_to_bytes(pf.C).hex(),
        "A_compressed": pr.g1_compress(pf.A).hex(), "B_compressed": pr.g2_compress(pf.B).hex(),
        "C_compressed": pr.g1_compress(pf.C).hex(),
    })

    # ---- PHGR13 on the toy ----
    rnd = [rng.fr() for _ in range(8)]
    st = pr.phgr13_setup(q, *rnd)
    pp = pr.phgr13_prove(st.EK, q, wit)
    ek = {f: b"".join(map(pr.g2_to_bytes if f == "ws" else pr.g1_to_bytes, getattr(st.EK, f))).hex()
          for f in ("vs", "ws", "ys", "vas", "was", "yas", "gsi", "vbs", "wbs", "ybs")}
    proof = {f: (pr.g2_to_bytes if f == "wss" else pr.g1_to_bytes)(getattr(pp, f)).hex()
             for f in ("vss", "vass", "wss", "wass", "yss", "yass", "hs", "gz")}
    dump("phgr13_toy.json", {
        "randomness": dict(zip(["s", "av", "aw", "ay", "rv", "rw", "beta", "gamma"], map(hx, rnd))),
        "ek": ek, "proof": proof,
    })
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
