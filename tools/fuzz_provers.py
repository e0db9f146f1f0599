"""Differential soak of Groth16Prove / PHGR13Prove (both key forms, device setups) against the oracle's statement-by-
statement restatement: random circuit sizes, IO splits, toxic waste and r, s; a corrupted witness must raise Apocalypse.
  python3 tools/fuzz_provers.py [seconds] [seed]        (GPU box; exits non-zero on the first mismatch)"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyref as pr, restate as rs  # noqa: E402
from playsnark_amd import api  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
ctx = api.Context(0)
fr = lambda: rnd.randrange(1, pr.R)
t_end = time.time() + budget
cases = 0
while time.time() < t_end:
    n = rnd.choice([2, 3, 4, 5, 8, 15, 16, 17, 31, 33, 63, 64, 65, 100, 127, 129, rnd.randrange(2, 260)])
    if cases % 8 == 7:  # keys of >= 1024 points: window tables, the tree tail over a table plan, multi-sums over both groups
        n = rnd.choice([1024, 1500, 2048, rnd.randrange(1024, 3000)])
    kind = rnd.choice(["synthetic", "synthetic", "bits"])
    c, sol = rs.synthetic_circuit(n, x0=rnd.randrange(2, 1000)) if kind == "synthetic" else rs.bit_circuit(n, seed=rnd.randrange(1 << 30))
    nio = rnd.randrange(1, c.nbVars)  # the reference's diff quirk: the first non-IO index is nbVars - nbIO
    if n >= 1024 and rnd.random() < 0.7:
        nio = c.nbVars - rnd.randrange(1, 6)  # most variables on the prover's side, as in a real circuit
    c = rs.SparseR1CS(c.nbVars, nio, c.left, c.right, c.out)
    q = api.QAP(ctx, c.nbVars, c.nbIO, c.left, c.right, c.out)
    dsol = api.Poly.upload(ctx, sol)
    bad = list(sol)
    bad[rnd.randrange(1, len(bad))] += 1
    info = dict(n=n, kind=kind, nbVars=c.nbVars, nbIO=c.nbIO, seed=seed, case=cases)
    # Groth16
    tox = [fr() for _ in range(5)]
    while tox[3] <= 2 * n:  # x off the interpolation nodes
        tox[3] = fr()
    want = rs.groth16_setup(c, *tox)
    tr, vk = api.NewGroth16TrustedSetup(q, *tox)
    r, s = fr(), fr()
    ctx.set_tail(rnd.choice([0, 0, 1, 2]))
    ref = rs.groth16_prove(want, c, sol, r, s, fast=n > 16)
    for key in (tr, tr.monomial_only()):
        p = api.Groth16Prove(key, q, dsol, r, s)
        if (p.A, p.B, p.C) != (ref.A, ref.B, ref.C):
            print("MISMATCH groth16", info)
            sys.exit(1)
    bad_ok = False
    try:
        api.Groth16Prove(tr, q, api.Poly.upload(ctx, [v % pr.R for v in bad]), r, s)
    except api.Apocalypse:
        bad_ok = True
    # a changed wire may still satisfy every gate it touches only if it touches none
    touched = any(bad[i] != sol[i] and any(i in dict(row) for m in (c.left, c.right, c.out) for row in m) for i in range(len(sol)))
    if kind == "synthetic" and touched and not bad_ok:  # (a flipped bit 0 -> 1 still satisfies b * b = b)
        print("NO APOCALYPSE groth16", info)
        sys.exit(1)
    # PHGR13
    tox = [fr() for _ in range(8)]
    while tox[0] <= 2 * n:
        tox[0] = fr()
    wantp = rs.phgr13_setup(c, *tox)
    ek, pvk = api.NewPHGR13TrustedSetup(q, *tox)
    refp = rs.phgr13_prove(wantp.EK, c, sol, fast=n > 16)
    for key in (ek, ek.monomial_only()):
        p = api.PHGR13Prove(key, q, dsol)
        for f in api.PHGR13Proof.FIELDS:
            if getattr(p, f) != getattr(refp, f):
                print("MISMATCH phgr13", f, info)
                sys.exit(1)
    ctx.set_tail(0)
    cases += 1
    if cases % 10 == 0:
        print(cases, "cases ok", flush=True)
print("fuzz ok:", cases, "circuits x 2 provers x 2 key forms, seed", seed)
