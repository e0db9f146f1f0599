"""Soak run for the multi-stream paths (pending-sum queue, multi-array sums, both provers): the same
small jobs repeated, every result compared with the first iteration's.  Prints a line every 50
iterations.  Not collected by pytest (no test_ prefix); lives under tests/ because it builds its inputs
with the oracle.  Usage (GPU box): python tests/soak_provers.py [iterations]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import coracle as co  # noqa: E402  (test infrastructure: builds the inputs only)
from oracle import pyref as pr  # noqa: E402
from oracle import restate as rs  # noqa: E402
from playsnark_amd import api  # noqa: E402


def main(iters):
    ctx = api.Context(0)
    rng = pr.SplitMix64(4242)
    jobs = []
    for n_gates in (4, 50, 300):
        if n_gates == 4:
            c, wit = rs.toy_circuit()
            sol = [pr.fr(v) for v in wit]
        else:
            c, sol = rs.synthetic_circuit(n_gates)
            c = rs.SparseR1CS(c.nbVars, c.nbVars - 3, c.left, c.right, c.out)
        q = api.QAP(ctx, c.nbVars, c.nbIO, c.left, c.right, c.out)
        g16, _ = api.NewGroth16TrustedSetup(q, *[rng.fr() for _ in range(5)])
        ek, _ = api.NewPHGR13TrustedSetup(q, *[rng.fr() for _ in range(8)])
        jobs.append((q, g16, ek, api.Poly.upload(ctx, sol), rng.fr(), rng.fr()))
    sums = []
    for gid, og, n in ((api.G1, co.G1, 700), (api.G2, co.G2, 90), (api.G1, co.G1, 5000)):
        sc = api.Poly.upload(ctx, [rng.fr() for _ in range(n)])
        pts = api.Points.upload(ctx, gid, og.gen_points(rng.fr(), rng.fr(), n))
        sums.append((gid, pts, sc))

    def one_round():
        out = []
        for q, g16, ek, sol, r, s in jobs:
            p = api.PHGR13Prove(ek, q, sol)
            out.append(tuple(getattr(p, f) for f in api.PHGR13Proof.FIELDS))
            g = api.Groth16Prove(g16, q, sol, r, s)
            out.append((g.A, g.B, g.C))
            out.append(q.Quotient(sol).download())
        order = [0, 1, 2, 0, 2, 1]
        launched = finished = 0
        while finished < len(order):
            while launched < len(order) and launched - finished < 3:
                api.msm_launch(ctx, sums[order[launched]][1], sums[order[launched]][2])
                launched += 1
            out.append(api.msm_finish(ctx, sums[order[finished]][0]))
            finished += 1
        out.append(tuple(api.msm_multi(ctx, [sums[0][1]] * 5, sums[0][2])))
        return out

    first = one_round()
    t0 = time.time()
    for it in range(1, iters + 1):
        assert one_round() == first, f"iteration {it} differs"
        if it % 50 == 0:
            print(f"soak: {it} iterations, {time.time() - t0:.1f} s", flush=True)
    print("soak ok", flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 300)
