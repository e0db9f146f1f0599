"""Wall time of the one-time monomial -> Lagrange conversion of a Groth16 key (ps_points_monomial_to_lagrange) at 2^LOG2N
constraints, each array checked byte for byte against the array the device setup computes from the toxic waste.
  LOG2N=16 python3 tools/pk_to_lagrange.py [xi xi2 xit]"""
import os
import random
import sys
import time

sys.path.insert(0, os.getcwd())
import bench  # noqa: E402
from playsnark_amd import api  # noqa: E402

ctx = api.Context(0)
n = 1 << int(os.environ.get("LOG2N", "16"))
which = sys.argv[1:] or ["xi", "xit", "xi2"]
nvars, L, Rm, O, sol = bench.synthetic_r1cs(n)
q = api.QAP.from_csr(ctx, nvars, nvars - 3, L, Rm, O)
rnd = random.Random(5)
fr = lambda: rnd.randrange(1 << 20, bench.R_MOD)
tr, _vk = api.NewGroth16TrustedSetup(q, fr(), fr(), fr(), fr(), fr())
ctx.sync()
for name, src, want, nodes in (("xi", tr.Xi, tr.LXi, 0), ("xit", tr.XiT, tr.LXiT, 1), ("xi2", tr.Xi2, tr.LXi2, 0)):
    if name not in which:
        continue
    t0 = time.perf_counter()
    got = src.to_lagrange(q, nodes)
    ctx.sync()
    dt = time.perf_counter() - t0
    same = got.download() == want.download()
    print("n = 2^%d  %-3s (%s, %d points): %.2f s, byte-identical to the setup's array: %s" % (
        n.bit_length() - 1, name, "G2" if name == "xi2" else "G1", len(src), dt, same), flush=True)
    assert same
