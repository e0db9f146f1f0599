"""Median of the quotient phase (and of the whole proof) over REPS Groth16 proofs at 2^LOG2N constraints, key as the reference's
setup makes it (MONOMIAL=1) or in Lagrange form; for A/B runs of two library builds on one box (PLAYSNARK_HIP_LIB)."""
import os, random, statistics, sys, time
sys.path.insert(0, os.getcwd())
import bench
from playsnark_amd import api
ctx = api.Context(0)
n = 1 << int(os.environ.get("LOG2N", "20"))
nvars, L, Rm, O, sol = bench.synthetic_r1cs(n)
q = api.QAP.from_csr(ctx, nvars, nvars - 3, L, Rm, O)
rnd = random.Random(1)
fr = lambda: rnd.randrange(1 << 20, bench.R_MOD)
dsol = api.Poly.upload(ctx, sol)
tr, vk = api.NewGroth16TrustedSetup(q, fr(), fr(), fr(), fr(), fr())
if os.environ.get("MONOMIAL"): tr = tr.monomial_only()
r, s = fr(), fr()
for _ in range(3): api.Groth16Prove(tr, q, dsol, r, s)
qs, ts = [], []
for _ in range(int(os.environ.get("REPS", "15"))):
    t0 = time.perf_counter(); api.Groth16Prove(tr, q, dsol, r, s); ts.append((time.perf_counter() - t0) * 1e3)
    qs.append(ctx.last_prove_phase_ms()["quotient"])
print("%s quotient median %.2f ms (min %.2f), proof median %.2f ms" % (os.environ.get("TAG", ""), statistics.median(qs), min(qs), statistics.median(ts)))
