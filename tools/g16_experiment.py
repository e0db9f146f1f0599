"""Groth16Prove at 2^20 constraints in a loop (for rocprofv3 timelines / PMC passes / A-B runs)."""
import sys, time, os, random
sys.path.insert(0, os.getcwd())
import bench
from playsnark_amd import api
ctx = api.Context(0)
if os.environ.get("NOTAB"): ctx.set_tables(False)
n = 1 << int(os.environ.get("LOG2N", "20"))
if os.environ.get("CIRCUIT") == "bits":  # n booleanity gates b*b = b over [const, b_1..b_n], a witness of random bits
    import numpy as np
    ptr = np.arange(n + 1, dtype=np.uint32); col = np.arange(1, n + 1, dtype=np.uint32); val = np.ones(n, dtype=np.int64)
    nvars, L, Rm, O = n + 1, (ptr, col, val), (ptr, col, val), (ptr, col, val)
    sol = [1] + np.random.RandomState(7).randint(0, 2, size=n).tolist()
    q = api.QAP.from_csr(ctx, nvars, n, L, Rm, O)
else:
    nvars, L, Rm, O, sol = bench.synthetic_r1cs(n)
    q = api.QAP.from_csr(ctx, nvars, nvars - 3, L, Rm, O)
rnd = random.Random(1)
fr = lambda: rnd.randrange(1 << 20, bench.R_MOD)
dsol = api.Poly.upload(ctx, sol)
ctx.sync(); t0 = time.perf_counter()
tr, vk = api.NewGroth16TrustedSetup(q, fr(), fr(), fr(), fr(), fr())
ctx.sync(); setup_ms = (time.perf_counter() - t0) * 1e3
if os.environ.get("MONOMIAL"): tr = tr.monomial_only()
r, s = fr(), fr()
t0 = time.perf_counter()
api.Groth16Prove(tr, q, dsol, r, s)
print("device setup ms %.1f, first proof (builds the window tables) ms %.1f" % (setup_ms, (time.perf_counter() - t0) * 1e3))
api.Groth16Prove(tr, q, dsol, r, s)
ts = []
for _ in range(int(os.environ.get("REPS", "5"))):
    t0 = time.perf_counter(); api.Groth16Prove(tr, q, dsol, r, s); ts.append((time.perf_counter() - t0) * 1e3)
print(os.environ.get("TAG", ""), "groth16 ms", [round(t, 2) for t in ts], {k: round(v, 2) for k, v in ctx.last_prove_phase_ms().items()})
proof = api.Groth16Prove(tr, q, dsol, r, s)
diff = 1 if os.environ.get("CIRCUIT") == "bits" else 3
io = api.Poly.upload(ctx, sol[:diff]) if os.environ.get("CIRCUIT") != "bits" else api.Poly.from_values(ctx, sol[:diff])
tv = []
for _ in range(3):
    t0 = time.perf_counter(); ok = api.Groth16Verify(ctx, tr.Alpha, tr.Beta2, vk["Gamma"], tr.Delta2, vk["IoLP"], proof, io); tv.append((time.perf_counter() - t0) * 1e3)
print("groth16 verify", ok, "ms", [round(t, 1) for t in tv])
