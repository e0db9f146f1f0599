"""PHGR13Prove at 2^20 constraints in a loop (for rocprofv3 timelines / A-B runs)."""
import os, random, sys, time
sys.path.insert(0, os.getcwd())
import bench
from playsnark_amd import api
ctx = api.Context(0)
if os.environ.get("NOTAB"): ctx.set_tables(False)
n = 1 << int(os.environ.get("LOG2N", "20"))
nvars, L, Rm, O, sol = bench.synthetic_r1cs(n)
q = api.QAP.from_csr(ctx, nvars, nvars - 3, L, Rm, O)
rnd = random.Random(1)
fr = lambda: rnd.randrange(1 << 20, bench.R_MOD)
dsol = api.Poly.upload(ctx, sol)
ek, vk = api.NewPHGR13TrustedSetup(q, *[fr() for _ in range(8)])
api.PHGR13Prove(ek, q, dsol)
api.PHGR13Prove(ek, q, dsol)
ts = []
for _ in range(int(os.environ.get("REPS", "5"))):
    t0 = time.perf_counter(); api.PHGR13Prove(ek, q, dsol); ts.append((time.perf_counter() - t0) * 1e3)
print(os.environ.get("TAG", ""), "phgr13 ms", [round(t, 2) for t in ts], {k: round(v, 2) for k, v in ctx.last_prove_phase_ms().items()})
proof = api.PHGR13Prove(ek, q, dsol)
diff = 3  # nbVars - nbIO: const, x, out
io = api.Poly.upload(ctx, sol[:diff])
args = (vk.vs.slice(0, diff), vk.ws.slice(0, diff), vk.ys.slice(0, diff))
tv = []
for _ in range(3):
    t0 = time.perf_counter(); ok = api.PHGR13Verify(ctx, vk.fixed_points(), *args, proof, io); tv.append((time.perf_counter() - t0) * 1e3)
print("phgr13 verify", ok, "ms", [round(t, 1) for t in tv])
