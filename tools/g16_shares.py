"""ps_groth16_prove_shard per rank on ONE GPU (world simulated ranks, one after the other): ms per share and its phases."""
import os, sys, time, random
sys.path.insert(0, os.getcwd())
import bench
from playsnark_amd import api
from playsnark_amd.dist import ShardedGroth16
ctx = api.Context(0)
n = 1 << int(os.environ.get("LOG2N", "20"))
world = int(os.environ.get("WORLD", "8"))
nvars, L, Rm, O, sol = bench.synthetic_r1cs(n)
q = api.QAP.from_csr(ctx, nvars, nvars - 3, L, Rm, O)
rnd = random.Random(1)
fr = lambda: rnd.randrange(1 << 20, bench.R_MOD)
dsol = api.Poly.upload(ctx, sol)
tr, vk = api.NewGroth16TrustedSetup(q, fr(), fr(), fr(), fr(), fr())
if os.environ.get("MONOMIAL"): tr = tr.monomial_only()
r, s = fr(), fr()
whole = api.Groth16Prove(tr, q, dsol, r, s)
api.Groth16Prove(tr, q, dsol, r, s)
t0 = time.perf_counter(); api.Groth16Prove(tr, q, dsol, r, s); print("unsharded ms %.2f" % ((time.perf_counter() - t0) * 1e3), ctx.last_prove_phase_ms())
sh = ShardedGroth16(ctx, None, world, 0)
for rep in range(2):
    parts = []
    for g in range(world):
        t0 = time.perf_counter(); parts.append(sh.partials(tr, q, dsol, r, s, rank=g)); ms = (time.perf_counter() - t0) * 1e3
        print("rep %d rank %d of %d: %.2f ms" % (rep, g, world, ms), {k: round(v, 2) for k, v in ctx.last_prove_phase_ms().items()}, ctx.last_msm_info())
    f = sh.fold(parts, r, s)
    assert (f.A, f.B, f.C) == (whole.A, whole.B, whole.C)
