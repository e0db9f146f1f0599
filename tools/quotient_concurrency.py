"""Is the quotient's NTT pipeline leaving the chip idle?  ps_qap_quotient (Groth16 route: A, B, h) at 2^LOG2N constraints from ONE
thread, then from TWO threads on two contexts (two QAP objects of the same circuit) at once: ms per quotient either way.  If two
at once take clearly less than twice one alone, independent work fills what the passes of a lone quotient leave idle."""
import os, sys, threading, time
sys.path.insert(0, os.getcwd())
import bench
from playsnark_amd import api
n = 1 << int(os.environ.get("LOG2N", "20"))
reps = int(os.environ.get("REPS", "10"))
nvars, L, Rm, O, sol = bench.synthetic_r1cs(n)
ctxs = [api.Context(0), api.Context(0)]
qs = [api.QAP.from_csr(c, nvars, nvars - 3, L, Rm, O) for c in ctxs]
sols = [api.Poly.upload(c, sol) for c in ctxs]
def run(i, k, route):
    for _ in range(k):
        if route == "ab": qs[i].computeAB(sols[i])
        else: qs[i].Quotient(sols[i])
        ctxs[i].sync()
for route in ("ab", "h"):
    run(0, 2, route); run(1, 2, route)
    t0 = time.perf_counter(); run(0, reps, route); one = (time.perf_counter() - t0) / reps * 1e3
    th = [threading.Thread(target=run, args=(i, reps, route)) for i in range(2)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; two = (time.perf_counter() - t0) / (2 * reps) * 1e3
    print("route %-2s: one at a time %.2f ms per quotient; two at once %.2f ms per quotient (%.2f per pair)" % (route, one, two, 2 * two), flush=True)
