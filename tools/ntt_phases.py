"""Phase timeline of ONE NTT pass (measurement build with -DPS_NTT_TUNE): every workgroup stamps s_memtime (100 MHz) at its
start, after its tile is loaded (first barrier), after its butterfly stages, and after its stores are issued.
On gfx950 the counter runs at the shader clock (a 60 us kernel spans ~1.2e5 ticks), and the XCDs' counters are not
synchronised: only per-workgroup differences mean anything; they are printed in kilo-cycles.
  PLAYSNARK_HIP_LIB=playsnark_amd/libps_tune.so PS_NTT_TRACE_LAUNCH=<i> python3 tools/ntt_phases.py [log2 size]
The traced launch is the i-th pass of the process (ps_qap_create's table building launches the first few hundred; with
POLY=1 the process only runs ps_poly_mul: 3 transforms of 2^p, launches 0..)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from playsnark_amd import api, _lib  # noqa: E402

p = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = api.Context(0)
n = 1 << (p - 1)
a = api.Poly.upload(ctx, [(i * 7 + 3) % 1000003 for i in range(n)])
for _ in range(2):
    a.Mul(a)
lib = _lib.lib
out = (C.c_uint64 * (4 * 8192))()
meta = (C.c_int * 6)()
lib.ps_debug_ntt_trace.argtypes = [C.c_void_p, C.c_void_p]
rc = lib.ps_debug_ntt_trace(out, meta)
assert rc == 0, "no traced launch (PS_NTT_TRACE_LAUNCH?)"
grid, k, logD, inv, pp, threads = list(meta)
rows = [(out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]) for i in range(grid)]
t0 = min(r[0] for r in rows)
us = lambda t: (t - t0) / 1000.0
print("pass of a 2^%d transform: %d workgroups x %d threads, k = %d stages, logD = %d, %s" % (pp, grid, threads, k, logD, "inverse" if inv else "forward"))
import statistics as st
d = lambda i, j: sorted((r[j] - r[i]) / 1000.0 for r in rows)
for name, i, j in (("load", 0, 1), ("stages", 1, 2), ("store issue", 2, 3), ("lifetime", 0, 3)):
    v = d(i, j)
    print("  per workgroup %-12s min %7.2f  median %7.2f  max %7.2f kilo-cycles" % (name, v[0], st.median(v), v[-1]))
