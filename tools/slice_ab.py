"""A/B of the slice length (and tail mode) on short sums: ms alone (best of 10) and three in flight.
Usage: python3 tools/slice_ab.py <g1|g2> <log2n> <slice,...> [tail modes]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from playsnark_amd import api  # noqa: E402
from playsnark_amd.dist import ShardedMsm  # noqa: E402

g, l = sys.argv[1], int(sys.argv[2])
slices = [int(x) for x in sys.argv[3].split(",")]
modes = [int(x) for x in (sys.argv[4].split(",") if len(sys.argv) > 4 else ["0"])]
gid = api.G1 if g == "g1" else api.G2
ctx = api.Context(0)
n = 1 << l
a = api.Poly.upload(ctx, bench.uniform_scalars_be32(n, 77 + l).tobytes())
pts = api.Points.from_scalars(ctx, gid, a).precompute(0)
sc = api.Poly.upload(ctx, bench.uniform_scalars_be32(n, 78 + l).tobytes())
m = ShardedMsm(ctx, gid, None, 1)
for mode in modes:
    for sl in slices:
        ctx.set_tail(mode)
        ctx.set_slice(sl)
        m.run_pipelined(pts, sc, 6, depth=3)
        ctx.sync()
        t0 = time.perf_counter()
        m.run_pipelined(pts, sc, 30, depth=3)
        ctx.sync()
        fl = (time.perf_counter() - t0) / 30 * 1e3
        best = 1e9
        for _ in range(10):
            t0 = time.perf_counter()
            m.run(pts, sc)
            best = min(best, (time.perf_counter() - t0) * 1e3)
        info = ctx.last_msm_info()
        print("%s 2^%d tail=%d slice=%-3d (M=%d c=%d)  %7.3f in flight  %7.3f alone" % (g, l, mode, sl, info["slice"], info["window_bits"], fl, best))
