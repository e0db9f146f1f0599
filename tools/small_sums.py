"""Short sums, no event timing: ms per sum three in flight and alone, G1 and G2, window table and plain plan.
Usage (GPU box, repository root): python3 tools/small_sums.py 10 13 16"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from playsnark_amd import api  # noqa: E402
from playsnark_amd.dist import ShardedMsm  # noqa: E402

ctx = api.Context(0)
for g, gid in (("g1", api.G1), ("g2", api.G2)):
    for l in [int(a) for a in sys.argv[1:]] or [10, 16]:
        n = 1 << l
        a = api.Poly.upload(ctx, bench.uniform_scalars_be32(n, 77 + l).tobytes())
        for table in (True, False):
            pts = api.Points.from_scalars(ctx, gid, a)
            if table:
                pts.precompute(0)
            sc = api.Poly.upload(ctx, bench.uniform_scalars_be32(n, 78 + l).tobytes())
            ctx.sync()
            m = ShardedMsm(ctx, gid, None, 1)
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
            print("%s 2^%-2d %-5s c=%-2d W=%-2d M=%-3d  %7.3f ms in flight  %7.3f ms alone (best of 10)" % (
                g, l, "table" if table else "plain", info["window_bits"], info["windows"], info["slice"], fl, best))
            pts.free()
            sc.free()
        a.free()
