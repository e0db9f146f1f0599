"""Thirty G1 sums of 2^k points, three in flight, window table (for tools/inflight_timeline.sh)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from playsnark_amd import api  # noqa: E402
from playsnark_amd.dist import ShardedMsm  # noqa: E402

l = int(sys.argv[1]) if len(sys.argv) > 1 else 16
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = api.Context(0)
n = 1 << l
a = api.Poly.upload(ctx, bench.uniform_scalars_be32(n, 77 + l).tobytes())
pts = api.Points.from_scalars(ctx, api.G1, a).precompute(0)
sc = api.Poly.upload(ctx, bench.uniform_scalars_be32(n, 78 + l).tobytes())
m = ShardedMsm(ctx, api.G1, None, 1)
m.run_pipelined(pts, sc, 6, depth=depth)
ctx.sync()
t0 = time.perf_counter()
m.run_pipelined(pts, sc, 30, depth=depth)
ctx.sync()
print("ms per sum in flight", (time.perf_counter() - t0) / 30 * 1e3)
