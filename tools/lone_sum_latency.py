import os, sys, time, statistics
sys.path.insert(0, os.getcwd())
import bench
from playsnark_amd import api
from playsnark_amd.dist import ShardedMsm
import numpy as np
ctx = api.Context(0)
n = 1 << 20
a = api.Poly.upload(ctx, bench.uniform_scalars_be32(n, 77).tobytes())
pts = api.Points.from_scalars(ctx, api.G1, a).precompute(0)
host = bench.uniform_scalars_be32(n, 78).tobytes()
sc = api.Poly.upload(ctx, host)
m = ShardedMsm(ctx, api.G1, None, 1)
for timing in (False, True, False, True):
    ctx.set_timing(timing)
    for _ in range(3): m.run(pts, sc)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); m.run(pts, sc); ts.append((time.perf_counter() - t0) * 1e3)
    hs = []
    api.blind_eval_host(ctx, pts, host)
    for _ in range(10):
        t0 = time.perf_counter(); api.blind_eval_host(ctx, pts, host); hs.append((time.perf_counter() - t0) * 1e3)
    print("timing %s: lone sum mean %.3f median %.3f min %.3f | S1 mean %.3f median %.3f min %.3f" % (timing, statistics.mean(ts), statistics.median(ts), min(ts), statistics.mean(hs), statistics.median(hs), min(hs)), flush=True)
