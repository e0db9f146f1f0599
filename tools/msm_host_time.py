"""Host-side cost of one MSM call: time to enqueue (launch), GPU wait, host fold.  Run on the GPU box."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from playsnark_amd import api
ctx = api.Context(0)
n = 1 << int(os.environ.get("LOG2N", "20"))
a = api.Poly.upload(ctx, bench.uniform_scalars_be32(n, 1).tobytes())
pts = api.Points.from_scalars(ctx, api.G1, a)
if os.environ.get("TABLE", "1") == "1":
    pts.precompute(0)
sc = api.Poly.upload(ctx, bench.uniform_scalars_be32(n, 2).tobytes())
for _ in range(2):
    api.msm_launch(ctx, pts, sc); api.msm_finish(ctx, api.G1)
tl = tf = 0
for _ in range(10):
    t0 = time.perf_counter(); api.msm_launch(ctx, pts, sc); t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter(); api.msm_finish(ctx, api.G1); t3 = time.perf_counter()
    tl += t1 - t0; tf += t3 - t2
    ts = t2 - t1
print("launch_ms", tl / 10 * 1e3, "gpu_wait_ms(last)", ts * 1e3, "host_fold_ms", tf / 10 * 1e3)
