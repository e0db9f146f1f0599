"""Differential soak of the MSM against the CPU oracle: random lengths, groups, window sizes (forced, automatic, window
tables), slice lengths, tail (chains / trees of lane-cooperative additions / automatic), scalar distributions (uniform,
short, skewed, zeros, r - 1, repeated points); every fourth case a multi-sum over arrays of BOTH groups (one sort, one
plan) or a burst of sums in flight.
  python3 tools/fuzz_msm.py [seconds] [seed]        (GPU box; exits non-zero on the first mismatch)"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import coracle as co, pyref as pr  # noqa: E402
from playsnark_amd import api  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
ctx = api.Context(0)
t_end = time.time() + budget
cases = 0
while time.time() < t_end:
    group = rnd.choice(["G1", "G1", "G2"])
    og, gid = getattr(co, group), getattr(api, group)
    n = rnd.choice([1, 2, 3, 7, 63, 64, 65, 255, 257, 1000, 1023, 1025, 4097, rnd.randrange(1, 6000), rnd.randrange(1, 20000)])
    if group == "G2":
        n = min(n, 3000)
    raw = og.gen_points(rnd.randrange(1, pr.R), rnd.randrange(1, pr.R), n)
    if rnd.random() < 0.2 and n > 4:  # repeated points (P + P and P - P inside buckets)
        nb = og.nb
        k = rnd.randrange(1, n)
        raw = raw[: k * nb] + raw[: (n - k) * nb]
    dist = rnd.choice(["uniform", "uniform", "short", "skew", "bits", "edge"])
    if dist == "uniform":
        sc = [rnd.randrange(pr.R) for _ in range(n)]
    elif dist == "short":
        bits = rnd.choice([1, 8, 20, 33, 64, 100])
        sc = [rnd.randrange(1 << bits) for _ in range(n)]
    elif dist == "skew":
        pool = [rnd.randrange(pr.R) for _ in range(rnd.choice([1, 2, 5]))]
        sc = [rnd.choice(pool) for _ in range(n)]
    elif dist == "bits":
        sc = [rnd.randrange(2) for _ in range(n)]
    else:
        sc = [rnd.choice([0, 1, pr.R - 1, pr.R - 2, (pr.R - 1) // 2, 1 << 254]) % pr.R for _ in range(n)]
    want = og.to_b(og.msm_pippenger(co.pack_fr(sc), raw, n, 4))
    pts = api.Points.upload(ctx, gid, raw)
    mode = rnd.choice(["auto", "forced", "table", "table"])
    c = 0
    if mode == "forced":
        c = rnd.randrange(4, 21)
        ctx.set_window(c)
    elif mode == "table":
        c = rnd.choice([0, 8, 9, 10, 13, 16, 20, rnd.randrange(8, 23)])
        pts.precompute(c)
    m = rnd.choice([0, 0, 1, 2, 5, 8, 32, 100])
    if m:
        ctx.set_slice(m)
    tail = rnd.choice([0, 0, 1, 2, 2])
    ctx.set_tail(tail)
    try:
        poly = api.Poly.upload(ctx, sc)
        got = poly.BlindEval(pts)
        if rnd.random() < 0.3 and n > 10:  # an index-range view of both
            f = rnd.randrange(0, n // 2)
            cnt = rnd.randrange(1, n - f)
            nb = og.nb
            got2 = poly.slice(f, cnt).BlindEval(pts.slice(f, cnt))
            want2 = og.to_b(og.msm_pippenger(co.pack_fr(sc[f:f + cnt]), raw[f * nb:(f + cnt) * nb], cnt, 4))
            if got2 != want2:
                print("MISMATCH (slice)", dict(group=group, n=n, dist=dist, mode=mode, c=c, m=m, first=f, count=cnt, seed=seed, case=cases))
                sys.exit(1)
        if cases % 4 == 3 and n <= 3000:
            # the same scalars over arrays of both groups in ONE call (PHGR13's shape), or three sums in flight
            other_name = "G2" if group == "G1" else "G1"
            oo, ogid = getattr(co, other_name), getattr(api, other_name)
            raw_o = oo.gen_points(rnd.randrange(1, pr.R), rnd.randrange(1, pr.R), n)
            pts_o = api.Points.upload(ctx, ogid, raw_o)
            if mode == "table":
                pts_o.precompute(c)
            want_o = oo.to_b(oo.msm_pippenger(co.pack_fr(sc), raw_o, n, 4))
            if rnd.random() < 0.5:
                got_m = api.msm_multi(ctx, [pts, pts_o, pts], poly)
                if got_m != [want, want_o, want]:
                    print("MISMATCH (multi)", dict(group=group, n=n, dist=dist, mode=mode, c=c, m=m, tail=tail, seed=seed, case=cases))
                    sys.exit(1)
            else:
                burst = [(pts, gid, want), (pts_o, ogid, want_o), (pts, gid, want), (pts_o, ogid, want_o)][:rnd.choice([3, 4])]  # up to PS_MSM_QUEUE pending
                for pp, _g, _w in burst:
                    api.msm_launch(ctx, pp, poly)
                got_f = [api.msm_finish(ctx, gg) for _p, gg, _w in burst]
                if got_f != [w for _p, _g, w in burst]:
                    print("MISMATCH (in flight)", dict(group=group, n=n, dist=dist, mode=mode, c=c, m=m, tail=tail, seed=seed, case=cases))
                    sys.exit(1)
    finally:
        ctx.set_window(0)
        ctx.set_slice(0)
        ctx.set_tail(0)
    if got != want:
        print("MISMATCH", dict(group=group, n=n, dist=dist, mode=mode, c=c, m=m, tail=tail, seed=seed, case=cases, info=ctx.last_msm_info()))
        sys.exit(1)
    cases += 1
    if cases % 50 == 0:
        print(cases, "cases ok", flush=True)
print("fuzz ok:", cases, "cases, seed", seed)
