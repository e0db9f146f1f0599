"""CPU, world_size 2, gloo: the multi-GPU exchange of the sharded MSM.  Each rank
owns an index-range shard, the per-rank partial sums are all_gathered and folded by the product's
ps_points_sum.  Without a GPU the local MSM of each rank is supplied by the oracle; the sharding,
the collective and the fold are the code under test."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from oracle import coracle as co, pyref as pr
from playsnark_amd import api
from playsnark_amd.dist import ShardedMsm, shard_range
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n = 301
rng = pr.SplitMix64(77)
sc = [rng.fr() for _ in range(n)]
for grp, gid in ((co.G1, api.G1), (co.G2, api.G2)):
    raw = grp.gen_points(123, 457, n)
    first, cnt = shard_range(n, rank, world)
    partial = grp.to_b(grp.msm_pippenger(co.pack_fr(sc[first:first + cnt]), raw[first * grp.nb:(first + cnt) * grp.nb], cnt, 1))
    total = ShardedMsm(None, gid, dist, world).combine(partial)
    want = grp.to_b(grp.msm_pippenger(co.pack_fr(sc), raw, n, 2))
    assert total == want, (rank, grp.name)
# the pipelined runner starts the exchange of sum i and collects it `depth` sums later (async all_gather): the local
# MSMs come from the oracle here, a different one per step, and every step's folded result must be that step's total
import playsnark_amd.api as api_mod
for depth in (1, 3):
    steps, launched, seen = 5, [], []
    raw = co.G1.gen_points(9, 11, n)
    vectors = [[rng.fr() for _ in range(n)] for _ in range(steps)]
    first, cnt = shard_range(n, rank, world)
    api_mod.msm_launch = lambda ctx, pts, sc: launched.append(len(launched))
    def finish(ctx, group, _state={"k": 0}):
        k = _state["k"] %% steps; _state["k"] += 1
        return co.G1.to_b(co.G1.msm_pippenger(co.pack_fr(vectors[k][first:first + cnt]), raw[first * 96:(first + cnt) * 96], cnt, 1))
    api_mod.msm_finish = finish
    sm = ShardedMsm(None, api.G1, dist, world)
    orig_finish = sm.combine_finish
    sm.combine_finish = lambda started: seen.append(orig_finish(started)) or seen[-1]
    last = sm.run_pipelined(None, None, steps, depth=depth)
    want = [co.G1.to_b(co.G1.msm_pippenger(co.pack_fr(v), raw, n, 2)) for v in vectors]
    assert seen == want and last == want[-1] and len(launched) == steps, (rank, depth)
covered = sorted(sum((list(range(*(lambda f, c: (f, f + c))(*shard_range(10, r, 3)))) for r in range(3)), []))
assert covered == list(range(10))
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_sharded_msm_exchange_two_ranks(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29617", str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert res.stdout.count("ok") == 2


def test_pipelined_runner_keeps_depth_and_order(monkeypatch):
    """Host logic of ShardedMsm.run_pipelined (the loop bench.py times): never more than `depth` sums
    pending, every launched sum is finished, results come back in launch order.  The GPU entry points
    are replaced by a recording FIFO; the queue itself is tested on the GPU (test_msm_gpu.py)."""
    from playsnark_amd import api
    from playsnark_amd.dist import ShardedMsm

    for depth in (1, 2, 3):
        for steps in (1, 2, 5):
            pending, log, counter = [], [], [0]

            def launch(ctx, points, scalars):
                assert len(pending) < depth
                counter[0] += 1
                pending.append(counter[0])
                log.append(("L", counter[0], len(pending)))

            def finish(ctx, group):
                k = pending.pop(0)
                log.append(("F", k, len(pending)))
                return k.to_bytes(4, "big")

            monkeypatch.setattr(api, "msm_launch", launch)
            monkeypatch.setattr(api, "msm_finish", finish)
            seen = []
            last = ShardedMsm(None, api.G1).run_pipelined(None, None, steps, on_step=lambda: seen.append(len(pending)), depth=depth)
            assert not pending and counter[0] == steps and len(seen) == steps
            assert [k for op, k, _ in log if op == "F"] == list(range(1, steps + 1))
            assert last == steps.to_bytes(4, "big")
            if steps >= depth:
                assert max(p for _, _, p in log) == depth


PHGR13_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from oracle import pyref as pr, restate as rs
from playsnark_amd import api
from playsnark_amd.dist import ShardedPHGR13
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ctx = api.Context(0)  # rehearsal: every rank on the one GPU of the box
rng = pr.SplitMix64(99)
c, sol = rs.synthetic_circuit(60)
c = rs.SparseR1CS(c.nbVars, c.nbVars - 3, c.left, c.right, c.out)
q = api.QAP(ctx, c.nbVars, c.nbIO, c.left, c.right, c.out)
ek, vk = api.NewPHGR13TrustedSetup(q, *[rng.fr() for _ in range(8)])
sol_dev = api.Poly.upload(ctx, sol)
whole = api.PHGR13Prove(ek, q, sol_dev)
got = ShardedPHGR13(ctx, dist, world, rank).prove(ek, q, sol_dev)
for f in api.PHGR13Proof.FIELDS:
    assert getattr(got, f) == getattr(whole, f), (rank, f)
from playsnark_amd.dist import ShardedGroth16
tr, _ = api.NewGroth16TrustedSetup(q, *[rng.fr() for _ in range(5)])
r, s = rng.fr(), rng.fr()
g_whole = api.Groth16Prove(tr, q, sol_dev, r, s)
g_got = ShardedGroth16(ctx, dist, world, rank).prove(tr, q, sol_dev, r, s)
assert (g_got.A, g_got.B, g_got.C) == (g_whole.A, g_whole.B, g_whole.C), rank
dist.destroy_process_group()
print("rank", rank, "ok")
"""


import pytest  # noqa: E402

LOCAL_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from oracle import pyref as pr, restate as rs
from playsnark_amd import api
from playsnark_amd.dist import ShardedGroth16Local, shard_range
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ctx = api.Context(0)  # rehearsal: every rank on the one GPU of the box
rng = pr.SplitMix64(1234)
c, sol = rs.synthetic_circuit(77)
c = rs.SparseR1CS(c.nbVars, c.nbVars - 3, c.left, c.right, c.out)
tox = [rng.fr() for _ in range(5)]
r, s = rng.fr(), rng.fr()
tr = rs.groth16_setup(c, *tox)
want = rs.groth16_prove(tr, c, sol, r, s)
def part(raw, total, nb, group):
    first, cnt = shard_range(total, rank, world)
    return api.Points.upload(ctx, group, raw[first * nb:(first + cnt) * nb])  # this rank uploads ITS range only
n, nn = c.nbGates, c.nbIO
key = api.Groth16Setup(tr.Alpha, tr.Beta, tr.Delta, tr.Beta2, tr.Delta2, part(tr.Xi, n, 96, api.G1), part(tr.Xi2, n, 192, api.G2),
                       part(tr.NioLP, nn, 96, api.G1), part(tr.XiT, n - 1, 96, api.G1))
q = api.QAP(ctx, c.nbVars, c.nbIO, c.left, c.right, c.out)
got = ShardedGroth16Local(ctx, dist, world, rank).prove(key, q, api.Poly.upload(ctx, sol), r, s)
assert (got.A, got.B, got.C) == (want.A, want.B, want.C), rank
bad = list(sol); bad[7] = (bad[7] + 1) %% pr.R
try:
    ShardedGroth16Local(ctx, dist, world, rank).prove(key, q, api.Poly.upload(ctx, bad), r, s)
    raise SystemExit("no apocalypse on rank %%d" %% rank)
except api.Apocalypse:
    pass
dist.destroy_process_group()
print("rank", rank, "ok")
"""


@pytest.mark.gpu
def test_groth16_three_ranks_with_local_keys_and_split_quotient(tmp_path):
    """Three processes (gloo, all on GPU 0): each uploads only its index ranges of the CRS; rank 0 interpolates A, rank 1 B,
    rank 2 computes h, three broadcasts hand them round; the folded proof equals the oracle's; a bad witness raises
    Apocalypse on every rank."""
    script = tmp_path / "worker.py"
    script.write_text(LOCAL_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
           "--master-port", "29619", str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert res.stdout.count("ok") == 3


@pytest.mark.gpu
def test_sharded_provers_two_ranks_on_one_gpu(tmp_path):
    """Two processes (gloo for the exchange, both on GPU 0): each proves its index ranges of a PHGR13 and
    of a Groth16 proof, the all_gather + fold gives every rank the unsharded proofs."""
    script = tmp_path / "worker.py"
    script.write_text(PHGR13_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29618", str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert res.stdout.count("ok") == 2


def test_sharded_phgr13_fold_on_the_host():
    """ShardedPHGR13.fold adds the ranks' partial proofs element by element (host code only:
    ps_points_sum); checked against the oracle's point additions, identity partials included."""
    from oracle import coracle as co, pyref as pr
    from playsnark_amd import api
    from playsnark_amd.dist import ShardedPHGR13

    rng = pr.SplitMix64(4711)
    parts, want = [], {}
    for g in range(3):
        p = {}
        for f in api.PHGR13Proof.FIELDS:
            grp = co.G2 if f == "wss" else co.G1
            pt = None if (g == 1 and f in ("hs", "wss")) else grp.mul(rng.fr())  # rank 1 had empty ranges there
            p[f] = grp.to_b(pt)
            want[f] = pt if f not in want else grp.add(want[f], pt)
        parts.append(p)
    folded = ShardedPHGR13.fold(parts)
    for f in api.PHGR13Proof.FIELDS:
        grp = co.G2 if f == "wss" else co.G1
        assert getattr(folded, f) == grp.to_b(want[f]), f
