"""First contact with RCCL on the one GPU a box has.  Started by tests/test_rccl_gpu.py as ONE rank of
torch.distributed.run with backend "nccl" (= RCCL on ROCm).  In this ONE process:
  * torch's bundled HIP runtime + an RCCL communicator (built by the all_reduce probe of bench.py),
  * libplaysnark_hip.so, linked to /opt/rocm's libamdhip64 -- a second HIP runtime next to torch's,
  * the exchange code of playsnark_amd/dist.py exactly as eight ranks run it: async all_gather_into_tensor of the
    pipelined MSM runner, the provers' blocking all_gather, the broadcast of the split quotient (ALWAYS_EXCHANGE makes
    a one-rank group run the collectives instead of skipping them),
every result compared with the oracle.  What this cannot show: xGMI traffic, scaling, N > 1 rendezvous."""
import datetime
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert world == 1 and torch.cuda.is_available()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=120))
probe = torch.ones(4, dtype=torch.uint8, device=dev)
dist.all_reduce(probe)  # the communicator is built here
assert dist.get_backend() == "nccl" and probe.cpu().tolist() == [1, 1, 1, 1]
print("RCCL communicator up:", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "?", flush=True)

from oracle import coracle as co, pyref as pr, restate as rs  # the checker
from playsnark_amd import api, dist as psd

psd.ALWAYS_EXCHANGE = True
ctx = api.Context(0)
rng = pr.SplitMix64(0x52434C)

# ---- ShardedMsm.run / run_pipelined over RCCL, both groups ----
for og, gid, n in ((co.G1, api.G1, 3000), (co.G2, api.G2, 700)):
    raw = og.gen_points(rng.fr(), rng.fr(), n)
    pts = api.Points.upload(ctx, gid, raw)
    sc = [rng.fr() for _ in range(n)]
    want = og.to_b(og.msm_pippenger(co.pack_fr(sc), raw, n, 4))
    poly = api.Poly.upload(ctx, sc)
    m = psd.ShardedMsm(ctx, gid, dist, world)
    assert m.run(pts, poly) == want, og.name
    seen = []
    orig = m.combine_finish
    m.combine_finish = lambda started: seen.append(orig(started)) or seen[-1]
    assert m.run_pipelined(pts, poly, 7, depth=3) == want and seen == [want] * 7, og.name
    # the exchanged tensor really went through the GPU: a device tensor of the partial's bytes
    started = psd.ShardedMsm(ctx, gid, dist, world).combine_start(want)
    assert started[1].is_cuda and started[2].is_cuda
    started[0].wait()
print("sharded MSM over RCCL ok", flush=True)

# ---- ShardedGroth16 / ShardedGroth16Local / ShardedPHGR13 ----
c, sol = rs.synthetic_circuit(64)
tw = [rng.fr() for _ in range(5)]
r, s = rng.fr(), rng.fr()
tr_o = rs.groth16_setup(c, *tw)
want = rs.groth16_prove(tr_o, c, [pr.fr(v) for v in sol], r, s, fast=True)
q = api.QAP(ctx, c.nbVars, c.nbIO, c.left, c.right, c.out)
dsol = api.Poly.upload(ctx, sol)
tr, _vk = api.NewGroth16TrustedSetup(q, *tw)
for key in (tr, tr.monomial_only()):
    got = psd.ShardedGroth16(ctx, dist, world, rank).prove(key, q, dsol, r, s)
    assert (got.A, got.B, got.C) == (want.A, want.B, want.C)
got = psd.ShardedGroth16Local(ctx, dist, world, rank).prove(tr.monomial_only(), q, dsol, r, s)
assert (got.A, got.B, got.C) == (want.A, want.B, want.C)
tw8 = [rng.fr() for _ in range(8)]
want_p = rs.phgr13_prove(rs.phgr13_setup(c, *tw8).EK, c, [pr.fr(v) for v in sol], fast=True)
ek, _pvk = api.NewPHGR13TrustedSetup(q, *tw8)
got_p = psd.ShardedPHGR13(ctx, dist, world, rank).prove(ek, q, dsol)
for f in api.PHGR13Proof.FIELDS:
    assert getattr(got_p, f) == getattr(want_p, f), f
print("sharded provers over RCCL ok", flush=True)

# a GPU tensor torch made and the library's kernels in one process, after the collectives: both runtimes still work
x = torch.arange(1 << 20, device=dev, dtype=torch.int64).sum().item()
assert x == (1 << 20) * ((1 << 20) - 1) // 2
assert api.Poly.upload(ctx, [5]).BlindEval(api.Points.upload(ctx, api.G1, co.G1.gen_points(1, 0, 1))) == co.G1.to_b(co.G1.mul(5))
dist.barrier()
dist.destroy_process_group()
print("rccl one rank ok", flush=True)
