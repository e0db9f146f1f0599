"""Multi-GPU sharding of the MSM: one process per GPU, `torch.distributed` for the exchange
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

sum_i k_i*P_i splits by index range: rank g owns (P_i, k_i) for
i in shard_range(n, g, G), runs a complete local Pippenger on its GPU and produces ONE affine
partial sum (96 B G1 / 192 B G2).  The only data-path collective is an all_gather of those
G partials followed by a local G-way point addition -- elliptic-curve addition is not an
ncclRedOp, so a true all_reduce does not exist, and shipping buckets instead of the reduced
partial would turn a latency-bound 96-byte exchange into a per-link-bound ring for no gain.
"""
from __future__ import annotations

from . import api

_WIRE = {api.G1: 96, api.G2: 192}


def shard_range(n: int, rank: int, world: int):
    """[first, first+count) owned by `rank`: contiguous, sizes differ by at most one."""
    base, extra = divmod(n, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


class ShardedMsm:
    def __init__(self, ctx, group: int, dist=None, world: int = 1):
        self.ctx, self.group, self.dist, self.world = ctx, group, dist, world

    def combine(self, partial: bytes) -> bytes:
        """all_gather the per-rank partial sums and fold them (every rank gets the result)."""
        if self.dist is None or self.world == 1:
            return partial
        import torch

        nb = _WIRE[self.group]
        backend = self.dist.get_backend()
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        mine = torch.frombuffer(bytearray(partial), dtype=torch.uint8).to(dev)
        gathered = torch.empty(self.world * nb, dtype=torch.uint8, device=dev)
        self.dist.all_gather_into_tensor(gathered, mine)
        return api.points_sum(self.group, gathered.cpu().numpy().tobytes())

    def run(self, points: "api.Points", scalars: "api.Poly") -> bytes:
        """Local MSM over this rank's shard, then the exchange."""
        api.msm_launch(self.ctx, points, scalars)
        return self.combine(api.msm_finish(self.ctx, self.group))

    def run_pipelined(self, points: "api.Points", scalars: "api.Poly", steps: int, on_step=None, depth: int = 2) -> bytes:
        """`steps` sums with up to `depth` in flight (ps_msm_launch ... / ps_msm_finish FIFO): later sums
        are enqueued before the oldest is folded and exchanged, so their sort and accumulation run
        beside its latency-bound tail.  Every sum is completed and combined; returns the last result."""
        result = None
        launched = finished = 0
        while finished < steps:
            while launched < steps and launched - finished < depth:
                api.msm_launch(self.ctx, points, scalars)
                launched += 1
            result = self.combine(api.msm_finish(self.ctx, self.group))
            finished += 1
            if on_step:
                on_step()
        return result
