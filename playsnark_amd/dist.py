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

# A one-rank group normally skips the exchange (its own partial is the result).  True runs the collective anyway -- the
# one-rank RCCL first-contact test (tests/rccl_one_rank.py) uses it to put an RCCL communicator, the library's own HIP
# runtime and the async all_gather_into_tensor of the pipelined runner into ONE process on the one GPU a box has.
ALWAYS_EXCHANGE = False


def shard_range(n: int, rank: int, world: int):
    """[first, first+count) owned by `rank`: contiguous, sizes differ by at most one."""
    base, extra = divmod(n, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


class ShardedMsm:
    def __init__(self, ctx, group: int, dist=None, world: int = 1):
        self.ctx, self.group, self.dist, self.world = ctx, group, dist, world

    def combine_start(self, partial: bytes):
        """Start the all_gather of the per-rank partial sums; combine_finish folds them.  The two halves let the pipelined
        runner keep the exchange of sum i in flight while sums i+1.. are folded: on the GPU box the RCCL kernel of a
        96-byte all_gather has to find a free CU among saturated ones, and a blocking collective per step would put that
        wait (milliseconds under load) on every step's critical path."""
        if self.dist is None or (self.world == 1 and not ALWAYS_EXCHANGE):
            return partial
        import torch

        nb = _WIRE[self.group]
        backend = self.dist.get_backend()
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        mine = torch.frombuffer(bytearray(partial), dtype=torch.uint8).to(dev)
        gathered = torch.empty(self.world * nb, dtype=torch.uint8, device=dev)
        work = self.dist.all_gather_into_tensor(gathered, mine, async_op=True)
        return (work, gathered, mine)

    def combine_finish(self, started) -> bytes:
        if isinstance(started, (bytes, bytearray)):
            return bytes(started)
        work, gathered, _mine = started
        work.wait()
        return api.points_sum(self.group, gathered.cpu().numpy().tobytes())

    def combine(self, partial: bytes) -> bytes:
        """all_gather the per-rank partial sums and fold them (every rank gets the result)."""
        return self.combine_finish(self.combine_start(partial))

    def run(self, points: "api.Points", scalars: "api.Poly") -> bytes:
        """Local MSM over this rank's shard, then the exchange."""
        api.msm_launch(self.ctx, points, scalars)
        return self.combine(api.msm_finish(self.ctx, self.group))

    def run_pipelined(self, points: "api.Points", scalars: "api.Poly", steps: int, on_step=None, depth: int = 2) -> bytes:
        """`steps` sums with up to `depth` in flight (ps_msm_launch ... / ps_msm_finish FIFO): later sums
        are enqueued before the oldest is folded and exchanged, so their sort and accumulation run
        beside its latency-bound tail; the exchange of a sum is started when its local result is folded and collected
        `depth` sums later.  Every sum is completed and combined; returns the last result."""
        result = None
        launched = finished = 0
        exchanges = []
        while finished < steps:
            while launched < steps and launched - finished < depth:
                api.msm_launch(self.ctx, points, scalars)
                launched += 1
            exchanges.append(self.combine_start(api.msm_finish(self.ctx, self.group)))
            finished += 1
            if len(exchanges) > depth:
                result = self.combine_finish(exchanges.pop(0))
            if on_step:
                on_step()
        while exchanges:
            result = self.combine_finish(exchanges.pop(0))
        return result


class ShardedPHGR13:
    """PHGR13Prove (pinochio.go:207-254) with its sums sharded over the ranks (BASELINE config #5).

    Every sum of the proof is linear in (points, scalars), so rank g takes the index range
    shard_range(len, g, G) of each evaluation-key array with the matching range of the scalars:
        hs                     h[range] over gsi[range]                      (pinochio.go:218)
        vss .. yass, gz        solution[diff:][range] over the nine arrays   (pinochio.go:231-242),
                               one digit sort for all of them (ps_msm_multi)
    and produces nine partial points; one all_gather of 8 x 96 + 192 bytes per rank and a local fold
    give every rank the proof.  The quotient h is not sharded (its NTTs would need an all-to-all): each
    rank computes it from the full witness -- deterministic, so identical everywhere.
    """

    G1_FIELDS = ("vss", "yss", "vass", "wass", "yass", "hs", "gz")

    def __init__(self, ctx, dist=None, world: int = 1, rank: int = 0, local: bool = False):
        """local = True: the evaluation key holds only THIS rank's index ranges (a rank uploads its shard of the CRS and
        nothing else); False: every rank holds the whole key and takes views."""
        self.ctx, self.dist, self.world, self.rank, self.local = ctx, dist, world, rank, local

    def partials(self, ek: "api.PHGR13EvalKey", qap: "api.QAP", solution: "api.Poly", rank=None) -> dict:
        """This rank's share of every proof element (affine bytes; the identity for an empty range)."""
        rank = self.rank if rank is None else rank
        ctx = self.ctx
        # The index-range split needs every array to match its scalars, as BlindEval does (algebra.go:350-352):
        # a longer array would be silently truncated by the slices below, where the unsharded prover panics.
        diff = qap.nbVars - qap.nbIO
        names = ("vs", "ws", "ys", "vas", "was", "yas", "vbs", "wbs", "ybs")
        if self.local:  # the arrays ARE the rank's ranges: their lengths must be exactly those
            nn = qap.nbIO  # len(ek.vs) of the whole key: the reference's non-IO count nbVars - diff (pinochio.go:122-125)
            want_gsi, want = shard_range(qap.nbGates - 1, rank, self.world)[1], shard_range(nn, rank, self.world)[1]
            if len(ek.gsi) != want_gsi or any(len(getattr(ek, f)) != want for f in names):
                raise api.LengthMismatch(f"rank {rank} of {self.world} does not hold its index range of the evaluation key")
            take = lambda arr, first, cnt: arr
        else:
            nn = len(ek.vs)
            for f in names:
                if len(getattr(ek, f)) != nn:
                    raise api.LengthMismatch(f"evaluation-key arrays of different lengths: vs {nn}, {f} {len(getattr(ek, f))}")
            if len(ek.gsi) != qap.nbGates - 1:
                raise api.LengthMismatch(f"mismatch of length between poly {qap.nbGates - 1} and blinded eval points {len(ek.gsi)}")
            take = lambda arr, first, cnt: arr.slice(first, cnt)
        if diff + nn > len(solution):
            raise api.LengthMismatch("evaluation-key array longer than the non-IO part of the solution")
        h = qap.Quotient(solution)  # raises Apocalypse exactly as the unsharded prover
        first, cnt = shard_range(len(h), rank, self.world)
        out = {"hs": h.slice(first, cnt).BlindEval(take(ek.gsi, first, cnt))}
        first, cnt = shard_range(nn, rank, self.world)
        arrays = [take(getattr(ek, f), first, cnt) for f in names]
        sums = api.msm_multi(ctx, arrays, solution.slice(diff + first, cnt))
        by = dict(zip(names, sums))
        out.update(vss=by["vs"], wss=by["ws"], yss=by["ys"], vass=by["vas"], wass=by["was"], yass=by["yas"])
        out["gz"] = api.points_sum(api.G1, by["vbs"] + by["wbs"] + by["ybs"])  # pinochio.go:242
        return out

    @staticmethod
    def fold(parts: list) -> "api.PHGR13Proof":
        """Element-wise sum of the ranks' partial proofs."""
        raw = api._lib.Phgr13Proof()
        for f in api.PHGR13Proof.FIELDS:
            grp = api.G2 if f == "wss" else api.G1
            total = api.points_sum(grp, b"".join(p[f] for p in parts))
            api.C.memmove(getattr(raw, f), total, len(total))
        return api.PHGR13Proof(raw)

    def prove(self, ek, qap, solution) -> "api.PHGR13Proof":
        mine = self.partials(ek, qap, solution)
        if self.dist is None or (self.world == 1 and not ALWAYS_EXCHANGE):
            return self.fold([mine])
        import torch

        order = api.PHGR13Proof.FIELDS
        blob = b"".join(mine[f] for f in order)
        backend = self.dist.get_backend()
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
        gathered = torch.empty(self.world * len(blob), dtype=torch.uint8, device=dev)
        self.dist.all_gather_into_tensor(gathered, t)
        flat = gathered.cpu().numpy().tobytes()
        parts = []
        for g in range(self.world):
            chunk, off, p = flat[g * len(blob):(g + 1) * len(blob)], 0, {}
            for f in order:
                nb = 192 if f == "wss" else 96
                p[f] = chunk[off:off + nb]
                off += nb
            parts.append(p)
        return self.fold(parts)


class ShardedGroth16:
    """Groth16Prove (groth16.go:122-211) with its three sums sharded over the ranks: ps_groth16_prove_shard
    gives this rank's partial A, B, C (its index ranges of Xi, Xi2, NioLP, XiT; rank 0 also the fixed
    points); one all_gather of 96 + 192 + 96 bytes per rank and a local fold give every rank the proof."""

    def __init__(self, ctx, dist=None, world: int = 1, rank: int = 0):
        self.ctx, self.dist, self.world, self.rank = ctx, dist, world, rank

    def partials(self, tr: "api.Groth16Setup", q: "api.QAP", sol: "api.Poly", r: int, s: int, rank=None):
        rank = self.rank if rank is None else rank
        A, B, Cc = api.C.create_string_buffer(96), api.C.create_string_buffer(192), api.C.create_string_buffer(96)
        pk = tr._struct()
        api._check(api.lib.ps_groth16_prove_shard(q.ctx._h, api.C.byref(pk), q._h, sol._h, api._be32(r), api._be32(s), rank,
                                                  self.world, A, B, Cc))
        return A.raw, B.raw, Cc.raw

    @staticmethod
    def fold(parts: list, r: int, s: int) -> "api.Groth16Proof":
        A = api.points_sum(api.G1, b"".join(p[0] for p in parts))
        B = api.points_sum(api.G2, b"".join(p[1] for p in parts))
        Cc = api.points_sum(api.G1, b"".join(p[2] for p in parts))
        return api.Groth16Proof(r, s, A, B, Cc)

    def prove(self, tr, q, sol, r: int, s: int) -> "api.Groth16Proof":
        mine = self.partials(tr, q, sol, r, s)
        if self.dist is None or (self.world == 1 and not ALWAYS_EXCHANGE):
            return self.fold([mine], r, s)
        import torch

        blob = b"".join(mine)
        backend = self.dist.get_backend()
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
        gathered = torch.empty(self.world * len(blob), dtype=torch.uint8, device=dev)
        self.dist.all_gather_into_tensor(gathered, t)
        flat = gathered.cpu().numpy().tobytes()
        parts = [(flat[g * 384:g * 384 + 96], flat[g * 384 + 96:g * 384 + 288], flat[g * 384 + 288:(g + 1) * 384])
                 for g in range(self.world)]
        return self.fold(parts, r, s)


def _broadcast_bytes(dist, payload, nbytes: int, src: int) -> bytes:
    """One vector from its owner to every rank (RCCL broadcast over xGMI on the GPU box, gloo in the CPU rehearsals)."""
    import torch

    backend = dist.get_backend()
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    if dist.get_rank() == src:
        t = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(dev)
    else:
        t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=src)
    return t.cpu().numpy().tobytes()


class ShardedGroth16Local:
    """Groth16Prove (groth16.go:122-211) over ranks that hold ONLY their index ranges of the CRS arrays: the key a rank
    passes in has Xi[range(n)], Xi2[range(n)], NioLP[range(nbIO)], XiT[range(n-1)] (shard_range) -- an eighth of the
    upload and of the HBM per rank at eight GPUs -- and the fixed points, which rank 0 uses.

    The three parts of the quotient are independent until the exchange: with three or more ranks,
    rank 0 interpolates A, rank 1 interpolates B, rank 2 computes h by the h-only route (which carries the divisibility
    test: Apocalypse), side by side; three broadcasts of 32 n bytes hand the vectors round and every rank keeps its
    ranges.  With fewer ranks each computes all three itself.  Sums per rank: B.Xi2 (G2), A.Xi, B.Xi, sol.NioLP, h.XiT over
    its ranges; C's share is N + H + s (A.Xi) + r (B.Xi) (two host scalar multiplications), rank 0 adds the fixed points;
    one all_gather of 96 + 192 + 96 bytes and a local fold give every rank the proof -- the bytes of the unsharded prover."""

    def __init__(self, ctx, dist=None, world: int = 1, rank: int = 0):
        self.ctx, self.dist, self.world, self.rank = ctx, dist, world, rank

    def quotient_ranges(self, q: "api.QAP", sol: "api.Poly", rank=None):
        """(A[range(n)], B[range(n)], h[range(n-1)]) of this rank as device vectors."""
        rank = self.rank if rank is None else rank
        n = q.nbGates
        fa, ca = shard_range(n, rank, self.world)
        fh, ch = shard_range(n - 1, rank, self.world)
        if self.dist is not None and self.world >= 3:
            mine = None
            if rank == 0:
                mine = q.interpolate(sol, 0).download_bytes()
            elif rank == 1:
                mine = q.interpolate(sol, 1).download_bytes()
            elif rank == 2:
                try:
                    mine = q.Quotient(sol).download_bytes()
                except api.Apocalypse:
                    mine = b"\xff" * (32 * (n - 1))  # not a field element: every rank sees the failure after the broadcast
            A = _broadcast_bytes(self.dist, mine, 32 * n, 0)
            B = _broadcast_bytes(self.dist, mine, 32 * n, 1)
            h = _broadcast_bytes(self.dist, mine, 32 * (n - 1), 2)
            if h[:32] == b"\xff" * 32:
                raise api.Apocalypse("apocalypse")
            up = lambda raw, first, cnt: api.Poly.upload(self.ctx, raw[32 * first:32 * (first + cnt)])
            return up(A, fa, ca), up(B, fa, ca), up(h, fh, ch)
        A, B, h = q.computeAB(sol)
        return A.slice(fa, ca), B.slice(fa, ca), h.slice(fh, ch)

    def partials(self, tr_local: "api.Groth16Setup", q: "api.QAP", sol: "api.Poly", r: int, s: int, rank=None):
        rank = self.rank if rank is None else rank
        ctx = self.ctx
        n, nn, diff = q.nbGates, q.nbIO, q.nbVars - q.nbIO  # len(NioLP) = nbVars - diff = nbIO (groth16.go:86-91)
        fq, cq = shard_range(nn, rank, self.world)
        want = {"Xi": shard_range(n, rank, self.world)[1], "Xi2": shard_range(n, rank, self.world)[1], "NioLP": cq,
                "XiT": shard_range(n - 1, rank, self.world)[1]}
        for name, cnt in want.items():
            if len(getattr(tr_local, name)) != cnt:
                raise api.LengthMismatch(f"rank {rank} of {self.world} does not hold its index range of {name}")
        A, B, h = self.quotient_ranges(q, sol, rank)
        sol_nio = sol.slice(diff + fq, cq)
        api.msm_launch(ctx, tr_local.Xi2, B)   # the longest point pass first
        api.msm_launch(ctx, tr_local.Xi, A)
        api.msm_launch(ctx, tr_local.Xi, B)
        pB2 = api.msm_finish(ctx, api.G2)
        api.msm_launch(ctx, tr_local.NioLP, sol_nio)
        pA = api.msm_finish(ctx, api.G1)
        api.msm_launch(ctx, tr_local.XiT, h)
        pB1, pN, pH = (api.msm_finish(ctx, api.G1) for _ in range(3))
        pC = api.points_sum(api.G1, pN + pH + api.points_lincomb(api.G1, pA + pB1, [s, r]))  # groth16.go:189-197
        if rank == 0:  # r Delta + Alpha ; s Delta2 + Beta2 ; rs Delta + s Alpha + r Beta (expanded from :149-200)
            pA = api.points_sum(api.G1, pA + api.points_lincomb(api.G1, tr_local.Delta + tr_local.Alpha, [r, 1]))
            pB2 = api.points_sum(api.G2, pB2 + api.points_lincomb(api.G2, tr_local.Delta2 + tr_local.Beta2, [s, 1]))
            rs_delta = api.points_lincomb(api.G1, api.points_lincomb(api.G1, tr_local.Delta, [s]), [r])
            pC = api.points_sum(api.G1, pC + rs_delta + api.points_lincomb(api.G1, tr_local.Alpha + tr_local.Beta, [s, r]))
        return pA, pB2, pC

    def prove(self, tr_local, q, sol, r: int, s: int) -> "api.Groth16Proof":
        mine = self.partials(tr_local, q, sol, r, s)
        if self.dist is None or (self.world == 1 and not ALWAYS_EXCHANGE):
            return ShardedGroth16.fold([mine], r, s)
        import torch

        blob = b"".join(mine)
        backend = self.dist.get_backend()
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
        gathered = torch.empty(self.world * len(blob), dtype=torch.uint8, device=dev)
        self.dist.all_gather_into_tensor(gathered, t)
        flat = gathered.cpu().numpy().tobytes()
        parts = [(flat[g * 384:g * 384 + 96], flat[g * 384 + 96:g * 384 + 288], flat[g * 384 + 288:(g + 1) * 384])
                 for g in range(self.world)]
        return ShardedGroth16.fold(parts, r, s)
