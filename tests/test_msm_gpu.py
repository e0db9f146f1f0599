"""GPU parity: the HIP MSM (through the C ABI) against the oracle, bit-exact on the affine bytes.

Reference behaviour under test: Poly.BlindEval (algebra.go:348-359), its int64-scalar twins
(groth16.go:176-178, pinochio.go:222-229) and the length-mismatch panic (algebra.go:350-352).
Method of the reference's own tests (algebra_test.go:21-35): compare against an independently
computed group element.
"""
import os

import pytest

pytestmark = pytest.mark.gpu

SEED = 0x706C6179736E61726B & 0xFFFFFFFFFFFFFFFF


def _report_ms(what, ms, budget_ms):
    """Wall-clock figures of the parity tests are REPORTED (a warning when over budget), never asserted: a slow or shared
    box must not turn a correct run red and hide the tests behind it under `pytest -x`."""
    import warnings

    line = f"{what}: {ms:.2f} ms (budget {budget_ms} ms)"
    print("[perf]", line)
    if ms > budget_ms:
        warnings.warn("over budget: " + line)



def _rng(pr, salt=0):
    return pr.SplitMix64(SEED + salt)


def _grp(ps_api, co, name):
    return (ps_api.G1, co.G1) if name == "g1" else (ps_api.G2, co.G2)


@pytest.mark.parametrize("name", ["g1", "g2"])
def test_fixed_base_matches_oracle(ps_api, ctx, co, pr, name):
    """Point.Mul(s, nil) = s*G (curve.go:25-31, algebra.go:373) on the device."""
    gid, og = _grp(ps_api, co, name)
    rng = _rng(pr, 1)
    ks = [0, 1, 2, 3, 255, 256, pr.R - 1, pr.R - 2, 1 << 254] + [rng.fr() for _ in range(23)]
    pts = ps_api.Points.from_scalars(ctx, gid, ps_api.Poly.upload(ctx, ks))
    got = og.unpack(pts.download())
    for k, g in zip(ks, got):
        assert g == og.mul(k), hex(k)


@pytest.mark.parametrize("name", ["g1", "g2"])
def test_points_roundtrip_and_validation(ps_api, ctx, co, pr, name):
    gid, og = _grp(ps_api, co, name)
    raw = og.gen_points(12345, 6789, 9) + og.to_b(None)
    pts = ps_api.Points.upload(ctx, gid, raw)
    assert len(pts) == 10 and pts.download() == raw
    assert pts.slice(3, 4).download() == raw[3 * og.nb : 7 * og.nb]
    bad = bytearray(raw)
    bad[og.nb - 1] ^= 1  # y of point 0 off the curve
    with pytest.raises(ps_api.PlaysnarkError) as e:
        ps_api.Points.upload(ctx, gid, bytes(bad))
    assert e.value.code == -3


@pytest.mark.parametrize("name", ["g1", "g2"])
@pytest.mark.parametrize("n", [0, 1, 2, 3, 5, 64, 333, 1024])
def test_blind_eval_small_vs_reference_loop(ps_api, ctx, co, pr, name, n):
    """TestAlgebraBlindEval / TestPinocchioCombine shape: GPU MSM == serial Mul+Add loop."""
    gid, og = _grp(ps_api, co, name)
    rng = _rng(pr, 100 + n)
    sc = [rng.fr() for _ in range(n)]
    raw = og.gen_points(rng.fr(), rng.fr(), n)
    got = ps_api.Poly.upload(ctx, sc).BlindEval(ps_api.Points.upload(ctx, gid, raw))
    want = og.to_b(og.blind_eval(sc, raw)) if n <= 64 else og.to_b(og.msm_pippenger(co.pack_fr(sc), raw, n, 4))
    assert got == want


@pytest.mark.parametrize("name", ["g1", "g2"])
def test_blind_eval_edge_scalars_and_points(ps_api, ctx, co, pr, name):
    """Zero / one / r-1 scalars, identity points, repeated points (forces the doubling branch
    of the bucket adder) and P, -P pairs (forces the cancellation branch)."""
    gid, og = _grp(ps_api, co, name)
    rng = _rng(pr, 7)
    base = og.unpack(og.gen_points(rng.fr(), rng.fr(), 6))
    P, Q = base[0], base[1]
    negP = (P[0], (pr.P - P[1]) % pr.P) if name == "g1" else (P[0], ((-P[1][0]) % pr.P, (-P[1][1]) % pr.P))
    k = rng.fr()
    cases = [
        ([0, 0, 0], [P, Q, base[2]]),
        ([1, 1, 1], [P, Q, base[2]]),
        ([pr.R - 1, pr.R - 1], [P, Q]),
        ([k, k, k, k, k], [P, P, P, P, P]),              # same bucket, same point: doubling
        ([k, k], [P, negP]),                              # same bucket, opposite points: identity
        ([k, pr.R - k], [P, P]),                          # k*P + (-k)*P = O
        ([k, 5, 7], [None, P, None]),                     # identity inputs
        ([3, 3, 3, 3, 3, 3], base),
        ([(1 << 255) % pr.R, (1 << 128) - 1, 1 << 64], base[:3]),
    ]
    for sc, pts in cases:
        raw = og.pack(pts)
        got = ps_api.Poly.upload(ctx, sc).BlindEval(ps_api.Points.upload(ctx, gid, raw))
        assert got == og.to_b(og.blind_eval(sc, raw)), (sc, name)


def test_blind_eval_length_mismatch_is_the_reference_panic(ps_api, ctx, co):
    raw = co.G1.gen_points(1, 1, 4)
    pts = ps_api.Points.upload(ctx, ps_api.G1, raw)
    with pytest.raises(ps_api.LengthMismatch) as e:
        ps_api.Poly.upload(ctx, [1, 2, 3]).BlindEval(pts)
    assert "mismatch of length between poly 3 and blinded eval points 4" in str(e.value)


@pytest.mark.parametrize("name", ["g1", "g2"])
def test_blind_eval_int64_witness_scalars(ps_api, ctx, co, pr, name):
    """computeSolCommit / NioLP loop: Value.ToFieldElement scalars incl. negatives, zeros, ones."""
    gid, og = _grp(ps_api, co, name)
    rng = _rng(pr, 9)
    n = 200
    vals = [0, 1, 1, 0, -1, 35, 9, 27, 30, -(1 << 62), (1 << 63) - 1] + [
        int(rng.next() % 1000) - 100 for _ in range(n - 11)
    ]
    raw = og.gen_points(rng.fr(), rng.fr(), n)
    pts = ps_api.Points.upload(ctx, gid, raw)
    got = ps_api.Poly.from_values(ctx, vals).BlindEval(pts)
    assert got == og.to_b(og.blind_eval_i64(vals, raw))
    # negatives are folded onto the negated point: the vector keeps the short-scalar plan
    assert ctx.last_msm_info()["windows"] * ctx.last_msm_info()["window_bits"] < 128
    extremes = [-(1 << 63), (1 << 63) - 1, -1, 1, 0, -(1 << 63) + 1] + [-(int(rng.next() % (1 << 62))) for _ in range(n - 6)]
    got = ps_api.Poly.from_values(ctx, extremes).BlindEval(pts)
    assert got == og.to_b(og.blind_eval_i64(extremes, raw))
    sl = ps_api.Poly.from_values(ctx, vals).slice(3, 50)  # a view keeps the folding
    assert sl.BlindEval(pts.slice(3, 50)) == og.to_b(og.blind_eval_i64(vals[3:53], raw[3 * og.nb : 53 * og.nb]))
    # all non-negative: the short-scalar plan (fewer windows) must give the same point
    pos = [abs(v) for v in vals]
    got = ps_api.Poly.from_values(ctx, pos).BlindEval(pts)
    assert got == og.to_b(og.blind_eval_i64(pos, raw))
    assert ctx.last_msm_info()["windows"] * ctx.last_msm_info()["window_bits"] < 128


@pytest.mark.parametrize("c,group", [(c, "G1") for c in (4, 5, 6, 7, 8, 9, 11, 13, 16, 18, 20)] + [(c, "G2") for c in (4, 7, 10, 16, 19)])
def test_window_size_does_not_change_the_result(ps_api, ctx, co, pr, c, group):
    """Forced window sizes; 18 and up give more than 2^20 buckets, which takes the one-level sort with returning
    atomics (k_digits_grouped / k_scatter) instead of the two-level counting sort.  The sizes also walk the shapes of the
    bucket reduction: fewer than 8 segments per set (c <= 6), one group of segments (7), bit jobs (8 and up), several
    workgroups per job and k_reduce_fin (18 and up)."""
    rng = _rng(pr, 11)
    n = 777 if group == "G1" else 200
    og = getattr(co, group)
    sc = [rng.fr() for _ in range(n)]
    raw = og.gen_points(rng.fr(), rng.fr(), n)
    pts = ps_api.Points.upload(ctx, getattr(ps_api, group), raw)
    want = og.to_b(og.msm_pippenger(co.pack_fr(sc), raw, n, 4))
    try:
        ctx.set_window(c)
        assert ps_api.Poly.upload(ctx, sc).BlindEval(pts) == want
        assert ctx.last_msm_info()["window_bits"] == c
    finally:
        ctx.set_window(0)


def test_skewed_scalars_heavy_buckets(ps_api, ctx, co, pr):
    """All-equal scalars put every point of a window into one bucket (SURVEY 8d regime iii):
    exercises slices that lie wholly inside one bucket and the multi-slice fix-up."""
    rng = _rng(pr, 13)
    n = 5000
    k = rng.fr()
    raw = co.G1.gen_points(rng.fr(), rng.fr(), n)
    pts = ps_api.Points.upload(ctx, ps_api.G1, raw)
    for sc in ([k] * n, [pr.R - 1] * n, [1] * (n // 2) + [0] * (n - n // 2)):
        want = co.G1.to_b(co.G1.msm_pippenger(co.pack_fr(sc), raw, n, 4))
        assert ps_api.Poly.upload(ctx, sc).BlindEval(pts) == want


def test_config2_g1_msm_2pow16_bit_exact(ps_api, ctx, co, pr):
    """BASELINE config #2: synthetic 2^16-point BLS12-381 G1 MSM, bit-exact vs the CPU oracle."""
    rng = _rng(pr, 16)
    n = 1 << 16
    sc = co.pack_fr([rng.fr() for _ in range(n)])
    raw = co.G1.gen_points(rng.fr(), rng.fr(), n)
    want = co.G1.to_b(co.G1.msm_pippenger(sc, raw, n, 8))
    got = ps_api.Poly.upload(ctx, sc).BlindEval(ps_api.Points.upload(ctx, ps_api.G1, raw))
    assert got == want


def test_g2_msm_2pow12_bit_exact(ps_api, ctx, co, pr):
    rng = _rng(pr, 17)
    n = 1 << 12
    sc = co.pack_fr([rng.fr() for _ in range(n)])
    raw = co.G2.gen_points(rng.fr(), rng.fr(), n)
    want = co.G2.to_b(co.G2.msm_pippenger(sc, raw, n, 8))
    got = ps_api.Poly.upload(ctx, sc).BlindEval(ps_api.Points.upload(ctx, ps_api.G2, raw))
    assert got == want


def test_dlog_identity_on_device_generated_points(ps_api, ctx, co, pr):
    """The reference's own oracle method (groth16_test.go:41-50): with P_i = a_i*G,
    sum k_i P_i == (sum k_i a_i)*G.  Points come from the device fixed-base kernel."""
    rng = _rng(pr, 19)
    n = 3000
    a = [rng.fr() for _ in range(n)]
    k = [rng.fr() for _ in range(n)]
    pts = ps_api.Points.from_scalars(ctx, ps_api.G1, ps_api.Poly.upload(ctx, a))
    got = ps_api.Poly.upload(ctx, k).BlindEval(pts)
    dlog = sum(x * y for x, y in zip(a, k)) % pr.R
    assert got == co.G1.to_b(co.G1.mul(dlog))


def test_points_sum_folds_partial_sums(ps_api, co, pr):
    rng = _rng(pr, 23)
    raw = co.G1.gen_points(rng.fr(), rng.fr(), 8) + co.G1.to_b(None)
    want = None
    for p in co.G1.unpack(raw):
        want = co.G1.add(want, p)
    assert ps_api.points_sum(ps_api.G1, raw) == co.G1.to_b(want)
    raw2 = co.G2.gen_points(rng.fr(), rng.fr(), 5)
    want = None
    for p in co.G2.unpack(raw2):
        want = co.G2.add(want, p)
    assert ps_api.points_sum(ps_api.G2, raw2) == co.G2.to_b(want)


@pytest.mark.parametrize("name", ["g1", "g2"])
def test_compressed_upload_decompresses_on_the_gpu(ps_api, ctx, co, pr, name):
    """ps_points_upload(PS_FMT_COMPRESSED): the batch square-root kernel against the oracle's
    decompression, including the identity and both y signs; a bad encoding is rejected."""
    gid, og = _grp(ps_api, co, name)
    comp = pr.g1_compress if name == "g1" else pr.g2_compress
    rng = _rng(pr, 31)
    pts = og.unpack(og.gen_points(rng.fr(), rng.fr(), 37)) + [None]
    pts += [(p[0], (pr.P - p[1]) % pr.P) if name == "g1" else (p[0], ((-p[1][0]) % pr.P, (-p[1][1]) % pr.P)) for p in pts[:5]]
    raw_c = b"".join(comp(p) for p in pts)
    dev = ps_api.Points.upload(ctx, gid, raw_c, fmt=ps_api.FMT_COMPRESSED)
    assert dev.download() == og.pack(pts)
    sc = [rng.fr() for _ in pts]
    assert ps_api.Poly.upload(ctx, sc).BlindEval(dev) == og.to_b(og.blind_eval(sc, og.pack(pts)))
    bad = bytearray(raw_c)
    bad[0] &= 0x7F  # compressed flag cleared
    with pytest.raises(ps_api.PlaysnarkError) as e:
        ps_api.Points.upload(ctx, gid, bytes(bad), fmt=ps_api.FMT_COMPRESSED)
    assert e.value.code == -3


@pytest.mark.parametrize("n", [0, 1, 77, 3000])
def test_msm_multi_shares_one_sort(ps_api, ctx, co, pr, n):
    """ps_msm_multi: several point arrays (G1 and G2 mixed) against ONE scalar vector -- the shape of
    the nine computeSolCommit calls (pinochio.go:231-241) -- equals one BlindEval per array and the
    oracle's sums; a wrong-length array is the reference's length panic (algebra.go:350-352)."""
    rng = _rng(pr, 7000 + n)
    sc = [rng.fr() for _ in range(n)]
    if n > 4:
        sc[1], sc[2], sc[3] = 0, pr.R - 1, sc[4]
    kinds = ["g1", "g2", "g1", "g1"]
    raws, pts = [], []
    for j, name in enumerate(kinds):
        gid, og = _grp(ps_api, co, name)
        raw = og.gen_points(rng.fr(), rng.fr(), n)
        if n > 4 and j == 2:  # identity points and a repeated point inside one array
            raw = og.to_b(None) + raw[og.nb : 2 * og.nb] * 2 + raw[3 * og.nb :]
        raws.append(raw)
        pts.append(ps_api.Points.upload(ctx, gid, raw))
    dsc = ps_api.Poly.upload(ctx, sc)
    got = ps_api.msm_multi(ctx, pts, dsc)
    for name, raw, p, g in zip(kinds, raws, pts, got):
        og = _grp(ps_api, co, name)[1]
        assert g == dsc.BlindEval(p)
        want = og.blind_eval(sc, raw) if n <= 64 else og.msm_pippenger(co.pack_fr(sc), raw, n, 4)
        assert g == og.to_b(want)
    if n:
        with pytest.raises(ps_api.LengthMismatch):
            ps_api.msm_multi(ctx, [pts[0], pts[0].slice(0, n - 1)], dsc)


def test_two_sums_in_flight_fifo(ps_api, ctx, co, pr):
    """Up to PS_MSM_QUEUE sums pending on one context: results come back oldest first and equal the
    one-at-a-time results (each pending sum has its own stream / workspace, accumulations chained in
    launch order); a launch beyond the queue depth is refused; an empty sum may sit in the queue
    (zero.Clone(), algebra.go:353)."""
    rng = _rng(pr, 8800)
    jobs = []
    for gid, og, n in ((ps_api.G1, co.G1, 5000), (ps_api.G2, co.G2, 700), (ps_api.G1, co.G1, 0), (ps_api.G1, co.G1, 64)):
        sc = [rng.fr() for _ in range(n)]
        raw = og.gen_points(rng.fr(), rng.fr(), n)
        pts, dsc = ps_api.Points.upload(ctx, gid, raw), ps_api.Poly.upload(ctx, sc)
        jobs.append((gid, pts, dsc, dsc.BlindEval(pts)))
    from playsnark_amd import _lib

    assert _lib.PS_MSM_QUEUE == 4
    for trio in ((0, 1, 3, 2), (1, 0, 2, 0), (2, 3, 0, 1), (3, 3, 3, 3)):  # PS_MSM_QUEUE sums fill the queue
        for j in trio:
            ps_api.msm_launch(ctx, jobs[j][1], jobs[j][2])
        with pytest.raises(ps_api.PlaysnarkError):  # the queue is full
            ps_api.msm_launch(ctx, jobs[trio[0]][1], jobs[trio[0]][2])
        for j in trio:
            assert ps_api.msm_finish(ctx, jobs[j][0]) == jobs[j][3]
    # steady-state pipelines of depth 2, 3 and 4: launch ahead, finish the oldest
    order = [0, 1, 3, 0, 0, 1, 2, 3, 1]
    for depth in (2, 3, 4):
        launched = finished = 0
        while finished < len(order):
            while launched < len(order) and launched - finished < depth:
                ps_api.msm_launch(ctx, jobs[order[launched]][1], jobs[order[launched]][2])
                launched += 1
            j = order[finished]
            assert ps_api.msm_finish(ctx, jobs[j][0]) == jobs[j][3]
            finished += 1
    with pytest.raises(ps_api.PlaysnarkError):
        ps_api.msm_finish(ctx, ps_api.G1)
    # the one-call form would pop somebody else's result: refused while sums are pending
    ps_api.msm_launch(ctx, jobs[0][1], jobs[0][2])
    with pytest.raises(ps_api.PlaysnarkError):
        jobs[1][2].BlindEval(jobs[1][1])
    assert ps_api.msm_finish(ctx, jobs[0][0]) == jobs[0][3]


def test_msm_multi_limits(ps_api, ctx, co, pr):
    """PS_MSM_MULTI_MAX arrays in one call; one more is refused; zero arrays is a no-op."""
    rng = _rng(pr, 9100)
    n = 40
    sc = [rng.fr() for _ in range(n)]
    dsc = ps_api.Poly.upload(ctx, sc)
    raws = [co.G1.gen_points(rng.fr(), rng.fr(), n) for _ in range(16)]
    pts = [ps_api.Points.upload(ctx, ps_api.G1, raw) for raw in raws]
    got = ps_api.msm_multi(ctx, pts, dsc)
    assert got == [co.G1.to_b(co.G1.blind_eval(sc, raw)) for raw in raws]
    with pytest.raises(ps_api.PlaysnarkError):
        ps_api.msm_multi(ctx, pts + [pts[0]], dsc)
    assert ps_api.msm_multi(ctx, [], dsc) == []


def test_pending_queue_random_schedule(ps_api, ctx, co, pr):
    """A seeded random schedule of launches and finishes over sums of different sizes and groups, with
    one-at-a-time calls and multi-array calls in between whenever the queue is empty: every result
    equals the one-at-a-time result."""
    from playsnark_amd import _lib

    rng = _rng(pr, 9200)
    jobs = []
    for gid, og, n in ((ps_api.G1, co.G1, 3), (ps_api.G1, co.G1, 900), (ps_api.G2, co.G2, 130), (ps_api.G1, co.G1, 4100),
                       (ps_api.G2, co.G2, 1), (ps_api.G1, co.G1, 0)):
        sc = [rng.fr() for _ in range(n)]
        raw = og.gen_points(rng.fr(), rng.fr(), n)
        pts, dsc = ps_api.Points.upload(ctx, gid, raw), ps_api.Poly.upload(ctx, sc)
        jobs.append((gid, pts, dsc, dsc.BlindEval(pts)))
    pending = []
    for step in range(60):
        r = rng.next() % 100
        if pending and (len(pending) == _lib.PS_MSM_QUEUE or r < 45):
            j = pending.pop(0)
            assert ps_api.msm_finish(ctx, jobs[j][0]) == jobs[j][3], (step, j)
        elif not pending and r >= 90:
            j = rng.next() % len(jobs)
            assert jobs[j][2].BlindEval(jobs[j][1]) == jobs[j][3]
            assert ps_api.msm_multi(ctx, [jobs[j][1], jobs[j][1]], jobs[j][2]) == [jobs[j][3]] * 2
        else:
            j = rng.next() % len(jobs)
            ps_api.msm_launch(ctx, jobs[j][1], jobs[j][2])
            pending.append(j)
    while pending:
        j = pending.pop(0)
        assert ps_api.msm_finish(ctx, jobs[j][0]) == jobs[j][3]


@pytest.mark.parametrize("name", ["g1", "g2"])
def test_compressed_download_and_key_files(ps_api, ctx, co, pr, name, tmp_path):
    """MarshalBinary form out of the GPU (pinochio.go:258-272): byte-equal to the oracle's compression,
    identity included; a key file written in either form loads back to the same points; a damaged file
    is refused."""
    gid, og = _grp(ps_api, co, name)
    rng = _rng(pr, 9500)
    n = 300
    raw = og.gen_points(rng.fr(), rng.fr(), n - 1) + og.to_b(None)
    pts = ps_api.Points.upload(ctx, gid, raw)
    compress = co.g1_compress if name == "g1" else co.g2_compress
    want = b"".join(compress(p) for p in og.unpack(raw))
    assert pts.download_compressed() == want
    assert pts.download_compressed(7, 5) == want[7 * og.nb // 2 : 12 * og.nb // 2]
    for compressed in (True, False):
        path = str(tmp_path / f"key_{name}_{int(compressed)}.psnk")
        pts.save(path, compressed=compressed)
        assert os.path.getsize(path) == 16 + n * (og.nb // 2 if compressed else og.nb)
        back = ps_api.Points.load(ctx, path)
        assert back.group == gid and back.download() == raw
    blob = bytearray(open(path, "rb").read())
    blob[3] ^= 1
    open(path, "wb").write(blob)
    with pytest.raises(ps_api.PlaysnarkError):
        ps_api.Points.load(ctx, path)
    open(path, "wb").write(bytes(blob[:3]) + b"K" + bytes(blob[4:-1]))  # one byte short
    with pytest.raises(ps_api.PlaysnarkError):
        ps_api.Points.load(ctx, path)


@pytest.mark.parametrize("name,log2n", [("g1", 20), ("g2", 18)])
def test_full_size_properties_2pow20(ps_api, ctx, co, pr, name, log2n):
    """BASELINE's full size (2^20 points), where the oracle would take minutes: size-independent
    properties of the sum.  (i) additivity over index ranges: MSM(all) = MSM(first part) + MSM(rest),
    for an uneven split; (ii) homogeneity on an adversarial vector: all scalars equal to k gives
    k * MSM(all ones) -- every window then has ONE bucket holding 2^20 entries (the heavy-bucket
    path at its extreme); (iii) the three-in-flight queue returns the same bytes as one at a time."""
    import numpy as np

    gid, og = _grp(ps_api, co, name)
    n = 1 << log2n
    rng = np.random.default_rng(20)
    raw_sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    raw_sc[:, 0] &= 0x3F  # < 2^254 < r: canonical without rejection
    sc = ps_api.Poly.upload(ctx, raw_sc.tobytes())
    seeds = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    seeds[:, 0] &= 0x3F
    pts = ps_api.Points.from_scalars(ctx, gid, ps_api.Poly.upload(ctx, seeds.tobytes()))
    whole = sc.BlindEval(pts)
    cut = n // 3 + 1
    parts = sc.slice(0, cut).BlindEval(pts.slice(0, cut)) + sc.slice(cut, n - cut).BlindEval(pts.slice(cut, n - cut))
    assert ps_api.points_sum(gid, parts) == whole
    # (iii) queue
    for _ in range(3):
        ps_api.msm_launch(ctx, pts, sc)
    assert [ps_api.msm_finish(ctx, gid) for _ in range(3)] == [whole] * 3
    # (ii) all-equal scalars
    k = pr.SplitMix64(77).fr()
    ones = ps_api.Poly.upload(ctx, (b"\x00" * 31 + b"\x01") * n)
    total = ones.BlindEval(pts)
    same = ps_api.Poly.upload(ctx, k.to_bytes(32, "big") * n)
    got = same.BlindEval(pts)
    want = ps_api.Poly.upload(ctx, [k]).BlindEval(ps_api.Points.upload(ctx, gid, total))
    assert got == want
    assert got == og.to_b(og.mul(k, og.from_b(total)))


def _uniform_be32(n, seed):
    """n scalars < 2^254 < r (canonical without rejection) as big-endian rows."""
    import numpy as np

    rows = np.random.default_rng(seed).integers(0, 256, size=(n, 32), dtype=np.uint8)
    rows[:, 0] &= 0x3F
    return rows


def _witness_i64(n, seed):
    """SURVEY 8d regime ii: 40-bit values, a tenth zeros, a tenth ones, a quarter negative."""
    import numpy as np

    rs = np.random.RandomState(seed)
    w = rs.randint(0, 1 << 40, size=n, dtype=np.int64)
    kind = rs.randint(0, 20, size=n)
    w[kind < 2] = 0
    w[(kind >= 2) & (kind < 4)] = 1
    w[kind >= 15] *= -1
    return w


def _i64_to_be32(w, R):
    """Value.ToFieldElement (curve.go:17-19) for a whole vector: v -> v mod r as big-endian rows."""
    import numpy as np

    n = len(w)
    out = np.zeros((n, 32), dtype=np.uint8)
    mag = np.abs(w).astype(np.uint64)
    out[:, 24:] = mag.astype(">u8").view(np.uint8).reshape(n, 8)
    neg = np.nonzero(w < 0)[0]
    if len(neg):  # r - |v|: few distinct patterns would not help; do it with Python ints on the negative rows only
        rows = [(R - int(m)).to_bytes(32, "big") for m in mag[neg]]
        out[neg] = np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(neg), 32)
    return out


@pytest.mark.parametrize("name,log2n", [("g1", 20), ("g2", 18)])
def test_blind_eval_full_size_bit_exact_vs_oracle(ps_api, ctx, co, pr, name, log2n):
    """BASELINE's size (2^20 G1 points; 2^18 for G2), bit-exact against the oracle's CPU Pippenger on the
    same inputs: uniform 255-bit scalars and the int64 witness regime (zeros, ones, negatives -- the
    short-scalar plan with folded signs and the heavy-bucket path).  The points are the oracle's own
    synthetic vector (k0 + i q) G, uploaded as bytes."""
    gid, og = _grp(ps_api, co, name)
    n = 1 << log2n
    threads = min(os.cpu_count() or 1, 16)
    raw = og.gen_points(0x1234567, 0x89ABCDEF, n)
    pts = ps_api.Points.upload(ctx, gid, raw)
    sc = _uniform_be32(n, 2020 + log2n).tobytes()
    assert ps_api.Poly.upload(ctx, sc).BlindEval(pts) == og.to_b(og.msm_pippenger(sc, raw, n, threads))
    w = _witness_i64(n, 3030 + log2n)
    got = ps_api.Poly.from_values(ctx, w.tolist()).BlindEval(pts)
    assert ctx.last_msm_info()["windows"] * ctx.last_msm_info()["window_bits"] < 128  # the short plan was taken
    assert got == og.to_b(og.msm_pippenger(_i64_to_be32(w, pr.R).tobytes(), raw, n, threads))


def test_config4_2pow24_points_in_8_index_range_shards(ps_api, ctx, co, pr):
    """BASELINE config #4 on one GPU: a 2^24-point G1 MSM as the 8 index-range shards that 8 ranks would
    take (playsnark_amd.dist.shard_range), folded with ps_points_sum exactly as after the all_gather --
    equal to the unsharded sum and, bit for bit, to the oracle's CPU Pippenger over the same 2^24 inputs.
    PS_SKIP_ORACLE_2P24=1 drops the 20-second CPU leg."""
    from playsnark_amd.dist import shard_range

    n, world = 1 << 24, 8
    seeds = ps_api.Poly.upload(ctx, _uniform_be32(n, 4040).tobytes())
    pts = ps_api.Points.from_scalars(ctx, ps_api.G1, seeds)
    del seeds
    sc_bytes = _uniform_be32(n, 5050).tobytes()
    sc = ps_api.Poly.upload(ctx, sc_bytes)
    whole = sc.BlindEval(pts)
    assert ctx.last_msm_info()["entries"] > 15 * n  # every scalar went through all of its windows
    parts = b""
    for g in range(world):
        first, cnt = shard_range(n, g, world)
        assert cnt == n // world
        ps_api.msm_launch(ctx, pts.slice(first, cnt), sc.slice(first, cnt))
        parts += ps_api.msm_finish(ctx, ps_api.G1)
    assert ps_api.points_sum(ps_api.G1, parts) == whole
    if os.environ.get("PS_SKIP_ORACLE_2P24") != "1":
        raw = pts.download()
        assert whole == co.G1.to_b(co.G1.msm_pippenger(sc_bytes, raw, n, min(os.cpu_count() or 1, 16)))


def test_inputs_produced_between_launches_are_ordered(ps_api, ctx, co, pr, hipmem):
    """ps_points_from_scalars / ps_scalars_from_device_be32 return before their kernels have run; a sum
    launched on a worker stream later in the same burst must still see their output."""
    rng = _rng(pr, 4711)
    n = 1 << 16
    raw = co.G1.gen_points(rng.fr(), rng.fr(), n)
    sc_bytes = _uniform_be32(n, 6060).tobytes()
    sc = ps_api.Poly.upload(ctx, sc_bytes)
    ptsA = ps_api.Points.upload(ctx, ps_api.G1, raw)
    wantA = co.G1.to_b(co.G1.msm_pippenger(sc_bytes, raw, n, 8))
    ks = _uniform_be32(n, 7070)
    dev = hipmem.alloc(32 * n, ks.tobytes())
    for _ in range(3):
        ps_api.msm_launch(ctx, ptsA, sc)                                 # the context's own stream is busy now
        k2 = ps_api.Poly.from_device_be32(ctx, dev, n)                   # asynchronous, on the context stream
        p2 = ps_api.Points.from_scalars(ctx, ps_api.G1, k2)              # asynchronous as well
        ps_api.msm_launch(ctx, p2, k2)                                   # runs on a worker stream
        assert ps_api.msm_finish(ctx, ps_api.G1) == wantA
        got = ps_api.msm_finish(ctx, ps_api.G1)
        del p2, k2
    # sum_i k_i (k_i G) = (sum k_i^2) G
    tot = sum(int.from_bytes(ks[i].tobytes(), "big") ** 2 for i in range(n)) % pr.R
    assert got == co.G1.to_b(co.G1.mul(tot))


def test_sum_too_long_for_the_32bit_sort_offsets_is_refused(ps_api, ctx, hipmem):
    """windows x length >= 2^32 would wrap the u32 offsets of the counting sort: PS_ERR_ARG, not a wrong point.
    Forced 4-bit windows (64 of them) reach the limit at 2^26 scalars."""
    n = 1 << 26
    k = ps_api.Poly.from_device_be32(ctx, hipmem.alloc(32 * n), n)
    pts = ps_api.Points.from_scalars(ctx, ps_api.G1, k)
    ctx.set_window(4)
    try:
        with pytest.raises(ps_api.PlaysnarkError) as e:
            ps_api.msm_launch(ctx, pts, k)
        assert e.value.code == -5 and "32-bit" in str(e.value)
        with pytest.raises(ps_api.PlaysnarkError):
            ps_api.msm_multi(ctx, [pts, pts], k)
    finally:
        ctx.set_window(0)
    half = k.slice(0, n // 2 - 1)
    ctx.set_window(4)
    try:
        assert half.BlindEval(pts.slice(0, n // 2 - 1)) == b"\x40" + bytes(95)  # just under the limit: runs (all-zero scalars)
    finally:
        ctx.set_window(0)


@pytest.mark.parametrize("name", ["g1", "g2"])
def test_points_subgroup_check_on_the_device(ps_api, ctx, co, pr, name, off_subgroup):
    gid, og = _grp(ps_api, co, name)
    good = og.gen_points(99, 101, 70) + og.to_b(None)
    assert ps_api.Points.upload(ctx, gid, good).in_subgroup()
    bad_pt = off_subgroup[0 if name == "g1" else 1]
    bad = good[: 33 * og.nb] + og.to_b(bad_pt) + good[34 * og.nb :]
    arr = ps_api.Points.upload(ctx, gid, bad)  # on the curve: the upload accepts it
    assert not arr.in_subgroup()
    assert arr.slice(0, 33).in_subgroup() and not arr.slice(33, 1).in_subgroup()
    # the identity must be 0x40 followed by zeros
    junk = bytearray(og.to_b(None))
    junk[-1] = 1
    with pytest.raises(ps_api.PlaysnarkError) as e:
        ps_api.Points.upload(ctx, gid, bytes(junk))
    assert e.value.code == -3


# ---- window tables (ps_points_precompute): the same sums, all windows in one bucket set ----
@pytest.mark.parametrize("name", ["g1", "g2"])
@pytest.mark.parametrize("n,wbits", [(1, 8), (5, 8), (333, 9), (1024, 12), (3000, 0)])
def test_window_table_sums_equal_the_plain_sums(ps_api, ctx, co, pr, name, n, wbits):
    """Same group element, bit for bit, as the plain path and the oracle: random scalars, the edge scalars 0 / 1 / r-1,
    identity points, repeated points and P / -P pairs in one bucket, int64 witnesses with negative values, slices."""
    gid, og = _grp(ps_api, co, name)
    rng = _rng(pr, 31000 + n)
    pts_list = og.unpack(og.gen_points(rng.fr(), rng.fr(), n))
    if n >= 5:
        P = pts_list[0]
        negP = (P[0], (pr.P - P[1]) % pr.P) if name == "g1" else (P[0], ((-P[1][0]) % pr.P, (-P[1][1]) % pr.P))
        pts_list[1], pts_list[2], pts_list[3] = P, negP, None
    raw = og.pack(pts_list)
    plain = ps_api.Points.upload(ctx, gid, raw)
    tab = ps_api.Points.upload(ctx, gid, raw).precompute(wbits)
    assert tab.table_window == (wbits or tab.table_window) and tab.table_window >= 8 and plain.table_window == 0
    k = rng.fr()
    vectors = [[rng.fr() for _ in range(n)], [k] * n, [0, 1, pr.R - 1, pr.R - 2, 2][:n] + [rng.fr() for _ in range(max(0, n - 5))]]
    for sc in vectors:
        want = og.to_b(og.blind_eval(sc, raw)) if n <= 64 else og.to_b(og.msm_pippenger(co.pack_fr(sc), raw, n, 4))
        poly = ps_api.Poly.upload(ctx, sc)
        got = poly.BlindEval(tab)
        assert ctx.last_msm_info()["window_bits"] == tab.table_window and ctx.last_msm_info()["buckets"] == 1 << (tab.table_window - 1)
        assert got == want and poly.BlindEval(plain) == want
    vals = [0, 1, -1, 35, -(1 << 62), (1 << 63) - 1, -(1 << 63)][:n] + [int(rng.next() % 2000) - 1000 for _ in range(max(0, n - 7))]
    assert ps_api.Poly.from_values(ctx, vals).BlindEval(tab) == og.to_b(og.blind_eval_i64(vals, raw))
    if n >= 333:  # an index-range shard of a precomputed array uses the same table (or, from 1024 points on, one of its own)
        first, cnt = n // 3, n // 2
        sc = vectors[0]
        got = ps_api.Poly.upload(ctx, sc[first:first + cnt]).BlindEval(tab.slice(first, cnt))
        assert ctx.last_msm_info()["window_table"] == 1 and (cnt >= 1024 or ctx.last_msm_info()["buckets"] == 1 << (tab.table_window - 1))
        assert got == og.to_b(og.msm_pippenger(co.pack_fr(sc[first:first + cnt]), raw[first * og.nb:(first + cnt) * og.nb], cnt, 4))


@pytest.mark.parametrize("name", ["g1", "g2"])
def test_index_range_views_get_a_window_table_of_their_own(ps_api, ctx, co, pr, name):
    """An index-range shard of a key whose table was built for the WHOLE array (an eighth of a 2^20-point key over
    its 20-bit table reduced 2^19 buckets for 2^17 points): the first sum over such a view builds a table for the view's own
    length (smaller windows, a fraction of the buckets), later sums and other contexts find it, multi-sums use it when every
    array of the call has one, the bytes are the oracle's, and ps_points_precompute(p, -1) releases it with the whole table."""
    gid, og = _grp(ps_api, co, name)
    rng = _rng(pr, 36000)
    n = 1 << 15 if name == "g1" else 1 << 14
    a = ps_api.Poly.upload(ctx, [rng.fr() for _ in range(n)])
    whole = ps_api.Points.from_scalars(ctx, gid, a).precompute(16)
    raw = whole.download()
    first, cnt = 4099, 2048
    view = whole.slice(first, cnt)
    sc = [rng.fr() for _ in range(cnt)]
    want = og.to_b(og.msm_pippenger(co.pack_fr(sc), raw[first * og.nb:(first + cnt) * og.nb], cnt, 4))
    poly = ps_api.Poly.upload(ctx, sc)
    for _ in range(2):  # built by the first sum, found by the second
        assert poly.BlindEval(view) == want
        info = ctx.last_msm_info()
        assert info["window_table"] == 1 and info["window_bits"] < 16 and info["buckets"] == 1 << (info["window_bits"] - 1)
    view_bits = info["window_bits"]
    # another context and a second handle on the same range find the same table
    cx2 = ps_api.Context(0)
    assert poly.BlindEval(whole.slice(first, cnt)) == want
    p2 = ps_api.Poly.upload(cx2, sc)
    assert p2.BlindEval(view) == want and cx2.last_msm_info()["window_bits"] == view_bits
    cx2.close()
    # a multi-sum over two views of the same range of two arrays; then with one array that has no table at all (plain plan)
    other = ps_api.Points.from_scalars(ctx, gid, ps_api.Poly.upload(ctx, [rng.fr() for _ in range(n)])).precompute(16)
    oraw = other.download()
    want_o = og.to_b(og.msm_pippenger(co.pack_fr(sc), oraw[first * og.nb:(first + cnt) * og.nb], cnt, 4))
    assert ps_api.msm_multi(ctx, [view, other.slice(first, cnt)], poly) == [want, want_o]
    assert ctx.last_msm_info()["window_bits"] == view_bits and ctx.last_msm_info()["window_table"] == 1
    bare = ps_api.Points.upload(ctx, gid, oraw)
    assert ps_api.msm_multi(ctx, [view, bare.slice(first, cnt)], poly) == [want, want_o]
    assert ctx.last_msm_info()["window_table"] == 0
    # a view almost as long as the array keeps the array's own table (nothing to gain), int64 witnesses over the view table
    long_view = whole.slice(1, n - 2)
    sc_long = [rng.fr() for _ in range(n - 2)]
    got = ps_api.Poly.upload(ctx, sc_long).BlindEval(long_view)
    assert ctx.last_msm_info()["window_table"] == 1 and (name == "g2" or ctx.last_msm_info()["window_bits"] == 16)
    assert got == og.to_b(og.msm_pippenger(co.pack_fr(sc_long), raw[og.nb:(n - 1) * og.nb], n - 2, 4))
    vals = [int(rng.next() % 2000) - 1000 for _ in range(cnt)]
    assert ps_api.Poly.from_values(ctx, vals).BlindEval(view) == og.to_b(og.blind_eval_i64(vals, raw[first * og.nb:(first + cnt) * og.nb]))
    # a view whose table does not fit the context's budget gets none (the plans it had before apply); a later request with room builds it
    v3 = whole.slice(9000, 3000)
    sc3 = [rng.fr() for _ in range(3000)]
    want3 = og.to_b(og.msm_pippenger(co.pack_fr(sc3), raw[9000 * og.nb:12000 * og.nb], 3000, 4))
    p3 = ps_api.Poly.upload(ctx, sc3)
    ctx.set_table_budget(4096)
    try:
        assert p3.BlindEval(v3) == want3  # (the array's own 16-bit table, or the plain plan where the model finds that cheaper)
        assert ctx.last_msm_info()["window_table"] == 0 or ctx.last_msm_info()["window_bits"] == 16
    finally:
        ctx.set_table_budget(-1)
    assert p3.BlindEval(v3) == want3 and ctx.last_msm_info()["window_bits"] < 16 and ctx.last_msm_info()["window_table"] == 1
    # tables off for the context: no table of any kind is read; release: the whole array's plain plan
    ctx.set_tables(False)
    try:
        assert poly.BlindEval(view) == want and ctx.last_msm_info()["window_table"] == 0
    finally:
        ctx.set_tables(True)
    assert poly.BlindEval(view) == want and ctx.last_msm_info()["window_bits"] == view_bits
    whole.precompute(-1)
    assert poly.BlindEval(view) == want and ctx.last_msm_info()["window_table"] == 0


@pytest.mark.parametrize("name", ["g1", "g2"])
def test_blind_eval_with_host_scalars_is_seam_s1(ps_api, ctx, co, pr, name):
    """ps_msm_be32 / ps_msm_i64 -- Poly.BlindEval (algebra.go:348-359) as the cgo shim calls it, scalars in host memory: the
    context keeps ONE upload vector between calls (no allocation per call), so lengths that grow, shrink and repeat, both
    scalar forms in turn, a table-carrying array, an empty sum and the length-mismatch panic all go through it."""
    import numpy as np

    gid, og = _grp(ps_api, co, name)
    rng = _rng(pr, 37000)
    nmax = 3000
    raw = og.gen_points(rng.fr(), rng.fr(), nmax)
    pts = ps_api.Points.upload(ctx, gid, raw).precompute(0)
    for n in (5, 3000, 64, 1, 2999, 700, 700):
        sc = [rng.fr() for _ in range(n)]
        view = pts.slice(0, n)
        want = og.to_b(og.msm_pippenger(co.pack_fr(sc), raw[:n * og.nb], n, 4))
        assert ps_api.blind_eval_host(ctx, view, co.pack_fr(sc)) == want
        vals = [int(rng.next() % 2001) - 1000 for _ in range(n)]
        vals[0] = -(1 << 63)
        assert ps_api.blind_eval_host(ctx, view, np.array(vals, dtype=np.int64)) == og.to_b(og.blind_eval_i64(vals, raw[:n * og.nb]))
        assert ps_api.Poly.upload(ctx, sc).BlindEval(view) == want  # device-resident scalars in between
    ident = bytes([0x40]) + bytes(og.nb - 1)
    assert ps_api.blind_eval_host(ctx, pts.slice(0, 0), b"") == ident
    with pytest.raises(ps_api.LengthMismatch):
        ps_api.blind_eval_host(ctx, pts, co.pack_fr([1, 2, 3]))


def test_view_tables_are_built_once_under_two_racing_threads(ps_api, co, pr):
    """Two host threads, each with its own context, race into the FIRST sums over views of one shared table-carrying array
    (the shares of a sharded prover): the view tables are built under the array's lock and published complete; every sum of
    both threads, over several views, is the oracle's."""
    import threading

    rng = _rng(pr, 38000)
    n = 1 << 14
    ctx0 = ps_api.Context(0)
    whole = ps_api.Points.from_scalars(ctx0, ps_api.G1, ps_api.Poly.upload(ctx0, [rng.fr() for _ in range(n)])).precompute(16)
    raw = whole.download()
    views = [(0, 2048), (2048, 2048), (5000, 1500), (n - 1024, 1024)]
    scs = {v: [rng.fr() for _ in range(v[1])] for v in views}
    want = {v: co.G1.to_b(co.G1.msm_pippenger(co.pack_fr(scs[v]), raw[v[0] * 96:(v[0] + v[1]) * 96], v[1], 4)) for v in views}
    errors, barrier = [], threading.Barrier(2)

    def worker(tid):
        try:
            cx = ps_api.Context(0)
            polys = {v: ps_api.Poly.upload(cx, scs[v]) for v in views}
            barrier.wait()
            for rep in range(3):
                for v in (views if tid == 0 else views[::-1]):
                    got = polys[v].BlindEval(whole.slice(*v))
                    assert got == want[v], (tid, rep, v)
                    assert cx.last_msm_info()["window_table"] == 1 and cx.last_msm_info()["window_bits"] < 16
            cx.close()
        except BaseException as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((tid, repr(e)))
            try:
                barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    ctx0.close()


def test_window_table_in_the_queue_in_multi_sums_and_rebuilt(ps_api, ctx, co, pr):
    """Tables in ps_msm_launch / ps_msm_finish bursts, in ps_msm_multi (all arrays with tables of one window size: the
    table plan; otherwise the plain plan), with ps_msm_set_window forcing the plain path, and rebuilt for another size."""
    rng = _rng(pr, 32000)
    n = 700
    sc = [rng.fr() for _ in range(n)]
    poly = ps_api.Poly.upload(ctx, sc)
    raws = [co.G1.gen_points(rng.fr(), rng.fr(), n), co.G2.gen_points(rng.fr(), rng.fr(), n), co.G1.gen_points(rng.fr(), rng.fr(), n)]
    gids = [ps_api.G1, ps_api.G2, ps_api.G1]
    ogs = [co.G1, co.G2, co.G1]
    want = [og.to_b(og.msm_pippenger(co.pack_fr(sc), raw, n, 4)) for og, raw in zip(ogs, raws)]
    arrs = [ps_api.Points.upload(ctx, g, raw).precompute(10) for g, raw in zip(gids, raws)]
    for a in arrs:
        ps_api.msm_launch(ctx, a, poly)
    assert [ps_api.msm_finish(ctx, g) for g in gids] == want
    assert ps_api.msm_multi(ctx, arrs, poly) == want and ctx.last_msm_info()["buckets"] == 1 << 9
    arrs[1].precompute(11)  # a different window size in the same call: the plain plan for all
    assert ps_api.msm_multi(ctx, arrs, poly) == want and ctx.last_msm_info()["buckets"] > 1 << 10
    assert poly.BlindEval(arrs[1]) == want[1] and ctx.last_msm_info()["window_bits"] == 11
    ctx.set_window(7)
    try:
        assert poly.BlindEval(arrs[0]) == want[0] and ctx.last_msm_info()["window_bits"] == 7
    finally:
        ctx.set_window(0)
    with pytest.raises(ps_api.PlaysnarkError):
        arrs[0].precompute(5)


def test_window_table_full_size_bit_exact_vs_oracle(ps_api, ctx, co, pr):
    """2^20 G1 points with their table (c = 20: 13 windows, one set of 2^19 buckets, the split reduction jobs), uniform
    and witness scalars, bit-exact against the oracle; c = 18 forced on 2^16 points for the same code path at a second size."""
    n = 1 << 20
    threads = min(os.cpu_count() or 1, 16)
    raw = co.G1.gen_points(0x7654321, 0xFEDCBA9, n)
    pts = ps_api.Points.upload(ctx, ps_api.G1, raw).precompute()
    c = pts.table_window
    assert c == 20  # the cost model's choice at this length (19 would leave an 8-bit top window)
    sc = _uniform_be32(n, 8080).tobytes()
    want = co.G1.to_b(co.G1.msm_pippenger(sc, raw, n, threads))
    got = ps_api.Poly.upload(ctx, sc).BlindEval(pts)
    info = ctx.last_msm_info()
    assert (info["window_bits"], info["windows"], info["buckets"]) == (c, 255 // c + 1, 1 << (c - 1)) and info["entries"] < (255 // c + 1) * n
    assert got == want
    assert pts.drop_table().table_window == 0  # released: the plain plan again, the same bytes
    assert ps_api.Poly.upload(ctx, sc).BlindEval(pts) == want and ctx.last_msm_info()["window_bits"] == 16
    pts.precompute(19)  # a window size whose top window is nearly empty: a million entries on 116 buckets
    assert ps_api.Poly.upload(ctx, sc).BlindEval(pts) == want and ctx.last_msm_info()["window_bits"] == 19
    pts.precompute(20)
    w = _witness_i64(n, 9090)
    got = ps_api.Poly.from_values(ctx, w.tolist()).BlindEval(pts)
    # an int64 witness over a 20-bit table would take 4 windows but 2^19 buckets to reduce: the cost model prefers the
    # plain plan's small windows for short scalars (1.28 -> 1.04 ms per sum), the table stays for full-width ones
    assert ctx.last_msm_info()["windows"] == 5 and ctx.last_msm_info()["window_bits"] == 13
    assert got == co.G1.to_b(co.G1.msm_pippenger(_i64_to_be32(w, pr.R).tobytes(), raw, n, threads))
    m = 1 << 16
    small = ps_api.Points.upload(ctx, ps_api.G1, raw[: 96 * m]).precompute(18)
    assert ps_api.Poly.upload(ctx, sc[: 32 * m]).BlindEval(small) == co.G1.to_b(co.G1.msm_pippenger(sc[: 32 * m], raw[: 96 * m], m, threads))


def test_wire_values_of_a_boolean_circuit_over_a_window_table(ps_api, ctx, co, pr):
    """Scalars that are all 0 or 1 (the values L.s of booleanity gates, which a Lagrange-form key makes the scalars of the
    sums) put 2^19 entries into ONE of 2^19 buckets: correct bytes, and no walk over the empty buckets (that walk, one
    dependent load per bucket, once cost 40 ms per sum; the time is reported, a slow box does not fail the parity test)."""
    import time

    import numpy as np

    n = 1 << 20
    raw_seeds = _uniform_be32(n, 1212).tobytes()
    pts = ps_api.Points.from_scalars(ctx, ps_api.G1, ps_api.Poly.upload(ctx, raw_seeds)).precompute()
    bits = np.random.RandomState(5).randint(0, 2, size=n).astype(np.uint8)
    rows = np.zeros((n, 32), dtype=np.uint8)
    rows[:, 31] = bits
    sc = ps_api.Poly.upload(ctx, rows.tobytes())
    sc.BlindEval(pts)  # warm-up
    ctx.sync()
    t0 = time.perf_counter()
    got = sc.BlindEval(pts)
    ms = (time.perf_counter() - t0) * 1e3
    _report_ms("2^20-point sum of 0/1 scalars over a window table", ms, 25)
    # sum of the selected points = (sum of their discrete logs) G
    seeds = np.frombuffer(raw_seeds, dtype=np.uint8).reshape(n, 32)
    total = sum(int.from_bytes(seeds[i].tobytes(), "big") for i in np.nonzero(bits)[0]) % pr.R
    assert got == co.G1.to_b(co.G1.mul(total))


@pytest.mark.gpu
@pytest.mark.parametrize("log2n", [17, 19])
def test_dense_buckets_do_not_take_the_heavy_path(ps_api, ctx, co, pr, log2n):
    """Lengths whose automatic table window leaves hundreds of entries per bucket (2^19 points over a 16-bit table: 256):
    the slice length follows the bucket size, so an average bucket spans a few slices and not the eight that route it to
    the heavy-bucket kernels -- 2^19 points once took 9.9 ms per sum, more than 2^21.  Additivity of the sum in the
    scalars for the bytes; the time is reported (a warning when over budget), never asserted."""
    import time

    n = 1 << log2n
    pts = ps_api.Points.from_scalars(ctx, ps_api.G1, ps_api.Poly.upload(ctx, _uniform_be32(n, 3131).tobytes())).precompute()
    a, b = _uniform_be32(n, 3132), _uniform_be32(n, 3133)
    pa, pb = ps_api.Poly.upload(ctx, a.tobytes()), ps_api.Poly.upload(ctx, b.tobytes())
    ra = pa.BlindEval(pts)
    info = ctx.last_msm_info()
    assert info["entries"] <= 5 * info["slice"] * info["buckets"], info  # an average bucket spans at most five slices
    ctx.sync()
    t0 = time.perf_counter()
    rb = pb.BlindEval(pts)
    ms = (time.perf_counter() - t0) * 1e3
    ia = [int.from_bytes(a[i].tobytes(), "big") for i in range(n)]
    ib = [int.from_bytes(b[i].tobytes(), "big") for i in range(n)]
    ab = ps_api.Poly.upload(ctx, [(x + y) % pr.R for x, y in zip(ia, ib)])
    assert ps_api.points_sum(ps_api.G1, ra + rb) == ab.BlindEval(pts)
    _report_ms(f"2^{log2n}-point sum, dense buckets", ms, 6)


@pytest.mark.parametrize("name", ["g1", "g2"])
@pytest.mark.parametrize("n,wbits,table", [(1, 0, False), (7, 0, False), (333, 0, False), (1024, 0, True), (3000, 9, True),
                                           (5000, 13, True), (1 << 14, 0, True), (1 << 14, 0, False)])
def test_tail_of_a_sum_chains_and_trees_give_the_oracle_bytes(ps_api, ctx, co, pr, name, n, wbits, table):
    """The tail of a sum (fix-up of cut buckets + bucket reduction) exists twice: work-efficient chains (long sums) and
    shallow trees of lane-cooperative additions (short sums, csrc/qtail.hpp).  Both must give the oracle's bytes for
    Poly.BlindEval (algebra.go:348-359) -- plain plan and window table, both groups, every slice length that changes the
    shape of the fix-up (cut buckets summed by 1, 2, .. 32 quads), uniform and skewed scalars."""
    gid, og = _grp(ps_api, co, name)
    if name == "g2" and n > 5000:
        n = 5000
    rng = _rng(pr, 4100 + n)
    raw = og.gen_points(rng.fr(), rng.fr(), n)
    pts = ps_api.Points.upload(ctx, gid, raw)
    if table:
        pts.precompute(wbits)
    k = rng.fr()
    vectors = [[rng.fr() for _ in range(n)],
               [k] * n,                                                     # one bucket per window holds everything
               [int(rng.next() & 1) for _ in range(n)],                     # wire values of a boolean circuit
               [1 if i & 1 else rng.fr() for i in range(n)],                # half ones: ONE heavy bucket among ordinary ones (the
                                                                            # quad-tree heavy kernels of a short sum, qtail.hpp)
               [0] * n]
    try:
        for vi, sc in enumerate(vectors):
            packed = co.pack_fr(sc)
            want = og.to_b(og.blind_eval(sc, raw)) if n <= 64 else og.to_b(og.msm_pippenger(packed, raw, n, 4))
            poly = ps_api.Poly.upload(ctx, sc)
            for mode in (1, 2):
                for sl in ((0,) if vi else (0, 2, 4, 16)):
                    ctx.set_tail(mode)
                    ctx.set_slice(sl)
                    assert poly.BlindEval(pts) == want, (name, n, wbits, table, vi, mode, sl)
    finally:
        ctx.set_tail(0)
        ctx.set_slice(0)


@pytest.mark.parametrize("n,table", [(2047, True), (2047, False), (300, True), (9000, True)])
def test_multi_sum_over_both_groups_shares_one_tail_plan(ps_api, ctx, co, pr, n, table):
    """computeSolCommit's sums (pinochio.go:231-241) go through ps_msm_multi: ONE digit sort and ONE plan for arrays of
    both groups (PHGR13: ws is G2).  The tree tail sizes its fix-up by lanes per point, which differ between the groups
    (4 / 8): a plan sized for the G1 arrays once put 64 quads of a bucket into a G2 block that holds 32 -- the G2 element of
    a 2^11-constraint PHGR13 proof came back as zeros (round 3, found by the two-thread prover test).  Chains and trees,
    with skewed scalars (many slices per bucket), against the oracle."""
    rng = _rng(pr, 5200 + n)
    kinds = ["g1", "g2", "g1"]
    raws, pts = [], []
    for name in kinds:
        gid, og = _grp(ps_api, co, name)
        raw = og.gen_points(rng.fr(), rng.fr(), n)
        p = ps_api.Points.upload(ctx, gid, raw)
        if table:
            p.precompute(0)
        raws.append(raw)
        pts.append(p)
    try:
        for sc in ([rng.fr() for _ in range(n)], [int(rng.next() & 1) for _ in range(n)], [rng.fr()] * n):
            dsc = ps_api.Poly.upload(ctx, sc)
            want = [_grp(ps_api, co, name)[1] for name in kinds]
            want = [og.to_b(og.msm_pippenger(co.pack_fr(sc), raw, n, 4)) for og, raw in zip(want, raws)]
            for mode in (1, 2, 0):
                ctx.set_tail(mode)
                assert ps_api.msm_multi(ctx, pts, dsc) == want, (n, table, mode)
    finally:
        ctx.set_tail(0)
