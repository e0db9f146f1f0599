"""GPU, BASELINE configs #3/#5 shape: a synthetic 2^k-constraint circuit proved end to end with the
CRS generated on the device (fixed-base kernel), checked at full size by the reference's own
method -- recompute the discrete logs of A, B, C from the retained toxic waste
(TestGroth16ProofGen, groth16_test.go:32-107) -- plus the QAP identity at a random point.

Runs at n = 2^16 and at BASELINE's n = 2^20 (configs #3 and #5) by default; PS_SCALE_LOG2N=k runs that one
size instead (e.g. 22) and prints the stage timings (used for DESIGN.md)."""
import json
import os
import time

import pytest

pytestmark = pytest.mark.gpu
SEED = 0x706C6179736E61726B & 0xFFFFFFFFFFFFFFFF


def _powers(x, n, shift, R):
    out, cur = [], shift % R
    for _ in range(n):
        out.append(cur)
        cur = cur * x % R
    return out


def _sizes():
    return [int(os.environ["PS_SCALE_LOG2N"])] if "PS_SCALE_LOG2N" in os.environ else [16, 20]


@pytest.mark.parametrize("log2n", _sizes())
def test_groth16_and_phgr13_at_scale(ps_api, ctx, co, pr, log2n):
    from oracle import restate as rs

    n = 1 << log2n
    R = pr.R
    rng = pr.SplitMix64(SEED + 2020)
    t = {}
    t_begin = time.time()

    def progress(what):  # a line per stage: long runs (PS_SCALE_LOG2N >= 22) must not look hung
        print(f"[scale 2^{log2n}] {what} at {time.time() - t_begin:.1f} s", flush=True)

    t0 = time.time()
    c, sol = rs.synthetic_circuit(n)
    # The reference splits the variables at diff = nbVars - nbIO (groth16.go:86, pinochio.go:122) and
    # treats sol[:diff] as the public part.  Declaring nbIO = nbVars - 3 makes that split fall after
    # (const, x, out): three public values for the verifier, ~n witness values for the prover's sums --
    # the shape BASELINE configs #3 / #5 mean (full-size G1 and G2 sums), under the reference's own rule.
    c = rs.SparseR1CS(c.nbVars, c.nbVars - 3, c.left, c.right, c.out)
    m, diff = c.nbVars, c.nbVars - c.nbIO
    assert diff == 3
    alpha, beta, delta, x, gamma = (rng.fr() for _ in range(5))
    u, v, w, zx = rs.var_poly_evals(c, x)
    t["host_circuit_and_setup_scalars_s"] = time.time() - t0
    progress("circuit and setup scalars on the host")

    # ---- CRS on the device: NewGroth16TrustedSetup (groth16.go:64-101), toxic waste retained for the checks
    up = lambda vals: ps_api.Poly.upload(ctx, vals)
    one = lambda grp, k: grp.to_b(grp.mul(k))
    t0 = time.time()
    q = ps_api.QAP(ctx, m, c.nbIO, c.left, c.right, c.out)
    ctx.sync()
    t["qap_create_s"] = time.time() - t0
    t0 = time.time()
    tr, vk = ps_api.NewGroth16TrustedSetup(q, alpha, beta, delta, x, gamma)
    ctx.sync()
    t["device_setup_s"] = time.time() - t0
    progress("QAP tables and Groth16 CRS on the device")
    # spot-check the device CRS against scalars recomputed on the host
    xi_s = _powers(x, n, 1, R)
    assert tr.Xi.download(5, 1) == one(co.G1, xi_s[5]) and tr.Xi2.download(n - 1, 1) == one(co.G2, xi_s[n - 1])
    assert tr.XiT.download(n - 2, 1) == one(co.G1, xi_s[n - 2] * pr.fr_div(zx, delta) % R)
    lin = lambda i, div: pr.fr_div((w[i] + beta * u[i] + alpha * v[i]) % R, div)
    assert len(tr.NioLP) == m - diff and len(vk["IoLP"]) == diff
    for i in (diff, diff + 1, m // 2, m - 1):
        assert tr.NioLP.download(i - diff, 1) == one(co.G1, lin(i, delta))
    assert vk["IoLP"].download() == b"".join(one(co.G1, lin(i, gamma)) for i in range(diff))
    dsol = up(sol)

    r, s = rng.fr(), rng.fr()
    ctx.set_tables(False)  # the plain plan first: no window tables anywhere yet
    ps_api.Groth16Prove(tr, q, dsol, r, s)  # warm-up (workspace allocation)
    t0 = time.time()
    plain_proof = ps_api.Groth16Prove(tr, q, dsol, r, s)
    t["groth16_prove_plain_plan_s"] = time.time() - t0
    ctx.set_tables(True)   # window tables for the CRS arrays (built once, inside the warm-up call)
    ps_api.Groth16Prove(tr, q, dsol, r, s)
    info = ctx.last_msm_info()
    assert info["buckets"] == 1 << (info["window_bits"] - 1)  # one bucket set: the table plan was taken
    t0 = time.time()
    proof = ps_api.Groth16Prove(tr, q, dsol, r, s)
    t["groth16_prove_s"] = time.time() - t0
    t["groth16_phase_ms"] = {k: round(v, 2) for k, v in ctx.last_prove_phase_ms().items()}
    assert (proof.A, proof.B, proof.C) == (plain_proof.A, plain_proof.B, plain_proof.C)
    # the key as the reference's setup makes it (monomial arrays only): coefficients, interpolations, the division
    mono_key = tr.monomial_only()
    ps_api.Groth16Prove(mono_key, q, dsol, r, s)
    t0 = time.time()
    mono = ps_api.Groth16Prove(mono_key, q, dsol, r, s)
    t["groth16_prove_monomial_key_s"] = time.time() - t0
    t["groth16_monomial_key_phase_ms"] = {k: round(v, 2) for k, v in ctx.last_prove_phase_ms().items()}
    assert (mono.A, mono.B, mono.C) == (proof.A, proof.B, proof.C)
    del mono_key
    progress("Groth16 proof")
    # BASELINE config #4 / #5's other half at full size: the shares of 8 ranks (index ranges of Xi, Xi2, NioLP, XiT through
    # ps_groth16_prove_shard, one after the other on this GPU), folded, are the unsharded proof byte for byte
    from playsnark_amd.dist import ShardedGroth16, ShardedPHGR13
    t0 = time.time()
    sh = ShardedGroth16(ctx, None, 8, 0)
    folded = sh.fold([sh.partials(tr, q, dsol, r, s, rank=g) for g in range(8)], r, s)
    t["groth16_8_shares_one_gpu_s"] = time.time() - t0
    assert (folded.A, folded.B, folded.C) == (proof.A, proof.B, proof.C)
    progress("Groth16 proof from the shares of 8 ranks")

    # ---- TestGroth16ProofGen at full size ----
    t0 = time.time()
    A_c, B_c, C_c, h = (p.download_bytes() for p in q.computeAggregatePoly(dsol))
    t["quotient_with_C_and_download_s"] = time.time() - t0
    import ctypes as C

    def poly_eval_bytes(raw, cnt, at):
        out = C.create_string_buffer(32)
        co.lib().or_poly_eval(raw, C.c_size_t(cnt), pr.fr_to_be32(at), out)
        return int.from_bytes(out.raw, "big")

    Ax, Bx, Cx, hx = (poly_eval_bytes(A_c, n, x), poly_eval_bytes(B_c, n, x), poly_eval_bytes(C_c, n, x),
                      poly_eval_bytes(h, n - 1, x))
    # aggregate polynomials really interpolate L.s, R.s, O.s: sum_i s_i u_i(x) == A(x)
    assert Ax == sum(ui * si for ui, si in zip(u, sol)) % R
    assert Bx == sum(vi * si for vi, si in zip(v, sol)) % R
    assert (Ax * Bx - Cx) % R == hx * zx % R  # QAP identity (pinocchio_test.go:147-155)
    a = (Ax + r * delta + alpha) % R
    b = (Bx + s * delta + beta) % R
    assert proof.A == one(co.G1, a)
    assert proof.B == one(co.G2, b)
    # sum_i (w_i + beta u_i + alpha v_i) sol_i / delta: one inversion for the whole sum
    res = sum((w[i] + beta * u[i] + alpha * v[i]) * sol[i] for i in range(diff, m)) % R
    res = pr.fr_div((res + hx * zx) % R, delta)
    cd = (res + s * a + r * b - r * s % R * delta) % R
    assert proof.C == one(co.G1, cd)
    progress("Groth16 discrete-log checks")

    # ---- PHGR13 on the same QAP (BASELINE config #5 shape): device setup, prove, discrete-log checks of
    # every proof element from the retained toxic waste (pinocchio_test.go:33-146), pairing verification
    tox = [rng.fr() for _ in range(8)]
    sp, av, aw, ay, rv, rw, pbeta, pgamma = tox
    del tr, vk, proof
    t0 = time.time()
    ek, pvk = ps_api.NewPHGR13TrustedSetup(q, *tox)
    ctx.sync()
    t["phgr13_device_setup_s"] = time.time() - t0
    ctx.set_tables(False)
    ps_api.PHGR13Prove(ek, q, dsol)  # warm-up
    t0 = time.time()
    plain_pp = ps_api.PHGR13Prove(ek, q, dsol)
    t["phgr13_prove_plain_plan_s"] = time.time() - t0
    ctx.set_tables(True)
    ps_api.PHGR13Prove(ek, q, dsol)
    assert ek.vas.table_window > 0 and ek.ws.table_window > 0
    t0 = time.time()
    pp = ps_api.PHGR13Prove(ek, q, dsol)
    t["phgr13_prove_s"] = time.time() - t0
    mono_ek = ek.monomial_only()
    ps_api.PHGR13Prove(mono_ek, q, dsol)
    t0 = time.time()
    mono_pp = ps_api.PHGR13Prove(mono_ek, q, dsol)
    t["phgr13_prove_monomial_key_s"] = time.time() - t0
    for f in ps_api.PHGR13Proof.FIELDS:
        assert getattr(pp, f) == getattr(plain_pp, f), f
        assert getattr(pp, f) == getattr(mono_pp, f), f
    t["phgr13_phase_ms"] = {k: round(v, 2) for k, v in ctx.last_prove_phase_ms().items()}
    progress("PHGR13 setup and proof")
    t0 = time.time()
    shp = ShardedPHGR13(ctx, None, 8, 0)
    folded_pp = shp.fold([shp.partials(ek, q, dsol, rank=g) for g in range(8)])
    t["phgr13_8_shares_one_gpu_s"] = time.time() - t0
    for f in ps_api.PHGR13Proof.FIELDS:
        assert getattr(folded_pp, f) == getattr(pp, f), f
    progress("PHGR13 proof from the shares of 8 ranks")
    us, vs_, ws_, zs = rs.var_poly_evals(c, sp)
    ry = rv * rw % R
    dot = lambda ev: sum(e * si for e, si in zip(ev[diff:], sol[diff:])) % R
    Vm, Wm, Ym = dot(us), dot(vs_), dot(ws_)
    assert pp.hs == one(co.G1, poly_eval_bytes(h, n - 1, sp))
    assert pp.vss == one(co.G1, rv * Vm % R) and pp.vass == one(co.G1, rv * av % R * Vm % R)
    assert pp.wss == one(co.G2, rw * Wm % R) and pp.wass == one(co.G1, rw * aw % R * Wm % R)
    assert pp.yss == one(co.G1, ry * Ym % R) and pp.yass == one(co.G1, ry * ay % R * Ym % R)
    assert pp.gz == one(co.G1, pbeta * (rv * Vm + rw * Wm + ry * Ym) % R)
    progress("PHGR13 discrete-log checks")
    t0 = time.time()
    io = up(sol[:diff])
    io_arrays = (pvk.vs.slice(0, diff), pvk.ws.slice(0, diff), pvk.ys.slice(0, diff))
    assert ps_api.PHGR13Verify(ctx, pvk.fixed_points(), *io_arrays, pp, io)
    t["phgr13_verify_s"] = time.time() - t0
    pp.gz = one(co.G1, 12345)
    assert not ps_api.PHGR13Verify(ctx, pvk.fixed_points(), *io_arrays, pp, io)
    print("SCALE " + json.dumps({"log2n": log2n, "n_vars": m, **{k: (round(val, 4) if isinstance(val, float) else val) for k, val in t.items()}}))


@pytest.mark.parametrize("log2n", _sizes())
def test_provers_on_a_booleanity_circuit_with_an_int64_witness(ps_api, ctx, co, pr, log2n):
    """The regime the reference itself lives in: Vector = []int (algebra.go:13).  2^k booleanity gates
    b*b = b, a witness of random bits uploaded as int64: the solution sums take the short-scalar plan
    (and the heavy-bucket path: half the entries fall into one bucket).  Checks: both proofs are accepted
    by the verifiers (pairings), rejected when tampered with, and byte-identical to the proofs obtained
    from the same witness uploaded as 32-byte field elements (the 255-bit plan)."""
    from oracle import restate as rs

    n = 1 << log2n
    rng = pr.SplitMix64(SEED + 3030)
    c, wit = rs.bit_circuit(n)
    diff = c.nbVars - c.nbIO
    q = ps_api.QAP(ctx, c.nbVars, c.nbIO, c.left, c.right, c.out)
    tox = [rng.fr() for _ in range(5)]
    ptox = [rng.fr() for _ in range(8)]
    tr, vk = ps_api.NewGroth16TrustedSetup(q, *tox)
    ek, pvk = ps_api.NewPHGR13TrustedSetup(q, *ptox)
    sol_i64 = ps_api.Poly.from_values(ctx, wit)
    sol_fr = ps_api.Poly.upload(ctx, wit)
    r, s = rng.fr(), rng.fr()
    t = {}
    out = {}
    for name, sol in (("int64", sol_i64), ("fr", sol_fr)):
        ps_api.Groth16Prove(tr, q, sol, r, s)  # warm-up
        t0 = time.time()
        g = ps_api.Groth16Prove(tr, q, sol, r, s)
        t[f"groth16_prove_{name}_s"] = time.time() - t0
        ps_api.PHGR13Prove(ek, q, sol)
        t0 = time.time()
        p = ps_api.PHGR13Prove(ek, q, sol)
        t[f"phgr13_prove_{name}_s"] = time.time() - t0
        out[name] = (g, p)
    g, p = out["int64"]
    g2, p2 = out["fr"]
    assert (g.A, g.B, g.C) == (g2.A, g2.B, g2.C)
    for f in ps_api.PHGR13Proof.FIELDS:
        assert getattr(p, f) == getattr(p2, f), f
    io = ps_api.Poly.from_values(ctx, wit[:diff])
    verify16 = lambda proof: ps_api.Groth16Verify(ctx, tr.Alpha, tr.Beta2, vk["Gamma"], tr.Delta2, vk["IoLP"], proof, io)
    assert verify16(g)
    assert not verify16(ps_api.Groth16Proof(r, s, g.A, g.B, co.G1.to_b(co.G1.mul(777))))
    io_arrays = (pvk.vs.slice(0, diff), pvk.ws.slice(0, diff), pvk.ys.slice(0, diff))
    assert ps_api.PHGR13Verify(ctx, pvk.fixed_points(), *io_arrays, p, io)
    p.vss = co.G1.to_b(co.G1.mul(777))
    assert not ps_api.PHGR13Verify(ctx, pvk.fixed_points(), *io_arrays, p, io)
    print("SCALE-BITS " + json.dumps({"log2n": log2n, **{k: round(v, 4) for k, v in t.items()}}))
