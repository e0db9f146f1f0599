"""Go / no-go measurement for batched-AFFINE bucket additions: the first round of a pairwise reduction of
the sorted entries of a 2^20-point G1 sum over its window table -- 16 affine pair additions per thread, one field inversion
per thread shared by Montgomery's trick, the points gathered twice from HBM (ps_debug_pair_add_probe, csrc/capi.hip) --
timed against the mixed additions of k_accumulate on the same entries.  A sample of the sums is checked with plain integers.
  python3 tools/pair_add_probe.py [log2n]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
import bench  # noqa: E402
from playsnark_amd import api  # noqa: E402
from playsnark_amd._lib import lib  # noqa: E402

l = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << l
ctx = api.Context(0)
a_host = bench.uniform_scalars_be32(n, 91).tobytes()
pts = api.Points.from_scalars(ctx, api.G1, api.Poly.upload(ctx, a_host)).precompute(0)
sc = api.Poly.upload(ctx, bench.uniform_scalars_be32(n, 92).tobytes())
ctx.set_timing(True)
for _ in range(3):
    sc.BlindEval(pts)
acc_ms = ctx.last_stage_ms()["accumulate"]
entries = ctx.last_msm_info()["entries"]
ctx.set_timing(False)
lib.ps_debug_pair_add_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_char_p,
                                        C.POINTER(C.c_uint32)]
res = {}
for real in (1, 0):
    ms = C.c_float()
    ent = (C.c_uint32 * 32)()
    sums = C.create_string_buffer(96 * 16)
    info = (C.c_uint32 * 4)()
    api._check(lib.ps_debug_pair_add_probe(ctx._h, pts._h, sc._h, real, C.byref(ms), ent, sums, info))
    res[real] = ms.value
    if real:  # check thread 0's sixteen sums: entry = index | window << 26 | sign << 31 over the table 2^(c w) P_index, P_i = a_i G
        E, T, c, W = info
        fb = int.from_bytes
        ok = 0
        for i in range(16):
            dl = []
            for e in (ent[2 * i], ent[2 * i + 1]):
                idx, w, neg = e & ((1 << 26) - 1), (e >> 26) & 31, e >> 31
                k = fb(a_host[32 * idx:32 * idx + 32], "big") * (1 << (c * w)) % bench.R_MOD
                dl.append(-k % bench.R_MOD if neg else k)
            ok += sums.raw[96 * i:96 * i + 96] == bench.fixed_base_mul_bytes("g1", (dl[0] + dl[1]) % bench.R_MOD)
        print("sample: %d of 16 affine sums equal (k1 + k2) G by discrete logarithm" % ok)
        assert ok == 16
pairs = info[1] * 16
print("2^%d points, window %d: %d sorted entries; k_accumulate (mixed additions, 8M + 2S): %.3f ms = %.0f ps per addition" % (
    l, info[2], entries, acc_ms, acc_ms * 1e9 / entries))
print("pair-add round, %d affine additions (half of the entries), one inversion per 16: %.3f ms = %.0f ps per addition" % (
    pairs, res[1], res[1] * 1e9 / pairs))
print("the same with the inversion replaced by a copy (what a FREE inversion would leave): %.3f ms = %.0f ps per addition" % (
    res[0], res[0] * 1e9 / pairs))
