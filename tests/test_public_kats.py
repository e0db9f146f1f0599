"""Known answers that NONE of this repository's code generated.

The reference's curve arithmetic lives in absent third-party Go modules and its tests hold no point vectors, so the
oracle's curve layer cannot be pinned by anything reference-held ("parity unpinned").  The next-best
pin: published BLS12-381 values, hard-coded below with their sources, checked against (1) integer arithmetic written
out in THIS file (curve equation, chord-and-tangent doubling, square roots by exponentiation), (2) the oracle's pure
Python twin, (3) the C oracle, (4) the product's host-side codec (ps_point_convert, ps_points_sum, ps_points_lincomb --
no device needed) and, on the GPU, (5) the fixed-base kernel and the MSM.

Sources
  [IETF]  draft-irtf-cfrg-pairing-friendly-curves, section 4.2.1 "BLS12-381": p, r, the generator coordinates of G1 and G2.
  [ZC]    ZCash protocol specification 5.4.9.2 / zkcrypto `bls12_381` serialisation notes: 48 / 96-byte compressed and
          96 / 192-byte uncompressed encodings, flag bits 0x80 (compressed) 0x40 (infinity) 0x20 (y lexicographically
          larger), Fp2 as c1 || c0.
  [ETH]   Ethereum consensus-layer BLS test vectors (and py_ecc's test-suite): the public keys of the secret keys 1, 2, 3
          are G1, 2 G1, 3 G1 compressed -- 97f1d3a7.., a572cbea.., 89ece308..; 2 G2 compressed is aa4edef9..
"""
import pytest

P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB   # [IETF]
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001                                     # [IETF]
# generator coordinates [IETF]
G1_X = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
G1_Y = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
G2_X0 = 0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8
G2_X1 = 0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E
G2_Y0 = 0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801
G2_Y1 = 0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE
# compressed multiples [ETH] / [ZC]
G1_1 = bytes.fromhex("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")
G1_2 = bytes.fromhex("a572cbea904d67468808c8eb50a9450c9721db309128012543902d0ac358a62ae28f75bb8f1c7c42c39a8c5529bf0f4e")
G1_3 = bytes.fromhex("89ece308f9d1f0131765212deca99697b112d61f9be9a5f1f3780a51335b3ff981747a0b2ca2179b96d2c0c9024e5224")
G2_1 = bytes.fromhex("93e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
                     "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8")
G2_2 = bytes.fromhex("aa4edef9c1ed7f729f520e47730a124fd70662a904ba1074728114d1031e1572c6c886f6b57ec72a6178288c47c33577"
                     "1638533957d540a9d2370f17cc7ed5863bc0b995b8825e0ee1ea1e1e4d00dbae81f14b0bf3611b78c952aacab827a053")
G1_INF_C, G1_INF_U = b"\xc0" + bytes(47), b"\x40" + bytes(95)   # [ZC]
G2_INF_C, G2_INF_U = b"\xc0" + bytes(95), b"\x40" + bytes(191)


# ---- arithmetic written out here: nothing below imports the oracle or the product ----
def _g1_decompress(b):
    assert len(b) == 48 and b[0] & 0x80 and not b[0] & 0x40
    x = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:], "big")
    y = pow((x * x * x + 4) % P, (P + 1) // 4, P)          # p = 3 mod 4
    assert y * y % P == (x * x * x + 4) % P, "x is not on the curve"
    if (y > (P - 1) // 2) != bool(b[0] & 0x20):
        y = P - y
    return x, y


def _g1_double(pt):
    x, y = pt
    lam = 3 * x * x * pow(2 * y, P - 2, P) % P
    x3 = (lam * lam - 2 * x) % P
    return x3, (lam * (x - x3) - y) % P


def _g1_add(p, q):
    lam = (q[1] - p[1]) * pow(q[0] - p[0], P - 2, P) % P
    x3 = (lam * lam - p[0] - q[0]) % P
    return x3, (lam * (p[0] - x3) - p[1]) % P


def _g1_unc(pt):
    return pt[0].to_bytes(48, "big") + pt[1].to_bytes(48, "big")


def _f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def _f2_inv(a):
    n = pow(a[0] * a[0] + a[1] * a[1], P - 2, P)
    return (a[0] * n % P, -a[1] * n % P)


def _g2_double(pt):
    x, y = pt
    x2 = _f2_mul(x, x)
    lam = _f2_mul((3 * x2[0] % P, 3 * x2[1] % P), _f2_inv((2 * y[0] % P, 2 * y[1] % P)))
    l2 = _f2_mul(lam, lam)
    x3 = ((l2[0] - 2 * x[0]) % P, (l2[1] - 2 * x[1]) % P)
    t = _f2_mul(lam, ((x[0] - x3[0]) % P, (x[1] - x3[1]) % P))
    return x3, ((t[0] - y[0]) % P, (t[1] - y[1]) % P)


def _g2_unc(pt):
    (x0, x1), (y0, y1) = pt
    return b"".join(v.to_bytes(48, "big") for v in (x1, x0, y1, y0))   # c1 || c0 [ZC]


def _g2_compress(pt):
    (x0, x1), (y0, y1) = pt
    larger = y1 > (P - 1) // 2 if y1 else y0 > (P - 1) // 2           # lexicographic: c1 first [ZC]
    b = bytearray(x1.to_bytes(48, "big") + x0.to_bytes(48, "big"))
    b[0] |= 0x80 | (0x20 if larger else 0)
    return bytes(b)


GEN1 = (G1_X, G1_Y)
GEN2 = ((G2_X0, G2_X1), (G2_Y0, G2_Y1))


def test_the_published_values_are_consistent_with_each_other():
    """The constants above against arithmetic written in this file: curve equations, encodings, doubling, 3G = 2G + G,
    group order.  Guards the test itself against a mistyped digit."""
    assert (G1_Y * G1_Y - G1_X ** 3 - 4) % P == 0
    x3 = _f2_mul(_f2_mul(GEN2[0], GEN2[0]), GEN2[0])
    y2 = _f2_mul(GEN2[1], GEN2[1])
    assert y2 == ((x3[0] + 4) % P, (x3[1] + 4) % P)                   # twist: y^2 = x^3 + 4(1 + u)
    z = -0xD201000000010000
    assert R == z ** 4 - z ** 2 + 1 and P == (z - 1) ** 2 * R // 3 + z  # the BLS12 family polynomials
    assert _g1_decompress(G1_1) == GEN1
    two, three = _g1_decompress(G1_2), _g1_decompress(G1_3)
    assert _g1_double(GEN1) == two and _g1_add(two, GEN1) == three
    assert _g2_compress(GEN2) == G2_1 and _g2_compress(_g2_double(GEN2)) == G2_2


def test_the_oracles_agree_with_the_published_values(co, pr):
    """pyref (the Python twin) and the C oracle: generator, 2G, 3G, -G, the identity, in both encodings, both groups."""
    two1, three1, two2 = _g1_decompress(G1_2), _g1_decompress(G1_3), _g2_double(GEN2)
    assert (pr.P, pr.R) == (P, R)
    assert pr.G1.gen == GEN1 and pr.G2.gen == GEN2
    for og in (pr.G1, co.G1):
        assert og.mul(1) == GEN1 and og.mul(2) == two1 and og.mul(3) == three1
        assert og.mul(R - 1) == (G1_X, P - G1_Y)
        assert og.add(og.mul(2), og.mul(1)) == three1
    for og in (pr.G2, co.G2):
        assert og.mul(1) == GEN2 and og.mul(2) == two2
    assert pr.G1.mul(R) is None and pr.G2.mul(R) is None
    for k, comp in ((1, G1_1), (2, G1_2), (3, G1_3)):
        pt = pr.G1.mul(k)
        assert pr.g1_compress(pt) == comp and co.g1_compress(pt) == comp
        assert pr.g1_decompress(comp) == pt and co.g1_decompress(comp) == pt
        assert pr.g1_to_bytes(pt) == _g1_unc(pt) == co.G1.to_b(pt)
    for k, comp in ((1, G2_1), (2, G2_2)):
        pt = pr.G2.mul(k)
        assert pr.g2_compress(pt) == comp and co.g2_compress(pt) == comp
        assert pr.g2_decompress(comp) == pt and co.g2_decompress(comp) == pt
        assert pr.g2_to_bytes(pt) == _g2_unc(pt) == co.G2.to_b(pt)
    neg = bytes([G1_1[0] | 0x20]) + G1_1[1:]                             # -G1: same x, the other sign bit
    assert neg.hex().startswith("b7f1d3a7") and pr.g1_compress((G1_X, P - G1_Y)) == neg
    assert pr.g1_compress(None) == G1_INF_C and pr.g2_compress(None) == G2_INF_C
    assert pr.g1_to_bytes(None) == G1_INF_U and pr.g2_to_bytes(None) == G2_INF_U
    assert pr.g1_decompress(G1_INF_C) is None and pr.g2_decompress(G2_INF_C) is None


def test_the_products_host_codec_agrees_with_the_published_values(ps_api):
    """ps_point_convert / ps_points_sum / ps_points_lincomb are host code of libplaysnark_hip.so: no device needed."""
    A, Cm = ps_api._lib.PS_FMT_AFFINE, ps_api._lib.PS_FMT_COMPRESSED
    two1, three1, two2 = _g1_decompress(G1_2), _g1_decompress(G1_3), _g2_double(GEN2)
    for comp, pt in ((G1_1, GEN1), (G1_2, two1), (G1_3, three1)):
        assert ps_api.point_convert(ps_api.G1, comp, Cm, A) == _g1_unc(pt)
        assert ps_api.point_convert(ps_api.G1, _g1_unc(pt), A, Cm) == comp
    for comp, pt in ((G2_1, GEN2), (G2_2, two2)):
        assert ps_api.point_convert(ps_api.G2, comp, Cm, A) == _g2_unc(pt)
        assert ps_api.point_convert(ps_api.G2, _g2_unc(pt), A, Cm) == comp
    assert ps_api.point_convert(ps_api.G1, G1_INF_C, Cm, A) == G1_INF_U
    assert ps_api.point_convert(ps_api.G2, G2_INF_U, A, Cm) == G2_INF_C
    g, g2 = _g1_unc(GEN1), _g2_unc(GEN2)
    assert ps_api.points_sum(ps_api.G1, g + g) == _g1_unc(two1)           # the exceptional P + P branch
    assert ps_api.points_sum(ps_api.G1, g + g + g) == _g1_unc(three1)
    assert ps_api.points_sum(ps_api.G2, g2 + g2) == _g2_unc(two2)
    assert ps_api.points_sum(ps_api.G1, g + _g1_unc((G1_X, P - G1_Y))) == G1_INF_U
    assert ps_api.points_lincomb(ps_api.G1, g + _g1_unc(two1), [1, 1]) == _g1_unc(three1)
    assert ps_api.points_lincomb(ps_api.G1, g, [R - 1]) == _g1_unc((G1_X, P - G1_Y))
    assert ps_api.points_lincomb(ps_api.G2, g2, [2]) == _g2_unc(two2)


@pytest.mark.gpu
def test_the_gpu_kernels_agree_with_the_published_values(ps_api, ctx):
    """Fixed-base kernel (k G for k = 0, 1, 2, 3, r - 1), compressed download, GPU decompression, and MSMs whose answers
    are published multiples: 1*G + 1*G = 2G (the bucket adder's doubling), 1*G + 1*(2G) = 3G, 2*G = 2G, (r-1)*G = -G."""
    A, Cm = ps_api._lib.PS_FMT_AFFINE, ps_api._lib.PS_FMT_COMPRESSED
    two1, three1, two2 = _g1_decompress(G1_2), _g1_decompress(G1_3), _g2_double(GEN2)
    pts = ps_api.Points.from_scalars(ctx, ps_api.G1, ps_api.Poly.upload(ctx, [0, 1, 2, 3, R - 1]))
    assert pts.download() == G1_INF_U + _g1_unc(GEN1) + _g1_unc(two1) + _g1_unc(three1) + _g1_unc((G1_X, P - G1_Y))
    assert pts.download_compressed() == G1_INF_C + G1_1 + G1_2 + G1_3 + bytes([G1_1[0] | 0x20]) + G1_1[1:]
    pts2 = ps_api.Points.from_scalars(ctx, ps_api.G2, ps_api.Poly.upload(ctx, [0, 1, 2]))
    assert pts2.download() == G2_INF_U + _g2_unc(GEN2) + _g2_unc(two2)
    assert pts2.download_compressed() == G2_INF_C + G2_1 + G2_2
    up = ps_api.Points.upload(ctx, ps_api.G1, G1_1 + G1_2 + G1_3, fmt=Cm)   # square roots on the GPU
    assert up.download() == _g1_unc(GEN1) + _g1_unc(two1) + _g1_unc(three1)
    up2 = ps_api.Points.upload(ctx, ps_api.G2, G2_1 + G2_2, fmt=Cm)
    assert up2.download() == _g2_unc(GEN2) + _g2_unc(two2)
    gg = ps_api.Points.upload(ctx, ps_api.G1, _g1_unc(GEN1) * 2)
    assert ps_api.Poly.upload(ctx, [1, 1]).BlindEval(gg) == _g1_unc(two1)
    assert ps_api.Poly.upload(ctx, [2, 1]).BlindEval(gg) == _g1_unc(three1)
    assert ps_api.Poly.upload(ctx, [R - 1, 0]).BlindEval(gg) == _g1_unc((G1_X, P - G1_Y))
    assert ps_api.Poly.upload(ctx, [1, R - 1]).BlindEval(gg) == G1_INF_U
    assert ps_api.Poly.upload(ctx, [1, 1, 0]).BlindEval(up) == _g1_unc(three1)
    assert ps_api.Poly.from_values(ctx, [3, -1]).BlindEval(gg) == _g1_unc(two1)
    gg2 = ps_api.Points.upload(ctx, ps_api.G2, _g2_unc(GEN2) * 2)
    assert ps_api.Poly.upload(ctx, [1, 1]).BlindEval(gg2) == _g2_unc(two2)
    assert ps_api.Poly.upload(ctx, [2, 0]).BlindEval(gg2) == _g2_unc(two2)
