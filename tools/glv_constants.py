"""Constants of the GLV split used by csrc/lagrange.hpp (ec_mul_glv), derived and checked with integers only.

BLS12-381: r = z^4 - z^2 + 1, so lambda = z^2 - 1 satisfies lambda^2 + lambda + 1 = 0 (mod r), and the curve (j = 0) and its
twist have the endomorphism phi(x, y) = (beta x, y), beta a primitive cube root of unity in Fp, acting on the subgroup of
order r as multiplication by lambda or by lambda^2 -- which one depends on beta, found here by comparing phi(G) with lambda G
on both generators.  Prints the values the header hard-codes (28-bit Montgomery limbs of beta are made by the library's own
conversion at start-up: only the plain integer is pinned)."""
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
Z = 0xD201000000010000
G1 = (0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
      0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1)
G2 = ((0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
       0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
      (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
       0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE))


class F1:
    zero, one = 0, 1
    add = staticmethod(lambda a, b: (a + b) % P)
    sub = staticmethod(lambda a, b: (a - b) % P)
    mul = staticmethod(lambda a, b: a * b % P)
    inv = staticmethod(lambda a: pow(a, P - 2, P))
    smul = staticmethod(lambda a, k: a * k % P)


class F2:
    zero, one = (0, 0), (1, 0)
    add = staticmethod(lambda a, b: ((a[0] + b[0]) % P, (a[1] + b[1]) % P))
    sub = staticmethod(lambda a, b: ((a[0] - b[0]) % P, (a[1] - b[1]) % P))
    mul = staticmethod(lambda a, b: ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P))
    smul = staticmethod(lambda a, k: (a[0] * k % P, a[1] * k % P))

    @staticmethod
    def inv(a):
        d = pow(a[0] * a[0] + a[1] * a[1], P - 2, P)
        return (a[0] * d % P, -a[1] * d % P)


def ec_add(F, p, q):
    if p is None: return q
    if q is None: return p
    if p[0] == q[0]:
        if p[1] != q[1] or p[1] == F.zero: return None
        x2 = F.mul(p[0], p[0])
        lam = F.mul(F.add(F.add(x2, x2), x2), F.inv(F.add(p[1], p[1])))
    else:
        lam = F.mul(F.sub(q[1], p[1]), F.inv(F.sub(q[0], p[0])))
    x3 = F.sub(F.sub(F.mul(lam, lam), p[0]), q[0])
    return (x3, F.sub(F.mul(lam, F.sub(p[0], x3)), p[1]))


def ec_mul(F, k, p):
    acc = None
    for bit in bin(k)[2:]:
        acc = ec_add(F, acc, acc)
        if bit == "1": acc = ec_add(F, acc, p)
    return acc


lam = Z * Z - 1
assert (lam * lam + lam + 1) % R == 0 and R == Z**4 - Z**2 + 1
# a primitive cube root of unity in Fp
g = 2
while pow(g, (P - 1) // 3, P) == 1: g += 1
b0 = pow(g, (P - 1) // 3, P)
assert b0 != 1 and pow(b0, 3, P) == 1
out = {}
for name, F, G in (("G1", F1, G1), ("G2", F2, G2)):
    want = ec_mul(F, lam, G)
    for beta in (b0, b0 * b0 % P):
        if (F.smul(G[0], beta), G[1]) == want:
            out[name] = beta
    assert name in out, name
print("lambda = 0x%x  (%d bits)" % (lam, lam.bit_length()))
print("r // lambda = 0x%x  (%d bits)" % (R // lam, (R // lam).bit_length()))
for name in ("G1", "G2"):
    print("beta(%s) = 0x%096x" % (name, out[name]))
print("same beta" if out["G1"] == out["G2"] else "different betas: beta(G2) = beta(G1)^2")
