// host_limb_check.cpp -- machine check of the lazy-limb contract of playsnark_amd/csrc/field.hpp / curve.hpp and of the
// host-side pairing arithmetic (pairing_math.inc), compiled FOR THE HOST with
//     g++ -std=c++17 -O1 -fsanitize=address,undefined -fno-sanitize-recover=all -pthread
// so that every signed 64-bit column sum, every shift and every array access of the product's field code runs under
// UBSan / ASan (GPU sanitizers are not available on the pool; the arithmetic is the same source on both sides).
//
// The contract (field.hpp): limbs are 28-bit, signed and lazy; "class c" means |limb| < c * 2^28.
//   f_mul(a, b)                 needs class(a) class(b) <= 8
//   f_mul2sub / f_mul2add       need  class(a) class(b) + class(c) class(d) <= 8
//   f_mul2add2sub               needs the four class products to add up to <= 8
//   fr_mul(a, b)                needs class(a) class(b) <= 11
// Each is driven with WORST-CASE operands of every admissible class combination (all limbs at +-(c 2^28 - 1), equal
// and alternating signs -- the extreme column sums) and with random lazy operands, and compared, after canonical
// reduction, with the same product on reduced operands; results must land in the documented output range.  Then the
// group law (madd / add / dbl chains, limb class of every stored coordinate tracked), the NTT butterfly sequences
// exactly as k_ntt_pass issues them (forward: no reduction for 32 stages; inverse: 16 unscaled stages), and the
// Miller loop + final exponentiation (bilinearity).  Exit code 0 = all checks passed.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <vector>

#include "../playsnark_amd/csrc/curve.hpp"

using namespace ps;

// what pairing_math.inc expects from msm.hpp / capi.hip
template <class F> static bool affine_is_identity(const Affine<F>& p) { return fp_all_zero(p.x) && fp_all_zero(p.y); }
namespace ps {
#include "../playsnark_amd/csrc/hostfield.inc"
}
#define PS_HOSTFIELD 1
#include "../playsnark_amd/csrc/pairing_math.inc"
template <class F> static Affine<typename HostField<F>::type> affine_to_host(const Affine<F>& a) {
    Affine<typename HostField<F>::type> r;
    r.x = to_host(a.x); r.y = to_host(a.y);
    return r;
}

static unsigned long long rng_state = 0x706c6179736e6172ull;
static unsigned long long rnd() {  // splitmix64
    unsigned long long z = (rng_state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
static int failures = 0;
#define CHECK(cond, ...)                                                            \
    do {                                                                            \
        if (!(cond)) {                                                              \
            failures++;                                                             \
            std::fprintf(stderr, "FAIL %s:%d: %s  ", __FILE__, __LINE__, #cond);    \
            std::fprintf(stderr, __VA_ARGS__);                                      \
            std::fprintf(stderr, "\n");                                             \
        }                                                                           \
    } while (0)

static int limb_class(const Fp& a) {  // smallest c with |l[i]| < c * 2^28 for all i
    long long m = 0;
    for (int i = 0; i < FP_L; i++) { long long v = a.l[i] < 0 ? -(long long)a.l[i] : a.l[i]; if (v > m) m = v; }
    return (int)(m >> 28) + 1;
}
static int limb_class(const Fr& a) {
    long long m = 0;
    for (int i = 0; i < FR_L; i++) { long long v = a.l[i] < 0 ? -(long long)a.l[i] : a.l[i]; if (v > m) m = v; }
    return (int)(m >> 28) + 1;
}
// The contract has two parts: the limb class, and |value| <= 16 p.  p has 381 bits, so the top limb (bits 364..) of an
// admissible value is at most 16 * p_top; the thirteen limbs below it are pushed to the edge of the class.
static const long long TOP_SPAN = 15ll * fp_mod28(FP_L - 1);
static Fp worst(int cls, int pattern) {  // pattern 0: all +, 1: all -, 2: alternating, 3: random signs
    Fp r;
    for (int i = 0; i < FP_L; i++) {
        i32 v = i == FP_L - 1 ? (i32)TOP_SPAN : (i32)(((long long)cls << 28) - 1);
        bool neg = pattern == 1 || (pattern == 2 && (i & 1)) || (pattern == 3 && (rnd() & 1));
        r.l[i] = neg ? -v : v;
    }
    return r;
}
static Fp random_lazy(int cls) {
    Fp r;
    for (int i = 0; i < FP_L; i++) {
        long long span = i == FP_L - 1 ? TOP_SPAN : ((long long)cls << 28) - 1;
        r.l[i] = (i32)((long long)(rnd() % (unsigned long long)(2 * span + 1)) - span);
    }
    return r;
}
static Fp reduced(const Fp& a) { return f_mul(a, fp_one()); }  // same residue, value in (-p/8, 9p/8); class(a) <= 8
static bool same_residue(const Fp& a, const Fp& b) { return fp_all_zero(f_sub(fp_canon(reduced(a)), fp_canon(reduced(b)))); }
static bool same_limbs(const Fp& a, const Fp& b) {  // the column-parallel forms must give the SAME limbs, not just the same residue
    for (int i = 0; i < FP_L; i++) if (a.l[i] != b.l[i]) return false;
    return true;
}
static void check_mul_output(const Fp& r, const char* what) {
    for (int i = 0; i < FP_L - 1; i++) CHECK(r.l[i] >= 0 && r.l[i] < (1 << 28), "%s: limb %d = %d outside [0, 2^28)", what, i, r.l[i]);
    CHECK(r.l[FP_L - 1] > -(1 << 25) && r.l[FP_L - 1] < (1 << 27), "%s: top limb %d", what, r.l[FP_L - 1]);
}

static void test_fp_products() {
    int combos = 0;
    for (int ca = 1; ca <= 8; ca++)
        for (int cb = 1; ca * cb <= 8; cb++)
            for (int pa = 0; pa < 4; pa++)
                for (int pb = 0; pb < 4; pb++) {
                    Fp a = worst(ca, pa), b = worst(cb, pb);
                    Fp r = f_mul(a, b);
                    check_mul_output(r, "f_mul");
                    CHECK(same_residue(r, f_mul(reduced(a), reduced(b))), "f_mul classes %d x %d patterns %d %d", ca, cb, pa, pb);
                    CHECK(same_limbs(r, f_mul_ilp(a, b)), "f_mul_ilp != f_mul, classes %d x %d patterns %d %d", ca, cb, pa, pb);
                    combos++;
                }
    for (int it = 0; it < 2000; it++) {
        int ca = 1 + (int)(rnd() % 8), cb = 1 + (int)(rnd() % (8 / ca));
        Fp a = random_lazy(ca), b = random_lazy(cb);
        Fp r = f_mul(a, b);
        CHECK(same_limbs(r, f_mul_ilp(a, b)), "f_mul_ilp != f_mul on random classes %d x %d", ca, cb);
        check_mul_output(r, "f_mul random");
        CHECK(same_residue(r, f_mul(reduced(a), reduced(b))), "f_mul random classes %d x %d", ca, cb);
        if (ca * ca <= 8) CHECK(same_residue(f_sqr(a), f_mul(reduced(a), reduced(a))), "f_sqr class %d", ca);
    }
    // two and four products under one reduction
    for (int c1 = 1; c1 <= 7; c1++)
        for (int c2 = 1; c1 + c2 <= 8; c2++)
            for (int pat = 0; pat < 4; pat++) {
                // class products c1 and c2: (c1 x 1) and (1 x c2), and the square-ish splits where they exist
                Fp a = worst(c1, pat), b = worst(1, (pat + 1) & 3), c = worst(1, (pat + 2) & 3), d = worst(c2, (pat + 3) & 3);
                Fp want_sub = f_sub(f_mul(reduced(a), reduced(b)), f_mul(reduced(c), reduced(d)));
                Fp want_add = f_add(f_mul(reduced(a), reduced(b)), f_mul(reduced(c), reduced(d)));
                Fp rs = f_mul2sub(a, b, c, d), ra = f_mul2add(a, b, c, d);
                check_mul_output(rs, "f_mul2sub");
                check_mul_output(ra, "f_mul2add");
                CHECK(same_limbs(rs, f_mul2sub_ilp(a, b, c, d)), "f_mul2sub_ilp != f_mul2sub, class products %d + %d", c1, c2);
                {
                    const Fp xs[2] = {a, c}, ys[2] = {b, d};
                    const bool plus[2] = {false, false};
                    CHECK(same_limbs(ra, f_mulsum_ilp<2>(xs, ys, plus)), "f_mulsum_ilp<2> != f_mul2add, class products %d + %d", c1, c2);
                }
                CHECK(same_residue(rs, want_sub), "f_mul2sub class products %d + %d", c1, c2);
                CHECK(same_residue(ra, want_add), "f_mul2add class products %d + %d", c1, c2);
                combos++;
            }
    for (int it = 0; it < 500; it++) {  // four products: class products 2 + 2 + 2 + 2 = 8 and random splits
        int cls[4] = {2, 2, 2, 2};
        if (it & 1) { cls[0] = 1 + (int)(rnd() % 5); cls[1] = 1 + (int)(rnd() % (6 - cls[0])); cls[2] = 1; cls[3] = 8 - cls[0] - cls[1] - 1; if (cls[3] < 1) cls[3] = 1; }
        Fp x[8];
        for (int k = 0; k < 4; k++) { x[2 * k] = it < 64 ? worst(cls[k], it & 3) : random_lazy(cls[k]); x[2 * k + 1] = it < 64 ? worst(1, (it >> 2) & 3) : random_lazy(1); }
        Fp r = f_mul2add2sub(x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]);
        check_mul_output(r, "f_mul2add2sub");
        {
            const Fp xs[4] = {x[0], x[2], x[4], x[6]}, ys[4] = {x[1], x[3], x[5], x[7]};
            const bool sg[4] = {false, false, true, true};
            CHECK(same_limbs(r, f_mulsum_ilp<4>(xs, ys, sg)), "f_mulsum_ilp<4> != f_mul2add2sub, classes %d %d %d %d", cls[0], cls[1], cls[2], cls[3]);
        }
        Fp want = f_sub(f_add(f_mul(reduced(x[0]), reduced(x[1])), f_mul(reduced(x[2]), reduced(x[3]))),
                        f_add(f_mul(reduced(x[4]), reduced(x[5])), f_mul(reduced(x[6]), reduced(x[7]))));
        CHECK(same_residue(r, want), "f_mul2add2sub classes %d %d %d %d", cls[0], cls[1], cls[2], cls[3]);
    }
    // f_norm: value unchanged, class back to ~1; f_is_zero on multiples of p inside its contract
    for (int it = 0; it < 500; it++) {
        Fp a = random_lazy(1 + (int)(rnd() % 8));
        Fp nrm = f_norm(a);
        CHECK(limb_class(nrm) <= 2, "f_norm left class %d", limb_class(nrm));
        CHECK(same_residue(a, nrm), "f_norm changed the value");
    }
    Fp kp = fp_zero();
    for (int k = -7; k <= 7; k++) {  // k p limb by limb stays inside class 8 for |k| <= 7
        Fp v;
        for (int i = 0; i < FP_L; i++) v.l[i] = (i32)((long long)k * fp_mod28(i));
        CHECK(f_is_zero(v), "f_is_zero(%d p)", k);
        v.l[0] += 1;
        CHECK(!f_is_zero(v), "f_is_zero(%d p + 1)", k);
    }
    (void)kp;
    std::printf("fp products: %d worst-case class combinations and 3000 random cases ok\n", combos);
}

// Fr: |value| < 64 r (the storage discipline of field.hpp) bounds the top limb; a class above 7 does not fit an i32 limb
static Fr fr_worst(int cls, int pattern) {
    Fr r;
    for (int i = 0; i < FR_L; i++) {
        i32 v = i == FR_L - 1 ? (i32)(63ll * fr_mod28(FR_L - 1)) : (i32)(((long long)cls << 28) - 1);
        bool neg = pattern == 1 || (pattern == 2 && (i & 1)) || (pattern == 3 && (rnd() & 1));
        r.l[i] = neg ? -v : v;
    }
    return r;
}
static Fr fr_random_canon() {
    u32 w[8];
    for (int i = 0; i < 8; i++) w[i] = (u32)rnd();
    w[7] &= 0x3fffffffu;  // < 2^254 < r
    return fr_to_mont(fr_from_words8(w));
}
static bool fr_same(const Fr& a, const Fr& b) {
    Fr d = fr_canon(fr_reduce(fr_sub(fr_reduce(a), fr_reduce(b))));
    for (int i = 0; i < FR_L; i++) if (d.l[i]) return false;
    return true;
}
static void test_fr_and_butterflies() {
    for (int ca = 1; ca <= 7; ca++)
        for (int cb = 1; cb <= 7 && ca * cb <= 11; cb++)
            for (int pat = 0; pat < 16; pat++) {
                Fr a = fr_worst(ca, pat & 3), b = fr_worst(cb, pat >> 2);
                Fr r = fr_mul(a, b);
                for (int i = 0; i < FR_L - 1; i++) CHECK(r.l[i] >= 0 && r.l[i] < (1 << 28), "fr_mul limb %d = %d", i, r.l[i]);
                CHECK(fr_same(r, fr_mul(fr_reduce(a), fr_reduce(b))), "fr_mul classes %d x %d", ca, cb);
            }
    // Forward (Cooley-Tukey) butterflies as k_ntt_pass issues them: a' = norm(a + w b), b' = norm(a - w b), no reduction
    // between stages; 32 stages is the deepest transform the field supports.  Checked against the same recurrence with a
    // canonical reduction after every stage.
    const int N = 64;
    std::vector<Fr> lazy(N), ref(N), tw(N);
    for (int i = 0; i < N; i++) { lazy[i] = ref[i] = fr_random_canon(); tw[i] = fr_random_canon(); }
    int worst_class = 0;
    for (int stage = 0; stage < 32; stage++) {
        const int half = 1 << (stage % 6);
        for (int i = 0; i < N; i++) {
            if (i & half) continue;
            Fr wb = fr_mul(lazy[i | half], tw[(i + stage) % N]);
            Fr a = lazy[i];
            lazy[i] = fr_norm(fr_add(a, wb));
            lazy[i | half] = fr_norm(fr_sub(a, wb));
            Fr wr = fr_mul(ref[i | half], tw[(i + stage) % N]);
            Fr ar = ref[i];
            ref[i] = fr_canon(fr_reduce(fr_add(ar, wr)));
            ref[i | half] = fr_canon(fr_reduce(fr_sub(ar, wr)));
        }
        for (int i = 0; i < N; i++) { int c = limb_class(lazy[i]); if (c > worst_class) worst_class = c; }
    }
    for (int i = 0; i < N; i++) CHECK(fr_same(lazy[i], ref[i]), "forward butterflies diverged at %d", i);
    CHECK(worst_class <= 3, "forward butterflies reached limb class %d", worst_class);
    // Inverse (Gentleman-Sande): a' = norm(a + b), b' = norm(a - b) * w; the all-sums path doubles per stage and ntt_run
    // divides the factor out after at most 16 stages.
    for (int i = 0; i < N; i++) lazy[i] = ref[i] = fr_random_canon();
    for (int stage = 0; stage < 16; stage++) {
        const int half = 1 << (stage % 6);
        for (int i = 0; i < N; i++) {
            if (i & half) continue;
            Fr a = lazy[i], b = lazy[i | half];
            lazy[i] = fr_norm(fr_add(a, b));
            lazy[i | half] = fr_mul(fr_norm(fr_sub(a, b)), tw[(i + stage) % N]);
            Fr ar = ref[i], br = ref[i | half];
            ref[i] = fr_canon(fr_reduce(fr_add(ar, br)));
            ref[i | half] = fr_canon(fr_reduce(fr_mul(fr_canon(fr_reduce(fr_sub(ar, br))), tw[(i + stage) % N])));
        }
    }
    Fr sc = fr_random_canon();
    for (int i = 0; i < N; i++) CHECK(fr_same(fr_mul(lazy[i], sc), fr_mul(ref[i], sc)), "inverse butterflies diverged at %d", i);
    std::printf("fr products and NTT butterfly sequences ok (forward: limb class <= %d after 32 stages)\n", worst_class);
}

template <class F> static int coord_class(const F& a);
template <> int coord_class<Fp>(const Fp& a) { return limb_class(a); }
template <> int coord_class<Fp2>(const Fp2& a) { int x = limb_class(a.c0), y = limb_class(a.c1); return x > y ? x : y; }
static Affine<Fp> gen_of(const Fp*) {
    Affine<Fp> g;
    constexpr i32 x[FP_L] = PS_G1_GEN28_X; constexpr i32 y[FP_L] = PS_G1_GEN28_Y;
    for (int i = 0; i < FP_L; i++) { g.x.l[i] = x[i]; g.y.l[i] = y[i]; }
    return g;
}
static Affine<Fp2> gen_of(const Fp2*) {
    Affine<Fp2> g;
    constexpr i32 x0[FP_L] = PS_G2_GEN28_X0; constexpr i32 x1[FP_L] = PS_G2_GEN28_X1;
    constexpr i32 y0[FP_L] = PS_G2_GEN28_Y0; constexpr i32 y1[FP_L] = PS_G2_GEN28_Y1;
    for (int i = 0; i < FP_L; i++) { g.x.c0.l[i] = x0[i]; g.x.c1.l[i] = x1[i]; g.y.c0.l[i] = y0[i]; g.y.c1.l[i] = y1[i]; }
    return g;
}
template <class F>
static Affine<F> small_multiple(u32 k) {
    Affine<F> g = gen_of((const F*)0), r;
    Xyzz<F> t = xyzz_mul_small<F>(xyzz_from_affine<F>(g.x, g.y), k);
    if (!xyzz_to_affine<F>(t, r.x, r.y)) { r.x = f_zero((const F*)0); r.y = f_zero((const F*)0); }
    return r;
}
template <class F>
static void test_group_law(const char* name) {
    // a bucket's life: a long chain of mixed additions (with repeats and P, -P pairs), partial sums added together,
    // doublings -- the stored coordinates must stay at limb class <= 2 and the result must not depend on the order
    const int K = 40;
    std::vector<Affine<F>> pts;
    std::vector<u32> ks;
    for (int i = 0; i < K; i++) { u32 k = 1 + (u32)(rnd() % 5000); if (i % 7 == 3) k = ks[i - 1]; ks.push_back(k); pts.push_back(small_multiple<F>(k)); }
    unsigned long long total = 0;
    Xyzz<F> acc = xyzz_identity<F>(), left = xyzz_identity<F>(), right = xyzz_identity<F>();
    int worst_cls = 0;
    for (int i = 0; i < K; i++) {
        xyzz_madd<F>(acc, pts[i].x, pts[i].y);
        xyzz_madd<F>(i < K / 2 ? left : right, pts[i].x, pts[i].y);
        total += ks[i];
        int c = std::max(std::max(coord_class<F>(acc.x), coord_class<F>(acc.y)), std::max(coord_class<F>(acc.zz), coord_class<F>(acc.zzz)));
        if (c > worst_cls) worst_cls = c;
    }
    xyzz_add<F>(left, right);
    F ax, ay, bx, by, cx, cy;
    CHECK(xyzz_to_affine<F>(acc, ax, ay) && xyzz_to_affine<F>(left, bx, by), "%s: chain ended in the identity", name);
    CHECK(fp_all_zero(f_sub(ax, bx)) && fp_all_zero(f_sub(ay, by)), "%s: (sum of halves) != serial sum", name);
    Affine<F> g = gen_of((const F*)0);
    u32 kw[8] = {(u32)total, (u32)(total >> 32), 0, 0, 0, 0, 0, 0};
    Xyzz<F> direct = xyzz_mul_scalar<F>(xyzz_from_affine<F>(g.x, g.y), kw);
    CHECK(xyzz_to_affine<F>(direct, cx, cy) && fp_all_zero(f_sub(ax, cx)) && fp_all_zero(f_sub(ay, cy)), "%s: chain != [sum k] G", name);
    CHECK(worst_cls <= 2, "%s: stored coordinates reached limb class %d", name, worst_cls);
    // P + (-P) and P + P through the mixed adder
    Xyzz<F> t = xyzz_from_affine<F>(pts[0].x, pts[0].y);
    xyzz_madd<F>(t, pts[0].x, f_neg(pts[0].y));
    CHECK(xyzz_is_identity(t), "%s: P + (-P) is not the identity", name);
    t = xyzz_from_affine<F>(pts[0].x, pts[0].y);
    xyzz_madd<F>(t, pts[0].x, pts[0].y);
    Xyzz<F> d = xyzz_dbl<F>(xyzz_from_affine<F>(pts[0].x, pts[0].y));
    F dx, dy, ex, ey;
    CHECK(xyzz_to_affine<F>(t, dx, dy) && xyzz_to_affine<F>(d, ex, ey) && fp_all_zero(f_sub(dx, ex)) && fp_all_zero(f_sub(dy, ey)), "%s: P + P != 2P", name);
    std::printf("%s group law ok (stored coordinates at limb class <= %d over %d mixed additions)\n", name, worst_cls, K);
}

// The pairing over one representation: NS = pairing_dev (Fp / Fp2, the lazy limbs) or pairing (Fq / Fq2, what the
// verifiers run); CONV brings the device-representation test points into it.
#define PAIRING_TESTS(NAME, NS, CONV1, CONV2, LABEL)                                                                              \
    static void NAME() {                                                                                                          \
        /* e(aP, bQ) == e(abP, Q) and != e((ab + 1)P, Q) */                                                                       \
        auto P = CONV1(small_multiple<Fp>(6)), P210 = CONV1(small_multiple<Fp>(210)), P211 = CONV1(small_multiple<Fp>(211));      \
        auto Q35 = CONV2(small_multiple<Fp2>(35)), Q1 = CONV2(small_multiple<Fp2>(1));                                            \
        CHECK(NS::pairing_product_is_one({{P, Q35}, {NS::neg_g1(P210), Q1}}), LABEL ": e(6P, 35Q) != e(210P, Q)");                \
        CHECK(!NS::pairing_product_is_one({{P, Q35}, {NS::neg_g1(P211), Q1}}), LABEL ": e(6P, 35Q) == e(211P, Q)");               \
        /* the fast path against the literal one: the projective Miller loop differs from the affine one only by factors in  */  \
        /* proper subfields (gone after the exponentiation), and final_exp is the cube of the plain (p^12 - 1) / r power      */  \
        const unsigned ks[3][2] = {{6, 35}, {1, 1}, {1234567, 89}};                                                               \
        for (auto& k : ks) {                                                                                                      \
            auto A = CONV1(small_multiple<Fp>(k[0]));                                                                             \
            auto B = CONV2(small_multiple<Fp2>(k[1]));                                                                            \
            NS::Fp12 slow = NS::final_exp_generic(NS::miller_affine(A, B));                                                       \
            NS::Fp12 mid = NS::final_exp_generic(NS::miller(A, B));                                                               \
            CHECK(NS::f12_eq(slow, mid), LABEL ": projective Miller loop != affine Miller loop after the exponentiation");        \
            NS::Fp12 fast = NS::final_exp(NS::miller(A, B));                                                                      \
            CHECK(NS::f12_eq(fast, NS::f12_mul(slow, NS::f12_mul(slow, slow))), LABEL ": final_exp != (plain exponentiation)^3"); \
            CHECK(!NS::f12_eq(fast, NS::f12_one()), LABEL ": degenerate pairing value");                                          \
        }                                                                                                                         \
        /* Frobenius: twelve applications are the identity, and f12_sqr agrees with the product */                                \
        NS::Fp12 f = NS::miller(P, Q35), g = f;                                                                                   \
        for (int i = 0; i < 12; i++) g = NS::f12_frob(g);                                                                         \
        CHECK(NS::f12_eq(f, g), LABEL ": frobenius^12 is not the identity");                                                      \
        CHECK(NS::f12_eq(NS::f12_sqr(f), NS::f12_mul(f, f)), LABEL ": f12_sqr != f * f");                                         \
        std::printf(LABEL ": pairing bilinearity ok, fast pairing path == literal path\n");                                       \
    }
#define IDENT(x) (x)
PAIRING_TESTS(test_pairing_dev, pairing_dev, IDENT, IDENT, "device representation")
PAIRING_TESTS(test_pairing_host, pairing, affine_to_host<Fp>, affine_to_host<Fp2>, "host field")

// hostfield.inc against field.hpp: conversions round-trip, every operation commutes with them, and the group law gives
// the same points in both representations
static void test_host_field() {
    for (int it = 0; it < 2000; it++) {
        Fp a = random_lazy(1 + it % 3), b = random_lazy(1 + (it / 3) % 2);  // any lazy value: to_host reduces first
        Fq ha = to_host(a), hb = to_host(b);
        CHECK(same_residue(to_device(ha), a), "to_device(to_host(a)) != a");
        Fp an = f_norm(a), bn = f_norm(b);
        CHECK(f_eq(to_host(f_mul(an, bn)), f_mul(ha, hb)), "host product != device product");
        CHECK(f_eq(to_host(f_norm(f_add(a, b))), f_add(ha, hb)), "host sum != device sum");
        CHECK(f_eq(to_host(f_norm(f_sub(a, b))), f_sub(ha, hb)), "host difference != device difference");
        CHECK(f_eq(to_host(f_norm(f_neg(a))), f_neg(ha)), "host negation != device negation");
        if (it < 20 && !f_is_zero(ha)) CHECK(f_eq(f_mul(f_inv(ha), ha), f_one((const Fq*)0)), "host inverse");
        Fp2 x = Fp2{an, bn}, y = Fp2{bn, f_norm(f_add(an, bn))};
        CHECK(f_eq(to_host(f_mul(x, y)), f_mul(to_host(x), to_host(y))), "host Fq2 product != device Fp2 product");
        CHECK(f_eq(to_host(f_sqr(x)), f_sqr(to_host(x))), "host Fq2 square != device Fp2 square");
    }
    CHECK(f_is_zero(to_host(fp_zero())) && f_eq(to_host(fp_one()), f_one((const Fq*)0)), "zero / one");
    uint8_t b1[48], b2[48];
    Fp g = gen_of((const Fp*)0).x;
    fp_to_be48(b1, g);
    fq_to_be48(b2, to_host(g));
    CHECK(!memcmp(b1, b2, 48), "wire bytes differ between the representations");
    // [k]G in both representations
    u32 kw[8] = {0x12345678u, 0x9abcdef0u, 0x0fedcba9u, 0x87654321u, 0x13579bdfu, 0x2468ace0u, 0x0badf00du, 0x0000beefu};
    Affine<Fp> g1 = gen_of((const Fp*)0);
    Affine<Fp2> g2 = gen_of((const Fp2*)0);
    Fp dx, dy; Fq hx, hy;
    bool ok1 = xyzz_to_affine<Fp>(xyzz_mul_scalar<Fp>(xyzz_from_affine<Fp>(g1.x, g1.y), kw), dx, dy);
    Affine<Fq> hg1 = affine_to_host<Fp>(g1);
    bool ok2 = xyzz_to_affine<Fq>(xyzz_mul_scalar<Fq>(xyzz_from_affine<Fq>(hg1.x, hg1.y), kw), hx, hy);
    CHECK(ok1 && ok2 && f_eq(to_host(dx), hx) && f_eq(to_host(dy), hy), "[k]G differs between the representations (G1)");
    Fp2 ex, ey; Fq2 kx, ky;
    ok1 = xyzz_to_affine<Fp2>(xyzz_mul_scalar<Fp2>(xyzz_from_affine<Fp2>(g2.x, g2.y), kw), ex, ey);
    Affine<Fq2> hg2 = affine_to_host<Fp2>(g2);
    ok2 = xyzz_to_affine<Fq2>(xyzz_mul_scalar<Fq2>(xyzz_from_affine<Fq2>(hg2.x, hg2.y), kw), kx, ky);
    CHECK(ok1 && ok2 && f_eq(to_host(ex), kx) && f_eq(to_host(ey), ky), "[k]G differs between the representations (G2)");
    std::printf("host field == device field\n");
}

int main() {
    test_fp_products();
    test_fr_and_butterflies();
    test_group_law<Fp>("G1");
    test_group_law<Fp2>("G2");
    test_host_field();
    test_pairing_dev();
    test_pairing_host();
    if (failures) { std::fprintf(stderr, "%d check(s) failed\n", failures); return 1; }
    std::printf("host_limb_check ok\n");
    return 0;
}
