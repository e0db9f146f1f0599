/* TEST INFRASTRUCTURE ONLY -- see oracle.h for the status header ("parity unpinned" at byte
 * level; pinned by the reference's fixed Fr cases + algebraic identities + public constants).
 *
 * Plain C11 + unsigned __int128, 64-bit limbs, Jacobian coordinates, CIOS Montgomery with
 * constants *derived at load time* from the two public moduli.  Deliberately different in
 * every implementation choice from the HIP product (32-bit limbs, XYZZ buckets, generated
 * constant tables) so that agreement between the two means something.
 */
#define _POSIX_C_SOURCE 200809L
#include "oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------------------ */
/* generic n-limb Montgomery arithmetic                                                  */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    int n;
    u64 m[6];   /* modulus */
    u64 inv;    /* -m^-1 mod 2^64 */
    u64 r1[6];  /* R mod m   (Montgomery one) */
    u64 r2[6];  /* R^2 mod m */
} mont_ctx;

static int ge_n(const u64* a, const u64* b, int n) {
    for (int i = n - 1; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static u64 add_n(u64* o, const u64* a, const u64* b, int n) {
    u128 c = 0;
    for (int i = 0; i < n; i++) { c += (u128)a[i] + b[i]; o[i] = (u64)c; c >>= 64; }
    return (u64)c;
}
static u64 sub_n(u64* o, const u64* a, const u64* b, int n) {
    u64 br = 0;
    for (int i = 0; i < n; i++) {
        u128 d = (u128)a[i] - b[i] - br;
        o[i] = (u64)d;
        br = (u64)(d >> 64) & 1;
    }
    return br;
}
static void mod_add(u64* o, const u64* a, const u64* b, const mont_ctx* c) {
    u64 t[6];
    u64 carry = add_n(t, a, b, c->n);
    if (carry || ge_n(t, c->m, c->n)) sub_n(t, t, c->m, c->n);
    memcpy(o, t, 8 * c->n);
}
static void mod_sub(u64* o, const u64* a, const u64* b, const mont_ctx* c) {
    u64 t[6];
    if (sub_n(t, a, b, c->n)) add_n(t, t, c->m, c->n);
    memcpy(o, t, 8 * c->n);
}
static int is_zero_n(const u64* a, int n) {
    u64 x = 0;
    for (int i = 0; i < n; i++) x |= a[i];
    return x == 0;
}
static void mod_neg(u64* o, const u64* a, const mont_ctx* c) {
    if (is_zero_n(a, c->n)) { memset(o, 0, 8 * c->n); return; }
    u64 t[6];
    sub_n(t, c->m, a, c->n);
    memcpy(o, t, 8 * c->n);
}
static void mont_mul(u64* o, const u64* a, const u64* b, const mont_ctx* c) {
    const int n = c->n;
    u64 t[8] = {0};
    for (int i = 0; i < n; i++) {
        u128 cy = 0;
        for (int j = 0; j < n; j++) {
            cy += (u128)a[j] * b[i] + t[j];
            t[j] = (u64)cy; cy >>= 64;
        }
        cy += t[n]; t[n] = (u64)cy; t[n + 1] = (u64)(cy >> 64);
        u64 mm = t[0] * c->inv;
        cy = (u128)mm * c->m[0] + t[0]; cy >>= 64;
        for (int j = 1; j < n; j++) {
            cy += (u128)mm * c->m[j] + t[j];
            t[j - 1] = (u64)cy; cy >>= 64;
        }
        cy += t[n]; t[n - 1] = (u64)cy; t[n] = t[n + 1] + (u64)(cy >> 64);
    }
    if (t[n] || ge_n(t, c->m, n)) sub_n(t, t, c->m, n);
    memcpy(o, t, 8 * n);
}
static void mont_init(mont_ctx* c, const u64* m, int n) {
    c->n = n;
    memset(c->m, 0, sizeof c->m);
    memcpy(c->m, m, 8 * n);
    u64 x = 1; /* Newton: x = m^-1 mod 2^64 */
    for (int i = 0; i < 6; i++) x *= 2 - m[0] * x;
    c->inv = (u64)0 - x;
    /* r1 = 2^(64n) mod m by doubling 1 */
    u64 t[6] = {1, 0, 0, 0, 0, 0};
    for (int i = 0; i < 64 * n; i++) mod_add(t, t, t, c);
    memcpy(c->r1, t, sizeof t);
    for (int i = 0; i < 64 * n; i++) mod_add(t, t, t, c);
    memcpy(c->r2, t, sizeof t);
}
static void mont_pow(u64* o, const u64* a, const u64* e, int elimbs, const mont_ctx* c) {
    u64 acc[6], base[6];
    memcpy(acc, c->r1, sizeof acc);
    memcpy(base, a, 8 * c->n);
    for (int i = 0; i < 64 * elimbs; i++) {
        if ((e[i >> 6] >> (i & 63)) & 1) mont_mul(acc, acc, base, c);
        mont_mul(base, base, base, c);
    }
    memcpy(o, acc, 8 * c->n);
}
static void mont_inv(u64* o, const u64* a, const mont_ctx* c) { /* Fermat */
    u64 e[6], two[6] = {2, 0, 0, 0, 0, 0};
    sub_n(e, c->m, two, c->n);
    mont_pow(o, a, e, c->n, c);
}
static void be_to_limbs(u64* o, const uint8_t* b, int n) {
    for (int i = 0; i < n; i++) {
        u64 v = 0;
        for (int j = 0; j < 8; j++) v = (v << 8) | b[(n - 1 - i) * 8 + j];
        o[i] = v;
    }
}
static void limbs_to_be(uint8_t* b, const u64* a, int n) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 8; j++) b[(n - 1 - i) * 8 + j] = (uint8_t)(a[i] >> (56 - 8 * j));
}

/* ------------------------------------------------------------------------------------ */
/* the two fields                                                                        */
/* ------------------------------------------------------------------------------------ */
static const u64 P_LIMBS[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                               0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const u64 R_LIMBS[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL,
                               0x73eda753299d7d48ULL};
static mont_ctx FP, FR;
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

typedef struct { u64 l[6]; } fp_t;
typedef struct { u64 l[4]; } fr_t;
typedef struct { fp_t c0, c1; } fp2_t;

static fp_t FP_ONE, FP_ZERO, FP_B1;
static fp2_t FP2_ONE, FP2_ZERO, FP2_B2;
static void init_curves(void);

static void do_init(void) {
    mont_init(&FP, P_LIMBS, 6);
    mont_init(&FR, R_LIMBS, 4);
    memset(&FP_ZERO, 0, sizeof FP_ZERO);
    memcpy(FP_ONE.l, FP.r1, 48);
    fp_t four = {{4, 0, 0, 0, 0, 0}};
    mont_mul(FP_B1.l, four.l, FP.r2, &FP);
    FP2_ZERO.c0 = FP_ZERO; FP2_ZERO.c1 = FP_ZERO;
    FP2_ONE.c0 = FP_ONE; FP2_ONE.c1 = FP_ZERO;
    FP2_B2.c0 = FP_B1; FP2_B2.c1 = FP_B1;
    init_curves();
}
static void ensure_init(void) { pthread_once(&g_once, do_init); }

/* Fp */
static void fp_add(fp_t* o, const fp_t* a, const fp_t* b) { mod_add(o->l, a->l, b->l, &FP); }
static void fp_sub(fp_t* o, const fp_t* a, const fp_t* b) { mod_sub(o->l, a->l, b->l, &FP); }
static void fp_neg(fp_t* o, const fp_t* a) { mod_neg(o->l, a->l, &FP); }
static void fp_mul(fp_t* o, const fp_t* a, const fp_t* b) { mont_mul(o->l, a->l, b->l, &FP); }
static void fp_sqr(fp_t* o, const fp_t* a) { mont_mul(o->l, a->l, a->l, &FP); }
static void fp_inv(fp_t* o, const fp_t* a) { mont_inv(o->l, a->l, &FP); }
static int fp_is_zero(const fp_t* a) { return is_zero_n(a->l, 6); }
static int fp_eq(const fp_t* a, const fp_t* b) { return memcmp(a->l, b->l, 48) == 0; }
static int fp_from_be(fp_t* o, const uint8_t* b) {
    u64 t[6];
    be_to_limbs(t, b, 6);
    if (ge_n(t, FP.m, 6)) return OR_ERR_ENCODING;
    mont_mul(o->l, t, FP.r2, &FP);
    return OR_OK;
}
static void fp_to_limbs(u64 t[6], const fp_t* a) {
    u64 one[6] = {1, 0, 0, 0, 0, 0};
    mont_mul(t, a->l, one, &FP);
}
static void fp_to_be(uint8_t* b, const fp_t* a) {
    u64 t[6];
    fp_to_limbs(t, a);
    limbs_to_be(b, t, 6);
}
/* y > (p-1)/2 ? (ZCash sign bit) */
static int fp_lex_larger(const fp_t* a) {
    u64 t[6], d[6];
    fp_to_limbs(t, a);
    add_n(d, t, t, 6); /* 2y < 2^382 fits */
    return !ge_n(FP.m, d, 6); /* 2y > p  (2y == p impossible, p odd) */
}
static int fp_sqrt(fp_t* o, const fp_t* a) { /* p = 3 mod 4 */
    u64 e[6], one[6] = {1, 0, 0, 0, 0, 0};
    add_n(e, FP.m, one, 6);
    for (int i = 0; i < 6; i++) e[i] = (e[i] >> 2) | (i < 5 ? e[i + 1] << 62 : 0);
    fp_t s, chk;
    mont_pow(s.l, a->l, e, 6, &FP);
    fp_sqr(&chk, &s);
    if (!fp_eq(&chk, a)) return 0;
    *o = s;
    return 1;
}

/* Fp2 = Fp[u]/(u^2+1) */
static void fp2_add(fp2_t* o, const fp2_t* a, const fp2_t* b) { fp_add(&o->c0, &a->c0, &b->c0); fp_add(&o->c1, &a->c1, &b->c1); }
static void fp2_sub(fp2_t* o, const fp2_t* a, const fp2_t* b) { fp_sub(&o->c0, &a->c0, &b->c0); fp_sub(&o->c1, &a->c1, &b->c1); }
static void fp2_neg(fp2_t* o, const fp2_t* a) { fp_neg(&o->c0, &a->c0); fp_neg(&o->c1, &a->c1); }
static void fp2_mul(fp2_t* o, const fp2_t* a, const fp2_t* b) {
    fp_t t0, t1, t2, t3;
    fp_mul(&t0, &a->c0, &b->c0);
    fp_mul(&t1, &a->c1, &b->c1);
    fp_mul(&t2, &a->c0, &b->c1);
    fp_mul(&t3, &a->c1, &b->c0);
    fp_sub(&o->c0, &t0, &t1);
    fp_add(&o->c1, &t2, &t3);
}
static void fp2_sqr(fp2_t* o, const fp2_t* a) { fp2_mul(o, a, a); }
static void fp2_inv(fp2_t* o, const fp2_t* a) {
    fp_t t0, t1;
    fp_sqr(&t0, &a->c0); fp_sqr(&t1, &a->c1); fp_add(&t0, &t0, &t1);
    fp_inv(&t0, &t0);
    fp_mul(&o->c0, &a->c0, &t0);
    fp_mul(&t1, &a->c1, &t0);
    fp_neg(&o->c1, &t1);
}
static int fp2_is_zero(const fp2_t* a) { return fp_is_zero(&a->c0) && fp_is_zero(&a->c1); }
static int fp2_eq(const fp2_t* a, const fp2_t* b) { return fp_eq(&a->c0, &b->c0) && fp_eq(&a->c1, &b->c1); }
static int fp2_lex_larger(const fp2_t* a) {
    if (!fp_is_zero(&a->c1)) return fp_lex_larger(&a->c1);
    return fp_lex_larger(&a->c0);
}
static void fp2_pow(fp2_t* o, const fp2_t* a, const u64* e, int elimbs) {
    fp2_t acc = FP2_ONE, base = *a;
    for (int i = 0; i < 64 * elimbs; i++) {
        if ((e[i >> 6] >> (i & 63)) & 1) fp2_mul(&acc, &acc, &base);
        fp2_sqr(&base, &base);
    }
    *o = acc;
}
static int fp2_sqrt(fp2_t* o, const fp2_t* a) { /* Adj & Rodriguez-Henriquez alg. 9 */
    if (fp2_is_zero(a)) { *o = FP2_ZERO; return 1; }
    u64 e[6], three[6] = {3, 0, 0, 0, 0, 0}, one[6] = {1, 0, 0, 0, 0, 0};
    sub_n(e, FP.m, three, 6);
    for (int i = 0; i < 6; i++) e[i] = (e[i] >> 2) | (i < 5 ? e[i + 1] << 62 : 0);
    fp2_t a1, alpha, a0, x0, x, m1, chk;
    fp2_pow(&a1, a, e, 6);
    fp2_mul(&alpha, &a1, a); fp2_mul(&alpha, &alpha, &a1);
    fp2_pow(&a0, &alpha, FP.m, 6); fp2_mul(&a0, &a0, &alpha);
    fp2_neg(&m1, &FP2_ONE);
    if (fp2_eq(&a0, &m1)) return 0;
    fp2_mul(&x0, &a1, a);
    if (fp2_eq(&alpha, &m1)) {
        fp2_t u = FP2_ZERO; u.c1 = FP_ONE;
        fp2_mul(&x, &u, &x0);
    } else {
        fp2_t b; fp2_add(&b, &FP2_ONE, &alpha);
        sub_n(e, FP.m, one, 6);
        for (int i = 0; i < 6; i++) e[i] = (e[i] >> 1) | (i < 5 ? e[i + 1] << 63 : 0);
        fp2_pow(&b, &b, e, 6);
        fp2_mul(&x, &b, &x0);
    }
    fp2_sqr(&chk, &x);
    if (!fp2_eq(&chk, a)) return 0;
    *o = x;
    return 1;
}

/* Fr */
static void fr_from_be(fr_t* o, const uint8_t* b) { /* reduces non-canonical input mod r */
    u64 t[4];
    be_to_limbs(t, b, 4);
    while (ge_n(t, FR.m, 4)) sub_n(t, t, FR.m, 4);
    mont_mul(o->l, t, FR.r2, &FR);
}
static void fr_to_be(uint8_t* b, const fr_t* a) {
    u64 t[4], one[4] = {1, 0, 0, 0};
    mont_mul(t, a->l, one, &FR);
    limbs_to_be(b, t, 4);
}
static void fr_add(fr_t* o, const fr_t* a, const fr_t* b) { mod_add(o->l, a->l, b->l, &FR); }
static void fr_sub(fr_t* o, const fr_t* a, const fr_t* b) { mod_sub(o->l, a->l, b->l, &FR); }
static void fr_mul(fr_t* o, const fr_t* a, const fr_t* b) { mont_mul(o->l, a->l, b->l, &FR); }
static void fr_inv(fr_t* o, const fr_t* a) { mont_inv(o->l, a->l, &FR); }
static void fr_neg(fr_t* o, const fr_t* a) { mod_neg(o->l, a->l, &FR); }
static int fr_is_zero(const fr_t* a) { return is_zero_n(a->l, 4); }
static void fr_set_u64(fr_t* o, u64 v) {
    u64 t[4] = {v, 0, 0, 0};
    mont_mul(o->l, t, FR.r2, &FR);
}
static void fr_set_i64(fr_t* o, int64_t v) { /* SetInt64 (curve.go:17-19): Euclidean mod r */
    if (v >= 0) { fr_set_u64(o, (u64)v); return; }
    fr_t t;
    fr_set_u64(&t, (u64)0 - (u64)v);
    fr_neg(o, &t);
}
/* canonical plain limbs of a be32 scalar (reduced mod r) */
static void scalar_limbs(u64 k[4], const uint8_t* b) {
    be_to_limbs(k, b, 4);
    while (ge_n(k, FR.m, 4)) sub_n(k, k, FR.m, 4);
}
static void scalar_limbs_i64(u64 k[4], int64_t v) {
    fr_t t;
    uint8_t b[32];
    fr_set_i64(&t, v);
    fr_to_be(b, &t);
    be_to_limbs(k, b, 4);
}

/* ------------------------------------------------------------------------------------ */
/* curve instantiations                                                                  */
/* ------------------------------------------------------------------------------------ */
#define CN(n) g1_##n
#define F fp_t
#define FADD fp_add
#define FSUB fp_sub
#define FMUL fp_mul
#define FSQR fp_sqr
#define FNEG fp_neg
#define FINV fp_inv
#define FISZERO fp_is_zero
#define FEQ fp_eq
#define FONE FP_ONE
#define FZERO FP_ZERO
#define FB FP_B1
#include "curve_tmpl.h"
#undef CN
#undef F
#undef FADD
#undef FSUB
#undef FMUL
#undef FSQR
#undef FNEG
#undef FINV
#undef FISZERO
#undef FEQ
#undef FONE
#undef FZERO
#undef FB

#define CN(n) g2_##n
#define F fp2_t
#define FADD fp2_add
#define FSUB fp2_sub
#define FMUL fp2_mul
#define FSQR fp2_sqr
#define FNEG fp2_neg
#define FINV fp2_inv
#define FISZERO fp2_is_zero
#define FEQ fp2_eq
#define FONE FP2_ONE
#define FZERO FP2_ZERO
#define FB FP2_B2
#include "curve_tmpl.h"
#undef CN

static g1_aff G1_GEN;
static g2_aff G2_GEN;

static const char* G1X = "17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb";
static const char* G1Y = "08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1";
static const char* G2X0 = "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8";
static const char* G2X1 = "13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e";
static const char* G2Y0 = "0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801";
static const char* G2Y1 = "0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be";

static void fp_from_hex(fp_t* o, const char* h) {
    uint8_t b[48];
    for (int i = 0; i < 48; i++) {
        unsigned v = 0;
        for (int j = 0; j < 2; j++) {
            char ch = h[2 * i + j];
            v = v * 16 + (unsigned)(ch <= '9' ? ch - '0' : ch - 'a' + 10);
        }
        b[i] = (uint8_t)v;
    }
    fp_from_be(o, b);
}
static void init_curves(void) {
    fp_from_hex(&G1_GEN.x, G1X); fp_from_hex(&G1_GEN.y, G1Y); G1_GEN.inf = 0;
    fp_from_hex(&G2_GEN.x.c0, G2X0); fp_from_hex(&G2_GEN.x.c1, G2X1);
    fp_from_hex(&G2_GEN.y.c0, G2Y0); fp_from_hex(&G2_GEN.y.c1, G2Y1); G2_GEN.inf = 0;
}

/* ---- byte (de)serialisation of points ---- */
static int g1_from_bytes(g1_aff* o, const uint8_t* b) {
    if (b[0] & 0x40) { o->inf = 1; o->x = FP_ZERO; o->y = FP_ZERO; return OR_OK; }
    o->inf = 0;
    if (fp_from_be(&o->x, b) || fp_from_be(&o->y, b + 48)) return OR_ERR_ENCODING;
    return OR_OK;
}
static void g1_to_bytes(uint8_t* b, const g1_aff* a) {
    if (a->inf) { memset(b, 0, 96); b[0] = 0x40; return; }
    fp_to_be(b, &a->x); fp_to_be(b + 48, &a->y);
}
static int g2_from_bytes(g2_aff* o, const uint8_t* b) {
    if (b[0] & 0x40) { o->inf = 1; o->x = FP2_ZERO; o->y = FP2_ZERO; return OR_OK; }
    o->inf = 0;
    if (fp_from_be(&o->x.c1, b) || fp_from_be(&o->x.c0, b + 48) || fp_from_be(&o->y.c1, b + 96) ||
        fp_from_be(&o->y.c0, b + 144))
        return OR_ERR_ENCODING;
    return OR_OK;
}
static void g2_to_bytes(uint8_t* b, const g2_aff* a) {
    if (a->inf) { memset(b, 0, 192); b[0] = 0x40; return; }
    fp_to_be(b, &a->x.c1); fp_to_be(b + 48, &a->x.c0);
    fp_to_be(b + 96, &a->y.c1); fp_to_be(b + 144, &a->y.c0);
}

/* ------------------------------------------------------------------------------------ */
/* exported: Fr                                                                          */
/* ------------------------------------------------------------------------------------ */
void or_fr_from_i64(int64_t v, uint8_t out[32]) { ensure_init(); fr_t t; fr_set_i64(&t, v); fr_to_be(out, &t); }
#define FR_BINOP(name, op)                                                        \
    void name(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) {        \
        ensure_init();                                                            \
        fr_t x, y;                                                                \
        fr_from_be(&x, a); fr_from_be(&y, b);                                     \
        op(&x, &x, &y);                                                           \
        fr_to_be(out, &x);                                                        \
    }
FR_BINOP(or_fr_add, fr_add)
FR_BINOP(or_fr_sub, fr_sub)
FR_BINOP(or_fr_mul, fr_mul)
void or_fr_inv(const uint8_t a[32], uint8_t out[32]) { ensure_init(); fr_t x; fr_from_be(&x, a); fr_inv(&x, &x); fr_to_be(out, &x); }

/* ------------------------------------------------------------------------------------ */
/* exported: Poly (algebra.go:89-243)                                                    */
/* ------------------------------------------------------------------------------------ */
static fr_t* poly_load(const uint8_t* b, size_t n) {
    fr_t* p = (fr_t*)malloc(sizeof(fr_t) * (n ? n : 1));
    for (size_t i = 0; i < n; i++) fr_from_be(&p[i], b + 32 * i);
    return p;
}
static void poly_store(uint8_t* b, const fr_t* p, size_t n) {
    for (size_t i = 0; i < n; i++) fr_to_be(b + 32 * i, &p[i]);
}
/* Poly.Mul schoolbook (algebra.go:92-105) */
static void poly_mul_fr(fr_t* out, const fr_t* a, size_t na, const fr_t* b, size_t nb) {
    memset(out, 0, sizeof(fr_t) * (na + nb - 1));
    for (size_t i = 0; i < na; i++)
        for (size_t j = 0; j < nb; j++) {
            fr_t t;
            fr_mul(&t, &a[i], &b[j]);
            fr_add(&out[i + j], &out[i + j], &t);
        }
}
void or_poly_mul(const uint8_t* a, size_t na, const uint8_t* b, size_t nb, uint8_t* out) {
    ensure_init();
    fr_t *pa = poly_load(a, na), *pb = poly_load(b, nb);
    fr_t* po = (fr_t*)malloc(sizeof(fr_t) * (na + nb));
    poly_mul_fr(po, pa, na, pb, nb);
    poly_store(out, po, na + nb - 1);
    free(pa); free(pb); free(po);
}
/* Poly.Eval Horner (algebra.go:107-115) */
static void poly_eval_fr(fr_t* out, const fr_t* p, size_t n, const fr_t* x) {
    fr_t v;
    memset(&v, 0, sizeof v);
    for (size_t j = n; j-- > 0;) { fr_mul(&v, &v, x); fr_add(&v, &v, &p[j]); }
    *out = v;
}
void or_poly_eval(const uint8_t* p, size_t n, const uint8_t x[32], uint8_t out[32]) {
    ensure_init();
    fr_t* pp = poly_load(p, n);
    fr_t xx, v;
    fr_from_be(&xx, x);
    poly_eval_fr(&v, pp, n, &xx);
    fr_to_be(out, &v);
    free(pp);
}
/* (*Poly).Div2 long division (algebra.go:140-159).  Each iteration takes the top
 * coefficient of r, forms tPoly = t*x^(len(r)-len(p2)), q += tPoly, r = (r - tPoly*p2)[:len-1].
 * The reference recomputes tPoly.Mul(p2) in full (O(n^2) per iteration => O(n^3)); the
 * numbers produced are those of this loop. */
static int poly_div2_fr(fr_t* q, fr_t* r, const fr_t* p, size_t np, const fr_t* d, size_t nd) {
    if (nd == 0) return OR_ERR_LENGTH;
    size_t rl = np;
    fr_t* rr = (fr_t*)malloc(sizeof(fr_t) * (np ? np : 1));
    memcpy(rr, p, sizeof(fr_t) * np);
    fr_t dinv;
    fr_inv(&dinv, &d[nd - 1]);
    while (rl > 0 && rl >= nd) {
        fr_t t;
        fr_mul(&t, &rr[rl - 1], &dinv);
        size_t deg = rl - nd;
        q[deg] = t; /* q.Add(tPoly): distinct degree every iteration */
        for (size_t j = 0; j < nd; j++) {
            fr_t m;
            fr_mul(&m, &t, &d[j]);
            fr_sub(&rr[deg + j], &rr[deg + j], &m);
        }
        rl--;
    }
    for (size_t i = 0; i + 1 < nd; i++) {
        if (i < rl) r[i] = rr[i]; else memset(&r[i], 0, sizeof(fr_t));
    }
    free(rr);
    return OR_OK;
}
int or_poly_div2(const uint8_t* p, size_t np, const uint8_t* d, size_t nd, uint8_t* q, uint8_t* r) {
    ensure_init();
    if (np < nd) return OR_ERR_LENGTH;
    fr_t *pp = poly_load(p, np), *pd = poly_load(d, nd);
    size_t nq = np - nd + 1;
    fr_t* pq = (fr_t*)calloc(nq, sizeof(fr_t));
    fr_t* pr = (fr_t*)calloc(nd, sizeof(fr_t));
    int rc = poly_div2_fr(pq, pr, pp, np, pd, nd);
    poly_store(q, pq, nq);
    poly_store(r, pr, nd - 1);
    free(pp); free(pd); free(pq); free(pr);
    return rc;
}
/* Interpolate + lagrangeBasis (algebra.go:254-338) on x = 1..n.  Literal: every basis
 * polynomial is built by repeated Mul with (x - m) and scaled by prod 1/(j - m). */
static void interpolate_fr(fr_t* out, const fr_t* ys, size_t n) {
    memset(out, 0, sizeof(fr_t) * n);
    fr_t* basis = (fr_t*)malloc(sizeof(fr_t) * (n + 1));
    fr_t* tmp = (fr_t*)malloc(sizeof(fr_t) * (n + 1));
    for (size_t j = 1; j <= n; j++) {
        size_t len = 1;
        basis[0] = *(fr_t*)FR.r1;
        fr_t acc = *(fr_t*)FR.r1;
        for (size_t m = 1; m <= n; m++) {
            if (m == j) continue;
            fr_t lin[2], den;
            fr_set_i64(&lin[0], -(int64_t)m);
            lin[1] = *(fr_t*)FR.r1;
            poly_mul_fr(tmp, basis, len, lin, 2);
            len++;
            memcpy(basis, tmp, sizeof(fr_t) * len);
            fr_set_i64(&den, (int64_t)j - (int64_t)m);
            fr_inv(&den, &den);
            fr_mul(&acc, &acc, &den);
        }
        fr_t sc;
        fr_mul(&sc, &acc, &ys[j - 1]);
        for (size_t i = 0; i < len; i++) {
            fr_t t;
            fr_mul(&t, &basis[i], &sc);
            fr_add(&out[i], &out[i], &t);
        }
    }
    free(basis); free(tmp);
}
void or_interpolate(const uint8_t* ys, size_t n, uint8_t* out) {
    ensure_init();
    fr_t* y = poly_load(ys, n);
    fr_t* o = (fr_t*)malloc(sizeof(fr_t) * (n ? n : 1));
    interpolate_fr(o, y, n);
    poly_store(out, o, n);
    free(y); free(o);
}

/* ------------------------------------------------------------------------------------ */
/* exported: points                                                                      */
/* ------------------------------------------------------------------------------------ */
int or_g1_mul(const uint8_t s[32], const uint8_t* p, uint8_t out[96]) {
    ensure_init();
    g1_aff a = G1_GEN, r;
    if (p && g1_from_bytes(&a, p)) return OR_ERR_ENCODING;
    u64 k[4];
    scalar_limbs(k, s);
    g1_jac j;
    g1_mul(&j, k, &a);
    g1_to_aff(&r, &j);
    g1_to_bytes(out, &r);
    return OR_OK;
}
int or_g2_mul(const uint8_t s[32], const uint8_t* p, uint8_t out[192]) {
    ensure_init();
    g2_aff a = G2_GEN, r;
    if (p && g2_from_bytes(&a, p)) return OR_ERR_ENCODING;
    u64 k[4];
    scalar_limbs(k, s);
    g2_jac j;
    g2_mul(&j, k, &a);
    g2_to_aff(&r, &j);
    g2_to_bytes(out, &r);
    return OR_OK;
}
int or_g1_add(const uint8_t a[96], const uint8_t b[96], uint8_t out[96]) {
    ensure_init();
    g1_aff x, y, r;
    if (g1_from_bytes(&x, a) || g1_from_bytes(&y, b)) return OR_ERR_ENCODING;
    g1_jac j;
    g1_from_aff(&j, &x);
    g1_madd(&j, &j, &y);
    g1_to_aff(&r, &j);
    g1_to_bytes(out, &r);
    return OR_OK;
}
int or_g2_add(const uint8_t a[192], const uint8_t b[192], uint8_t out[192]) {
    ensure_init();
    g2_aff x, y, r;
    if (g2_from_bytes(&x, a) || g2_from_bytes(&y, b)) return OR_ERR_ENCODING;
    g2_jac j;
    g2_from_aff(&j, &x);
    g2_madd(&j, &j, &y);
    g2_to_aff(&r, &j);
    g2_to_bytes(out, &r);
    return OR_OK;
}
int or_g1_neg(const uint8_t a[96], uint8_t out[96]) {
    ensure_init();
    g1_aff x, r;
    if (g1_from_bytes(&x, a)) return OR_ERR_ENCODING;
    g1_neg_aff(&r, &x);
    g1_to_bytes(out, &r);
    return OR_OK;
}
int or_g1_on_curve(const uint8_t a[96]) {
    ensure_init();
    g1_aff x;
    if (g1_from_bytes(&x, a)) return 0;
    return g1_on_curve(&x);
}
int or_g2_on_curve(const uint8_t a[192]) {
    ensure_init();
    g2_aff x;
    if (g2_from_bytes(&x, a)) return 0;
    return g2_on_curve(&x);
}
int or_g1_compress(const uint8_t a[96], uint8_t out[48]) {
    ensure_init();
    g1_aff x;
    if (g1_from_bytes(&x, a)) return OR_ERR_ENCODING;
    if (x.inf) { memset(out, 0, 48); out[0] = 0xC0; return OR_OK; }
    fp_to_be(out, &x.x);
    out[0] |= 0x80 | (fp_lex_larger(&x.y) ? 0x20 : 0);
    return OR_OK;
}
int or_g1_decompress(const uint8_t a[48], uint8_t out[96]) {
    ensure_init();
    g1_aff r;
    if (a[0] & 0x40) { r.inf = 1; g1_to_bytes(out, &r); return OR_OK; }
    uint8_t xb[48];
    memcpy(xb, a, 48);
    xb[0] &= 0x1F;
    r.inf = 0;
    if (fp_from_be(&r.x, xb)) return OR_ERR_ENCODING;
    fp_t t;
    fp_sqr(&t, &r.x); fp_mul(&t, &t, &r.x); fp_add(&t, &t, &FP_B1);
    if (!fp_sqrt(&r.y, &t)) return OR_ERR_ENCODING;
    if (fp_lex_larger(&r.y) != !!(a[0] & 0x20)) fp_neg(&r.y, &r.y);
    g1_to_bytes(out, &r);
    return OR_OK;
}
int or_g2_compress(const uint8_t a[192], uint8_t out[96]) {
    ensure_init();
    g2_aff x;
    if (g2_from_bytes(&x, a)) return OR_ERR_ENCODING;
    if (x.inf) { memset(out, 0, 96); out[0] = 0xC0; return OR_OK; }
    fp_to_be(out, &x.x.c1); fp_to_be(out + 48, &x.x.c0);
    out[0] |= 0x80 | (fp2_lex_larger(&x.y) ? 0x20 : 0);
    return OR_OK;
}
int or_g2_decompress(const uint8_t a[96], uint8_t out[192]) {
    ensure_init();
    g2_aff r;
    if (a[0] & 0x40) { r.inf = 1; g2_to_bytes(out, &r); return OR_OK; }
    uint8_t xb[48];
    memcpy(xb, a, 48);
    xb[0] &= 0x1F;
    r.inf = 0;
    if (fp_from_be(&r.x.c1, xb) || fp_from_be(&r.x.c0, a + 48)) return OR_ERR_ENCODING;
    fp2_t t;
    fp2_sqr(&t, &r.x); fp2_mul(&t, &t, &r.x); fp2_add(&t, &t, &FP2_B2);
    if (!fp2_sqrt(&r.y, &t)) return OR_ERR_ENCODING;
    if (fp2_lex_larger(&r.y) != !!(a[0] & 0x20)) fp2_neg(&r.y, &r.y);
    g2_to_bytes(out, &r);
    return OR_OK;
}

/* Poly.BlindEval (algebra.go:348-359): acc = acc.Add(acc, tmp.Mul(p[i], blindedPoint[i])) */
#define DEFINE_BLIND_EVAL(G, PB)                                                                     \
    static int G##_blind_eval_k(const u64* k, const uint8_t* points, size_t n, uint8_t* out) {       \
        G##_jac acc;                                                                                 \
        G##_jac_set_inf(&acc);                                                                       \
        for (size_t i = 0; i < n; i++) {                                                             \
            G##_aff p;                                                                               \
            if (G##_from_bytes(&p, points + (size_t)PB * i)) return OR_ERR_ENCODING;                 \
            G##_jac t;                                                                               \
            G##_mul(&t, k + 4 * i, &p);                                                              \
            G##_add(&acc, &acc, &t);                                                                 \
        }                                                                                            \
        G##_aff r;                                                                                   \
        G##_to_aff(&r, &acc);                                                                        \
        G##_to_bytes(out, &r);                                                                       \
        return OR_OK;                                                                                \
    }                                                                                                \
    int or_##G##_blind_eval(const uint8_t* scalars, size_t ns, const uint8_t* points, size_t np,     \
                            uint8_t* out) {                                                          \
        ensure_init();                                                                               \
        if (ns != np) return OR_ERR_LENGTH; /* algebra.go:350-352 */                                 \
        u64* k = (u64*)malloc(32 * (ns ? ns : 1));                                                   \
        for (size_t i = 0; i < ns; i++) scalar_limbs(k + 4 * i, scalars + 32 * i);                   \
        int rc = G##_blind_eval_k(k, points, ns, out);                                               \
        free(k);                                                                                     \
        return rc;                                                                                   \
    }                                                                                                \
    int or_##G##_blind_eval_i64(const int64_t* scalars, const uint8_t* points, size_t n,             \
                                uint8_t* out) {                                                      \
        ensure_init();                                                                               \
        u64* k = (u64*)malloc(32 * (n ? n : 1));                                                     \
        for (size_t i = 0; i < n; i++) scalar_limbs_i64(k + 4 * i, scalars[i]);                      \
        int rc = G##_blind_eval_k(k, points, n, out);                                                \
        free(k);                                                                                     \
        return rc;                                                                                   \
    }                                                                                                \
    int or_##G##_msm_pippenger(const uint8_t* scalars, const uint8_t* points, size_t n, int threads, \
                               uint8_t* out) {                                                       \
        ensure_init();                                                                               \
        u64* k = (u64*)malloc(32 * (n ? n : 1));                                                     \
        G##_aff* pts = (G##_aff*)malloc(sizeof(G##_aff) * (n ? n : 1));                              \
        for (size_t i = 0; i < n; i++) {                                                             \
            scalar_limbs(k + 4 * i, scalars + 32 * i);                                               \
            if (G##_from_bytes(&pts[i], points + (size_t)PB * i)) { free(k); free(pts); return OR_ERR_ENCODING; } \
        }                                                                                            \
        G##_jac acc;                                                                                 \
        G##_pippenger(&acc, k, pts, n, threads);                                                     \
        G##_aff r;                                                                                   \
        G##_to_aff(&r, &acc);                                                                        \
        G##_to_bytes(out, &r);                                                                       \
        free(k); free(pts);                                                                          \
        return OR_OK;                                                                                \
    }                                                                                                \
    /* GeneratePowersCommit (algebra.go:371-384) */                                                  \
    int or_##G##_powers_commit(const uint8_t e[32], const uint8_t shift[32], size_t power,           \
                               uint8_t* out) {                                                       \
        ensure_init();                                                                               \
        fr_t ee, sh, si = *(fr_t*)FR.r1, tmp;                                                        \
        fr_from_be(&ee, e); fr_from_be(&sh, shift);                                                  \
        G##_jac* js = (G##_jac*)malloc(sizeof(G##_jac) * (power + 1));                               \
        for (size_t i = 0; i <= power; i++) {                                                        \
            if (i) fr_mul(&si, &si, &ee);                                                            \
            fr_mul(&tmp, &si, &sh);                                                                  \
            uint8_t b[32]; u64 k[4];                                                                 \
            fr_to_be(b, &tmp); be_to_limbs(k, b, 4);                                                 \
            G##_mul(&js[i], k, &G##_GEN_REF);                                                        \
        }                                                                                            \
        G##_aff* as = (G##_aff*)malloc(sizeof(G##_aff) * (power + 1));                               \
        G##_batch_to_aff(as, js, power + 1);                                                         \
        for (size_t i = 0; i <= power; i++) G##_to_bytes(out + (size_t)PB * i, &as[i]);              \
        free(js); free(as);                                                                          \
        return OR_OK;                                                                                \
    }                                                                                                \
    int or_##G##_gen_points(const uint8_t k0[32], const uint8_t q[32], size_t n, uint8_t* out) {     \
        ensure_init();                                                                               \
        u64 kk[4], qq[4];                                                                            \
        scalar_limbs(kk, k0); scalar_limbs(qq, q);                                                   \
        G##_jac cur, step;                                                                           \
        G##_mul(&cur, kk, &G##_GEN_REF);                                                             \
        G##_mul(&step, qq, &G##_GEN_REF);                                                            \
        G##_aff stepa;                                                                               \
        G##_to_aff(&stepa, &step);                                                                   \
        G##_jac* js = (G##_jac*)malloc(sizeof(G##_jac) * (n ? n : 1));                               \
        for (size_t i = 0; i < n; i++) { js[i] = cur; G##_madd(&cur, &cur, &stepa); }                \
        G##_aff* as = (G##_aff*)malloc(sizeof(G##_aff) * (n ? n : 1));                               \
        G##_batch_to_aff(as, js, n);                                                                 \
        for (size_t i = 0; i < n; i++) G##_to_bytes(out + (size_t)PB * i, &as[i]);                   \
        free(js); free(as);                                                                          \
        return OR_OK;                                                                                \
    }

#define g1_GEN_REF G1_GEN
#define g2_GEN_REF G2_GEN
DEFINE_BLIND_EVAL(g1, 96)
DEFINE_BLIND_EVAL(g2, 192)

/* ------------------------------------------------------------------------------------ */
/* exported: QAP (qap.go)                                                                */
/* ------------------------------------------------------------------------------------ */
/* ToQAP + qapInterpolate (qap.go:35-93): transpose, lift with SetInt64, Interpolate. */
int or_to_qap_dense(const int64_t* L, const int64_t* Rm, const int64_t* O, size_t n, size_t m,
                    uint8_t* left, uint8_t* right, uint8_t* outp, uint8_t* z) {
    ensure_init();
    const int64_t* mats[3] = {L, Rm, O};
    uint8_t* outs[3] = {left, right, outp};
    fr_t* ys = (fr_t*)malloc(sizeof(fr_t) * n);
    fr_t* poly = (fr_t*)malloc(sizeof(fr_t) * n);
    for (int k = 0; k < 3; k++)
        for (size_t v = 0; v < m; v++) {
            for (size_t g = 0; g < n; g++) fr_set_i64(&ys[g], mats[k][g * m + v]);
            interpolate_fr(poly, ys, n);
            poly_store(outs[k] + 32 * n * v, poly, n);
        }
    /* z = prod (x - i), qap.go:41-55 */
    fr_t* zz = (fr_t*)calloc(n + 2, sizeof(fr_t));
    fr_t* tmp = (fr_t*)calloc(n + 2, sizeof(fr_t));
    size_t zl = 0;
    for (size_t i = 1; i <= n; i++) {
        fr_t lin[2];
        fr_set_i64(&lin[0], -(int64_t)i);
        lin[1] = *(fr_t*)FR.r1;
        if (zl == 0) { zz[0] = lin[0]; zz[1] = lin[1]; zl = 2; }
        else { poly_mul_fr(tmp, zz, zl, lin, 2); zl++; memcpy(zz, tmp, sizeof(fr_t) * zl); }
    }
    poly_store(z, zz, n + 1);
    free(ys); free(poly); free(zz); free(tmp);
    return OR_OK;
}
/* computeAggregatePoly (qap.go:164-175): sum_i polys[i] * sol[i] */
void or_aggregate_poly(const uint8_t* polys, size_t n, size_t m, const uint8_t* sol, uint8_t* out) {
    ensure_init();
    fr_t* acc = (fr_t*)calloc(n, sizeof(fr_t));
    for (size_t v = 0; v < m; v++) {
        fr_t s;
        fr_from_be(&s, sol + 32 * v);
        for (size_t j = 0; j < n; j++) {
            fr_t c, t;
            fr_from_be(&c, polys + 32 * (n * v + j));
            fr_mul(&t, &c, &s);
            fr_add(&acc[j], &acc[j], &t);
        }
    }
    poly_store(out, acc, n);
    free(acc);
}
/* Quotient (qap.go:151-162): px = left.Mul(right).Sub(out); hx, rem = px.Div2(z) */
static int quotient_fr(fr_t* h, const fr_t* A, const fr_t* B, const fr_t* C, const fr_t* z, size_t n) {
    size_t np = 2 * n - 1;
    fr_t* px = (fr_t*)malloc(sizeof(fr_t) * np);
    poly_mul_fr(px, A, n, B, n);
    for (size_t i = 0; i < n; i++) fr_sub(&px[i], &px[i], &C[i]);
    fr_t* q = (fr_t*)calloc(np, sizeof(fr_t));
    fr_t* r = (fr_t*)calloc(n + 1, sizeof(fr_t));
    int rc = OR_OK;
    if (np >= n + 1) {
        rc = poly_div2_fr(q, r, px, np, z, n + 1);
        for (size_t i = 0; i < n && rc == OR_OK; i++)
            if (!fr_is_zero(&r[i])) rc = OR_ERR_NOT_DIVISIBLE; /* "apocalypse" */
        memcpy(h, q, sizeof(fr_t) * (n - 1));
    } else { /* n == 1: px has 1 coeff < len(z) = 2: loop never runs, q empty, rem = px */
        if (!fr_is_zero(&px[0])) rc = OR_ERR_NOT_DIVISIBLE;
    }
    free(px); free(q); free(r);
    return rc;
}
int or_quotient_from_aggregates(const uint8_t* A, const uint8_t* B, const uint8_t* C, const uint8_t* z,
                                size_t n, uint8_t* h) {
    ensure_init();
    fr_t *a = poly_load(A, n), *b = poly_load(B, n), *c = poly_load(C, n), *zz = poly_load(z, n + 1);
    fr_t* hh = (fr_t*)calloc(n, sizeof(fr_t));
    int rc = quotient_fr(hh, a, b, c, zz, n);
    poly_store(h, hh, n - 1);
    free(a); free(b); free(c); free(zz); free(hh);
    return rc;
}
int or_quotient_from_values(const uint8_t* yA, const uint8_t* yB, const uint8_t* yC, size_t n,
                            uint8_t* A, uint8_t* B, uint8_t* C, uint8_t* h) {
    ensure_init();
    const uint8_t* ins[3] = {yA, yB, yC};
    uint8_t* outs[3] = {A, B, C};
    fr_t* polys[3];
    /* Lagrange interpolation on {1..n} with the weights w_j = 1/prod_{m!=j}(j-m) and the
     * master polynomial z(x)/(x-j) by synthetic division: same polynomial as
     * Interpolate (algebra.go:254-281), O(n^2) instead of O(n^3). */
    fr_t* zz = (fr_t*)calloc(n + 2, sizeof(fr_t));
    fr_t* tmp = (fr_t*)calloc(n + 2, sizeof(fr_t));
    zz[0] = *(fr_t*)FR.r1;
    size_t zl = 1;
    for (size_t i = 1; i <= n; i++) {
        fr_t lin[2];
        fr_set_i64(&lin[0], -(int64_t)i);
        lin[1] = *(fr_t*)FR.r1;
        poly_mul_fr(tmp, zz, zl, lin, 2);
        zl++;
        memcpy(zz, tmp, sizeof(fr_t) * zl);
    }
    fr_t* w = (fr_t*)malloc(sizeof(fr_t) * n);
    for (size_t j = 1; j <= n; j++) {
        fr_t acc = *(fr_t*)FR.r1;
        for (size_t m = 1; m <= n; m++) {
            if (m == j) continue;
            fr_t d;
            fr_set_i64(&d, (int64_t)j - (int64_t)m);
            fr_mul(&acc, &acc, &d);
        }
        fr_inv(&w[j - 1], &acc);
    }
    fr_t* quo = (fr_t*)malloc(sizeof(fr_t) * (n + 1));
    for (int k = 0; k < 3; k++) {
        fr_t* y = poly_load(ins[k], n);
        polys[k] = (fr_t*)calloc(n, sizeof(fr_t));
        for (size_t j = 1; j <= n; j++) {
            /* quo = z / (x - j): quo[n-1] = 1, quo[i-1] = z[i] + j*quo[i] */
            fr_t jj;
            fr_set_u64(&jj, j);
            quo[n - 1] = zz[n];
            for (size_t i = n - 1; i >= 1; i--) {
                fr_t t;
                fr_mul(&t, &jj, &quo[i]);
                fr_add(&quo[i - 1], &zz[i], &t);
            }
            fr_t sc;
            fr_mul(&sc, &w[j - 1], &y[j - 1]);
            for (size_t i = 0; i < n; i++) {
                fr_t t;
                fr_mul(&t, &quo[i], &sc);
                fr_add(&polys[k][i], &polys[k][i], &t);
            }
        }
        poly_store(outs[k], polys[k], n);
        free(y);
    }
    fr_t* hh = (fr_t*)calloc(n, sizeof(fr_t));
    int rc = quotient_fr(hh, polys[0], polys[1], polys[2], zz, n);
    poly_store(h, hh, n - 1);
    for (int k = 0; k < 3; k++) free(polys[k]);
    free(zz); free(tmp); free(w); free(quo); free(hh);
    return rc;
}

double or_now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}


/* ------------------------------------------------------------------------------------ */
/* B1-h: a FAST CPU statement of the same polynomials -- the aggregate     */
/* polynomials A, B, C of computeAggregatePoly (qap.go:164-175) as interpolants of L.s, R.s, O.s    */
/* on {1..n}, and h = (A*B - C)/z of QAP.Quotient (qap.go:151-162) -- for sizes where the literal   */
/* O(n^3) restatement above cannot run (n = 2^16 .. 2^20).  Quasi-linear: NTT products over Fr,     */
/* values -> Newton coefficients by one convolution with (-1)^j / j!, Newton -> monomial by a       */
/* recursive product tree, exact division by z through a power-series inverse of rev(z).           */
/* Scalar code, 64-bit Montgomery limbs, recursion instead of the GPU's level-wise passes; pinned   */
/* against or_quotient_from_values (the literal algorithm) in tests/test_oracle_golden.py.          */
/* ------------------------------------------------------------------------------------ */
static fr_t FQ_ROOT[33]; /* FQ_ROOT[k]: primitive 2^k-th root of unity */
static pthread_once_t fq_once = PTHREAD_ONCE_INIT;
static void fq_init(void) {
    /* 7 generates Fr^*; w = 7^((r-1)/2^32) has order 2^32 (checked: w^(2^31) = -1) */
    fr_t g, w, minus1;
    fr_set_u64(&g, 7);
    u64 e[4];
    memcpy(e, R_LIMBS, sizeof e);
    e[0] -= 1; /* r - 1, low limb is ...00000001 */
    /* shift right by 32 */
    for (int i = 0; i < 4; i++) e[i] = (e[i] >> 32) | (i + 1 < 4 ? e[i + 1] << 32 : 0);
    mont_pow(w.l, g.l, e, 4, &FR);
    FQ_ROOT[32] = w;
    for (int k = 31; k >= 0; k--) fr_mul(&FQ_ROOT[k], &FQ_ROOT[k + 1], &FQ_ROOT[k + 1]);
    fr_set_i64(&minus1, -1);
    if (memcmp(&FQ_ROOT[1], &minus1, sizeof(fr_t)) != 0 || memcmp(&FQ_ROOT[0], FR.r1, sizeof(fr_t)) != 0) {
        fprintf(stderr, "oracle: 7 is not a generator of Fr^* ?\n");
        abort();
    }
}
static int fq_log2_ceil(size_t v) { int l = 0; while (((size_t)1 << l) < v) l++; return l; }

/* in-place radix-2 NTT of 2^lg points (natural order in and out); inverse includes the 1/n scaling */
static void fq_ntt(fr_t* a, int lg, int inverse) {
    const size_t n = (size_t)1 << lg;
    for (size_t i = 1, j = 0; i < n; i++) { /* bit reversal */
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { fr_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    for (int s = 1; s <= lg; s++) {
        const size_t half = (size_t)1 << (s - 1);
        fr_t wlen = FQ_ROOT[s];
        if (inverse) fr_inv(&wlen, &wlen);
        fr_t* tw = (fr_t*)malloc(sizeof(fr_t) * half);
        tw[0] = *(fr_t*)FR.r1;
        for (size_t k = 1; k < half; k++) fr_mul(&tw[k], &tw[k - 1], &wlen);
        for (size_t i = 0; i < n; i += 2 * half)
            for (size_t k = 0; k < half; k++) {
                fr_t u = a[i + k], v;
                fr_mul(&v, &a[i + k + half], &tw[k]);
                fr_add(&a[i + k], &u, &v);
                fr_sub(&a[i + k + half], &u, &v);
            }
        free(tw);
    }
    if (inverse) {
        fr_t ninv;
        fr_set_u64(&ninv, (u64)n);
        fr_inv(&ninv, &ninv);
        for (size_t i = 0; i < n; i++) fr_mul(&a[i], &a[i], &ninv);
    }
}
/* out[0..nout) = (a*b)[0..nout); out may alias neither input */
static void fq_mul(fr_t* out, size_t nout, const fr_t* a, size_t na, const fr_t* b, size_t nb) {
    if (na == 0 || nb == 0) { memset(out, 0, sizeof(fr_t) * nout); return; }
    const size_t full = na + nb - 1;
    if (na < 32 || nb < 32) {
        fr_t* t = (fr_t*)malloc(sizeof(fr_t) * full);
        poly_mul_fr(t, a, na, b, nb);
        for (size_t i = 0; i < nout; i++) { if (i < full) out[i] = t[i]; else memset(&out[i], 0, sizeof(fr_t)); }
        free(t);
        return;
    }
    const int lg = fq_log2_ceil(full);
    const size_t S = (size_t)1 << lg;
    fr_t* fa = (fr_t*)calloc(S, sizeof(fr_t));
    fr_t* fb = (fr_t*)calloc(S, sizeof(fr_t));
    memcpy(fa, a, sizeof(fr_t) * na);
    memcpy(fb, b, sizeof(fr_t) * nb);
    fq_ntt(fa, lg, 0);
    fq_ntt(fb, lg, 0);
    for (size_t i = 0; i < S; i++) fr_mul(&fa[i], &fa[i], &fb[i]);
    fq_ntt(fa, lg, 1);
    for (size_t i = 0; i < nout; i++) { if (i < full) out[i] = fa[i]; else memset(&out[i], 0, sizeof(fr_t)); }
    free(fa); free(fb);
}
/* Newton coefficients d[lo..hi) on the nodes 1, 2, ..: returns N(x) = sum_{k in [lo,hi)} d_k prod_{i=lo+1..k}(x - i)
 * (hi-lo coefficients) in `nout`, and Z(x) = prod_{i=lo+1..hi}(x - i) (hi-lo+1 coefficients, monic) in `zout` */
static void fq_newton_to_monomial(const fr_t* d, size_t lo, size_t hi, fr_t* nout, fr_t* zout) {
    const size_t len = hi - lo;
    if (len == 1) {
        nout[0] = d[lo];
        fr_set_i64(&zout[0], -(int64_t)(lo + 1));
        zout[1] = *(fr_t*)FR.r1;
        return;
    }
    const size_t mid = lo + len / 2, ll = mid - lo, lr = hi - mid;
    fr_t* nl = (fr_t*)malloc(sizeof(fr_t) * ll);
    fr_t* zl = (fr_t*)malloc(sizeof(fr_t) * (ll + 1));
    fr_t* nr = (fr_t*)malloc(sizeof(fr_t) * lr);
    fr_t* zr = (fr_t*)malloc(sizeof(fr_t) * (lr + 1));
    fq_newton_to_monomial(d, lo, mid, nl, zl);
    fq_newton_to_monomial(d, mid, hi, nr, zr);
    /* N = N_left + Z_left * N_right  (degree < len) ; Z = Z_left * Z_right */
    fr_t* t = (fr_t*)malloc(sizeof(fr_t) * (len + 1));
    fq_mul(t, len, zl, ll + 1, nr, lr);
    for (size_t i = 0; i < len; i++) { nout[i] = t[i]; if (i < ll) fr_add(&nout[i], &nout[i], &nl[i]); }
    fq_mul(zout, len + 1, zl, ll + 1, zr, lr + 1);
    free(nl); free(zl); free(nr); free(zr); free(t);
}
/* monomial coefficients (n of them) of the interpolant of ys on the nodes 1..n; z (n+1 coefficients) as a by-product */
static void fq_interpolate(const fr_t* ys, size_t n, const fr_t* invfact, fr_t* out, fr_t* z) {
    /* divided differences on equally spaced nodes: d_k = sum_j (y_j / j!) * (-1)^(k-j) / (k-j)!,  j = 0..k (y_j = f(j+1)) */
    fr_t* u = (fr_t*)malloc(sizeof(fr_t) * n);
    fr_t* v = (fr_t*)malloc(sizeof(fr_t) * n);
    for (size_t j = 0; j < n; j++) {
        fr_mul(&u[j], &ys[j], &invfact[j]);
        v[j] = invfact[j];
        if (j & 1) fr_neg(&v[j], &v[j]);
    }
    fr_t* d = (fr_t*)malloc(sizeof(fr_t) * n);
    fq_mul(d, n, u, n, v, n);
    fr_t* zz = (fr_t*)malloc(sizeof(fr_t) * (n + 1));
    fq_newton_to_monomial(d, 0, n, out, zz);
    if (z) memcpy(z, zz, sizeof(fr_t) * (n + 1));
    free(u); free(v); free(d); free(zz);
}
typedef struct { const fr_t* ys; size_t n; const fr_t* invfact; fr_t* out; fr_t* z; } fq_job;
static void* fq_job_run(void* p) {
    fq_job* j = (fq_job*)p;
    fq_interpolate(j->ys, j->n, j->invfact, j->out, j->z);
    return NULL;
}
/* g = f^-1 mod x^m for f[0] != 0 (Newton iteration g <- g (2 - f g)) */
static void fq_series_inverse(const fr_t* f, size_t nf, size_t m, fr_t* g) {
    memset(g, 0, sizeof(fr_t) * m);
    fr_inv(&g[0], &f[0]);
    fr_t two;
    fr_set_u64(&two, 2);
    fr_t* t = (fr_t*)malloc(sizeof(fr_t) * (m + 1));
    fr_t* g2 = (fr_t*)malloc(sizeof(fr_t) * (m + 1));
    for (size_t cur = 1; cur < m;) {
        const size_t nxt = 2 * cur < m ? 2 * cur : m;
        fq_mul(t, nxt, f, nf < nxt ? nf : nxt, g, cur);
        for (size_t i = 0; i < nxt; i++) fr_neg(&t[i], &t[i]);
        fr_add(&t[0], &t[0], &two);
        fq_mul(g2, nxt, g, cur, t, nxt);
        memcpy(g, g2, sizeof(fr_t) * nxt);
        cur = nxt;
    }
    free(t); free(g2);
}
int or_fast_quotient(const uint8_t* yA, const uint8_t* yB, const uint8_t* yC, size_t n, uint8_t* A, uint8_t* B, uint8_t* C, uint8_t* h) {
    ensure_init();
    pthread_once(&fq_once, fq_init);
    if (n < 2) return OR_ERR_LENGTH;
    fr_t *ya = poly_load(yA, n), *yb = poly_load(yB, n), *yc = poly_load(yC, n);
    int rc = OR_OK;
    for (size_t j = 0; j < n && rc == OR_OK; j++) { /* z | A*B - C  <=>  the gate equation holds at every root of z */
        fr_t t;
        fr_mul(&t, &ya[j], &yb[j]);
        if (memcmp(&t, &yc[j], sizeof t) != 0) rc = OR_ERR_NOT_DIVISIBLE;
    }
    fr_t* invfact = (fr_t*)malloc(sizeof(fr_t) * n);
    {
        fr_t f = *(fr_t*)FR.r1, k;
        for (size_t j = 1; j < n; j++) { fr_set_u64(&k, (u64)j); fr_mul(&f, &f, &k); }
        fr_inv(&f, &f); /* 1/(n-1)! */
        for (size_t j = n; j-- > 0;) { invfact[j] = f; if (j) { fr_set_u64(&k, (u64)j); fr_mul(&f, &f, &k); } }
    }
    fr_t *pa = (fr_t*)malloc(sizeof(fr_t) * n), *pb = (fr_t*)malloc(sizeof(fr_t) * n), *pc = (fr_t*)malloc(sizeof(fr_t) * n);
    fr_t* z = (fr_t*)malloc(sizeof(fr_t) * (n + 1));
    { /* the three interpolations are independent: one thread each */
        fq_job jobs[3] = {{ya, n, invfact, pa, z}, {yb, n, invfact, pb, NULL}, {yc, n, invfact, pc, NULL}};
        pthread_t th[2];
        pthread_create(&th[0], NULL, fq_job_run, &jobs[1]);
        pthread_create(&th[1], NULL, fq_job_run, &jobs[2]);
        fq_job_run(&jobs[0]);
        pthread_join(th[0], NULL);
        pthread_join(th[1], NULL);
    }
    poly_store(A, pa, n); poly_store(B, pb, n); poly_store(C, pc, n);
    if (rc == OR_OK) {
        /* P = A*B - C, degree <= 2n-2; h = P / z: reversed, rev(h) = rev(P) * rev(z)^-1 mod x^(n-1) */
        const size_t np = 2 * n - 1, m = n - 1;
        fr_t* P = (fr_t*)malloc(sizeof(fr_t) * np);
        fq_mul(P, np, pa, n, pb, n);
        for (size_t i = 0; i < n; i++) fr_sub(&P[i], &P[i], &pc[i]);
        fr_t *rz = (fr_t*)malloc(sizeof(fr_t) * (n + 1)), *rp = (fr_t*)malloc(sizeof(fr_t) * m), *g = (fr_t*)malloc(sizeof(fr_t) * m);
        for (size_t i = 0; i <= n; i++) rz[i] = z[n - i];
        for (size_t i = 0; i < m; i++) rp[i] = P[np - 1 - i];
        fq_series_inverse(rz, n + 1, m, g);
        fr_t* rh = (fr_t*)malloc(sizeof(fr_t) * m);
        fq_mul(rh, m, rp, m, g, m);
        fr_t* hh = (fr_t*)malloc(sizeof(fr_t) * m);
        for (size_t i = 0; i < m; i++) hh[i] = rh[m - 1 - i];
        /* the division must be exact: h * z == P */
        fr_t* chk = (fr_t*)malloc(sizeof(fr_t) * np);
        fq_mul(chk, np, hh, m, z, n + 1);
        if (memcmp(chk, P, sizeof(fr_t) * np) != 0) rc = OR_ERR_NOT_DIVISIBLE;
        poly_store(h, hh, m);
        free(P); free(rz); free(rp); free(g); free(rh); free(hh); free(chk);
    }
    free(ya); free(yb); free(yc); free(invfact); free(pa); free(pb); free(pc); free(z);
    return rc;
}
