/* TEST INFRASTRUCTURE ONLY -- CPU restatement (plain C) of the playsnark prover hot path.
 *
 * This is the *oracle*: the checker the HIP path is compared against.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product library
 * (playsnark_amd/csrc -> libplaysnark_hip.so) never links, includes or calls anything here.
 *
 * PARITY STATUS: "parity unpinned" at byte level -- the reference is Go, the image has no Go
 * toolchain, the curve arithmetic lives in absent third-party modules (go.mod:5-13:
 * drand/kyber v1.1.3, drand/kyber-bls12381 v0.2.1-0.20200920171356-02a6d1c7cc77,
 * kilic/bls12-381 v0.0.0-20200820230200-6b2c19996391) and the reference's tests hold no fixed
 * point/proof vectors.  The restatement is pinned by the reference's fixed Fr cases, its
 * algebraic identity tests (restated in tests/test_oracle_*.py) and the public BLS12-381
 * constants; it is cross-checked against an independent pure-Python twin (oracle/pyref.py)
 * through the committed fixtures in tests/golden/.
 *
 * Byte formats (all big-endian):
 *   Fr scalar  : 32 B canonical (kyber Scalar.MarshalBinary [upstream])
 *   G1 affine  : 96 B  x||y, identity = 0x40 then zeros (ZCash uncompressed)
 *   G2 affine  : 192 B x_c1||x_c0||y_c1||y_c0, identity likewise
 *   compressed : 48 B / 96 B ZCash (what kyber Point.MarshalBinary emits [upstream],
 *                pinochio.go:256-275)
 */
#ifndef PLAYSNARK_ORACLE_H
#define PLAYSNARK_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OR_OK 0
#define OR_ERR_LENGTH (-1)        /* algebra.go:350-352 panic: length mismatch          */
#define OR_ERR_NOT_DIVISIBLE (-2) /* qap.go:158-160 / pinochio.go:214-216 "apocalypse" */
#define OR_ERR_ENCODING (-3)

/* ---- Fr (kyber.Scalar; curve.go:17-31) ---- */
void or_fr_from_i64(int64_t v, uint8_t out[32]); /* SetInt64, curve.go:17-19 */
void or_fr_add(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]);
void or_fr_sub(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]);
void or_fr_mul(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]);
void or_fr_inv(const uint8_t a[32], uint8_t out[32]);

/* ---- Poly = []Element, lowest degree first (algebra.go:89-243) ---- */
void or_poly_mul(const uint8_t* a, size_t na, const uint8_t* b, size_t nb, uint8_t* out /* na+nb-1 */);
void or_poly_eval(const uint8_t* p, size_t n, const uint8_t x[32], uint8_t out[32]);
/* Div2 (algebra.go:140-159): q gets np-nd+1 coefficients, r gets nd-1 (padded with zeros). */
int or_poly_div2(const uint8_t* p, size_t np, const uint8_t* d, size_t nd, uint8_t* q, uint8_t* r);
/* Interpolate (algebra.go:254-281): out(i+1) = ys[i]; n coefficients. O(n^3) as the reference. */
void or_interpolate(const uint8_t* ys, size_t n, uint8_t* out);

/* ---- G1 / G2 (kyber.Point; curve.go:25-45) ---- */
/* Point.Mul(s, p): p == NULL means the generator (algebra.go:373). double-and-add. */
int or_g1_mul(const uint8_t s[32], const uint8_t* p96_or_null, uint8_t out[96]);
int or_g2_mul(const uint8_t s[32], const uint8_t* p192_or_null, uint8_t out[192]);
int or_g1_add(const uint8_t a[96], const uint8_t b[96], uint8_t out[96]);
int or_g2_add(const uint8_t a[192], const uint8_t b[192], uint8_t out[192]);
int or_g1_neg(const uint8_t a[96], uint8_t out[96]);
int or_g1_on_curve(const uint8_t a[96]);
int or_g2_on_curve(const uint8_t a[192]);
int or_g1_compress(const uint8_t a[96], uint8_t out[48]);
int or_g1_decompress(const uint8_t a[48], uint8_t out[96]);
int or_g2_compress(const uint8_t a[192], uint8_t out[96]);
int or_g2_decompress(const uint8_t a[96], uint8_t out[192]);

/* Poly.BlindEval (algebra.go:348-359): serial sum of Point.Mul + Point.Add.  This IS the
 * reference's "MSM"; np != ns returns OR_ERR_LENGTH like the panic at :350-352. */
int or_g1_blind_eval(const uint8_t* scalars, size_t ns, const uint8_t* points, size_t np, uint8_t out[96]);
int or_g2_blind_eval(const uint8_t* scalars, size_t ns, const uint8_t* points, size_t np, uint8_t out[192]);
/* same sum, int64 scalars lifted by SetInt64 (groth16.go:176-178, pinochio.go:222-229) */
int or_g1_blind_eval_i64(const int64_t* scalars, const uint8_t* points, size_t n, uint8_t out[96]);
int or_g2_blind_eval_i64(const int64_t* scalars, const uint8_t* points, size_t n, uint8_t out[192]);

/* Fast CPU Pippenger over the same inputs (the "fair" CPU baseline B1 of BASELINE.md; checked
 * against or_*_blind_eval in tests).  threads >= 1 (pthreads, windows split across threads). */
int or_g1_msm_pippenger(const uint8_t* scalars, const uint8_t* points, size_t n, int threads, uint8_t out[96]);
int or_g2_msm_pippenger(const uint8_t* scalars, const uint8_t* points, size_t n, int threads, uint8_t out[192]);

/* GeneratePowersCommit (algebra.go:371-384): out[i] = (shift * e^i) * G, i = 0..power. */
int or_g1_powers_commit(const uint8_t e[32], const uint8_t shift[32], size_t power, uint8_t* out);
int or_g2_powers_commit(const uint8_t e[32], const uint8_t shift[32], size_t power, uint8_t* out);

/* Synthetic point vectors: out[i] = (k0 + i*q) * G  (running add + batch normalisation). */
int or_g1_gen_points(const uint8_t k0[32], const uint8_t q[32], size_t n, uint8_t* out);
int or_g2_gen_points(const uint8_t k0[32], const uint8_t q[32], size_t n, uint8_t* out);

/* ---- QAP (qap.go) on a dense int R1CS, rows = gates, cols = variables (r1cs.go:78-101) ---- */
/* ToQAP/qapInterpolate (qap.go:35-93): polys out as m*n Fr each (variable-major); z as n+1 */
int or_to_qap_dense(const int64_t* L, const int64_t* Rm, const int64_t* O, size_t n, size_t m,
                    uint8_t* left, uint8_t* right, uint8_t* outp, uint8_t* z);
/* computeAggregatePoly (qap.go:164-175) from the per-variable polys; sol as Fr be32. */
void or_aggregate_poly(const uint8_t* polys /* m*n */, size_t n, size_t m, const uint8_t* sol, uint8_t* out /* n */);
/* Quotient (qap.go:151-162): literally Mul, Sub, Div2 on the aggregate polys. h gets n-1. */
int or_quotient_from_aggregates(const uint8_t* A, const uint8_t* B, const uint8_t* C, const uint8_t* z, size_t n, uint8_t* h);
/* Convenience for synthetic circuits: aggregate polys obtained by interpolating the three
 * value vectors yA = L.s, yB = R.s, yC = O.s on {1..n} (mathematically identical to
 * qap.go:164-175 by linearity), then the literal Mul/Sub/Div2.  O(n^3). */
int or_quotient_from_values(const uint8_t* yA, const uint8_t* yB, const uint8_t* yC, size_t n,
                            uint8_t* A, uint8_t* B, uint8_t* C, uint8_t* h);

/* CPU timing helper for bench.py's cpu_baseline leg: seconds for one call of the named op */
/* B1-h: the same polynomials by a quasi-linear CPU algorithm (NTT products, Newton basis,
 * product tree, series division) for n up to 2^20.  A, B, C: n coefficients; h: n-1.  OR_ERR_NOT_DIVISIBLE <=> "apocalypse". */
int or_fast_quotient(const uint8_t* yA, const uint8_t* yB, const uint8_t* yC, size_t n, uint8_t* A, uint8_t* B, uint8_t* C, uint8_t* h);

double or_now(void);

#ifdef __cplusplus
}
#endif
#endif
