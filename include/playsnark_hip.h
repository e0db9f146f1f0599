/* playsnark_hip.h -- C ABI of the MI355X-native prover hot path for nikkolasg/playsnark.
 *
 * The reference is a single Go package with no FFI; this header is the boundary a cgo shim
 * binds (INTEGRATION.md shows the shim).  Each entry point names the reference function it
 * replaces (file:line under /root/reference).  Plain pointers and sizes only; all host
 * buffers are caller-owned and never retained past return (cgo pointer rules); handles are
 * opaque, library-owned and explicitly destroyed.  Functions return 0 or a negative
 * PS_ERR_*; nothing aborts or throws across the boundary.  A ps_ctx is not thread-safe;
 * distinct contexts are.
 *
 * Byte formats (big-endian, what kyber's MarshalBinary produces [upstream]):
 *   scalar      32 B canonical Fr element
 *   PS_FMT_AFFINE      G1 96 B  x||y ;  G2 192 B  x_c1||x_c0||y_c1||y_c0 ; identity = 0x40,0,0,...
 *   PS_FMT_COMPRESSED  G1 48 B / G2 96 B ZCash compressed (flags 0x80 | 0x40 inf | 0x20 sign)
 * Outputs are always PS_FMT_AFFINE unless stated.
 */
#ifndef PLAYSNARK_HIP_H
#define PLAYSNARK_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PS_OK 0
#define PS_ERR_LENGTH (-1)        /* algebra.go:350-352 panic "mismatch of length between poly ..." */
#define PS_ERR_NOT_DIVISIBLE (-2) /* qap.go:158-160, pinochio.go:214-216 panic("apocalypse")        */
#define PS_ERR_ENCODING (-3)      /* non-canonical field element / point not on the curve           */
#define PS_ERR_HIP (-4)           /* HIP runtime failure; see ps_last_error()                        */
#define PS_ERR_ARG (-5)           /* NULL handle, unsupported size, sanityCheck (qap.go:177-189)    */
#define PS_ERR_NO_DEVICE (-6)     /* no gfx950 device visible: the library has no CPU fallback      */

#define PS_FMT_AFFINE 0
#define PS_FMT_COMPRESSED 1

#define PS_G1 1
#define PS_G2 2

typedef struct ps_ctx ps_ctx;         /* one device + stream + workspace                       */
typedef struct ps_points ps_points;   /* device-resident point vector (a CRS array)            */
typedef struct ps_scalars ps_scalars; /* device-resident Fr vector                             */
typedef struct ps_qap ps_qap;         /* device-resident sparse QAP + per-n tables             */

/* ABI revision of this header.  Structs only ever grow at the tail, and only together with this number; a caller checks
 * ps_abi_version() == PS_ABI_VERSION once after loading the library.  Key / device structs (ps_groth16_pk, ps_phgr13_ek,
 * ps_groth16_device, ..) MUST be zero-initialised by the caller (memset or `= {0}`): an optional array that is NULL is
 * "not present", and garbage in a member added by a later revision would be dereferenced.
 *   1  round 1;  2  Lagrange-form key arrays (lxi / lxi2 / lxi_t / lgsi), multi-device entries;
 *   3  ps_msm_info.window_table, ps_msm_set_tail, ps_ctx_set_table_budget, ps_qap_is_valid, ps_microbench_mad;
 *   4  ps_points_monomial_to_lagrange; index-range views build window tables of their own; PS_MSM_QUEUE 3 -> 4 (no struct changed) */
#define PS_ABI_VERSION 4
int ps_abi_version(void);
const char* ps_last_error(void);
const char* ps_version(void);
int ps_device_count(void);

/* ---- context ---- */
/* Thread safety: a ps_ctx (and the ps_qap made on it) belongs to one host thread at a time; distinct contexts may run
 * on distinct threads concurrently, ALSO over the same ps_points / ps_scalars arrays (a key uploaded once, proved with
 * from several threads): the arrays are read-only to the sums, their reference counts are atomic, and a window table a
 * prover attaches on first use is built once under the array's own lock.  ps_points_precompute with another window
 * size or -1 (release), and ps_*_free, must not race with sums over the same array. */
int ps_ctx_create(int device, ps_ctx** out);
void ps_ctx_destroy(ps_ctx* ctx);
int ps_ctx_sync(ps_ctx* ctx);
/* Raw HIP stream of the context (hipStream_t as void*), for callers that time with events. */
void* ps_ctx_stream(ps_ctx* ctx);

/* ---- CRS / evaluation-key arrays: the []G1 / []G2 slices of Groth16Setup
 *      (groth16.go:30-61: Xi, Xi2, NioLP, XiT) and PHGR13EvalKey (pinochio.go:37-62) ---- */
/* Encodings are validated (canonical field elements, curve equation, identity = 0x40 then zeros / 0xC0 then
 * zeros): PS_ERR_ENCODING otherwise.  Membership in the order-r subgroup is NOT tested here (it costs a
 * scalar multiplication per point); arrays that come from an untrusted source go through
 * ps_points_check_subgroup, as kyber's UnmarshalBinary would have rejected such points [upstream]. */
int ps_points_upload(ps_ctx* ctx, int group /*PS_G1|PS_G2*/, const uint8_t* pts, size_t n, int fmt,
                     ps_points** out);
/* Window table of a resident array: T[w][i] = 2^(c w) P[i] for every window w of a c-bit signed-digit
 * decomposition (c = window_bits; 0 picks it from the array length: 20 at 2^20 points, 13 windows).  CRS arrays are
 * fixed across proofs (groth16.go:30-61, pinochio.go:37-62), so the table is built once; every later sum over
 * the array -- or over a ps_points_slice of it -- then lets all its windows share ONE bucket set: 13 instead of
 * 16 bucket additions per 255-bit scalar and a sixteenth of the buckets to reduce.  Results are the same group
 * elements, bit for bit.  Costs (255 / c + 1) rows of 128 B (G1) / 256 B (G2) per point in device memory (1.7 GB for
 * 2^20 G1 points at c = 20);
 * at most 2^26 - 1 points.  Sums take the plain path when a table is absent, when ps_msm_set_window
 * forces a window size, when the arrays of a ps_msm_multi call do not all carry tables of one window size, or when the
 * cost model prefers it (short scalars: an int64 witness).  window_bits = -1 releases the table (after waiting for the
 * device: sums may still read it). */
int ps_points_precompute(ps_ctx* ctx, ps_points* p, int window_bits);
int ps_points_table_window(const ps_points* p); /* window bits of the table, 0 = none */
/* ps_groth16_prove / ps_phgr13_prove build the tables of their CRS arrays themselves on first use (keys of at least
 * 32 points; cached per key; about 9 GB for a 2^20-constraint Groth16 key, 15 GB for a PHGR13 one).
 * enable = 0 keeps the provers on the plain plan. */
int ps_ctx_set_tables(ps_ctx* ctx, int enable);
/* Memory policy of those tables.  A prover builds a table only when it fits: `bytes` >= 0 caps ONE table (0: none fit),
 * negative (the default) = what hipMemGetInfo reports as free, less a sixteenth of the device (at least 2 GiB) kept for
 * the sums' own workspaces.  A table that does not fit -- or whose allocation fails -- is not an error: the sums over that
 * array take the plain plan, which needs no memory beyond the array (ps_msm_info.window_table tells which ran), and the
 * array is not asked again until ps_points_precompute(p, -1) clears the mark.  An explicit ps_points_precompute still
 * fails with PS_ERR_HIP when the allocation fails. */
int ps_ctx_set_table_budget(ps_ctx* ctx, long long bytes);
/* *ok = 1 iff [r]P = O for every point of the array (GPU, ~400 group operations per point). */
int ps_points_check_subgroup(ps_ctx* ctx, const ps_points* p, int* ok);
/* out[i] = scalars[i] * G (fixed base).  GeneratePowersCommit (algebra.go:371-384) and the
 * commit loops of fullLinearPoly (groth16.go:254-264) / generateEvalCommit (pinochio.go:381-388)
 * reduce to this once the exponents are known. */
int ps_points_from_scalars(ps_ctx* ctx, int group, const ps_scalars* k, ps_points** out);
int ps_points_download(ps_ctx* ctx, const ps_points* p, size_t first, size_t n, uint8_t* out);
/* Same with the output format chosen: PS_FMT_COMPRESSED gives the 48 / 96-byte form of MarshalBinary
 * (pinochio.go:258-272), compressed on the GPU -- the at-rest form of key files. */
int ps_points_download_fmt(ps_ctx* ctx, const ps_points* p, size_t first, size_t n, int fmt, uint8_t* out);
size_t ps_points_len(const ps_points* p);
int ps_points_group(const ps_points* p);
/* A view of [first, first+n) sharing storage with `p` (index-range sharding, multi-GPU). */
int ps_points_slice(const ps_points* p, size_t first, size_t n, ps_points** out);
void ps_points_free(ps_points* p);

/* ---- scalar vectors: Poly = []Element (algebra.go:89) / Vector = []Value (algebra.go:13) ---- */
int ps_scalars_upload(ps_ctx* ctx, const uint8_t* be32, size_t n, ps_scalars** out);
/* Value.ToFieldElement = SetInt64 (curve.go:17-19): negatives map to r - |v|.  Sums over such a vector use
 * the short scalar |v| (64 bits: a quarter of the windows) and, for a negative value, the negated point --
 * the same group element for points of order r, which every CRS point is. */
int ps_scalars_upload_i64(ps_ctx* ctx, const int64_t* v, size_t n, ps_scalars** out);
/* Wrap n big-endian 32-byte scalars already resident in device memory (e.g. a torch tensor's
 * data_ptr()); the bytes are converted into a library-owned vector. */
int ps_scalars_from_device_be32(ps_ctx* ctx, const void* d_be32, size_t n, ps_scalars** out);
int ps_scalars_download(ps_ctx* ctx, const ps_scalars* s, size_t first, size_t n, uint8_t* out_be32);
size_t ps_scalars_len(const ps_scalars* s);
int ps_scalars_slice(const ps_scalars* s, size_t first, size_t n, ps_scalars** out);
void ps_scalars_free(ps_scalars* s);

/* ---- MSM: Poly.BlindEval (algebra.go:348-359), sumBlind (groth16.go:134-141), the NioLP loop
 *      (groth16.go:173-179), computeSolCommit (pinochio.go:222-229) ----
 * out = sum_i scalars[i] * points[i].  len(scalars) != len(points) returns PS_ERR_LENGTH, the
 * reference's panic at algebra.go:350-352.  `out` is 96 B (G1) or 192 B (G2), affine.
 * Size limit: windows x length < 2^32 digits (32-bit sort offsets): 2^28 full-width scalars per call,
 * PS_ERR_ARG beyond (split with ps_points_slice / ps_scalars_slice and add the parts with ps_points_sum).
 * Ordering: arrays returned by the asynchronous producers (ps_points_from_scalars,
 * ps_scalars_from_device_be32) may be passed on at once -- every sum waits for its own inputs. */
int ps_msm(ps_ctx* ctx, const ps_points* points, const ps_scalars* scalars, uint8_t* out);
/* Host-buffer convenience forms (upload + ps_msm). */
int ps_msm_be32(ps_ctx* ctx, const ps_points* points, const uint8_t* scalars_be32, size_t n, uint8_t* out);
int ps_msm_i64(ps_ctx* ctx, const ps_points* points, const int64_t* scalars, size_t n, uint8_t* out);
/* Asynchronous form: ps_msm_launch enqueues the sum and leaves the per-window sums on the device;
 * ps_msm_finish() waits for the OLDEST pending sum and folds it on the host.  Up to PS_MSM_QUEUE sums
 * may be pending on a context (PS_ERR_ARG beyond that): each runs on its own internal stream and
 * workspace, the accumulations chained in launch order, so a caller that keeps the queue full
 * (launch i+3, then finish i) hides each sum's sort and its latency-bound tail -- bucket fix-up,
 * reduction, host fold -- under its neighbours' accumulations.  Four is the measured optimum for 2^20-point sums
 * (2.70 ms per sum with three pending, 2.63 with four, 2.9 with five or six: the tail of the oldest sum is starved by
 * the accumulations queued behind it).  The one-call forms (ps_msm, ps_msm_be32, ps_msm_i64, ps_msm_multi, the provers)
 * need an empty queue (PS_ERR_ARG otherwise). */
#ifndef PS_MSM_QUEUE
#define PS_MSM_QUEUE 4
#endif
int ps_msm_launch(ps_ctx* ctx, const ps_points* points, const ps_scalars* scalars);
int ps_msm_finish(ps_ctx* ctx, uint8_t* out);
/* k sums over ONE scalar vector: out[i] = sum_j scalars[j] * points[i][j].  The digit sort runs once and
 * is shared; the arrays may mix G1 and G2.  This is the shape of computeSolCommit called nine times on
 * solution[diff:] (pinochio.go:231-241).  k <= PS_MSM_MULTI_MAX; every array must have the scalars'
 * length (PS_ERR_LENGTH otherwise, algebra.go:350-352). */
#define PS_MSM_MULTI_MAX 16
int ps_msm_multi(ps_ctx* ctx, const ps_points* const* points, size_t k, const ps_scalars* scalars, uint8_t* const* out);
/* Host-side conversion of ONE point between PS_FMT_AFFINE and PS_FMT_COMPRESSED (what the shim
 * needs to feed proof elements back to kyber's UnmarshalBinary).  Validates the encoding. */
int ps_point_convert(int group, int in_fmt, int out_fmt, const uint8_t* in, uint8_t* out);
/* Sum of k affine points (the per-GPU partial sums of a sharded MSM, after the RCCL gather). */
int ps_points_sum(int group, const uint8_t* pts, size_t k, uint8_t* out);
/* sum_i scalars[i] * points[i] for a handful of points (k <= 64) on the host: the fixed-point terms of a proof
 * (r Delta + Alpha ...) and the weighting of partial sums computed elsewhere.  Affine in, affine out. */
int ps_points_lincomb(int group, const uint8_t* pts, const uint8_t* scalars_be32, size_t k, uint8_t* out);
/* One sum over index-range shards held by several devices of THIS process: shard d is summed on ctxs[d] (the devices
 * work side by side), the partial sums are folded on the host.  The in-process form of the sharded MSM of SURVEY 8e
 * (a cgo caller cannot wrap a function call in one process per GPU). */
int ps_msm_multi_device(ps_ctx* const* ctxs, const ps_points* const* pts, const ps_scalars* const* scalars, size_t ndev, uint8_t* out);
/* Tuning / introspection of the last MSM on this context. */
typedef struct {
    int window_bits;   /* c */
    int windows;       /* W */
    uint64_t entries;  /* non-zero digits = bucket additions issued */
    uint64_t buckets;  /* W * 2^(c-1) */
    int slice;         /* sorted entries per accumulation thread */
    int window_table;  /* 1: the sum ran over the array's window table (ps_points_precompute), 0: the plain plan */
} ps_msm_info;
int ps_msm_last_info(ps_ctx* ctx, ps_msm_info* out);
int ps_msm_set_window(ps_ctx* ctx, int window_bits /* 0 = automatic, else 4..20 */);
int ps_msm_set_slice(ps_ctx* ctx, int entries /* sorted entries per accumulation thread; 0 = automatic */);
/* The tail of a sum (fix-up of cut buckets, bucket reduction): 0 = automatic (short sums -- fewer than 2^21 digits --
 * take shallow trees of lane-cooperative point additions, long ones the work-efficient chains), 1 = chains, 2 = trees.
 * Same group element, same bytes, either way (A/B runs and tests). */
int ps_msm_set_tail(ps_ctx* ctx, int mode);
/* Per-stage device time of the sum finished last, measured with HIP events on the streams its kernels
 * run on.  Stages: 0 digits (+counter memset), 1 scan, 2 scatter, 3 queue (bucket memset, and with
 * several sums in flight the wait for the previous sum's accumulation), 4 accumulate (the dominant
 * kernel, bracketed tightly), 5 fix-up, 6 bucket reduction. */
/* Measured issue rate of v_mad_u64_u32 (lane-operations per second, two waves per SIMD on every CU; ~1 ms): the
 * integer roofline the bucket additions are priced against, measured on the chip and at the clocks of the run. */
int ps_microbench_mad(ps_ctx* ctx, double* lane_mads_per_s);
#define PS_MSM_STAGES 7
int ps_ctx_set_timing(ps_ctx* ctx, int enable);
int ps_msm_last_stage_ms(ps_ctx* ctx, float ms[PS_MSM_STAGES]);

/* ---- QAP quotient: QAP.Quotient (qap.go:151-162) + computeAggregatePoly (qap.go:164-175) ----
 * The R1CS matrices (r1cs.go:78-101: rows = gates, columns = variables) are given in CSR with
 * int64 coefficients (the reference's Value = int, algebra.go:11).  The QAP domain is the
 * reference's {1..n} (qap.go:42-55, algebra.go:256-258). */
typedef struct {
    const uint32_t* row_ptr; /* n+1 */
    const uint32_t* col;     /* nnz */
    const int64_t* val;      /* nnz */
} ps_csr;
int ps_qap_create(ps_ctx* ctx, size_t n_gates, size_t n_vars, size_t n_io, const ps_csr* L, const ps_csr* R,
                  const ps_csr* O, ps_qap** out);
void ps_qap_free(ps_qap* q);
/* sol: n_vars scalars.  Outputs (any may be NULL): A, B, C aggregate polynomials (n coefficients
 * each) and h (n-1 coefficients), all device-resident.  PS_ERR_NOT_DIVISIBLE <=> "apocalypse". */
int ps_qap_quotient(ps_ctx* ctx, const ps_qap* q, const ps_scalars* sol, ps_scalars** A, ps_scalars** B,
                    ps_scalars** C, ps_scalars** h);
/* (*QAP).IsValid (qap.go:107-148): *valid = 1 iff left(x) right(x) - out(x) is divisible by z(x), i.e. iff the solution
 * satisfies every gate.  PS_ERR_ARG <=> sanityCheck's panic (qap.go:177-189). */
int ps_qap_is_valid(ps_ctx* ctx, const ps_qap* q, const ps_scalars* sol, int* valid);

/* computeAggregatePoly (qap.go:164-175) for ONE of the three polynomials: which = 0 left (A), 1 right (B), 2 out (C);
 * n coefficients.  No divisibility test (that needs all three: ps_qap_quotient).  The three parts of the quotient --
 * A, B and h through the h-only route -- are independent, so three GPUs can compute them side by side. */
int ps_qap_interpolate(ps_ctx* ctx, const ps_qap* q, const ps_scalars* sol, int which, ps_scalars** out);

/* Poly.Mul (algebra.go:92-105): out = a * b, len(a)+len(b)-1 coefficients (NTT product). */
int ps_poly_mul(ps_ctx* ctx, const ps_scalars* a, const ps_scalars* b, ps_scalars** out);

/* ---- whole-function drivers ---- */
typedef struct { /* the prover's part of Groth16Setup (groth16.go:30-61) */
    uint8_t alpha[96], beta[96], delta[96]; /* G1 */
    uint8_t beta2[192], delta2[192];        /* G2 */
    const ps_points* xi;                    /* n   G1 */
    const ps_points* xi2;                   /* n   G2 (declared []G1 at groth16.go:60) */
    const ps_points* nio_lp;                /* n_vars - (n_vars - n_io) ... see `diff` note */
    const ps_points* xi_t;                  /* n-1 G1 */
    /* Optional (NULL = absent): the same CRS in LAGRANGE form, as ps_groth16_setup also emits it --
     *   lxi[j-1]  = l_j(x) G1,  lxi2[j-1] = l_j(x) G2         l_j the Lagrange basis of the QAP domain {1..n} (qap.go:42-55)
     *   lxi_t[k-1] = lambda_k(x) t(x)/delta G1                 lambda_k the Lagrange basis of the nodes n+1..2n-1
     * With all three present the prover needs no polynomial in coefficient form: A(x) G = sum_j (L.s)_j lxi_j, likewise B,
     * and h(x) t(x)/delta G = sum_k h(n+k) lxi_t_k -- the interpolations and the division of the quotient (three quarters of
     * it) disappear, the proof is the same group elements.  A key made by the reference's NewGroth16TrustedSetup has only
     * the monomial arrays above; then the coefficients are computed as QAP.Quotient does. */
    const ps_points* lxi;                   /* n   G1 */
    const ps_points* lxi2;                  /* n   G2 */
    const ps_points* lxi_t;                 /* n-1 G1 */
} ps_groth16_pk;
/* Groth16Prove (groth16.go:122-211).  r, s are inputs (the reference draws them at :148,:158 and
 * keeps them in the proof, :203-206).  diff = n_vars - n_io is used as the first non-IO index
 * exactly as the reference does (groth16.go:175-177).
 * Internal forms, all giving the same group elements: C as one sum with the scalars s a_j + r b_j, or -- Lagrange-form keys
 * of >= 2^19 constraints -- B in G1 as a sum of its own over the wire values and s A + r B1 added on the host (DESIGN.md
 * section 5; PS_G16_B1_MIN_N in the environment of ps_ctx_create moves that threshold: a knob for tests and measurements). */
int ps_groth16_prove(ps_ctx* ctx, const ps_groth16_pk* pk, const ps_qap* q, const ps_scalars* sol,
                     const uint8_t r_be32[32], const uint8_t s_be32[32], uint8_t A[96], uint8_t B[192],
                     uint8_t C[96]);

/* One rank's share of Groth16Prove when the sums are sharded over `world` GPUs (one process each): rank g
 * takes its index range of every CRS array, rank 0 also the fixed points; A_part / B_part / C_part of all
 * ranks add up (ps_points_sum, after an all_gather) to the A, B, C of ps_groth16_prove.  Every rank
 * computes the quotient itself (beside its sums: the quotient runs on a stream of its own). */
int ps_groth16_prove_shard(ps_ctx* ctx, const ps_groth16_pk* pk, const ps_qap* q, const ps_scalars* sol,
                           const uint8_t r_be32[32], const uint8_t s_be32[32], int rank, int world, uint8_t A_part[96],
                           uint8_t B_part[192], uint8_t C_part[96]);

/* Groth16Prove over the devices of one process, every device holding only ITS index range of the CRS arrays: device d
 * of ndev holds Xi[range(n)], Xi2[range(n)], NioLP[range(nbIO)], XiT[range(n-1)] with range = the d-th of ndev
 * contiguous parts whose sizes differ by at most one (PS_ERR_LENGTH otherwise); the fixed points are read from dev[0].pk.
 * Each device has its own context, its own ps_qap of the circuit and its own copy of the solution.  With three or more
 * devices the parts of the quotient (A, B, h) are computed side by side on devices 0, 1, 2.  Same proof bytes as
 * ps_groth16_prove. */
typedef struct {
    ps_ctx* ctx;
    const ps_qap* qap;
    const ps_scalars* sol;
    ps_groth16_pk pk; /* rank-local arrays */
} ps_groth16_device;
int ps_groth16_prove_multi(const ps_groth16_device* dev, size_t ndev, const uint8_t r_be32[32], const uint8_t s_be32[32],
                           uint8_t A[96], uint8_t B[192], uint8_t C[96]);

typedef struct { /* PHGR13EvalKey (pinochio.go:37-62); ws is G2, every other array is G1 */
    const ps_points *vs, *ws, *ys, *vas, *was, *yas, *gsi, *vbs, *wbs, *ybs;
    /* Optional (NULL = absent): lgsi[k-1] = lambda_k(s) G1, the Lagrange form of gsi on the nodes n+1..2n-1 (n-1 points, as
     * ps_phgr13_setup emits it): hs = sum_k h(n+k) lgsi_k without interpolating h. */
    const ps_points* lgsi;
} ps_phgr13_ek;
typedef struct { /* PHGR13Proof (pinochio.go:180-203) */
    uint8_t vss[96], vass[96], wss[192], wass[96], yss[96], yass[96], hs[96], gz[96];
} ps_phgr13_proof;
/* PHGR13Prove (pinochio.go:207-254). */
int ps_phgr13_prove(ps_ctx* ctx, const ps_phgr13_ek* ek, const ps_qap* q, const ps_scalars* sol,
                    ps_phgr13_proof* out);

/* Host wall-clock split of the last ps_groth16_prove / ps_phgr13_prove on this context, in ms:
 * [0] quotient h(x) (SpMV, gate check, interpolations, division), [1] scalar preparation (Groth16) or
 * the h(s) sum (PHGR13), [2] the remaining sums incl. host folds, [3] total.  For reports only. */
#define PS_PROVE_PHASES 4
int ps_prove_last_phase_ms(ps_ctx* ctx, float ms[PS_PROVE_PHASES]);

/* ---- trusted setup on the device ---- */
typedef struct { uint8_t alpha[32], beta[32], delta[32], x[32], gamma[32]; } ps_groth16_toxic; /* groth16.go:15-26 */
typedef struct { /* type Groth16Setup (groth16.go:30-61) without the toxic waste */
    uint8_t alpha[96], beta[96], delta[96];        /* G1 */
    uint8_t beta2[192], delta2[192], gamma[192];   /* G2 */
    ps_points *xi, *xi2, *io_lp, *nio_lp, *xi_t;    /* caller frees with ps_points_free */
    ps_points *lxi, *lxi2, *lxi_t;                  /* the Lagrange form (see ps_groth16_pk); caller frees */
} ps_groth16_crs;
/* NewGroth16TrustedSetup (groth16.go:64-101) with the toxic waste supplied by the caller. */
int ps_groth16_setup(ps_ctx* ctx, const ps_qap* q, const ps_groth16_toxic* tw, ps_groth16_crs* out);

typedef struct { uint8_t s[32], av[32], aw[32], ay[32], rv[32], rw[32], beta[32], gamma[32]; } ps_phgr13_toxic; /* draw order of pinochio.go:99-138 */
typedef struct { /* PHGR13Setup (pinochio.go:28-35) without the toxic waste */
    ps_points *gsi, *vs, *ws, *ys, *vas, *was, *yas, *vbs, *wbs, *ybs;                  /* PHGR13EvalKey, pinochio.go:37-62 */
    uint8_t av[192], aw[96], ay[192], gamma[192], bgamma[96], bgamma2[192], yts[192]; /* PHGR13VerifKey, pinochio.go:64-91 */
    ps_points *vk_vs, *vk_ws, *vk_ys; /* vk.vs, vk.ws (G2), vk.ys over ALL variables; vs/ws/ys above are their [diff:] views */
    ps_points* lgsi;                  /* the Lagrange form of gsi (see ps_phgr13_ek) */
} ps_phgr13_crs;
/* NewPHGR13TrustedSetup (pinochio.go:93-176) with the toxic waste supplied by the caller. */
int ps_phgr13_setup(ps_ctx* ctx, const ps_qap* q, const ps_phgr13_toxic* tw, ps_phgr13_crs* out);
void ps_phgr13_crs_free(ps_phgr13_crs* crs); /* frees the 14 arrays */

/* ---- a reference-made key onto the fast route, without the toxic waste ----
 * The reference's setups emit the monomial arrays only -- Xi, Xi2, XiT (groth16.go:79-97), gsi (pinochio.go:101) -- and a prover
 * given those interpolates and divides (28 ms per Groth16 proof at 2^20 constraints where the Lagrange form of the same key
 * takes 18).  ps_groth16_setup emits both forms but needs the toxic waste, which "must be delete[d] after a trusted setup"
 * (groth16.go:13-14).  This is the one-time conversion of an array {x^i P}, i < cnt, into {l_j(x) P}, j < cnt, over the
 * group elements alone (the transposed interpolation of csrc/lagrange.hpp: ~230 cnt scalar multiplications of points; seconds
 * at 2^16 gates, under a minute at 2^20).  nodes = 0: the QAP domain 1..n, cnt = n (Xi -> lxi, Xi2 -> lxi2);
 * nodes = 1: the nodes n+1..2n-1, cnt = n-1 (XiT -> lxi_t, gsi -> lgsi).  Either group.  The result is byte-identical to the
 * array ps_groth16_setup / ps_phgr13_setup emit from the toxic waste; the caller frees it.  PS_ERR_LENGTH when the array is
 * not exactly cnt points long.  The points must lie in the subgroup of order r, as the points of a key do (the scalar
 * multiplications split their scalars with the curve's endomorphism, which acts as a scalar only there). */
int ps_points_monomial_to_lagrange(ps_ctx* ctx, const ps_qap* q, const ps_points* mono, int nodes, ps_points** out);

/* ---- verifiers (host-side ate pairing; the IO commitments go through the GPU MSM) ---- */
typedef struct { /* the verifier's part of Groth16Setup (groth16.go:30-61) */
    uint8_t alpha[96];                        /* G1 */
    uint8_t beta2[192], gamma[192], delta2[192]; /* G2 */
    const ps_points* io_lp;                   /* IoLP, nbVars - nbIO G1 points (`diff` convention) */
} ps_groth16_vk;
/* Groth16Verify (groth16.go:214-233); io = sol[:diff] as in groth16_test.go:29.  *ok = 1/0.
 * Both verifiers and ps_pairing_equal treat every point they are given as untrusted: canonical encoding, on
 * the curve and in the order-r subgroup ([r]P = O), PS_ERR_ENCODING otherwise -- the checks the reference gets
 * from kyber's UnmarshalBinary before its Verify functions ever see a point [upstream]. */
int ps_groth16_verify(ps_ctx* ctx, const ps_groth16_vk* vk, const ps_scalars* io, const uint8_t A[96], const uint8_t B[192],
                      const uint8_t C[96], int* ok);
typedef struct { /* PHGR13VerifKey (pinochio.go:64-91); the *_io arrays are vk.vs[:diff], vk.ws[:diff], vk.ys[:diff] */
    uint8_t av[192], aw[96], ay[192], gamma[192], bgamma[96], bgamma2[192], yts[192];
    const ps_points *vs_io, *ws_io, *ys_io; /* G1, G2, G1 */
} ps_phgr13_vk;
/* PHGR13Verify (pinochio.go:281-378).  *ok = 1/0. */
int ps_phgr13_verify(ps_ctx* ctx, const ps_phgr13_vk* vk, const ps_scalars* io, const ps_phgr13_proof* proof, int* ok);
/* e(a1, b1) == e(a2, b2) ?  (Pair, curve.go:36-38; host only) */
int ps_pairing_equal(const uint8_t a1_g1[96], const uint8_t b1_g2[192], const uint8_t a2_g1[96], const uint8_t b2_g2[192],
                     int* equal);

#ifdef __cplusplus
}
#endif
#endif
