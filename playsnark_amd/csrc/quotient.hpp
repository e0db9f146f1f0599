// QAP quotient on the reference's interpolation domain {1..n}  (qap.go:151-175).
//
// Reference:  A = sum_i s_i u_i(x), B, C likewise (computeAggregatePoly, qap.go:164-175), then
// h = (A*B - C) / z with z = prod_{i=1..n}(x - i) by schoolbook Mul and an O(n^3) long division
// (algebra.go:92-105,140-159), panic("apocalypse") when the remainder is non-zero.
//
// Here the same *polynomials* (bit-identical monomial coefficients; they are unique) come from:
//   1. y_A = L.s, y_B = R.s, y_C = O.s                        CSR SpMV over Fr
//      (A is the degree < n interpolant of y_A on {1..n}, by linearity of qap.go:168-173)
//   2. remainder == 0  <=>  y_A[j]*y_B[j] == y_C[j] for every gate j (z has the simple roots
//      1..n), checked point-wise: this is the "apocalypse" test
//   3. values on {1..n} -> Newton coefficients on the nodes 1,2,..: one convolution
//      d_k = sum_j (y_{j+1}/j!) * ((-1)^(k-j)/(k-j)!)
//   4. Newton -> monomial: blocks of 64 by a wave-wide Horner in the Newton basis, then
//      log2(n/64) levels  N = N_left + Z_left * N_right  with the subproduct tree
//      Z = prod (x - i) kept in NTT form in HBM (n log n Fr elements; 288 GB makes that free)
//   5. h = floor(A*B / z) (C = A*B mod z never needs interpolating): reversed power-series
//      division with rev(z)^-1 mod x^(n-1) precomputed per n              [Groth16Prove: A, B needed]
//   6. h alone: values of A, B, C on the nodes n+1..2n-1 by one convolution each, h there point-wise,
//      then steps 3-4 once on a second tree over the shifted nodes     [PHGR13Prove, QAP.Quotient]
// The element-wise work around the transforms (padding, products with stored transforms, adding halves
// back) is folded into the first load / last store of the NTT passes (ntt.hpp, NttFuse).
// Everything per-n (factorials, tree, z, inverse series) is built once in ps_qap_create.
#pragma once
#include <vector>

#include "ntt.hpp"

namespace ps {

struct QuotientCache {  // per-context NTT state
    NttTables tabs;
};
static inline void quotient_cache_free(QuotientCache* q) {
    if (!q) return;
    for (Fr* t : {q->tabs.fwd, q->tabs.inv, q->tabs.cfwd, q->tabs.cinv})
        if (t) (void)hipFree(t);
    delete q;
}

// ---------------------------------------------------------------------------------------
// wave-wide kernels on blocks of 64 coefficients (lane l holds coefficient l)
// ---------------------------------------------------------------------------------------
__device__ inline Fr fr_shfl_up1(const Fr& v) {  // lane l gets lane l-1's value, lane 0 gets 0
    Fr r;
#pragma unroll
    for (int j = 0; j < FR_L; j++)  // DPP wave_shr:1 (0x138), bound_ctrl: lane 0 reads 0 -- one v_mov_dpp per limb
        r.l[j] = __builtin_amdgcn_update_dpp(0, v.l[j], 0x138, 0xF, 0xF, true);
    return r;
}
__device__ inline Fr fr_shfl(const Fr& v, int src) {  // broadcast of lane `src` (uniform): v_readlane
    Fr r;
#pragma unroll
    for (int j = 0; j < FR_L; j++) r.l[j] = __builtin_amdgcn_readlane(v.l[j], src);
    return r;
}

// v * 2^-28 mod r (lazy): one Montgomery step.  r = 1 mod 2^28, so the digit is a negation and limb 0 of r
// contributes just the digit itself.
__device__ inline Fr fr_div_2p28(const Fr& v) {
    Fr o;
    i64 acc = (i64)v.l[0];
    const i64 m = (i64)((0u - (u32)acc) & FP_MASK);
    acc = (acc + m) >> 28;
#pragma unroll
    for (int i = 1; i < FR_L; i++) {
        acc += (i64)v.l[i] + m * (i64)fr_mod28(i);
        o.l[i - 1] = (i32)((u32)acc & FP_MASK);
        acc >>= 28;
    }
    o.l[FR_L - 1] = (i32)acc;
    return o;
}
// (up - c * coef) * 2^-28 mod r for a plain integer c < 2^28: 10 + 10 multiply-adds instead of the 200 of a
// full Montgomery product by the Montgomery form of c.
__device__ inline Fr fr_sub_mul_small_div_2p28(const Fr& up, const Fr& coef, u32 c) {
    // Every factor is handed over as a 32-bit SIGNED value (c < 2^28 and the Montgomery digit m < 2^28 are; the limbs are), so
    // that each term is one v_mad_i64_i32: with c as an unsigned 64-bit factor the compiler built every product from two
    // unsigned multiply-adds and a borrow chain -- 150 instructions per trip of k_newton_base's loop instead of 95.
    Fr o;
    const i32 nc = -(i32)c;
    i64 acc = (i64)up.l[0] + (i64)nc * (i64)coef.l[0];
    const i32 m = (i32)((0u - (u32)acc) & FP_MASK);
    acc = (acc + (i64)FP_MASK) >> 28;  // = (acc + m r_0) >> 28 with r_0 = 1: the low bits cancel, a ceiling (field.hpp, fr_mul)
#pragma unroll
    for (int i = 1; i < FR_L; i++) {
        acc += (i64)up.l[i] + (i64)nc * (i64)coef.l[i] + (i64)m * (i64)fr_mod28(i);
        o.l[i - 1] = (i32)((u32)acc & FP_MASK);
        acc >>= 28;
    }
    o.l[FR_L - 1] = (i32)acc;
    return o;
}

// block b: N(x) = sum_{k=0..63} d[64b+k] * prod_{i=64b+1}^{64b+k} (x - (off + i)), in place.
// Horner in the Newton basis: poly = poly*(x - (off+64b+k+1)) + d[64b+k], k = 63..0.
// off = 0 for the QAP domain {1..n}; off = n for the nodes n+1.. of the h-only path.
// The nodes are small integers, so each step multiplies by a plain c < 2^28 and divides by 2^28 (one
// Montgomery step, see above) instead of a full field multiplication.  To keep every term at the same
// scale, d[64b+L] enters with the factor 2^(-28 (63 - L)) (lane_scale), so all terms
// end with the factor 2^(-28*63); `unscale` = 2^(28*63) (Montgomery form) removes it.
// A batch of conversions (the two interpolations of a small circuit, interpolate2_on_1_to_n): `member64` blocks per member,
// the nodes start again at off + 1 for every member.
// (Round 4: the scale of d[64b+L] is set ONCE, by a product with lane_scale[L] = 2^(-28 (63 - L)), instead of one Montgomery step
// per loop trip for the lanes L < k -- a divergent ~45 instructions in 63 of the 64 trips: the kernel was 0.67 ms of the 9.4 ms
// quotient at 2^20 gates.)
__global__ void __launch_bounds__(256) k_newton_base(Fr* __restrict__ d, u32 nblocks64, u64 off, Fr unscale, u32 member64,
                                                     const Fr* __restrict__ lane_scale) {
    const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= nblocks64) return;
    const u64 base = 64ull * wave;
    const Fr mine = fr_mul(d[base + lane], lane_scale[lane]);
    Fr coef = fr_zero();
    u32 c = (u32)(off + 64ull * (wave % member64) + 64);  // node of step k = 63
    for (int k = 63; k >= 0; k--) {
        Fr up = fr_shfl_up1(coef);
        Fr dk = fr_shfl(mine, k);
        coef = fr_sub_mul_small_div_2p28(up, coef, c);
        if (lane == 0) coef = fr_add(coef, dk);
        c -= 1;
    }
    d[base + lane] = fr_mul(coef, unscale);
}

// block b: lower 64 coefficients of prod_{i=64b+1}^{64b+64} (x - (off + i))  (the x^64 term is implied)
__global__ void __launch_bounds__(256) k_subproduct_base(Fr* __restrict__ out, u32 nblocks64, u64 off) {
    const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= nblocks64) return;
    const u64 base = 64ull * wave;
    Fr coef = lane == 0 ? fr_one() : fr_zero();
    Fr c = fr_from_u64(off + base + 1);
    const Fr one = fr_one();
    for (int k = 0; k < 64; k++) {
        Fr up = fr_shfl_up1(coef);
        coef = fr_norm(fr_sub(up, fr_mul(c, coef)));
        c = fr_norm(fr_add(c, one));
    }
    out[base + lane] = coef;
}

// ---------------------------------------------------------------------------------------
// subproduct-tree kernels (the level steps themselves ride on the NTT passes, see NttFuse)
// ---------------------------------------------------------------------------------------
// full[node*2t + i] = i < t ? F[node*t + i] : (i == t ? 1 : 0)
__global__ void __launch_bounds__(256) k_tree_expand(Fr* __restrict__ full, const Fr* __restrict__ F, u64 total2, int logt) {
    u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total2) return;
    u64 t = 1ull << logt;
    u64 node = idx >> (logt + 1), i = idx & (2 * t - 1);
    full[idx] = i < t ? F[node * t + i] : (i == t ? fr_one() : fr_zero());
}
// prod[p*2t + i] = full[(2p)*2t + i] * full[(2p+1)*2t + i]
__global__ void __launch_bounds__(256) k_tree_pair_mul(Fr* __restrict__ prod, const Fr* __restrict__ full, u64 total, int log2t) {
    u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    u64 p = idx >> log2t, i = idx & ((1ull << log2t) - 1);
    prod[idx] = fr_mul(full[((2 * p) << log2t) + i], full[((2 * p + 1) << log2t) + i]);
}
// cyclic wrap of the monic x^(2t) term: coefficient 0 of every node carries a spurious +1
__global__ void __launch_bounds__(256) k_tree_fix(Fr* __restrict__ F2, u64 nodes, int log2t) {
    u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nodes) return;
    F2[p << log2t] = fr_norm(fr_sub(F2[p << log2t], fr_one()));
}
// zhat[p*2t + i] = full[(2p)*2t + i]  (left children only)
__global__ void __launch_bounds__(256) k_tree_take_left(Fr* __restrict__ zhat, const Fr* __restrict__ full, u64 total, int log2t) {
    u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    u64 p = idx >> log2t, i = idx & ((1ull << log2t) - 1);
    zhat[idx] = full[((2 * p) << log2t) + i];
}

// ---------------------------------------------------------------------------------------
// per-proof kernels
// ---------------------------------------------------------------------------------------
constexpr u32 SPMV_LONG_ROW = 512;  // rows with more non-zeros are summed by a whole workgroup

__global__ void __launch_bounds__(256) k_spmv(const u32* __restrict__ row_ptr, const u32* __restrict__ col,
                                              const Fr* __restrict__ val, const Fr* __restrict__ x, Fr* __restrict__ y, u32 n) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    if (row_ptr[r + 1] - row_ptr[r] > SPMV_LONG_ROW) return;  // k_spmv_long_rows owns it
    Fr acc = fr_zero();
    for (u32 e = row_ptr[r]; e < row_ptr[r + 1]; e++) {
        acc = fr_norm(fr_add(acc, fr_mul(val[e], x[col[e]])));
        if (((e - row_ptr[r]) & 31u) == 31u) acc = fr_reduce(acc);  // dense rows: keep |value| small
    }
    y[r] = acc;
}
// one workgroup per long row (e.g. the `const` variable's column in a transposed R1CS matrix)
__global__ void __launch_bounds__(256) k_spmv_long_rows(const u32* __restrict__ row_ptr, const u32* __restrict__ col,
                                                        const Fr* __restrict__ val, const Fr* __restrict__ x, Fr* __restrict__ y,
                                                        const u32* __restrict__ long_rows) {
    __shared__ Fr sm[256];
    const u32 r = long_rows[blockIdx.x];
    Fr acc = fr_zero();
    u32 cnt = 0;
    for (u32 e = row_ptr[r] + threadIdx.x; e < row_ptr[r + 1]; e += blockDim.x) {
        acc = fr_norm(fr_add(acc, fr_mul(val[e], x[col[e]])));
        if ((++cnt & 31u) == 0) acc = fr_reduce(acc);
    }
    sm[threadIdx.x] = fr_reduce(acc);
    __syncthreads();
    for (u32 stride = 128; stride > 0; stride >>= 1) {
        if (threadIdx.x < stride) sm[threadIdx.x] = fr_norm(fr_add(sm[threadIdx.x], sm[threadIdx.x + stride]));
        __syncthreads();
    }
    if (threadIdx.x == 0) y[r] = sm[0];
}
__global__ void __launch_bounds__(256) k_check_gates(const Fr* __restrict__ yA, const Fr* __restrict__ yB,
                                                     const Fr* __restrict__ yC, u32 n, u32* __restrict__ flag) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    if (!fr_is_zero(fr_sub(fr_mul(yA[r], yB[r]), yC[r]))) atomicOr(flag, 1u);
}
// q[i] = i < cnt ? P[top - i] : 0, i < total
__global__ void __launch_bounds__(256) k_rev_take(Fr* __restrict__ q, const Fr* __restrict__ P, u64 top, u64 cnt, u64 total) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    q[i] = i < cnt ? P[top - i] : fr_zero();
}
// t[i] = (i == 0 ? 2 : 0) - t[i]
__global__ void __launch_bounds__(256) k_two_minus(Fr* __restrict__ t, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr v = fr_neg(t[i]);
    if (i == 0) { Fr one = fr_one(); v = fr_add(v, fr_add(one, one)); }
    t[i] = fr_norm(v);
}
// out (plain limbs) = s*A + r*B, element-wise; A, B, s, r in Montgomery form
__global__ void __launch_bounds__(256) k_fr_lincomb_plain(u32* __restrict__ out, const Fr* __restrict__ A, const Fr* __restrict__ B,
                                                          Fr s, Fr r, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[8];
    fr_to_words8(w, fr_from_mont(fr_add(fr_mul(A[i], s), fr_mul(B[i], r))));
#pragma unroll
    for (int j = 0; j < 8; j++) out[8 * i + j] = w[j];
}
// ---- h-only path: values of A, B, C on the nodes n+1 .. 2n-1 by one convolution each ----
// Lagrange on {1..n}: P(x) = z(x) * sum_j w_j / (x - j) with w_j = P(j) / z'(j),
// z'(j) = (j-1)! (n-j)! (-1)^(n-j).  out[j-1] = w_j for j <= n, 0 up to `total`.
__global__ void __launch_bounds__(256) k_lagrange_weights(Fr* __restrict__ out, const Fr* __restrict__ y,
                                                          const Fr* __restrict__ invfact, u64 n, u64 total) {
    u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= total) return;
    if (j >= n) { out[j] = fr_zero(); return; }
    Fr w = fr_mul(fr_mul(y[j], invfact[j]), invfact[n - 1 - j]);
    out[j] = ((n - 1 - j) & 1) ? fr_neg(w) : w;
}
// the same for A, B, C at once: out = [w^A | w^B | w^C], each part L = 2^logL long (a batch of THREE transforms: the passes take any
// multiple of a tile -- until round 4's end the batch was padded to four with a transform of zeros, a third more work)
struct FrPtr3 { const Fr* p[3]; };
__global__ void __launch_bounds__(256) k_lagrange_weights3(Fr* __restrict__ out, FrPtr3 ys, const Fr* __restrict__ invfact, u64 n, int logL) {
    u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (3ull << logL)) return;
    const u64 j = idx & ((1ull << logL) - 1);
    const u32 k = (u32)(idx >> logL);
    if (j >= n) { out[idx] = fr_zero(); return; }
    const Fr* y = k == 0 ? ys.p[0] : k == 1 ? ys.p[1] : ys.p[2];
    Fr w = fr_mul(fr_mul(y[j], invfact[j]), invfact[n - 1 - j]);
    out[idx] = ((n - 1 - j) & 1) ? fr_neg(w) : w;
}
// out[d-1] = 1/d = (d-1)! / d!  for d = 1..cnt, 0 up to `total`
__global__ void __launch_bounds__(256) k_reciprocals(Fr* __restrict__ out, const Fr* __restrict__ fact,
                                                     const Fr* __restrict__ invfact, u64 cnt, u64 total) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    out[i] = i < cnt ? fr_mul(fact[i], invfact[i + 1]) : fr_zero();
}
// hv[i] = h(n+1+i) = z(n+1+i) * S_A * S_B - S_C with S_P = (w_P * recip)[n+1+i], i < n-1;
// z(n+k) = (n+k-1)! / (k-1)!.  The S_* arrays hold the cyclic convolutions (S_P(k) at index n-2+k).
__global__ void __launch_bounds__(256) k_h_values(Fr* __restrict__ hv, const Fr* __restrict__ SA, const Fr* __restrict__ SB,
                                                  const Fr* __restrict__ SC, const Fr* __restrict__ fact,
                                                  const Fr* __restrict__ invfact, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 >= n) return;
    const u64 idx = n - 1 + i;
    Fr zs = fr_mul(fact[n + i], invfact[i]);
    hv[i] = fr_norm(fr_sub(fr_mul(fr_mul(zs, SA[idx]), SB[idx]), SC[idx]));
}
// ---- trusted-setup helpers ----
struct FrPow2Table { Fr p[32]; };  // p[k] = x^(2^k)
// out[i] = shift * x^i  (GeneratePowersCommit's exponents, algebra.go:371-384)
__global__ void __launch_bounds__(256) k_fr_powers(Fr* __restrict__ out, FrPow2Table tab, Fr shift, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr acc = shift;
    for (int k = 0; k < 32; k++)
        if ((i >> k) & 1) acc = fr_mul(acc, tab.p[k]);
    out[i] = acc;
}
// l_j(x) = zx * w_j / (x - j), w_j = (-1)^(n-j) / ((j-1)! (n-j)!), j = 1..n   (the Lagrange basis
// behind Interpolate, algebra.go:254-338, evaluated at the secret point)
__global__ void __launch_bounds__(256) k_lagrange_at(Fr* __restrict__ out, const Fr* __restrict__ invfact, Fr x, Fr zx, u64 n) {
    u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    u64 j = idx + 1;
    Fr d = fr_inv(fr_norm(fr_sub(x, fr_from_u64(j))));
    Fr w = fr_mul(invfact[j - 1], invfact[n - j]);
    if ((n - j) & 1) w = fr_neg(w);
    out[idx] = fr_mul(fr_mul(zx, w), d);
}
// partial products of (x - j), j = 1..n: one value per block
__global__ void __launch_bounds__(256) k_prod_x_minus_j(Fr* __restrict__ partial, Fr x, u64 n) {
    __shared__ Fr sm[256];
    u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    sm[threadIdx.x] = idx < n ? fr_norm(fr_sub(x, fr_from_u64(idx + 1))) : fr_one();
    __syncthreads();
    for (u32 stride = 128; stride > 0; stride >>= 1) {
        if (threadIdx.x < stride) sm[threadIdx.x] = fr_mul(sm[threadIdx.x], sm[threadIdx.x + stride]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}
// out[i] = (beta*u[i] + alpha*v[i] + w[i]) * (i < diff ? ginv : dinv)   (linearPolyForVar, groth16.go:238-250)
__global__ void __launch_bounds__(256) k_linear_poly(Fr* __restrict__ out, const Fr* __restrict__ u, const Fr* __restrict__ v,
                                                     const Fr* __restrict__ w, Fr alpha, Fr beta, Fr ginv, Fr dinv, u64 diff, u64 m) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    Fr lp = fr_norm(fr_add(fr_add(fr_mul(beta, u[i]), fr_mul(alpha, v[i])), w[i]));
    out[i] = fr_mul(lp, i < diff ? ginv : dinv);
}
// out[i] = scale * in[i]   (generateEvalCommit's `Mul(p.Eval(s), shift)`, pinochio.go:381-388)
__global__ void __launch_bounds__(256) k_fr_scale(Fr* __restrict__ out, const Fr* __restrict__ in, Fr scale, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fr_mul(in[i], scale);
}
__global__ void __launch_bounds__(64) k_set_one(Fr* __restrict__ p, u64 idx) {
    if (threadIdx.x == 0 && blockIdx.x == 0) p[idx] = fr_one();
}

// ---------------------------------------------------------------------------------------
// per-n tables
// ---------------------------------------------------------------------------------------
struct QapTables {
    u64 n = 0, np = 0;
    int lognp = 0;
    Fr* invfact = nullptr;       // np : 1/j!
    Fr* vhat = nullptr;          // 2np: NTT of v_j = (-1)^j / j!
    std::vector<Fr*> zhat;       // [logs] -> np elements, logs = 7..lognp
    Fr* z = nullptr;             // n+1 coefficients of prod (x - i), i = 1..n
    Fr* ghat = nullptr;          // NTT_{2^ph} of rev(z)^-1 mod x^(n-1)
    int ph = 0, pp = 0;          // log sizes of the division / product transforms
    // h-only path: h is interpolated from its values on the nodes n+1 .. 2n-1
    Fr* fact2 = nullptr;         // 2np: j!            (invfact above has 2np entries as well)
    Fr* rhat = nullptr;          // 2np: NTT of 1/d, d = 1..2n-2
    u64 np_h = 0;                // power of two >= max(64, n-1)
    int lognp_h = 0;
    Fr* vhat_h = nullptr;        // 2np_h: NTT of v (== vhat when np_h == np)
    std::vector<Fr*> zhat_h;     // subproduct tree over the nodes n+1 .. n+np_h
    // work buffers
    Fr *t1 = nullptr, *data = nullptr, *scratch = nullptr, *pa = nullptr, *pb = nullptr;
    Fr* newton_scale = nullptr;  // 64: 2^(-28 (63 - L)), the scale lane L's coefficient enters k_newton_base with
    Fr* s4 = nullptr;            // 8np: the two interpolations of Groth16's route as ONE batch of transforms (any n, round 4) ...
    bool batch_h = false;        // ... and the three convolutions of the h-values path as a batch of three (every size since round 4's end)
    std::vector<void*> owned;
    void free_all() {
        for (void* p : owned) (void)hipFree(p);
        owned.clear();
        zhat.clear();
    }
};

#ifndef PS_QT_BATCH_MAX_LOG
#define PS_QT_BATCH_MAX_LOG 40
#endif
#ifndef QT_BATCH_MAX_L
#define QT_BATCH_MAX_L (1ull << PS_QT_BATCH_MAX_LOG)  // transforms up to this length leave most of the chip idle: the h-values path batches its three convolutions
#endif
#define QT_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return _e; } while (0)

static inline hipError_t qt_alloc(QapTables& qt, Fr** p, u64 count) {
    hipError_t e = hipMalloc((void**)p, sizeof(Fr) * (count ? count : 1));
    if (e == hipSuccess) qt.owned.push_back(*p);
    return e;
}

// Newton coefficients on the nodes off+1, off+2, .. (np of them, zero beyond the true length) -> monomial
// coefficients, in place
// `members` (a power of two) conversions at once: data and scratch hold members x np elements, every member on the same nodes
static inline hipError_t newton_to_monomial(const NttTables& tabs, hipStream_t st, u64 np, int lognp, const std::vector<Fr*>& zhat,
                                            u64 off, Fr* data, Fr* scratch, const Fr* lane_scale, u64 members = 1) {
    Fr unscale = fr_from_u64(1ull << 28);  // 2^(28*63), Montgomery form
    {
        const Fr two28 = unscale;
        for (int i = 1; i < 63; i++) unscale = fr_mul(unscale, two28);
    }
    hipLaunchKernelGGL(k_newton_base, dim3(nblk(np * members)), dim3(256), 0, st, data, (u32)(np * members / 64), off, unscale, (u32)(np / 64),
                       lane_scale);
    for (int logs = 7; logs <= lognp; logs++) {
        // scratch = NTT(upper halves of the nodes, zero-padded) * zhat ; data = lower halves + INTT(scratch):
        // prepare, multiply and combine ride on the first load / last store of the two transforms
        NttFuse f;
        f.ld = NTT_LD_UPPER_HALF; f.ld_src = data; f.logs = logs;
        NttFuse g;
        g.st = NTT_ST_COMBINE; g.st_dst = data; g.logs = logs;
        QT_TRY(ntt_conv(tabs, st, scratch, np * members, logs, f, g, zhat[logs], np - 1));
    }
    return hipGetLastError();
}

// values y[0..cnt) = f(off+1 .. off+cnt)  ->  monomial coefficients of the degree < cnt interpolant, in qt.data.
// The divided differences of equally spaced nodes do not depend on `off`: d_k = sum_j (y_j / j!) (-1)^(k-j) / (k-j)!.
static inline hipError_t interpolate_on_nodes(const NttTables& tabs, hipStream_t st, const QapTables& qt, const Fr* y, u64 cnt,
                                              u64 np, int lognp, const Fr* vhat, const std::vector<Fr*>& zhat, u64 off) {
    // t1 = INTT(NTT(y_j / j!, zero-padded) * vhat); data = its first cnt entries, zero up to np
    NttFuse f;
    f.ld = NTT_LD_SCALE_PAD; f.ld_src = y; f.ld_aux = qt.invfact; f.cnt = cnt;
    QT_TRY(hipMemsetAsync(qt.data, 0, sizeof(Fr) * np, st));
    NttFuse g;
    g.st = NTT_ST_TAKE; g.st_dst = qt.data; g.cnt = cnt;
    QT_TRY(ntt_conv(tabs, st, qt.t1, 2 * np, lognp + 1, f, g, vhat));
    return newton_to_monomial(tabs, st, np, lognp, zhat, off, qt.data, qt.scratch, qt.newton_scale);
}
// values y[0..n) = f(1..n)  ->  monomial coefficients of the degree < n interpolant, in qt.data
static inline hipError_t interpolate_on_1_to_n(const NttTables& tabs, hipStream_t st, const QapTables& qt, const Fr* y) {
    return interpolate_on_nodes(tabs, st, qt, y, qt.n, qt.np, qt.lognp, qt.vhat, qt.zhat, 0);
}

// Two interpolations on {1..n} as ONE batch (small circuits, where a transform is a handful of workgroups and every launch is
// latency; qt.s4 = [t1: 2 x 2np | data: 2 x np | scratch: 2 x np]): the coefficients of member b end up in data[b * np ..].
static inline hipError_t interpolate2_on_1_to_n(const NttTables& tabs, hipStream_t st, const QapTables& qt, const Fr* y0, const Fr* y1,
                                                Fr* out0, Fr* out1) {
    const u64 np = qt.np, L = 2 * np;
    Fr *t1 = qt.s4, *data = qt.s4 + 2 * L, *scratch = data + 2 * np;
    // y_j / j! of both members on the way in, the first n entries of both convolutions on the way out (zero up to np)
    NttFuse f;
    f.ld = NTT_LD_SCALE_PAD; f.ld_src = y0; f.ld_src2 = y1; f.ld_aux = qt.invfact; f.cnt = qt.n; f.batch_log = qt.lognp + 1;
    NttFuse g;
    g.st = NTT_ST_TAKE; g.st_dst = data; g.cnt = qt.n; g.batch_log = qt.lognp + 1; g.st_member_log = qt.lognp;
    if (qt.n < np) QT_TRY(hipMemsetAsync(data, 0, sizeof(Fr) * 2 * np, st));
    QT_TRY(ntt_conv(tabs, st, t1, 2 * L, qt.lognp + 1, f, g, qt.vhat, L - 1));
    QT_TRY(newton_to_monomial(tabs, st, np, qt.lognp, qt.zhat, 0, data, scratch, qt.newton_scale, 2));
    QT_TRY(hipMemcpyAsync(out0, data, sizeof(Fr) * qt.n, hipMemcpyDeviceToDevice, st));
    QT_TRY(hipMemcpyAsync(out1, data + np, sizeof(Fr) * qt.n, hipMemcpyDeviceToDevice, st));
    return hipGetLastError();
}

// h alone (PHGR13Prove and QAP.Quotient need no A, B coefficients): with y_P = P(1..n),
//   P(n+k) = z(n+k) * sum_j w_j^P / (n+k-j)        one cyclic convolution of length 2np per polynomial
//   h(n+k) = z(n+k) S_A(k) S_B(k) - S_C(k)         k = 1..n-1 (z(n+k) != 0)
// and h (degree <= n-2) is the interpolant of those n-1 values on the nodes n+1..2n-1: ONE values ->
// monomial conversion instead of two plus a product and a division.  Same polynomial, bit for bit.
// hv[k-1] = h(n+k), k = 1..n-1, into qt.scratch: three cyclic convolutions and one element-wise kernel.  These values
// are all a prover needs of h when its key carries the Lagrange-form points of the nodes n+1..2n-1 (ps_groth16_pk.lxi_t,
// ps_phgr13_ek.lgsi): h(x) G = sum_k h(n+k) lambda_k(x) G, no interpolation at all.
static inline hipError_t quotient_h_values(const NttTables& tabs, hipStream_t st, const QapTables& qt, const Fr* yA, const Fr* yB,
                                           const Fr* yC) {
    const u64 n = qt.n, L = 2 * qt.np;
    if (n < 2) return hipSuccess;
    const Fr* ys[3] = {yA, yB, yC};
    Fr* S[3] = {qt.t1, qt.pa, qt.pb};
    if (qt.s4 && qt.batch_h) {
        // Short transforms are latency: a 2^11-point transform is two workgroups walking eleven butterfly stages, ~40 us for
        // the forward and inverse pair whatever the chip could do beside it (kernel trace of Groth16Prove on 2^10 constraints:
        // 0.23 of the quotient's 0.37 ms).  The three convolutions share the kernel 1/d, so they run as one batch of THREE
        // transforms -- one set of launches instead of three.  (Until the end of round 4 the batch was padded to four with a
        // transform of zeros and used only below 2^16 gates, where the padding cost less than the launches: as a batch of
        // three it wins at every size -- 2^16 gates 0.49 -> 0.33 ms, 2^18 0.77 -> 0.63, 2^20 2.16 -> 2.00.)
        const int logL = qt.lognp + 1;
        hipLaunchKernelGGL(k_lagrange_weights3, dim3(nblk(3 * L)), dim3(256), 0, st, qt.s4, FrPtr3{{yA, yB, yC}}, qt.invfact, n, logL);
        QT_TRY(ntt_conv(tabs, st, qt.s4, 3 * L, logL, NttFuse(), NttFuse(), qt.rhat, L - 1));
        for (int k = 0; k < 3; k++) S[k] = qt.s4 + (u64)k * L;
    } else
    for (int k = 0; k < 3; k++) {
        hipLaunchKernelGGL(k_lagrange_weights, dim3(nblk(L)), dim3(256), 0, st, S[k], ys[k], qt.invfact, n, L);
        QT_TRY(ntt_conv(tabs, st, S[k], L, qt.lognp + 1, NttFuse(), NttFuse(), qt.rhat));
    }
    hipLaunchKernelGGL(k_h_values, dim3(nblk(n - 1)), dim3(256), 0, st, qt.scratch, (const Fr*)S[0], (const Fr*)S[1], (const Fr*)S[2],
                       (const Fr*)qt.fact2, (const Fr*)qt.invfact, n);
    return hipGetLastError();
}
static inline hipError_t quotient_h_only(const NttTables& tabs, hipStream_t st, const QapTables& qt, const Fr* yA, const Fr* yB,
                                         const Fr* yC, Fr* h_out) {
    const u64 n = qt.n;
    if (n < 2) return hipSuccess;
    QT_TRY(quotient_h_values(tabs, st, qt, yA, yB, yC));
    Fr* hv = qt.scratch;
    QT_TRY(interpolate_on_nodes(tabs, st, qt, hv, n - 1, qt.np_h, qt.lognp_h, qt.vhat_h, qt.zhat_h, n));
    return hipMemcpyAsync(h_out, qt.data, sizeof(Fr) * (n - 1), hipMemcpyDeviceToDevice, st);
}

// h = floor(A*B / z): n-1 coefficients into h_out (Montgomery); A, B have n coefficients
static inline hipError_t quotient_from_AB(const NttTables& tabs, hipStream_t st, const QapTables& qt, const Fr* A, const Fr* B, Fr* h_out) {
    const u64 n = qt.n;
    if (n < 2) return hipSuccess;
    const u64 Sp = 1ull << qt.pp, Sh = 1ull << qt.ph;
    // pb = INTT(NTT(A) * NTT(B)): the zero-padding rides on the loads, the product on B's last store
    NttFuse fa;
    fa.ld = NTT_LD_PAD; fa.ld_src = A; fa.cnt = n;
    QT_TRY(ntt_run<false>(tabs, st, qt.pa, Sp, qt.pp, fa));
    NttFuse fb;
    fb.ld = NTT_LD_PAD; fb.ld_src = B; fb.cnt = n;
    QT_TRY(ntt_conv(tabs, st, qt.pb, Sp, qt.pp, fb, NttFuse(), qt.pa));
    // q_i = P_{2n-2-i}, i < n-1, times the inverse series; h_{n-2-i} = (q*g)_i
    if (qt.ph == 0) {  // n = 2: one coefficient, transforms of size 1
        hipLaunchKernelGGL(k_rev_take, dim3(1), dim3(256), 0, st, qt.pa, (const Fr*)qt.pb, 2 * n - 2, n - 1, Sh);
        hipLaunchKernelGGL(k_fr_pointwise_mul, dim3(1), dim3(256), 0, st, qt.pa, (const Fr*)qt.pa, (const Fr*)qt.ghat, Sh);
        hipLaunchKernelGGL(k_rev_take, dim3(1), dim3(256), 0, st, h_out, (const Fr*)qt.pa, n - 2, n - 1, n - 1);
        return hipGetLastError();
    }
    NttFuse fq;
    fq.ld = NTT_LD_REV_PAD; fq.ld_src = qt.pb; fq.top = 2 * n - 2; fq.cnt = n - 1;
    NttFuse fh;
    fh.st = NTT_ST_REV_TAKE; fh.st_dst = h_out; fh.top = n - 2; fh.cnt = n - 1;
    QT_TRY(ntt_conv(tabs, st, qt.pa, Sh, qt.ph, fq, fh, qt.ghat));
    return hipGetLastError();
}

// Build every per-n table.  Host arithmetic is limited to the factorial table (n' field
// multiplications and one inversion, with the library's own host-compiled field code).
// j! and 1/j! for j < 2np, and v_j = (-1)^j / j! for j < np (host arithmetic: 4np multiplications, 1 inversion)
struct FactTables {
    std::vector<Fr> fact, inv, v;
};
static inline u64 qap_np(u64 n) { return 1ull << ilog2_ceil(n < 64 ? 64 : n); }
static inline FactTables fact_tables(u64 np) {
    FactTables t;
    const u64 nf = 2 * np;
    t.fact.resize(nf); t.inv.resize(nf); t.v.resize(np);
    Fr f = fr_one();
    for (u64 j = 0; j < nf; j++) {
        if (j > 0) f = fr_mul(f, fr_from_u64(j));
        t.fact[j] = f;
    }
    Fr finv = fr_inv(t.fact[nf - 1]);
    for (u64 j = nf; j-- > 0;) {  // walk down: 1/(j-1)! = (1/j!) * j
        t.inv[j] = finv;
        if (j > 0) finv = fr_mul(finv, fr_from_u64(j));
    }
    for (u64 j = 0; j < np; j++) t.v[j] = (j & 1) ? fr_neg(t.inv[j]) : t.inv[j];
    return t;
}

static inline hipError_t qap_tables_build(NttTables& tabs, hipStream_t st, QapTables& qt, u64 n, const FactTables* ft = nullptr) {
    qt.n = n;
    int lg = ilog2_ceil(n < 64 ? 64 : n);
    qt.lognp = lg;
    qt.np = 1ull << lg;
    const u64 np = qt.np;
    qt.pp = ilog2_ceil(2 * n - 1);
    qt.ph = n >= 2 ? ilog2_ceil(2 * (n - 1) - 1 > 0 ? 2 * (n - 1) - 1 : 1) : 0;
    QT_TRY(ntt_tables_ensure(tabs, lg + 1, st));
    const u64 Sp = 1ull << qt.pp, Sh = 1ull << qt.ph;
    const u64 big = 2 * np > Sp ? 2 * np : Sp;
    QT_TRY(qt_alloc(qt, &qt.invfact, 2 * np));
    QT_TRY(qt_alloc(qt, &qt.fact2, 2 * np));
    QT_TRY(qt_alloc(qt, &qt.rhat, 2 * np));
    QT_TRY(qt_alloc(qt, &qt.vhat, 2 * np));
    QT_TRY(qt_alloc(qt, &qt.t1, big));
    QT_TRY(qt_alloc(qt, &qt.data, np));
    QT_TRY(qt_alloc(qt, &qt.scratch, big));
    QT_TRY(qt_alloc(qt, &qt.pa, big));
    QT_TRY(qt_alloc(qt, &qt.pb, big));
    // Both interpolations of Groth16's route run as one batch at every size: with twice the workgroups per pass the loads of
    // one round hide under the butterflies of the other (A/B on one box at 2^20: quotient 10.6 -> 9.9 ms).  The h-values path
    // batches its three convolutions as FOUR transforms (batches are powers of two) only while that is latency, not work
    // (at 2^20 the batch of four costs 3.1 ms against 2.5 for three single transforms).
    QT_TRY(qt_alloc(qt, &qt.s4, 8 * np));
    qt.batch_h = 2 * np <= QT_BATCH_MAX_L;
    QT_TRY(qt_alloc(qt, &qt.z, n + 1));
    {
        Fr tab[64];
        const Fr inv28 = fr_inv(fr_from_u64(1ull << 28));
        tab[63] = fr_one();
        for (int L = 62; L >= 0; L--) tab[L] = fr_mul(tab[L + 1], inv28);
        QT_TRY(qt_alloc(qt, &qt.newton_scale, 64));
        QT_TRY(hipMemcpyAsync(qt.newton_scale, tab, sizeof tab, hipMemcpyHostToDevice, st));
        QT_TRY(hipStreamSynchronize(st));  // (tab is on this stack frame)
    }
    QT_TRY(qt_alloc(qt, &qt.ghat, Sh));
    // ---- factorials up to 2np - 1 (host; ps_qap_create computes them on a thread of its own) ----
    {
        FactTables local;
        if (!ft) { local = fact_tables(np); ft = &local; }
        const u64 nf = 2 * np;
        QT_TRY(hipMemcpyAsync(qt.invfact, ft->inv.data(), sizeof(Fr) * nf, hipMemcpyHostToDevice, st));
        QT_TRY(hipMemcpyAsync(qt.fact2, ft->fact.data(), sizeof(Fr) * nf, hipMemcpyHostToDevice, st));
        QT_TRY(hipMemcpyAsync(qt.t1, ft->v.data(), sizeof(Fr) * np, hipMemcpyHostToDevice, st));
        QT_TRY(hipStreamSynchronize(st));
    }
    hipLaunchKernelGGL(k_fr_copy_pad, dim3(nblk(2 * np)), dim3(256), 0, st, qt.vhat, qt.t1, np, 2 * np);
    QT_TRY(ntt_run<false>(tabs, st, qt.vhat, 2 * np, lg + 1));
    // h-only path: v again at the size of its own interpolation, and the reciprocals 1/d, d <= 2n-2
    {
        int lgh = ilog2_ceil(n - 1 < 64 ? 64 : n - 1);
        qt.lognp_h = lgh;
        qt.np_h = 1ull << lgh;
        if (qt.np_h == np) {
            qt.vhat_h = qt.vhat;
        } else {
            QT_TRY(qt_alloc(qt, &qt.vhat_h, 2 * qt.np_h));
            hipLaunchKernelGGL(k_fr_copy_pad, dim3(nblk(2 * qt.np_h)), dim3(256), 0, st, qt.vhat_h, qt.t1, qt.np_h, 2 * qt.np_h);
            QT_TRY(ntt_run<false>(tabs, st, qt.vhat_h, 2 * qt.np_h, lgh + 1));
        }
        hipLaunchKernelGGL(k_reciprocals, dim3(nblk(2 * np)), dim3(256), 0, st, qt.rhat, (const Fr*)qt.fact2, (const Fr*)qt.invfact,
                           n >= 2 ? 2 * n - 2 : (u64)0, 2 * np);
        QT_TRY(ntt_run<false>(tabs, st, qt.rhat, 2 * np, lg + 1));
    }
    // ---- subproduct trees: nodes 1..np (the QAP domain) and n+1..n+np_h (the h-only path) ----
    Fr *F = qt.data, *F2 = qt.scratch, *full = qt.t1;  // F: np, full: 2np, F2: np
    auto build_tree = [&](u64 tnp, int tlg, u64 off, std::vector<Fr*>& zhat) -> hipError_t {
        zhat.assign(tlg + 1, nullptr);
        F = qt.data; F2 = qt.scratch;
        hipLaunchKernelGGL(k_subproduct_base, dim3(nblk(tnp)), dim3(256), 0, st, F, (u32)(tnp / 64), off);
        for (int logt = 6; logt < tlg; logt++) {
            const int log2t = logt + 1;
            hipLaunchKernelGGL(k_tree_expand, dim3(nblk(2 * tnp)), dim3(256), 0, st, full, F, 2 * tnp, logt);
            QT_TRY(ntt_run<false>(tabs, st, full, 2 * tnp, log2t));
            Fr* zh;
            QT_TRY(qt_alloc(qt, &zh, tnp));
            zhat[log2t] = zh;
            hipLaunchKernelGGL(k_tree_take_left, dim3(nblk(tnp)), dim3(256), 0, st, zh, full, tnp, log2t);
            hipLaunchKernelGGL(k_tree_pair_mul, dim3(nblk(tnp)), dim3(256), 0, st, F2, full, tnp, log2t);
            QT_TRY(ntt_run<true>(tabs, st, F2, tnp, log2t));
            hipLaunchKernelGGL(k_tree_fix, dim3(nblk(tnp >> log2t)), dim3(256), 0, st, F2, tnp >> log2t, log2t);
            Fr* tmp = F; F = F2; F2 = tmp;
        }
        return hipGetLastError();
    };
    if (n >= 2) QT_TRY(build_tree(qt.np_h, qt.lognp_h, n, qt.zhat_h));
    QT_TRY(build_tree(np, lg, 0, qt.zhat));  // last: F holds the top node of the QAP domain
    // ---- z = prod_{i=1..n} (x - i) ----
    if (n == np) {  // F now holds the single top node: lower np coefficients, monic
        QT_TRY(hipMemcpyAsync(qt.z, F, sizeof(Fr) * n, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_set_one, dim3(1), dim3(64), 0, st, qt.z, n);
    } else {  // z is the Newton basis polynomial N_n: convert the unit vector e_n
        QT_TRY(hipMemsetAsync(qt.pa, 0, sizeof(Fr) * np, st));
        hipLaunchKernelGGL(k_set_one, dim3(1), dim3(64), 0, st, qt.pa, n);
        QT_TRY(newton_to_monomial(tabs, st, np, lg, qt.zhat, 0, qt.pa, qt.pb, qt.newton_scale));
        QT_TRY(hipMemcpyAsync(qt.z, qt.pa, sizeof(Fr) * (n + 1), hipMemcpyDeviceToDevice, st));
    }
    // ---- g = rev(z)^-1 mod x^(n-1) by Newton iteration ----
    if (n >= 2) {
        const u64 m = n - 1;
        Fr *f = qt.data, *g = qt.pa, *t = qt.pb;  // f: np >= m ; g, t: big
        hipLaunchKernelGGL(k_rev_take, dim3(nblk(m)), dim3(256), 0, st, f, qt.z, n, m, m);  // f_i = z_{n-i}
        QT_TRY(hipMemsetAsync(g, 0, sizeof(Fr) * big, st));
        hipLaunchKernelGGL(k_set_one, dim3(1), dim3(64), 0, st, g, (u64)0);
        for (u64 cur = 1; cur < m;) {
            u64 nxt = 2 * cur < m ? 2 * cur : m;
            QT_TRY(poly_mul_dev(tabs, st, f, nxt, g, cur, t, nxt, qt.t1, qt.scratch));
            hipLaunchKernelGGL(k_two_minus, dim3(nblk(nxt)), dim3(256), 0, st, t, nxt);
            QT_TRY(poly_mul_dev(tabs, st, g, cur, t, nxt, g, nxt, qt.t1, qt.scratch));
            cur = nxt;
        }
        hipLaunchKernelGGL(k_fr_copy_pad, dim3(nblk(Sh)), dim3(256), 0, st, qt.ghat, g, m, Sh);
        QT_TRY(ntt_run<false>(tabs, st, qt.ghat, Sh, qt.ph));
    }
    QT_TRY(hipStreamSynchronize(st));
    return hipGetLastError();
}

}  // namespace ps
