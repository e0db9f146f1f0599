// A CRS array in monomial form, {x^i P}_{i < cnt}, into its Lagrange form {l_j(x) P}_{j < cnt} on the reference's
// interpolation nodes -- WITHOUT the secret point x.
//
// Why.  The reference's setups emit the monomial arrays only (Xi, Xi2, XiT: groth16.go:79-97; gsi: pinochio.go:101), and a
// prover that is handed those interpolates A and B and divides for h: 28 ms per proof at 2^20 constraints where the Lagrange
// form of the same key takes 18 ms.  The device setup emits both forms, but it needs the toxic waste,
// which the reference itself says "must be delete[d] after a trusted setup" (groth16.go:13-14).  This is the way from a
// reference-made key onto the fast route: a one-time linear map over group elements.
//
// How.  For every polynomial a of degree < cnt with values y on the nodes,  sum_i a_i (x^i P) = sum_j y_j (l_j(x) P), and
// a = M y with M the values -> monomial map of quotient.hpp (interpolate_on_nodes):
//     M = T_top ... T_7 . B . [pad] . C . D        D: y_j / j!     C: convolution with (-1)^k / k! (Newton coefficients)
//                                                  B: Newton -> monomial on blocks of 64      T_s: N = N_left + Z_left N_right
// so the Lagrange-form points are  L = M^T G = D^T C^T [take] B^T T_7^T ... T_top^T G: the SAME stages transposed (Tellegen),
// run backwards, with field multiplications replaced by scalar multiplications of points:
//     T_s^T   per node of size s:  left half unchanged, right_k = sum_j Z_left[j] g[k + j]: a cyclic correlation, i.e. an
//             NTT over points (k_ec_ntt_stage: butterflies (P + wQ, P - wQ)), times the stored transform of the reversed
//             Z_left, inverse NTT over points
//     B^T     e_k = <g, N_k>,  T_0 = g,  T_{k+1}[i] = T_k[i+1] - c_k T_k[i],  e_k = T_k[0]   (c_k the node: a small scalar)
//     C^T     the correlation with (-1)^k / k!, one NTT over points of size 2 np
//     D^T     L_j = (1 / j!) e_j
// ~230 cnt scalar multiplications of 255 bits (cnt log^2 cnt / 2 butterflies; each a GLV split over fixed signed windows,
// ec_mul_glv below): seconds at 2^16, under a minute at 2^20 -- once per key.  Every point stays in XYZZ form between stages (no inversions); one batch normalisation at the end gives the
// same canonical affine bytes ps_groth16_setup emits for l_j(x) P (tests/test_prover_gpu.py).
#pragma once
#include "msm.hpp"
#include "quotient.hpp"

namespace ps {

// buf[i] = i < cnt ? in[i] : O, i < total   (the identity is stored as (0, 0) in affine arrays)
template <class KF>
__global__ void __launch_bounds__(256, 1) k_ec_from_affine(const Affine<typename FieldTraits<KF>::Store>* __restrict__ in, u32 cnt, u32 total,
                                                           Xyzz<typename FieldTraits<KF>::Store>* __restrict__ buf) {
    const u32 i = logical_tid<KF>();
    if (i >= total) return;
    Xyzz<KF> r = xyzz_identity<KF>();
    if (i < cnt) {
        const Affine<KF> a = ld_affine<KF>(&in[i]);
        if (!affine_is_identity<KF>(a)) r = xyzz_from_affine<KF>(a.x, a.y);
    }
    st_xyzz<KF>(&buf[i], r);
}
// dst[i] = i < cnt ? src[i] : O, i < total
template <class KF>
__global__ void __launch_bounds__(256, 1) k_ec_pad(const Xyzz<typename FieldTraits<KF>::Store>* __restrict__ src, u32 cnt, u32 total,
                                                   Xyzz<typename FieldTraits<KF>::Store>* __restrict__ dst) {
    const u32 i = logical_tid<KF>();
    if (i >= total) return;
    st_xyzz<KF>(&dst[i], i < cnt ? ld_xyzz<KF>(&src[i]) : xyzz_identity<KF>());
}

// k * P for a 256-bit plain scalar, skipping the leading zero bits (xyzz_mul_scalar walks all 256)
template <class KF>
__device__ inline Xyzz<KF> ec_mul_words(const Xyzz<KF>& p, const u32* k) {
    int top = 255;
    while (top >= 0 && !((k[top >> 5] >> (top & 31)) & 1)) top--;
    Xyzz<KF> acc = xyzz_identity<KF>();
    if (xyzz_is_identity(p)) return acc;
#pragma unroll 1
    for (int bit = top; bit >= 0; bit--) {
        acc = xyzz_dbl<KF>(acc);
        if ((k[bit >> 5] >> (bit & 31)) & 1) xyzz_add<KF>(acc, p);
    }
    return acc;
}

// ---- full-width scalars: GLV split + fixed signed windows ----
// The lanes of a wave hold different scalars, so a double-and-add loop pays its addition on every bit (some lane always has
// the bit set): 255 (dbl + add).  Instead: k = k1 + k2 lambda with k2 = floor(k / lambda), lambda = z^2 - 1 (both halves
// below 2^128, since lambda^2 + lambda + 1 = r), lambda P = phi(P) = (beta x, y) on the subgroup of order r -- the points of
// a key are in it; for any other point the result is NOT k P -- and both halves in signed radix-16 digits taken from k' = k
// + 0x88..8 (digit_i = nibble_i(k') - 8: independent of each other, like the MSM's digits), over ONE table {1..8} P: 33
// rounds of 4 dbl + 2 add + one product by beta, the same schedule on every lane.  ~2 200 field products instead of ~5 900.
// (Constants: gen_constants.py, derived and checked in tools/glv_constants.py.)
PS_HD constexpr i32 fp_glv_beta28(int g2, int i) {
    constexpr i32 b1[FP_L] = PS_FP28_GLV_BETA_G1;
    constexpr i32 b2[FP_L] = PS_FP28_GLV_BETA_G2;
    return g2 ? b2[i] : b1[i];
}
template <class KF> struct GlvPhi;
template <> struct GlvPhi<Fp> {
    PS_INL static void apply(Xyzz<Fp>& q) {
        Fp b;
#pragma unroll
        for (int i = 0; i < FP_L; i++) b.l[i] = fp_glv_beta28(0, i);
        q.x = f_mul(q.x, b);
    }
};
template <> struct GlvPhi<Fp2s> {
    PS_INL static void apply(Xyzz<Fp2s>& q) {  // beta is in Fp: each lane of the pair scales its own component
        Fp b;
#pragma unroll
        for (int i = 0; i < FP_L; i++) b.l[i] = fp_glv_beta28(1, i);
        q.x = Fp2s{f_mul(q.x.v, b)};
    }
};
// k (8 words, < r) -> q = floor(k / lambda), rem = k mod lambda (4 words each) by binary long division: ~4 000 integer
// instructions, 0.5 % of the multiplication they steer
__device__ inline void glv_split(const u32* k, u32* rem, u32* q) {
    constexpr u32 lam[4] = PS_GLV_LAMBDA;
    u32 r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0;
    q[0] = q[1] = q[2] = q[3] = 0;
    u32 qov = 0;  // (bits of the quotient above 2^128: none for k < r; kept so that a wrong input cannot pass unnoticed)
#pragma unroll 1
    for (int bit = 255; bit >= 0; bit--) {
        r4 = (r4 << 1) | (r3 >> 31);
        r3 = (r3 << 1) | (r2 >> 31);
        r2 = (r2 << 1) | (r1 >> 31);
        r1 = (r1 << 1) | (r0 >> 31);
        r0 = (r0 << 1) | ((k[bit >> 5] >> (bit & 31)) & 1u);
        // rem >= lambda ?
        const u64 d0 = (u64)r0 - lam[0], d1 = (u64)r1 - lam[1] - ((d0 >> 32) & 1), d2 = (u64)r2 - lam[2] - ((d1 >> 32) & 1),
                  d3 = (u64)r3 - lam[3] - ((d2 >> 32) & 1), d4 = (u64)r4 - ((d3 >> 32) & 1);
        const bool ge = !((d4 >> 32) & 1);
        if (ge) { r0 = (u32)d0; r1 = (u32)d1; r2 = (u32)d2; r3 = (u32)d3; r4 = (u32)d4; }
        if (bit >= 128) qov |= ge ? 1u : 0u;
        else if (ge) q[bit >> 5] |= 1u << (bit & 31);
    }
    rem[0] = r0; rem[1] = r1; rem[2] = r2; rem[3] = r3;
    if (qov) { q[0] = q[1] = q[2] = q[3] = 0xffffffffu; }  // unreachable for k < r
}
// w[0..4] = v[0..3] + 0x8888...8 (32 nibbles): digit i of v in [-8, 7] is nibble i of w minus 8, digit 32 = w[4] (0 or 1)
__device__ inline void glv_bias(const u32* v, u32* w) {
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        c += (u64)v[i] + 0x88888888ull;
        w[i] = (u32)c;
        c >>= 32;
    }
    w[4] = (u32)c;
}
// (forced inline: the out-of-line xyzz_add / xyzz_dbl pass their 224-byte operands through scratch memory -- 2^18 points: 2.41 ->
// 1.86 s G1, 6.70 -> 5.71 s G2)
#if defined(PS_GLV_OUTLINE)
#define PS_GLV_ADD(a, t) xyzz_add<KF>(a, t)
#define PS_GLV_DBL(a) xyzz_dbl<KF>(a)
#else
#define PS_GLV_ADD(a, t) xyzz_add_inl<KF>(a, t)
#define PS_GLV_DBL(a) xyzz_dbl_inl<KF>(a)
#endif
template <class KF>
__device__ inline void glv_add_digit(Xyzz<KF>& acc, const Xyzz<KF>* T, int d, bool phi) {
    if (d == 0) return;
    Xyzz<KF> t = T[(d < 0 ? -d : d) - 1];
    if (d < 0) t.y = f_neg(t.y);
    if (phi) GlvPhi<KF>::apply(t);
    PS_GLV_ADD(acc, t);
}
template <class KF>
__device__ inline Xyzz<KF> ec_mul_glv(const Xyzz<KF>& p, const u32* k) {
    if (xyzz_is_identity(p)) return xyzz_identity<KF>();
    u32 k1[4], k2[4], w1[5], w2[5];
    glv_split(k, k1, k2);
    glv_bias(k1, w1);
    glv_bias(k2, w2);
    Xyzz<KF> T[8];  // j P, j = 1..8 (private memory: 8 x 224 bytes per lane)
    T[0] = p;
    T[1] = xyzz_dbl<KF>(p);
    T[2] = T[1]; xyzz_add<KF>(T[2], p);
    T[3] = xyzz_dbl<KF>(T[1]);
    T[4] = T[3]; xyzz_add<KF>(T[4], p);
    T[5] = xyzz_dbl<KF>(T[2]);
    T[6] = T[5]; xyzz_add<KF>(T[6], p);
    T[7] = xyzz_dbl<KF>(T[3]);
    Xyzz<KF> acc = xyzz_identity<KF>();
    glv_add_digit<KF>(acc, T, (int)w1[4], false);  // the top digits: 0 or 1
    glv_add_digit<KF>(acc, T, (int)w2[4], true);
#pragma unroll 1
    for (int i = 31; i >= 0; i--) {
#pragma unroll 1
        for (int j = 0; j < 4; j++) acc = PS_GLV_DBL(acc);
        glv_add_digit<KF>(acc, T, (int)((w1[i >> 3] >> ((i & 7) * 4)) & 15u) - 8, false);
        glv_add_digit<KF>(acc, T, (int)((w2[i >> 3] >> ((i & 7) * 4)) & 15u) - 8, true);
    }
    return acc;
}
template <class KF>
__device__ inline Xyzz<KF> ec_mul_fr(const Xyzz<KF>& p, const Fr& s_mont) {
    u32 k[8];
    fr_to_words8(k, fr_from_mont(s_mont));
#if defined(PS_EC_MUL_PLAIN)  // the double-and-add loop this replaced (A/B and cross-check builds)
    return ec_mul_words<KF>(p, k);
#else
    return ec_mul_glv<KF>(p, k);
#endif
}

// Waves per SIMD of the two kernels that multiply points by full-width scalars (a lone wave issues a dependent multiply-add
// every other slot).
#ifndef PS_EC_WAVES_G1
#define PS_EC_WAVES_G1 2
#endif
#ifndef PS_EC_WAVES_G2
#define PS_EC_WAVES_G2 1
#endif
#define PS_EC_WAVES(KF) (FieldTraits<KF>::LANES == 1 ? PS_EC_WAVES_G1 : PS_EC_WAVES_G2)

// One stage of the NTT over points, in place: `total` points = a batch of transforms of 2^p, half-distance 2^logh.
// Indexing and twiddles are those of the scalar transform (ntt.hpp, k_ntt_pass): forward = Cooley-Tukey (P + wQ, P - wQ)
// on natural input from the largest distance down, one twiddle w = omega^bitrev(block) per block; inverse = Gentleman-Sande
// (P + Q, (P - Q) w^-1) from the smallest distance up.  The inverse's factor 2^-p rides in the stored scalar transform.
template <class KF, bool INV>
__global__ void __launch_bounds__(256, PS_EC_WAVES(KF)) k_ec_ntt_stage(Xyzz<typename FieldTraits<KF>::Store>* __restrict__ buf, u32 total, int p, int logh,
                                                         const Fr* __restrict__ tw, int log_tab) {
    const u32 bf = logical_tid<KF>();
    if (bf >= total / 2) return;
    const u32 pos = bf & ((1u << logh) - 1u), blk_all = bf >> logh;
    const u32 i0 = (blk_all << (logh + 1)) | pos, i1 = i0 + (1u << logh);
    const int M = p - 1 - logh;  // 2^M blocks per transform at this stage
    Xyzz<KF> P = ld_xyzz<KF>(&buf[i0]), Q = ld_xyzz<KF>(&buf[i1]);
    Fr w = fr_one();
    const bool has_w = M > 0;
    if (has_w) {
        const u32 blk = (i0 & ((1u << p) - 1u)) >> (logh + 1);
        w = tw[(size_t)(__brev(blk) >> (32 - M)) << (log_tab - 1 - M)];
    }
    if (!INV) {
        if (has_w) Q = ec_mul_fr<KF>(Q, w);
        Xyzz<KF> s = P;
        xyzz_add<KF>(s, Q);
        xyzz_add<KF>(P, xyzz_neg<KF>(Q));
        st_xyzz<KF>(&buf[i0], s);
        st_xyzz<KF>(&buf[i1], P);
    } else {
        Xyzz<KF> s = P;
        xyzz_add<KF>(s, Q);
        xyzz_add<KF>(P, xyzz_neg<KF>(Q));
        if (has_w) P = ec_mul_fr<KF>(P, w);
        st_xyzz<KF>(&buf[i0], s);
        st_xyzz<KF>(&buf[i1], P);
    }
}

// buf[i] = scal[i & mask] * buf[i], i < total   (scalars in Montgomery form)
template <class KF>
__global__ void __launch_bounds__(256, PS_EC_WAVES(KF)) k_ec_scale(Xyzz<typename FieldTraits<KF>::Store>* __restrict__ buf, u32 total,
                                                     const Fr* __restrict__ scal, u64 mask) {
    const u32 i = logical_tid<KF>();
    if (i >= total) return;
    st_xyzz<KF>(&buf[i], ec_mul_fr<KF>(ld_xyzz<KF>(&buf[i]), scal[(u64)i & mask]));
}

// T_s^T, last step: the right half of every node of size 2^logs takes the first half of the node's correlation
template <class KF>
__global__ void __launch_bounds__(256, 1) k_ec_take_right(Xyzz<typename FieldTraits<KF>::Store>* __restrict__ g,
                                                          const Xyzz<typename FieldTraits<KF>::Store>* __restrict__ corr, u32 total, int logs) {
    const u32 idx = logical_tid<KF>();  // over total / 2
    if (idx >= total / 2) return;
    const u32 half = 1u << (logs - 1);
    const u32 node = idx >> (logs - 1), k = idx & (half - 1);
    st_xyzz<KF>(&g[((size_t)node << logs) + half + k], ld_xyzz<KF>(&corr[((size_t)node << logs) + k]));
}

// B^T, step k of 64 on every block of 64 (see the header): E[64 b + k] = T[64 b]; T'[i] = T[i + 1] - c_k T[i] for i < 63 - k,
// c_k = off + 64 (b mod member64) + k + 1 the node (a plain integer below 2^32).  Ping-pong between tin and tout.
template <class KF>
__global__ void __launch_bounds__(256, 1) k_ec_base_step(const Xyzz<typename FieldTraits<KF>::Store>* __restrict__ tin,
                                                         Xyzz<typename FieldTraits<KF>::Store>* __restrict__ tout,
                                                         Xyzz<typename FieldTraits<KF>::Store>* __restrict__ E, u32 total, int k, u64 off, u32 member64) {
    const u32 idx = logical_tid<KF>();
    if (idx >= total) return;
    const u32 b = idx >> 6, i = idx & 63u;
    if ((int)i > 63 - k) return;
    const Xyzz<KF> t = ld_xyzz<KF>(&tin[idx]);
    if (i == 0) st_xyzz<KF>(&E[((size_t)b << 6) + k], t);
    if ((int)i >= 63 - k) return;
    const u32 ck = (u32)(off + 64ull * (b % member64) + (u64)k + 1ull);
    u32 words[8] = {ck, 0, 0, 0, 0, 0, 0, 0};
    Xyzz<KF> r = ld_xyzz<KF>(&tin[idx + 1]);
    xyzz_add<KF>(r, xyzz_neg<KF>(ec_mul_words<KF>(t, words)));
    st_xyzz<KF>(&tout[idx], r);
}

// ---- scalar side: the stored transforms of the reversed multipliers ----
// dst[node s + i] = src[node s + ((s - i) mod s)], s = 2^logs   (the sequence read backwards, cyclically)
__global__ void __launch_bounds__(256) k_fr_rev_mod(Fr* __restrict__ dst, const Fr* __restrict__ src, u64 total, int logs) {
    const u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const u64 s = 1ull << logs, i = idx & (s - 1);
    dst[idx] = src[(idx - i) + ((s - i) & (s - 1))];
}
// dst[i] = v[(-i) mod 2np], v_j = (-1)^j / j! for j < np and 0 beyond; i < 2np
__global__ void __launch_bounds__(256) k_fr_v_rev(Fr* __restrict__ dst, const Fr* __restrict__ invfact, u64 np) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * np) return;
    const u64 j = (2 * np - i) & (2 * np - 1);
    dst[i] = j < np ? ((j & 1) ? fr_neg(invfact[j]) : invfact[j]) : fr_zero();
}

}  // namespace ps
