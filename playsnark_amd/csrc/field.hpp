// BLS12-381 prime-field arithmetic for gfx950 (and for the host-side glue of the same library).
//
// Two representations, chosen from measured gfx950 instruction rates (profiles/r01_microbench_valu.txt):
// v_mad_u64_u32 / v_mad_i64_i32 run at half rate (~28 T lane-ops/s chip-wide) while every carry
// instruction (v_addc_co_u32, v_lshl_add_u64) costs almost as much as a multiply and needs
// software wait states behind an SGPR carry.
//   Fp (381 bit, the MSM's coordinate field): UNSATURATED -- 14 signed limbs of 28 bits,
//      Montgomery R = 2^392.  Products accumulate in 64-bit columns with no carry handling
//      (14 * 2^58 fits), add/sub are 14 independent v_add_u32, reduction is lazy.  One
//      multiplication is 392 mads + ~110 cheap ops, fully inlined (no call ABI, no scratch).
//   Fr (255 bit, the NTT / quotient field): the same unsaturated scheme with 10 limbs, R = 2^280
//      (200 mads per product).  Scalars at the MSM boundary stay plain 8 x 32-bit words (FrSat).
// All loops are fully unrolled with compile-time indices so elements live in VGPRs.
//
// This is product code: it replaces, for the hot path, the arithmetic the reference gets
// from kilic/bls12-381 through kyber (call sites algebra.go:100-101,111-112,356;
// groth16.go:138,149-152,189-200).  It shares no code with oracle/.
#pragma once
#include <stdint.h>

#include "bls12_381_constants.h"

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define PS_HD __host__ __device__
#else
#define PS_HD
#endif
#define PS_INL PS_HD inline __attribute__((always_inline))

namespace ps {

typedef uint32_t u32;
typedef uint64_t u64;

struct FpParams {
    static constexpr int N = 12;
    static constexpr u32 INV = PS_FP_INV32;
    PS_HD static constexpr u32 mod(int i) { constexpr u32 v[N] = PS_FP_MOD; return v[i]; }
    PS_HD static constexpr u32 r1(int i) { constexpr u32 v[N] = PS_FP_R1; return v[i]; }
    PS_HD static constexpr u32 r2(int i) { constexpr u32 v[N] = PS_FP_R2; return v[i]; }
};
struct FrParams {
    static constexpr int N = 8;
    static constexpr u32 INV = PS_FR_INV32;
    PS_HD static constexpr u32 mod(int i) { constexpr u32 v[N] = PS_FR_MOD; return v[i]; }
    PS_HD static constexpr u32 r1(int i) { constexpr u32 v[N] = PS_FR_R1; return v[i]; }
    PS_HD static constexpr u32 r2(int i) { constexpr u32 v[N] = PS_FR_R2; return v[i]; }
};

template <class P>
struct Fe {
    static constexpr int N = P::N;
    u32 l[P::N];
};
typedef Fe<FrParams> FrSat;  // plain / saturated 8 x 32-bit scalars (MSM digit extraction, I/O)

// ---------------------------------------------------------------------------------------
// basic helpers
// ---------------------------------------------------------------------------------------
template <class P>
PS_INL Fe<P> fe_zero() {
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::N; i++) r.l[i] = 0;
    return r;
}
template <class P>
PS_INL Fe<P> fe_one() {  // Montgomery one
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::N; i++) r.l[i] = P::r1(i);
    return r;
}
template <class P>
PS_INL bool fe_is_zero(const Fe<P>& a) {
    u32 x = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) x |= a.l[i];
    return x == 0;
}
template <class P>
PS_INL bool fe_eq(const Fe<P>& a, const Fe<P>& b) {
    u32 x = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) x |= a.l[i] ^ b.l[i];
    return x == 0;
}

// r = a - mod if a >= mod (a < 2*mod, optional extra top word `hi`)
template <class P>
PS_INL void fe_reduce_once(u32* t, u32 hi) {
    u32 s[P::N];
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        u64 d = (u64)t[i] - P::mod(i) - borrow;
        s[i] = (u32)d;
        borrow = (u32)(d >> 63);
    }
    bool ge = (hi != 0) | (borrow == 0);
#pragma unroll
    for (int i = 0; i < P::N; i++) t[i] = ge ? s[i] : t[i];
}

template <class P>
PS_INL Fe<P> fe_add(const Fe<P>& a, const Fe<P>& b) {
    Fe<P> r;
    u32 carry = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        u64 s = (u64)a.l[i] + b.l[i] + carry;
        r.l[i] = (u32)s;
        carry = (u32)(s >> 32);
    }
    fe_reduce_once<P>(r.l, carry);
    return r;
}
template <class P>
PS_INL Fe<P> fe_sub(const Fe<P>& a, const Fe<P>& b) {
    Fe<P> r;
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        u64 d = (u64)a.l[i] - b.l[i] - borrow;
        r.l[i] = (u32)d;
        borrow = (u32)(d >> 63);
    }
    // add back the modulus when the subtraction borrowed
    u32 mask = 0u - borrow;
    u32 carry = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        u64 s = (u64)r.l[i] + (P::mod(i) & mask) + carry;
        r.l[i] = (u32)s;
        carry = (u32)(s >> 32);
    }
    return r;
}
template <class P>
PS_INL Fe<P> fe_neg(const Fe<P>& a) {
    Fe<P> r;
    u32 borrow = 0;
    u32 nz = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) nz |= a.l[i];
    u32 mask = nz ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        u64 d = (u64)(P::mod(i) & mask) - a.l[i] - borrow;
        r.l[i] = (u32)d;
        borrow = (u32)(d >> 63);
    }
    return r;
}
template <class P>
PS_INL Fe<P> fe_dbl(const Fe<P>& a) { return fe_add<P>(a, a); }

// ---------------------------------------------------------------------------------------
// Montgomery multiplication, CIOS over 32-bit limbs: N*N v_mad_u64_u32 for the product and
// N*N for the interleaved reduction.  a*b + t + c never overflows 64 bits
// ((2^32-1)^2 + 2(2^32-1) = 2^64-1), so no carry word beyond the running 64-bit value.
// ---------------------------------------------------------------------------------------
template <class P>
PS_INL void mont_mul_raw(u32* r, const u32* a, const u32* b) {
    constexpr int N = P::N;
    u32 t[N + 2];
#pragma unroll
    for (int i = 0; i < N + 2; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        u64 c = 0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            u64 x = (u64)a[j] * b[i] + t[j] + c;
            t[j] = (u32)x;
            c = x >> 32;
        }
        u64 x = (u64)t[N] + c;
        t[N] = (u32)x;
        t[N + 1] = (u32)(x >> 32);
        u32 m = t[0] * P::INV;
        c = ((u64)m * P::mod(0) + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < N; j++) {
            u64 y = (u64)m * P::mod(j) + t[j] + c;
            t[j - 1] = (u32)y;
            c = y >> 32;
        }
        x = (u64)t[N] + c;
        t[N - 1] = (u32)x;
        t[N] = t[N + 1] + (u32)(x >> 32);
    }
    fe_reduce_once<P>(t, t[N]);
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = t[i];
}

// The multiply is kept out of line on the device: one copy (~6 KB of ISA) stays resident in
// the instruction cache instead of ten inlined copies per point addition.  Fp (48 B) is
// passed and returned in VGPRs by the AMDGPU calling convention.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PS_MUL_INLINE)
#define PS_MUL_ATTR __device__ __noinline__
#else
#define PS_MUL_ATTR PS_INL
#endif

template <class P>
PS_MUL_ATTR Fe<P> fe_mul(Fe<P> a, Fe<P> b) {
    Fe<P> r;
    mont_mul_raw<P>(r.l, a.l, b.l);
    return r;
}
template <class P>
PS_MUL_ATTR Fe<P> fe_sqr(Fe<P> a) {
    Fe<P> r;
    mont_mul_raw<P>(r.l, a.l, a.l);
    return r;
}

// to / from Montgomery form
template <class P>
PS_INL Fe<P> fe_to_mont(const Fe<P>& a) {
    Fe<P> r2;
#pragma unroll
    for (int i = 0; i < P::N; i++) r2.l[i] = P::r2(i);
    return fe_mul<P>(a, r2);
}
template <class P>
PS_INL Fe<P> fe_from_mont(const Fe<P>& a) {
    Fe<P> one = fe_zero<P>();
    one.l[0] = 1;
    return fe_mul<P>(a, one);
}

// a^e for a plain little-endian exponent of NE 32-bit words (square-and-multiply, LSB first)
template <class P, int NE>
PS_HD inline Fe<P> fe_pow(const Fe<P>& a, const u32* e) {
    Fe<P> acc = fe_one<P>();
    Fe<P> base = a;
    for (int i = 0; i < 32 * NE; i++) {
        if ((e[i >> 5] >> (i & 31)) & 1) acc = fe_mul<P>(acc, base);
        base = fe_sqr<P>(base);
    }
    return acc;
}
// Fermat inversion a^(mod-2); 0 -> 0
template <class P>
PS_HD inline Fe<P> fe_inv(const Fe<P>& a) {
    u32 e[P::N];
    u32 borrow = 0;
    for (int i = 0; i < P::N; i++) {
        u64 d = (u64)P::mod(i) - (i == 0 ? 2u : 0u) - borrow;
        e[i] = (u32)d;
        borrow = (u32)(d >> 63);
    }
    return fe_pow<P, P::N>(a, e);
}

// plain (non-Montgomery) comparison a > b on little-endian limb arrays
template <int N>
PS_INL bool limbs_gt(const u32* a, const u32* b) {
    bool gt = false, decided = false;
#pragma unroll
    for (int i = N - 1; i >= 0; i--) {
        if (!decided && a[i] != b[i]) { gt = a[i] > b[i]; decided = true; }
    }
    return gt;
}
template <class P>
PS_INL bool fe_is_canonical(const u32* a) {  // a < mod ?
    u32 m[P::N];
#pragma unroll
    for (int i = 0; i < P::N; i++) m[i] = P::mod(i);
    return limbs_gt<P::N>(m, a);
}

// big-endian bytes <-> plain limbs
template <int N>
PS_HD inline void limbs_from_be(u32* l, const uint8_t* b) {
    for (int i = 0; i < N; i++) {
        const uint8_t* p = b + 4 * (N - 1 - i);
        l[i] = ((u32)p[0] << 24) | ((u32)p[1] << 16) | ((u32)p[2] << 8) | (u32)p[3];
    }
}
template <int N>
PS_HD inline void limbs_to_be(uint8_t* b, const u32* l) {
    for (int i = 0; i < N; i++) {
        uint8_t* p = b + 4 * (N - 1 - i);
        p[0] = (uint8_t)(l[i] >> 24); p[1] = (uint8_t)(l[i] >> 16); p[2] = (uint8_t)(l[i] >> 8); p[3] = (uint8_t)l[i];
    }
}


// =======================================================================================
// Fp: unsaturated 14 x 28-bit signed limbs, Montgomery R = 2^392
// =======================================================================================
// Value V = sum l[i] * 2^(28 i).  Limbs are int32 and may be negative or exceed 28 bits
// ("lazy"): add/sub/neg never carry.  Magnitude class c means |l[i]| < c * 2^28.
//   * f_mul(a, b) needs class(a) * class(b) <= 8 and |A|, |B| <= 16 p; its result has limbs
//     0..12 in [0, 2^28), a small signed top limb, and value in (-p/8, 9p/8)   ("class 1").
//   * f_norm() is one parallel carry-save step that brings any class <= 8 value back to class ~1
//     without changing V.
//   * f_is_zero / f_eq work modulo p for |V| < 3p (every call site passes differences of
//     class-1 values or multiplication results).
typedef int32_t i32;
typedef int64_t i64;

struct Fp {
    static constexpr int L = 14;
    i32 l[14];
};
constexpr int FP_L = 14;
constexpr u32 FP_MASK = (1u << 28) - 1u;
constexpr u32 FP_INV28 = PS_FP28_INV;

PS_HD constexpr i32 fp_mod28(int i) { constexpr i32 v[FP_L] = PS_FP28_MOD; return v[i]; }
PS_HD constexpr i32 fp_r1_28(int i) { constexpr i32 v[FP_L] = PS_FP28_R1; return v[i]; }
PS_HD constexpr i32 fp_r2_28(int i) { constexpr i32 v[FP_L] = PS_FP28_R2; return v[i]; }

PS_INL Fp fp_zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < FP_L; i++) r.l[i] = 0;
    return r;
}
PS_INL Fp fp_one() {  // Montgomery one
    Fp r;
#pragma unroll
    for (int i = 0; i < FP_L; i++) r.l[i] = fp_r1_28(i);
    return r;
}
PS_INL Fp f_add(const Fp& a, const Fp& b) {
    Fp r;
#pragma unroll
    for (int i = 0; i < FP_L; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
PS_INL Fp f_sub(const Fp& a, const Fp& b) {
    Fp r;
#pragma unroll
    for (int i = 0; i < FP_L; i++) r.l[i] = a.l[i] - b.l[i];
    return r;
}
PS_INL Fp f_neg(const Fp& a) {
    Fp r;
#pragma unroll
    for (int i = 0; i < FP_L; i++) r.l[i] = -a.l[i];
    return r;
}
// one parallel carry-save step: limbs 0..12 into [0, 2^28) + a small carry, value unchanged
PS_INL Fp f_norm(const Fp& a) {
    Fp r;
    r.l[0] = (i32)((u32)a.l[0] & FP_MASK);
#pragma unroll
    for (int i = 1; i < FP_L - 1; i++) r.l[i] = (i32)((u32)a.l[i] & FP_MASK) + (a.l[i - 1] >> 28);
    r.l[FP_L - 1] = a.l[FP_L - 1] + (a.l[FP_L - 2] >> 28);
    return r;
}
// full sequential carry: limbs 0..12 in [0, 2^28), signed top limb; unique for a given V
PS_INL Fp fp_propagate(const Fp& a) {
    Fp r;
    i32 c = 0;
#pragma unroll
    for (int i = 0; i < FP_L - 1; i++) {
        i32 t = a.l[i] + c;
        r.l[i] = (i32)((u32)t & FP_MASK);
        c = t >> 28;
    }
    r.l[FP_L - 1] = a.l[FP_L - 1] + c;
    return r;
}

#if defined(__HIP_DEVICE_COMPILE__)
// The same products as one chain of multiply-adds per column (generated: tools/gen_fp_chain.py; the why is at fr_mul_chain):
// the compiler's code sums each column as independent chains and pays a 64-bit add per merge -- 260 quarter-rate adds per
// mixed addition of the bucket accumulation, 5.7 % of its issue time.
#include "fp_chain.inc"
#endif
// Montgomery product, product scanning: column k sums a[i]*b[k-i] and m[i]*p[k-i] in one signed
// 64-bit accumulator (v_mad_i64_i32 chains), m[k] = -acc/p mod 2^28 clears the low 28 bits.
PS_INL Fp f_mul(const Fp& a, const Fp& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PS_FP_MUL_NO_CHAIN)
    return f_mul_chain(a, b);
#endif
    Fp r;
    i32 m[FP_L];
    i64 acc = 0;
#pragma unroll
    for (int k = 0; k < FP_L; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (i64)a.l[i] * (i64)b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (i64)m[i] * (i64)fp_mod28(k - i);
        m[k] = (i32)(((u32)acc * FP_INV28) & FP_MASK);
        acc += (i64)m[k] * (i64)fp_mod28(0);
        acc >>= 28;
    }
#pragma unroll
    for (int k = FP_L; k < 2 * FP_L - 1; k++) {
#pragma unroll
        for (int i = k - FP_L + 1; i < FP_L; i++) acc += (i64)a.l[i] * (i64)b.l[k - i];
#pragma unroll
        for (int i = k - FP_L + 1; i < FP_L; i++) acc += (i64)m[i] * (i64)fp_mod28(k - i);
        r.l[k - FP_L] = (i32)((u32)acc & FP_MASK);
        acc >>= 28;
    }
    r.l[FP_L - 1] = (i32)acc;
    return r;
}
PS_INL Fp f_sqr(const Fp& a) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PS_FP_MUL_NO_CHAIN)
    return f_sqr_chain(a);
#endif
    return f_mul(a, a);  // the compiler folds the symmetric products
}

// a*b - c*d with ONE Montgomery reduction (588 mads instead of 784): both products accumulate in
// the same columns.  Needs class(a)*class(b) + class(c)*class(d) <= 8.
PS_INL Fp f_mul2sub(const Fp& a, const Fp& b, const Fp& c, const Fp& d) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PS_FP_MUL_NO_CHAIN)
    return f_mul2sub_chain(a, b, c, d);
#endif
    Fp r;
    i32 m[FP_L];
    i64 acc = 0;
#pragma unroll
    for (int k = 0; k < FP_L; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) {
            acc += (i64)a.l[i] * (i64)b.l[k - i];
            acc -= (i64)c.l[i] * (i64)d.l[k - i];
        }
#pragma unroll
        for (int i = 0; i < k; i++) acc += (i64)m[i] * (i64)fp_mod28(k - i);
        m[k] = (i32)(((u32)acc * FP_INV28) & FP_MASK);
        acc += (i64)m[k] * (i64)fp_mod28(0);
        acc >>= 28;
    }
#pragma unroll
    for (int k = FP_L; k < 2 * FP_L - 1; k++) {
#pragma unroll
        for (int i = k - FP_L + 1; i < FP_L; i++) {
            acc += (i64)a.l[i] * (i64)b.l[k - i];
            acc -= (i64)c.l[i] * (i64)d.l[k - i];
        }
#pragma unroll
        for (int i = k - FP_L + 1; i < FP_L; i++) acc += (i64)m[i] * (i64)fp_mod28(k - i);
        r.l[k - FP_L] = (i32)((u32)acc & FP_MASK);
        acc >>= 28;
    }
    r.l[FP_L - 1] = (i32)acc;
    return r;
}

// a*b + c*d with one reduction (the complex product of the lane-split Fp2, below)
PS_INL Fp f_mul2add(const Fp& a, const Fp& b, const Fp& c, const Fp& d) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PS_FP_MUL_NO_CHAIN)
    return f_mul2add_chain(a, b, c, d);
#endif
    Fp r;
    i32 m[FP_L];
    i64 acc = 0;
#pragma unroll
    for (int k = 0; k < FP_L; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) {
            acc += (i64)a.l[i] * (i64)b.l[k - i];
            acc += (i64)c.l[i] * (i64)d.l[k - i];
        }
#pragma unroll
        for (int i = 0; i < k; i++) acc += (i64)m[i] * (i64)fp_mod28(k - i);
        m[k] = (i32)(((u32)acc * FP_INV28) & FP_MASK);
        acc += (i64)m[k] * (i64)fp_mod28(0);
        acc >>= 28;
    }
#pragma unroll
    for (int k = FP_L; k < 2 * FP_L - 1; k++) {
#pragma unroll
        for (int i = k - FP_L + 1; i < FP_L; i++) {
            acc += (i64)a.l[i] * (i64)b.l[k - i];
            acc += (i64)c.l[i] * (i64)d.l[k - i];
        }
#pragma unroll
        for (int i = k - FP_L + 1; i < FP_L; i++) acc += (i64)m[i] * (i64)fp_mod28(k - i);
        r.l[k - FP_L] = (i32)((u32)acc & FP_MASK);
        acc >>= 28;
    }
    r.l[FP_L - 1] = (i32)acc;
    return r;
}
// a1*b1 + a2*b2 - s1*t1 - s2*t2 under ONE Montgomery reduction: 980 multiply-adds instead of 2 x 588 (the lane-split
// Fp2 form of a*b - c*d, below).  The 64-bit columns hold 14 terms of each product plus the reduction terms, so the
// operands' limb classes must satisfy class(a1)class(b1) + .. + class(s2)class(t2) <= 8.
PS_INL Fp f_mul2add2sub(const Fp& a1, const Fp& b1, const Fp& a2, const Fp& b2, const Fp& s1, const Fp& t1, const Fp& s2,
                        const Fp& t2) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PS_FP_MUL_NO_CHAIN)
    return f_mul2add2sub_chain(a1, b1, a2, b2, s1, t1, s2, t2);
#endif
    Fp r;
    i32 m[FP_L];
    i64 acc = 0;
#pragma unroll
    for (int k = 0; k < FP_L; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) {
            acc += (i64)a1.l[i] * (i64)b1.l[k - i];
            acc += (i64)a2.l[i] * (i64)b2.l[k - i];
            acc -= (i64)s1.l[i] * (i64)t1.l[k - i];
            acc -= (i64)s2.l[i] * (i64)t2.l[k - i];
        }
#pragma unroll
        for (int i = 0; i < k; i++) acc += (i64)m[i] * (i64)fp_mod28(k - i);
        m[k] = (i32)(((u32)acc * FP_INV28) & FP_MASK);
        acc += (i64)m[k] * (i64)fp_mod28(0);
        acc >>= 28;
    }
#pragma unroll
    for (int k = FP_L; k < 2 * FP_L - 1; k++) {
#pragma unroll
        for (int i = k - FP_L + 1; i < FP_L; i++) {
            acc += (i64)a1.l[i] * (i64)b1.l[k - i];
            acc += (i64)a2.l[i] * (i64)b2.l[k - i];
            acc -= (i64)s1.l[i] * (i64)t1.l[k - i];
            acc -= (i64)s2.l[i] * (i64)t2.l[k - i];
        }
#pragma unroll
        for (int i = k - FP_L + 1; i < FP_L; i++) acc += (i64)m[i] * (i64)fp_mod28(k - i);
        r.l[k - FP_L] = (i32)((u32)acc & FP_MASK);
        acc >>= 28;
    }
    r.l[FP_L - 1] = (i32)acc;
    return r;
}

// sum_k (+/-) a_k b_k under ONE Montgomery reduction -- the same column sums as f_mul / f_mul2sub / f_mul2add /
// f_mul2add2sub, hence the same limbs bit for bit, but written for a wave that is ALONE on its SIMD (the lane-cooperative
// additions of the short sums' tail, qtail.hpp): every column keeps its own 64-bit accumulator, so the 196 multiply-adds
// of a product are independent of each other and the only serial chain is the 14 Montgomery digits (about five dependent
// instructions each).  The product-scanning forms above run all 392 multiply-adds of a product through one accumulator:
// fine when two waves share a SIMD or a thread has several products in flight, 2 x slower when neither holds (measured:
// a quad addition 8-10 us instead of ~5).  Contract as for the fused forms: sum_k class(a_k) class(b_k) <= 8.
template <int NPROD>
PS_INL Fp f_mulsum_ilp(const Fp* a, const Fp* b, const bool* neg) {
    i64 col[2 * FP_L];
#pragma unroll
    for (int k = 0; k < 2 * FP_L; k++) col[k] = 0;
#pragma unroll
    for (int q = 0; q < NPROD; q++) {
#pragma unroll
        for (int i = 0; i < FP_L; i++) {
#pragma unroll
            for (int j = 0; j < FP_L; j++) {
                if (neg[q]) col[i + j] -= (i64)a[q].l[i] * (i64)b[q].l[j];
                else col[i + j] += (i64)a[q].l[i] * (i64)b[q].l[j];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < FP_L; k++) {
        const i32 m = (i32)(((u32)col[k] * FP_INV28) & FP_MASK);
#pragma unroll
        for (int j = 0; j < FP_L; j++) col[k + j] += (i64)m * (i64)fp_mod28(j);
        col[k + 1] += col[k] >> 28;
    }
    Fp r;
#pragma unroll
    for (int k = FP_L; k < 2 * FP_L - 1; k++) {
        r.l[k - FP_L] = (i32)((u32)col[k] & FP_MASK);
        col[k + 1] += col[k] >> 28;
    }
    r.l[FP_L - 1] = (i32)col[2 * FP_L - 1];
    return r;
}
PS_INL Fp f_mul_ilp(const Fp& a, const Fp& b) {
    const bool neg[1] = {false};
    return f_mulsum_ilp<1>(&a, &b, neg);
}
PS_INL Fp f_mul2sub_ilp(const Fp& a, const Fp& b, const Fp& c, const Fp& d) {  // a b - c d
    const Fp x[2] = {a, c}, y[2] = {b, d};
    const bool neg[2] = {false, true};
    return f_mulsum_ilp<2>(x, y, neg);
}

// Out-of-line copy for code paths where ten inlined multiplications per group operation would
// not fit the instruction cache (the Fp2 tower of G2, cold exceptional cases).
#if defined(PS_FP2_INLINE)
PS_INL Fp fp_mul_call(const Fp& a, const Fp& b) { return f_mul(a, b); }
#elif defined(__HIP_DEVICE_COMPILE__)
__host__ __device__ __attribute__((noinline)) Fp fp_mul_call(Fp a, Fp b) { return f_mul(a, b); }
#else
PS_HD inline Fp fp_mul_call(Fp a, Fp b) { return f_mul(a, b); }
#endif
// ... and of the squaring (three quarters of the multiplier work): the exponentiation loops
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PS_FP2_INLINE)
__host__ __device__ __attribute__((noinline)) Fp fp_sqr_call(Fp a) { return f_sqr(a); }
#else
PS_HD inline Fp fp_sqr_call(Fp a) { return f_sqr(a); }
#endif

// V == 0 (mod p)?  Valid for any lazy value with class <= 8 and |V| <= 16p.
// Filter: V = k*p with |k| <= 16 forces (V mod 2^28) * p^-1 = k (mod 2^28); computed from limb 0
// alone (higher limbs are multiples of 2^28).  Anything else is non-zero -- the common case, ~6
// instructions.  Survivors (probability 2^-23) take the exact path: one Montgomery reduction
// brings V into (-p/8, 9p/8), where zero means the carried form equals 0 or p.
PS_HD inline bool fp_is_zero_exact(Fp a) {
    Fp one = fp_zero();
    one.l[0] = 1;
    Fp t = fp_propagate(fp_mul_call(a, one));
    u32 d0 = 0, dp = 0;
#pragma unroll
    for (int i = 0; i < FP_L; i++) {
        d0 |= (u32)t.l[i];
        dp |= (u32)(t.l[i] ^ fp_mod28(i));
    }
    return d0 == 0 || dp == 0;
}
PS_INL bool fp_all_zero(const Fp& a) {  // every limb literally zero (canonical inputs)
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < FP_L; i++) o |= (u32)a.l[i];
    return o == 0;
}
PS_INL bool f_is_zero(const Fp& a) {
    if (fp_all_zero(a)) return true;
    constexpr u32 PINV_POS = ((1u << 28) - FP_INV28) & FP_MASK;  // p^-1 mod 2^28
    u32 q = (((u32)a.l[0] & FP_MASK) * PINV_POS) & FP_MASK;
    if (q > 32u && q < (1u << 28) - 32u) return false;
    return fp_is_zero_exact(a);
}
PS_INL bool f_eq(const Fp& a, const Fp& b) { return f_is_zero(f_sub(a, b)); }

// canonical representative in [0, p), limbs in [0, 2^28): for V in (-2p, 3p)
PS_INL Fp fp_canon(const Fp& a) {
    Fp t = fp_propagate(a);
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {  // negative -> add p (twice covers (-2p, 0))
        bool negv = t.l[FP_L - 1] < 0;
        Fp u;
#pragma unroll
        for (int i = 0; i < FP_L; i++) u.l[i] = t.l[i] + (negv ? fp_mod28(i) : 0);
        t = fp_propagate(u);
    }
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {  // >= p -> subtract p (twice covers [p, 3p))
        Fp u;
#pragma unroll
        for (int i = 0; i < FP_L; i++) u.l[i] = t.l[i] - fp_mod28(i);
        u = fp_propagate(u);
        bool ge = u.l[FP_L - 1] >= 0;
#pragma unroll
        for (int i = 0; i < FP_L; i++) t.l[i] = ge ? u.l[i] : t.l[i];
    }
    return t;
}

// plain 384-bit little-endian words (12 x u32) <-> 28-bit limbs
PS_INL Fp fp_from_words12(const u32* w) {
    Fp r;
#pragma unroll
    for (int j = 0; j < FP_L; j++) {
        int bit = 28 * j, wi = bit >> 5, sh = bit & 31;
        u64 two = (u64)w[wi] | (wi + 1 < 12 ? (u64)w[wi + 1] << 32 : 0ull);
        r.l[j] = (i32)((u32)(two >> sh) & FP_MASK);
    }
    return r;
}
PS_INL void fp_to_words12(u32* w, const Fp& a) {  // a canonical (limbs in [0, 2^28))
#pragma unroll
    for (int i = 0; i < 12; i++) w[i] = 0;
#pragma unroll
    for (int j = 0; j < FP_L; j++) {
        int bit = 28 * j, wi = bit >> 5, sh = bit & 31;
        u64 v = (u64)(u32)a.l[j] << sh;
        w[wi] |= (u32)v;
        if (wi + 1 < 12) w[wi + 1] |= (u32)(v >> 32);
    }
}
// plain canonical value -> Montgomery form (canonical limbs), and back
PS_INL Fp fp_to_mont(const Fp& plain) {
    Fp r2;
#pragma unroll
    for (int i = 0; i < FP_L; i++) r2.l[i] = fp_r2_28(i);
    return fp_canon(f_mul(plain, r2));
}
PS_INL Fp fp_from_mont(const Fp& a) {
    Fp one = fp_zero();
    one.l[0] = 1;
    return fp_canon(f_mul(a, one));
}
// 48 big-endian bytes <-> Montgomery Fp.  ok is cleared for a non-canonical encoding (>= p).
PS_HD inline Fp fp_from_be48(const uint8_t* p, bool& ok) {
    u32 w[12];
    for (int i = 0; i < 12; i++) {
        const uint8_t* q = p + 4 * (11 - i);
        w[i] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
    }
    ok = ok && fe_is_canonical<FpParams>(w);
    return fp_to_mont(fp_from_words12(w));
}
PS_HD inline void fp_to_be48(uint8_t* p, const Fp& a) {
    u32 w[12];
    fp_to_words12(w, fp_from_mont(a));
    for (int i = 0; i < 12; i++) {
        uint8_t* q = p + 4 * (11 - i);
        q[0] = (uint8_t)(w[i] >> 24); q[1] = (uint8_t)(w[i] >> 16); q[2] = (uint8_t)(w[i] >> 8); q[3] = (uint8_t)w[i];
    }
}

// a^(p-2); 0 -> 0.  Square-and-multiply over the bits of p - 2 (a loop, not unrolled).
PS_HD inline Fp f_inv(const Fp& a) {
    u32 e[12];
#pragma unroll
    for (int i = 0; i < 12; i++) e[i] = FpParams::mod(i);
    e[0] -= 2;  // p is odd and its low word is > 2
    Fp acc = fp_one();
    Fp base = a;
    for (int i = 0; i < 381; i++) {
        if ((e[i >> 5] >> (i & 31)) & 1) acc = fp_mul_call(acc, base);
        base = fp_sqr_call(base);
    }
    return acc;
}

PS_INL Fp f_zero(const Fp*) { return fp_zero(); }
PS_INL Fp f_one(const Fp*) { return fp_one(); }
PS_INL Fp f_dbl(const Fp& a) { return f_add(a, a); }

// ---------------------------------------------------------------------------------------
// Fp2 = Fp[u]/(u^2 + 1).  Multiplications go through the out-of-line Fp product; Karatsuba
// sums are carry-save normalised first so every Fp product sees class <= 2 operands, and the
// results are normalised back to class ~1.
// ---------------------------------------------------------------------------------------
struct Fp2 {
    Fp c0, c1;
};
PS_INL Fp2 f_zero(const Fp2*) { return Fp2{fp_zero(), fp_zero()}; }
PS_INL Fp2 f_one(const Fp2*) { return Fp2{fp_one(), fp_zero()}; }
PS_INL Fp2 f_add(const Fp2& a, const Fp2& b) { return Fp2{f_add(a.c0, b.c0), f_add(a.c1, b.c1)}; }
PS_INL Fp2 f_sub(const Fp2& a, const Fp2& b) { return Fp2{f_sub(a.c0, b.c0), f_sub(a.c1, b.c1)}; }
PS_INL Fp2 f_neg(const Fp2& a) { return Fp2{f_neg(a.c0), f_neg(a.c1)}; }
PS_INL Fp2 f_dbl(const Fp2& a) { return f_add(a, a); }
PS_INL Fp2 f_norm(const Fp2& a) { return Fp2{f_norm(a.c0), f_norm(a.c1)}; }
PS_INL Fp2 f_mul(const Fp2& a, const Fp2& b) {
    Fp a0 = f_norm(a.c0), a1 = f_norm(a.c1), b0 = f_norm(b.c0), b1 = f_norm(b.c1);
    Fp t0 = fp_mul_call(a0, b0);
    Fp t1 = fp_mul_call(a1, b1);
    Fp s = fp_mul_call(f_add(a0, a1), f_add(b0, b1));
    return Fp2{f_norm(f_sub(t0, t1)), f_norm(f_sub(f_sub(s, t0), t1))};
}
PS_INL Fp2 f_sqr(const Fp2& a) {
    Fp a0 = f_norm(a.c0), a1 = f_norm(a.c1);
    Fp t = fp_mul_call(f_add(a0, a1), f_sub(a0, a1));
    Fp m = fp_mul_call(a0, a1);
    return Fp2{t, f_norm(f_add(m, m))};
}
PS_INL Fp2 f_mul2sub(const Fp2& a, const Fp2& b, const Fp2& c, const Fp2& d) { return f_sub(f_mul(a, b), f_mul(c, d)); }
PS_INL bool f_is_zero(const Fp2& a) { return f_is_zero(a.c0) && f_is_zero(a.c1); }
PS_INL bool f_eq(const Fp2& a, const Fp2& b) { return f_eq(a.c0, b.c0) && f_eq(a.c1, b.c1); }
PS_HD inline Fp2 f_inv(const Fp2& a) {
    Fp a0 = f_norm(a.c0), a1 = f_norm(a.c1);
    Fp d = f_inv(f_norm(f_add(fp_mul_call(a0, a0), fp_mul_call(a1, a1))));
    return Fp2{fp_mul_call(a0, d), f_neg(fp_mul_call(a1, d))};
}
PS_INL bool fp_all_zero(const Fp2& a) { return fp_all_zero(a.c0) && fp_all_zero(a.c1); }

// ---- square roots and the ZCash "lexicographically larger" bit (point decompression) ----
PS_HD constexpr i32 fp_half28(int i) { constexpr i32 v[FP_L] = PS_FP28_HALF; return v[i]; }
// a^e for a plain exponent given as 12 little-endian words (loop, out-of-line products)
PS_HD inline Fp fp_pow_words(const Fp& a, const u32* e, int nbits) {
    Fp acc = fp_one();
    Fp base = a;
    for (int i = 0; i < nbits; i++) {
        if ((e[i >> 5] >> (i & 31)) & 1) acc = fp_mul_call(acc, base);
        base = fp_sqr_call(base);
    }
    return acc;
}
PS_HD inline void fp_exp_words(u32* e, int delta, int shift) {  // e = (p + delta) >> shift
    i64 c = delta;
    for (int i = 0; i < 12; i++) {
        i64 v = (i64)FpParams::mod(i) + c;
        e[i] = (u32)v;
        c = v >> 32;
    }
    for (int i = 0; i < 12; i++) e[i] = (e[i] >> shift) | ((i + 1 < 12 && shift) ? e[i + 1] << (32 - shift) : 0u);
}
// sqrt in Fp (p = 3 mod 4): a^((p+1)/4); ok=false when a is a non-residue
PS_HD inline Fp fp_sqrt(const Fp& a, bool& ok) {
    u32 e[12];
    fp_exp_words(e, 1, 2);  // (p+1)/4
    Fp s = fp_pow_words(a, e, 380);
    ok = ok && f_is_zero(f_sub(fp_mul_call(s, s), a));
    return s;
}
// canonical plain value > (p-1)/2 ?
PS_HD inline bool fp_lex_larger(const Fp& a_mont) {
    Fp v = fp_from_mont(a_mont);  // canonical plain limbs
    bool gt = false, decided = false;
    for (int i = FP_L - 1; i >= 0; i--) {
        if (!decided && v.l[i] != fp_half28(i)) { gt = v.l[i] > fp_half28(i); decided = true; }
    }
    return gt;
}
PS_HD inline Fp2 fp2_pow_words(const Fp2& a, const u32* e, int nbits);
PS_HD inline bool fp_lex_larger(const Fp2& a_mont) {
    Fp c1 = fp_from_mont(a_mont.c1);
    if (!fp_all_zero(c1)) return fp_lex_larger(a_mont.c1);
    return fp_lex_larger(a_mont.c0);
}
PS_INL Fp2 fp_canon(const Fp2& a) { return Fp2{fp_canon(a.c0), fp_canon(a.c1)}; }
PS_HD inline Fp2 fp2_pow_words(const Fp2& a, const u32* e, int nbits) {
    Fp2 acc = f_one((const Fp2*)0);
    Fp2 base = a;
    for (int i = 0; i < nbits; i++) {
        if ((e[i >> 5] >> (i & 31)) & 1) acc = f_mul(acc, base);
        base = f_sqr(base);
    }
    return acc;
}
// sqrt in Fp2 = Fp[u]/(u^2+1), p = 3 mod 4 (Adj & Rodriguez-Henriquez, alg. 9)
PS_HD inline Fp2 fp_sqrt(const Fp2& a, bool& ok) {
    if (f_is_zero(a)) return f_zero((const Fp2*)0);
    u32 e[12];
    fp_exp_words(e, -3, 2);  // (p-3)/4
    Fp2 a1 = fp2_pow_words(a, e, 380);
    Fp2 alpha = f_mul(f_mul(a1, a1), a);
    Fp2 conj = Fp2{alpha.c0, f_neg(alpha.c1)};  // alpha^p (Frobenius)
    Fp2 a0 = f_mul(conj, alpha);
    Fp2 minus_one = f_neg(f_one((const Fp2*)0));
    if (f_eq(a0, minus_one)) { ok = false; return a; }
    Fp2 x0 = f_mul(a1, a);
    Fp2 x;
    if (f_eq(alpha, minus_one)) {
        x = Fp2{f_neg(x0.c1), x0.c0};  // u * x0
    } else {
        fp_exp_words(e, -1, 1);  // (p-1)/2
        Fp2 b = fp2_pow_words(f_add(f_one((const Fp2*)0), alpha), e, 381);
        x = f_mul(b, x0);
    }
    ok = ok && f_eq(f_sqr(x), a);
    return x;
}


// =======================================================================================
// Fr: unsaturated 10 x 28-bit signed limbs, Montgomery R = 2^280 (same discipline as Fp)
// =======================================================================================
// f_mul contract: class(a)*class(b) <= 11, |A|*|B| <= 2^22 r^2; result limbs 0..8 in [0, 2^28),
// value in (-r/8, 9r/8).  Stored vectors are kept at class ~1 (fr_norm) and |V| < ~64 r.
struct Fr {
    i32 l[10];
};
constexpr int FR_L = 10;
constexpr u32 FR_INV28 = PS_FR28_INV;
PS_HD constexpr i32 fr_mod28(int i) { constexpr i32 v[FR_L] = PS_FR28_MOD; return v[i]; }
PS_HD constexpr i32 fr_r1_28(int i) { constexpr i32 v[FR_L] = PS_FR28_R1; return v[i]; }
PS_HD constexpr i32 fr_r2_28(int i) { constexpr i32 v[FR_L] = PS_FR28_R2; return v[i]; }

PS_INL Fr fr_zero() {
    Fr r;
#pragma unroll
    for (int i = 0; i < FR_L; i++) r.l[i] = 0;
    return r;
}
PS_INL Fr fr_one() {
    Fr r;
#pragma unroll
    for (int i = 0; i < FR_L; i++) r.l[i] = fr_r1_28(i);
    return r;
}
PS_INL Fr fr_add(const Fr& a, const Fr& b) {
    Fr r;
#pragma unroll
    for (int i = 0; i < FR_L; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
PS_INL Fr fr_sub(const Fr& a, const Fr& b) {
    Fr r;
#pragma unroll
    for (int i = 0; i < FR_L; i++) r.l[i] = a.l[i] - b.l[i];
    return r;
}
PS_INL Fr fr_neg(const Fr& a) {
    Fr r;
#pragma unroll
    for (int i = 0; i < FR_L; i++) r.l[i] = -a.l[i];
    return r;
}
PS_INL Fr fr_norm(const Fr& a) {  // one parallel carry-save step
    Fr r;
    r.l[0] = (i32)((u32)a.l[0] & FP_MASK);
#pragma unroll
    for (int i = 1; i < FR_L - 1; i++) r.l[i] = (i32)((u32)a.l[i] & FP_MASK) + (a.l[i - 1] >> 28);
    r.l[FR_L - 1] = a.l[FR_L - 1] + (a.l[FR_L - 2] >> 28);
    return r;
}
PS_INL Fr fr_propagate(const Fr& a) {
    Fr r;
    i32 c = 0;
#pragma unroll
    for (int i = 0; i < FR_L - 1; i++) {
        i32 t = a.l[i] + c;
        r.l[i] = (i32)((u32)t & FP_MASK);
        c = t >> 28;
    }
    r.l[FR_L - 1] = a.l[FR_L - 1] + c;
    return r;
}
#if defined(__HIP_DEVICE_COMPILE__)
// The Montgomery product as ONE chain of multiply-adds per column, carried in the NEGATED domain: N_k = -(column sum).  Then
// the Montgomery digit is N_k mod 2^28 itself (r = 1 mod 2^28), the carry into the next column is N_k >> 28 (the floor of the
// negated sum is the negated CEILING the positive sum needs, see fr_mul), the modulus enters as the constants -r_j, and the
// result's limbs come out as -(N_k mod 2^28): non-positive limbs, the same value, inside the lazy-limb contract like any
// difference.  194 multiply-adds, 19 shifts, 19 masks, 20 negations; no 64-bit add.  (Left to the compiler, the products of a
// column are summed as independent chains -- good for a lone wave's latency, but every merge is a 64-bit add that costs the
// issue time of a multiply-add: 27 of them per product.)  The code is generated: tools/gen_fr_chain.py.
#include "fr_chain.inc"
#endif
PS_INL Fr fr_mul(const Fr& a_in, const Fr& b_in) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PS_FR_MUL_NO_CHAIN)
    // On the device: the single-chain form above.  With a statement per multiply-add (194 compiler-inserted wait states per
    // product) it won only where the chip was full of waves (2^20 gates: -2.5 %) and lost 10 % at 2^10 .. 2^16 gates; with a
    // statement per column it wins 2-6 % at every size (A/B against the code below, 2^10 .. 2^20 gates).
    return fr_mul_chain(a_in, b_in);
#endif
    Fr r;
    i32 m[FR_L];
    i64 acc = 0;
    // Operand limbs that are fresh 32-bit sums (the lazy additions of a butterfly) must reach the products AS 32-bit values:
    // otherwise the compiler widens the sum (sext(x + y) = sext(x) + sext(y), no signed overflow) and multiplies 64 x 64 bits --
    // two v_mul_lo_u32, a v_mad_u64_u32 and a v_add3_u32 where one v_mad_i64_i32 does it.  The first operand only: the second
    // is a twiddle or a constant, often in SGPRs.  (At four waves per SIMD the barrier cost more in spilled registers than it
    // saved; at three -- ntt.hpp, PS_NTT_WAVES -- the quotient at 2^20 gates goes from 8.58 to 8.37 ms.)
    Fr a = a_in;
    const Fr& b = b_in;  // (a barrier on this one too: measured, nothing)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PS_FR_MUL_NO_BARRIER)
#pragma unroll
    for (int i = 0; i < FR_L; i++) asm("" : "+v"(a.l[i]));
#endif
    // The top limb of r is 7: left as a literal the compiler multiplies by it with a v_mul_lo_u32 and adds the product with a
    // 64-bit add (two quarter-rate instructions) where a multiply-add does both -- so it is handed over as an opaque scalar.
    i32 mod_top = fr_mod28(FR_L - 1);
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+s"(mod_top));
#endif
    auto modl = [&](int idx) -> i64 { return idx == FR_L - 1 ? (i64)mod_top : (i64)fr_mod28(idx); };
#pragma unroll
    for (int k = 0; k < FR_L; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (i64)a.l[i] * (i64)b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (i64)m[i] * modl(k - i);
        // r = 1 mod 2^28 (two-adicity 32), so -r^-1 = -1 mod 2^28 and the Montgomery digit is a negation,
        // not a (quarter-rate) v_mul_lo_u32
        static_assert(FR_INV28 == FP_MASK, "m = -acc mod 2^28 needs r = 1 mod 2^28");
        m[k] = (i32)((0u - (u32)acc) & FP_MASK);
        // acc + m[k] * r_0 with r_0 = 1: the low 28 bits cancel, so (acc + m[k]) >> 28 = ceil(acc / 2^28) = (acc + 2^28 - 1) >> 28
        // -- a CONSTANT addend, which the compiler folds into the column's first multiply-add, where m[k] (a register) costs a
        // 64-bit add of its own: ten quarter-rate instructions less per product.
        static_assert(fr_mod28(0) == 1, "the ceiling trick needs r = 1 mod 2^28");
        acc += (i64)FP_MASK;
        acc >>= 28;
    }
#pragma unroll
    for (int k = FR_L; k < 2 * FR_L - 1; k++) {
#pragma unroll
        for (int i = k - FR_L + 1; i < FR_L; i++) acc += (i64)a.l[i] * (i64)b.l[k - i];
#pragma unroll
        for (int i = k - FR_L + 1; i < FR_L; i++) acc += (i64)m[i] * modl(k - i);
        r.l[k - FR_L] = (i32)((u32)acc & FP_MASK);
        acc >>= 28;
    }
    r.l[FR_L - 1] = (i32)acc;
    return r;
}
// canonical representative in [0, r) for V in (-2r, 3r)
PS_INL Fr fr_canon(const Fr& a) {
    Fr t = fr_propagate(a);
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {
        bool negv = t.l[FR_L - 1] < 0;
        Fr u;
#pragma unroll
        for (int i = 0; i < FR_L; i++) u.l[i] = t.l[i] + (negv ? fr_mod28(i) : 0);
        t = fr_propagate(u);
    }
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {
        Fr u;
#pragma unroll
        for (int i = 0; i < FR_L; i++) u.l[i] = t.l[i] - fr_mod28(i);
        u = fr_propagate(u);
        bool ge = u.l[FR_L - 1] >= 0;
#pragma unroll
        for (int i = 0; i < FR_L; i++) t.l[i] = ge ? u.l[i] : t.l[i];
    }
    return t;
}
PS_INL Fr fr_reduce(const Fr& a) { return fr_mul(a, fr_one()); }  // same residue, value back in (-r/8, 9r/8)
PS_HD inline bool fr_is_zero(const Fr& a) {  // V == 0 (mod r); filter on limb 0, exact path behind it
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < FR_L; i++) o |= (u32)a.l[i];
    if (o == 0) return true;
    constexpr u32 RINV_POS = ((1u << 28) - FR_INV28) & FP_MASK;  // r^-1 mod 2^28
    u32 q = (((u32)a.l[0] & FP_MASK) * RINV_POS) & FP_MASK;
    if (q > 4096u && q < (1u << 28) - 4096u) return false;  // |V| < 4096 r by the storage discipline
    Fr t = fr_canon(fr_reduce(a));
    u32 d = 0;
#pragma unroll
    for (int i = 0; i < FR_L; i++) d |= (u32)t.l[i];
    return d == 0;
}
// plain 256-bit little-endian words <-> 28-bit limbs (plain, canonical)
PS_INL Fr fr_from_words8(const u32* w) {
    Fr r;
#pragma unroll
    for (int j = 0; j < FR_L; j++) {
        int bit = 28 * j, wi = bit >> 5, sh = bit & 31;
        u64 two = (u64)w[wi] | (wi + 1 < 8 ? (u64)w[wi + 1] << 32 : 0ull);
        r.l[j] = (i32)((u32)(two >> sh) & FP_MASK);
    }
    return r;
}
PS_INL void fr_to_words8(u32* w, const Fr& a) {  // a canonical
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = 0;
#pragma unroll
    for (int j = 0; j < FR_L; j++) {
        int bit = 28 * j, wi = bit >> 5, sh = bit & 31;
        u64 v = (u64)(u32)a.l[j] << sh;
        if (wi < 8) w[wi] |= (u32)v;
        if (wi + 1 < 8) w[wi + 1] |= (u32)(v >> 32);
    }
}
PS_INL Fr fr_to_mont(const Fr& plain) {  // plain canonical limbs -> Montgomery form
    Fr r2;
#pragma unroll
    for (int i = 0; i < FR_L; i++) r2.l[i] = fr_r2_28(i);
    return fr_mul(plain, r2);
}
PS_INL Fr fr_from_mont(const Fr& a) {  // -> canonical plain limbs
    Fr one = fr_zero();
    one.l[0] = 1;
    return fr_canon(fr_mul(a, one));
}
PS_INL Fr fr_from_u64(u64 v) {
    Fr a = fr_zero();
    a.l[0] = (i32)(v & FP_MASK);
    a.l[1] = (i32)((v >> 28) & FP_MASK);
    a.l[2] = (i32)(v >> 56);
    return fr_to_mont(a);
}
PS_HD inline Fr fr_inv(const Fr& a) {  // a^(r-2)
    u32 e[8];
    for (int i = 0; i < 8; i++) e[i] = FrParams::mod(i);
    e[0] -= 2;  // low word of r is 1: borrow
    // r = ...00000001: r - 2 = ...ffffffff with a borrow from word 1
    e[0] = 0xffffffffu; e[1] = FrParams::mod(1) - 1u;
    Fr acc = fr_one();
    Fr base = fr_reduce(a);
    for (int i = 0; i < 255; i++) {
        if ((e[i >> 5] >> (i & 31)) & 1) acc = fr_mul(acc, base);
        base = fr_mul(base, base);
    }
    return acc;
}

// ---------------------------------------------------------------------------------------
// Fp2s: an Fp2 element SPLIT ACROSS A LANE PAIR (device kernels only).  Even lane holds c0, odd
// lane c1; the partner's limbs come through DPP quad_perm [1,0,3,2] (a full-rate v_mov_dpp, no
// LDS).  Per-lane register state is that of the G1 kernels, so the G2 bucket kernels inline
// everything and spill nothing; the price is 2 x 588 mads per product instead of Karatsuba's
// 3 x 392.  Control flow must be uniform inside a pair (every kernel guarantees it: both lanes
// carry the same logical thread).
// ---------------------------------------------------------------------------------------
struct Fp2s {
    Fp v;
};
#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline int pair_lane() { return (int)(threadIdx.x & 1u); }
__device__ inline i32 pair_swap(i32 x) { return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, false); }   // quad_perm [1,0,3,2]
__device__ inline i32 pair_bcast0(i32 x) { return __builtin_amdgcn_mov_dpp(x, 0xA0, 0xF, 0xF, false); } // quad_perm [0,0,2,2]: the even lane's value
__device__ inline i32 pair_bcast1(i32 x) { return __builtin_amdgcn_mov_dpp(x, 0xF5, 0xF, 0xF, false); } // quad_perm [1,1,3,3]: the odd lane's value
#else  // host pass: never executed, present so that __device__ code parses
PS_HD inline int pair_lane() { return 0; }
PS_HD inline i32 pair_swap(i32 x) { return x; }
PS_HD inline i32 pair_bcast0(i32 x) { return x; }
PS_HD inline i32 pair_bcast1(i32 x) { return x; }
#endif
PS_INL Fp pair_swap(const Fp& a) {
    Fp r;
#pragma unroll
    for (int i = 0; i < FP_L; i++) r.l[i] = pair_swap(a.l[i]);
    return r;
}
PS_INL Fp pair_bcast0(const Fp& a) {
    Fp r;
#pragma unroll
    for (int i = 0; i < FP_L; i++) r.l[i] = pair_bcast0(a.l[i]);
    return r;
}
PS_INL Fp pair_bcast1(const Fp& a) {
    Fp r;
#pragma unroll
    for (int i = 0; i < FP_L; i++) r.l[i] = pair_bcast1(a.l[i]);
    return r;
}
// the partner's component with the sign the complex product wants: even lane -a1, odd lane a0.  (x ^ m) - m negates
// where m is all ones: one v_xad_u32 per limb, no select.
PS_INL Fp pair_cross(const Fp& a) {
    const u32 m = pair_lane() ? 0u : 0xffffffffu;
    Fp r;
#pragma unroll
    for (int i = 0; i < FP_L; i++) r.l[i] = (i32)(((u32)pair_swap(a.l[i]) ^ m) - m);
    return r;
}
PS_INL Fp2s f_zero(const Fp2s*) { return Fp2s{fp_zero()}; }
PS_INL Fp2s f_one(const Fp2s*) { return Fp2s{pair_lane() ? fp_zero() : fp_one()}; }
PS_INL Fp2s f_add(const Fp2s& a, const Fp2s& b) { return Fp2s{f_add(a.v, b.v)}; }
PS_INL Fp2s f_sub(const Fp2s& a, const Fp2s& b) { return Fp2s{f_sub(a.v, b.v)}; }
PS_INL Fp2s f_neg(const Fp2s& a) { return Fp2s{f_neg(a.v)}; }
PS_INL Fp2s f_dbl(const Fp2s& a) { return Fp2s{f_add(a.v, a.v)}; }
PS_INL Fp2s f_norm(const Fp2s& a) { return Fp2s{f_norm(a.v)}; }
PS_INL Fp2s fp_canon(const Fp2s& a) { return Fp2s{fp_canon(a.v)}; }
// (a0 + a1 u)(b0 + b1 u): even lane a0 b0 - a1 b1, odd lane a1 b0 + a0 b1 -- on both lanes
//     mine(a) * b0 + cross(a) * b1,   cross(a) = -a1 on the even lane, a0 on the odd one,
// with b0, b1 broadcast inside the pair: three DPP moves and one xor-add per limb, no per-lane selects.
// Needs class(a)*class(b) <= 4.
PS_INL Fp2s f_mul(const Fp2s& a, const Fp2s& b) {
    return Fp2s{f_mul2add(a.v, pair_bcast0(b.v), pair_cross(a.v), pair_bcast1(b.v))};
}
// (a0 + a1)(a0 - a1) on the even lane, 2 a0 a1 on the odd lane
PS_INL Fp2s f_sqr(const Fp2s& a) {
    const bool odd = pair_lane() != 0;
    Fp me = f_norm(a.v);
    Fp ot = pair_swap(me);
    Fp x, y;
#pragma unroll
    for (int i = 0; i < FP_L; i++) {
        x.l[i] = odd ? 2 * me.l[i] : me.l[i] + ot.l[i];
        y.l[i] = odd ? ot.l[i] : me.l[i] - ot.l[i];
    }
    return Fp2s{f_mul(x, y)};
}
// a*b - c*d in Fp2, four real products per lane under one reduction.  Operands are brought to limb class ~1
// first (two of them arrive as differences in the mixed addition), which keeps the column sums inside 64 bits.
PS_INL Fp2s f_mul2sub(const Fp2s& a, const Fp2s& b, const Fp2s& c, const Fp2s& d) {
    const Fp an = f_norm(a.v), bn = f_norm(b.v), cn = f_norm(c.v), dn = f_norm(d.v);
    return Fp2s{f_mul2add2sub(an, pair_bcast0(bn), pair_cross(an), pair_bcast1(bn), cn, pair_bcast0(dn), pair_cross(cn), pair_bcast1(dn))};
}
// the same two forms with independent column accumulators (f_mulsum_ilp): the lane-cooperative additions of qtail.hpp
PS_INL Fp2s f_mul_ilp(const Fp2s& a, const Fp2s& b) {
    const Fp x[2] = {a.v, pair_cross(a.v)}, y[2] = {pair_bcast0(b.v), pair_bcast1(b.v)};
    const bool neg[2] = {false, false};
    return Fp2s{f_mulsum_ilp<2>(x, y, neg)};
}
PS_INL Fp2s f_mul2sub_ilp(const Fp2s& a, const Fp2s& b, const Fp2s& c, const Fp2s& d) {
    const Fp an = f_norm(a.v), bn = f_norm(b.v), cn = f_norm(c.v), dn = f_norm(d.v);
    const Fp x[4] = {an, pair_cross(an), cn, pair_cross(cn)}, y[4] = {pair_bcast0(bn), pair_bcast1(bn), pair_bcast0(dn), pair_bcast1(dn)};
    const bool neg[4] = {false, false, true, true};
    return Fp2s{f_mulsum_ilp<4>(x, y, neg)};
}
PS_INL bool f_is_zero(const Fp2s& a) {
    i32 z = f_is_zero(a.v) ? 1 : 0;
    return (z & pair_swap(z)) != 0;
}
PS_INL bool f_eq(const Fp2s& a, const Fp2s& b) { return f_is_zero(f_sub(a, b)); }
PS_INL bool fp_all_zero(const Fp2s& a) {
    i32 z = fp_all_zero(a.v) ? 1 : 0;
    return (z & pair_swap(z)) != 0;
}

}  // namespace ps
