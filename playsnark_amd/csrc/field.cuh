// BLS12-381 prime-field arithmetic for gfx950 (and for the host-side glue of the same library).
//
// Representation: little-endian 32-bit limbs (Fp: 12, Fr: 8), Montgomery form, canonical
// (< modulus).  CDNA4 has no 64x64 multiplier; the widest integer multiply is
// v_mad_u64_u32 (32x32+64 -> 64), so 32-bit limbs are the native width.  All loops are
// fully unrolled with compile-time indices so every element lives in VGPRs (no scratch).
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
typedef Fe<FpParams> Fp;
typedef Fe<FrParams> Fr;

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

// ---------------------------------------------------------------------------------------
// Fp2 = Fp[u]/(u^2 + 1)
// ---------------------------------------------------------------------------------------
struct Fp2 {
    Fp c0, c1;
};

// Uniform free-function interface so curve code can be written once for Fp and Fp2.
PS_INL Fp f_zero(const Fp*) { return fe_zero<FpParams>(); }
PS_INL Fp f_one(const Fp*) { return fe_one<FpParams>(); }
PS_INL Fp f_add(const Fp& a, const Fp& b) { return fe_add<FpParams>(a, b); }
PS_INL Fp f_sub(const Fp& a, const Fp& b) { return fe_sub<FpParams>(a, b); }
PS_INL Fp f_neg(const Fp& a) { return fe_neg<FpParams>(a); }
PS_INL Fp f_mul(const Fp& a, const Fp& b) { return fe_mul<FpParams>(a, b); }
PS_INL Fp f_sqr(const Fp& a) { return fe_sqr<FpParams>(a); }
PS_INL bool f_is_zero(const Fp& a) { return fe_is_zero<FpParams>(a); }
PS_INL bool f_eq(const Fp& a, const Fp& b) { return fe_eq<FpParams>(a, b); }
PS_HD inline Fp f_inv(const Fp& a) { return fe_inv<FpParams>(a); }

PS_INL Fp2 f_zero(const Fp2*) { return Fp2{fe_zero<FpParams>(), fe_zero<FpParams>()}; }
PS_INL Fp2 f_one(const Fp2*) { return Fp2{fe_one<FpParams>(), fe_zero<FpParams>()}; }
PS_INL Fp2 f_add(const Fp2& a, const Fp2& b) { return Fp2{f_add(a.c0, b.c0), f_add(a.c1, b.c1)}; }
PS_INL Fp2 f_sub(const Fp2& a, const Fp2& b) { return Fp2{f_sub(a.c0, b.c0), f_sub(a.c1, b.c1)}; }
PS_INL Fp2 f_neg(const Fp2& a) { return Fp2{f_neg(a.c0), f_neg(a.c1)}; }
PS_INL Fp2 f_mul(const Fp2& a, const Fp2& b) {
    // Karatsuba: 3 Fp multiplications
    Fp t0 = f_mul(a.c0, b.c0);
    Fp t1 = f_mul(a.c1, b.c1);
    Fp s = f_mul(f_add(a.c0, a.c1), f_add(b.c0, b.c1));
    return Fp2{f_sub(t0, t1), f_sub(f_sub(s, t0), t1)};
}
PS_INL Fp2 f_sqr(const Fp2& a) {
    // (a0+a1)(a0-a1) + 2 a0 a1 u : 2 Fp multiplications
    Fp t = f_mul(f_add(a.c0, a.c1), f_sub(a.c0, a.c1));
    Fp m = f_mul(a.c0, a.c1);
    return Fp2{t, f_add(m, m)};
}
PS_INL bool f_is_zero(const Fp2& a) { return f_is_zero(a.c0) & f_is_zero(a.c1); }
PS_INL bool f_eq(const Fp2& a, const Fp2& b) { return f_eq(a.c0, b.c0) & f_eq(a.c1, b.c1); }
PS_HD inline Fp2 f_inv(const Fp2& a) {
    Fp d = f_inv(f_add(f_sqr(a.c0), f_sqr(a.c1)));
    return Fp2{f_mul(a.c0, d), f_neg(f_mul(a.c1, d))};
}

}  // namespace ps
