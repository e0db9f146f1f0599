// BLS12-381 G1 / G2 group law for the MSM kernels, written once over the coordinate field
// F (Fp for G1, Fp2 for G2).  Curve: y^2 = x^3 + b, a = 0.
//
// Bucket accumulators use extended Jacobian "XYZZ" coordinates (X, Y, ZZ, ZZZ with
// x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2): mixed addition of an affine point costs 8M + 2S
// (madd-2008-s) against 7M + 4S + many more additions for Jacobian, and needs no
// inversion.  The identity is ZZ = 0.  Every exceptional case (identity operands, P + P,
// P + (-P)) is handled, because bit-exactness with the reference's serial
// Point.Mul/Point.Add loop (algebra.go:355-357) must hold for *all* inputs, including
// repeated points and adversarial scalars.
//
// Lazy-limb discipline (field.hpp): coordinates stored in an accumulator are kept at limb class
// ~1 by f_norm() on X3 / Y3 (ZZ, ZZZ are multiplication outputs); every product below then sees
// operand classes whose product is <= 4 and values <= 6p, inside f_mul's contract (<= 8, <= 16p).
// The rare exceptional branches are out of line so the hot loop stays small.
#pragma once
#include "field.hpp"

namespace ps {

template <class F>
struct Affine {  // identity encoded as inf != 0
    F x, y;
};

template <class F>
struct Xyzz {
    F x, y, zz, zzz;
};

template <class F>
PS_INL Xyzz<F> xyzz_identity() {
    Xyzz<F> r;
    r.x = f_zero((const F*)0);
    r.y = f_zero((const F*)0);
    r.zz = f_zero((const F*)0);
    r.zzz = f_zero((const F*)0);
    return r;
}
template <class F>
PS_INL bool xyzz_is_identity(const Xyzz<F>& p) { return f_is_zero(p.zz); }

template <class F>
PS_INL Xyzz<F> xyzz_from_affine(const F& x, const F& y) {
    Xyzz<F> r;
    r.x = x;
    r.y = y;
    r.zz = f_one((const F*)0);
    r.zzz = f_one((const F*)0);
    return r;
}

#if defined(__HIP_DEVICE_COMPILE__)
#define PS_COLD __host__ __device__ __attribute__((noinline))
#else
#define PS_COLD PS_HD inline
#endif

// 2*(x, y) for an affine point (mdbl-2008-s-1)
template <class F>
PS_COLD Xyzz<F> xyzz_dbl_affine(F x, F y) {  // by value: a reference would pin the caller's point in scratch
    F u = f_add(y, y);
    F v = f_sqr(u);
    F w = f_mul(u, v);
    F s = f_mul(x, v);
    F xx = f_sqr(x);
    F m = f_norm(f_add(f_add(xx, xx), xx));
    Xyzz<F> r;
    r.x = f_norm(f_sub(f_sub(f_sqr(m), s), s));
    r.y = f_norm(f_sub(f_mul(m, f_sub(s, r.x)), f_mul(w, y)));
    r.zz = v;
    r.zzz = w;
    return r;
}

// 2*P (dbl-2008-s-1, a = 0)
template <class F>
PS_COLD Xyzz<F> xyzz_dbl(Xyzz<F> p) {
    if (xyzz_is_identity(p)) return p;
    F u = f_add(p.y, p.y);
    F v = f_sqr(u);
    F w = f_mul(u, v);
    F s = f_mul(p.x, v);
    F xx = f_sqr(p.x);
    F m = f_norm(f_add(f_add(xx, xx), xx));
    Xyzz<F> r;
    r.x = f_norm(f_sub(f_sub(f_sqr(m), s), s));
    r.y = f_norm(f_sub(f_mul(m, f_sub(s, r.x)), f_mul(w, p.y)));
    r.zz = f_mul(v, p.zz);
    r.zzz = f_mul(w, p.zzz);
    return r;
}

// acc += (x2, y2), affine and not the identity (madd-2008-s): 8M + 2S.  _inl is forced inline for the accumulation
// kernel: left to the inliner, the lane-pair instance became an out-of-line call (both operands through scratch, the G2
// sum 8.2 -> 11.2 ms) the day a second kernel started using it.
template <class F>
PS_INL void xyzz_madd_inl(Xyzz<F>& acc, const F& x2, const F& y2) {
    if (xyzz_is_identity(acc)) {
        acc = xyzz_from_affine<F>(x2, y2);
        return;
    }
    F u2 = f_mul(x2, acc.zz);
    F s2 = f_mul(y2, acc.zzz);
    F p = f_sub(u2, acc.x);
    F r = f_sub(s2, acc.y);
    if (f_is_zero(p)) {
        if (f_is_zero(r)) acc = xyzz_dbl_affine<F>(x2, y2);
        else acc = xyzz_identity<F>();
        return;
    }
    F pp = f_sqr(p);
    F ppp = f_mul(p, pp);
    F q = f_mul(acc.x, pp);
    F x3 = f_norm(f_sub(f_sub(f_sub(f_sqr(r), ppp), q), q));
    F y3 = f_norm(f_mul2sub(r, f_sub(q, x3), acc.y, ppp));  // classes 2*2 + 1*1
    acc.zz = f_mul(acc.zz, pp);
    acc.zzz = f_mul(acc.zzz, ppp);
    acc.x = x3;
    acc.y = y3;
}
template <class F>
PS_HD inline void xyzz_madd(Xyzz<F>& acc, const F& x2, const F& y2) { xyzz_madd_inl<F>(acc, x2, y2); }

// acc += q (add-2008-s): 12M + 2S.  _inl is forced inline for the latency-bound reduction kernels
// (an out-of-line call passes both 224-byte operands through scratch memory).
template <class F>
PS_INL void xyzz_add_inl(Xyzz<F>& acc, const Xyzz<F>& q) {
    if (xyzz_is_identity(q)) return;
    if (xyzz_is_identity(acc)) {
        acc = q;
        return;
    }
    F u1 = f_mul(acc.x, q.zz);
    F u2 = f_mul(q.x, acc.zz);
    F s1 = f_mul(acc.y, q.zzz);
    F s2 = f_mul(q.y, acc.zzz);
    F p = f_sub(u2, u1);
    F r = f_sub(s2, s1);
    if (f_is_zero(p)) {
        if (f_is_zero(r)) acc = xyzz_dbl<F>(acc);
        else acc = xyzz_identity<F>();
        return;
    }
    F pp = f_sqr(p);
    F ppp = f_mul(p, pp);
    F qq = f_mul(u1, pp);
    F x3 = f_norm(f_sub(f_sub(f_sub(f_sqr(r), ppp), qq), qq));
    F y3 = f_norm(f_mul2sub(r, f_sub(qq, x3), s1, ppp));
    acc.zz = f_mul(f_mul(acc.zz, q.zz), pp);
    acc.zzz = f_mul(f_mul(acc.zzz, q.zzz), ppp);
    acc.x = x3;
    acc.y = y3;
}

template <class F>
PS_HD inline void xyzz_add(Xyzz<F>& acc, const Xyzz<F>& q) { xyzz_add_inl<F>(acc, q); }

// 2*P, forced inline (the Horner loop of the bucket reduction)
template <class F>
PS_INL Xyzz<F> xyzz_dbl_inl(const Xyzz<F>& p) {
    if (xyzz_is_identity(p)) return p;
    F u = f_add(p.y, p.y);
    F v = f_sqr(u);
    F w = f_mul(u, v);
    F s = f_mul(p.x, v);
    F xx = f_sqr(p.x);
    F m = f_norm(f_add(f_add(xx, xx), xx));
    Xyzz<F> r;
    r.x = f_norm(f_sub(f_sub(f_sqr(m), s), s));
    r.y = f_norm(f_sub(f_mul(m, f_sub(s, r.x)), f_mul(w, p.y)));
    r.zz = f_mul(v, p.zz);
    r.zzz = f_mul(w, p.zzz);
    return r;
}

template <class F>
PS_INL Xyzz<F> xyzz_neg(const Xyzz<F>& p) {
    Xyzz<F> r = p;
    r.y = f_neg(p.y);
    return r;
}

// k * P for a small non-negative k (double-and-add, MSB first)
template <class F>
PS_HD inline Xyzz<F> xyzz_mul_small(const Xyzz<F>& p, u32 k) {
    Xyzz<F> acc = xyzz_identity<F>();
    for (int bit = 31; bit >= 0; bit--) {
        acc = xyzz_dbl<F>(acc);
        if ((k >> bit) & 1) xyzz_add<F>(acc, p);
    }
    return acc;
}

// k * P for a 256-bit plain little-endian scalar (8 x u32)
template <class F>
PS_HD inline Xyzz<F> xyzz_mul_scalar(const Xyzz<F>& p, const u32* k) {
    Xyzz<F> acc = xyzz_identity<F>();
    for (int bit = 255; bit >= 0; bit--) {
        acc = xyzz_dbl<F>(acc);
        if ((k[bit >> 5] >> (bit & 31)) & 1) xyzz_add<F>(acc, p);
    }
    return acc;
}

// normalise; returns false for the identity
template <class F>
PS_HD inline bool xyzz_to_affine(const Xyzz<F>& p, F& x, F& y) {
    if (xyzz_is_identity(p)) return false;
    // 1/ZZZ, then 1/ZZ = ZZZ^-2 * ZZ^2 ... cheaper: one inversion of ZZ*ZZZ
    F t = f_inv(f_mul(p.zz, p.zzz));
    F izz = f_mul(t, p.zzz);
    F izzz = f_mul(t, p.zz);
    x = fp_canon(f_mul(p.x, izz));  // stored affine coordinates are always canonical
    y = fp_canon(f_mul(p.y, izzz));
    return true;
}

}  // namespace ps
