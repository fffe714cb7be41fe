// ec.cuh -- short-Weierstrass (a = 0) G1 arithmetic in extended Jacobian ("XYZZ") coordinates.
//
// x = X/ZZ, y = Y/ZZZ with ZZ^3 = ZZZ^2; infinity <=> ZZ == 0.  XYZZ has the cheapest mixed
// addition (8M+2S, EFD madd-2008-s), which is what the MSM bucket accumulation spends its time
// in.  The boundary type of the reference is ark-ec's Jacobian `Projective{X,Y,Z}`
// (primitives/src/pcs/univariate_kzg/mod.rs:109-111 calls `.into_affine()` on it); an XYZZ
// point maps to the Jacobian representative (X*ZZ, Y*ZZZ, ZZ) -- see xyzz_to_jacobian.
#pragma once
#include "fp.cuh"

// Every template below is generic over a field class F (device: Fp<Params> on 32-bit limbs;
// host tail of the MSM: Fp64<Params> on 64-bit limbs, hostfp.hpp) offering + - * sqr dbl neg
// is_zero one() zero().

namespace mzk {

template <class F>
struct Affine {            // packed x||y, Montgomery; (0,0) encodes infinity (not on any b != 0 curve)
    F x, y;
    MZK_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
};

template <class F>
struct XYZZ {
    F x, y, zz, zzz;
    MZK_HD bool is_inf() const { return zz.is_zero(); }
    MZK_HD static XYZZ inf() {
        XYZZ r;
        r.x = F::one(); r.y = F::one(); r.zz = F::zero(); r.zzz = F::zero();
        return r;
    }
    MZK_HD static XYZZ from_affine(const Affine<F>& p) {
        if (p.is_inf()) return inf();
        XYZZ r;
        r.x = p.x; r.y = p.y; r.zz = F::one(); r.zzz = F::one();
        return r;
    }
};

// 2*P for affine P (EFD mdbl-2008-s-1)
template <class F>
MZK_HD XYZZ<F> xyzz_dbl_affine(const Affine<F>& p) {
    if (p.is_inf() || p.y.is_zero()) return XYZZ<F>::inf();
    XYZZ<F> r;
    F u = dbl(p.y);
    F v = sqr(u);
    F w = u * v;
    F s = p.x * v;
    F x2 = sqr(p.x);
    F m = dbl(x2) + x2;
    r.x = sqr(m) - dbl(s);
    r.y = m * (s - r.x) - w * p.y;
    r.zz = v;
    r.zzz = w;
    return r;
}

// 2*P (EFD dbl-2008-s-1)
template <class F>
MZK_HD XYZZ<F> xyzz_dbl(const XYZZ<F>& p) {
    if (p.is_inf() || p.y.is_zero()) return XYZZ<F>::inf();
    XYZZ<F> r;
    F u = dbl(p.y);
    F v = sqr(u);
    F w = u * v;
    F s = p.x * v;
    F x2 = sqr(p.x);
    F m = dbl(x2) + x2;
    r.x = sqr(m) - dbl(s);
    r.y = m * (s - r.x) - w * p.y;
    r.zz = v * p.zz;
    r.zzz = w * p.zzz;
    return r;
}

// P + Q, Q affine (EFD madd-2008-s) with every exceptional case handled:
// Q = inf, P = inf, P = Q (doubling), P = -Q (infinity).
template <class F>
MZK_HD XYZZ<F> xyzz_madd(const XYZZ<F>& p, const Affine<F>& q) {
    if (q.is_inf()) return p;
    if (p.is_inf()) return XYZZ<F>::from_affine(q);
    F u2 = q.x * p.zz;
    F s2 = q.y * p.zzz;
    F pp_ = u2 - p.x;
    F rr = s2 - p.y;
    if (pp_.is_zero()) {
        if (rr.is_zero()) return xyzz_dbl_affine(q);
        return XYZZ<F>::inf();
    }
    XYZZ<F> r;
    F pp = sqr(pp_);
    F ppp = pp_ * pp;
    F qq = p.x * pp;
    r.x = sqr(rr) - ppp - dbl(qq);
    r.y = rr * (qq - r.x) - p.y * ppp;
    r.zz = p.zz * pp;
    r.zzz = p.zzz * ppp;
    return r;
}

// P + Q (EFD add-2008-s), exceptional cases handled
template <class F>
MZK_HD XYZZ<F> xyzz_add(const XYZZ<F>& p, const XYZZ<F>& q) {
    if (q.is_inf()) return p;
    if (p.is_inf()) return q;
    F u1 = p.x * q.zz;
    F u2 = q.x * p.zz;
    F s1 = p.y * q.zzz;
    F s2 = q.y * p.zzz;
    F pp_ = u2 - u1;
    F rr = s2 - s1;
    if (pp_.is_zero()) {
        if (rr.is_zero()) return xyzz_dbl(p);
        return XYZZ<F>::inf();
    }
    XYZZ<F> r;
    F pp = sqr(pp_);
    F ppp = pp_ * pp;
    F qq = u1 * pp;
    r.x = sqr(rr) - ppp - dbl(qq);
    r.y = rr * (qq - r.x) - s1 * ppp;
    r.zz = p.zz * q.zz * pp;
    r.zzz = p.zzz * q.zzz * ppp;
    return r;
}

template <class F>
MZK_HD Affine<F> affine_neg(const Affine<F>& p) {
    Affine<F> r;
    r.x = p.x;
    r.y = neg(p.y);
    return r;
}

// Jacobian representative (X*ZZ, Y*ZZZ, ZZ): x = X*ZZ/ZZ^2, y = Y*ZZZ/ZZ^3 (ZZ^3 = ZZZ^2).
template <class F>
MZK_HD void xyzz_to_jacobian(const XYZZ<F>& p, F& X, F& Y, F& Z) {
    if (p.is_inf()) {
        X = F::one(); Y = F::one(); Z = F::zero();
        return;
    }
    X = p.x * p.zz;
    Y = p.y * p.zzz;
    Z = p.zz;
}

template <class F>
MZK_HD Affine<F> xyzz_to_affine(const XYZZ<F>& p) {
    Affine<F> r;
    if (p.is_inf()) {
        r.x = F::zero(); r.y = F::zero();
        return r;
    }
    F zi = inv(p.zzz);               // 1/ZZZ
    F zzi = sqr(zi * p.zz);          // (ZZ/ZZZ)^2 = ZZ^2/ZZ^3 = 1/ZZ
    r.x = p.x * zzi;
    r.y = p.y * zi;
    return r;
}

}  // namespace mzk
