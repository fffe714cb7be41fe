// ecx.cuh -- G1 bucket arithmetic on the reduced-radix field (fx.cuh) for curves whose base field
// leaves plenty of head-room in 29-bit limbs (BLS12-381: 14 x 29 = 406 bits for a 381-bit p, so any
// product of values below 2^12 p comes out below 1.5 p and value bounds never bind; only the 32-bit
// limb capacity has to be tracked).  Same formulas as ec.cuh (EFD madd-2008-s / add-2008-s /
// dbl-2008-s-1 / mdbl-2008-s-1), same exceptional cases.
//
// BN254 Fq runs on 9 limbs (261 bits for a 254-bit p: R' / p = 169, 7 bits of head-room; 81 instead of the 100 limb products of the
// 10-limb form of rounds 1-3).  There the value bounds DO bind: a product of A p and B p comes out below (A B / 169 + 1) p, so the
// pads are the smallest that cover their subtrahends -- X::XSUB_XY = 8 p (an accumulator coordinate), XSUB_PQ = 4 p (PPP + 2Q),
// XSUB_2S = 4 p (2S) instead of 64 / 32 / 8 p -- and the accumulator invariant is X, Y < X::XKXY p = 8 p.  The comments below give the
// generous field's bounds; tools/ecx_bounds.py propagates both sets through every line (and refuses 9 limbs with the wide pads).
//
// Limb classes:  M = every limb < 2^29 (an fx_mul result, or a canonical value)
//                N = every limb < 2^29 + 8 (after fx_norm)
// Invariant of an accumulator: X, Y in N with value < XKXY p (64p; BN254: 8p); ZZ, ZZZ in M; infinity <=> ZZ == 0 (all limbs).
#pragma once
#include "ec.cuh"
#include "fx.cuh"

namespace mzk {

template <class X>
struct AffineX {               // x, y canonical in R'-form; (0,0) = infinity
    Fx<X> x, y;
    MZK_HD bool is_inf() const {
        uint32_t a = 0;
#pragma unroll
        for (int i = 0; i < X::XN; i++) a |= x.l[i] | y.l[i];
        return a == 0;
    }
};

template <class X>
struct XYZZX {
    Fx<X> x, y, zz, zzz;
    MZK_HD bool is_inf() const {
        uint32_t a = 0;
#pragma unroll
        for (int i = 0; i < X::XN; i++) a |= zz.l[i];
        return a == 0;
    }
    MZK_HD static XYZZX inf() {
        XYZZX r;
        r.x = Fx<X>::one(); r.y = Fx<X>::one(); r.zz = Fx<X>::zero(); r.zzz = Fx<X>::zero();
        return r;
    }
    MZK_HD static XYZZX from_affine(const AffineX<X>& p) {
        if (p.is_inf()) return inf();
        XYZZX r;
        r.x = p.x; r.y = p.y; r.zz = Fx<X>::one(); r.zzz = Fx<X>::one();
        return r;
    }
};

// -y for a canonical y: 2p - y, limbs < 2^30 (usable directly as a multiplicand)
template <class X>
MZK_HD Fx<X> fx_neg_m(const Fx<X>& y) {
    Fx<X> r;
#pragma unroll
    for (int i = 0; i < X::XN; i++) r.l[i] = X::XSUB2[i] - y.l[i];
    return r;
}

// 2*P for affine P (mdbl-2008-s-1); cold path
template <class X>
MZK_HD XYZZX<X> xyzzx_dbl_affine(const AffineX<X>& p) {
    XYZZX<X> r;
    const Fx<X> u = fx_norm(fx_add(p.y, p.y));                              // 2y, N
    const Fx<X> v = fx_sqr(u);                                              // M
    const Fx<X> w = fx_mul(u, v);                                           // M
    const Fx<X> s = fx_mul(p.x, v);                                         // M
    const Fx<X> x2 = fx_sqr(p.x);                                           // M
    const Fx<X> m = fx_norm(fx_add(fx_add(x2, x2), x2));                    // 3x^2, N, < 6p
    const Fx<X> mm = fx_sqr(m);                                             // M
    r.x = fx_norm(fx_sub_pad<X>(mm, fx_add(s, s), X::XSUB_2S));                               // M + 8p - 2S: N, < 10p
    const Fx<X> d = fx_norm(fx_sub_pad<X>(s, r.x, X::XSUB_XY));              // S - X3, N
    r.y = fx_norm(fx_sub2(fx_mul(m, d), fx_mul(w, p.y)));                   // N, < 4p
    r.zz = v;
    r.zzz = w;
    return r;
}

// 2*P (dbl-2008-s-1); cold path
template <class X>
MZK_HD XYZZX<X> xyzzx_dbl(const XYZZX<X>& p) {
    if (p.is_inf()) return p;
    XYZZX<X> r;
    const Fx<X> u = fx_norm(fx_add(p.y, p.y));                              // N
    const Fx<X> v = fx_sqr(u);
    const Fx<X> w = fx_mul(u, v);
    const Fx<X> s = fx_mul(p.x, v);
    const Fx<X> x2 = fx_sqr(p.x);
    const Fx<X> m = fx_norm(fx_add(fx_add(x2, x2), x2));
    const Fx<X> mm = fx_sqr(m);
    r.x = fx_norm(fx_sub_pad<X>(mm, fx_add(s, s), X::XSUB_2S));
    const Fx<X> d = fx_norm(fx_sub_pad<X>(s, r.x, X::XSUB_XY));
    r.y = fx_norm(fx_sub2(fx_mul(m, d), fx_mul(w, p.y)));
    r.zz = fx_mul(v, p.zz);
    r.zzz = fx_mul(w, p.zzz);
    return r;
}

// P +- Q, Q affine and canonical; `negate` adds -Q.  10 products -- the two of Y3 share one Montgomery reduction (fx_mul2) --, 4 fx_norm.
template <class X>
MZK_HD XYZZX<X> xyzzx_madd(const XYZZX<X>& p, const AffineX<X>& q, bool negate) {
    if (q.is_inf()) return p;
    const Fx<X> qy = negate ? fx_neg_m(q.y) : q.y;                          // limbs < 2^30
    if (p.is_inf()) {
        XYZZX<X> r;
        r.x = q.x; r.y = fx_norm(qy); r.zz = Fx<X>::one(); r.zzz = Fx<X>::one();
        return r;
    }
    const Fx<X> u2 = fx_mul(q.x, p.zz);                                     // M
    const Fx<X> s2 = fx_mul(qy, p.zzz);                                     // M
    const Fx<X> pp_ = fx_norm(fx_sub_pad<X>(u2, p.x, X::XSUB_XY));           // U2 - X1, N
    const Fx<X> rr_ = fx_norm(fx_sub_pad<X>(s2, p.y, X::XSUB_XY));           // S2 - Y1, N
    const Fx<X> pp = fx_sqr(pp_);                                           // M
    const Fx<X> rr2 = fx_sqr(rr_);                                          // M
    if (fx_is_zero_m(pp)) {                                                 // U2 == X1 (mod p)
        if (fx_is_zero_m(rr2)) {                                            // same point: double it
            AffineX<X> qq;
            qq.x = q.x;
            qq.y = fx_norm(qy);
            return xyzzx_dbl_affine(qq);
        }
        return XYZZX<X>::inf();                                             // P = -Q
    }
    XYZZX<X> r;
    const Fx<X> ppp = fx_mul(pp_, pp);                                      // M
    const Fx<X> qv = fx_mul(p.x, pp);                                       // M
    r.x = fx_norm(fx_sub_pad<X>(rr2, fx_add(ppp, fx_add(qv, qv)), X::XSUB_PQ));              // R^2 - PPP - 2Q: N, < 34p
    const Fx<X> d = fx_norm(fx_sub_pad<X>(qv, r.x, X::XSUB_XY));             // Q - X3, N
    r.y = fx_mul2(rr_, d, p.y, fx_neg_m(ppp));                              // R (Q - X3) - Y1 PPP in ONE reduction: M, < 2p
    r.zz = fx_mul(p.zz, pp);                                                // M
    r.zzz = fx_mul(p.zzz, ppp);                                             // M
    return r;
}

// P + Q, both accumulators.  14 products (two of them fused into one reduction).
template <class X>
MZK_HD XYZZX<X> xyzzx_add(const XYZZX<X>& p, const XYZZX<X>& q) {
    if (q.is_inf()) return p;
    if (p.is_inf()) return q;
    const Fx<X> u1 = fx_mul(p.x, q.zz), u2 = fx_mul(q.x, p.zz);             // M
    const Fx<X> s1 = fx_mul(p.y, q.zzz), s2 = fx_mul(q.y, p.zzz);           // M
    const Fx<X> pp_ = fx_norm(fx_sub2(u2, u1));                             // N, < 4p
    const Fx<X> rr_ = fx_norm(fx_sub2(s2, s1));                             // N
    const Fx<X> pp = fx_sqr(pp_);
    const Fx<X> rr2 = fx_sqr(rr_);
    if (fx_is_zero_m(pp)) {
        if (fx_is_zero_m(rr2)) return xyzzx_dbl(p);
        return XYZZX<X>::inf();
    }
    XYZZX<X> r;
    const Fx<X> ppp = fx_mul(pp_, pp);
    const Fx<X> qv = fx_mul(u1, pp);
    r.x = fx_norm(fx_sub_pad<X>(rr2, fx_add(ppp, fx_add(qv, qv)), X::XSUB_PQ));
    const Fx<X> d = fx_norm(fx_sub_pad<X>(qv, r.x, X::XSUB_XY));
    r.y = fx_mul2(rr_, d, s1, fx_neg_m(ppp));
    r.zz = fx_mul(fx_mul(p.zz, q.zz), pp);
    r.zzz = fx_mul(fx_mul(p.zzz, q.zzz), ppp);
    return r;
}

#if defined(__HIPCC__)
// ---- one addition on FOUR lanes ----------------------------------------------------------------------------------------------
// A level of the bucket reduction is ONE addition deep: when it has fewer additions than the chip has lanes, its time is the latency
// of the 14 dependent-ish products of xyzzx_add on a lone lane (~17 us with loads).  The products of an addition form four rounds of
// at most four independent ones, so a QUAD of lanes (lane & 3 = role) takes one addition in four product-latencies: lane k holds
// coordinate k (x, y, zz, zzz) of both operands and ends with coordinate k of the sum; operands move by DPP quad permutes (a VALU
// move per limb, no LDS, no barrier).  All four lanes run the same instruction stream; a lane's product of a round may be unused.
//   round 1   L0 U1 = x1 zz2     L1 S1 = y1 zzz2    L2 U2 = x2 zz1     L3 S2 = y2 zzz1      (own first operand x partner's second)
//             d = partner's - own:  L0 P = U2 - U1,  L1 R = S2 - S1  (L2 -P, L3 -R)
//   round 2   L0 PP = P^2        L1 RR = R^2        L2 zz1 zz2         L3 zzz1 zzz2
//   round 3   L0 PPP = P PP      L1 Q = U1 PP       L2 ZZ3 = zz1 zz2 PP
//             L1 X3 = RR - PPP - 2Q, Q - X3
//   round 4   L0 S1 PPP          L1 R (Q - X3)      L3 ZZZ3 = zzz1 zzz2 PPP;   L1 Y3 = R (Q - X3) - S1 PPP
// Same operations, pads and operand classes as xyzzx_add except Y3, which is a padded difference of two products (class N, < 4p,
// as in xyzzx_dbl) instead of one fused product; tools/ecx_bounds.py (add_quad) walks it for both fields.  Exceptional operands
// (an infinity, P = +-Q) are detected on the lanes that see them, OR-ed over the quad, and sent through xyzzx_add on all four lanes.
template <int CTRL, class X>
__device__ __forceinline__ Fx<X> fx_quad_perm(const Fx<X>& a) {
    Fx<X> r;
#pragma unroll
    for (int i = 0; i < X::XN; i++) r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.l[i], CTRL, 0xF, 0xF, true);
    return r;
}
template <class X>
__device__ __forceinline__ Fx<X> fx_sel(bool c, const Fx<X>& a, const Fx<X>& b) {
    Fx<X> r;
#pragma unroll
    for (int i = 0; i < X::XN; i++) r.l[i] = c ? a.l[i] : b.l[i];
    return r;
}
constexpr int QP_SWAP2 = 0x4E, QP_SWAP1 = 0xB1, QP_L0 = 0x00, QP_L1 = 0x55, QP_L2 = 0xAA, QP_L3 = 0xFF;       // quad_perm controls

// ca, cb: coordinate `role` of the accumulators a and b; returns coordinate `role` of a + b.  The four lanes of a quad must all be active.
template <class X>
__device__ __forceinline__ Fx<X> xyzzx_add_quad(const Fx<X>& ca, const Fx<X>& cb, int role) {
    const bool r0 = role == 0, r1 = role == 1, hi = role >= 2;
    uint32_t za = 0, zb = 0;
#pragma unroll
    for (int i = 0; i < X::XN; i++) { za |= ca.l[i]; zb |= cb.l[i]; }
    int special = (role == 2 && (za == 0 || zb == 0)) ? 1 : 0;                 // an operand at infinity (zz == 0)
    const Fx<X> p1 = fx_mul(ca, fx_quad_perm<QP_SWAP2>(cb));                    // U1 | S1 | U2 | S2
    const Fx<X> d = fx_norm(fx_sub2(fx_quad_perm<QP_SWAP2>(p1), p1));           // P | R | -P | -R: N, < 4p
    const Fx<X> p2 = fx_mul(fx_sel(hi, ca, d), fx_sel(hi, cb, d));              // PP | RR | zz1 zz2 | zzz1 zzz2
    if (r0 && fx_is_zero_m(p2)) special = 1;                                    // U2 == U1: the same point or inverse points
    special |= __builtin_amdgcn_update_dpp(0, special, QP_SWAP1, 0xF, 0xF, true);
    special |= __builtin_amdgcn_update_dpp(0, special, QP_SWAP2, 0xF, 0xF, true);
    if (special) {                                                              // (quad-uniform) every lane adds the whole points
        XYZZX<X> a, b;
        a.x = fx_quad_perm<QP_L0>(ca); a.y = fx_quad_perm<QP_L1>(ca); a.zz = fx_quad_perm<QP_L2>(ca); a.zzz = fx_quad_perm<QP_L3>(ca);
        b.x = fx_quad_perm<QP_L0>(cb); b.y = fx_quad_perm<QP_L1>(cb); b.zz = fx_quad_perm<QP_L2>(cb); b.zzz = fx_quad_perm<QP_L3>(cb);
        const XYZZX<X> r = xyzzx_add(a, b);
        return fx_sel(hi, fx_sel(role == 2, r.zz, r.zzz), fx_sel(r0, r.x, r.y));
    }
    const Fx<X> pp = fx_quad_perm<QP_L0>(p2);
    const Fx<X> u1 = fx_quad_perm<QP_L0>(p1);
    const Fx<X> p3 = fx_mul(fx_sel(r0, d, fx_sel(r1, u1, p2)), pp);             // PPP | Q | ZZ3 | -
    const Fx<X> ppp = fx_quad_perm<QP_L0>(p3);
    const Fx<X> x3 = fx_norm(fx_sub_pad<X>(p2, fx_add(ppp, fx_add(p3, p3)), X::XSUB_PQ));        // L1: RR - PPP - 2Q
    const Fx<X> qx = fx_norm(fx_sub_pad<X>(p3, x3, X::XSUB_XY));                // L1: Q - X3
    const Fx<X> s1 = fx_quad_perm<QP_L1>(p1);
    const Fx<X> p4 = fx_mul(fx_sel(r0, s1, fx_sel(r1, d, p2)), fx_sel(r1, qx, ppp));             // S1 PPP | R (Q - X3) | - | ZZZ3
    const Fx<X> y3 = fx_norm(fx_sub2(p4, fx_quad_perm<QP_L0>(p4)));             // L1: N, < 4p
    const Fx<X> x3_0 = fx_quad_perm<QP_L1>(x3);
    return fx_sel(hi, fx_sel(role == 2, p3, p4), fx_sel(r0, x3_0, y3));
}
#endif

template <class X>
MZK_HD XYZZ<Fp<X>> xyzzx_to_boundary(const XYZZX<X>& p) {
    XYZZ<Fp<X>> r;
    if (p.is_inf()) return XYZZ<Fp<X>>::inf();
    r.x = fx_to_boundary<X>(p.x); r.y = fx_to_boundary<X>(p.y);
    r.zz = fx_to_boundary<X>(p.zz); r.zzz = fx_to_boundary<X>(p.zzz);
    return r;
}

}  // namespace mzk
