// fx.cuh -- reduced-radix ("lazy") prime-field arithmetic for gfx950: XN limbs of 29 bits in
// 32-bit VGPRs, Montgomery radix R' = 2^(29*XN).
//
// Why (profiles/r01_valu_ubench2.txt, profiles/r01_fp_bench.txt): on MI355X every VOP3 instruction
// -- v_mad_u64_u32 included -- issues at the same rate, ~1.4x a VOP2 add, and carry-chained adds are
// as dear as multiplies.  With 32-bit limbs each partial product costs two instructions
// (v_mad_u64_u32 + v_addc_co_u32, fp_mul_gen.cuh); with 29-bit limbs a whole column of partial
// products fits a 64-bit accumulator (28 * 2^58 < 2^63), so a product is ONE v_mad_u64_u32, and
// additions are plain limb-wise v_add_u32 with no carry chain at all.
//
// Values are kept lazily reduced: an element stands for its integer value modulo p, may exceed p by
// a small factor (head-room 2^HEADROOM_BITS), and its limbs may exceed 2^29 after additions.  The
// bounds every operation needs are stated on the operation; the kernels' comments track them.
//   class M  : output of fx_mul: every limb < 2^29, value < 2p.
// Memory images: the boundary's 32-bit-limb Montgomery form (R = 2^(32*N)) is unpacked/packed at
// the edges (fx_unpack / fx_pack_canonical).  Multiplying a boundary-form value by a constant held
// in R'-form leaves it in boundary form: fx_mul(x*R, w*R') = x*w*R, so NTT data never changes form.
#pragma once
#include "fp.cuh"

namespace mzk {

constexpr int XL = 29;
constexpr uint32_t XMASK = (1u << XL) - 1;

template <class X>
struct Fx {
    static constexpr int N = X::XN;
    uint32_t l[N];
    MZK_HD static Fx zero() {
        Fx r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = 0;
        return r;
    }
    MZK_HD static Fx from_const(const uint32_t (&c)[X::XN]) {
        Fx r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = c[i];
        return r;
    }
    MZK_HD static Fx one() { return from_const(X::XONE); }
};

// a*b/R' mod p, lazily reduced.  Requires sum over any column of limb products < 2^64: with XN <= 14
// that holds when every limb of a and of b is < 2^29 + 2^27 (weakly normalised operands, one of them
// may be twice that).  Value bound: a < A*p, b < B*p  =>  result < p*(A*B/2^HEADROOM + 1).
// Result is class M (limbs < 2^29) whenever that bound is <= 2p.
template <class X>
MZK_HD Fx<X> fx_mul(const Fx<X>& x, const Fx<X>& y) {
    constexpr int N = X::XN;
    uint32_t m[N];
    Fx<X> t;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)x.l[i] * y.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * X::XP[k - i];
        m[k] = ((uint32_t)acc * X::XINV) & XMASK;
        acc += (uint64_t)m[k] * X::XP[0];
        acc >>= XL;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
        for (int i = k - N + 1; i < N; i++) {
            acc += (uint64_t)x.l[i] * y.l[k - i];
            acc += (uint64_t)m[i] * X::XP[k - i];
        }
        t.l[k - N] = (uint32_t)acc & XMASK;
        acc >>= XL;
    }
    t.l[N - 1] = (uint32_t)acc;
    return t;
}
// (x*y + u*v)/R' mod p with ONE Montgomery reduction: 3 N^2 multiply-adds instead of the 4 N^2 of two products (the m*p half is
// shared), and the sum needs no pad, subtraction or normalisation.  Column sums must stay below 2^64: with N <= 14,
//   14 * max|x_i y_j| + 14 * max|u_i v_j| + 14 * 2^58 < 2^64,
// which at N = 14 (BLS12-381 Fq) holds for x, y, u of class N / M (limbs < 2^29 + 8) and v < 2^30 (a negated class-M value,
// fx_neg_m): 14 (1 + 2 + 1) 2^58 = 56 * 2^58 -- what ecx.cuh passes.  It does NOT hold for merely weakly normalised x, y, u
// (< 2^29 + 2^27): 14 (1.57 + 2.5 + 1) 2^58 = 70.9 * 2^58 > 2^64; normalise such operands first (fx_norm).
// Value bound: result < p (A B + C D) / 2^HEADROOM + p.
template <class X>
MZK_HD Fx<X> fx_mul2(const Fx<X>& x, const Fx<X>& y, const Fx<X>& u, const Fx<X>& v) {
    constexpr int N = X::XN;
    uint32_t m[N];
    Fx<X> t;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) {
            acc += (uint64_t)x.l[i] * y.l[k - i];
            acc += (uint64_t)u.l[i] * v.l[k - i];
        }
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * X::XP[k - i];
        m[k] = ((uint32_t)acc * X::XINV) & XMASK;
        acc += (uint64_t)m[k] * X::XP[0];
        acc >>= XL;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
        for (int i = k - N + 1; i < N; i++) {
            acc += (uint64_t)x.l[i] * y.l[k - i];
            acc += (uint64_t)u.l[i] * v.l[k - i];
            acc += (uint64_t)m[i] * X::XP[k - i];
        }
        t.l[k - N] = (uint32_t)acc & XMASK;
        acc >>= XL;
    }
    t.l[N - 1] = (uint32_t)acc;
    return t;
}
// x^2/R': the off-diagonal products x_i*x_j (i < j) are taken once against the doubled operand 2*x_j
// (limbs < 2^30: no overflow, and a column still sums to < 2^64), so the x*x half needs N(N+1)/2
// multiply-adds instead of N^2 -- 23 % fewer v_mad_u64_u32 per squaring at 14 limbs.  Same contract as fx_mul.
template <class X>
MZK_HD Fx<X> fx_sqr(const Fx<X>& x) {
    constexpr int N = X::XN;
    uint32_t m[N], x2[N];
    Fx<X> t;
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < N; i++) x2[i] = x.l[i] << 1;
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) acc += (uint64_t)x.l[i] * x2[k - i];
        if (k % 2 == 0) acc += (uint64_t)x.l[k / 2] * x.l[k / 2];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * X::XP[k - i];
        m[k] = ((uint32_t)acc * X::XINV) & XMASK;
        acc += (uint64_t)m[k] * X::XP[0];
        acc >>= XL;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
        for (int i = k - N + 1; 2 * i < k; i++) acc += (uint64_t)x.l[i] * x2[k - i];
        if (k % 2 == 0) acc += (uint64_t)x.l[k / 2] * x.l[k / 2];
#pragma unroll
        for (int i = k - N + 1; i < N; i++) acc += (uint64_t)m[i] * X::XP[k - i];
        t.l[k - N] = (uint32_t)acc & XMASK;
        acc >>= XL;
    }
    t.l[N - 1] = (uint32_t)acc;
    return t;
}

template <class X>
MZK_HD Fx<X> fx_add(const Fx<X>& a, const Fx<X>& b) {
    Fx<X> r;
#pragma unroll
    for (int i = 0; i < X::XN; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
// a - b + K*p with the pad's limbs spread so no limb underflows (constants.cuh: XSUB2 needs b's limbs
// < 2^29, XSUB8 < 2^30 - 1, XSUB32 < 2^31 - 3; the value of b must be < K*p).
template <class X>
MZK_HD Fx<X> fx_sub_pad(const Fx<X>& a, const Fx<X>& b, const uint32_t (&pad)[X::XN]) {
    Fx<X> r;
#pragma unroll
    for (int i = 0; i < X::XN; i++) r.l[i] = a.l[i] + (pad[i] - b.l[i]);
    return r;
}
template <class X> MZK_HD Fx<X> fx_sub2(const Fx<X>& a, const Fx<X>& b) { return fx_sub_pad<X>(a, b, X::XSUB2); }
template <class X> MZK_HD Fx<X> fx_sub8(const Fx<X>& a, const Fx<X>& b) { return fx_sub_pad<X>(a, b, X::XSUB8); }
template <class X> MZK_HD Fx<X> fx_sub32(const Fx<X>& a, const Fx<X>& b) { return fx_sub_pad<X>(a, b, X::XSUB32); }

// one round of carries, all limbs at once: limbs < 2^32 in, limbs < 2^29 + 8 out (top limb keeps all)
template <class X>
MZK_HD Fx<X> fx_norm(const Fx<X>& a) {
    constexpr int N = X::XN;
    Fx<X> r;
    r.l[0] = a.l[0] & XMASK;
#pragma unroll
    for (int i = 1; i < N - 1; i++) r.l[i] = (a.l[i] & XMASK) + (a.l[i - 1] >> XL);
    r.l[N - 1] = a.l[N - 1] + (a.l[N - 2] >> XL);
    return r;
}
// full carry propagation: every limb < 2^29 (the value must be < 2^(29*XN))
template <class X>
MZK_HD Fx<X> fx_carry(const Fx<X>& a) {
    constexpr int N = X::XN;
    Fx<X> r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N - 1; i++) {
        uint32_t t = a.l[i] + c;
        r.l[i] = t & XMASK;
        c = t >> XL;
    }
    r.l[N - 1] = a.l[N - 1] + c;
    return r;
}
// value < 2p with fully carried limbs -> canonical [0, p)
template <class X>
MZK_HD Fx<X> fx_cond_sub_p(const Fx<X>& a) {
    constexpr int N = X::XN;
    Fx<X> d;
    uint32_t b = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint32_t t = a.l[i] - X::XP[i] - b;
        b = t >> 31;
        d.l[i] = t & XMASK;
    }
    Fx<X> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = b ? a.l[i] : d.l[i];
    return r;
}
// lazily reduced (value < 2p, any limb state that fx_carry accepts) -> canonical
template <class X>
MZK_HD Fx<X> fx_canonical(const Fx<X>& a) {
    return fx_cond_sub_p(fx_carry(a));
}
// class M value == 0 mod p ?  (value < 2p, limbs fully carried: it is 0 or p)
template <class X>
MZK_HD bool fx_is_zero_m(const Fx<X>& a) {
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < X::XN; i++) { z |= a.l[i]; e |= a.l[i] ^ X::XP[i]; }
    return z == 0 || e == 0;
}

// boundary image (X::N little-endian 32-bit words, any value < 2^(32*N)) -> 29-bit limbs (< 2^29 each)
template <class X>
MZK_HD Fx<X> fx_unpack(const uint32_t* w) {
    constexpr int N = X::XN, W = X::N;
    Fx<X> r;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int bit = XL * i, wi = bit >> 5, s = bit & 31;
        uint32_t v = wi < W ? (w[wi] >> s) : 0u;
        if (s > 32 - XL && wi + 1 < W) v |= w[wi + 1] << (32 - s);
        r.l[i] = v & XMASK;
    }
    return r;
}
// fully carried limbs, value < 2^(32*X::N) -> boundary image
template <class X>
MZK_HD void fx_pack(uint32_t* w, const Fx<X>& a) {
    constexpr int N = X::XN, W = X::N;
#pragma unroll
    for (int j = 0; j < W; j++) {
        const int bit = 32 * j, i0 = bit / XL, o = bit - XL * i0;
        uint32_t v = a.l[i0] >> o;
        if (i0 + 1 < N) v |= a.l[i0 + 1] << (XL - o);
        if (2 * XL - o < 32 && i0 + 2 < N) v |= a.l[i0 + 2] << (2 * XL - o);
        w[j] = v;
    }
}

// boundary <-> internal
template <class X>
MZK_HD Fx<X> fx_from_boundary(const Fp<X>& a) {                // x*R (packed) -> x*R' canonical
    return fx_canonical(fx_mul(fx_unpack<X>(a.l), Fx<X>::from_const(X::XTO)));
}
template <class X>
MZK_HD Fp<X> fx_to_boundary(const Fx<X>& a) {                  // lazy x*R' (limbs N) -> x*R canonical, packed
    Fp<X> r;
    fx_pack<X>(r.l, fx_canonical(fx_mul(a, Fx<X>::from_const(X::XFROM))));
    return r;
}

// 1/a in the internal form (a R' -> a^-1 R'), fx_inv(0) = 0: a^(p-2) with 4-bit windows taken from the top -- 32 X::N squarings
// and at most 8 X::N products against a^1 .. a^15 -- on the reduced-radix product: about half the time of the bitwise Fermat
// power on 32-bit limbs (fp.cuh inv) that a lone wave spends on the one inversion of a batch.  The loops keep ONE call site each of
// fx_sqr and fx_mul (code size: the instruction cache), the table is indexed at run time (on the device it lives in scratch).
// a: any lazy value fx_mul accepts (limbs < 2^29 + 2^27, value < 2^HEADROOM p).
template <class X>
MZK_HD Fx<X> fx_inv(const Fx<X>& a) {
    Fx<X> tab[16];
    tab[0] = Fx<X>::one();
    tab[1] = fx_mul(a, tab[0]);                                  // class M
#pragma unroll 1
    for (int i = 2; i < 16; i++) tab[i] = fx_mul(tab[i - 1], tab[1]);
    uint32_t e[X::N];                                            // p - 2 (p is odd and > 2: no borrow beyond word 0 unless it is 1)
#pragma unroll
    for (int i = 0; i < X::N; i++) e[i] = X::MOD[i];
    uint32_t borrow = e[0] < 2 ? 1u : 0u;
    e[0] -= 2;
#pragma unroll
    for (int i = 1; i < X::N; i++) { const uint32_t t = e[i]; e[i] = t - borrow; borrow = (borrow && t == 0) ? 1u : 0u; }
    Fx<X> acc = tab[0];
#pragma unroll 1
    for (int d = 8 * X::N - 1; d >= 0; d--) {
#pragma unroll 1
        for (int k = 0; k < 4; k++) acc = fx_sqr(acc);
        const uint32_t nib = (e[d >> 3] >> ((d & 7) * 4)) & 15u;
        if (nib) acc = fx_mul(acc, tab[nib]);
    }
    return acc;
}

#if defined(__HIPCC__)
template <class X>
MZK_D Fx<X> fx_load_packed(const uint32_t* __restrict__ p) {
    Fp<X> t = load_fp<X>(p);
    return fx_unpack<X>(t.l);
}
template <class X>
MZK_D void fx_store_packed(uint32_t* __restrict__ p, const Fx<X>& fully_carried) {
    Fp<X> t;
    fx_pack<X>(t.l, fully_carried);
    store_fp<X>(p, t);
}
#endif

}  // namespace mzk
