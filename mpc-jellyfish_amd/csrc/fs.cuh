// fs.cuh -- signed, lazily reduced 29-bit-limb arithmetic for products whose one operand is a CONSTANT known when the plan is
// built (NTT twiddles): Barrett reduction with a precomputed quotient ("Shoup" form) instead of a Montgomery product.
//
// Why.  A Montgomery product of 9-limb operands (fx.cuh) costs 2 * 81 multiply-adds: a*b and m*p in full.  When w is a constant,
// wq = floor(w * 2^261 / p) can be stored beside it, and
//        q^ = floor(a * wq / 2^261)     -- only the HIGH half of that product is needed (columns 7..16: 53 multiply-adds)
//        r  = a * w - q^ * p            -- only the LOW 261 bits of both products are needed (2 * 45 multiply-adds)
// is congruent to a*w with |r| < 3p: 143 multiply-adds instead of 162, and -- unlike a Montgomery product -- no factor R^-1, so
// NTT data keep whatever form they arrive in.  Limbs are SIGNED here (v_mad_i64_i32): a - b needs no multiple-of-p pad, and a
// butterfly's two outputs need no renormalisation before they are multiplied again (bounds below).
//
// Representation: Fs = 9 int32 limbs, value = sum l_i 2^(29 i), any sign.
//   class C : limbs 0..7 in [0, 2^29), limb 8 signed -- what fs_mulc and fs_carry return.
// Contracts are stated on each function; ntt_fx.cuh tracks them per stage.
#pragma once
#include "fx.cuh"

namespace mzk {

constexpr int FS_N = 9;
constexpr int FS_TW_WORDS = 20;          // a twiddle record in global memory: w (9 limbs), wq (9 limbs), 2 words of padding = 5 x 16 B

template <class X>
struct Fs {
    int32_t l[FS_N];
    MZK_HD static Fs zero() {
        Fs r;
#pragma unroll
        for (int i = 0; i < FS_N; i++) r.l[i] = 0;
        return r;
    }
};

struct FsTw {                            // one constant operand: w canonical (limbs < 2^29), wq = floor(w 2^261 / p) (limbs < 2^29)
    int32_t w[FS_N], q[FS_N];
};

// a * w (mod p), lazily.  Requires |a.l[i]| <= 2^30 (column sums: 9 * 2^30 * 2^29 + 9 * 2^29 * 2^29 + carry < 2^63) and
// |value(a)| < 2^261.  Returns class C with value in (-2p, 3p):
//   q^ = floor(a wq / 2^261) computed from columns 7..16 of the product differs from floor(a w / p) by -2 .. +1
//   (wq's own rounding contributes less than 1 in absolute value because |a| < 2^261; the dropped columns 0..6 less than 2^-24).
// -p's limbs as values the compiler cannot see through: both scalar fields have limbs of special shape (BLS12-381: 1 and 2^29 - 8;
// BN254: 2^28 + 1), and clang turns q * (-1) or q * (8 - 2^29) into 64-bit shift / subtract sequences of two to four VOP3
// instructions where one v_mad_i64_i32 does (26 of a product's 143 multiply-adds on BLS12-381, profiles/r02_ntt_isa.txt).
template <class X>
MZK_HD int32_t fs_neg_p_limb(int i) {
    int32_t v = -(int32_t)X::XP[i];
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(v));
#endif
    return v;
}
// acc += x * y as ONE v_mad_i64_i32 whose addend is the running accumulator.  Left to itself clang starts every column's sum from
// zero (for instruction-level parallelism) and then needs a 64-bit add per column to bring the carry in: 18 extra VOP3 instructions
// per product.  The transform runs 6 waves per SIMD, so a serial chain of multiply-adds per wave costs nothing.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MZK_FS_NO_ASM)
__device__ __forceinline__ void fs_mad(int64_t& acc, int32_t x, int32_t y) {
    asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y) : "vcc");
}
__device__ __forceinline__ void fs_mad_s(int64_t& acc, int32_t x, int32_t y_sgpr) {
    asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "s"(y_sgpr) : "vcc");
}
#else
MZK_HD void fs_mad(int64_t& acc, int32_t x, int32_t y) { acc += (int64_t)x * y; }
MZK_HD void fs_mad_s(int64_t& acc, int32_t x, int32_t y) { acc += (int64_t)x * y; }
#endif
template <class X>
MZK_HD Fs<X> fs_mulc(const Fs<X>& a, const FsTw& t) {
    static_assert(X::XN == FS_N, "9 limbs of 29 bits");
    constexpr int N = FS_N;
    int32_t q[N], np[N];
#pragma unroll
    for (int i = 0; i < N; i++) np[i] = fs_neg_p_limb<X>(i);
    int64_t acc = 0;
#pragma unroll
    for (int k = N - 2; k <= 2 * N - 2; k++) {                     // columns 7 .. 16 of a x wq
#pragma unroll
        for (int i = (k - (N - 1) > 0 ? k - (N - 1) : 0); i <= (k < N - 1 ? k : N - 1); i++) fs_mad(acc, a.l[i], t.q[k - i]);
        if (k >= N) q[k - N] = (int32_t)((uint32_t)acc & XMASK);  // units of 2^261 start at column 9
        acc >>= XL;
    }
    q[N - 1] = (int32_t)acc;                                        // |q^| < 2^261: the top limb fits
    Fs<X> r;
    acc = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {                                   // low 261 bits of a w - q^ p
#pragma unroll
        for (int i = 0; i <= k; i++) {
            fs_mad(acc, a.l[i], t.w[k - i]);
            fs_mad_s(acc, q[i], np[k - i]);
        }
        if (k < N - 1) {
            r.l[k] = (int32_t)((uint32_t)acc & XMASK);
            acc >>= XL;
        }
    }
    r.l[N - 1] = ((int32_t)((uint32_t)acc << 3)) >> 3;             // |r| < 3p < 2^257: bits 232..260, sign-extended
    return r;
}

// one round of carries, all limbs at once: |limbs| < 2^31 in, limbs 0..7 in [-4, 2^29 + 4) out (limb 8 keeps its sign)
template <class X>
MZK_HD Fs<X> fs_norm(const Fs<X>& a) {
    Fs<X> r;
    r.l[0] = a.l[0] & (int32_t)XMASK;
#pragma unroll
    for (int i = 1; i < FS_N - 1; i++) r.l[i] = (a.l[i] & (int32_t)XMASK) + (a.l[i - 1] >> XL);
    r.l[FS_N - 1] = a.l[FS_N - 1] + (a.l[FS_N - 2] >> XL);
    return r;
}
// full carry propagation -> class C
template <class X>
MZK_HD Fs<X> fs_carry(const Fs<X>& a) {
    Fs<X> r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < FS_N - 1; i++) {
        const int32_t t = a.l[i] + c;                                // |a.l[i]| <= 2^31 - 2^3: no overflow
        r.l[i] = t & (int32_t)XMASK;
        c = t >> XL;
    }
    r.l[FS_N - 1] = a.l[FS_N - 1] + c;
    return r;
}
template <class X>
MZK_HD Fs<X> fs_add(const Fs<X>& a, const Fs<X>& b) {
    Fs<X> r;
#pragma unroll
    for (int i = 0; i < FS_N; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
template <class X>
MZK_HD Fs<X> fs_sub(const Fs<X>& a, const Fs<X>& b) {
    Fs<X> r;
#pragma unroll
    for (int i = 0; i < FS_N; i++) r.l[i] = a.l[i] - b.l[i];
    return r;
}

// any lazy value with |limbs| < 2^31 - 2^3 and |value| < 2^261  ->  the canonical representative in [0, p), fully carried.
// Quotient by p from the top limb alone: after the carries T = l[8] = floor(v / 2^232) and 0 <= v - T 2^232 < 2^232.  With
// SRC = floor(2^272 / p), T SRC / 2^40 differs from T 2^232 / p by less than |T| 2^-40 < 2^-11 (either way, T being signed), and
// v / p exceeds T 2^232 / p by less than 2^232 / p < 2^-21; so qh = floor(T SRC / 2^40 - 2^-10) never exceeds floor(v / p) and
// is at most one below it: v - qh p lies in [0, 2p).  |qh| < 2^8, so that subtraction is one carry chain.
template <class X>
MZK_HD Fx<X> fs_canonical(const Fs<X>& a) {
    const Fs<X> c = fs_carry(a);
    const int32_t qh = (int32_t)(((int64_t)c.l[FS_N - 1] * (int64_t)X::SRC - (1ll << 30)) >> 40);
    Fx<X> r;
    int64_t acc = 0;
#pragma unroll
    for (int i = 0; i < FS_N - 1; i++) {
        acc += (int64_t)qh * (-(int32_t)X::XP[i]) + c.l[i];
        r.l[i] = (uint32_t)acc & XMASK;
        acc >>= XL;
    }
    acc += (int64_t)qh * (-(int32_t)X::XP[FS_N - 1]) + c.l[FS_N - 1];
    r.l[FS_N - 1] = (uint32_t)acc;                                   // value now in [0, 2p)
    return fx_cond_sub_p(r);
}

// boundary image (8 little-endian words, any value < 2^256) -> class C (non-negative)
template <class X>
MZK_HD Fs<X> fs_unpack(const uint32_t* w) {
    const Fx<X> u = fx_unpack<X>(w);
    Fs<X> r;
#pragma unroll
    for (int i = 0; i < FS_N; i++) r.l[i] = (int32_t)u.l[i];
    return r;
}

// plan-time construction of a constant operand from its boundary Montgomery image a = w R:
//   w   = a / R
//   wq  = floor(w 2^261 / p) = (w 2^261 - rho) / p with rho = w 2^261 mod p; the division is exact and wq < 2^261, so
//         wq = (-rho) * p^-1 mod 2^261 -- one field product for rho and one truncated 9 x 9 limb product, no long division.
template <class X>
inline FsTw fs_make_tw(const Fp<X>& a) {
    static_assert(X::XN == FS_N && X::N == 8, "256-bit scalar fields");
    const Fp<X> w = from_mont(a);
    Fp<X> rm;
    for (int i = 0; i < 8; i++) rm.l[i] = X::XRM[i];
    const Fp<X> rho = from_mont(a * rm);                             // w 2^261 mod p, canonical
    const Fx<X> wl = fx_unpack<X>(w.l), rl = fx_unpack<X>(rho.l);
    uint32_t neg[FS_N];                                              // (2^261 - rho) mod 2^261
    uint32_t borrow = 0;
    for (int i = 0; i < FS_N; i++) {
        const uint32_t t = 0u - rl.l[i] - borrow;
        neg[i] = t & XMASK;
        borrow = (rl.l[i] + borrow) != 0 ? 1u : 0u;
    }
    FsTw t;
    uint64_t acc = 0;
    for (int k = 0; k < FS_N; k++) {                                 // low 261 bits of neg * p^-1
        for (int i = 0; i <= k; i++) acc += (uint64_t)neg[i] * X::XPINV[k - i];
        t.q[k] = (int32_t)((uint32_t)acc & XMASK);
        acc >>= XL;
    }
    for (int i = 0; i < FS_N; i++) t.w[i] = (int32_t)wl.l[i];
    return t;
}

}  // namespace mzk
