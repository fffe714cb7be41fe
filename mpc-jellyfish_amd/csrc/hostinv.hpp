// hostinv.hpp -- host-side modular inversion by Bernstein-Yang division steps ("safegcd", eprint 2019/266, the half-delta variant),
// for every field of constants.cuh (Fr and Fq of both curves): what h64::inv (hostfp.hpp) runs since the second half of round 5.
//
// Why: every commit group of a proof ends on the host with ONE inversion in Fq (Jacobian -> affine by Montgomery's trick,
// msm.hip jac_to_affine_host) and round 5 inverts a few scalars; the Fermat power a^(p-2) they used is 381 squarings + ~190 products
// on six 64-bit limbs -- ~25 us on the GPU box's EPYC, during which the card idles.  At 2^10..2^15 gates that is 3-6 % of a proof
// (tools/host_inv_bench.cpp for the figures).
//
// Same scheme as the device's fp_inv.cuh (batches of 30 division steps on the low bits, a 2 x 2 matrix per batch applied to the
// full-width (f, g) and, modulo p, to the cofactors (d, e)), written for any limb count: NL = 9 signed 30-bit limbs for the 254 / 255-bit
// fields, 13 for BLS12-381's 381-bit Fq.  The host can stop as soon as g = 0 (no lanes wait for each other).  The number of batches
// is capped at the PROVEN bound of the plain-delta variant, floor((49 bits + 57) / 17) steps, which the half-delta variant never exceeds;
// should the cap be hit with g != 0 the caller falls back to the Fermat power (has not been seen: tests/test_host_inv.py).
// Product code: not shared with oracle/.
#pragma once
#include <cstdint>

namespace mzk {
namespace hinv {

constexpr int32_t M30 = (int32_t)((1u << 30) - 1);
template <int NL> struct Sg { int32_t v[NL]; };
struct Mat { int32_t u, v, q, r; };

template <int NL>
inline Sg<NL> from_words(const uint32_t* a, int n_words) {
    Sg<NL> r;
    for (int i = 0; i < NL; i++) {
        const int bit = 30 * i, w = bit >> 5, s = bit & 31;
        uint64_t x = w < n_words ? (uint64_t)a[w] >> s : 0;
        if (s > 2 && w + 1 < n_words) x |= (uint64_t)a[w + 1] << (32 - s);
        r.v[i] = (int32_t)((uint32_t)x & (uint32_t)M30);
    }
    return r;
}
// a value in [0, 2^(32 n_words)) with limbs in [0, 2^30) -> words
template <int NL>
inline void to_words(const Sg<NL>& a, uint32_t* out, int n_words) {
    for (int w = 0; w < n_words; w++) {
        const int bit = 32 * w, i = bit / 30, s = bit % 30;
        uint64_t x = (uint64_t)(uint32_t)a.v[i] >> s;
        if (i + 1 < NL) x |= (uint64_t)(uint32_t)a.v[i + 1] << (30 - s);
        if (60 - s < 32 && i + 2 < NL) x |= (uint64_t)(uint32_t)a.v[i + 2] << (60 - s);
        out[w] = (uint32_t)x;
    }
}
// 30 division steps on the low bits: zeta = -(delta + 1/2); returns the new zeta, t = 2^30 * (transition matrix)
inline int32_t divsteps_30(int32_t zeta, uint32_t f0, uint32_t g0, Mat& t) {
    uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
    for (int i = 0; i < 30; i++) {
        uint32_t m1 = (uint32_t)(zeta >> 31);
        const uint32_t m2 = 0u - (g & 1u);
        const uint32_t x = (f ^ m1) - m1, y = (u ^ m1) - m1, z = (v ^ m1) - m1;
        g += x & m2; q += y & m2; r += z & m2;
        m1 &= m2;
        zeta = (int32_t)((uint32_t)zeta ^ m1) - 1;
        f += g & m1; u += q & m1; v += r & m1;
        g >>= 1; u <<= 1; v <<= 1;
    }
    t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
    return zeta;
}
// (f, g) <- t (f, g) / 2^30, exactly
template <int NL>
inline void update_fg(Sg<NL>& f, Sg<NL>& g, const Mat& t) {
    const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
    int64_t cf = u * f.v[0] + v * g.v[0], cg = q * f.v[0] + r * g.v[0];
    cf >>= 30; cg >>= 30;
    for (int i = 1; i < NL; i++) {
        const int64_t fi = f.v[i], gi = g.v[i];
        cf += u * fi + v * gi;
        cg += q * fi + r * gi;
        f.v[i - 1] = (int32_t)cf & M30; cf >>= 30;
        g.v[i - 1] = (int32_t)cg & M30; cg >>= 30;
    }
    f.v[NL - 1] = (int32_t)cf;
    g.v[NL - 1] = (int32_t)cg;
}
// (d, e) <- t (d, e) / 2^30 mod p: a multiple of p is added that clears the low 30 bits; d, e stay in (-2p, p)
template <int NL>
inline void update_de(Sg<NL>& d, Sg<NL>& e, const Mat& t, const Sg<NL>& mod, uint32_t mod_inv30) {
    const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
    const int32_t sd = d.v[NL - 1] >> 31, se = e.v[NL - 1] >> 31;
    int32_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
    int64_t cd = u * d.v[0] + v * e.v[0], ce = q * d.v[0] + r * e.v[0];
    md -= (int32_t)((mod_inv30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
    me -= (int32_t)((mod_inv30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
    cd += (int64_t)mod.v[0] * md;
    ce += (int64_t)mod.v[0] * me;
    cd >>= 30; ce >>= 30;
    for (int i = 1; i < NL; i++) {
        const int64_t di = d.v[i], ei = e.v[i];
        cd += u * di + v * ei + (int64_t)mod.v[i] * md;
        ce += q * di + r * ei + (int64_t)mod.v[i] * me;
        d.v[i - 1] = (int32_t)cd & M30; cd >>= 30;
        e.v[i - 1] = (int32_t)ce & M30; ce >>= 30;
    }
    d.v[NL - 1] = (int32_t)cd;
    e.v[NL - 1] = (int32_t)ce;
}
// r in (-2p, p), negated when `sign` is negative, into [0, p)
template <int NL>
inline void normalize(Sg<NL>& r, int32_t sign, const Sg<NL>& mod) {
    int32_t add = r.v[NL - 1] >> 31;
    const int32_t ng = sign >> 31;
    for (int i = 0; i < NL; i++) r.v[i] = ((r.v[i] + (mod.v[i] & add)) ^ ng) - ng;
    for (int i = 0; i < NL - 1; i++) { r.v[i + 1] += r.v[i] >> 30; r.v[i] &= M30; }
    add = r.v[NL - 1] >> 31;
    for (int i = 0; i < NL; i++) r.v[i] += mod.v[i] & add;
    for (int i = 0; i < NL - 1; i++) { r.v[i + 1] += r.v[i] >> 30; r.v[i] &= M30; }
}

// x^-1 mod p for the INTEGER 0 < x < p given as P::N words; false: the step cap was reached with g != 0 (out untouched).
// `batches_out`: how many batches of 30 steps ran (the bench prints it).
template <class P>
inline bool inverse_words(const uint32_t* x, uint32_t* out, int* batches_out = nullptr) {
    constexpr int W = P::N, BITS = 32 * W;                                   // (the top word of every modulus here is not full: bits <= 32 W - 1)
    constexpr int NL = (BITS + 2 + 29) / 30;                                 // a sign bit and 2p: 9 limbs for 8 words, 13 for 12
    constexpr int MAX_BATCHES = ((49 * BITS + 57) / 17 + 29) / 30;
    const Sg<NL> mod = from_words<NL>(P::MOD, W);
    const uint32_t mod_inv30 = (0u - P::INV) & (uint32_t)M30;                // p^-1 mod 2^30 (P::INV = -p^-1 mod 2^32)
    Sg<NL> d, e, f = mod, g = from_words<NL>(x, W);
    for (int i = 0; i < NL; i++) { d.v[i] = 0; e.v[i] = 0; }
    e.v[0] = 1;
    int32_t zeta = -1;
    int it = 0;
    for (; it < MAX_BATCHES; it++) {
        int32_t any = 0;
        for (int i = 0; i < NL; i++) any |= g.v[i];
        if (!any) break;
        Mat t;
        zeta = divsteps_30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
        update_de<NL>(d, e, t, mod, mod_inv30);
        update_fg<NL>(f, g, t);
    }
    if (batches_out) *batches_out = it;
    int32_t any = 0;
    for (int i = 0; i < NL; i++) any |= g.v[i];
    if (any) return false;
    // g = 0: f = +-gcd = +-1 and d = +-x^-1
    normalize<NL>(d, f.v[NL - 1], mod);
    to_words<NL>(d, out, W);
    return true;
}

}  // namespace hinv
}  // namespace mzk
