// fp_inv.cuh -- modular inversion in a 256-bit scalar field by Bernstein-Yang division steps ("safegcd", eprint 2019/266; the
// half-delta variant with 30-bit batches that 32-bit CPUs use): 20 batches of 30 division steps on the LOW 30 bits of (f, g) -- each
// batch a 2 x 2 integer matrix with entries up to 2^30 -- applied to the full-width f, g (exact division by 2^30) and, modulo p, to
// the pair (d, e) that tracks g's and f's cofactors.  Everything is adds, shifts and 36 + 54 signed multiply-adds per batch
// (v_mad_i64_i32): ~17 K instructions per inversion where the Fermat power with 4-bit windows on the reduced-radix product (fx.cuh
// fx_inv) takes ~74 K.  The prover inverts ONCE per thread in its batched divisions (plonk.cuh fr_batch_div_kernel: the grand
// products of rounds 2 and 2.5), and the launch lasts as long as that one inversion on a lone wave.
//
// 590 division steps suffice for any odd modulus below 2^256 and any 0 <= g < f (the bound computed for this variant); 20 x 30 = 600
// are run, no early exit (lanes of a wave would wait for the slowest anyway).
// Representation: 9 signed limbs of 30 bits, value = sum v[i] 2^(30 i); f, g exact integers; d, e in (-2p, p) between batches.
#pragma once
#include "fp.cuh"

namespace mzk {

struct Sg30 { int32_t v[9]; };
struct SgMat { int32_t u, v, q, r; };
constexpr int32_t SG_M30 = (int32_t)((1u << 30) - 1);

template <class P>
MZK_HD Sg30 sg_modulus() {
    static_assert(P::N == 8, "256-bit fields");
    Sg30 m;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 30 * i, w = bit >> 5, s = bit & 31;
        uint64_t x = (uint64_t)P::MOD[w] >> s;
        if (s > 2 && w + 1 < 8) x |= (uint64_t)P::MOD[w + 1] << (32 - s);
        m.v[i] = (int32_t)((uint32_t)x & (uint32_t)SG_M30);
    }
    return m;
}
MZK_HD Sg30 sg_from_words(const uint32_t* a) {
    Sg30 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 30 * i, w = bit >> 5, s = bit & 31;
        uint64_t x = (uint64_t)a[w] >> s;
        if (s > 2 && w + 1 < 8) x |= (uint64_t)a[w + 1] << (32 - s);
        r.v[i] = (int32_t)((uint32_t)x & (uint32_t)SG_M30);
    }
    return r;
}
// a value in [0, 2^256) with limbs in [0, 2^30) -> 8 words
MZK_HD void sg_to_words(const Sg30& a, uint32_t* out) {
#pragma unroll
    for (int w = 0; w < 8; w++) {
        const int bit = 32 * w, i = bit / 30, s = bit % 30;                  // word w starts inside limb i at bit s
        uint64_t x = (uint64_t)(uint32_t)a.v[i] >> s;
        x |= (uint64_t)(uint32_t)a.v[i + 1] << (30 - s);
        if (60 - s < 32 && i + 2 < 9) x |= (uint64_t)(uint32_t)a.v[i + 2] << (60 - s);
        out[w] = (uint32_t)x;
    }
}

// 30 division steps on the low bits: zeta = -(delta + 1/2); returns the new zeta, t = 2^30 * (transition matrix)
MZK_HD int32_t sg_divsteps_30(int32_t zeta, uint32_t f0, uint32_t g0, SgMat& t) {
    uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll 1
    for (int i = 0; i < 30; i++) {
        uint32_t m1 = (uint32_t)(zeta >> 31);                                // zeta < 0
        const uint32_t m2 = 0u - (g & 1u);                                   // g odd
        const uint32_t x = (f ^ m1) - m1, y = (u ^ m1) - m1, z = (v ^ m1) - m1;      // f, u, v negated when zeta < 0
        g += x & m2; q += y & m2; r += z & m2;
        m1 &= m2;                                                            // zeta < 0 and g odd: swap
        zeta = (int32_t)((uint32_t)zeta ^ m1) - 1;                           // -zeta - 2, or zeta - 1
        f += g & m1; u += q & m1; v += r & m1;
        g >>= 1; u <<= 1; v <<= 1;
    }
    t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
    return zeta;
}
// (f, g) <- t (f, g) / 2^30, exactly
MZK_HD void sg_update_fg(Sg30& f, Sg30& g, const SgMat& t) {
    const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
    int64_t cf = u * f.v[0] + v * g.v[0], cg = q * f.v[0] + r * g.v[0];
    cf >>= 30; cg >>= 30;                                                    // (the low 30 bits are zero)
#pragma unroll
    for (int i = 1; i < 9; i++) {
        const int64_t fi = f.v[i], gi = g.v[i];
        cf += u * fi + v * gi;
        cg += q * fi + r * gi;
        f.v[i - 1] = (int32_t)cf & SG_M30; cf >>= 30;
        g.v[i - 1] = (int32_t)cg & SG_M30; cg >>= 30;
    }
    f.v[8] = (int32_t)cf;
    g.v[8] = (int32_t)cg;
}
// (d, e) <- t (d, e) / 2^30 mod p: a multiple of p is added that clears the low 30 bits; d, e stay in (-2p, p)
template <class P>
MZK_HD void sg_update_de(Sg30& d, Sg30& e, const SgMat& t, const Sg30& mod, uint32_t mod_inv30) {
    const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
    const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;                      // sign masks
    int32_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
    int64_t cd = u * d.v[0] + v * e.v[0], ce = q * d.v[0] + r * e.v[0];
    md -= (int32_t)((mod_inv30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)SG_M30);
    me -= (int32_t)((mod_inv30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)SG_M30);
    cd += (int64_t)mod.v[0] * md;
    ce += (int64_t)mod.v[0] * me;
    cd >>= 30; ce >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        const int64_t di = d.v[i], ei = e.v[i];
        cd += u * di + v * ei + (int64_t)mod.v[i] * md;
        ce += q * di + r * ei + (int64_t)mod.v[i] * me;
        d.v[i - 1] = (int32_t)cd & SG_M30; cd >>= 30;
        e.v[i - 1] = (int32_t)ce & SG_M30; ce >>= 30;
    }
    d.v[8] = (int32_t)cd;
    e.v[8] = (int32_t)ce;
}
// r in (-2p, p), negated when `sign` is negative, into [0, p)
MZK_HD void sg_normalize(Sg30& r, int32_t sign, const Sg30& mod) {
    int32_t add = r.v[8] >> 31;
    const int32_t ng = sign >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = ((r.v[i] + (mod.v[i] & add)) ^ ng) - ng;
#pragma unroll
    for (int i = 0; i < 8; i++) { r.v[i + 1] += r.v[i] >> 30; r.v[i] &= SG_M30; }
    add = r.v[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] += mod.v[i] & add;
#pragma unroll
    for (int i = 0; i < 8; i++) { r.v[i + 1] += r.v[i] >> 30; r.v[i] &= SG_M30; }
}

// x^-1 mod p for the INTEGER x in [0, p) given as 8 words; 0 -> 0
template <class P>
MZK_HD void sg_inverse_words(const uint32_t* x, uint32_t* out) {
    const Sg30 mod = sg_modulus<P>();
    const uint32_t mod_inv30 = (0u - P::INV) & (uint32_t)SG_M30;             // p^-1 mod 2^30 (P::INV = -p^-1 mod 2^32)
    Sg30 d, e, f = mod, g = sg_from_words(x);
#pragma unroll
    for (int i = 0; i < 9; i++) { d.v[i] = 0; e.v[i] = 0; }
    e.v[0] = 1;
    int32_t zeta = -1;
#pragma unroll 1
    for (int it = 0; it < 20; it++) {
        SgMat t;
        zeta = sg_divsteps_30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
        sg_update_de<P>(d, e, t, mod, mod_inv30);
        sg_update_fg(f, g, t);
    }
    sg_normalize(d, f.v[8], mod);                                            // g = 0, f = +-1: d = +-x^-1
    sg_to_words(d, out);
}

// 1 / a for a in Montgomery form (a R -> a^-1 R), inv(0) = 0: the integer inverse is a^-1 R^-1, times R^2 is one Montgomery product by R^3
template <class P>
MZK_HD Fp<P> inv_safegcd(const Fp<P>& a) {
    Fp<P> y;
    sg_inverse_words<P>(a.l, y.l);
    const Fp<P> r2 = Fp<P>::from_const(P::R2);
    return y * (r2 * r2);
}

}  // namespace mzk
