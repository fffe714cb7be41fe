// hostfp.hpp -- host-side Montgomery field on 64-bit limbs (same memory image as the device's
// 32-bit-limb Fp<P>).  Used for scalars only: the challenges, evaluations and linearisation coefficients of the prover rounds
// (csrc/prover.hip), the EC sum of <= 8 partial commitments, Jacobian -> affine.
// Product code: not shared with oracle/.
#pragma once
#include <cstdint>
#include <array>
#include <cstring>

#include "constants.cuh"
#include "hostinv.hpp"

namespace mzk {

template <class P>
struct Fp64 {
    static constexpr int N = P::N / 2;
    using u128 = unsigned __int128;
    uint64_t l[N];

    static uint64_t c64(const uint32_t* c, int i) { return (uint64_t)c[2 * i] | ((uint64_t)c[2 * i + 1] << 32); }
    static uint64_t mod(int i) { return c64(P::MOD, i); }
    static uint64_t inv64() {
        // -p^-1 mod 2^64 by Newton iteration from the 32-bit constant
        uint64_t p0 = mod(0), x = (uint64_t)(0u - P::INV);   // p^-1 mod 2^32
        x *= 2 - p0 * x;                                       // mod 2^64
        return 0 - x;
    }
    static Fp64 zero() { Fp64 r; std::memset(r.l, 0, sizeof r.l); return r; }
    static Fp64 one() { Fp64 r; for (int i = 0; i < N; i++) r.l[i] = c64(P::R1, i); return r; }
    static Fp64 from_words(const uint32_t* w) { Fp64 r; std::memcpy(r.l, w, sizeof r.l); return r; }
    void to_words(uint32_t* w) const { std::memcpy(w, l, sizeof l); }
    bool is_zero() const { uint64_t a = 0; for (int i = 0; i < N; i++) a |= l[i]; return a == 0; }
    bool operator==(const Fp64& o) const { return std::memcmp(l, o.l, sizeof l) == 0; }

    static bool geq_mod(const uint64_t* t) {
        for (int i = N - 1; i >= 0; i--) {
            uint64_t m = mod(i);
            if (t[i] > m) return true;
            if (t[i] < m) return false;
        }
        return true;
    }
    static void sub_mod(uint64_t* t) {
        uint64_t b = 0;
        for (int i = 0; i < N; i++) { u128 d = (u128)t[i] - mod(i) - b; t[i] = (uint64_t)d; b = (uint64_t)(d >> 64) & 1; }
    }
    friend Fp64 operator+(const Fp64& a, const Fp64& b) {
        Fp64 r; uint64_t c = 0;
        for (int i = 0; i < N; i++) { u128 s = (u128)a.l[i] + b.l[i] + c; r.l[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
        if (geq_mod(r.l)) sub_mod(r.l);
        return r;
    }
    friend Fp64 operator-(const Fp64& a, const Fp64& b) {
        Fp64 r; uint64_t bw = 0;
        for (int i = 0; i < N; i++) { u128 d = (u128)a.l[i] - b.l[i] - bw; r.l[i] = (uint64_t)d; bw = (uint64_t)(d >> 64) & 1; }
        if (bw) { uint64_t c = 0; for (int i = 0; i < N; i++) { u128 s = (u128)r.l[i] + mod(i) + c; r.l[i] = (uint64_t)s; c = (uint64_t)(s >> 64); } }
        return r;
    }
    friend Fp64 operator*(const Fp64& x, const Fp64& y) {
        static const uint64_t INV = inv64();
        uint64_t t[N];
        for (int j = 0; j < N; j++) t[j] = 0;
        for (int i = 0; i < N; i++) {
            u128 a = (u128)x.l[0] * y.l[i] + t[0];
            uint64_t m = (uint64_t)a * INV;
            u128 c = (u128)m * mod(0) + (uint64_t)a;
            for (int j = 1; j < N; j++) {
                a = (u128)x.l[j] * y.l[i] + t[j] + (uint64_t)(a >> 64);
                c = (u128)m * mod(j) + (uint64_t)a + (uint64_t)(c >> 64);
                t[j - 1] = (uint64_t)c;
            }
            t[N - 1] = (uint64_t)(c >> 64) + (uint64_t)(a >> 64);
        }
        if (geq_mod(t)) sub_mod(t);
        Fp64 r; std::memcpy(r.l, t, sizeof t);
        return r;
    }
};
template <class P> inline Fp64<P> sqr(const Fp64<P>& a) { return a * a; }
template <class P> inline Fp64<P> dbl(const Fp64<P>& a) { return a + a; }
template <class P> inline Fp64<P> neg(const Fp64<P>& a) { return Fp64<P>::zero() - a; }

// ---- helpers of the host-side prover logic (csrc/prover.hip, host/mzk_host.hpp) ----------------------------
// (their own namespace: fp.cuh has device-side functions of the same names on Fp<P>)
namespace h64 {
template <class P>
Fp64<P> from_u64(uint64_t v) {
    Fp64<P> a = Fp64<P>::zero(), r2;
    a.l[0] = v;
    for (int i = 0; i < Fp64<P>::N; i++) r2.l[i] = Fp64<P>::c64(P::R2, i);
    return a * r2;
}
template <class P>
Fp64<P> pow_u64(Fp64<P> b, uint64_t e) {
    Fp64<P> acc = Fp64<P>::one();
    for (; e; e >>= 1) {
        if (e & 1) acc = acc * b;
        b = b * b;
    }
    return acc;
}
template <class P>
Fp64<P> inv_fermat(const Fp64<P>& a) {                       // a^(p-2): the checker of the division-step inverse below, and its fallback
    uint64_t e[Fp64<P>::N];
    for (int i = 0; i < Fp64<P>::N; i++) e[i] = Fp64<P>::mod(i);
    e[0] -= 2;                                               // every modulus here is odd and > 2: no borrow beyond limb 0
    Fp64<P> acc = Fp64<P>::one(), b = a;
    for (int i = 0; i < Fp64<P>::N; i++)
        for (int k = 0; k < 64; k++) {
            if ((e[i] >> k) & 1) acc = acc * b;
            b = b * b;
        }
    return acc;
}
// 1 / a for a in Montgomery form (a R -> a^-1 R), inv(0) = 0, by division steps (hostinv.hpp: ~4 us in BLS12-381's Fq where the Fermat power
// takes ~25).  The integer inverse of the image a R is a^-1 R^-1; one Montgomery product by R^3 makes it a^-1 R.
template <class P>
Fp64<P> inv(const Fp64<P>& a) {
    if (a.is_zero()) return a;
    static const Fp64<P> r3 = [] {
        Fp64<P> r2;
        for (int i = 0; i < Fp64<P>::N; i++) r2.l[i] = Fp64<P>::c64(P::R2, i);
        return r2 * r2;
    }();
    uint32_t x[P::N], y[P::N];
    a.to_words(x);
    if (!hinv::inverse_words<P>(x, y)) return inv_fermat(a);   // (the proven step cap cannot be reached: hostinv.hpp)
    return Fp64<P>::from_words(y) * r3;
}
template <class P>
std::array<uint64_t, Fp64<P>::N> canonical(const Fp64<P>& a) {   // Montgomery -> integer
    Fp64<P> one = Fp64<P>::zero();
    one.l[0] = 1;
    Fp64<P> c = a * one;
    std::array<uint64_t, Fp64<P>::N> r;
    std::memcpy(r.data(), c.l, sizeof c.l);
    return r;
}
template <class P>
Fp64<P> root_of_unity(int log_n) {
    Fp64<P> w = Fp64<P>::from_words(P::ROOT);
    for (int i = log_n; i < P::TWO_ADICITY; i++) w = w * w;
    return w;
}
}  // namespace h64

}  // namespace mzk
