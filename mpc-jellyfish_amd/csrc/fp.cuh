// fp.cuh -- Montgomery prime-field arithmetic on 32-bit limbs for gfx950 (and host).
//
// Element = P::N little-endian uint32 limbs holding a*R mod p, R = 2^(32*N), always fully
// reduced.  This is byte-identical to ark-ff's Fp<MontBackend<_,N/2>,N/2> (u64 limbs), the
// in-memory form the reference hands across the boundary, so no
// repacking is needed between the Rust side and the device.
//
// Cost model measured on MI355X (tools/valu_ubench.hip, profiles/r01_valu_ubench.txt):
// v_mad_u64_u32 ~2.1x a v_add_u32, each carry-chained add (v_add_co/v_addc) ~1.6x.  The CIOS
// inner step below compiles to one v_mad_u64_u32 plus the carry adds; all loops are fully
// unrolled so the modulus limbs become literal/SGPR operands and everything lives in VGPRs.
#pragma once
#include <cstdint>

#include "constants.cuh"

#if defined(__HIPCC__)
#define MZK_HD __host__ __device__ __forceinline__
#define MZK_D __device__ __forceinline__
#else
#define MZK_HD inline
#define MZK_D inline
#endif

namespace mzk {

template <class P>
struct Fp {
    static constexpr int N = P::N;
    uint32_t l[N];

    MZK_HD static Fp zero() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = 0;
        return r;
    }
    MZK_HD static Fp one() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = P::R1[i];
        return r;
    }
    MZK_HD static Fp from_const(const uint32_t (&c)[P::N]) {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = c[i];
        return r;
    }
    MZK_HD bool is_zero() const {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < N; i++) acc |= l[i];
        return acc == 0;
    }
    MZK_HD bool operator==(const Fp& o) const {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < N; i++) acc |= l[i] ^ o.l[i];
        return acc == 0;
    }
    MZK_HD bool operator!=(const Fp& o) const { return !(*this == o); }
};

// r = a - b, returns borrow (0/1)
template <int N>
MZK_HD uint32_t sub_limbs(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t d = (uint64_t)a[i] - b[i] - borrow;
        r[i] = (uint32_t)d;
        borrow = (d >> 32) & 1;
    }
    return (uint32_t)borrow;
}
template <int N>
MZK_HD uint32_t add_limbs(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t s = (uint64_t)a[i] + b[i] + carry;
        r[i] = (uint32_t)s;
        carry = s >> 32;
    }
    return (uint32_t)carry;
}

// conditional final subtraction: t in [0, 2p) -> [0, p)   (every modulus here has a clear top bit)
template <class P>
MZK_HD void reduce_once(uint32_t* t) {
    constexpr int N = P::N;
    uint32_t d[N];
    uint32_t borrow = sub_limbs<N>(d, t, P::MOD);
#pragma unroll
    for (int i = 0; i < N; i++) t[i] = borrow ? t[i] : d[i];
}

template <class P>
MZK_HD Fp<P> operator+(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> r;
    add_limbs<P::N>(r.l, a.l, b.l);  // no carry out: 2p < 2^(32N)
    reduce_once<P>(r.l);
    return r;
}
template <class P>
MZK_HD Fp<P> operator-(const Fp<P>& a, const Fp<P>& b) {
    constexpr int N = P::N;
    Fp<P> r;
    uint32_t borrow = sub_limbs<N>(r.l, a.l, b.l);
    uint32_t t[N];
    add_limbs<N>(t, r.l, P::MOD);
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = borrow ? t[i] : r.l[i];
    return r;
}
template <class P>
MZK_HD Fp<P> neg(const Fp<P>& a) {
    Fp<P> r;
    sub_limbs<P::N>(r.l, P::MOD, a.l);
    bool z = a.is_zero();
#pragma unroll
    for (int i = 0; i < P::N; i++) r.l[i] = z ? 0u : r.l[i];
    return r;
}
template <class P>
MZK_HD Fp<P> dbl(const Fp<P>& a) {
    return a + a;
}

// Montgomery product a*b/R mod p: CIOS, multiplication and reduction chains interleaved
// ("no-carry" variant, valid because the top bit of every modulus is clear).
template <class P>
MZK_HD Fp<P> mont_mul_cios(const Fp<P>& x, const Fp<P>& y) {
    constexpr int N = P::N;
    uint32_t t[N];
#pragma unroll
    for (int j = 0; j < N; j++) t[j] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t a = (uint64_t)x.l[0] * y.l[i] + t[0];
        uint32_t m = (uint32_t)a * P::INV;
        uint64_t c = (uint64_t)m * P::MOD[0] + (uint32_t)a;
#pragma unroll
        for (int j = 1; j < N; j++) {
            a = (uint64_t)x.l[j] * y.l[i] + t[j] + (a >> 32);
            c = (uint64_t)m * P::MOD[j] + (uint32_t)a + (c >> 32);
            t[j - 1] = (uint32_t)c;
        }
        t[N - 1] = (uint32_t)(c >> 32) + (uint32_t)(a >> 32);
    }
    reduce_once<P>(t);
    Fp<P> r;
#pragma unroll
    for (int j = 0; j < N; j++) r.l[j] = t[j];
    return r;
}
}  // namespace mzk
#if defined(__HIP_DEVICE_COMPILE__)
#include "fp_mul_gen.cuh"     // mont_mul_fips{8,12}, mont_sqr_fips{8,12}: one asm block per column
#endif
namespace mzk {

// Device: FIPS in inline asm -- 2*N^2 (v_mad_u64_u32 + v_addc_co_u32) pairs and nothing else;
// hipcc's lowering of the portable CIOS form spends ~55 % of its instructions on v_mov and
// 64-bit adds (profiles/r01_fp_bench.txt).  Host: the portable form.
template <class P>
MZK_HD Fp<P> operator*(const Fp<P>& x, const Fp<P>& y) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (P::N == 8) return mont_mul_fips8(x, y);
    else return mont_mul_fips12(x, y);
#else
    return mont_mul_cios(x, y);
#endif
}
template <class P>
MZK_HD Fp<P> sqr(const Fp<P>& x) {
    // a dedicated FIPS squaring (mont_sqr_fips*, 23 % fewer multiply-adds) measured no faster than
    // the product on gfx950: the per-column doubling eats the saving (profiles/r01_fp_bench.txt)
    return x * x;
}

// canonical integer (as limbs) <-> Montgomery
template <class P>
MZK_HD Fp<P> to_mont(const Fp<P>& a) {
    return a * Fp<P>::from_const(P::R2);
}
template <class P>
MZK_HD Fp<P> from_mont(const Fp<P>& a) {
    Fp<P> one = Fp<P>::zero();
    one.l[0] = 1;
    return a * one;
}

// a^e, e = plain little-endian 32-bit limbs
template <class P>
MZK_HD Fp<P> pow_limbs(const Fp<P>& a, const uint32_t* e, int ne) {
    Fp<P> acc = Fp<P>::one(), base = a;
    for (int i = 0; i < ne; i++)
        for (int b = 0; b < 32; b++) {
            if ((e[i] >> b) & 1) acc = acc * base;
            base = sqr(base);
        }
    return acc;
}
template <class P>
MZK_HD Fp<P> pow_u64(const Fp<P>& a, uint64_t e) {               // as many squarings as e has bits (the power tables' threads wait on this)
    Fp<P> acc = Fp<P>::one(), base = a;
    for (; e; e >>= 1) {
        if (e & 1) acc = acc * base;
        if (e > 1) base = sqr(base);
    }
    return acc;
}
// Fermat inverse a^(p-2); inv(0) = 0
template <class P>
MZK_HD Fp<P> inv(const Fp<P>& a) {
    uint32_t e[P::N], two[P::N];
    for (int i = 0; i < P::N; i++) two[i] = 0;
    two[0] = 2;
    sub_limbs<P::N>(e, P::MOD, two);
    return pow_limbs(a, e, P::N);
}
template <class P>
MZK_HD Fp<P> from_u64(uint64_t v) {
    Fp<P> a = Fp<P>::zero();
    a.l[0] = (uint32_t)v;
    a.l[1] = (uint32_t)(v >> 32);
    return to_mont(a);
}

#if defined(__HIPCC__)
// 16-byte vector load/store of a field element (N is a multiple of 4)
template <class P>
MZK_D Fp<P> load_fp(const uint32_t* __restrict__ p) {
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < P::N / 4; i++) {
        uint4 v = reinterpret_cast<const uint4*>(p)[i];
        r.l[4 * i] = v.x; r.l[4 * i + 1] = v.y; r.l[4 * i + 2] = v.z; r.l[4 * i + 3] = v.w;
    }
    return r;
}
template <class P>
MZK_D void store_fp(uint32_t* __restrict__ p, const Fp<P>& a) {
#pragma unroll
    for (int i = 0; i < P::N / 4; i++)
        reinterpret_cast<uint4*>(p)[i] = make_uint4(a.l[4 * i], a.l[4 * i + 1], a.l[4 * i + 2], a.l[4 * i + 3]);
}
#endif

}  // namespace mzk
