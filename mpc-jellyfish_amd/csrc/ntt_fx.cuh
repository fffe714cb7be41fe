// ntt_fx.cuh -- the multi-pass NTT of ntt.cuh on the reduced-radix field (fx.cuh): butterflies cost one
// v_mad_u64_u32 per partial product and carry-free additions.
//
// Same decomposition, tile geometry, coalescing and natural-order scatter as ntt.cuh (see there for the
// index algebra and the reference call sites it replaces: prover.rs:545-567, 672;
// constraint_system.rs:1172-1257).  What changes:
//   * data stay in the boundary's Montgomery form (x*R, R = 2^256) as integers, but are held as 9 limbs of
//     29 bits inside the kernel; twiddles are stored in R'-form (w*R', R' = 2^261), so
//     fx_mul(data, twiddle) = data*w keeps the boundary form and no conversion is ever needed;
//   * values are lazily reduced: a butterfly adds at most 2p to a value, a pass has at most 9 stages and
//     every pass ends in a multiplication (inter-pass twiddle, or the final scaling), so values stay
//     below 20p < 2^260 and every product comes out below 1.3p.  Between passes the 8-word image may hold
//     a non-canonical value < 2^256; only the last pass canonicalises;
//   * both directions run decimation-in-time (a DIF butterfly would double lazy values at every stage).
//     Forward coset scaling is folded into the stage twiddles as before (g_k = h^(S_k), free); the inverse
//     applies N^-1 * h^-j in the final pass -- N^-1 rides in the one multiplication every final pass does
//     anyway, a coset costs one more (the prover has one coset iNTT per proof, prover.rs:672).
// LDS: 9 words per element in three planes (b128, b128, b32): a 2048-element tile is 72 KiB, so two
// 512-thread workgroups share a CU (one loads/stores while the other computes); the <= 511 stage twiddles
// of a pass (24 KiB) are read through L1 -- lanes of a wave share them.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "fx.cuh"
#include "ntt.cuh"

namespace mzk {

constexpr int NTTX_THREADS = 512;
constexpr int NTTX_TW_WORDS = 12;        // a 9-limb twiddle padded to three 16-byte words in global memory

struct NttxPassArgs {
    uint32_t* in;
    uint32_t* out;
    const uint32_t* stage_tw;  // (R-1) entries of NTTX_TW_WORDS words, entry (half-1+j)
    const uint32_t* t_lo;      // inter-pass twiddles, low LB bits of the exponent (R'-form)
    const uint32_t* t_hi;      // high bits
    const uint32_t* t_full;    // middle passes: the whole table w_N^(P r s), entry (r << log_s) + s  (N / P entries: small); else null
    const uint32_t* f_lo;      // final pass: N^-1 h^-j = f_lo[j & mask] * f_hi[j >> LB]   (inverse coset), else null
    const uint32_t* f_hi;
    const uint32_t* f_one;     // final pass without coset: the single multiplier (R' mod p, or N^-1 R')
    unsigned long long in_stride, out_stride;
    unsigned long long in_len;
    int log_n, log_r, log_c, log_s, log_p;
    int log_lb;
    int is_first, is_final, n_pass;
    int skip;                  // first pass of a zero-padded input (in_len <= N >> skip): the first `skip` stages are copies
    int log_radix[NTT_MAX_PASSES];
};

template <class X>
__device__ __forceinline__ Fx<X> ldsx_load(const uint4* pa, const uint4* pb, const uint32_t* pc, int idx) {
    Fx<X> r;
    const uint4 a = pa[idx], b = pb[idx];
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = pc[idx];
    return r;
}
template <class X>
__device__ __forceinline__ void ldsx_store(uint4* pa, uint4* pb, uint32_t* pc, int idx, const Fx<X>& v) {
    pa[idx] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    pb[idx] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    pc[idx] = v.l[8];
}
template <class X>
__device__ __forceinline__ Fx<X> twx_load(const uint32_t* __restrict__ table, unsigned long long idx) {
    const uint4* src = reinterpret_cast<const uint4*>(table + idx * NTTX_TW_WORDS);
    const uint4 a = src[0], b = src[1], c = src[2];
    Fx<X> r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = c.x;
    return r;
}

// grid = (N / (R*C), batch), block = NTTX_THREADS
template <class X>
__global__ __launch_bounds__(NTTX_THREADS) void nttx_pass_kernel(NttxPassArgs a) {
    static_assert(X::XN == 9 && X::N == 8, "256-bit scalar fields: 8 boundary words, 9 limbs of 29 bits");
    extern __shared__ uint4 ldsx[];
    const int R = 1 << a.log_r, C = 1 << a.log_c, TILE = R * C;
    uint4* da = ldsx;
    uint4* db = da + TILE;
    uint32_t* dc = reinterpret_cast<uint32_t*>(db + TILE);
    const int tid = threadIdx.x;
    const unsigned long long tile = blockIdx.x;
    uint32_t* in = a.in + (unsigned long long)blockIdx.y * a.in_stride * 8;
    uint32_t* out = a.out + (unsigned long long)blockIdx.y * a.out_stride * 8;

    // ---- tile geometry (as ntt.cuh) --------------------------------------------------------------------
    unsigned long long base = 0, c0 = 0, p1 = 1, i1_0 = 0, rest = 0, rev_rest = 0;
    if (!a.is_final) {
        const unsigned long long tiles_per_blk = 1ull << (a.log_s - a.log_c);
        const unsigned long long blk = tile >> (a.log_s - a.log_c);
        c0 = (tile & (tiles_per_blk - 1)) << a.log_c;
        base = (blk << (a.log_r + a.log_s)) + c0;
    } else {
        const int log_r1 = a.n_pass > 1 ? a.log_radix[0] : 0;
        const int log_p1 = a.log_p - log_r1;
        p1 = 1ull << log_p1;
        rest = tile & (p1 - 1);
        i1_0 = (tile >> log_p1) << a.log_c;
        unsigned long long x = rest;
        for (int k = a.n_pass - 2; k >= 1; k--) {
            const unsigned long long d = x & ((1ull << a.log_radix[k]) - 1);
            x >>= a.log_radix[k];
            rev_rest = (rev_rest << a.log_radix[k]) | d;
        }
    }

    // ---- load: boundary words -> 29-bit limbs, bit-reversed rows (decimation in time) --------------------
    // Zero-padded input (the quotient round transforms polynomials of degree n + 2 on 8n points): rows r >= R >> skip are
    // zero for every tile, so in bit-reversed order each run of 2^skip positions holds one non-zero value first, and the
    // first `skip` stages -- butterflies whose multiplied operand is zero -- merely copy it across the run.
    if (a.skip > 0) {
        const int rows = R >> a.skip, reps = 1 << a.skip;
        for (int e = tid; e < rows * C; e += NTTX_THREADS) {
            const int c = e & (C - 1), r = e >> a.log_c;
            const unsigned long long g = base + ((unsigned long long)r << a.log_s) + c;
            Fx<X> v = Fx<X>::zero();
            if (g < a.in_len) v = fx_load_packed<X>(in + g * 8);
            const int pos = (int)bitrev((unsigned)r, a.log_r) * C + c;
            for (int t = 0; t < reps; t++) ldsx_store<X>(da, db, dc, pos + t * C, v);
        }
    } else
    for (int e = tid; e < TILE; e += NTTX_THREADS) {
        int r, c;
        unsigned long long g;
        if (!a.is_final) {
            c = e & (C - 1);
            r = e >> a.log_c;
            g = base + ((unsigned long long)r << a.log_s) + c;
        } else {
            r = e & (R - 1);
            c = e >> a.log_r;
            g = (((i1_0 + c) * p1 + rest) << a.log_r) + r;
        }
        Fx<X> v = Fx<X>::zero();
        if (!(a.is_first && g >= a.in_len)) v = fx_load_packed<X>(in + g * 8);
        ldsx_store<X>(da, db, dc, (int)bitrev((unsigned)r, a.log_r) * C + c, v);
    }
    __syncthreads();

    // ---- R-point transforms: one butterfly per thread per stage -------------------------------------------
    const int nbf = TILE >> 1;
    for (int s = a.skip; s < a.log_r; s++) {
        const int half = 1 << s;
        for (int bt = tid; bt < nbf; bt += NTTX_THREADS) {
            const int c = bt & (C - 1);
            const int jj = bt >> a.log_c;
            const int j = jj & (half - 1);
            const int lo_i = (((jj >> s) << (s + 1)) + j) * C + c, hi_i = lo_i + half * C;
            const Fx<X> lo = fx_norm(ldsx_load<X>(da, db, dc, lo_i));               // limbs < 2^29 + 8
            const Fx<X> hi = fx_norm(ldsx_load<X>(da, db, dc, hi_i));
            const Fx<X> t = fx_mul(hi, twx_load<X>(a.stage_tw, half - 1 + j));       // class M, value < 1.3p; the table is L1-resident
            ldsx_store<X>(da, db, dc, lo_i, fx_add(lo, t));                          // value + 1.3p, limbs < 2^30 + 8
            ldsx_store<X>(da, db, dc, hi_i, fx_sub2(lo, t));                         // value + 2p,  limbs < 2^31
        }
        __syncthreads();
    }

    // ---- store ------------------------------------------------------------------------------------------
    for (int e = tid; e < TILE; e += NTTX_THREADS) {
        const int c = e & (C - 1);
        const int r = e >> a.log_c;
        const Fx<X> v = fx_norm(ldsx_load<X>(da, db, dc, r * C + c));
        if (!a.is_final) {
            Fx<X> tw;
            if (a.t_full) {                                                          // one product instead of two
                tw = twx_load<X>(a.t_full, ((unsigned long long)r << a.log_s) + c0 + c);
            } else {
                const unsigned long long ex = ((unsigned long long)r * (c0 + c)) << a.log_p;
                tw = fx_mul(twx_load<X>(a.t_lo, ex & ((1ull << a.log_lb) - 1)), twx_load<X>(a.t_hi, ex >> a.log_lb));
            }
            const unsigned long long g = base + ((unsigned long long)r << a.log_s) + c;
            fx_store_packed<X>(out + g * 8, fx_mul(v, tw));                          // < 1.3p < 2^256: fully carried limbs
        } else {
            const unsigned long long rev = (i1_0 + c) + (rev_rest << (a.n_pass > 1 ? a.log_radix[0] : 0));
            const unsigned long long g = rev + ((unsigned long long)r << a.log_p);
            Fx<X> f;
            if (a.f_lo) f = fx_mul(twx_load<X>(a.f_lo, g & ((1ull << a.log_lb) - 1)), twx_load<X>(a.f_hi, g >> a.log_lb));
            else f = twx_load<X>(a.f_one, 0);
            fx_store_packed<X>(out + g * 8, fx_canonical(fx_mul(v, f)));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side: plan tables in R'-form
// ------------------------------------------------------------------------------------------------
struct NttxPlanHost {
    int log_n = 0, n_pass = 0, log_lb = 0;
    bool inverse = false, coset = false;
    int log_radix[NTT_MAX_PASSES] = {0, 0, 0, 0};
    std::vector<uint32_t> stage_tw[NTT_MAX_PASSES];
    std::vector<uint32_t> t_lo, t_hi, f_lo, f_hi, f_one;
    std::vector<uint32_t> t_full[NTT_MAX_PASSES];       // middle passes only
};

// boundary-form field element (x*R) -> twiddle record (x*R' as 9 limbs, padded)
template <class X>
inline void nttx_put(std::vector<uint32_t>& dst, size_t idx, const Fp<X>& a) {
    // (x*R) * (R' mod p) / R = x*R' as a canonical integer
    Fp<X> rp;
    fx_pack<X>(rp.l, Fx<X>::from_const(X::XONE));
    const Fp<X> v = a * rp;
    const Fx<X> f = fx_unpack<X>(v.l);
    for (int i = 0; i < 9; i++) dst[idx * NTTX_TW_WORDS + i] = f.l[i];
    for (int i = 9; i < NTTX_TW_WORDS; i++) dst[idx * NTTX_TW_WORDS + i] = 0;
}

template <class X>
void nttx_build_plan(NttxPlanHost& pl, int log_n, bool inverse, const uint32_t* coset_mont /* nullable */, int scale = 0) {
    using F = Fp<X>;
    pl.log_n = log_n;
    pl.inverse = inverse;
    pl.coset = coset_mont != nullptr;
    ntt_choose_radices(log_n, pl.log_radix, &pl.n_pass);
    F h = F::one();
    if (coset_mont)
        for (int i = 0; i < 8; i++) h.l[i] = coset_mont[i];
    F w = F::from_const(X::ROOT);
    for (int i = log_n; i < X::TWO_ADICITY; i++) w = sqr(w);
    const F w_dir = inverse ? inv(w) : w;
    int log_p = 0;
    for (int k = 0; k < pl.n_pass; k++) {
        const int lr = pl.log_radix[k];
        const int R = 1 << lr;
        const int log_s = log_n - log_p - lr;
        // forward: inputs pre-scaled by h^j  =>  stage twiddles W_M[j] = w_M^j g^(R/M), g = h^(S_k); inverse: plain
        const F g = inverse ? F::one() : pow_u64(h, 1ull << log_s);
        pl.stage_tw[k].assign((size_t)(R > 1 ? R - 1 : 1) * NTTX_TW_WORDS, 0);
        for (int s = 0; s < lr; s++) {
            const int half = 1 << s, M = 2 * half;
            const F wm = pow_u64(w_dir, 1ull << (log_n - (s + 1)));
            F cur = pow_u64(g, (uint64_t)(R / M));
            for (int j = 0; j < half; j++) {
                nttx_put<X>(pl.stage_tw[k], (size_t)(half - 1 + j), cur);
                cur = cur * wm;
            }
        }
        log_p += lr;
    }
    // Inter-pass twiddles of pass k: w_N^(P_k r s), r < R_k, s < S_k -- the same for every block, so the table has N / P_k
    // entries.  For the first pass that is N entries (kept two-level: t_lo, t_hi, one extra product per element); for the
    // middle passes it is N / R_1 or less (<= 3 MB), read through L2: one product per element.
    log_p = pl.log_radix[0];
    for (int k = 1; k + 1 < pl.n_pass; k++) {
        const int lr = pl.log_radix[k];
        const int log_s = log_n - log_p - lr;
        const size_t R = (size_t)1 << lr, S = (size_t)1 << log_s;
        pl.t_full[k].assign(R * S * NTTX_TW_WORDS, 0);
        const F wp = pow_u64(w_dir, 1ull << log_p);                 // w_N^P
        F row = F::one();                                           // (w_N^P)^r
        for (size_t r = 0; r < R; r++) {
            F cur = F::one();
            for (size_t sidx = 0; sidx < S; sidx++) { nttx_put<X>(pl.t_full[k], (r << log_s) + sidx, cur); cur = cur * row; }
            row = row * wp;
        }
        log_p += lr;
    }
    pl.log_lb = (log_n + 1) / 2;
    const size_t nlo = (size_t)1 << pl.log_lb, nhi = (size_t)1 << (log_n - pl.log_lb);
    pl.t_lo.assign(nlo * NTTX_TW_WORDS, 0);
    pl.t_hi.assign(nhi * NTTX_TW_WORDS, 0);
    F cur = F::one();
    for (size_t i = 0; i < nlo; i++) { nttx_put<X>(pl.t_lo, i, cur); cur = cur * w_dir; }
    const F step = cur;
    cur = F::one();
    for (size_t i = 0; i < nhi; i++) { nttx_put<X>(pl.t_hi, i, cur); cur = cur * step; }
    // final-pass multiplier(s)
    // scale: 1 = leave the output in the internal form x * R' (R' = 32 R), 2 = take an internal-form input back to x * R;
    // folded into the multiplication the final pass performs anyway (plonk.cuh)
    F ninv = inverse ? inv(from_u64<X>(1ull << log_n)) : F::one();
    if (scale == 1) ninv = ninv * from_u64<X>(32);
    if (scale == 2) ninv = ninv * inv(from_u64<X>(32));
    pl.f_one.assign(NTTX_TW_WORDS, 0);
    nttx_put<X>(pl.f_one, 0, ninv);
    if (inverse && pl.coset) {
        const F hi = inv(h);
        pl.f_lo.assign(nlo * NTTX_TW_WORDS, 0);
        pl.f_hi.assign(nhi * NTTX_TW_WORDS, 0);
        cur = F::one();
        for (size_t i = 0; i < nlo; i++) { nttx_put<X>(pl.f_lo, i, cur); cur = cur * hi; }
        const F hstep = cur;
        cur = ninv;
        for (size_t i = 0; i < nhi; i++) { nttx_put<X>(pl.f_hi, i, cur); cur = cur * hstep; }
    }
}

}  // namespace mzk
