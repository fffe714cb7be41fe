// plonk.cuh -- TurboPlonk quotient evaluation on the device.
//
// Replaces the m-point closure of Prover::compute_quotient_polynomial
// (plonk/src/proof_system/prover.rs:605-659) with its helpers
// compute_quotient_circuit_contribution (:677-709) and
// compute_quotient_copy_constraint_contribution (:719-759), one instance, no Plookup.
// Everything the closure needs that does not depend on the witness is resident in HBM per proving
// key: the coset evaluations of the 13 selectors and W sigmas (the reference recomputes these 18
// size-8n FFTs in every proof, prover.rs:552-558), the evaluation points x_i = g*w_m^i and the
// inverses 1/(n (x_i - 1)) that the reference obtains by one field inversion per point (:755-756).
// Selector order: q_lc[4], q_mul[2], q_hash[4], q_o, q_c, q_ecc (relation/src/constants.rs:18-22).
#pragma once
#include <hip/hip_runtime.h>

#include "fp.cuh"
#include "fp_inv.cuh"
#include "fx.cuh"
#include "poly.cuh"

namespace mzk {

constexpr int PLK_THREADS = 256;
constexpr int PLK_SELECTORS = 13;
constexpr int PLK_WIRES = 5;
constexpr int PLK_MAX_WIRES = 6;              // UltraPlonk adds the range/lookup wire (relation/src/constants.rs)
constexpr int PLK_RATIO = 8;                 // m / n for TurboPlonk and UltraPlonk

struct QuotientArgs {
    const uint32_t* sel;      // [13 (14)][m] coset evaluations; UltraPlonk: q_lookup last
    const uint32_t* sig;      // [W][m]
    const uint32_t* wire;     // [W][m]
    const uint32_t* z;        // [m]
    const uint32_t* pi;       // [m]
    const uint32_t* xs;       // [m]  x_i = g * w_m^i
    const uint32_t* inv_den;  // [m]  1 / (n * (x_i - 1))
    uint32_t* out;            // [m]
    unsigned long long m;     // points per stream in this launch: 8n (whole quotient domain) or n (one residue class mod 8)
    unsigned long long fstride, ostride;   // elements between consecutive fixed (sel, sig, tab) / online (wire, h) polynomials
    unsigned int next_off;    // index distance of the point w_n * x: 8 in natural order, 1 inside a residue class
    int zh_class;             // >= 0: the residue class of every point of this launch; < 0: i % 8
    uint32_t k[PLK_MAX_WIRES][8];     // coset representatives k_j (Montgomery)
    uint32_t alpha[8], alpha2[8], beta[8], gamma[8];
    uint32_t zh_inv[PLK_RATIO][8];    // 1 / Z_H(x_i), period 8 in i
    // ---- UltraPlonk only (prover.rs:773-888)
    const uint32_t* tab;      // [4][m] range, key, table_dom_sep, q_dom_sep
    const uint32_t* h;        // [2][m] sorted-vector polynomials h_1, h_2
    const uint32_t* pl;       // [m] Plookup product polynomial
    const uint32_t* inv_den_n;// [m]  w^(n-1) / (n * (x_i - w^(n-1)))
    uint32_t tau[8], alpha3[8], w_inv[8];
    // bit j: selector j is the zero polynomial (found at registration): its gate term is neither read nor computed.  A circuit without
    // Rescue / elliptic-curve / multiplication gates -- the reference's bench circuit is additions only -- skips 16 + 3 + 4 of the
    // 56 products per point and 7 of the 28 operand streams.
    uint32_t sel_zero;
    // ---- several residue classes in one launch (grid.y = class; small circuits: plonk.hip quotient_chunked_run): class c reads the fixed
    // tables, x_i, 1 / (n (x_i - 1)) and writes its output c * cls_fixed words on, reads the online evaluations c * cls_online words on
    int n_cls;                // <= 1: one class, the pointers as they are
    int zh_cls[PLK_RATIO];    // zh_class of class c
    unsigned long long cls_fixed, cls_online;
};
// the pointers of class blockIdx.y, as LOCAL values: the kernels leave their argument struct untouched -- a kernel that writes to its
// by-value argument makes clang copy all of it (1 KB, with run-time-indexed arrays: zh_inv, k) into scratch memory, which cost the
// 2^20-gate quotient round 0.8 ms when the class shift was first written that way
struct QuotientPtrs {
    const uint32_t *sel, *sig, *wire, *z, *pi, *xs, *inv_den, *tab, *h, *pl, *inv_den_n;
    uint32_t* out;
    int zh_class;
};
__device__ __forceinline__ QuotientPtrs quotient_class_ptrs(const QuotientArgs& a) {
    QuotientPtrs q{a.sel, a.sig, a.wire, a.z, a.pi, a.xs, a.inv_den, a.tab, a.h, a.pl, a.inv_den_n, a.out, a.zh_class};
    if (a.n_cls > 1) {
        const unsigned c = blockIdx.y;
        const size_t f = (size_t)c * a.cls_fixed, o = (size_t)c * a.cls_online;
        q.sel += f; q.sig += f; q.xs += f; q.inv_den += f; q.out += f;
        if (q.tab) q.tab += f;
        if (q.inv_den_n) q.inv_den_n += f;
        q.wire += o; q.z += o;
        if (q.pi) q.pi += o;
        if (q.h) q.h += o;
        if (q.pl) q.pl += o;
        q.zh_class = a.zh_cls[c];
    }
    return q;
}

template <class P>
__device__ __forceinline__ Fp<P> arg_fp(const uint32_t (&a)[8]) {
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = a[i];
    return r;
}

// ---- the quotient kernels run on the reduced-radix field (fx.cuh: 9 limbs of 29 bits, lazy reduction) ------------------
// Every operand is held in the INTERNAL Montgomery form x * R', R' = 2^261 = 32 R, packed canonically in 8 words: the
// proving key's tables are stored that way, the challenges arrive that way (plonk.hip multiplies them by 32), and the coset
// NTT that produces the online evaluations folds the factor 32 into the multiplication its final pass does anyway -- as the
// inverse NTT of the result folds 1/32 (ntt.hip `scale`).  So fx_mul(a, b) = a b / R' stays in the form, at no cost.
// Bounds (H = X::HEADROOM_BITS, 6 for BLS12-381 Fr, 7 for BN254 Fr): fx_mul wants A * B <= 2^H for operands below A p, B p
// and limbs below 2^29 + 2^27; its result is class M: limbs < 2^29, value < 2 p (here < 1.35 p).  Sums are re-normalised
// (fx_norm: limbs < 2^29 + 8, value unchanged) before a limb can reach 2^32, i.e. after at most six class-M terms.
template <class X>
__device__ __forceinline__ Fx<X> arg_fx(const uint32_t (&a)[8]) { return fx_unpack<X>(a); }
template <class X>
__device__ __forceinline__ Fx<X> ldx(const uint32_t* __restrict__ p) { return fx_load_packed<X>(p); }       // canonical: A = 1, limbs < 2^29
// lazy value below 2^H p / any limb state fx_mul accepts  ->  canonical, packed
template <class X>
__device__ __forceinline__ void stx(uint32_t* __restrict__ p, const Fx<X>& v) {
    fx_store_packed<X>(p, fx_canonical(fx_mul(fx_norm(v), Fx<X>::one())));             // v * R' / R' = v, value < 1.1 p before the last subtraction
}

template <class X, bool ULTRA>
__global__ __launch_bounds__(PLK_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void plonk_quotient_kernel(QuotientArgs a) {
    using F = Fx<X>;
    constexpr int W = ULTRA ? 6 : 5;
    const QuotientPtrs q = quotient_class_ptrs(a);
    const unsigned long long i = (unsigned long long)blockIdx.x * PLK_THREADS + threadIdx.x;
    if (i >= a.m) return;
    const unsigned long long m = a.m;
    const unsigned long long inext = (i + a.next_off) % m;            // the point w_n * x_i (prover.rs:611, 627, 804)
    const unsigned long long fs = a.fstride, os = a.ostride;
    F w[W];
#pragma unroll
    for (int j = 0; j < W; j++) w[j] = ldx<X>(q.wire + ((size_t)j * os + i) * 8);
    auto sel = [&](int j) { return ldx<X>(q.sel + ((size_t)j * fs + i) * 8); };
    // ---- gate identity (prover.rs:696-708); value bounds in units of p on the right
    // Every term below is optional (sel_zero, uniform over the launch); the bounds on the right are those with all of them present.
    const uint32_t sz = a.sel_zero;
    auto has = [&](int j) { return !((sz >> j) & 1u); };
    F t = F::zero();                                                                   // q_c + pi                     2
    if (has(11)) t = sel(11);
    if (q.pi) t = fx_add(t, ldx<X>(q.pi + i * 8));                                     // (null: the public-input polynomial is zero)
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (has(j)) t = fx_add(t, fx_mul(sel(j), w[j]));                              // q_lc                         6.1, limbs < 6 * 2^29
    t = fx_norm(t);
    if (has(4) || has(5) || has(12)) {
        const F w01 = fx_mul(w[0], w[1]), w23 = fx_mul(w[2], w[3]);                   // class M
        if (has(4)) t = fx_add(t, fx_mul(sel(4), w01));                               // q_mul                        8.2
        if (has(5)) t = fx_add(t, fx_mul(sel(5), w23));
        if (has(12)) t = fx_add(t, fx_mul(sel(12), fx_mul(fx_mul(w01, w23), w[4])));  // q_ecc                        9.3, limbs < 4 * 2^29 + 8
        t = fx_norm(t);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {                                                     // q_hash * w^5                 13.5, limbs < 5 * 2^29 + 8
        if (!has(6 + j)) continue;
        const F w2 = fx_sqr(w[j]);
        t = fx_add(t, fx_mul(sel(6 + j), fx_mul(fx_sqr(w2), w[j])));
    }
    if (has(10)) t = fx_sub2(t, fx_mul(sel(10), w[4]));                               // - q_o w4 (+ 2p)               15.5
    t = fx_norm(t);
    // ---- copy constraints (prover.rs:741-758)
    const F alpha = arg_fx<X>(a.alpha), beta = arg_fx<X>(a.beta), gamma = arg_fx<X>(a.gamma);
    const F z_x = ldx<X>(q.z + i * 8);
    const F z_xw = ldx<X>(q.z + inext * 8);
    const F xb = fx_mul(ldx<X>(q.xs + i * 8), beta);                                  // class M
    F acc1 = z_x, acc2 = z_xw;
#pragma unroll
    for (int j = 0; j < W; j++) {
        const F wg = fx_add(w[j], gamma);                                             // 2, limbs < 2^30
        acc1 = fx_mul(acc1, fx_norm(fx_add(wg, fx_mul(arg_fx<X>(a.k[j]), xb))));      // factor 3.1 -> product < 1.06
        acc2 = fx_mul(acc2, fx_norm(fx_add(wg, fx_mul(ldx<X>(q.sig + ((size_t)j * fs + i) * 8), beta))));
    }
    const F t1 = fx_norm(fx_add(t, fx_mul(alpha, fx_norm(fx_sub2(acc1, acc2)))));     // acc1 - acc2 + 2p: 3.2;  t1: 16.6
    const F t2 = fx_mul(arg_fx<X>(a.alpha2), fx_mul(fx_norm(fx_sub2(z_x, F::one())), ldx<X>(q.inv_den + i * 8)));   // z - 1 (+ 2p): 3
    const F zh = arg_fx<X>(a.zh_inv[q.zh_class >= 0 ? q.zh_class : (int)(i % PLK_RATIO)]);
    stx<X>(q.out + i * 8, fx_add(fx_mul(t1, zh), t2));                                // 16.6 / 2^H + 1 + 1.05 < 2.4         prover.rs:657
}

// UltraPlonk, second launch: out[i] += t_lookup_1 * zh_inv + t_lookup_2 (compute_quotient_plookup_contribution,
// prover.rs:773-888).  Kept apart from the gate / copy-constraint kernel: together the 45 operand streams need > 320
// VGPRs (one wave per SIMD); field addition is exact, so the split changes nothing in the result.
template <class X>
__global__ __launch_bounds__(PLK_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void plonk_quotient_lookup_kernel(QuotientArgs a) {
    using F = Fx<X>;
    const QuotientPtrs q = quotient_class_ptrs(a);
    const unsigned long long i = (unsigned long long)blockIdx.x * PLK_THREADS + threadIdx.x;
    if (i >= a.m) return;
    const unsigned long long m = a.m;
    const unsigned long long inext = (i + a.next_off) % m;
    const unsigned long long fs = a.fstride, os = a.ostride;
    auto wire = [&](int j, unsigned long long at) { return ldx<X>(q.wire + ((size_t)j * os + at) * 8); };
    auto tab = [&](int j, unsigned long long at) { return ldx<X>(q.tab + ((size_t)j * fs + at) * 8); };
    const F alpha = arg_fx<X>(a.alpha), beta = arg_fx<X>(a.beta), gamma = arg_fx<X>(a.gamma);
    const F tau = arg_fx<X>(a.tau), alpha3 = arg_fx<X>(a.alpha3);
    // first + q tau (ds + tau (a0 + tau (a1 + tau a2))): every sum has two terms (value <= 2.1, limbs < 2^30): no norm needed
    // except for the multiplicand rule (limbs < 2^29 + 2^27): normalise the sums that are multiplied
    auto merged = [&](const F& first, const F& qtau, const F& ds, const F& a0, const F& a1, const F& a2) {
        const F in3 = fx_norm(fx_add(a1, fx_mul(tau, a2)));
        const F in2 = fx_norm(fx_add(a0, fx_mul(tau, in3)));
        const F in1 = fx_norm(fx_add(ds, fx_mul(tau, in2)));
        return fx_add(first, fx_mul(qtau, in1));                                      // 2.1, limbs < 2^30
    };
    const F qtau = fx_mul(ldx<X>(q.sel + ((size_t)13 * fs + i) * 8), tau);
    const F qtau_next = fx_mul(ldx<X>(q.sel + ((size_t)13 * fs + inext) * 8), tau);
    const F mt = merged(tab(0, i), qtau, tab(2, i), tab(1, i), wire(3, i), wire(4, i));
    const F mt_next = merged(tab(0, inext), qtau_next, tab(2, inext), tab(1, inext), wire(3, inext), wire(4, inext));
    const F ml = merged(wire(5, i), qtau, tab(3, i), wire(0, i), wire(1, i), wire(2, i));
    const F h1 = ldx<X>(q.h + i * 8), h1n = ldx<X>(q.h + inext * 8);
    const F h2 = ldx<X>(q.h + (os + i) * 8), h2n = ldx<X>(q.h + (os + inext) * 8);
    const F p = ldx<X>(q.pl + i * 8), pn = ldx<X>(q.pl + inext * 8);
    const F lag_n = ldx<X>(q.inv_den_n + i * 8), lag_1 = ldx<X>(q.inv_den + i * 8);
    const F pm1 = fx_norm(fx_sub2(p, F::one()));                                      // p - 1 (+ 2p): 3
    // result_2 = alpha^3 (term_h + alpha (term_p1 + alpha term_p2))
    const F inner = fx_norm(fx_add(fx_mul(pm1, lag_1), fx_mul(alpha, fx_mul(pm1, lag_n))));            // 2.2
    const F t2 = fx_mul(alpha3, fx_norm(fx_add(fx_mul(fx_norm(fx_sub2(h1, h2n)), lag_n), fx_mul(alpha, inner))));   // (h1 - h2n + 2p) <= 3
    const F b1 = fx_add(beta, F::one());                                              // 2, limbs < 2^30
    const F g1 = fx_mul(gamma, fx_norm(b1));                                          // class M
    // left = p (1 + beta) (gamma + ml) (g1 + mt + beta mt_next)
    F left = fx_mul(p, fx_norm(b1));
    left = fx_mul(left, fx_norm(fx_add(gamma, ml)));                                  // factor 3.1
    left = fx_mul(left, fx_norm(fx_add(fx_add(g1, mt), fx_mul(beta, fx_norm(mt_next)))));              // factor 1.1 + 2.1 + 1.1 = 4.3
    // right = p(wX) (g1 + h1 + beta h1(wX)) (g1 + h2 + beta h2(wX))
    F right = fx_mul(pn, fx_norm(fx_add(fx_add(g1, h1), fx_mul(beta, h1n))));         // factor 3.2
    right = fx_mul(right, fx_norm(fx_add(fx_add(g1, h2), fx_mul(beta, h2n))));
    const F xm = fx_norm(fx_sub2(ldx<X>(q.xs + i * 8), arg_fx<X>(a.w_inv)));          // x - w^-1 (+ 2p): 3
    const F term3 = fx_mul(xm, fx_norm(fx_sub2(left, right)));                        // (3.2) * 3 / 2^H + 1
    const F t1 = fx_mul(fx_sqr(alpha3), term3);
    const F zh = arg_fx<X>(a.zh_inv[q.zh_class >= 0 ? q.zh_class : (int)(i % PLK_RATIO)]);
    const F prev = ldx<X>(q.out + i * 8);
    stx<X>(q.out + i * 8, fx_add(fx_add(prev, t2), fx_mul(t1, zh)));                  // 1 + 1.1 + 1.1
}

// xs[i] = g * w^i and inv_den[i] = 1/(n (xs[i] - 1)), 16 points per thread with one shared inversion
template <class P>
__global__ __launch_bounds__(PLK_THREADS) void plonk_domain_tables_kernel(const uint32_t* __restrict__ w_mont, const uint32_t* __restrict__ n_mont,
                                                                           unsigned long long m, uint32_t* __restrict__ xs, uint32_t* __restrict__ inv_den) {
    using F = Fp<P>;
    constexpr int B = 16;
    const unsigned long long t = (unsigned long long)blockIdx.x * PLK_THREADS + threadIdx.x;
    const unsigned long long start = t * B;
    if (start >= m) return;
    const F w = load_fp<P>(w_mont), nf = load_fp<P>(n_mont), g = F::from_const(P::GENERATOR);
    F x = pow_u64(w, start) * g;
    F pref[B];
    F run = F::one();
    const int cnt = (int)(start + B <= m ? B : m - start);
    for (int j = 0; j < cnt; j++) {
        store_fp<P>(xs + (start + j) * 8, x);
        const F d = nf * (x - F::one());
        pref[j] = run;                         // product of the denominators before j
        run = run * d;
        store_fp<P>(inv_den + (start + j) * 8, d);     // parked; overwritten below
        x = x * w;
    }
    F inv_run = inv(run);
    for (int j = cnt - 1; j >= 0; j--) {
        const F d = load_fp<P>(inv_den + (start + j) * 8);
        store_fp<P>(inv_den + (start + j) * 8, inv_run * pref[j]);
        inv_run = inv_run * d;
    }
}

// ---- coset-chunked quotient -----------------------------------------------------------------------
// The 8n-point coset g*H_8n splits into 8 residue classes mod 8: class k is the coset h_k * H_n with h_k = g * w_8n^k.
// A polynomial p of degree < 2n evaluates on class k as the size-n coset NTT (offset h_k) of p mod (X^n - h_k^n):
//   folded[j] = p[j] + h_k^n * p[n + j].
template <class P>
__global__ __launch_bounds__(PLK_THREADS) void plonk_fold_kernel(const uint32_t* __restrict__ src, unsigned long long src_stride, unsigned long long in_len,
                                                                  unsigned long long n, int rows, FrArg c_mont,
                                                                  uint32_t* __restrict__ dst) {
    using F = Fp<P>;
    const unsigned long long t = (unsigned long long)blockIdx.x * PLK_THREADS + threadIdx.x;
    if (t >= n * (unsigned long long)rows) return;
    const unsigned long long row = t / n, j = t % n;
    const uint32_t* p = src + row * src_stride * 8;
    F v = j < in_len ? load_fp<P>(p + j * 8) : F::zero();
    if (n + j < in_len) v = v + fr_arg<P>(c_mont) * load_fp<P>(p + (n + j) * 8);
    store_fp<P>(dst + t * 8, v);
}

// The same fold for polynomials of at most n + 4 coefficients, where only the first in_len - n <= 4 coefficients change: the size-n coset
// NTT then reads p's first n coefficients in place and takes elements 0..3 from here (ntt_fx.cuh, NttxPassArgs::patch).
//   patch[row][j] = p[j] + h_k^n * p[n + j],  j < 4
template <class P>
__global__ void plonk_fold_patch_kernel(const uint32_t* __restrict__ src, unsigned long long src_stride, unsigned long long in_len, unsigned long long n, int rows,
                                        FrArg c_mont, uint32_t* __restrict__ patch) {
    using F = Fp<P>;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * 4) return;
    const unsigned long long row = t / 4, j = t % 4;
    const uint32_t* p = src + row * src_stride * 8;
    F v = (j < in_len && j < n) ? load_fp<P>(p + j * 8) : F::zero();
    if (n + j < in_len) v = v + fr_arg<P>(c_mont) * load_fp<P>(p + (n + j) * 8);
    store_fp<P>(patch + (size_t)t * 8, v);
}
// ... for several classes at once (grid.y = class, patches class-major)
struct FoldClasses { FrArg c[PLK_RATIO]; };
template <class P>
__global__ void plonk_fold_patch_classes_kernel(const uint32_t* __restrict__ src, unsigned long long src_stride, unsigned long long in_len, unsigned long long n, int rows,
                                                FoldClasses cs, uint32_t* __restrict__ patch) {
    using F = Fp<P>;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * 4) return;
    const unsigned long long row = t / 4, j = t % 4;
    const uint32_t* p = src + row * src_stride * 8;
    F v = (j < in_len && j < n) ? load_fp<P>(p + j * 8) : F::zero();
    if (n + j < in_len) v = v + fr_arg<P>(cs.c[blockIdx.y]) * load_fp<P>(p + (n + j) * 8);
    store_fp<P>(patch + ((size_t)blockIdx.y * rows * 4 + t) * 8, v);
}

// r_k = t mod (X^n - c_k), c_k = g^n w_8^k, for s of the 8 classes k  ->  the s coefficient slabs T_q of t = sum_{q<s} X^(qn) T_q:
//   r_k[j] = sum_q c_k^q T_q[j]   =>   T_q[j] = sum_k (V^-1)[q][k] r_k[j],  V[k][q] = c_k^q  (an s x s Vandermonde system per j).
// deg t = W (n + 1) + 2 (prover.rs:916-919, 1126-1128) is below (W + 1) n, so s = W + 1 classes determine it: 6 of 8 for TurboPlonk,
// 7 of 8 for UltraPlonk; with all 8 classes V^-1 is the 8-point inverse DFT scaled by g^(-nq) / 8.
//
// One class fewer (round 3): t = sum_{q<s} X^(qn) T_q + X^(sn) T_top with s = W classes, where T_top -- the W + 3 coefficients of t at
// and above X^(Wn) -- is known without any evaluation (plonk_quotient_top_kernel below).  Then r_k[j] - c_k^s T_top[j] (j < W + 3)
// are the remainders of the degree-< sn part and the same s x s system recovers it.
struct CombineArgs {
    const uint32_t* r;         // [ncl][n] class-major
    uint32_t* out;             // [ncl * n]  (the caller zeroes the slabs above)
    unsigned long long n;
    int ncl;
    uint32_t mat[8][8][8];     // mat[q][k] = (V^-1)[q][k], Montgomery
    const uint32_t* top;       // null, or the n_top coefficients of t from X^(ncl n) on (boundary form)
    int n_top;
    uint32_t ctop[8][8];       // c_k^ncl
};
template <class P>
__global__ __launch_bounds__(PLK_THREADS) void plonk_combine_kernel(CombineArgs a) {
    using F = Fp<P>;
    const unsigned long long j = (unsigned long long)blockIdx.x * PLK_THREADS + threadIdx.x;
    if (j >= a.n) return;
    F r[8];
#pragma unroll
    for (int k = 0; k < 8; k++) r[k] = k < a.ncl ? load_fp<P>(a.r + ((size_t)k * a.n + j) * 8) : F::zero();
    if (a.top && j < (unsigned long long)a.n_top) {
        const F tj = load_fp<P>(a.top + j * 8);
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k < a.ncl) r[k] = r[k] - arg_fp<P>(a.ctop[k]) * tj;
        store_fp<P>(a.out + ((size_t)a.ncl * a.n + j) * 8, tj);
    }
#pragma unroll 1
    for (int q = 0; q < a.ncl; q++) {
        F acc = arg_fp<P>(a.mat[q][0]) * r[0];
#pragma unroll
        for (int k = 1; k < 8; k++)
            if (k < a.ncl) acc = acc + arg_fp<P>(a.mat[q][k]) * r[k];
        store_fp<P>(a.out + ((size_t)q * a.n + j) * 8, acc);
    }
}

// ---- round 2: the permutation grand product ---------------------------------------
// Replaces the serial loop of Arithmetization::compute_prod_permutation_polynomial
// (relation/src/constraint_system.rs:1197-1223), which performs one field division per gate:
//   z[0] = 1,  z[j+1] = z[j] * prod_i (w_ij + gamma + beta k_i w^j) / prod_i (w_ij + gamma + beta sigma_ij),  j < n-1.
// Here: numerators and denominators gate by gate, the n denominators inverted with one Fermat power per `chunk` of them, then a
// three-phase parallel prefix product.
constexpr int SCAN_T = 256;           // threads per scan workgroup
constexpr int SCAN_E = 8;             // elements per thread
constexpr int SCAN_BLOCK = SCAN_T * SCAN_E;

struct PermArgs {
    const uint32_t* wire;      // [W][n] wire values (evaluations on H)
    const uint32_t* sigma;     // [W][n] extended permutation values sigma_i(w^j)
    const uint32_t* omega;     // [n] w^j
    uint32_t* ratio;           // [n] out: numerator of ratio[j] for j < n-1, 1 at n-1
    uint32_t* den;             // [n] out: its denominator
    unsigned long long n;
    int W;                     // 5 (TurboPlonk) or 6 (UltraPlonk)
    uint32_t k[PLK_MAX_WIRES][8];
    uint32_t beta[8], gamma[8];
};

// one gate per thread (coalesced): prod_i (w_ij + gamma + beta k_i w^j) and prod_i (w_ij + gamma + beta sigma_ij)
template <class P>
__global__ __launch_bounds__(PLK_THREADS) void plonk_perm_terms_kernel(PermArgs a) {
    using F = Fp<P>;
    const unsigned long long j = (unsigned long long)blockIdx.x * PLK_THREADS + threadIdx.x;
    if (j >= a.n) return;
    F nu = F::one(), de = F::one();
    if (j + 1 < a.n) {
        const F beta = arg_fp<P>(a.beta), gamma = arg_fp<P>(a.gamma);
        const F bw = beta * load_fp<P>(a.omega + j * 8);
#pragma unroll 1
        for (int i = 0; i < a.W; i++) {
            const F wg = load_fp<P>(a.wire + ((size_t)i * a.n + j) * 8) + gamma;
            nu = nu * (wg + bw * arg_fp<P>(a.k[i]));
            de = de * (wg + beta * load_fp<P>(a.sigma + ((size_t)i * a.n + j) * 8));
        }
    }
    store_fp<P>(a.ratio + j * 8, nu);
    store_fp<P>(a.den + j * 8, de);
}

// io[j] = io[j] / den[j] for all j < n.  Thread t of T owns the elements t, t + T, t + 2T, .. (coalesced across the wave) and
// inverts their product once (Montgomery's trick; the prefix products are parked in `pref`): 3 products per element and one
// inversion (fp_inv.cuh: ~70 product-equivalents; the launch lasts about as long as it on a lone wave) per n / T elements.  The host picks T so that the chip has a wave per SIMD
// before chunks grow.  A zero denominator -- probability ~ n / r; the reference would panic on 1 / 0 -- zeroes its chunk.
template <class P, class X>
__global__ __launch_bounds__(PLK_THREADS) void fr_batch_div_kernel(uint32_t* __restrict__ io, const uint32_t* __restrict__ den, unsigned long long n,
                                                                   unsigned long long T, uint32_t* __restrict__ pref) {
    using F = Fp<P>;
    const unsigned long long t = (unsigned long long)blockIdx.x * PLK_THREADS + threadIdx.x;
    if (t >= T) return;
    F run = F::one();
    unsigned long long j = t;
#pragma unroll 1
    for (; j < n; j += T) {
        store_fp<P>(pref + j * 8, run);
        run = run * load_fp<P>(den + j * 8);
    }
    F inv_run = inv_safegcd(run);                               // (fp_inv.cuh: division steps, ~17 K instructions; round 4's Fermat power on the reduced-radix product took ~74 K)
#pragma unroll 1
    for (j -= T;; j -= T) {                                     // the elements again, last to first (j ends at t)
        const F x = inv_run * load_fp<P>(pref + j * 8);
        inv_run = inv_run * load_fp<P>(den + j * 8);
        store_fp<P>(io + j * 8, load_fp<P>(io + j * 8) * x);
        if (j < T) break;
    }
}
// threads of fr_batch_div_kernel for n elements: chunks of 4 until every SIMD has a wave (2^16 threads), then longer chunks up to 64
inline unsigned long long batch_div_threads(unsigned long long n) {
    unsigned long long chunk = n >> 16;
    if (chunk < 4) chunk = 4;
    if (chunk > 64) chunk = 64;
    const unsigned long long T = (n + chunk - 1) / chunk;
    return T ? T : 1;
}

// phase 1: inclusive products inside each 2048-element block (in place), block total to totals[block]
template <class P>
__global__ __launch_bounds__(SCAN_T) void fr_scan_mul_block_kernel(uint32_t* __restrict__ data, unsigned long long n, uint32_t* __restrict__ totals) {
    using F = Fp<P>;
    __shared__ uint4 sh[2 * SCAN_T];
    const unsigned long long base = (unsigned long long)blockIdx.x * SCAN_BLOCK + (unsigned long long)threadIdx.x * SCAN_E;
    F v[SCAN_E];
    F run = F::one();
#pragma unroll
    for (int q = 0; q < SCAN_E; q++) {
        v[q] = base + q < n ? load_fp<P>(data + (base + q) * 8) : F::one();
        run = run * v[q];
        v[q] = run;
    }
    // Hillis-Steele over the SCAN_T thread totals
    auto put = [&](int i, const F& x) { sh[2 * i] = make_uint4(x.l[0], x.l[1], x.l[2], x.l[3]); sh[2 * i + 1] = make_uint4(x.l[4], x.l[5], x.l[6], x.l[7]); };
    auto get = [&](int i) { F x; uint4 p = sh[2 * i], q = sh[2 * i + 1]; x.l[0] = p.x; x.l[1] = p.y; x.l[2] = p.z; x.l[3] = p.w; x.l[4] = q.x; x.l[5] = q.y; x.l[6] = q.z; x.l[7] = q.w; return x; };
    F incl = run;
    put(threadIdx.x, incl);
    __syncthreads();
    for (int d = 1; d < SCAN_T; d <<= 1) {
        F other = F::one();
        const bool has = (int)threadIdx.x >= d;
        if (has) other = get(threadIdx.x - d);
        __syncthreads();
        if (has) incl = incl * other;
        put(threadIdx.x, incl);
        __syncthreads();
    }
    const F excl = threadIdx.x ? get(threadIdx.x - 1) : F::one();
#pragma unroll
    for (int q = 0; q < SCAN_E; q++)
        if (base + q < n) store_fp<P>(data + (base + q) * 8, excl * v[q]);
    if (threadIdx.x == SCAN_T - 1) store_fp<P>(totals + (size_t)blockIdx.x * 8, incl);
}

// phase 2: exclusive scan of the block totals by one workgroup (n_blocks <= 64 K: loops in chunks of 1024)
template <class P>
__global__ __launch_bounds__(1024) void fr_scan_mul_totals_kernel(uint32_t* __restrict__ totals, unsigned int n_blocks) {
    using F = Fp<P>;
    __shared__ uint4 sh[2 * 1024];
    auto put = [&](int i, const F& x) { sh[2 * i] = make_uint4(x.l[0], x.l[1], x.l[2], x.l[3]); sh[2 * i + 1] = make_uint4(x.l[4], x.l[5], x.l[6], x.l[7]); };
    auto get = [&](int i) { F x; uint4 p = sh[2 * i], q = sh[2 * i + 1]; x.l[0] = p.x; x.l[1] = p.y; x.l[2] = p.z; x.l[3] = p.w; x.l[4] = q.x; x.l[5] = q.y; x.l[6] = q.z; x.l[7] = q.w; return x; };
    F carry = F::one();
    for (unsigned int c0 = 0; c0 < n_blocks; c0 += 1024) {
        const unsigned int i = c0 + threadIdx.x;
        const F mine = i < n_blocks ? load_fp<P>(totals + (size_t)i * 8) : F::one();
        F incl = mine;
        put(threadIdx.x, incl);
        __syncthreads();
        const int live = (int)min(1024u, n_blocks - c0);             // the threads beyond hold ones: a 2^15-gate product has 16 block totals, four steps instead of ten
        for (int d = 1; d < live; d <<= 1) {
            F other = F::one();
            const bool has = (int)threadIdx.x >= d;
            if (has) other = get(threadIdx.x - d);
            __syncthreads();
            if (has) incl = incl * other;
            put(threadIdx.x, incl);
            __syncthreads();
        }
        const F excl = carry * (threadIdx.x ? get(threadIdx.x - 1) : F::one());
        if (i < n_blocks) store_fp<P>(totals + (size_t)i * 8, excl);
        carry = carry * get(1023);
        __syncthreads();
    }
}

// phase 3: out[0] = 1, out[j+1] = prefix[block(j)] * incl[j]  (the exclusive product, shifted by one)
template <class P>
__global__ __launch_bounds__(PLK_THREADS) void fr_scan_mul_apply_kernel(const uint32_t* __restrict__ incl, const uint32_t* __restrict__ totals,
                                                                         unsigned long long n, uint32_t* __restrict__ out) {
    using F = Fp<P>;
    const unsigned long long j = (unsigned long long)blockIdx.x * PLK_THREADS + threadIdx.x;
    if (j >= n) return;
    if (j == 0) store_fp<P>(out, F::one());
    if (j + 1 < n) store_fp<P>(out + (j + 1) * 8, load_fp<P>(totals + (j / SCAN_BLOCK) * 8) * load_fp<P>(incl + j * 8));
}

// ---- the top W + 3 coefficients of the quotient without evaluating anything -------------------------------------------
// t Z_H = num, Z_H = X^n - 1, deg t = W (n + 1) + 2 =: D_t, deg num <= D_t + n =: D.  For an index i > D_t:  num[i] = t[i - n] - t[i] =
// t[i - n], so the K = W + 3 coefficients of t from X^(Wn) on are num[(W + 1) n + i'], i' < K (needs n > W + 2) -- the top K
// coefficients of the numerator, and the top coefficients of a product of polynomials are the product, truncated to K terms, of the
// factors' top coefficients ("reversed" power series: s_f[rho] = f[deg_bound(f) - rho]).  Which terms of the closure
// (prover.rs:605-659, 677-759) reach that window:
//   alpha z(X) prod_j (w_j + beta k_j X + gamma)  and  - alpha z(wX) prod_j (w_j + beta sigma_j + gamma):   bound (n + 2) + W (n + 1) = D
//   q_ecc w_0 w_1 w_2 w_3 w_4  and  q_hash_j w_j^5:   bound (n - 1) + 5 (n + 1) = 6n + 4 = D - 3 for TurboPlonk (W = 5); below the
//       window for UltraPlonk (D - (n + 4))
//   everything else (q_lc w, q_mul w w, q_o w, pi, alpha^2 (z - 1) L_1, the Plookup terms: <= 5n + 3) lies below (W + 1) n.
// z has n + 3, the wires n + 2 coefficients (mask_polynomial, prover.rs:463-486), sigma_j and the selectors n.
struct TopArgs {
    const uint32_t* polys;     // rows: W wires, then z (coefficients, boundary form)
    unsigned long long stride, in_len, n;
    const uint32_t* top_fixed; // [W + 5][8] coefficients n-8 .. n-1 of sigma_0..W-1, q_hash_0..3, q_ecc
    int W, K, gate_shift;      // gate_shift: D - (6n + 4) if that is < K, else K (no gate term)
    uint32_t alpha[8], beta[8], gamma[8];
    uint32_t bk[PLK_MAX_WIRES][8];     // beta k_j
    uint32_t wpow[12][8];      // w_n^(2 - rho), rho < K
    uint32_t* out;             // K elements: out[i'] = t[W n + i']
};
constexpr int PLK_TOP_MAX = 9;         // K = W + 3 <= 9
constexpr int PLK_TOP_THREADS = 512;   // 8 waves: two permutation products, the q_ecc product, four q_hash products (one idle)
template <class P>
__global__ __launch_bounds__(PLK_TOP_THREADS) void plonk_quotient_top_kernel(TopArgs a) {
    using F = Fp<P>;
    constexpr int KP = PLK_TOP_MAX;
    // series: 0 P1, 1 P2, 2 E, 3..6 H_j, 7..12 F1_j, 13..18 F2_j, 19..23 W_j, 24 Q_ecc, 25..28 Q_hash_j, 29..32 hash powers
    __shared__ uint32_t ser[33][KP][8];
    __shared__ uint32_t part[8][KP][KP][8];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int K = a.K, W = a.W;
    const unsigned long long n = a.n;
    const F beta = arg_fp<P>(a.beta), gamma = arg_fp<P>(a.gamma);
    auto coef = [&](int row, long long idx) {
        return (idx >= 0 && (unsigned long long)idx < a.in_len) ? load_fp<P>(a.polys + ((size_t)row * a.stride + (size_t)idx) * 8) : F::zero();
    };
    auto fixed = [&](int row, long long idx) {                  // coefficient idx of a fixed polynomial (zero outside n-8 .. n-1)
        const long long t = idx - ((long long)n - 8);
        return (t >= 0 && t < 8) ? load_fp<P>(a.top_fixed + ((size_t)row * 8 + (size_t)t) * 8) : F::zero();
    };
    auto put = [&](int s, int rho, const F& v) {
        for (int q = 0; q < 8; q++) ser[s][rho][q] = v.l[q];
    };
    auto get = [&](int s, int rho) {
        F v;
        for (int q = 0; q < 8; q++) v.l[q] = ser[s][rho][q];
        return v;
    };
    for (int e = tid; e < 33 * KP; e += PLK_TOP_THREADS) {
        const int s = e / KP, rho = e % KP;
        F v = F::zero();
        if (rho < K) {
            const long long iw = (long long)n + 1 - rho;        // index in a wire polynomial (bound n + 1)
            const long long iq = (long long)n - 1 - rho;        // index in a fixed polynomial (bound n - 1)
            if (s == 0) v = coef(W, (long long)n + 2 - rho);
            else if (s == 1) v = coef(W, (long long)n + 2 - rho) * arg_fp<P>(a.wpow[rho]);
            else if (s >= 7 && s < 7 + W) {
                const int j = s - 7;
                v = coef(j, iw);
                if (iw == 1) v = v + arg_fp<P>(a.bk[j]);
                if (iw == 0) v = v + gamma;
            } else if (s >= 13 && s < 13 + W) {
                const int j = s - 13;
                v = coef(j, iw) + beta * fixed(j, iw);
                if (iw == 0) v = v + gamma;
            } else if (s >= 19 && s < 24) v = coef(s - 19, iw);
            else if (s == 24) v = fixed(W + 4, iq);
            else if (s >= 25 && s < 29) v = fixed(W + (s - 25), iq);
        }
        put(s, rho, v);
    }
    __syncthreads();
    const bool gate = a.gate_shift < K;
    const int n_steps = W > 5 ? W : 5;
    for (int step = 0; step < n_steps; step++) {
        int dst = -1, sa = 0, sb = 0;
        if (wave == 0 && step < W) { dst = 0; sa = 0; sb = 7 + step; }
        else if (wave == 1 && step < W) { dst = 1; sa = 1; sb = 13 + step; }
        else if (gate && wave == 2 && step < 5) { dst = 2; sa = step == 0 ? 24 : 2; sb = 19 + step; }
        else if (gate && wave >= 3 && wave < 7 && step < 4) {
            const int j = wave - 3;
            if (step == 0) { dst = 29 + j; sa = 19 + j; sb = 19 + j; }           // w^2
            else if (step == 1) { dst = 29 + j; sa = 29 + j; sb = 29 + j; }       // w^4
            else if (step == 2) { dst = 29 + j; sa = 29 + j; sb = 19 + j; }       // w^5
            else { dst = 3 + j; sa = 29 + j; sb = 25 + j; }                       // q_hash_j w^5
        }
        if (dst >= 0)
            for (int p = lane; p < K * K; p += 64) {
                const int rho = p / K, i = p % K;
                if (i <= rho) {
                    const F v = get(sa, i) * get(sb, rho - i);
                    for (int q = 0; q < 8; q++) part[wave][rho][i][q] = v.l[q];
                }
            }
        __syncthreads();
        F acc = F::zero();
        if (dst >= 0 && lane < K)
            for (int i = 0; i <= lane; i++) {
                F v;
                for (int q = 0; q < 8; q++) v.l[q] = part[wave][lane][i][q];
                acc = acc + v;
            }
        __syncthreads();
        if (dst >= 0 && lane < K) put(dst, lane, acc);
        __syncthreads();
    }
    if (tid < K) {
        const int rho = tid;
        F v = arg_fp<P>(a.alpha) * (get(0, rho) - get(1, rho));
        if (gate && rho >= a.gate_shift) {
            const int g = rho - a.gate_shift;
            v = v + get(2, g);
            for (int j = 0; j < 4; j++) v = v + get(3 + j, g);
        }
        store_fp<P>(a.out + (size_t)(K - 1 - rho) * 8, v);       // rho = D - ((W + 1) n + i')  =>  i' = W + 2 - rho = K - 1 - rho
    }
}

}  // namespace mzk
