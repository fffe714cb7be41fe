// poly.cuh -- dense-polynomial primitives of prover rounds 4 and 5, on device-resident coefficient
// vectors ("mul_poly/axpy chains, Horner evals, division by (X - z)"):
//   evaluate        DensePolynomial::evaluate            plonk/src/proof_system/prover.rs:216-235 (a14)
//   lincomb         mul_poly + poly additions            prover.rs:302-358, 1115-1122, 497-501   (a15, a16)
//   div_linear      &batch_poly / &(X - z)               prover.rs:504-506                        (a16)
// The reference evaluates with a serial Horner per polynomial and divides with a serial synthetic
// division; here both are a strided Horner / a prefix sum so that every access is coalesced.
#pragma once
#include <hip/hip_runtime.h>

#include "fp.cuh"
#include "fx.cuh"

namespace mzk {

constexpr int POLY_THREADS = 256;
constexpr int POLY_EVAL_T = 16384;           // least number of threads (= stride) per polynomial in the strided Horner
constexpr int POLY_EVAL_T_MAX = 1 << 18;     // ... grown while a thread's chain of dependent products would exceed 32 (poly.hip eval_run)
constexpr int POLY_MAX_TERMS = 32;

template <class P>
__device__ __forceinline__ Fp<P> block_sum(Fp<P> v, uint4* sh) {
    // tree sum over the 256 threads of a workgroup; result valid in thread 0
    auto put = [&](int i, const Fp<P>& x) { sh[2 * i] = make_uint4(x.l[0], x.l[1], x.l[2], x.l[3]); sh[2 * i + 1] = make_uint4(x.l[4], x.l[5], x.l[6], x.l[7]); };
    auto get = [&](int i) { Fp<P> x; uint4 p = sh[2 * i], q = sh[2 * i + 1]; x.l[0] = p.x; x.l[1] = p.y; x.l[2] = p.z; x.l[3] = p.w; x.l[4] = q.x; x.l[5] = q.y; x.l[6] = q.z; x.l[7] = q.w; return x; };
    put(threadIdx.x, v);
    __syncthreads();
    for (int d = POLY_THREADS / 2; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) { v = v + get(threadIdx.x + d); put(threadIdx.x, v); }
        __syncthreads();
    }
    return v;
}

// a field element as a kernel argument (no host-to-device copy in front of the launch)
struct FrArg { uint32_t l[8]; };
template <class P>
__host__ __device__ inline Fp<P> fr_arg(const FrArg& a) {
    Fp<P> r;
    for (int i = 0; i < 8; i++) r.l[i] = a.l[i];
    return r;
}
template <class P>
inline FrArg to_fr_arg(const Fp<P>& v) {
    FrArg a;
    for (int i = 0; i < 8; i++) a.l[i] = v.l[i];
    return a;
}

// partial[poly][block] = sum over the block's threads t of x^t * ( sum_k c[t + k T] y^k ),  y = x^T.
// grid = (T / 256, batch)
template <class P>
__global__ __launch_bounds__(POLY_THREADS) void poly_eval_partial_kernel(const uint32_t* __restrict__ coeffs, unsigned long long stride, unsigned long long len,
                                                                          const uint32_t* __restrict__ xpow /* [T] */, FrArg y_mont,
                                                                          unsigned long long T, uint32_t* __restrict__ partial) {
    using F = Fp<P>;
    __shared__ uint4 sh[2 * POLY_THREADS];
    const unsigned long long t = (unsigned long long)blockIdx.x * POLY_THREADS + threadIdx.x;
    const uint32_t* c = coeffs + (unsigned long long)blockIdx.y * stride * 8;
    const F y = fr_arg<P>(y_mont);
    F acc = F::zero();
    if (t < len) {
        const unsigned long long kmax = (len - 1 - t) / T;                       // highest k with t + k T < len
        acc = load_fp<P>(c + (t + kmax * T) * 8);
        for (unsigned long long k = kmax; k-- > 0;) acc = acc * y + load_fp<P>(c + (t + k * T) * 8);
        acc = acc * load_fp<P>(xpow + t * 8);
    }
    const F s = block_sum<P>(acc, sh);
    if (threadIdx.x == 0) store_fp<P>(partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8, s);
}
// out[poly] = sum of that polynomial's partials; grid = batch, 256 threads
template <class P>
__global__ __launch_bounds__(POLY_THREADS) void poly_eval_final_kernel(const uint32_t* __restrict__ partial, int per_poly, uint32_t* __restrict__ out) {
    using F = Fp<P>;
    __shared__ uint4 sh[2 * POLY_THREADS];
    F acc = F::zero();
    for (int i = threadIdx.x; i < per_poly; i += POLY_THREADS) acc = acc + load_fp<P>(partial + ((size_t)blockIdx.x * per_poly + i) * 8);
    const F s = block_sum<P>(acc, sh);
    if (threadIdx.x == 0) store_fp<P>(out + (size_t)blockIdx.x * 8, s);
}

// out[j] = w^j (Montgomery) for j < n, 16 per thread, for up to two bases in one launch (grid.y = base).
// The host -- where a field product costs 30 ns -- sends w^(d 4^L), d = 1..3, for every base-4 digit position L of an exponent
// as kernel arguments (PowTable: 1.3 KB per base), so a thread's first power w^(16 t) is a product of at most 12 of them (at most 6
// up to 2^16 elements) and its 16 outputs are that times w^(4 a) times w^b: a chain of 8 dependent products where round 4's form
// (square-and-multiply per thread, then 16 products in sequence) had up to 60 -- the table of a 2^15-coefficient polynomial 29 -> 8 us,
// and a proof launches six of them (rounds 4 and 5), now three.
constexpr int POW_LEVELS = 14;                 // exponents below 4^14 = 2^28
struct PowTable { uint32_t p[POW_LEVELS][3][8]; };
struct PowArgs {
    PowTable t[2];
    unsigned long long n;
    uint32_t* out[2];
};
template <class P>
__global__ __launch_bounds__(POLY_THREADS) void fr_powers_tab_kernel(PowArgs a) {
    using F = Fp<P>;
    const unsigned long long start = ((unsigned long long)blockIdx.x * POLY_THREADS + threadIdx.x) * 16;
    if (start >= a.n) return;
    const PowTable& tb = a.t[blockIdx.y];
    uint32_t* __restrict__ out = a.out[blockIdx.y];
    auto entry = [&](int L, unsigned d) {
        F r;
#pragma unroll
        for (int q = 0; q < 8; q++) r.l[q] = tb.p[L][d - 1][q];
        return r;
    };
    F x = F::one();
    bool have = false;
#pragma unroll 1
    for (int L = 2; L < POW_LEVELS; L++) {
        const unsigned d = (unsigned)(start >> (2 * L)) & 3u;
        if (d == 0) continue;
        x = have ? x * entry(L, d) : entry(L, d);
        have = true;
    }
#pragma unroll 1
    for (unsigned hi = 0; hi < 4; hi++) {
        const F xa = hi ? x * entry(1, hi) : x;
#pragma unroll 1
        for (unsigned lo = 0; lo < 4; lo++) {
            const unsigned long long j = start + 4 * hi + lo;
            if (j < a.n) store_fp<P>(out + j * 8, lo ? xa * entry(0, lo) : xa);
        }
    }
}
// the launch: count = 1 or 2 bases, n powers each
template <class P>
inline PowTable make_pow_table(const Fp<P>& w) {
    PowTable t;
    Fp<P> e = w;
    for (int L = 0; L < POW_LEVELS; L++) {
        const Fp<P> e2 = e * e, e3 = e2 * e;
        for (int q = 0; q < 8; q++) { t.p[L][0][q] = e.l[q]; t.p[L][1][q] = e2.l[q]; t.p[L][2][q] = e3.l[q]; }
        e = e2 * e2;
    }
    return t;
}
template <class P>
inline void launch_powers(hipStream_t st, const Fp<P>* w, int count, unsigned long long n, uint32_t* const* out) {
    if (n == 0 || count <= 0) return;
    PowArgs a;
    for (int q = 0; q < 2; q++) { a.t[q] = make_pow_table<P>(w[q < count ? q : 0]); a.out[q] = out[q < count ? q : 0]; }
    a.n = n;
    hipLaunchKernelGGL((fr_powers_tab_kernel<P>), dim3((unsigned)(((n + 15) / 16 + POLY_THREADS - 1) / POLY_THREADS), (unsigned)count), dim3(POLY_THREADS), 0, st, a);
}

struct LincombArgs {
    const uint32_t* poly[POLY_MAX_TERMS];
    unsigned long long len[POLY_MAX_TERMS];
    uint32_t scalar[POLY_MAX_TERMS][8];
    int n_terms;
    unsigned long long out_len;
    uint32_t* out;
};
// out[j] = sum_k scalar_k * poly_k[j]   (poly_k[j] = 0 for j >= len_k); out may alias one of the inputs
template <class P>
__global__ __launch_bounds__(POLY_THREADS) void poly_lincomb_kernel(LincombArgs a) {
    using F = Fp<P>;
    const unsigned long long j = (unsigned long long)blockIdx.x * POLY_THREADS + threadIdx.x;
    if (j >= a.out_len) return;
    F acc = F::zero();
    for (int k = 0; k < a.n_terms; k++) {
        if (j < a.len[k]) {
            F s;
#pragma unroll
            for (int q = 0; q < 8; q++) s.l[q] = a.scalar[k][q];
            acc = acc + s * load_fp<P>(a.poly[k] + j * 8);
        }
    }
    store_fp<P>(a.out + j * 8, acc);
}

// ---- division by (X - z): q_k = sum_{i>k} c_i z^(i-k-1) = z^-(k+1) * S_k,  S_k = sum_{i>k} c_i z^i ----
// step 1: t[i] = c_i * z^i  (zpow table);  step 2: inclusive suffix sums inside 2048-blocks + block totals;
// step 3: scan of totals;  step 4: q_k = (suffix_excl_k) * zinv^(k+1).
constexpr int DIV_E = 8;
constexpr int DIV_BLOCK = POLY_THREADS * DIV_E;

template <class P>
__global__ __launch_bounds__(POLY_THREADS) void poly_div_scale_kernel(const uint32_t* __restrict__ c, const uint32_t* __restrict__ zpow, unsigned long long len,
                                                                       uint32_t* __restrict__ t) {
    const unsigned long long i = (unsigned long long)blockIdx.x * POLY_THREADS + threadIdx.x;
    if (i < len) store_fp<P>(t + i * 8, load_fp<P>(c + i * 8) * load_fp<P>(zpow + i * 8));
}
// in-place inclusive suffix sum inside each block (indices descending), block total to totals[block]
template <class P>
__global__ __launch_bounds__(POLY_THREADS) void fr_suffix_add_block_kernel(uint32_t* __restrict__ data, unsigned long long n, uint32_t* __restrict__ totals) {
    using F = Fp<P>;
    __shared__ uint4 sh[2 * POLY_THREADS];
    auto put = [&](int i, const F& x) { sh[2 * i] = make_uint4(x.l[0], x.l[1], x.l[2], x.l[3]); sh[2 * i + 1] = make_uint4(x.l[4], x.l[5], x.l[6], x.l[7]); };
    auto get = [&](int i) { F x; uint4 p = sh[2 * i], q = sh[2 * i + 1]; x.l[0] = p.x; x.l[1] = p.y; x.l[2] = p.z; x.l[3] = p.w; x.l[4] = q.x; x.l[5] = q.y; x.l[6] = q.z; x.l[7] = q.w; return x; };
    // thread r (reversed order) owns elements base .. base + E - 1, processed from the top
    const int r = POLY_THREADS - 1 - threadIdx.x;           // r = 0 owns the highest indices
    const unsigned long long base = (unsigned long long)blockIdx.x * DIV_BLOCK + (unsigned long long)threadIdx.x * DIV_E;
    F v[DIV_E];
    F run = F::zero();
#pragma unroll
    for (int q = DIV_E - 1; q >= 0; q--) {
        const F x = base + q < n ? load_fp<P>(data + (base + q) * 8) : F::zero();
        run = run + x;
        v[q] = run;
    }
    F incl = run;
    put(r, incl);
    __syncthreads();
    for (int d = 1; d < POLY_THREADS; d <<= 1) {
        F other = F::zero();
        const bool has = r >= d;
        if (has) other = get(r - d);
        __syncthreads();
        if (has) incl = incl + other;
        put(r, incl);
        __syncthreads();
    }
    const F excl = r ? get(r - 1) : F::zero();
#pragma unroll
    for (int q = 0; q < DIV_E; q++)
        if (base + q < n) store_fp<P>(data + (base + q) * 8, excl + v[q]);
    if (r == POLY_THREADS - 1) store_fp<P>(totals + (size_t)blockIdx.x * 8, incl);
}
// totals[b] <- sum of totals of blocks above b (exclusive suffix), one 1024-thread workgroup
template <class P>
__global__ __launch_bounds__(1024) void fr_suffix_add_totals_kernel(uint32_t* __restrict__ totals, unsigned int n_blocks) {
    using F = Fp<P>;
    __shared__ uint4 sh[2 * 1024];
    auto put = [&](int i, const F& x) { sh[2 * i] = make_uint4(x.l[0], x.l[1], x.l[2], x.l[3]); sh[2 * i + 1] = make_uint4(x.l[4], x.l[5], x.l[6], x.l[7]); };
    auto get = [&](int i) { F x; uint4 p = sh[2 * i], q = sh[2 * i + 1]; x.l[0] = p.x; x.l[1] = p.y; x.l[2] = p.z; x.l[3] = p.w; x.l[4] = q.x; x.l[5] = q.y; x.l[6] = q.z; x.l[7] = q.w; return x; };
    const unsigned int per = (n_blocks + 1023) / 1024;
    const int r = 1023 - (int)threadIdx.x;                       // r = 0 owns the highest blocks
    const unsigned int lo = threadIdx.x * per, hi = min(n_blocks, lo + per);
    F mine = F::zero();
    for (unsigned int b = lo; b < hi; b++) mine = mine + load_fp<P>(totals + (size_t)b * 8);
    F incl = mine;
    put(r, incl);
    __syncthreads();
    const int live = (int)((n_blocks + per - 1) / per);             // threads that own blocks (the highest r): sums from further away are zeros
    for (int d = 1; d < live; d <<= 1) {
        F other = F::zero();
        const bool has = r >= d;
        if (has) other = get(r - d);
        __syncthreads();
        if (has) incl = incl + other;
        put(r, incl);
        __syncthreads();
    }
    F run = r ? get(r - 1) : F::zero();                          // everything above this thread's blocks
    for (unsigned int b = hi; b-- > lo;) {
        const F t = load_fp<P>(totals + (size_t)b * 8);
        store_fp<P>(totals + (size_t)b * 8, run);
        run = run + t;
    }
}
// q[k] = (S_incl[k+1] + above(block(k+1))) * zinv^(k+1), k < len - 1
template <class P>
__global__ __launch_bounds__(POLY_THREADS) void poly_div_finish_kernel(const uint32_t* __restrict__ sfx, const uint32_t* __restrict__ totals,
                                                                        const uint32_t* __restrict__ zinvpow, unsigned long long len, uint32_t* __restrict__ q,
                                                                        uint32_t* __restrict__ rem) {
    const unsigned long long k = (unsigned long long)blockIdx.x * POLY_THREADS + threadIdx.x;
    if (k == 0 && rem) store_fp<P>(rem, load_fp<P>(sfx) + load_fp<P>(totals));       // sum_i c_i z^i = p(z): the remainder
    if (k + 1 >= len) return;
    const Fp<P> s = load_fp<P>(sfx + (k + 1) * 8) + load_fp<P>(totals + ((k + 1) / DIV_BLOCK) * 8);
    store_fp<P>(q + k * 8, s * load_fp<P>(zinvpow + (k + 1) * 8));
}

// ---- mask_polynomial (prover.rs:463-486): p += (b_0 + b_1 X + .. + b_{h}) (X^n - 1), for up to 8 polynomials at once.
// p has n coefficients (an iNTT output) in a row of at least n + h + 1 slots: p[j] -= b_j, p[n + j] = b_j.
constexpr int MASK_MAX_ROWS = 8;
constexpr int MASK_MAX_BLIND = 4;
// number of coefficients up to and including the highest non-zero one (0 for the zero polynomial) -> *out, which must be zero
// on entry; one 32-byte element per thread, one atomic per workgroup that sees a non-zero element.  The prover's only guard
// against an unsatisfied witness is the quotient's degree (prover.rs:915-918, `WrongQuotientPolyDegree`).
static __global__ __launch_bounds__(POLY_THREADS) void poly_degree_kernel(const uint32_t* __restrict__ c, unsigned long long len, unsigned long long* __restrict__ out) {
    __shared__ unsigned long long best;
    if (threadIdx.x == 0) best = 0;
    __syncthreads();
    const unsigned long long i = (unsigned long long)blockIdx.x * POLY_THREADS + threadIdx.x;
    if (i < len) {
        const uint4 a = reinterpret_cast<const uint4*>(c + i * 8)[0], b = reinterpret_cast<const uint4*>(c + i * 8)[1];
        if (a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w) atomicMax(&best, i + 1);
    }
    __syncthreads();
    if (threadIdx.x == 0 && best) atomicMax(out, best);
}

struct MaskArgs {
    uint32_t* rows[MASK_MAX_ROWS];
    unsigned long long n;
    int n_rows, n_blind;
    uint32_t blind[MASK_MAX_ROWS][MASK_MAX_BLIND][8];       // Montgomery
};
template <class P>
__global__ void poly_mask_kernel(MaskArgs a) {
    using F = Fp<P>;
    const int t = threadIdx.x;
    if (t >= a.n_rows * a.n_blind) return;
    const int r = t / a.n_blind, j = t % a.n_blind;
    F b;
#pragma unroll
    for (int q = 0; q < 8; q++) b.l[q] = a.blind[r][j][q];
    uint32_t* p = a.rows[r];
    store_fp<P>(p + (size_t)j * 8, load_fp<P>(p + (size_t)j * 8) - b);
    store_fp<P>(p + (a.n + j) * 8, b);
}

// ---- a13: split_quotient_polynomial (plonk/src/proof_system/prover.rs:902-960) in one launch ----------------------------------------
// The quotient's W(n + 1) + 3 coefficients into W rows of `stride` (>= n + 3) slots: row i takes the coefficients [i (n + 2), (i + 1)(n + 2))
// (the last row: what is left up to the expected degree), row i < W - 1 gets the blinder b_i as coefficient n + 2 and row i > 0 loses
// b_{i-1} from its constant term (:946-957); every other slot of a row is zeroed.  Two lanes per 32-byte element (16 B each) for the copy.
constexpr int SPLIT_MAX_ROWS = 8;
struct SplitArgs {
    const uint32_t* q;
    uint32_t* out;
    unsigned long long n, stride, expected;     // expected = W (n + 1) + 2: the quotient's degree
    int W;
    uint32_t blind[SPLIT_MAX_ROWS - 1][8];      // Montgomery
};
template <class P>
__global__ __launch_bounds__(256) void poly_split_quotient_kernel(SplitArgs a) {
    using F = Fp<P>;
    const unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const unsigned int i = blockIdx.y;
    if (t >= 2 * a.stride) return;
    const unsigned long long j = t >> 1;
    const unsigned long long lo = (unsigned long long)i * (a.n + 2), hi = (int)i < a.W - 1 ? lo + a.n + 2 : a.expected + 1;
    uint4* dst = reinterpret_cast<uint4*>(a.out + ((size_t)i * a.stride + j) * 8) + (t & 1);
    const bool edge = (j == 0 && i > 0) || (j == a.n + 2 && (int)i < a.W - 1);
    if (!edge) {
        *dst = lo + j < hi ? reinterpret_cast<const uint4*>(a.q + (lo + j) * 8)[t & 1] : make_uint4(0, 0, 0, 0);
        return;
    }
    if (t & 1) return;                                                       // the two edge elements of a row: one lane does the field arithmetic
    F v;
    if (j == 0) {
        F b;
#pragma unroll
        for (int k = 0; k < 8; k++) b.l[k] = a.blind[i - 1][k];
        v = (lo < hi ? load_fp<P>(a.q + lo * 8) : F::zero()) - b;
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) v.l[k] = a.blind[i][k];
    }
    store_fp<P>(a.out + ((size_t)i * a.stride + j) * 8, v);
}

// ---- a1: the gather of Arithmetization::compute_wire_polynomials (relation/src/constraint_system.rs:1225-1247) -----------------------
// out[t] = witness[wire_variables[t]] for the W * n cells of the wire table: 32-byte elements, two lanes per element (16 B each),
// so a wave reads and writes whole 128-byte lines where the indices run (the bench circuit's mostly do).  Field-independent.
static __global__ __launch_bounds__(256) void wire_gather_kernel(const uint4* __restrict__ witness, unsigned long long n_vars, const uint32_t* __restrict__ vars,
                                                          unsigned long long count, uint4* __restrict__ out, unsigned int* __restrict__ bad) {
    const unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= 2 * count) return;
    const unsigned long long cell = t >> 1;
    const uint32_t v = vars[cell];
    if (v >= n_vars) { if (bad) atomicOr(bad, 1u); out[t] = make_uint4(0, 0, 0, 0); return; }
    out[t] = witness[2ull * v + (t & 1)];
}

// ---- division by the vanishing polynomial of a proof-linking domain (proof_linking.rs:119-158) -------------------------
// Z_D(X) = prod_{i < count} (X - rho g^i).  When Z_D divides p, the quotient is p(x) / Z_D(x) pointwise on a coset that
// avoids the roots; the kernel below multiplies the coset evaluations by 1 / Z_D(x): thread t owns the K points
// t, t + T, .., reads the roots from a table and inverts its K products with one field inversion.
struct DivRootsArgs {
    uint32_t* evals;
    const uint32_t* roots;                                   // [count] rho g^i, internal form
    unsigned long long threads;                              // T: n_points = K * T
    unsigned int count;
    uint32_t h[8], w[8], w_step[8];                          // coset offset, domain generator, w^T (internal form; the fold kernel borrows h for its multiplier)
};
// The kernel runs on the reduced-radix field (fx.cuh): the coset evaluations arrive in the INTERNAL form x * R' of the quotient kernels
// (forward NTT with scale = 1, ntt.hip), constants and roots are passed in that form, and the inverse NTT takes the factor back
// (scale = 2).  A product is one v_mad_u64_u32 per limb pair instead of a multiply-add + carry-add pair.
// Bounds (H = HEADROOM_BITS >= 6): z is class M (< 2p), x - r + 2p < 4p: fx_mul wants 2 * 4 <= 2^H and gives < 1.13 p.
template <class X>
__device__ __forceinline__ Fx<X> fx_inverse(const Fx<X>& a) {              // a^(p-2), a class M
    Fx<X> out = Fx<X>::one();
    bool started = false;
    uint32_t ex[X::N];                                                       // p - 2 (BLS12-381's r ends in ...00000001: the borrow runs one word up)
    uint32_t borrow = 2;
#pragma unroll
    for (int w = 0; w < X::N; w++) {
        ex[w] = X::MOD[w] - borrow;
        borrow = X::MOD[w] < borrow ? 1u : 0u;
    }
    for (int w = X::N - 1; w >= 0; w--) {
        const uint32_t e = ex[w];
        for (int b = 31; b >= 0; b--) {
            if (started) out = fx_sqr(out);
            if ((e >> b) & 1u) { out = started ? fx_mul(out, a) : a; started = true; }
        }
    }
    return out;
}
template <class X, int K>
__device__ __forceinline__ void poly_div_roots_pointwise_fx_body(const DivRootsArgs& a) {
    using F = Fx<X>;
    const unsigned long long t = (unsigned long long)blockIdx.x * POLY_THREADS + threadIdx.x;
    if (t >= a.threads) return;
    F x[K], z[K];
    const F step = fx_unpack<X>(a.w_step);
    {
        F acc = fx_unpack<X>(a.h), base = fx_unpack<X>(a.w);                  // h * w^t
        for (unsigned long long e = t; e; e >>= 1) {
            if (e & 1) acc = fx_mul(acc, base);
            base = fx_sqr(base);
        }
        x[0] = acc;
    }
    z[0] = F::one();
#pragma unroll
    for (int k = 1; k < K; k++) { x[k] = fx_mul(x[k - 1], step); z[k] = F::one(); }
    for (unsigned int i = 0; i < a.count; i++) {
        const F r = fx_load_packed<X>(a.roots + (size_t)i * 8);               // canonical, same address in every lane
#pragma unroll
        for (int k = 0; k < K; k++) z[k] = fx_mul(z[k], fx_norm(fx_sub2(x[k], r)));
    }
    x[0] = z[0];
#pragma unroll
    for (int k = 1; k < K; k++) x[k] = fx_mul(x[k - 1], z[k]);
    F iv = fx_inverse<X>(x[K - 1]);
#pragma unroll
    for (int k = K - 1; k > 0; k--) {
        const F zi = fx_mul(iv, x[k - 1]);
        iv = fx_mul(iv, z[k]);
        z[k] = zi;
    }
    z[0] = iv;
#pragma unroll
    for (int k = 0; k < K; k++) {
        uint32_t* e = a.evals + (t + (unsigned long long)k * a.threads) * 8;
        fx_store_packed<X>(e, fx_canonical(fx_mul(fx_load_packed<X>(e), z[k])));
    }
}
// K = 1 or 2 points per thread on small domains (many waves per SIMD hide the dependent chain), 4 on large ones (the eight
// 9-limb values stay in registers at two waves per SIMD; at 8 points the arrays go to scratch)
template <class X, int K>
__global__ __launch_bounds__(POLY_THREADS) void poly_div_roots_pointwise_fx_kernel(DivRootsArgs a) {
    poly_div_roots_pointwise_fx_body<X, K>(a);
}
// out[j] = sum_k c^k p[j + k N], j < N:  p mod (X^N - c) -- p on the N points x with x^N = c is this polynomial on them
template <class P>
__global__ __launch_bounds__(POLY_THREADS) void poly_fold_kernel(const uint32_t* __restrict__ p, unsigned long long len, unsigned long long N, DivRootsArgs a,
                                                                  uint32_t* __restrict__ out) {
    using F = Fp<P>;
    const unsigned long long j = (unsigned long long)blockIdx.x * POLY_THREADS + threadIdx.x;
    if (j >= N) return;
    F acc = F::zero();
    if (j < len) {
        F c;
#pragma unroll
        for (int q = 0; q < 8; q++) c.l[q] = a.h[q];                     // the fold multiplier travels in a.h
        const unsigned long long kmax = (len - 1 - j) / N;
        acc = load_fp<P>(p + (j + kmax * N) * 8);
        for (unsigned long long k = kmax; k-- > 0;) acc = acc * c + load_fp<P>(p + (j + k * N) * 8);
    }
    store_fp<P>(out + j * 8, acc);
}
// out[i] = src[((first + i) mod order) * step]: the evaluations of p at the roots, out of an NTT over a domain that contains them
template <class P>
__global__ void poly_gather_roots_kernel(const uint32_t* __restrict__ src, unsigned long long first, unsigned long long order_mask, unsigned long long step,
                                         unsigned long long count, uint32_t* __restrict__ out) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) store_fp<P>(out + i * 8, load_fp<P>(src + ((first + i) & order_mask) * step * 8));
}

}  // namespace mzk
