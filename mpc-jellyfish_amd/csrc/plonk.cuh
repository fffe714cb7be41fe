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
#include "poly.cuh"

namespace mzk {

constexpr int PLK_THREADS = 256;
constexpr int PLK_SELECTORS = 13;
constexpr int PLK_WIRES = 5;
constexpr int PLK_RATIO = 8;                 // m / n for TurboPlonk and UltraPlonk

struct QuotientArgs {
    const uint32_t* sel;      // [13][m] coset evaluations
    const uint32_t* sig;      // [W][m]
    const uint32_t* wire;     // [W][m]
    const uint32_t* z;        // [m]
    const uint32_t* pi;       // [m]
    const uint32_t* xs;       // [m]  x_i = g * w_m^i
    const uint32_t* inv_den;  // [m]  1 / (n * (x_i - 1))
    uint32_t* out;            // [m]
    unsigned long long m;
    uint32_t k[PLK_WIRES][8];         // coset representatives k_j (Montgomery)
    uint32_t alpha[8], alpha2[8], beta[8], gamma[8];
    uint32_t zh_inv[PLK_RATIO][8];    // 1 / Z_H(x_i), period 8 in i
};

template <class P>
__device__ __forceinline__ Fp<P> arg_fp(const uint32_t (&a)[8]) {
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = a[i];
    return r;
}

template <class P>
__global__ __launch_bounds__(PLK_THREADS) void plonk_quotient_kernel(QuotientArgs a) {
    using F = Fp<P>;
    const unsigned long long i = (unsigned long long)blockIdx.x * PLK_THREADS + threadIdx.x;
    if (i >= a.m) return;
    const unsigned long long m = a.m;
    F w[PLK_WIRES];
#pragma unroll
    for (int j = 0; j < PLK_WIRES; j++) w[j] = load_fp<P>(a.wire + ((size_t)j * m + i) * 8);
    auto sel = [&](int j) { return load_fp<P>(a.sel + ((size_t)j * m + i) * 8); };
    // ---- gate identity (prover.rs:696-708)
    F t = sel(11) + load_fp<P>(a.pi + i * 8);                       // q_c + pi
#pragma unroll
    for (int j = 0; j < 4; j++) t = t + sel(j) * w[j];              // q_lc
    const F w01 = w[0] * w[1], w23 = w[2] * w[3];
    t = t + sel(4) * w01 + sel(5) * w23;                            // q_mul
    t = t + sel(12) * (w01 * w23 * w[4]);                           // q_ecc
#pragma unroll
    for (int j = 0; j < 4; j++) {                                   // q_hash * w^5
        const F w2 = sqr(w[j]);
        t = t + sel(6 + j) * (sqr(w2) * w[j]);
    }
    t = t - sel(10) * w[4];                                         // q_o
    // ---- copy constraints (prover.rs:741-758)
    const F alpha = arg_fp<P>(a.alpha), beta = arg_fp<P>(a.beta), gamma = arg_fp<P>(a.gamma);
    const F z_x = load_fp<P>(a.z + i * 8);
    const F z_xw = load_fp<P>(a.z + ((i + PLK_RATIO) % m) * 8);
    const F xb = load_fp<P>(a.xs + i * 8) * beta;
    F acc1 = z_x, acc2 = z_xw;
#pragma unroll
    for (int j = 0; j < PLK_WIRES; j++) {
        const F wg = w[j] + gamma;
        acc1 = acc1 * (wg + arg_fp<P>(a.k[j]) * xb);
        acc2 = acc2 * (wg + load_fp<P>(a.sig + ((size_t)j * m + i) * 8) * beta);
    }
    const F t1 = t + alpha * (acc1 - acc2);
    const F t2 = arg_fp<P>(a.alpha2) * ((z_x - F::one()) * load_fp<P>(a.inv_den + i * 8));
    store_fp<P>(a.out + i * 8, t1 * arg_fp<P>(a.zh_inv[i % PLK_RATIO]) + t2);          // prover.rs:657
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

// ---- round 2: the permutation grand product ---------------------------------------
// Replaces the serial loop of Arithmetization::compute_prod_permutation_polynomial
// (relation/src/constraint_system.rs:1197-1223), which performs one field division per gate:
//   z[0] = 1,  z[j+1] = z[j] * prod_i (w_ij + gamma + beta k_i w^j) / prod_i (w_ij + gamma + beta sigma_ij),  j < n-1.
// Here: ratios with one shared inversion per 8 gates, then a three-phase parallel prefix product.
constexpr int PERM_B = 8;             // gates per thread in the ratio kernel
constexpr int SCAN_T = 256;           // threads per scan workgroup
constexpr int SCAN_E = 8;             // elements per thread
constexpr int SCAN_BLOCK = SCAN_T * SCAN_E;

struct PermArgs {
    const uint32_t* wire;      // [W][n] wire values (evaluations on H)
    const uint32_t* sigma;     // [W][n] extended permutation values sigma_i(w^j)
    const uint32_t* omega;     // [n] w^j
    uint32_t* ratio;           // [n] out: ratio[j] for j < n-1, ratio[n-1] = 1
    unsigned long long n;
    uint32_t k[PLK_WIRES][8];
    uint32_t beta[8], gamma[8];
};

template <class P>
__global__ __launch_bounds__(PLK_THREADS) void plonk_perm_ratio_kernel(PermArgs a) {
    using F = Fp<P>;
    const unsigned long long t = (unsigned long long)blockIdx.x * PLK_THREADS + threadIdx.x;
    const unsigned long long start = t * PERM_B;
    if (start >= a.n) return;
    const F beta = arg_fp<P>(a.beta), gamma = arg_fp<P>(a.gamma);
    F num[PERM_B], pref[PERM_B];
    F run = F::one();
#pragma unroll
    for (int q = 0; q < PERM_B; q++) {
        const unsigned long long j = start + q;
        F nu = F::one(), de = F::one();
        if (j + 1 < a.n) {
            const F bw = beta * load_fp<P>(a.omega + j * 8);
#pragma unroll
            for (int i = 0; i < PLK_WIRES; i++) {
                const F wg = load_fp<P>(a.wire + ((size_t)i * a.n + j) * 8) + gamma;
                nu = nu * (wg + bw * arg_fp<P>(a.k[i]));
                de = de * (wg + beta * load_fp<P>(a.sigma + ((size_t)i * a.n + j) * 8));
            }
        }
        num[q] = nu;
        pref[q] = run;
        run = run * de;
        store_fp<P>(a.ratio + j * 8, de);                   // parked
    }
    F inv_run = inv(run);            // a zero denominator (probability ~ n/r) yields 0, as 1/0 would panic in the reference
#pragma unroll
    for (int q = PERM_B - 1; q >= 0; q--) {
        const unsigned long long j = start + q;
        const F de = load_fp<P>(a.ratio + j * 8);
        store_fp<P>(a.ratio + j * 8, num[q] * (inv_run * pref[q]));
        inv_run = inv_run * de;
    }
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
        for (int d = 1; d < 1024; d <<= 1) {
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

}  // namespace mzk
