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

}  // namespace mzk
