// plookup.cuh -- the witness-dependent Plookup builders of UltraPlonk on the device.
//
//   merged lookup table / merged lookup witness     relation/src/constraint_system.rs:1290-1309, 1441-1480
//   compute_lookup_sorted_vec_polynomials           constraint_system.rs:1370-1417  (HashMap merge in table order)
//   compute_lookup_prod_polynomial                  constraint_system.rs:1311-1368  (one field division per row)
//
// The reference counts lookups in a HashMap<F, usize> and then walks the table, emitting every table entry once
// plus one copy per lookup at the FIRST entry holding that value.  Here: an open-addressing hash table in HBM
// keyed by the 256-bit value and holding the smallest table index with that value, per-index lookup counters,
// an exclusive scan of (1 + count) and a gather by binary search.  The product is the same shared-inversion
// ratio + prefix-product pipeline as the permutation product (plonk.cuh).
#pragma once
#include <hip/hip_runtime.h>

#include "fp.cuh"
#include "plonk.cuh"

namespace mzk {

constexpr uint32_t PLK_EMPTY = 0xFFFFFFFFu;
constexpr int PLK_SCAN_T = 1024;
constexpr int PLK_SCAN_E = 4;
constexpr int PLK_SCAN_BLOCK = PLK_SCAN_T * PLK_SCAN_E;

struct MergeArgs {
    const uint32_t* wire;        // [6][n] wire values on H
    const uint32_t* range;       // [n] table polynomials' values on H (pk)
    const uint32_t* key;
    const uint32_t* table_dom_sep;
    const uint32_t* q_dom_sep;
    const uint32_t* q_lookup;
    uint32_t* table;             // [n] out: merged table values
    uint32_t* lookup;            // [n] out: merged lookup witness values
    unsigned long long n;
    uint32_t tau[8];
};

// first + q_lookup * tau * (dom_sep + tau * (a0 + tau * (a1 + tau * a2)))   (structs.rs:926-956)
template <class P>
__device__ __forceinline__ Fp<P> plookup_merge(const Fp<P>& first, const Fp<P>& q_lookup, const Fp<P>& tau, const Fp<P>& dom_sep,
                                               const Fp<P>& a0, const Fp<P>& a1, const Fp<P>& a2) {
    return first + q_lookup * tau * (dom_sep + tau * (a0 + tau * (a1 + tau * a2)));
}

template <class P>
__global__ __launch_bounds__(PLK_THREADS) void plookup_merge_kernel(MergeArgs a) {
    using F = Fp<P>;
    const unsigned long long i = (unsigned long long)blockIdx.x * PLK_THREADS + threadIdx.x;
    if (i >= a.n) return;
    const F tau = arg_fp<P>(a.tau), ql = load_fp<P>(a.q_lookup + i * 8);
    auto w = [&](int j) { return load_fp<P>(a.wire + ((size_t)j * a.n + i) * 8); };
    store_fp<P>(a.table + i * 8, plookup_merge<P>(load_fp<P>(a.range + i * 8), ql, tau, load_fp<P>(a.table_dom_sep + i * 8), load_fp<P>(a.key + i * 8), w(3), w(4)));
    store_fp<P>(a.lookup + i * 8, plookup_merge<P>(w(5), ql, tau, load_fp<P>(a.q_dom_sep + i * 8), w(0), w(1), w(2)));
}

__device__ __forceinline__ bool fr_words_equal(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1) {
    return ((a0.x ^ b0.x) | (a0.y ^ b0.y) | (a0.z ^ b0.z) | (a0.w ^ b0.w) | (a1.x ^ b1.x) | (a1.y ^ b1.y) | (a1.z ^ b1.z) | (a1.w ^ b1.w)) == 0;
}
__device__ __forceinline__ uint32_t fr_words_hash(const uint4& a0, const uint4& a1) {
    uint32_t h = a0.x * 0x9E3779B1u;
    h = (h ^ (h >> 15)) + a0.y * 0x85EBCA77u;
    h = (h ^ (h >> 13)) + a0.z * 0xC2B2AE3Du;
    h = (h ^ (h >> 16)) + a0.w * 0x27D4EB2Fu;
    h = (h ^ (h >> 15)) + a1.x * 0x165667B1u;
    h = (h ^ (h >> 13)) + (a1.y ^ a1.z ^ a1.w) * 0x9E3779B1u;
    return h ^ (h >> 16);
}

// slots[h] = smallest table index holding that value.  Entries equal to their predecessor are never the first
// occurrence (the zero padding of the range table is one long run) and are skipped.
__global__ __launch_bounds__(PLK_THREADS) void plookup_hash_insert_kernel(const uint4* __restrict__ table, unsigned long long n, uint32_t* __restrict__ slots, uint32_t mask) {
    const unsigned long long i = (unsigned long long)blockIdx.x * PLK_THREADS + threadIdx.x;
    if (i >= n) return;
    const uint4 v0 = table[2 * i], v1 = table[2 * i + 1];
    if (i > 0 && fr_words_equal(v0, v1, table[2 * i - 2], table[2 * i - 1])) return;
    uint32_t h = fr_words_hash(v0, v1) & mask;
    for (uint32_t probe = 0; probe <= mask; probe++) {            // load factor <= 1/4: terminates long before the bound
        const uint32_t cur = atomicCAS(&slots[h], PLK_EMPTY, (uint32_t)i);
        if (cur == PLK_EMPTY) return;
        if (fr_words_equal(v0, v1, table[2 * (size_t)cur], table[2 * (size_t)cur + 1])) { atomicMin(&slots[h], (uint32_t)i); return; }
        h = (h + 1) & mask;
    }
}

// count[first index of lookup[j]] += 1 for j < n_lookups; a value absent from the table raises *missing
// ("some lookup variables might be outside the table", constraint_system.rs:1410-1412)
__global__ __launch_bounds__(PLK_THREADS) void plookup_hash_count_kernel(const uint4* __restrict__ table, const uint4* __restrict__ lookup, unsigned long long n_lookups,
                                                                          const uint32_t* __restrict__ slots, uint32_t mask, uint32_t* __restrict__ count,
                                                                          uint32_t* __restrict__ missing) {
    const unsigned long long j = (unsigned long long)blockIdx.x * PLK_THREADS + threadIdx.x;
    const bool valid = j < n_lookups;
    uint32_t idx = PLK_EMPTY;
    if (valid) {
        const uint4 v0 = lookup[2 * j], v1 = lookup[2 * j + 1];
        uint32_t h = fr_words_hash(v0, v1) & mask;
        for (uint32_t probe = 0; probe <= mask; probe++) {
            const uint32_t cur = slots[h];
            if (cur == PLK_EMPTY) break;
            if (fr_words_equal(v0, v1, table[2 * (size_t)cur], table[2 * (size_t)cur + 1])) { idx = cur; break; }
            h = (h + 1) & mask;
        }
        if (idx == PLK_EMPTY) atomicAdd(missing, 1u);
    }
    // most rows of a circuit look up the same value (0): one atomic per wavefront when all lanes agree
    const unsigned long long live = __ballot(valid && idx != PLK_EMPTY);
    if (live == 0) return;
    const int leader = __ffsll((long long)live) - 1;
    const uint32_t first = __shfl(idx, leader);
    const bool mine = valid && idx != PLK_EMPTY;
    if (__all(!mine || idx == first)) {
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&count[first], (uint32_t)__popcll(live));
    } else if (mine) {
        atomicAdd(&count[idx], 1u);
    }
}

// exclusive scan of (1 + count[i]), i < n: phase 1 per block of 4096, phase 2 over block totals, phase 3 folded into the gather
__global__ __launch_bounds__(PLK_SCAN_T) void plookup_scan_block_kernel(const uint32_t* __restrict__ count, unsigned long long n, uint32_t* __restrict__ pos,
                                                                        uint32_t* __restrict__ totals) {
    __shared__ uint32_t sh[PLK_SCAN_T];
    const unsigned long long base = (unsigned long long)blockIdx.x * PLK_SCAN_BLOCK + (unsigned long long)threadIdx.x * PLK_SCAN_E;
    uint32_t v[PLK_SCAN_E], run = 0;
#pragma unroll
    for (int q = 0; q < PLK_SCAN_E; q++) {
        v[q] = run;
        run += base + q < n ? 1u + count[base + q] : 0u;
    }
    sh[threadIdx.x] = run;
    __syncthreads();
    uint32_t incl = run;
    for (int d = 1; d < PLK_SCAN_T; d <<= 1) {
        const uint32_t other = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0u;
        __syncthreads();
        incl += other;
        sh[threadIdx.x] = incl;
        __syncthreads();
    }
    const uint32_t excl = incl - run;
#pragma unroll
    for (int q = 0; q < PLK_SCAN_E; q++)
        if (base + q < n) pos[base + q] = excl + v[q];
    if (threadIdx.x == PLK_SCAN_T - 1) totals[blockIdx.x] = incl;
}
__global__ __launch_bounds__(1024) void plookup_scan_totals_kernel(uint32_t* __restrict__ totals, unsigned int n_blocks) {
    __shared__ uint32_t sh[1024];
    uint32_t carry = 0;
    for (unsigned int c0 = 0; c0 < n_blocks; c0 += 1024) {
        const unsigned int i = c0 + threadIdx.x;
        const uint32_t mine = i < n_blocks ? totals[i] : 0u;
        uint32_t incl = mine;
        sh[threadIdx.x] = incl;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const uint32_t other = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0u;
            __syncthreads();
            incl += other;
            sh[threadIdx.x] = incl;
            __syncthreads();
        }
        if (i < n_blocks) totals[i] = carry + incl - mine;
        carry += sh[1023];
        __syncthreads();
    }
}
__global__ __launch_bounds__(PLK_THREADS) void plookup_scan_apply_kernel(uint32_t* __restrict__ pos, const uint32_t* __restrict__ totals, unsigned long long n) {
    const unsigned long long i = (unsigned long long)blockIdx.x * PLK_THREADS + threadIdx.x;
    if (i < n) pos[i] += totals[i / PLK_SCAN_BLOCK];
}

// sorted[p] = table[i] for pos[i] <= p < pos[i] + 1 + count[i]: the largest i with pos[i] <= p
__global__ __launch_bounds__(PLK_THREADS) void plookup_gather_kernel(const uint4* __restrict__ table, const uint32_t* __restrict__ pos, unsigned long long n,
                                                                      unsigned long long out_len, uint4* __restrict__ sorted) {
    const unsigned long long p = (unsigned long long)blockIdx.x * PLK_THREADS + threadIdx.x;
    if (p >= out_len) return;
    unsigned long long lo = 0, hi = n;                    // invariant: pos[lo] <= p, (hi == n or pos[hi] > p)
    while (hi - lo > 1) {
        const unsigned long long mid = (lo + hi) >> 1;
        if (pos[mid] <= p) lo = mid; else hi = mid;
    }
    sorted[2 * p] = table[2 * lo];
    sorted[2 * p + 1] = table[2 * lo + 1];
}

struct LookupProdArgs {
    const uint32_t* table;     // [n] merged table
    const uint32_t* lookup;    // [n] merged lookup witness
    const uint32_t* sorted;    // [2n-1]
    uint32_t* ratio;           // [n] out: a_j for j < n-2, 1 beyond
    uint32_t* den;             // [n] out: b_j (fr_batch_div_kernel then forms a_j / b_j)
    unsigned long long n;
    uint32_t beta[8], gamma[8];
};

// one row per thread: the numerator and denominator of the Plookup product's step j (constraint_system.rs:1340-1362)
template <class P>
__global__ __launch_bounds__(PLK_THREADS) void plookup_terms_kernel(LookupProdArgs a) {
    using F = Fp<P>;
    const unsigned long long j = (unsigned long long)blockIdx.x * PLK_THREADS + threadIdx.x;
    if (j >= a.n) return;
    F nu = F::one(), de = F::one();
    if (j + 2 < a.n) {
        const F beta = arg_fp<P>(a.beta), gamma = arg_fp<P>(a.gamma);
        const F b1 = beta + F::one(), g1 = gamma * b1;
        nu = b1 * (gamma + load_fp<P>(a.lookup + j * 8)) * (g1 + load_fp<P>(a.table + j * 8) + beta * load_fp<P>(a.table + (j + 1) * 8));
        de = (g1 + load_fp<P>(a.sorted + j * 8) + beta * load_fp<P>(a.sorted + (j + 1) * 8)) *
             (g1 + load_fp<P>(a.sorted + (a.n - 1 + j) * 8) + beta * load_fp<P>(a.sorted + (a.n + j) * 8));
    }
    store_fp<P>(a.ratio + j * 8, nu);
    store_fp<P>(a.den + j * 8, de);
}

// product_vec.push(F::one()) after the loop (constraint_system.rs:1364): the last value is the literal one
template <class P>
__global__ void plookup_set_last_one_kernel(uint32_t* __restrict__ out, unsigned long long n) {
    if (threadIdx.x == 0 && blockIdx.x == 0) store_fp<P>(out + (n - 1) * 8, Fp<P>::one());
}

}  // namespace mzk
