// msm.cuh -- Pippenger multi-scalar multiplication on G1 for gfx950.
//
// Replaces ark-ec's VariableBaseMSM::msm_bigint at the reference call sites
//   primitives/src/pcs/univariate_kzg/mod.rs:109-111 (commit) and :151-155 (open).
// Contract: result = sum_i k_i * P_i as a Jacobian point; only the group
// element is pinned, not the algorithm.  Pipeline (all on one stream):
//   1. msm_digits    signed base-2^c digits of every scalar (16-bit, window-major)
//   2. msm_sort<COUNT> / msm_scan / msm_sort<SCATTER>   counting sort of point indices by (window,
//                    bucket); one workgroup per (2048-bucket range, window), LDS atomics only
//   3. msm_order     buckets of each window ranked by load (equal-length loops within a wave)
//   4. msm_accumulate  one thread per (window,bucket): XYZZ += +-P (mixed add, all in VGPRs)
//   5. msm_fold x log2(M)  in-place recursive halving: after level l the main array keeps
//                    sum_i B_i folded to M/2^l entries and T_j (at offset M/2^j) the partial sums
//                    of the buckets whose index bit (log2 M - j) is set, so that
//                    S_w = X[0] + sum_j 2^(log2 M - j) X[M/2^j]     (weights i+1 for bucket i)
//                    A level is one EC addition deep; levels with few additions run them on quads of lanes
//                    (msm_fold_quad, ecx.cuh xyzzx_add_quad: 8 us instead of 17 us per level).
//   6. msm_collect + host Horner over the (1 + log2 M) points per window (hostfp.hpp).
// Algorithmic bytes: N * (2*|Fq| + 32) per MSM.
#pragma once
#include <hip/hip_runtime.h>

#include "ec.cuh"
#include "ecx.cuh"

namespace mzk {

constexpr int MSM_THREADS = 256;
constexpr int MSM_ACC_THREADS = 128;
// heavy buckets (msm_heavy_* below): entries one thread of a level-1 run sums / entries per level-1 run / items a run of a later
// level may take (32 per thread before the tree) / counters per window (runs of levels 0, B, C; parts of levels 2, 3)
constexpr uint32_t MSM_HEAVY_PER_THREAD = 16;
constexpr uint32_t MSM_HEAVY_RUN = MSM_ACC_THREADS * MSM_HEAVY_PER_THREAD;
constexpr uint32_t MSM_HEAVY_FANIN = MSM_ACC_THREADS * 32;
constexpr int MSM_HEAVY_COUNTERS = 6;
// a bucket is HEAVY when it holds more than MSM_HEAVY_RUN entries beyond the cap: its entries all go to the msm_heavy_* kernels (a workgroup
// per run of MSM_HEAVY_RUN entries: worth it only for runs that fill it -- with the threshold at 2 or 4 caps, 2^20 32-bit scalars (4096
// buckets of 256) went from 3.0 to 4.6 ms and 4096 repeated values from 5.8 to 21 ms); between the cap and that it is over-long: its first
// `cap` entries stay with the regular accumulation, the rest with the msm_long_* kernels
__host__ __device__ inline bool msm_is_heavy(uint32_t count, uint32_t cap) { return count > cap + MSM_HEAVY_RUN; }

// window size of the plain path: ~log2(n) - 2 (mean bucket load ~8: short dependent chains, enough buckets to
// fill the chip even for small n), clamped to [4, 16]
inline int msm_choose_window(unsigned long long n) {
    int lg = 0;
    while ((1ull << (lg + 1)) <= n) lg++;
    int c = lg - 2;
    if (c < 4) c = 4;
    if (c > 16) c = 16;
    return c;
}
inline int msm_num_windows(int scalar_bits, int c) { return (scalar_bits + 1 + c - 1) / c; }

// Signed digit of window w: returns magnitude (0..2^(c-1)) and sign.  The recoding carries
// upward, so digits are produced low to high by one thread per scalar.
struct DigitIter {
    uint32_t k[8];
    uint32_t carry;
    __device__ __forceinline__ void next(int w, int c, uint32_t& mag, uint32_t& negative) {
        const int off = w * c, word = off >> 5, bit = off & 31;
        uint32_t buf = 0;
        if (word < 8) {
            uint64_t two = k[word];
            if (word + 1 < 8) two |= (uint64_t)k[word + 1] << 32;
            buf = (uint32_t)(two >> bit);
        }
        uint32_t coef = (buf & ((1u << c) - 1)) + carry;
        carry = coef > (1u << (c - 1)) ? 1u : 0u;          // digit in (-2^(c-1), 2^(c-1)]
        negative = carry;
        mag = carry ? (1u << c) - coef : coef;
    }
};

template <class FR>
__device__ __forceinline__ void load_scalar(DigitIter& it, const uint32_t* __restrict__ scalars, unsigned long long i, int is_mont) {
    Fp<FR> s = load_fp<FR>(scalars + i * 8);
    if (is_mont) s = from_mont(s);
#pragma unroll
    for (int q = 0; q < 8; q++) it.k[q] = s.l[q];
    it.carry = 0;
}

// ---- bucket sort without global atomics ----------------------------------------------------------
// digits[w*n + i] = sign<<15 | (magnitude-1), MSM_EMPTY for a zero digit.  (sign=1, magnitude=2^15)
// cannot occur because negative digits have magnitude < 2^(c-1), so 0xFFFF is free.
constexpr uint32_t MSM_EMPTY = 0xFFFFu;
constexpr int MSM_RANGE_LOG = 11;            // buckets per sorting workgroup (LDS histogram of 2048 bins)
constexpr int MSM_SORT_THREADS = 1024;

template <class FR>
__global__ __launch_bounds__(MSM_THREADS) void msm_digits_kernel(const uint32_t* __restrict__ scalars, unsigned long long n, int is_mont,
                                                                  int c, int n_win, uint16_t* __restrict__ digits, unsigned long long stride) {
    const unsigned long long i = (unsigned long long)blockIdx.x * MSM_THREADS + threadIdx.x;
    if (i >= n) return;
    DigitIter it;
    load_scalar<FR>(it, scalars, i, is_mont);
    for (int w = 0; w < n_win; w++) {
        uint32_t mag, ng;
        it.next(w, c, mag, ng);
        digits[(size_t)w * stride + i] = (uint16_t)(mag ? ((ng << 15) | (mag - 1)) : MSM_EMPTY);
    }
}

// One workgroup per (bucket range, window): it scans the window's n digits and keeps those whose
// bucket lies in its range -- counts them in LDS (pass COUNT) or hands out positions from LDS
// cursors (pass SCATTER).  All atomics are LDS atomics; global writes go to the range's own segment.
template <bool SCATTER>
__global__ __launch_bounds__(MSM_SORT_THREADS) void msm_sort_kernel(const uint16_t* __restrict__ digits, unsigned long long n, unsigned long long stride, uint32_t M,
                                                                     uint32_t* __restrict__ hist, const uint32_t* __restrict__ offs,
                                                                     uint32_t* __restrict__ sorted) {
    __shared__ uint32_t bins[1 << MSM_RANGE_LOG];
    const uint32_t range = blockIdx.x, w = blockIdx.y, tid = threadIdx.x;
    const uint32_t rsize = M < (1u << MSM_RANGE_LOG) ? M : (1u << MSM_RANGE_LOG);
    const uint32_t rbase = range * rsize;
    for (uint32_t j = tid; j < rsize; j += MSM_SORT_THREADS) bins[j] = SCATTER ? offs[(size_t)w * M + rbase + j] : 0u;
    __syncthreads();
    const uint16_t* dw = digits + (size_t)w * stride;      // stride is a multiple of 8: 16-B aligned rows
    uint32_t* out = sorted + (size_t)w * n;
    // 8 digits (16 B) per thread per step
    const unsigned long long n8 = n >> 3;
    for (unsigned long long q = tid; q < n8; q += MSM_SORT_THREADS) {
        const uint4 v = reinterpret_cast<const uint4*>(dw)[q];
        const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t d = (wd[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
            const uint32_t b = d & 0x7FFFu;
            if (d != MSM_EMPTY && b - rbase < rsize) {
                if (SCATTER) out[atomicAdd(&bins[b - rbase], 1u)] = (uint32_t)(q * 8 + k) | ((d >> 15) << 31);
                else atomicAdd(&bins[b - rbase], 1u);
            }
        }
    }
    for (unsigned long long i = (n8 << 3) + tid; i < n; i += MSM_SORT_THREADS) {
        const uint32_t d = dw[i], b = d & 0x7FFFu;
        if (d != MSM_EMPTY && b - rbase < rsize) {
            if (SCATTER) out[atomicAdd(&bins[b - rbase], 1u)] = (uint32_t)i | ((d >> 15) << 31);
            else atomicAdd(&bins[b - rbase], 1u);
        }
    }
    if (!SCATTER) {
        __syncthreads();
        for (uint32_t j = tid; j < rsize; j += MSM_SORT_THREADS) hist[(size_t)w * M + rbase + j] = bins[j];
    }
}

// per-window exclusive scan; one 1024-thread workgroup per window
__global__ __launch_bounds__(1024) void msm_scan_kernel(const uint32_t* __restrict__ hist, uint32_t* __restrict__ offs, uint32_t M) {
    __shared__ uint32_t part[1024];
    const int w = blockIdx.x, t = threadIdx.x;
    const uint32_t per = (M + 1023) / 1024;
    const uint32_t lo = t * per, hi = min(M, lo + per);
    uint32_t sum = 0;
    for (uint32_t b = lo; b < hi; b++) sum += hist[(size_t)w * M + b];
    part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {            // Hillis-Steele inclusive scan of the partials
        uint32_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (uint32_t b = lo; b < hi; b++) {
        offs[(size_t)w * M + b] = run;
        run += hist[(size_t)w * M + b];
    }
}

// order[w*M + rank] = bucket, buckets of a window listed by descending point count (counting sort on the
// count, clamped to 1023), so the 64 lanes of an accumulation wave run loops of equal length.
// Three small kernels; a workgroup owns MSM_ORDER_SLICE consecutive buckets of one window.
constexpr uint32_t MSM_ORDER_SLICE = 8192;
__device__ __forceinline__ uint32_t order_key(uint32_t count) { return 1023u - min(count, 1023u); }

// keycnt[w][key] += number of buckets of this slice with that key   (keycnt zeroed first)
__global__ __launch_bounds__(1024) void msm_order_hist_kernel(const uint32_t* __restrict__ hist, uint32_t M, uint32_t* __restrict__ keycnt) {
    __shared__ uint32_t cnt[1024];
    const uint32_t w = blockIdx.y, t = threadIdx.x;
    const uint32_t lo = blockIdx.x * MSM_ORDER_SLICE, hi = min(M, lo + MSM_ORDER_SLICE);
    cnt[t] = 0;
    __syncthreads();
    for (uint32_t b = lo + t; b < hi; b += 1024) atomicAdd(&cnt[order_key(hist[(size_t)w * M + b])], 1u);
    __syncthreads();
    if (cnt[t]) atomicAdd(&keycnt[(size_t)w * 1024 + t], cnt[t]);
}
// keycnt[w][key] -> exclusive start of that key in the window's order array (in place)
__global__ __launch_bounds__(1024) void msm_order_scan_kernel(uint32_t* __restrict__ keycnt) {
    __shared__ uint32_t part[1024];
    const uint32_t w = blockIdx.x, t = threadIdx.x;
    const uint32_t mine = keycnt[(size_t)w * 1024 + t];
    part[t] = mine;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        uint32_t v = t >= (uint32_t)d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    keycnt[(size_t)w * 1024 + t] = part[t] - mine;
}
// each slice reserves, per key, a range of the window's order array and fills it
__global__ __launch_bounds__(1024) void msm_order_scatter_kernel(const uint32_t* __restrict__ hist, uint32_t M, uint32_t* __restrict__ keycur,
                                                                 uint32_t* __restrict__ order) {
    __shared__ uint32_t cnt[1024];
    const uint32_t w = blockIdx.y, t = threadIdx.x;
    const uint32_t lo = blockIdx.x * MSM_ORDER_SLICE, hi = min(M, lo + MSM_ORDER_SLICE);
    cnt[t] = 0;
    __syncthreads();
    for (uint32_t b = lo + t; b < hi; b += 1024) atomicAdd(&cnt[order_key(hist[(size_t)w * M + b])], 1u);
    __syncthreads();
    const uint32_t mine = cnt[t];
    __syncthreads();
    cnt[t] = mine ? atomicAdd(&keycur[(size_t)w * 1024 + t], mine) : 0u;      // start of this slice's range for key t
    __syncthreads();
    for (uint32_t b = lo + t; b < hi; b += 1024) {
        const uint32_t pos = atomicAdd(&cnt[order_key(hist[(size_t)w * M + b])], 1u);
        order[(size_t)w * M + pos] = b;
    }
}

// The two-level sort's form of the ranking (round 5): msm_order_hist's key counts arrive from the sort itself (pre_fine / pre_huge_scan count
// the keys of their buckets into keycnt), every workgroup scans the 1024 key totals of its window for itself (what msm_order_scan did in a
// launch of its own) and reserves its ranges from a zeroed cursor array -- and, visiting every bucket's count anyway, registers the
// over-long and heavy ones (what msm_long_find_kernel did after the accumulation; NULL desc_count: not here).  Three launches fewer per MSM.
struct LongDesc;
struct HeavyRun;
__device__ __forceinline__ void msm_long_register(uint32_t w, uint32_t b, uint32_t c, uint32_t offs_t, int n_win, uint32_t cap, uint32_t desc_cap,
                                                  LongDesc* __restrict__ desc, uint32_t* __restrict__ desc_count, uint32_t run_cap,
                                                  uint32_t h1_cap, HeavyRun* __restrict__ heavy_runs);
__global__ __launch_bounds__(1024) void msm_order_place_kernel(const uint32_t* __restrict__ hist, const uint32_t* __restrict__ offs, uint32_t M,
                                                               const uint32_t* __restrict__ keycnt, uint32_t* __restrict__ keycur, uint32_t* __restrict__ order,
                                                               int n_win, uint32_t cap, uint32_t desc_cap, LongDesc* __restrict__ desc,
                                                               uint32_t* __restrict__ desc_count, uint32_t run_cap, uint32_t h1_cap, HeavyRun* __restrict__ heavy_runs) {
    __shared__ uint32_t cnt[1024];
    __shared__ uint32_t part[1024];
    const uint32_t w = blockIdx.y, t = threadIdx.x;
    const uint32_t lo = blockIdx.x * MSM_ORDER_SLICE, hi = min(M, lo + MSM_ORDER_SLICE);
    const uint32_t total = keycnt[(size_t)w * 1024 + t];
    part[t] = total;
    cnt[t] = 0;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        uint32_t v = t >= (uint32_t)d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    const uint32_t key_start = part[t] - total;                        // exclusive start of key t in the window's order array
    for (uint32_t b = lo + t; b < hi; b += 1024) atomicAdd(&cnt[order_key(hist[(size_t)w * M + b])], 1u);
    __syncthreads();
    const uint32_t mine = cnt[t];
    __syncthreads();
    cnt[t] = mine ? key_start + atomicAdd(&keycur[(size_t)w * 1024 + t], mine) : 0u;      // start of this slice's range for key t
    __syncthreads();
    for (uint32_t b = lo + t; b < hi; b += 1024) {
        const uint32_t c = hist[(size_t)w * M + b];
        const uint32_t pos = atomicAdd(&cnt[order_key(c)], 1u);
        order[(size_t)w * M + pos] = b;
        if (desc_count && c > cap) msm_long_register(w, b, c, offs[(size_t)w * M + b], n_win, cap, desc_cap, desc, desc_count, run_cap, h1_cap, heavy_runs);
    }
}

template <class FQ>
__device__ __forceinline__ Affine<Fp<FQ>> load_affine(const uint32_t* __restrict__ bases, unsigned long long idx) {
    Affine<Fp<FQ>> p;
    const uint32_t* src = bases + idx * (2 * FQ::N);
    p.x = load_fp<FQ>(src);
    p.y = load_fp<FQ>(src + FQ::N);
    return p;
}
template <class FQ>
__device__ __forceinline__ XYZZ<Fp<FQ>> load_xyzz(const uint32_t* __restrict__ buf, unsigned long long idx) {
    XYZZ<Fp<FQ>> p;
    const uint32_t* src = buf + idx * (4 * FQ::N);
    p.x = load_fp<FQ>(src);
    p.y = load_fp<FQ>(src + FQ::N);
    p.zz = load_fp<FQ>(src + 2 * FQ::N);
    p.zzz = load_fp<FQ>(src + 3 * FQ::N);
    return p;
}
template <class FQ>
__device__ __forceinline__ void store_xyzz(uint32_t* __restrict__ buf, unsigned long long idx, const XYZZ<Fp<FQ>>& p) {
    uint32_t* dst = buf + idx * (4 * FQ::N);
    store_fp<FQ>(dst, p.x);
    store_fp<FQ>(dst + FQ::N, p.y);
    store_fp<FQ>(dst + 2 * FQ::N, p.zz);
    store_fp<FQ>(dst + 3 * FQ::N, p.zzz);
}

// ---- EC back ends -----------------------------------------------------------------------------------
// The bucket kernels are written once against a small policy.  EcFx keeps points in the reduced-radix form of
// fx.cuh / ecx.cuh (one v_mad_u64_u32 per partial product and carry-free additions; 14 limbs for BLS12-381 Fq, 10 for
// BN254 Fq), with the SRS converted once at registration and results converted back when they are collected: it is
// what the library runs.  EcFp keeps points in the boundary's 32-bit-limb Montgomery form; it stays as the baseline
// that tools/madd_bench.hip compares against.
template <class FQ>
struct EcFp {
    using Field = FQ;
    using Aff = Affine<Fp<FQ>>;
    using Pt = XYZZ<Fp<FQ>>;
    static constexpr int AFF_WORDS = 2 * FQ::N, PT_WORDS = 4 * FQ::N;
    static __device__ __forceinline__ Aff load_aff(const uint32_t* __restrict__ bases, unsigned long long idx) { return load_affine<FQ>(bases, idx); }
    static __device__ __forceinline__ Pt load_pt(const uint32_t* __restrict__ buf, unsigned long long idx) { return load_xyzz<FQ>(buf, idx); }
    static __device__ __forceinline__ void store_pt(uint32_t* __restrict__ buf, unsigned long long idx, const Pt& p) { store_xyzz<FQ>(buf, idx, p); }
    static __device__ __forceinline__ Pt inf() { return Pt::inf(); }
    static __device__ __forceinline__ Pt madd(const Pt& a, Aff q, bool negate) {
        if (negate) q.y = neg(q.y);
        return xyzz_madd(a, q);
    }
    static __device__ __forceinline__ Pt add(const Pt& a, const Pt& b) { return xyzz_add(a, b); }
    static __device__ __forceinline__ XYZZ<Fp<FQ>> to_boundary(const Pt& p) { return p; }
};

template <class X>
__device__ __forceinline__ void load_words(uint32_t* dst, const uint32_t* __restrict__ src, int n_words) {
    // n_words is a multiple of 4 and src is 16-byte aligned
    for (int i = 0; i < n_words / 4; i++) {
        const uint4 v = reinterpret_cast<const uint4*>(src)[i];
        dst[4 * i] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;
    }
}

template <class X>
struct EcFx {
    using Field = X;
    using Aff = AffineX<X>;
    using Pt = XYZZX<X>;
    static constexpr int AFF_WORDS = 2 * X::XN, PT_WORDS = 4 * X::XN;      // BLS12-381: 28 / 56 words; BN254: 18 / 36
    static_assert((4 * X::XN) % 4 == 0 && (2 * X::XN) % 2 == 0, "points are whole 16-byte / 8-byte words");
    static __device__ __forceinline__ Aff load_aff(const uint32_t* __restrict__ bases, unsigned long long idx) {
        uint32_t w[AFF_WORDS];
        if constexpr (AFF_WORDS % 4 == 0) {
            const uint4* src = reinterpret_cast<const uint4*>(bases + idx * AFF_WORDS);
#pragma unroll
            for (int i = 0; i < AFF_WORDS / 4; i++) {
                const uint4 v = src[i];
                w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
            }
        } else {                                                           // 18 words (BN254 on 9 limbs): 72-byte points, 8-byte aligned
            const uint2* src = reinterpret_cast<const uint2*>(bases + idx * AFF_WORDS);
#pragma unroll
            for (int i = 0; i < AFF_WORDS / 2; i++) {
                const uint2 v = src[i];
                w[2 * i] = v.x; w[2 * i + 1] = v.y;
            }
        }
        Aff p;
#pragma unroll
        for (int i = 0; i < X::XN; i++) { p.x.l[i] = w[i]; p.y.l[i] = w[X::XN + i]; }
        return p;
    }
    static __device__ __forceinline__ Pt load_pt(const uint32_t* __restrict__ buf, unsigned long long idx) {
        uint32_t w[PT_WORDS];
        const uint4* src = reinterpret_cast<const uint4*>(buf + idx * PT_WORDS);
#pragma unroll
        for (int i = 0; i < PT_WORDS / 4; i++) {
            const uint4 v = src[i];
            w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
        }
        Pt p;
#pragma unroll
        for (int i = 0; i < X::XN; i++) {
            p.x.l[i] = w[i]; p.y.l[i] = w[X::XN + i]; p.zz.l[i] = w[2 * X::XN + i]; p.zzz.l[i] = w[3 * X::XN + i];
        }
        return p;
    }
    static __device__ __forceinline__ void store_pt(uint32_t* __restrict__ buf, unsigned long long idx, const Pt& p) {
        uint32_t w[PT_WORDS];
#pragma unroll
        for (int i = 0; i < X::XN; i++) {
            w[i] = p.x.l[i]; w[X::XN + i] = p.y.l[i]; w[2 * X::XN + i] = p.zz.l[i]; w[3 * X::XN + i] = p.zzz.l[i];
        }
        uint4* dst = reinterpret_cast<uint4*>(buf + idx * PT_WORDS);
#pragma unroll
        for (int i = 0; i < PT_WORDS / 4; i++) dst[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
    }
    static __device__ __forceinline__ Pt inf() { return Pt::inf(); }
    static __device__ __forceinline__ Pt madd(const Pt& a, const Aff& q, bool negate) { return xyzzx_madd(a, q, negate); }
    static __device__ __forceinline__ Pt add(const Pt& a, const Pt& b) { return xyzzx_add(a, b); }
    static __device__ __forceinline__ XYZZ<Fp<X>> to_boundary(const Pt& p) { return xyzzx_to_boundary(p); }
};

// boundary SRS (packed x||y, R-form) -> internal affine table of EcFx (x||y, 29-bit limbs, R'-form)
template <class X>
__global__ __launch_bounds__(MSM_THREADS) void srs_to_internal_kernel(const uint32_t* __restrict__ xy, unsigned long long n, uint32_t* __restrict__ out) {
    const unsigned long long i = (unsigned long long)blockIdx.x * MSM_THREADS + threadIdx.x;
    if (i >= n) return;
    const Fp<X> x = load_fp<X>(xy + i * 2 * X::N), y = load_fp<X>(xy + i * 2 * X::N + X::N);
    const Fx<X> fx = fx_from_boundary<X>(x), fy = fx_from_boundary<X>(y);
    uint32_t* dst = out + i * 2 * X::XN;
#pragma unroll
    for (int k = 0; k < X::XN; k++) { dst[k] = fx.l[k]; dst[X::XN + k] = fy.l[k]; }
}

// one thread per (window, bucket): sum of +-P over the bucket's sorted run
template <class EC>
__global__ __launch_bounds__(MSM_ACC_THREADS) void msm_accumulate_kernel(const uint32_t* __restrict__ bases, unsigned long long n,
                                                                          const uint32_t* __restrict__ offs, const uint32_t* __restrict__ hist,
                                                                          const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ order,
                                                                          uint32_t M, int n_win, uint32_t cap, uint32_t* __restrict__ long_count,
                                                                          uint32_t* __restrict__ buckets, uint8_t* __restrict__ occ) {
    if (long_count && blockIdx.x == 0)                         // what msm_long_find_kernel counts into: saves a fill (per window: 1 + MSM_HEAVY_COUNTERS words);
        for (unsigned i = threadIdx.x; i < (unsigned)n_win * (1 + MSM_HEAVY_COUNTERS); i += MSM_ACC_THREADS) long_count[i] = 0u;      // NULL: the sort has cleared and filled them already
    const unsigned long long t0 = (unsigned long long)blockIdx.x * MSM_ACC_THREADS + threadIdx.x;
    if (t0 >= (unsigned long long)n_win * M) return;
    const unsigned long long w = t0 / M;
    const unsigned long long t = w * M + order[t0];            // lanes of a wave take buckets of equal load
    const uint32_t start = offs[t];
    // the rest of an over-long bucket: msm_long_* kernels; ALL of a heavy one: msm_heavy_* (a lone chain of `cap` additions would
    // outlast the whole launch)
    const uint32_t h = hist[t];
    const uint32_t cnt = msm_is_heavy(h, cap) ? 0u : min(h, cap);
    const uint32_t* list = sorted + w * n + start;
    typename EC::Pt acc = EC::inf();
    for (uint32_t k = 0; k < cnt; k++) {
        const uint32_t e = list[k];
        acc = EC::madd(acc, EC::load_aff(bases, e & 0x7fffffffu), (e >> 31) != 0);
    }
    // occ[t] == 0: the bucket holds infinity and its 224 bytes are NOT written (nor read by the reduction): a sparse polynomial -- the
    // bench circuit commits two zero wires -- then costs what its few occupied buckets cost, not 2^19 bucket writes and their folding
    const bool empty = acc.is_inf();
    occ[t] = empty ? 0 : 1;
    if (!empty) EC::store_pt(buckets, t, acc);
}

// Small and medium MSMs are latency bound: few buckets, each a chain of dependent mixed adds (~12 us apiece when a wave runs
// alone).  There every bucket is split over S = 2^log_split threads -- thread s takes entries s, s + S, ... of the run -- and
// msm_split_combine_kernel adds the S partial sums as a tree: chains S times shorter for (S - 1) M extra additions, log_split deep.
//
// LARGE plain-path MSMs, round 5 (last session): only the TAIL is split.  The buckets of rank < rank0 (ranks are window-major, by descending
// load within a window: `order`) take one thread each and write their bucket directly; the buckets of rank >= rank0 -- what the launch
// dispatches last -- are split.  With several bucket sets the last ranks are a whole window, light buckets AND heavy ones: a launch of
// whole-bucket threads ends with the chip draining for the length of its chains (2.65-2.77 ms at 2^20 pairs), halving EVERY bucket costs an
// addition per bucket (2.35 + 0.13 ms), splitting the last eighth of the ranks four ways 2.36 + 0.07.  (One bucket set -- the table path --
// is ranked as a whole, so its launch ends on its lightest buckets and gains nothing: not used there.)  rank0 = 0: every bucket is split.
// `sub` is laid out by RANK: the partial sums of the bucket of rank rank0 + r are sub[r S .. r S + S).
// ONE_SET: one bucket set (the table path); it also gives the two paths' launches different names for the profiler.
template <class EC, bool ONE_SET>
__global__ __launch_bounds__(MSM_ACC_THREADS) void msm_accumulate_split_kernel(const uint32_t* __restrict__ bases, unsigned long long n,
                                                                                const uint32_t* __restrict__ offs, const uint32_t* __restrict__ hist,
                                                                                const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ order,
                                                                                uint32_t M, int n_win, uint32_t cap, int log_split, uint32_t* __restrict__ long_count,
                                                                                uint32_t* __restrict__ sub, unsigned long long rank0,
                                                                                uint32_t* __restrict__ buckets, uint8_t* __restrict__ occ) {
    if (long_count && blockIdx.x == 0)
        for (unsigned i = threadIdx.x; i < (unsigned)n_win * (1 + MSM_HEAVY_COUNTERS); i += MSM_ACC_THREADS) long_count[i] = 0u;
    const unsigned long long t1 = (unsigned long long)blockIdx.x * MSM_ACC_THREADS + threadIdx.x;
    const bool whole = t1 < rank0;                                     // (rank0 is a multiple of the wave size: a wave is of one kind)
    const unsigned long long u = whole ? 0ull : t1 - rank0;
    const unsigned long long t0 = whole ? t1 : rank0 + (u >> log_split);
    if (t0 >= (unsigned long long)n_win * M) return;
    const uint32_t S = whole ? 1u : 1u << log_split, part = whole ? 0u : (uint32_t)(u & (S - 1));
    const unsigned long long w = ONE_SET ? 0ull : t0 / M;
    const unsigned long long t = w * M + order[t0];
    const uint32_t start = offs[t];
    const uint32_t h = hist[t];
    const uint32_t cnt = msm_is_heavy(h, cap) ? 0u : min(h, cap);
    const uint32_t* list = sorted + w * n + start;
    typename EC::Pt acc = EC::inf();
    for (uint32_t k = part; k < cnt; k += S) {
        const uint32_t e = list[k];
        acc = EC::madd(acc, EC::load_aff(bases, e & 0x7fffffffu), (e >> 31) != 0);
    }
    if (whole) {
        const bool empty = acc.is_inf();
        occ[t] = empty ? 0 : 1;
        if (!empty) EC::store_pt(buckets, t, acc);
    } else {
        EC::store_pt(sub, u, acc);
    }
}
// A workgroup takes 2 * MSM_ACC_THREADS consecutive partial sums (2 * MSM_ACC_THREADS / S whole buckets).  Level 0: every thread adds
// one adjacent pair; level l: the first MSM_ACC_THREADS >> l threads add the pairs of the level before, which they find in LDS
// (word-major, so lanes touch consecutive banks) -- the active lanes stay dense and idle waves skip the addition altogether.
template <class EC>
__device__ __forceinline__ void pt_lds_put(uint32_t* lds, int slot, const typename EC::Pt& p) {
    constexpr int N = EC::Field::XN;
#pragma unroll
    for (int i = 0; i < N; i++) {
        lds[i * MSM_ACC_THREADS + slot] = p.x.l[i];
        lds[(N + i) * MSM_ACC_THREADS + slot] = p.y.l[i];
        lds[(2 * N + i) * MSM_ACC_THREADS + slot] = p.zz.l[i];
        lds[(3 * N + i) * MSM_ACC_THREADS + slot] = p.zzz.l[i];
    }
}
template <class EC>
__device__ __forceinline__ typename EC::Pt pt_lds_get(const uint32_t* lds, int slot) {
    constexpr int N = EC::Field::XN;
    typename EC::Pt p;
#pragma unroll
    for (int i = 0; i < N; i++) {
        p.x.l[i] = lds[i * MSM_ACC_THREADS + slot];
        p.y.l[i] = lds[(N + i) * MSM_ACC_THREADS + slot];
        p.zz.l[i] = lds[(2 * N + i) * MSM_ACC_THREADS + slot];
        p.zzz.l[i] = lds[(3 * N + i) * MSM_ACC_THREADS + slot];
    }
    return p;
}
// (the bucket of tail rank r is the one of rank rank0 + r: window-major `order`, as in the accumulation)
template <class EC, bool ONE_SET>
__global__ __launch_bounds__(MSM_ACC_THREADS) void msm_split_combine_kernel(const uint32_t* __restrict__ sub, unsigned long long n_buckets, int log_split,
                                                                             const uint32_t* __restrict__ order, uint32_t M, unsigned long long rank0,
                                                                             uint32_t* __restrict__ buckets, uint8_t* __restrict__ occ) {
    __shared__ uint32_t lds[MSM_ACC_THREADS * EC::PT_WORDS];
    const int tid = threadIdx.x;
    const unsigned long long total = n_buckets << log_split;                 // partial sums in all (even)
    const unsigned long long base = (unsigned long long)blockIdx.x * (2 * MSM_ACC_THREADS);
    typename EC::Pt acc = EC::inf();
    // ONE call site of EC::add for all levels: a second copy of its ~50 KB of straight-line code would evict the first from the
    // 64 KB instruction cache, and a lone wave then waits for every line of it
#pragma unroll 1
    for (int lvl = 0; lvl < log_split; lvl++) {
        const bool mine = tid < (MSM_ACC_THREADS >> lvl) && base + ((unsigned long long)(2 * tid) << lvl) < total;
        typename EC::Pt a = EC::inf(), b = EC::inf();
        if (lvl == 0) {
            if (mine) { a = EC::load_pt(sub, base + 2ull * tid); b = EC::load_pt(sub, base + 2ull * tid + 1); }
        } else {
            __syncthreads();
            if (mine) { a = pt_lds_get<EC>(lds, 2 * tid); b = pt_lds_get<EC>(lds, 2 * tid + 1); }
            __syncthreads();
        }
        if (mine) {
            acc = EC::add(a, b);
            if (lvl == log_split - 1) {
                const unsigned long long t0 = rank0 + (base >> log_split) + tid, w = ONE_SET ? 0ull : t0 / M, t = w * M + order[t0];
                const bool empty = acc.is_inf();
                occ[t] = empty ? 0 : 1;
                if (!empty) EC::store_pt(buckets, t, acc);
            } else pt_lds_put<EC>(lds, tid, acc);
        }
    }
}

// ---- over-long buckets (skewed scalars) ------------------------------------------------------------
// A bucket with more than `cap` points would serialise one thread for its whole run (all-equal
// scalars put n points into one bucket per window).  Its first `cap` points stay with the regular
// accumulation; the remainder is cut into chunks of `cap`, each summed by its own thread, and the
// chunk sums are tree-reduced per bucket.  With uniformly random scalars no bucket is long and the
// three kernels below exit at once.
struct LongDesc { uint32_t bucket, start, len, idx_in_run, run_len; };

// one thread per (window, bucket): a long bucket reserves its run of descriptors with one atomic
// (desc_count must be zeroed first); runs of different buckets land in arrival order, each contiguous
//
// HEAVY buckets (round 3): more than MSM_HEAVY_RUN entries beyond the cap -- all-equal scalars, the short top digit of small scalars
// (witness VALUES are often 8 / 32 / 64-bit numbers), a handful of buckets holding most of the points.  One thread per chunk of `cap`
// and ONE workgroup per window for the tree (below) leaves the chip idle there.  A heavy bucket's remainder is cut into level-1 runs of
// MSM_HEAVY_RUN entries (ALL of them: the regular accumulation skips a heavy bucket -- a lone chain of `cap` dependent additions
// would outlast the launch): a workgroup per run, MSM_HEAVY_PER_THREAD mixed adds per thread (msm_heavy_chunk_kernel), then workgroup
// trees over the 128 partial sums of a run and over the runs of a bucket (msm_heavy_reduce_kernel, levels A, B, C: up to
// MSM_HEAVY_FANIN items each), the last of which adds into the bucket.  dest: bit 31 set = slot of the next level's parts.
struct HeavyRun { uint32_t src, n, dest, p1; };           // p1 (level-1 runs): where the run's partial sums start in h1
constexpr uint32_t MSM_HEAVY_DEST_PART = 0x80000000u;
// the heavy kernels of the MSMs of ONE batch run as one launch per level (blockIdx.z = MSM): their chains of a few dozen dependent
// additions are latency, and five of them in sequence were 1.6 of the 3.6 ms of a round 1 committed from small witness values
constexpr int MSM_HEAVY_JOBS = 8;
struct HeavyJob {
    const uint32_t* bases;
    const uint32_t* sorted;
    unsigned long long n;              // stride between the windows' lists (0: one combined list)
    const HeavyRun* runs;
    const uint32_t* count;
    uint32_t *h1, *h2, *h3, *buckets;
    uint8_t* occ;
    uint32_t run_cap, h1_cap, M;       // runs / level-1 partial sums per window
    // the over-long buckets of the same MSM (round 5: their chunk sums run in the launch of the heavy level-1 sums, msm_rare_leaf_kernel)
    const LongDesc* desc;
    const uint32_t* desc_count;
    uint32_t* parts;
    uint32_t desc_cap;
};
struct HeavyJobs { HeavyJob j[MSM_HEAVY_JOBS]; };
// counters of window w at heavy_count + w * MSM_HEAVY_COUNTERS: [0] level-1 runs, [1] level-B runs, [2] level-C runs, [3] parts of level 2, [4] of level 3,
// [5] level-1 partial sums (one per MSM_HEAVY_PER_THREAD entries of a run)
__device__ __forceinline__ void msm_heavy_push(uint32_t b, uint32_t start, uint32_t rem, uint32_t run_cap, uint32_t h1_cap, uint32_t* __restrict__ cnt,
                                               HeavyRun* __restrict__ runs0, HeavyRun* __restrict__ runsB, HeavyRun* __restrict__ runsC) {
    const uint32_t n1 = (rem + MSM_HEAVY_RUN - 1) / MSM_HEAVY_RUN;
    // the level-1 partial sums of this bucket: a range of the window's h1 array.  The array is sized OPTIMISTICALLY (msm.hip: a fraction of
    // the worst case "every entry sits in a heavy bucket", remembered per device context): a bucket that does not fit registers nothing, the
    // counter keeps counting what WOULD have been needed, and the host -- which reads the counters with the results -- grows the array and
    // runs the group again.  Round 5: 1.33 GB of scratch for five 2^20-pair MSMs -> what their heavy buckets really hold.
    const uint32_t need = (rem + MSM_HEAVY_PER_THREAD - 1) / MSM_HEAVY_PER_THREAD + n1;                     // (rounding slack: one per run)
    const uint32_t p1 = atomicAdd(&cnt[5], need);
    if (p1 + need > h1_cap) return;
    const uint32_t r0 = atomicAdd(&cnt[0], n1);
    if (r0 + n1 > run_cap) return;                                     // cannot happen: run_cap >= 2 * entries / MSM_HEAVY_RUN + 2 (a heavy bucket holds more than MSM_HEAVY_RUN entries)
    constexpr uint32_t PER_RUN = MSM_HEAVY_RUN / MSM_HEAVY_PER_THREAD;
    if (n1 == 1) { runs0[r0] = HeavyRun{start, rem, b, p1}; return; }
    const uint32_t p2 = atomicAdd(&cnt[3], n1);
    for (uint32_t i = 0; i < n1; i++)
        runs0[r0 + i] = HeavyRun{start + i * MSM_HEAVY_RUN, min(MSM_HEAVY_RUN, rem - i * MSM_HEAVY_RUN), MSM_HEAVY_DEST_PART | (p2 + i), p1 + i * PER_RUN};
    const uint32_t n2 = (n1 + MSM_HEAVY_FANIN - 1) / MSM_HEAVY_FANIN;
    const uint32_t rb = atomicAdd(&cnt[1], n2);
    if (n2 == 1) { runsB[rb] = HeavyRun{p2, n1, b, 0}; return; }
    const uint32_t p3 = atomicAdd(&cnt[4], n2);
    for (uint32_t j = 0; j < n2; j++)
        runsB[rb + j] = HeavyRun{p2 + j * MSM_HEAVY_FANIN, min(MSM_HEAVY_FANIN, n1 - j * MSM_HEAVY_FANIN), MSM_HEAVY_DEST_PART | (p3 + j), 0};
    runsC[atomicAdd(&cnt[2], 1u)] = HeavyRun{p3, n2, b, 0};
}

// bucket b of window w holds c > cap entries from offs_t on: descriptors of its chunks (over-long) or runs (heavy)
__device__ __forceinline__ void msm_long_register(uint32_t w, uint32_t b, uint32_t c, uint32_t offs_t, int n_win, uint32_t cap, uint32_t desc_cap,
                                                  LongDesc* __restrict__ desc, uint32_t* __restrict__ desc_count, uint32_t run_cap,
                                                  uint32_t h1_cap, HeavyRun* __restrict__ heavy_runs);

__global__ __launch_bounds__(256) void msm_long_find_kernel(const uint32_t* __restrict__ hist, const uint32_t* __restrict__ offs, uint32_t M, int n_win,
                                                            uint32_t cap, uint32_t desc_cap, LongDesc* __restrict__ desc,
                                                            uint32_t* __restrict__ desc_count, uint32_t run_cap, uint32_t h1_cap, HeavyRun* __restrict__ heavy_runs) {
    const unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (unsigned long long)n_win * M) return;
    const uint32_t c = hist[t];
    if (c <= cap) return;
    msm_long_register((uint32_t)(t / M), (uint32_t)(t % M), c, offs[t], n_win, cap, desc_cap, desc, desc_count, run_cap, h1_cap, heavy_runs);
}
__device__ __forceinline__ void msm_long_register(uint32_t w, uint32_t b, uint32_t c, uint32_t offs_t, int n_win, uint32_t cap, uint32_t desc_cap,
                                                  LongDesc* __restrict__ desc, uint32_t* __restrict__ desc_count, uint32_t run_cap,
                                                  uint32_t h1_cap, HeavyRun* __restrict__ heavy_runs) {
    if (msm_is_heavy(c, cap)) {                                        // heavy: its own kernels take ALL its entries
        HeavyRun* base = heavy_runs + (size_t)w * 3 * run_cap;
        msm_heavy_push(b, offs_t, c, run_cap, h1_cap, desc_count + n_win + (size_t)w * MSM_HEAVY_COUNTERS, base, base + run_cap, base + 2 * (size_t)run_cap);
        return;
    }
    const uint32_t nch = (c - cap + cap - 1) / cap;
    const uint32_t pos = atomicAdd(&desc_count[w], nch);
    if (pos + nch > desc_cap) return;                        // optimistic size (msm.hip): all of the bucket's chunks or none -- the host sees desc_count > desc_cap and runs the group again
    for (uint32_t j = 0; j < nch; j++) {
        LongDesc d;
        d.bucket = b;
        d.start = offs_t + cap * (j + 1);
        d.len = min(cap, c - cap * (j + 1));
        d.idx_in_run = j;
        d.run_len = nch;
        desc[(size_t)w * desc_cap + pos + j] = d;
    }
}

// one thread per chunk descriptor
template <class EC>
__global__ __launch_bounds__(MSM_ACC_THREADS) void msm_long_chunk_kernel(const uint32_t* __restrict__ bases, unsigned long long n,
                                                                          const uint32_t* __restrict__ sorted, const LongDesc* __restrict__ desc,
                                                                          const uint32_t* __restrict__ desc_count, uint32_t desc_cap,
                                                                          uint32_t* __restrict__ parts) {
    const uint32_t w = blockIdx.y;
    const uint32_t i = blockIdx.x * MSM_ACC_THREADS + threadIdx.x;
    if (i >= min(desc_count[w], desc_cap)) return;
    const LongDesc d = desc[(size_t)w * desc_cap + i];
    const uint32_t* list = sorted + (size_t)w * n + d.start;
    typename EC::Pt acc = EC::inf();
    for (uint32_t k = 0; k < d.len; k++) {
        const uint32_t e = list[k];
        acc = EC::madd(acc, EC::load_aff(bases, e & 0x7fffffffu), (e >> 31) != 0);
    }
    EC::store_pt(parts, (size_t)w * desc_cap + i, acc);
}

// The leaf sums of BOTH rare paths in one launch (round 5; bit-identical to the two kernels above and below, which it replaces in
// msm.hip): workgroups [0, long_blocks) take a chunk descriptor per thread (msm_long_chunk_kernel), the others a heavy level-1 run per
// workgroup, MSM_HEAVY_PER_THREAD entries per thread (msm_heavy_chunk_kernel).  Either way a thread sums `cnt` consecutive entries of the
// sorted list into one partial sum: ONE loop, one call site of EC::madd.  blockIdx.y = window, blockIdx.z = MSM of the batch.
template <class EC>
__global__ __launch_bounds__(MSM_ACC_THREADS) void msm_rare_leaf_kernel(HeavyJobs jobs, uint32_t long_blocks) {
    const HeavyJob& jb = jobs.j[blockIdx.z];
    const uint32_t w = blockIdx.y, t = threadIdx.x;
    const bool is_long = blockIdx.x < long_blocks;
    const uint32_t heavy_cnt = is_long ? 0u : min(jb.count[(size_t)w * MSM_HEAVY_COUNTERS], jb.run_cap);
    const HeavyRun* runs = jb.runs + (size_t)w * 3 * jb.run_cap;
    const uint32_t r_step = is_long ? 1u : gridDim.x - long_blocks;
    uint32_t r = is_long ? 0u : blockIdx.x - long_blocks;
    for (;; r += r_step) {
        const uint32_t* list;
        uint32_t cnt;
        uint32_t* out;
        size_t out_idx;
        if (is_long) {
            if (r) break;                                              // one descriptor per thread
            const uint32_t i = blockIdx.x * MSM_ACC_THREADS + t;
            if (i >= min(jb.desc_count[w], jb.desc_cap)) break;
            const LongDesc d = jb.desc[(size_t)w * jb.desc_cap + i];
            list = jb.sorted + (size_t)w * jb.n + d.start;
            cnt = d.len;
            out = jb.parts;
            out_idx = (size_t)w * jb.desc_cap + i;
        } else {
            if (r >= heavy_cnt) break;
            const HeavyRun run = runs[r];
            const uint32_t lo = t * MSM_HEAVY_PER_THREAD;
            if (lo >= run.n) continue;
            list = jb.sorted + (size_t)w * jb.n + run.src + lo;
            cnt = min(run.n, lo + MSM_HEAVY_PER_THREAD) - lo;
            out = jb.h1;
            out_idx = (size_t)w * jb.h1_cap + run.p1 + t;
        }
        typename EC::Pt acc = EC::inf();
        for (uint32_t k = 0; k < cnt; k++) {
            const uint32_t e = list[k];
            acc = EC::madd(acc, EC::load_aff(jb.bases, e & 0x7fffffffu), (e >> 31) != 0);
        }
        EC::store_pt(out, out_idx, acc);
    }
}

// one thread per long bucket: its run of chunk sums (at most MSM_HEAVY_RUN / cap + 1 of them: longer buckets are heavy, see above) is added
// to the bucket one after the other -- chains of a few additions, in parallel over the runs.  (Until round 3 ONE workgroup per window
// walked a pairwise tree over all runs: 3 - 30 ms once tens of thousands of buckets were over-long.)  ONE call site of EC::add.
template <class EC>
__global__ __launch_bounds__(MSM_ACC_THREADS) void msm_long_combine_kernel(HeavyJobs jobs) {
    const HeavyJob& jb = jobs.j[blockIdx.z];
    const LongDesc* __restrict__ desc = jb.desc;
    const uint32_t* __restrict__ desc_count = jb.desc_count;
    const uint32_t desc_cap = jb.desc_cap, M = jb.M;
    const uint32_t* __restrict__ parts = jb.parts;
    uint32_t* __restrict__ buckets = jb.buckets;
    uint8_t* __restrict__ occ = jb.occ;
    const uint32_t w = blockIdx.y;
    const uint32_t cnt = min(desc_count[w], desc_cap);
    const LongDesc* dw = desc + (size_t)w * desc_cap;
    const size_t pbase = (size_t)w * desc_cap;
    for (uint32_t i = blockIdx.x * MSM_ACC_THREADS + threadIdx.x; i < cnt; i += gridDim.x * MSM_ACC_THREADS) {
        const LongDesc d = dw[i];
        if (d.idx_in_run != 0) continue;
        const size_t bi = (size_t)w * M + d.bucket;
        typename EC::Pt acc = occ[bi] ? EC::load_pt(buckets, bi) : EC::inf();       // (the first `cap` points may sum to infinity)
        for (uint32_t j = 0; j < d.run_len; j++) acc = EC::add(acc, EC::load_pt(parts, pbase + i + j));
        const bool empty = acc.is_inf();
        occ[bi] = empty ? 0 : 1;
        if (!empty) EC::store_pt(buckets, bi, acc);
    }
}

// level 1 of a heavy bucket: one workgroup per run of <= MSM_HEAVY_RUN entries, thread t sums its MSM_HEAVY_PER_THREAD entries -> h1[run.p1 + t]
template <class EC>
__global__ __launch_bounds__(MSM_ACC_THREADS) void msm_heavy_chunk_kernel(HeavyJobs jobs) {
    const HeavyJob& jb = jobs.j[blockIdx.z];
    const uint32_t w = blockIdx.y, t = threadIdx.x, run_cap = jb.run_cap;
    const uint32_t cnt = min(jb.count[(size_t)w * MSM_HEAVY_COUNTERS], run_cap);
    const HeavyRun* runs = jb.runs + (size_t)w * 3 * run_cap;
    for (uint32_t r = blockIdx.x; r < cnt; r += gridDim.x) {
        const HeavyRun run = runs[r];
        const uint32_t lo = t * MSM_HEAVY_PER_THREAD;
        if (lo >= run.n) continue;
        const uint32_t hi = min(run.n, lo + MSM_HEAVY_PER_THREAD);
        const uint32_t* list = jb.sorted + (size_t)w * jb.n + run.src;
        typename EC::Pt acc = EC::inf();
        for (uint32_t k = lo; k < hi; k++) {
            const uint32_t e = list[k];
            acc = EC::madd(acc, EC::load_aff(jb.bases, e & 0x7fffffffu), (e >> 31) != 0);
        }
        EC::store_pt(jb.h1, (size_t)w * jb.h1_cap + run.p1 + t, acc);
    }
}
// levels A (the 128 partial sums of a level-1 run), B and C (the runs of a bucket): a workgroup per run; a thread first sums the items
// t, t + 128, .. of the run, then the 128 sums go through an LDS tree; the last level of a bucket stores into it (the regular
// accumulation leaves a heavy bucket empty).  ONE call site of EC::add (see msm_split_combine_kernel).
template <class EC>
__global__ __launch_bounds__(MSM_ACC_THREADS) void msm_heavy_reduce_kernel(HeavyJobs jobs, int level /* 0 = A, 1 = B, 2 = C */) {
    __shared__ uint32_t lds[MSM_ACC_THREADS * EC::PT_WORDS];
    const HeavyJob& jb = jobs.j[blockIdx.z];
    const uint32_t w = blockIdx.y, run_cap = jb.run_cap, M = jb.M;
    const int tid = threadIdx.x;
    const uint32_t* parts_in = level == 0 ? jb.h1 : (level == 1 ? jb.h2 : jb.h3);
    uint32_t* parts_out = level == 0 ? jb.h2 : jb.h3;
    uint32_t* buckets = jb.buckets;
    uint8_t* occ = jb.occ;
    const uint32_t cnt = min(jb.count[(size_t)w * MSM_HEAVY_COUNTERS + level], run_cap);
    const HeavyRun* runs = jb.runs + ((size_t)w * 3 + level) * run_cap;
    const size_t in_base = (size_t)w * (level == 0 ? jb.h1_cap : run_cap), out_base = (size_t)w * run_cap;
    for (uint32_t r = blockIdx.x; r < cnt; r += gridDim.x) {
        const HeavyRun run = runs[r];
        const uint32_t n_items = level == 0 ? (run.n + MSM_HEAVY_PER_THREAD - 1) / MSM_HEAVY_PER_THREAD : run.n;
        const size_t src = in_base + (level == 0 ? (size_t)run.p1 : (size_t)run.src);
        const int per = (int)((n_items + MSM_ACC_THREADS - 1) / MSM_ACC_THREADS);          // items per thread (uniform bound)
        int log_w = 0;                                                 // the tree is as wide as the run has sums: 2^log_w >= min(128, n_items)
        while ((1u << log_w) < n_items && log_w < 7) log_w++;
        typename EC::Pt acc = (uint32_t)tid < n_items ? EC::load_pt(parts_in, src + tid) : EC::inf();
        const bool to_bucket = !(run.dest & MSM_HEAVY_DEST_PART);
        const size_t bi = (size_t)w * M + run.dest;
        const int n_steps = (per - 1) + log_w;
#pragma unroll 1
        for (int s = 0; s < n_steps; s++) {
            typename EC::Pt a = EC::inf(), b = EC::inf();
            bool act = false;
            if (s < per - 1) {                                         // the thread's own items, one after the other
                const uint32_t q = (uint32_t)tid + (uint32_t)(s + 1) * MSM_ACC_THREADS;
                if (q < n_items) { a = acc; b = EC::load_pt(parts_in, src + q); act = true; }
            } else {                                                   // tree level l: thread t < 64 >> l adds the sums of threads 2t and 2t + 1
                const int l = s - (per - 1);
                __syncthreads();
                if (tid < ((1 << log_w) >> l)) pt_lds_put<EC>(lds, tid, acc);
                __syncthreads();
                if (tid < ((1 << log_w) >> (l + 1))) { a = pt_lds_get<EC>(lds, 2 * tid); b = pt_lds_get<EC>(lds, 2 * tid + 1); act = true; }
            }
            if (act) acc = EC::add(a, b);
        }
        if (tid == 0) {
            if (to_bucket) {
                const bool empty = acc.is_inf();
                occ[bi] = empty ? 0 : 1;
                if (!empty) EC::store_pt(buckets, bi, acc);
            } else {
                EC::store_pt(parts_out, out_base + (run.dest & ~MSM_HEAVY_DEST_PART), acc);
            }
        }
    }
}

// level with half-size h: segment 0 is the main array (base 0), segment j >= 1 is T_j (base M>>j);
// X[base+i] += X[base+h+i].  grid covers n_win * nseg * h threads.  occ[] (one byte per bucket slot, set by the accumulation) says
// which slots hold a point: an empty right operand costs nothing, an empty left one a copy -- the reduction of a sparse bucket set
// moves and adds only what is there.
template <class EC>
__device__ __forceinline__ void fold_one(uint32_t* __restrict__ buckets, uint8_t* __restrict__ occ, unsigned long long ia, unsigned long long ib) {
    if (!occ[ib]) return;
    typename EC::Pt b = EC::load_pt(buckets, ib);
    if (!occ[ia]) {
        EC::store_pt(buckets, ia, b);
        occ[ia] = 1;
        return;
    }
    typename EC::Pt a = EC::load_pt(buckets, ia);
    EC::store_pt(buckets, ia, EC::add(a, b));
}
template <class EC>
__global__ __launch_bounds__(MSM_ACC_THREADS) void msm_fold_kernel(uint32_t* __restrict__ buckets, uint8_t* __restrict__ occ, uint32_t M, uint32_t h, int nseg, int n_win) {
    const unsigned long long t = (unsigned long long)blockIdx.x * MSM_ACC_THREADS + threadIdx.x;
    const unsigned long long per_win = (unsigned long long)nseg * h;
    if (t >= per_win * n_win) return;
    const unsigned long long w = t / per_win, r = t % per_win;
    const uint32_t seg = (uint32_t)(r / h), i = (uint32_t)(r % h);
    const unsigned long long base = w * M + (seg ? (M >> seg) : 0u);
    fold_one<EC>(buckets, occ, base + i, base + h + i);
}

// TWO levels in one launch (round 5): half sizes h (level nseg) and h / 2 (level nseg + 1).  A thread owns the items i and i + h / 2 of one
// segment: both folds of the first level, then the second level's fold of their sums; the thread of the MAIN segment also folds the segment
// the first level leaves behind (the main array's upper half, T_nseg: its two sources are the operands this thread has just read).
// Nothing a thread touches is touched by another.  h even.
template <class EC>
__global__ __launch_bounds__(MSM_ACC_THREADS) void msm_fold2_kernel(uint32_t* __restrict__ buckets, uint8_t* __restrict__ occ, uint32_t M, uint32_t h, int nseg, int n_win) {
    const uint32_t q = h >> 1;
    const unsigned long long t = (unsigned long long)blockIdx.x * MSM_ACC_THREADS + threadIdx.x;
    const unsigned long long per_win = (unsigned long long)nseg * q;
    if (t >= per_win * n_win) return;
    const unsigned long long w = t / per_win, r = t % per_win;
    const uint32_t seg = (uint32_t)(r / q), i = (uint32_t)(r % q);
    const unsigned long long base = w * M + (seg ? (M >> seg) : 0u);
#pragma unroll 1
    for (int step = 0; step < 4; step++) {                              // ONE call site of the addition (instruction cache: msm_split_combine_kernel)
        if (step == 3 && seg) break;
        const unsigned long long ia = base + (step == 1 ? q : (step == 3 ? h : 0u)) + i;
        const unsigned long long ib = step < 2 ? ia + h : ia + q;
        fold_one<EC>(buckets, occ, ia, ib);
    }
}
template <class EC>
__global__ __launch_bounds__(MSM_ACC_THREADS) void msm_fold2_quad_kernel(uint32_t* buckets, uint8_t* occ, uint32_t M, uint32_t h, int nseg, int n_win);

// the last levels of the recursive halving in ONE launch: a 256-thread workgroup per
// bucket set walks the levels with a workgroup barrier between them (each level is one EC add deep,
// so separate launches would pay a launch gap per level for a few hundred threads of work)
// collect (nullable): the workgroup also writes its bucket set's log_m + 1 results (what msm_collect_kernel does in a launch of its own)
template <class EC>
__device__ __forceinline__ void collect_set(const uint32_t* buckets, const uint8_t* occ, uint32_t M, int log_m, unsigned long long w, uint32_t* out, int j) {
    const unsigned long long slot = w * M + (j ? (M >> j) : 0u);
    typename EC::Pt p = occ[slot] ? EC::load_pt(buckets, slot) : EC::inf();
    store_xyzz<typename EC::Field>(out, (size_t)w * (log_m + 1) + j, EC::to_boundary(p));
}
template <class EC>
__global__ __launch_bounds__(256) void msm_fold_tail_kernel(uint32_t* __restrict__ buckets, uint8_t* __restrict__ occ, uint32_t M, int log_m, int first_lvl,
                                                            uint32_t* __restrict__ collect) {
    const unsigned long long w = blockIdx.x;
    for (int lvl = first_lvl; lvl <= log_m; lvl++) {
        const uint32_t h = M >> lvl;
        const uint32_t items = (uint32_t)lvl * h;
        for (uint32_t r = threadIdx.x; r < items; r += 256) {
            const uint32_t seg = r / h, i = r % h;
            const unsigned long long base = w * M + (seg ? (M >> seg) : 0u);
            fold_one<EC>(buckets, occ, base + i, base + h + i);
        }
        __threadfence_block();
        __syncthreads();
    }
    if (collect && (int)threadIdx.x <= log_m) collect_set<EC>(buckets, occ, M, log_m, w, collect, (int)threadIdx.x);
}

// The same levels with FOUR lanes per addition (ecx.cuh, xyzzx_add_quad): lane k of a quad loads, adds and stores coordinate k.  For
// levels with fewer additions than a quarter of the chip's lanes (msm.hip picks per level): such a level lasts as long as ONE addition,
// and a quad finishes one in four product latencies instead of fourteen.
template <class EC>
__device__ __forceinline__ void fold_one_quad(uint32_t* buckets, uint8_t* occ, unsigned long long ia, unsigned long long ib, int role) {
    using X = typename EC::Field;
    constexpr int XN = X::XN;
    if (!occ[ib]) return;                                                 // (quad-uniform, as every branch below)
    uint32_t* pa = buckets + ia * EC::PT_WORDS + role * XN;
    const uint32_t* pb = buckets + ib * EC::PT_WORDS + role * XN;
    Fx<X> cb;
#pragma unroll
    for (int i = 0; i < XN; i++) cb.l[i] = pb[i];
    if (!occ[ia]) {
#pragma unroll
        for (int i = 0; i < XN; i++) pa[i] = cb.l[i];
        occ[ia] = 1;                                                      // every lane of the quad: each then re-reads what IT wrote (msm_fold2_quad_kernel reads it again without a barrier)
        return;
    }
    Fx<X> ca;
#pragma unroll
    for (int i = 0; i < XN; i++) ca.l[i] = pa[i];
    const Fx<X> r = xyzzx_add_quad<X>(ca, cb, role);
#pragma unroll
    for (int i = 0; i < XN; i++) pa[i] = r.l[i];
}
template <class EC>
__global__ __launch_bounds__(MSM_ACC_THREADS) void msm_fold_quad_kernel(uint32_t* buckets, uint8_t* occ, uint32_t M, uint32_t h, int nseg, int n_win) {
    const unsigned long long t = ((unsigned long long)blockIdx.x * MSM_ACC_THREADS + threadIdx.x) >> 2;
    const int role = threadIdx.x & 3;
    const unsigned long long per_win = (unsigned long long)nseg * h;
    if (t >= per_win * n_win) return;
    const unsigned long long w = t / per_win, r = t % per_win;
    const uint32_t seg = (uint32_t)(r / h), i = (uint32_t)(r % h);
    const unsigned long long base = w * M + (seg ? (M >> seg) : 0u);
    fold_one_quad<EC>(buckets, occ, base + i, base + h + i, role);
}
constexpr int MSM_TAIL_QUAD_THREADS = 512;
template <class EC>
__global__ __launch_bounds__(MSM_TAIL_QUAD_THREADS) void msm_fold_tail_quad_kernel(uint32_t* buckets, uint8_t* occ, uint32_t M, int log_m, int first_lvl,
                                                                                    uint32_t* collect) {
    const unsigned long long w = blockIdx.x;
    const uint32_t quad = threadIdx.x >> 2;
    const int role = threadIdx.x & 3;
    for (int lvl = first_lvl; lvl <= log_m; lvl++) {
        const uint32_t h = M >> lvl;
        const uint32_t items = (uint32_t)lvl * h;
        for (uint32_t r = quad; r < items; r += MSM_TAIL_QUAD_THREADS / 4) {
            const uint32_t seg = r / h, i = r % h;
            const unsigned long long base = w * M + (seg ? (M >> seg) : 0u);
            fold_one_quad<EC>(buckets, occ, base + i, base + h + i, role);
        }
        __threadfence_block();
        __syncthreads();
    }
    if (collect && (int)threadIdx.x <= log_m) collect_set<EC>(buckets, occ, M, log_m, w, collect, (int)threadIdx.x);
}
// two levels in one launch on quads of lanes (msm_fold2_kernel's item order)
template <class EC>
__global__ __launch_bounds__(MSM_ACC_THREADS) void msm_fold2_quad_kernel(uint32_t* buckets, uint8_t* occ, uint32_t M, uint32_t h, int nseg, int n_win) {
    const uint32_t q = h >> 1;
    const unsigned long long t = ((unsigned long long)blockIdx.x * MSM_ACC_THREADS + threadIdx.x) >> 2;
    const int role = threadIdx.x & 3;
    const unsigned long long per_win = (unsigned long long)nseg * q;
    if (t >= per_win * n_win) return;
    const unsigned long long w = t / per_win, r = t % per_win;
    const uint32_t seg = (uint32_t)(r / q), i = (uint32_t)(r % q);
    const unsigned long long base = w * M + (seg ? (M >> seg) : 0u);
#pragma unroll 1
    for (int step = 0; step < 4; step++) {
        if (step == 3 && seg) break;
        const unsigned long long ia = base + (step == 1 ? q : (step == 3 ? h : 0u)) + i;
        const unsigned long long ib = step < 2 ? ia + h : ia + q;
        fold_one_quad<EC>(buckets, occ, ia, ib, role);
    }
}

// out[w][0] = X[w*M], out[w][j] = X[w*M + (M>>j)], j = 1..log2 M, converted to the boundary form
template <class EC>
__global__ void msm_collect_kernel(const uint32_t* __restrict__ buckets, const uint8_t* __restrict__ occ, uint32_t M, int log_m, int n_win, uint32_t* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = log_m + 1;
    if (t >= n_win * per) return;
    const int w = t / per, j = t % per;
    const unsigned long long slot = (unsigned long long)w * M + (j ? (M >> j) : 0u);
    typename EC::Pt p = occ[slot] ? EC::load_pt(buckets, slot) : EC::inf();
    store_xyzz<typename EC::Field>(out, t, EC::to_boundary(p));
}

// ------------------------------------------------------------------------------------------------
// fixed-base powers: out[i] = beta^i * G as affine points (testing SRS, srs.rs:118-153 with g = G)
// ------------------------------------------------------------------------------------------------
// table[k] = 2^k * G, k < 256: one thread walks the doublings (XYZZ), then 256 threads normalise
template <class FQ>
__global__ void g1_pow2_table_kernel(uint32_t* __restrict__ table_xyzz, const uint32_t* __restrict__ g_xy = nullptr) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    using F = Fp<FQ>;
    Affine<F> g;
    if (g_xy) {                                            // a base point of the caller's (universal_setup_for_testing draws g = G1::rand)
        g.x = load_fp<FQ>(g_xy);
        g.y = load_fp<FQ>(g_xy + FQ::N);
    } else {
        g.x = F::from_const(FQ::GEN_X);
        g.y = F::from_const(FQ::GEN_Y);
    }
    XYZZ<F> cur = XYZZ<F>::from_affine(g);
    for (int k = 0; k < 256; k++) {
        store_xyzz<FQ>(table_xyzz, k, cur);
        cur = xyzz_dbl(cur);
    }
}
template <class FQ>
__global__ void g1_table_to_affine_kernel(const uint32_t* __restrict__ table_xyzz, uint32_t* __restrict__ table_xy, int n) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    Affine<Fp<FQ>> a = xyzz_to_affine(load_xyzz<FQ>(table_xyzz, k));
    store_fp<FQ>(table_xy + (size_t)k * 2 * FQ::N, a.x);
    store_fp<FQ>(table_xy + (size_t)k * 2 * FQ::N + FQ::N, a.y);
}

// scalars: canonical 8-word little-endian; out affine
template <class FQ>
__global__ __launch_bounds__(MSM_ACC_THREADS) void g1_fixed_base_kernel(const uint32_t* __restrict__ table, const uint32_t* __restrict__ scalars,
                                                                         unsigned long long n, uint32_t* __restrict__ out_xy) {
    const unsigned long long i = (unsigned long long)blockIdx.x * MSM_ACC_THREADS + threadIdx.x;
    if (i >= n) return;
    using F = Fp<FQ>;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int wd = 0; wd < 8; wd++) {
        uint32_t bits = scalars[i * 8 + wd];
        for (int b = 0; b < 32; b++) {
            if ((bits >> b) & 1) acc = xyzz_madd(acc, load_affine<FQ>(table, (unsigned long long)(wd * 32 + b)));
        }
    }
    Affine<F> a = xyzz_to_affine(acc);
    store_fp<FQ>(out_xy + i * 2 * FQ::N, a.x);
    store_fp<FQ>(out_xy + i * 2 * FQ::N + FQ::N, a.y);
}

// out[i] = beta^i (canonical), i < n: chunked so threads are independent
template <class FR>
__global__ __launch_bounds__(MSM_THREADS) void fr_powers_kernel(const uint32_t* __restrict__ beta_canon, unsigned long long n, uint32_t* __restrict__ out_canon) {
    const unsigned long long CH = 64;
    const unsigned long long t = (unsigned long long)blockIdx.x * MSM_THREADS + threadIdx.x;
    const unsigned long long start = t * CH;
    if (start >= n) return;
    using F = Fp<FR>;
    F b;
#pragma unroll
    for (int q = 0; q < 8; q++) b.l[q] = beta_canon[q];
    b = to_mont(b);
    F p = pow_u64(b, start);
    const unsigned long long end = start + CH < n ? start + CH : n;
    for (unsigned long long i = start; i < end; i++) {
        store_fp<FR>(out_canon + i * 8, from_mont(p));
        p = p * b;
    }
}

// ---- Lagrange-basis SRS from an SRS WITHOUT trapdoor: the inverse NTT over the group ---------------------------------------------
// S_j = [beta^j]g = sum_i (w^i)^j [L_i(beta)]g (X^j interpolated on H), i.e. S is the forward transform of the Lagrange-basis points:
//     [L_i(beta)]g = (1/n) sum_j w^(-ij) S_j.
// Decimation in frequency (natural order in, bit-reversed out) on an XYZZ array: a stage with half-size h maps (a, b) = (A[i], A[i + h])
// to (a + b, w_(2h)^(-k) (a - b)), k = i mod h -- a point addition, a subtraction and ONE 255-bit scalar multiplication per butterfly:
// (n / 2) log2 n of them.  A one-off per SRS and domain size.  The finishing pass scales by 1/n, undoes the bit reversal and normalises.
// Rounds 1-3 ran it bitwise (255 doublings + ~128 additions) on the 32-bit-limb arithmetic: 1.18 s at n = 2^20 on BLS12-381; since round 4:
// the reduced-radix arithmetic of the MSM (EcFx: 29-bit lazy limbs, ecx.cuh) with a signed 4-bit window per scalar multiplication --
// 0.58 s (BN254: 0.23 s; tools/lagrange_key_time.py).  A butterfly's multiple of D = a - b is sum_w d_w 16^w D with digits d_w in [-7, 8]: 7 additions build
// D .. 8D (in the thread's slice of a global scratch array: 8 x 224 bytes do not fit registers beside the accumulator), then 64 windows of
// four doublings and at most one addition -- 252 doublings + ~60 additions + 7 against 255 + ~128 of the bitwise form, each on products
// of 2 N'^2 single multiply-adds instead of 2 N^2 multiply-add / carry pairs.  The scaling by 1/n rides in the twiddles: an element is
// multiplied the first time it lands in the upper half of a butterfly, which happens in the FIRST block of a stage (every later block
// descends from an upper half already scaled), so that block's twiddles are w^(-k) / n (and its k = 0 element, which has none, gets
// 1/n alone: log2 n scalar multiplications in all, plus element 0 at the end) instead of n more multiplications in a finishing pass.
template <class X>
__global__ __launch_bounds__(MSM_ACC_THREADS) void ecx_ntt_load_kernel(const uint32_t* __restrict__ xy, unsigned long long n, uint32_t* __restrict__ a) {
    const unsigned long long i = (unsigned long long)blockIdx.x * MSM_ACC_THREADS + threadIdx.x;
    if (i >= n) return;
    using EC = EcFx<X>;
    const Fp<X> x = load_fp<X>(xy + i * 2 * X::N), y = load_fp<X>(xy + i * 2 * X::N + X::N);
    AffineX<X> p;
    p.x = fx_from_boundary<X>(x);
    p.y = fx_from_boundary<X>(y);
    EC::store_pt(a, i, XYZZX<X>::from_affine(p));                // (0, 0) = infinity on both sides
}
template <class X>
__device__ __forceinline__ XYZZX<X> xyzzx_neg(const XYZZX<X>& p) {            // -Y = K p - Y, K p the pad that covers an accumulator coordinate
    XYZZX<X> r = p;
    r.y = fx_norm(fx_sub_pad<X>(Fx<X>::zero(), p.y, X::XSUB_XY));
    return r;
}
// k * d, k a canonical 256-bit integer; tab: this thread's 8 slots of the scratch table (slot s of thread t at tab + (s * stride + t) points)
template <class X>
__device__ __forceinline__ XYZZX<X> ecx_scalar_mul(const XYZZX<X>& d, const uint32_t (&k)[8], uint32_t* __restrict__ tab, unsigned long long t, unsigned long long stride) {
    using EC = EcFx<X>;
    if (d.is_inf()) return d;
    XYZZX<X> m = d;
    EC::store_pt(tab, t, m);                                                  // 1 D
#pragma unroll 1
    for (int s = 1; s < 8; s++) {
        m = s == 1 ? xyzzx_dbl(m) : xyzzx_add(m, d);                          // 2 D, then + D each
        EC::store_pt(tab, (unsigned long long)s * stride + t, m);
    }
    // signed base-16 digits d_w = nibble + carry, minus 16 when above 8: the carry runs upwards, so they are recoded low to high first --
    // magnitudes 0..8 packed eight to a word, signs one bit each, a 65th digit for the last carry
    uint32_t dig[9];
    uint32_t sgn[2] = {0, 0};
    uint32_t carry = 0;
#pragma unroll
    for (int w8 = 0; w8 < 8; w8++) {
        uint32_t word = k[w8], out = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            uint32_t v = ((word >> (4 * q)) & 15u) + carry;
            carry = v > 8u ? 1u : 0u;                                         // digits in [-7, 8]
            const uint32_t mag = carry ? 16u - v : v;
            out |= mag << (4 * q);
            sgn[(w8 * 8 + q) >> 5] |= carry << ((w8 * 8 + q) & 31);
        }
        dig[w8] = out;
    }
    dig[8] = carry;
    XYZZX<X> acc = XYZZX<X>::inf();
#pragma unroll 1
    for (int w = 64; w >= 0; w--) {
        if (w < 64) {
#pragma unroll 1
            for (int q = 0; q < 4; q++) acc = xyzzx_dbl(acc);
        }
        const uint32_t mag = w == 64 ? dig[8] : (dig[w >> 3] >> (4 * (w & 7))) & 15u;
        if (mag) {
            XYZZX<X> e = EC::load_pt(tab, (unsigned long long)(mag - 1) * stride + t);
            if (w < 64 && ((sgn[w >> 5] >> (w & 31)) & 1u)) e = xyzzx_neg(e);
            acc = xyzzx_add(acc, e);
        }
    }
    return acc;
}
template <class FR, class X>
__global__ __launch_bounds__(MSM_ACC_THREADS) void ecx_ntt_stage_kernel(uint32_t* __restrict__ a, unsigned long long n, unsigned long long h,
                                                                        const uint32_t* __restrict__ winv_canon, uint32_t* __restrict__ tab) {
    using EC = EcFx<X>;
    const unsigned long long t = (unsigned long long)blockIdx.x * MSM_ACC_THREADS + threadIdx.x;
    if (t >= n / 2) return;
    const unsigned long long k = t % h, i = (t / h) * 2 * h + k, j = i + h;
    const bool scale = t < h;                                  // first block: its upper half is scaled by 1/n here (winv_canon + 8)
    const XYZZX<X> p = EC::load_pt(a, i), q = EC::load_pt(a, j);
    EC::store_pt(a, i, xyzzx_add(p, q));
    XYZZX<X> d = xyzzx_add(p, xyzzx_neg(q));
    if (k || scale) {                                          // times w_(2h)^(-k) = (w_n^-1)^(k n / (2h)) [/ n]
        Fp<FR> w, ni;
#pragma unroll
        for (int q8 = 0; q8 < 8; q8++) { w.l[q8] = winv_canon[q8]; ni.l[q8] = winv_canon[8 + q8]; }
        Fp<FR> twm = pow_u64(to_mont(w), k * (n / (2 * h)));
        if (scale) twm = twm * to_mont(ni);
        const Fp<FR> tw = from_mont(twm);
        uint32_t kk[8];
#pragma unroll
        for (int q8 = 0; q8 < 8; q8++) kk[q8] = tw.l[q8];
        d = ecx_scalar_mul<X>(d, kk, tab, t, n / 2);
    }
    EC::store_pt(a, j, d);
}
// out[bitrev(i)] = affine(ninv * A[i]) in the boundary form
template <class X>
__global__ __launch_bounds__(MSM_ACC_THREADS) void ecx_ntt_finish_kernel(const uint32_t* __restrict__ a, unsigned long long n, int log_n,
                                                                         const uint32_t* __restrict__ ninv_canon, uint32_t* __restrict__ tab,
                                                                         uint32_t* __restrict__ out_xy) {
    using EC = EcFx<X>;
    const unsigned long long i = (unsigned long long)blockIdx.x * MSM_ACC_THREADS + threadIdx.x;
    if (i >= n) return;
    XYZZX<X> p = EC::load_pt(a, i);
    if (i == 0) {                                              // the one element that never was in an upper half
        uint32_t kk[8];
#pragma unroll
        for (int q = 0; q < 8; q++) kk[q] = ninv_canon[q];
        p = ecx_scalar_mul<X>(p, kk, tab, 0, 1);
    }
    unsigned long long r = 0;
    for (int b = 0; b < log_n; b++) r |= ((i >> b) & 1ull) << (log_n - 1 - b);
    Affine<Fp<X>> q;
    if (p.is_inf()) { q.x = Fp<X>::zero(); q.y = Fp<X>::zero(); }
    else q = xyzz_to_affine(xyzzx_to_boundary(p));
    store_fp<X>(out_xy + r * 2 * X::N, q.x);
    store_fp<X>(out_xy + r * 2 * X::N + X::N, q.y);
}
// out[j] = S_(n + j) - S_j = [beta^j (beta^n - 1)]g, j < n_extra
template <class FQ>
__global__ void ec_ntt_extra_kernel(const uint32_t* __restrict__ xy, unsigned long long n, unsigned int n_extra, uint32_t* __restrict__ out_xy) {
    const unsigned int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_extra) return;
    Affine<Fp<FQ>> lo = load_affine<FQ>(xy, j);
    lo.y = neg(lo.y);
    const XYZZ<Fp<FQ>> d = xyzz_madd(XYZZ<Fp<FQ>>::from_affine(load_affine<FQ>(xy, n + j)), lo);
    Affine<Fp<FQ>> q;
    if (d.is_inf()) { q.x = Fp<FQ>::zero(); q.y = Fp<FQ>::zero(); }
    else q = xyzz_to_affine(d);
    store_fp<FQ>(out_xy + (size_t)j * 2 * FQ::N, q.x);
    store_fp<FQ>(out_xy + (size_t)j * 2 * FQ::N + FQ::N, q.y);
}

// Lagrange-basis scalars of the testing SRS: out[i] = L_i(beta) = w^i (beta^n - 1) / (n (beta - w^i)), i < n = 2^log_n (w the primitive
// n-th root of unity, so that sum_i v_i L_i(X) interpolates v on H), then out[n + j] = beta^j (beta^n - 1), j < n_extra -- the
// commitments of X^j Z_H(X), what a masked polynomial p + (b_0 + b_1 X + ..) Z_H adds to the commitment of p.  Canonical form.
// 64 consecutive i per thread with one shared inversion.
template <class FR>
__global__ __launch_bounds__(MSM_THREADS) void fr_lagrange_kernel(const uint32_t* __restrict__ beta_canon, int log_n, unsigned int n_extra,
                                                                  uint32_t* __restrict__ out_canon) {
    using F = Fp<FR>;
    constexpr int CH = 64;
    const unsigned long long n = 1ull << log_n;
    const unsigned long long t = (unsigned long long)blockIdx.x * MSM_THREADS + threadIdx.x;
    const unsigned long long start = t * CH;
    if (start >= n + n_extra) return;
    F b;
#pragma unroll
    for (int q = 0; q < 8; q++) b.l[q] = beta_canon[q];
    b = to_mont(b);
    const F vanish = pow_u64(b, n) - F::one();
    if (start >= n) return;                                    // (thread 0 writes the n_extra entries)
    F w = F::from_const(FR::ROOT);
    for (int i = log_n; i < FR::TWO_ADICITY; i++) w = sqr(w);
    const unsigned long long end = start + CH < n ? start + CH : n;
    const int cnt = (int)(end - start);
    F x = pow_u64(w, start);
    F pref[CH], xs[CH];
    F run = F::one();
    for (int j = 0; j < cnt; j++) {
        F d = b - x;
        if (d.is_zero()) d = F::one();
        xs[j] = x;
        pref[j] = run;
        run = run * d;
        x = x * w;
    }
    const F c = vanish * inv(from_u64<FR>(n));
    F inv_run = inv(run);
    for (int j = cnt - 1; j >= 0; j--) {
        F d = b - xs[j];
        const bool hit = d.is_zero();
        if (hit) d = F::one();
        F li = xs[j] * c * (inv_run * pref[j]);
        if (vanish.is_zero()) li = hit ? F::one() : F::zero();  // beta on the domain: L_i(beta) is 1 at beta = w^i, 0 elsewhere
        store_fp<FR>(out_canon + (start + j) * 8, from_mont(li));
        inv_run = inv_run * d;
    }
    if (t == 0) {
        F p = vanish;
        for (unsigned int j = 0; j < n_extra; j++) {
            store_fp<FR>(out_canon + (n + j) * 8, from_mont(p));
            p = p * b;
        }
    }
}

}  // namespace mzk
