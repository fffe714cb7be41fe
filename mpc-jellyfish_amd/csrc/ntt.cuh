// ntt.cuh -- multi-pass radix-2^r NTT / iNTT (plain and coset) over a 256-bit scalar field.
//
// Replaces ark-poly's Radix2EvaluationDomain::{fft,ifft}_in_place as driven by the reference at
//   relation/src/constraint_system.rs:1172,1189,1221,1240,1257   (ifft, size n)
//   plonk/src/proof_system/prover.rs:545-567                     (coset fft, size m = 8n)
//   plonk/src/proof_system/prover.rs:672                         (coset ifft, size m)
// Conventions: natural order in and out, Montgomery elements,
//   forward: out[i] = sum_j c[j] (h w^i)^j ;  inverse: c[j] = h^-j N^-1 sum_i e[i] w^(-ij).
//
// Decomposition.  N = R_1 R_2 ... R_K (each R_k = 2^r_k <= 2^9).  Pass k works on P_k = R_1..R_{k-1}
// independent contiguous blocks of R_k * S_k elements and transforms along the stride-S_k axis:
//     j = j1*S + j2,  i = i1 + R*i2:
//     X[i1 + R i2] = sum_j2 w_N^(j2 i1) w_S^(j2 i2) ( sum_j1 x[j1 S + j2] w_R^(j1 i1) ).
// A workgroup stages a tile of R_k rows x C contiguous columns in LDS (C*32 B coalesced
// segments), runs the R_k-point transforms there with LDS-resident stage twiddles, multiplies
// by the inter-pass twiddle w_N^(P i_k j2) on the way out, and stores in place.  The last pass
// (stride 1) reads C rows of R_K contiguous elements and scatters them digit-reversed, C
// contiguous elements at a time, into the second buffer -- so output is in natural order.
//
// Coset scaling is free: the forward pass k runs a decimation-in-time transform whose stage
// twiddles are  W_M[j] = w_M^j * g_k^(R/M), g_k = h^(S_k)  (inputs pre-scaled by h^j);
// the inverse pass k runs decimation-in-frequency with V_M[j] = (W_M[j])^-1, g_k = h^(P_k)
// (outputs post-scaled by h^-j).  N^-1 is folded into the first pass's inter-pass twiddle
// table (or applied at the store when K = 1).
//
// Roofline: 64*N algorithmic bytes per transform; K passes move 64*N*K bytes.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "fp.cuh"

namespace mzk {

constexpr int NTT_MAX_PASSES = 4;
constexpr int NTT_TILE_LOG = 11;        // R*C <= 2048 elements = 64 KiB of LDS
constexpr int NTT_MAX_LOG_R = 9;
constexpr int NTT_THREADS = 256;

struct NttPassArgs {
    uint32_t* in;              // pass input (batch 0)
    uint32_t* out;             // pass output (== in for non-final passes)
    const uint32_t* stage_tw;  // R-1 stage twiddles, entry (half-1+j)
    const uint32_t* t_lo;      // inter-pass twiddles, low  LB bits of the exponent
    const uint32_t* t_hi;      // inter-pass twiddles, high bits (pass 1 of an inverse: times N^-1)
    const uint32_t* scale;     // optional final multiplier (inverse, K = 1), else nullptr
    unsigned long long in_stride, out_stride;  // elements between consecutive polynomials of the batch
    unsigned long long in_len;        // first pass only: elements >= in_len read as zero
    int log_n, log_r, log_c, log_s, log_p;
    int log_lb;                // bits indexed by t_lo
    int is_first, is_final, n_pass;
    int log_radix[NTT_MAX_PASSES];
};

__device__ __forceinline__ unsigned bitrev(unsigned x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

template <class P>
__device__ __forceinline__ Fp<P> lds_load(const uint4* plane0, const uint4* plane1, int idx) {
    Fp<P> r;
    uint4 a = plane0[idx], b = plane1[idx];
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
template <class P>
__device__ __forceinline__ void lds_store(uint4* plane0, uint4* plane1, int idx, const Fp<P>& v) {
    plane0[idx] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    plane1[idx] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// One pass over one tile.  grid = (N / (R*C), batch), block = NTT_THREADS.
// LDS: [2 planes][R*C] uint4 for the tile, then [2 planes][R] uint4 for the stage twiddles.
template <class P, bool INVERSE>
__global__ __launch_bounds__(NTT_THREADS) void ntt_pass_kernel(NttPassArgs a) {
    static_assert(P::N == 8, "scalar fields are 8 x 32-bit limbs");
    extern __shared__ uint4 lds[];
    const int R = 1 << a.log_r, C = 1 << a.log_c, TILE = R * C;
    uint4* d0 = lds;
    uint4* d1 = lds + TILE;
    uint4* w0 = lds + 2 * TILE;
    uint4* w1 = w0 + R;
    const int tid = threadIdx.x;
    const unsigned long long tile = blockIdx.x;
    uint32_t* in = a.in + (unsigned long long)blockIdx.y * a.in_stride * 8;
    uint32_t* out = a.out + (unsigned long long)blockIdx.y * a.out_stride * 8;

    // stage twiddles -> LDS
    for (int i = tid; i < R - 1; i += NTT_THREADS) {
        const uint4* src = reinterpret_cast<const uint4*>(a.stage_tw) + 2 * i;
        w0[i] = src[0];
        w1[i] = src[1];
    }

    // ---- tile geometry -------------------------------------------------------------------
    unsigned long long base;      // non-final: index of element (row 0, col 0); final: unused
    unsigned long long c0 = 0;    // non-final: first column (j2) of the tile
    unsigned long long p1 = 1, i1_0 = 0, rest = 0, rev_rest = 0;
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
        for (int k = a.n_pass - 2; k >= 1; k--) {       // digits i_{K-1} ... i_2 of `rest`
            unsigned long long d = x & ((1ull << a.log_radix[k]) - 1);
            x >>= a.log_radix[k];
            rev_rest = (rev_rest << a.log_radix[k]) | d;
        }
        base = 0;
    }

    // ---- load tile into LDS ----------------------------------------------------------------
    for (int e = tid; e < TILE; e += NTT_THREADS) {
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
        Fp<P> v;
        if (a.is_first && g >= a.in_len) v = Fp<P>::zero();
        else v = load_fp<P>(in + g * 8);
        const int row = INVERSE ? r : (int)bitrev((unsigned)r, a.log_r);
        lds_store<P>(d0, d1, row * C + c, v);
    }
    __syncthreads();

    // ---- R-point transforms along the rows, all C columns at once ---------------------------
    const int nbf = TILE >> 1;
    for (int st = 0; st < a.log_r; st++) {
        const int s = INVERSE ? (a.log_r - 1 - st) : st;
        const int half = 1 << s;
        for (int bt = tid; bt < nbf; bt += NTT_THREADS) {
            const int c = bt & (C - 1);
            const int jj = bt >> a.log_c;
            const int j = jj & (half - 1);
            const int lo_row = ((jj >> s) << (s + 1)) + j;
            const int lo_i = lo_row * C + c, hi_i = lo_i + half * C;
            Fp<P> lo = lds_load<P>(d0, d1, lo_i);
            Fp<P> hi = lds_load<P>(d0, d1, hi_i);
            Fp<P> w = lds_load<P>(w0, w1, half - 1 + j);
            if (!INVERSE) {
                Fp<P> t = hi * w;
                lds_store<P>(d0, d1, lo_i, lo + t);
                lds_store<P>(d0, d1, hi_i, lo - t);
            } else {
                lds_store<P>(d0, d1, lo_i, lo + hi);
                lds_store<P>(d0, d1, hi_i, (lo - hi) * w);
            }
        }
        __syncthreads();
    }

    // ---- store ------------------------------------------------------------------------------
    for (int e = tid; e < TILE; e += NTT_THREADS) {
        const int c = e & (C - 1);
        const int r = e >> a.log_c;                          // transform output index i_k
        const int row = INVERSE ? (int)bitrev((unsigned)r, a.log_r) : r;
        Fp<P> v = lds_load<P>(d0, d1, row * C + c);
        unsigned long long g;
        if (!a.is_final) {
            const unsigned long long ex = ((unsigned long long)r * (c0 + c)) << a.log_p;
            Fp<P> tl = load_fp<P>(a.t_lo + (ex & ((1ull << a.log_lb) - 1)) * 8);
            Fp<P> th = load_fp<P>(a.t_hi + (ex >> a.log_lb) * 8);
            v = v * (tl * th);
            g = base + ((unsigned long long)r << a.log_s) + c;
        } else {
            if (a.scale) v = v * load_fp<P>(a.scale);
            const unsigned long long rev = (i1_0 + c) + (rev_rest << (a.n_pass > 1 ? a.log_radix[0] : 0));
            g = rev + ((unsigned long long)r << a.log_p);
        }
        store_fp<P>(out + g * 8, v);
    }
}

// ------------------------------------------------------------------------------------------------
// host side: plans
// ------------------------------------------------------------------------------------------------
struct NttPlanHost {
    int log_n = 0, n_pass = 0, log_lb = 0;
    bool inverse = false;
    int log_radix[NTT_MAX_PASSES] = {0, 0, 0, 0};
    std::vector<uint32_t> stage_tw[NTT_MAX_PASSES];   // (R_k - 1) * 8 words
    std::vector<uint32_t> t_lo, t_hi, t_hi_scaled, n_inv;
};

inline void ntt_choose_radices(int log_n, int* log_radix, int* n_pass) {
    if (log_n <= NTT_MAX_LOG_R) {
        *n_pass = 1;
        log_radix[0] = log_n;
        return;
    }
    // passes of about 8 bits; beyond 2^24 a fourth pass would cost more (its inter-pass twiddles and 64 B/element of
    // traffic) than 9-bit radices do (512-row tiles of 4 columns: 128-byte segments), so 2^25..2^27 stay at three passes
    int k = (log_n + 7) / 8;
    if (k == 4 && log_n <= 3 * NTT_MAX_LOG_R) k = 3;
    *n_pass = k;
    int basebits = log_n / k, extra = log_n % k;
    for (int i = 0; i < k; i++) log_radix[i] = basebits + (i < extra ? 1 : 0);
}

// Builds every table of a plan on the host (a few thousand field multiplications).
template <class P>
void ntt_build_plan(NttPlanHost& pl, int log_n, bool inverse, const uint32_t* coset_mont /* nullable */) {
    using F = Fp<P>;
    pl.log_n = log_n;
    pl.inverse = inverse;
    ntt_choose_radices(log_n, pl.log_radix, &pl.n_pass);
    F h = F::one();
    if (coset_mont)
        for (int i = 0; i < 8; i++) h.l[i] = coset_mont[i];
    // w_N
    F w = F::from_const(P::ROOT);
    for (int i = log_n; i < P::TWO_ADICITY; i++) w = sqr(w);
    F w_dir = inverse ? inv(w) : w;
    // stage twiddles per pass
    int log_p = 0;
    for (int k = 0; k < pl.n_pass; k++) {
        const int lr = pl.log_radix[k];
        const int R = 1 << lr;
        const int log_s = log_n - log_p - lr;
        // g_k = h^(S_k) forward, h^(P_k) inverse; the inverse tables hold V_M[j] = W_M[j]^-1
        F g = pow_u64(h, 1ull << (inverse ? log_p : log_s));
        if (inverse) g = inv(g);
        pl.stage_tw[k].assign((size_t)(R > 1 ? R - 1 : 1) * 8, 0);
        for (int s = 0; s < lr; s++) {
            const int half = 1 << s, M = 2 * half;
            F wm = pow_u64(w_dir, 1ull << (log_n - (s + 1)));    // w_M^(+-1) = w_N^(+-N/M)
            F cur = pow_u64(g, (uint64_t)(R / M));               // g^(+-R/M)
            for (int j = 0; j < half; j++) {
                for (int i = 0; i < 8; i++) pl.stage_tw[k][(size_t)(half - 1 + j) * 8 + i] = cur.l[i];
                cur = cur * wm;
            }
        }
        log_p += lr;
    }
    // inter-pass twiddles w_dir^e, e = e_hi * 2^LB + e_lo
    pl.log_lb = (log_n + 1) / 2;
    const size_t nlo = (size_t)1 << pl.log_lb, nhi = (size_t)1 << (log_n - pl.log_lb);
    pl.t_lo.resize(nlo * 8);
    pl.t_hi.resize(nhi * 8);
    pl.t_hi_scaled.resize(nhi * 8);
    F ninv = inv(from_u64<P>(1ull << log_n));
    pl.n_inv.assign(ninv.l, ninv.l + 8);
    F cur = F::one();
    for (size_t i = 0; i < nlo; i++) {
        for (int q = 0; q < 8; q++) pl.t_lo[i * 8 + q] = cur.l[q];
        cur = cur * w_dir;
    }
    F step = cur;                                                 // w_dir^(2^LB)
    cur = F::one();
    for (size_t i = 0; i < nhi; i++) {
        F sc = inverse ? cur * ninv : cur;
        for (int q = 0; q < 8; q++) { pl.t_hi[i * 8 + q] = cur.l[q]; pl.t_hi_scaled[i * 8 + q] = sc.l[q]; }
        cur = cur * step;
    }
}

}  // namespace mzk
