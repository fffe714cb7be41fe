// ntt_fx.cuh -- the multi-pass NTT of ntt.cuh on signed, lazily reduced 29-bit limbs (fs.cuh): every product of the transform has a
// plan-time constant operand, so it is a Barrett product with a precomputed quotient (143 multiply-adds) rather than a Montgomery
// product (162), data never change form, and the twiddle between pass 1 and pass 2 costs no product at all.
//
// Same decomposition, tile geometry, coalescing and natural-order scatter as ntt.cuh (see there for the index algebra and the
// reference call sites it replaces: prover.rs:545-567, 672; constraint_system.rs:1172-1257).  What this file adds:
//
//   * Folded first boundary.  With i = i1 S_1 + i2 and k = k1 + R_1 k2,
//         X[k] = sum_i2 (h w^k1)^i2  w_S1^(i2 k2)  ( sum_i1 x[i1 S_1 + i2] (h^S_1)^i1 w_R1^(i1 k1) ):
//     after pass 1, block k1 is a size-S_1 COSET transform with offset e_k1 = h w^k1, and a coset offset rides in the stage
//     twiddles for free (pass k uses g = e_k1^(S_k)).  So passes k >= 2 read their stage twiddles from a table indexed by k1
//     (R_1 x (R_k - 1) records; in the final pass k1 is the tile COLUMN) and the N-entry twiddle w^(k1 i2) -- two products per
//     element from a two-level table before -- disappears.  The boundaries after pass 2 keep an explicit product from a table of
//     N / R_1 or fewer entries (a per-element table for them would be N entries).
//   * Signed lazy limbs.  A butterfly is  t = hi * W (fs_mulc, |t| < 3p, limbs in [0, 2^29));  lo' = lo + t;  hi' = lo - t  with no
//     multiple-of-p pad and ONE normalisation (of lo) per butterfly: limbs stay within 2^30, which is what fs_mulc accepts.
//     Values grow by less than 3p per stage and nothing reduces them between pass 1 and pass 2: |v| < 2^256 + (9 + 9) 3p < 60p, and
//     2^261 / p is 70 (BLS12-381) / 169 (BN254); every later pass starts from a product (|v| < 3p).  nttx_build_plan asserts it.
//   * Between passes the data live in the scratch buffer as the 9 limbs themselves (planes of 16 + 16 + 4 bytes per element, the
//     LDS layout): no packing, no unpacking, and lazy signed values need no reduction to be stored.  Only the last pass
//     canonicalises (fs_canonical: quotient by p from the top limb) and packs into the boundary's 32-byte form.
//   * Both directions run decimation-in-time; the forward coset offset is in the stage twiddles, the inverse applies N^-1 h^-j
//     (and the internal-form factors 32 / 1/32 of the quotient kernels, plonk.cuh) as final products; a plain forward transform
//     has no final product at all.
// LDS: 9 words per element in three planes (b128, b128, b32): a 1024-element tile is 36 KiB, three 512-thread workgroups share a CU
// (one butterfly per thread per stage).  Measured (profiles/r02_nttprof_*): 80 M VALU wave-instructions per pass at 2^22 (114 M for
// the Montgomery version), 190 us per pass = ~85 % of the issue rate of that instruction mix.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "fs.cuh"
#include "ntt.cuh"

namespace mzk {

constexpr int NTTX_THREADS = 512;

struct NttxPassArgs {
    uint32_t* in;
    uint32_t* out;
    const uint32_t* stage_tw;  // records of FS_TW_WORDS words.  pass 1: entry (half-1+j); pass k >= 2: entry k1 (R-1) + (half-1+j)
    const uint32_t* t_full;    // middle passes k >= 2: w_N^(P r s), entry (r << log_s) + s  (N / P entries); else null
    const uint32_t* f_lo;      // final pass, inverse coset: N^-1 h^-j = f_lo[j & mask] * f_hi[j >> LB]; else null
    const uint32_t* f_hi;
    const uint32_t* f_one;     // final pass with a single factor (N^-1 and / or the internal-form factor); null: no final product
    unsigned long long in_stride, out_stride;      // elements between the polynomials of a batch (packed: 8 words each; planes: 9)
    unsigned long long in_len;
    unsigned long long n;                          // plane length of the scratch layout
    int log_n, log_r, log_c, log_s, log_p;
    int log_lb;
    int is_first, is_final, n_pass;
    int skip;                  // first pass of a zero-padded input (in_len <= N >> skip): the first `skip` stages are copies
    int in_planes, out_planes; // 1: the 9-limb plane layout of the scratch buffer; 0: the boundary's packed 8 words
    int skip_batch;            // -1, or a batch entry whose workgroups exit at once (the class-wise quotient transforms rows 0 .. W + 4 of a slab but
                               // has no use for an all-zero public-input row in the middle of them)
    const uint32_t* patch;     // first pass, nullable: elements g < 4 of batch entry y are read from patch[y * 4 + g] (packed) instead of `in` --
                               // the class-wise quotient transforms p mod (X^n - c), which differs from p's first n coefficients in 2-3 places
    int log_radix[NTT_MAX_PASSES];
};

template <class X>
__device__ __forceinline__ Fs<X> ldss_load(const int4* pa, const int4* pb, const int32_t* pc, int idx) {
    Fs<X> r;
    const int4 a = pa[idx], b = pb[idx];
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = pc[idx];
    return r;
}
template <class X>
__device__ __forceinline__ void ldss_store(int4* pa, int4* pb, int32_t* pc, int idx, const Fs<X>& v) {
    pa[idx] = make_int4(v.l[0], v.l[1], v.l[2], v.l[3]);
    pb[idx] = make_int4(v.l[4], v.l[5], v.l[6], v.l[7]);
    pc[idx] = v.l[8];
}
// scratch planes of one polynomial: limbs 0-3 at word 4g, limbs 4-7 at word 4n + 4g, limb 8 at word 8n + g
template <class X>
__device__ __forceinline__ Fs<X> planes_get(const uint32_t* __restrict__ base, unsigned long long n, unsigned long long g) {
    Fs<X> r;
    const int4 a = reinterpret_cast<const int4*>(base)[g], b = reinterpret_cast<const int4*>(base + 4 * n)[g];
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = reinterpret_cast<const int32_t*>(base + 8 * n)[g];
    return r;
}
template <class X>
__device__ __forceinline__ void planes_put(uint32_t* __restrict__ base, unsigned long long n, unsigned long long g, const Fs<X>& v) {
    reinterpret_cast<int4*>(base)[g] = make_int4(v.l[0], v.l[1], v.l[2], v.l[3]);
    reinterpret_cast<int4*>(base + 4 * n)[g] = make_int4(v.l[4], v.l[5], v.l[6], v.l[7]);
    reinterpret_cast<int32_t*>(base + 8 * n)[g] = v.l[8];
}
template <class X>
__device__ __forceinline__ Fs<X> fs_load_packed(const uint32_t* __restrict__ p) {
    const Fp<X> t = load_fp<X>(p);
    return fs_unpack<X>(t.l);
}
__device__ __forceinline__ FsTw tws_load(const uint32_t* __restrict__ table, unsigned long long idx) {
    const uint4* src = reinterpret_cast<const uint4*>(table + idx * FS_TW_WORDS);
    const uint4 a = src[0], b = src[1], c = src[2], d = src[3], e = src[4];
    FsTw t;
    t.w[0] = a.x; t.w[1] = a.y; t.w[2] = a.z; t.w[3] = a.w;
    t.w[4] = b.x; t.w[5] = b.y; t.w[6] = b.z; t.w[7] = b.w;
    t.w[8] = c.x; t.q[0] = c.y; t.q[1] = c.z; t.q[2] = c.w;
    t.q[3] = d.x; t.q[4] = d.y; t.q[5] = d.z; t.q[6] = d.w;
    t.q[7] = e.x; t.q[8] = e.y;
    return t;
}

// One pass over one tile.  grid = (N / (R*C), batch), block = NTTX_THREADS.  R4: stage PAIRS in registers (below), for 2048-element tiles.
// The batch entry and the plan's tables come from a selector: a plain launch reads blockIdx.y and the tables of its ONE plan in the
// argument struct at their points of use (NttxPlainSel: the code of rounds 1-4); a launch over several cosets picks them per class
// (NttxClassSel, nttx_pass_classes_kernel below).
//   y_in / y_out: batch entry of the input / of the output and the patch; y_row: what skip_batch is compared with
struct NttxPlainSel {
    const NttxPassArgs& a;
    __device__ __forceinline__ unsigned y_in() const { return blockIdx.y; }
    __device__ __forceinline__ unsigned y_out() const { return blockIdx.y; }
    __device__ __forceinline__ unsigned y_row() const { return blockIdx.y; }
    __device__ __forceinline__ const uint32_t* stage_tw() const { return a.stage_tw; }
    __device__ __forceinline__ const uint32_t* t_full() const { return a.t_full; }
    __device__ __forceinline__ const uint32_t* f_lo() const { return a.f_lo; }
    __device__ __forceinline__ const uint32_t* f_hi() const { return a.f_hi; }
    __device__ __forceinline__ const uint32_t* f_one() const { return a.f_one; }
};
template <class X, bool R4, class SEL>
__device__ __forceinline__ void nttx_pass_body(const NttxPassArgs& a, const SEL sel) {
    static_assert(X::XN == 9 && X::N == 8, "256-bit scalar fields: 8 boundary words, 9 limbs of 29 bits");
    extern __shared__ int4 ldsx[];
    if ((int)sel.y_row() == a.skip_batch) return;
    const int R = 1 << a.log_r, C = 1 << a.log_c, TILE = R * C;
    int4* da = ldsx;
    int4* db = da + TILE;
    int32_t* dc = reinterpret_cast<int32_t*>(db + TILE);
    const int tid = threadIdx.x;
    const unsigned long long tile = blockIdx.x;
    const uint32_t* in = a.in + (unsigned long long)sel.y_in() * a.in_stride * (a.in_planes ? 9 : 8);
    uint32_t* out = a.out + (unsigned long long)sel.y_out() * a.out_stride * (a.out_planes ? 9 : 8);

    // ---- tile geometry (as ntt.cuh) --------------------------------------------------------------------
    const int log_r1 = a.n_pass > 1 ? a.log_radix[0] : 0;
    unsigned long long base = 0, c0 = 0, p1 = 1, i1_0 = 0, rest = 0, rev_rest = 0, k1_blk = 0;
    if (!a.is_final) {
        const unsigned long long tiles_per_blk = 1ull << (a.log_s - a.log_c);
        const unsigned long long blk = tile >> (a.log_s - a.log_c);
        c0 = (tile & (tiles_per_blk - 1)) << a.log_c;
        base = (blk << (a.log_r + a.log_s)) + c0;
        k1_blk = a.is_first ? 0 : (blk >> (a.log_p - log_r1));        // the pass-1 output index of this block
    } else {
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

    // ---- load: bit-reversed rows (decimation in time) ------------------------------------------------------
    // Zero-padded input (the quotient round transforms polynomials of degree n + 2 on 8n points): rows r >= R >> skip are
    // zero for every tile, so in bit-reversed order each run of 2^skip positions holds one non-zero value first, and the
    // first `skip` stages -- butterflies whose multiplied operand is zero -- merely copy it across the run.
    if (a.skip > 0) {
        const int rows = R >> a.skip, reps = 1 << a.skip;
        for (int e = tid; e < rows * C; e += NTTX_THREADS) {
            const int c = e & (C - 1), r = e >> a.log_c;
            const unsigned long long g = base + ((unsigned long long)r << a.log_s) + c;
            Fs<X> v = Fs<X>::zero();
            if (g < a.in_len) v = fs_load_packed<X>(in + g * 8);
            const int pos = (int)bitrev((unsigned)r, a.log_r) * C + c;
            for (int t = 0; t < reps; t++) ldss_store<X>(da, db, dc, pos + t * C, v);
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
        Fs<X> v;
#ifdef MZK_NTT_DIAG_NOIO
        v = Fs<X>::zero(); v.l[0] = (int32_t)g;
#else
        if (a.in_planes) v = planes_get<X>(in, a.n, g);                                  // lazy: |limbs| <= 2^30
        else if (a.patch && g < 4) v = fs_load_packed<X>(a.patch + ((unsigned long long)sel.y_out() * 4 + g) * 8);
        else if (a.is_first && g >= a.in_len) v = Fs<X>::zero();
        else v = fs_load_packed<X>(in + g * 8);                                        // fresh: limbs in [0, 2^29)
#endif
        ldss_store<X>(da, db, dc, (int)bitrev((unsigned)r, a.log_r) * C + c, v);
    }
    __syncthreads();

    // ---- R-point transforms: one butterfly per thread per stage -------------------------------------------
    // Invariant: every LDS value has |limbs| <= 2^30.  lo is normalised (limbs in [-4, 2^29 + 4)) unless it is fresh data,
    // t = hi W is class C, so both outputs stay within 2^30.
    const int nbf = TILE >> 1;
    const int tw_rows = R - 1;
    int s0 = a.skip;
    // R4 (round 4): two stages per LDS round trip.  A thread takes the four rows b, b + h, b + 2h, b + 3h (h = 2^s) of one column through
    // stages s and s + 1 in registers -- the same four constant-operand products and normalisations as two radix-2 stages, half the
    // barriers and LDS traffic, 36 data registers instead of 18: 120 VGPRs, four waves per SIMD, which is what 2048-element tiles (72 KB of
    // LDS, every thread one group of four rows) leave anyway.  Measured (profiles/r04_ntt_radix4.txt, bit-identical outputs): with
    // 2048-element tiles 0.612 -> 0.592 ms at 2^22, 2.31 -> 2.16 ms at 2^24, 0.190 -> 0.184 at 2^20 -- but 0.070 -> 0.084 ms at 2^16 (too few
    // workgroups), and on 1024-element tiles (half the threads idle in the paired steps) 0.654 ms at 2^22: the dispatcher takes this form
    // for transforms of at least 2^21 points (ntt.hip).
    if constexpr (R4)
    for (; s0 + 1 < a.log_r; s0 += 2) {
        const int s = s0, half = 1 << s;
        const bool fresh = !a.in_planes && s == a.skip;
        for (int g = tid; g < (TILE >> 2); g += NTTX_THREADS) {
            const int c = g & (C - 1);
            const int jj = g >> a.log_c;
            const int j = jj & (half - 1);
            const int i0 = ((((jj >> s) << (s + 2)) + j) * C) + c, st = half * C;
            Fs<X> x0 = ldss_load<X>(da, db, dc, i0), x2 = ldss_load<X>(da, db, dc, i0 + 2 * st);
            const Fs<X> x1 = ldss_load<X>(da, db, dc, i0 + st), x3 = ldss_load<X>(da, db, dc, i0 + 3 * st);
            if (!fresh) { x0 = fs_norm(x0); x2 = fs_norm(x2); }
            const unsigned long long k1 = a.is_first ? 0ull : (a.is_final ? i1_0 + c : k1_blk);
            const uint32_t* tw = sel.stage_tw() + k1 * tw_rows * FS_TW_WORDS;
            const FsTw w1 = tws_load(tw, (unsigned long long)(half - 1 + j));
            const Fs<X> t1 = fs_mulc<X>(x1, w1), t3 = fs_mulc<X>(x3, w1);
            const Fs<X> y0 = fs_norm(fs_add(x0, t1)), y1 = fs_norm(fs_sub(x0, t1));
            const Fs<X> u2 = fs_mulc<X>(fs_add(x2, t3), tws_load(tw, (unsigned long long)(2 * half - 1 + j)));
            const Fs<X> u3 = fs_mulc<X>(fs_sub(x2, t3), tws_load(tw, (unsigned long long)(3 * half - 1 + j)));
            ldss_store<X>(da, db, dc, i0, fs_add(y0, u2));
            ldss_store<X>(da, db, dc, i0 + 2 * st, fs_sub(y0, u2));
            ldss_store<X>(da, db, dc, i0 + st, fs_add(y1, u3));
            ldss_store<X>(da, db, dc, i0 + 3 * st, fs_sub(y1, u3));
        }
        __syncthreads();
    }
#ifdef MZK_NTT_DIAG_NOSTAGES                         // (diagnostic builds only, tools/ntt_diag.sh: what a pass costs without its butterflies / without its HBM traffic)
    for (int s = a.log_r; s < a.log_r; s++) {
#else
    for (int s = s0; s < a.log_r; s++) {
#endif
        const int half = 1 << s;
        const bool fresh = !a.in_planes && s == a.skip;
        for (int bt = tid; bt < nbf; bt += NTTX_THREADS) {
            const int c = bt & (C - 1);
            const int jj = bt >> a.log_c;
            const int j = jj & (half - 1);
            const int lo_i = (((jj >> s) << (s + 1)) + j) * C + c, hi_i = lo_i + half * C;
            Fs<X> lo = ldss_load<X>(da, db, dc, lo_i);
            const Fs<X> hi = ldss_load<X>(da, db, dc, hi_i);
            if (!fresh) lo = fs_norm(lo);
            const unsigned long long k1 = a.is_first ? 0ull : (a.is_final ? i1_0 + c : k1_blk);
            const Fs<X> t = fs_mulc<X>(hi, tws_load(sel.stage_tw(), k1 * tw_rows + (half - 1 + j)));
            ldss_store<X>(da, db, dc, lo_i, fs_add(lo, t));
            ldss_store<X>(da, db, dc, hi_i, fs_sub(lo, t));
        }
        __syncthreads();
    }

    // ---- store ------------------------------------------------------------------------------------------
    for (int e = tid; e < TILE; e += NTTX_THREADS) {
        const int c = e & (C - 1);
        const int r = e >> a.log_c;
        Fs<X> v = ldss_load<X>(da, db, dc, r * C + c);
        if (!a.is_final) {
            const unsigned long long g = base + ((unsigned long long)r << a.log_s) + c;
            if (!a.is_first) v = fs_mulc<X>(v, tws_load(sel.t_full(), ((unsigned long long)r << a.log_s) + c0 + c));   // boundaries after pass 2
#ifdef MZK_NTT_DIAG_NOIO
            if (v.l[0] == 0x7fffffff && v.l[8] == 12345) planes_put<X>(out, a.n, g, v);
#else
            planes_put<X>(out, a.n, g, v);                                           // pass 1: stored as it is (folded boundary)
#endif
        } else {
            const unsigned long long rev = (i1_0 + c) + (rev_rest << log_r1);
            const unsigned long long g = rev + ((unsigned long long)r << a.log_p);
            if (sel.f_lo()) {
                v = fs_mulc<X>(v, tws_load(sel.f_lo(), g & ((1ull << a.log_lb) - 1)));
                v = fs_mulc<X>(v, tws_load(sel.f_hi(), g >> a.log_lb));
            } else if (sel.f_one()) {
                v = fs_mulc<X>(v, tws_load(sel.f_one(), 0));
            }
#ifdef MZK_NTT_DIAG_NOIO
            const Fx<X> cv = fs_canonical<X>(v);
            if (cv.l[0] == 0x7fffffff && cv.l[8] == 12345) fx_store_packed<X>(out + g * 8, cv);
#else
            fx_store_packed<X>(out + g * 8, fs_canonical<X>(v));
#endif
        }
    }
}

template <class X, bool R4>
__global__ __launch_bounds__(NTTX_THREADS) void nttx_pass_kernel(NttxPassArgs a) {
    nttx_pass_body<X, R4>(a, NttxPlainSel{a});
}

// Several transforms of ONE shape over DIFFERENT cosets in one launch: the quotient round's residue classes (plonk.hip) -- at 2^15
// gates a class's batch is 192 workgroups on 256 CUs and the round a chain of 5 x 5 launches; here batch entry y belongs to class
// y / rows and takes that class's tables.  The first pass of a forward transform reads the SHARED input rows (entry y % rows of `in`),
// everything else is class-major.  Radix-2 stages only (small transforms).
constexpr int NTTX_MAX_CLASSES = 8;
struct NttxClasses {
    unsigned rows;                    // batch entries per class
    int shared_in;                    // first pass: 1 = every class reads input row y % rows; 0 = class-major input
    const uint32_t* stage_tw[NTTX_MAX_CLASSES];
    const uint32_t* t_full[NTTX_MAX_CLASSES];
    const uint32_t* f_lo[NTTX_MAX_CLASSES];
    const uint32_t* f_hi[NTTX_MAX_CLASSES];
    const uint32_t* f_one[NTTX_MAX_CLASSES];
};
struct NttxClassSel {
    unsigned yi, y, row;
    const uint32_t *p_stage, *p_full, *p_lo, *p_hi, *p_one;
    __device__ __forceinline__ unsigned y_in() const { return yi; }
    __device__ __forceinline__ unsigned y_out() const { return y; }
    __device__ __forceinline__ unsigned y_row() const { return row; }
    __device__ __forceinline__ const uint32_t* stage_tw() const { return p_stage; }
    __device__ __forceinline__ const uint32_t* t_full() const { return p_full; }
    __device__ __forceinline__ const uint32_t* f_lo() const { return p_lo; }
    __device__ __forceinline__ const uint32_t* f_hi() const { return p_hi; }
    __device__ __forceinline__ const uint32_t* f_one() const { return p_one; }
};
template <class X>
__global__ __launch_bounds__(NTTX_THREADS) void nttx_pass_classes_kernel(NttxPassArgs a, NttxClasses mc) {
    const unsigned y = blockIdx.y, c = y / mc.rows, row = y - c * mc.rows;
    nttx_pass_body<X, false>(a, NttxClassSel{(a.is_first && mc.shared_in) ? row : y, y, row, mc.stage_tw[c], mc.t_full[c], mc.f_lo[c], mc.f_hi[c], mc.f_one[c]});
}

// ------------------------------------------------------------------------------------------------
// Persistent form of the same pass (round 3).  Measured on MI355X (tools/ntt_diag.sh): a 2^22 pass is 0.155 ms of butterflies when
// it touches no HBM and 0.067 ms of HBM traffic (4.3 TB/s) when it runs no butterflies, but 0.194 ms as launched above -- one
// workgroup per tile loads, transforms and stores in sequence, and only the OTHER workgroups of its CU fill its memory phases.
// Here a workgroup walks over tiles t = blockIdx.x, += gridDim.x (tile index fastest, then the polynomial of the batch) and, right
// after a tile has been staged into LDS, issues the global loads of its NEXT tile into registers (2 elements of 8 / 9 words per
// thread): they complete under the stages of the current tile and are written to LDS once its results have left.
// ------------------------------------------------------------------------------------------------
struct NttxTile {
    unsigned long long base, c0, p1, i1_0, rest, rev_rest, k1_blk;
    const uint32_t* in;
    uint32_t* out;
};
template <class X>
__device__ __forceinline__ NttxTile nttx_tile(const NttxPassArgs& a, unsigned long long t, int log_tiles, int log_r1) {
    NttxTile g;
    const unsigned long long tile = t & ((1ull << log_tiles) - 1), poly = t >> log_tiles;          // (tiles per polynomial: a power of two)
    g.in = a.in + poly * a.in_stride * (a.in_planes ? 9 : 8);
    g.out = a.out + poly * a.out_stride * (a.out_planes ? 9 : 8);
    g.base = g.c0 = g.i1_0 = g.rest = g.rev_rest = g.k1_blk = 0;
    g.p1 = 1;
    if (!a.is_final) {
        const unsigned long long tiles_per_blk = 1ull << (a.log_s - a.log_c);
        const unsigned long long blk = tile >> (a.log_s - a.log_c);
        g.c0 = (tile & (tiles_per_blk - 1)) << a.log_c;
        g.base = (blk << (a.log_r + a.log_s)) + g.c0;
        g.k1_blk = a.is_first ? 0 : (blk >> (a.log_p - log_r1));
    } else {
        const int log_p1 = a.log_p - log_r1;
        g.p1 = 1ull << log_p1;
        g.rest = tile & (g.p1 - 1);
        g.i1_0 = (tile >> log_p1) << a.log_c;
        unsigned long long x = g.rest;
        for (int k = a.n_pass - 2; k >= 1; k--) {
            const unsigned long long d = x & ((1ull << a.log_radix[k]) - 1);
            x >>= a.log_radix[k];
            g.rev_rest = (g.rev_rest << a.log_radix[k]) | d;
        }
    }
    return g;
}
// element e of a tile: where it comes from (global index, -1: a zero of the padding) and its row / column in the tile
__device__ __forceinline__ void nttx_src(const NttxPassArgs& a, const NttxTile& g, int e, long long& gidx, int& r, int& c) {
    const int C = 1 << a.log_c, R = 1 << a.log_r;
    if (!a.is_final) {
        c = e & (C - 1);
        r = e >> a.log_c;
        gidx = (long long)(g.base + ((unsigned long long)r << a.log_s) + c);
    } else {
        r = e & (R - 1);
        c = e >> a.log_r;
        gidx = (long long)((((g.i1_0 + c) * g.p1 + g.rest) << a.log_r) + r);
    }
    if (!a.in_planes && a.is_first && (unsigned long long)gidx >= a.in_len) gidx = -1;
}
template <class X>
__device__ __forceinline__ void nttx_fetch(const NttxPassArgs& a, const NttxTile& g, long long gidx, uint32_t (&raw)[9]) {
    if (gidx < 0) {
#pragma unroll
        for (int i = 0; i < 9; i++) raw[i] = 0;
        return;
    }
    if (a.in_planes) {
        const int4 p = reinterpret_cast<const int4*>(g.in)[gidx], q = reinterpret_cast<const int4*>(g.in + 4 * a.n)[gidx];
        raw[0] = p.x; raw[1] = p.y; raw[2] = p.z; raw[3] = p.w; raw[4] = q.x; raw[5] = q.y; raw[6] = q.z; raw[7] = q.w;
        raw[8] = (uint32_t)reinterpret_cast<const int32_t*>(g.in + 8 * a.n)[gidx];
    } else {
        const uint4 p = reinterpret_cast<const uint4*>(g.in + gidx * 8)[0], q = reinterpret_cast<const uint4*>(g.in + gidx * 8)[1];
        raw[0] = p.x; raw[1] = p.y; raw[2] = p.z; raw[3] = p.w; raw[4] = q.x; raw[5] = q.y; raw[6] = q.z; raw[7] = q.w;
        raw[8] = 0;
    }
}
template <class X>
__device__ __forceinline__ Fs<X> nttx_decode(const NttxPassArgs& a, const uint32_t (&raw)[9]) {
    Fs<X> v;
    if (a.in_planes) {
#pragma unroll
        for (int i = 0; i < 9; i++) v.l[i] = (int32_t)raw[i];
        return v;
    }
    return fs_unpack<X>(raw);
}

// workgroup barrier that orders LDS traffic only.  __syncthreads() is fence + barrier, and on gfx9-family hardware one counter
// (vmcnt) tracks global loads AND stores: its fence waits for every outstanding global access -- the prefetch of the next tile
// included, which would then overlap a single stage.  The stages exchange data through LDS alone, so lgkmcnt(0) is what they need.
__device__ __forceinline__ void nttx_lds_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}
__device__ __forceinline__ FsTw tws_load_lds(const int4* tw, int idx) {
    const int4* src = tw + idx * 5;
    const int4 a = src[0], b = src[1], c = src[2], d = src[3], e = src[4];
    FsTw t;
    t.w[0] = a.x; t.w[1] = a.y; t.w[2] = a.z; t.w[3] = a.w;
    t.w[4] = b.x; t.w[5] = b.y; t.w[6] = b.z; t.w[7] = b.w;
    t.w[8] = c.x; t.q[0] = c.y; t.q[1] = c.z; t.q[2] = c.w;
    t.q[3] = d.x; t.q[4] = d.y; t.q[5] = d.z; t.q[6] = d.w;
    t.q[7] = e.x; t.q[8] = e.y;
    return t;
}

// grid = (min(tiles * batch, what the chip holds at once)), block = NTTX_THREADS.  TW_LDS (every pass but the final one, whose
// twiddles depend on the tile column): the R - 1 stage-twiddle records of the tile's block are staged in LDS behind the tile --
// the stages then issue no global load at all, so the next tile's prefetch is not waited for by an in-order twiddle load.
template <class X, bool TW_LDS>
__global__ __launch_bounds__(NTTX_THREADS) void nttx_pass_persistent_kernel(NttxPassArgs a, int log_tiles, unsigned long long total) {
    static_assert(X::XN == 9 && X::N == 8, "256-bit scalar fields: 8 boundary words, 9 limbs of 29 bits");
    extern __shared__ int4 ldsx[];
    const int R = 1 << a.log_r, C = 1 << a.log_c, TILE = R * C;
    int4* da = ldsx;
    int4* db = da + TILE;
    int32_t* dc = reinterpret_cast<int32_t*>(db + TILE);
    const int tid = threadIdx.x;
    const int log_r1 = a.n_pass > 1 ? a.log_radix[0] : 0;
    // elements per thread: e0 = tid, e1 = tid + NTTX_THREADS (a tile has at most 2 * NTTX_THREADS of them); with a zero-padded first
    // pass only the rows r < R >> skip exist (the others are zeros that the first `skip` stages would merely copy)
    const int n_src = a.skip > 0 ? (R >> a.skip) * C : TILE;
    const int e0 = tid, e1 = tid + NTTX_THREADS;
    unsigned long long t = blockIdx.x;
    if (t >= total) return;
    NttxTile g = nttx_tile<X>(a, t, log_tiles, log_r1);
    int4* tw = reinterpret_cast<int4*>(dc + TILE);                                   // [R - 1][5] when TW_LDS
    unsigned long long tw_k1 = ~0ull;
    uint32_t raw0[9], raw1[9];
    long long g0 = -1, g1 = -1;
    int r0 = 0, c0 = 0, r1 = 0, c1 = 0;
    if (e0 < n_src) { nttx_src(a, g, e0, g0, r0, c0); nttx_fetch<X>(a, g, g0, raw0); }
    if (e1 < n_src) { nttx_src(a, g, e1, g1, r1, c1); nttx_fetch<X>(a, g, g1, raw1); }
    for (;;) {
        // ---- stage the fetched elements: bit-reversed rows (decimation in time) -------------------------------------
        if (a.skip > 0) {
            const int reps = 1 << a.skip;
            if (e0 < n_src) {
                const Fs<X> v = nttx_decode<X>(a, raw0);
                const int pos = (int)bitrev((unsigned)r0, a.log_r) * C + c0;
                for (int q = 0; q < reps; q++) ldss_store<X>(da, db, dc, pos + q * C, v);
            }
            if (e1 < n_src) {
                const Fs<X> v = nttx_decode<X>(a, raw1);
                const int pos = (int)bitrev((unsigned)r1, a.log_r) * C + c1;
                for (int q = 0; q < reps; q++) ldss_store<X>(da, db, dc, pos + q * C, v);
            }
        } else {
            if (e0 < n_src) ldss_store<X>(da, db, dc, (int)bitrev((unsigned)r0, a.log_r) * C + c0, nttx_decode<X>(a, raw0));
            if (e1 < n_src) ldss_store<X>(da, db, dc, (int)bitrev((unsigned)r1, a.log_r) * C + c1, nttx_decode<X>(a, raw1));
        }
        if (TW_LDS && tw_k1 != g.k1_blk) {                         // pass 1: once per workgroup; later passes: when the block's k1 changes
            const int4* src = reinterpret_cast<const int4*>(a.stage_tw + g.k1_blk * (unsigned long long)(R - 1) * FS_TW_WORDS);
            for (int i = tid; i < (R - 1) * 5; i += NTTX_THREADS) tw[i] = src[i];
            tw_k1 = g.k1_blk;
        }
        nttx_lds_barrier();                                        // the tile (and its twiddles) are in LDS; the previous tile's stores may still be draining
        // ---- the next tile's loads go out now and land under the stages below ----------------------------------------
        const unsigned long long tn = t + gridDim.x;
        const bool more = tn < total;
        NttxTile gn = g;
        if (more) {
            gn = nttx_tile<X>(a, tn, log_tiles, log_r1);
            if (e0 < n_src) { nttx_src(a, gn, e0, g0, r0, c0); nttx_fetch<X>(a, gn, g0, raw0); }
            if (e1 < n_src) { nttx_src(a, gn, e1, g1, r1, c1); nttx_fetch<X>(a, gn, g1, raw1); }
        }
        // ---- R-point transforms: one butterfly per thread per stage (as nttx_pass_kernel) ------------------------------
        const int nbf = TILE >> 1;
        const int tw_rows = R - 1;
        for (int s = a.skip; s < a.log_r; s++) {
            const int half = 1 << s;
            const bool fresh = !a.in_planes && s == a.skip;
            for (int bt = tid; bt < nbf; bt += NTTX_THREADS) {
                const int c = bt & (C - 1);
                const int jj = bt >> a.log_c;
                const int j = jj & (half - 1);
                const int lo_i = (((jj >> s) << (s + 1)) + j) * C + c, hi_i = lo_i + half * C;
                Fs<X> lo = ldss_load<X>(da, db, dc, lo_i);
                const Fs<X> hi = ldss_load<X>(da, db, dc, hi_i);
                if (!fresh) lo = fs_norm(lo);
                Fs<X> tt;
                if (TW_LDS) tt = fs_mulc<X>(hi, tws_load_lds(tw, half - 1 + j));
                else {
                    const unsigned long long k1 = a.is_first ? 0ull : (a.is_final ? g.i1_0 + c : g.k1_blk);
                    tt = fs_mulc<X>(hi, tws_load(a.stage_tw, k1 * tw_rows + (half - 1 + j)));
                }
                ldss_store<X>(da, db, dc, lo_i, fs_add(lo, tt));
                ldss_store<X>(da, db, dc, hi_i, fs_sub(lo, tt));
            }
            nttx_lds_barrier();                                    // LDS only: the next tile's loads stay in flight
        }
        // ---- store --------------------------------------------------------------------------------------------------
        for (int e = tid; e < TILE; e += NTTX_THREADS) {
            const int c = e & (C - 1);
            const int r = e >> a.log_c;
            Fs<X> v = ldss_load<X>(da, db, dc, r * C + c);
            if (!a.is_final) {
                const unsigned long long gi = g.base + ((unsigned long long)r << a.log_s) + c;
                if (!a.is_first) v = fs_mulc<X>(v, tws_load(a.t_full, ((unsigned long long)r << a.log_s) + g.c0 + c));   // boundaries after pass 2
                planes_put<X>(g.out, a.n, gi, v);
            } else {
                const unsigned long long rev = (g.i1_0 + c) + (g.rev_rest << log_r1);
                const unsigned long long gi = rev + ((unsigned long long)r << a.log_p);
                if (a.f_lo) {
                    v = fs_mulc<X>(v, tws_load(a.f_lo, gi & ((1ull << a.log_lb) - 1)));
                    v = fs_mulc<X>(v, tws_load(a.f_hi, gi >> a.log_lb));
                } else if (a.f_one) {
                    v = fs_mulc<X>(v, tws_load(a.f_one, 0));
                }
                fx_store_packed<X>(g.out + gi * 8, fs_canonical<X>(v));
            }
        }
        if (!more) break;
        nttx_lds_barrier();                                        // every value of this tile has left LDS (its global stores may still be in flight)
        t = tn;
        g = gn;
    }
}

// ------------------------------------------------------------------------------------------------
// host side: plan tables
// ------------------------------------------------------------------------------------------------
struct NttxPlanHost {
    int log_n = 0, n_pass = 0, log_lb = 0;
    bool inverse = false, coset = false, final_factor = false;
    int log_radix[NTT_MAX_PASSES] = {0, 0, 0, 0};
    std::vector<uint32_t> stage_tw[NTT_MAX_PASSES];
    std::vector<uint32_t> f_lo, f_hi, f_one;
    std::vector<uint32_t> t_full[NTT_MAX_PASSES];       // middle passes k >= 2 only
};

// boundary-form field element (x*R) -> constant-operand record (x and floor(x 2^261 / p) as 9 limbs each, padded)
template <class X>
inline void nttx_put(std::vector<uint32_t>& dst, size_t idx, const Fp<X>& a) {
    const FsTw t = fs_make_tw<X>(a);
    uint32_t* d = dst.data() + idx * FS_TW_WORDS;
    for (int i = 0; i < FS_N; i++) { d[i] = (uint32_t)t.w[i]; d[FS_N + i] = (uint32_t)t.q[i]; }
    d[2 * FS_N] = d[2 * FS_N + 1] = 0;
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
    const F h_fwd = inverse ? F::one() : h;                 // the inverse applies h^-j at the end instead
    const int lr1 = pl.log_radix[0];
    int log_p = 0;
    for (int k = 0; k < pl.n_pass; k++) {
        const int lr = pl.log_radix[k];
        const int R = 1 << lr;
        const int log_s = log_n - log_p - lr;
        const size_t rows = (size_t)(R > 1 ? R - 1 : 1);
        // Pass 1: g = h^(S_1).  Pass k >= 2, block k1: the sub-transform's offset is e_k1 = h w^k1, so g = e_k1^(S_k) =
        // h^(S_k) (w^(S_k))^k1; stage twiddles W_M[j] = w_M^j g^(R/M) (inputs pre-scaled by g^j: decimation in time).
        const size_t blocks = k == 0 ? 1 : (size_t)1 << lr1;
        pl.stage_tw[k].assign(blocks * rows * FS_TW_WORDS, 0);
        const F g0 = pow_u64(h_fwd, 1ull << log_s);
        const F gstep = k == 0 ? F::one() : pow_u64(w_dir, 1ull << log_s);
        F g = g0;
        for (size_t k1 = 0; k1 < blocks; k1++) {
            for (int s = 0; s < lr; s++) {
                const int half = 1 << s, M = 2 * half;
                const F wm = pow_u64(w_dir, 1ull << (log_n - (s + 1)));
                F cur = pow_u64(g, (uint64_t)(R / M));
                for (int j = 0; j < half; j++) {
                    nttx_put<X>(pl.stage_tw[k], k1 * rows + (size_t)(half - 1 + j), cur);
                    cur = cur * wm;
                }
            }
            g = g * gstep;
        }
        log_p += lr;
    }
    // Boundaries after pass 2: w_N^(P_k r s), r < R_k, s < S_k -- the same for every block, N / P_k <= N / R_1 entries, one
    // product per element.  (The boundary after pass 1 is folded into the tables above.)
    log_p = lr1;
    for (int k = 1; k + 1 < pl.n_pass; k++) {
        const int lr = pl.log_radix[k];
        const int log_s = log_n - log_p - lr;
        const size_t R = (size_t)1 << lr, S = (size_t)1 << log_s;
        pl.t_full[k].assign(R * S * FS_TW_WORDS, 0);
        const F wp = pow_u64(w_dir, 1ull << log_p);                 // w_N^P
        F row = F::one();                                           // (w_N^P)^r
        for (size_t r = 0; r < R; r++) {
            F cur = F::one();
            for (size_t sidx = 0; sidx < S; sidx++) { nttx_put<X>(pl.t_full[k], (r << log_s) + sidx, cur); cur = cur * row; }
            row = row * wp;
        }
        log_p += lr;
    }
    // final-pass factor(s)
    // scale: 1 = leave the output in the internal form x * R' (R' = 32 R), 2 = take an internal-form input back to x * R
    // (plonk.cuh); a plain forward transform has none: the final pass only canonicalises
    F ninv = inverse ? inv(from_u64<X>(1ull << log_n)) : F::one();
    if (scale == 1) ninv = ninv * from_u64<X>(32);
    if (scale == 2) ninv = ninv * inv(from_u64<X>(32));
    pl.final_factor = inverse || scale != 0;
    pl.f_one.assign(FS_TW_WORDS, 0);
    nttx_put<X>(pl.f_one, 0, ninv);
    pl.log_lb = (log_n + 1) / 2;
    if (inverse && pl.coset) {
        const size_t nlo = (size_t)1 << pl.log_lb, nhi = (size_t)1 << (log_n - pl.log_lb);
        const F hi = inv(h);
        pl.f_lo.assign(nlo * FS_TW_WORDS, 0);
        pl.f_hi.assign(nhi * FS_TW_WORDS, 0);
        F cur = F::one();
        for (size_t i = 0; i < nlo; i++) { nttx_put<X>(pl.f_lo, i, cur); cur = cur * hi; }
        const F hstep = cur;
        cur = ninv;
        for (size_t i = 0; i < nhi; i++) { nttx_put<X>(pl.f_hi, i, cur); cur = cur * hstep; }
    }
}

// |v| < 2^256 + 3p * (stages of pass 1 + stages of pass 2) must stay below 2^261 (fs_mulc's contract): true for every size the
// radix chooser produces (at most 9 + 9 stages); checked when a plan is built
template <class X>
inline bool nttx_growth_ok(const NttxPlanHost& pl) {
    const int stages = pl.log_radix[0] + (pl.n_pass > 1 ? pl.log_radix[1] : 0);
    // p > 2^(BITS-1): 2^256 / p < 2^(257 - BITS); the bound (2^(257-BITS) + 3 stages) p < 2^261 holds if the factor < 2^(261-BITS)
    const double factor = (double)(1ull << (257 - X::BITS)) + 3.0 * stages;
    return factor < (double)(1ull << (261 - X::BITS));
}

}  // namespace mzk
