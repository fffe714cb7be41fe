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
// Coset scaling is free in the forward direction: pass k runs a decimation-in-time transform whose stage
// twiddles are  W_M[j] = w_M^j * g_k^(R/M), g_k = h^(S_k)  (inputs pre-scaled by h^j).  The inverse runs
// decimation-in-time as well and applies N^-1 h^-j in the multiplication every final pass performs (ntt_fx.cuh).
//
// Roofline: 64*N algorithmic bytes per transform; K passes move 64*N*K bytes.
//
// This header keeps the decomposition and its shared definitions; the pass kernel and the plan tables live in
// ntt_fx.cuh (reduced-radix field, both directions decimation-in-time, N^-1 / coset factors in the final pass).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <vector>

#include "fp.cuh"

namespace mzk {

constexpr int NTT_MAX_PASSES = 4;
#ifndef MZK_NTT_TILE_LOG
#define MZK_NTT_TILE_LOG 10                            // (a -D override exists for A/B builds: profiles/r02_ntt_tile_ab.txt)
#endif
// R*C <= 1024 elements = 36 KiB of LDS at 9 limbs (ntt_fx.cuh): three 512-thread workgroups per CU, one butterfly per thread per
// stage.  2048-element tiles (two workgroups per CU) run the same at 2^22 and above and 25 % slower at 2^16..2^18.
constexpr int NTT_TILE_LOG = MZK_NTT_TILE_LOG;
constexpr int NTT_MAX_LOG_R = 9;

__device__ __forceinline__ unsigned bitrev(unsigned x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

inline void ntt_choose_radices(int log_n, int* log_radix, int* n_pass) {
    // EXPERIMENT switch (profiles/r05_ntt_two_pass.txt): MZK_NTT_RADICES="11,11" forces the pass structure of transforms whose size is
    // the sum of the listed radices (with MZK_NTT_TILE_LOG_RT=12 for 4096-element tiles); nothing ships with it
    if (const char* f = std::getenv("MZK_NTT_RADICES")) {
        int r[NTT_MAX_PASSES], k = 0, sum = 0;
        for (const char* q = f; *q && k < NTT_MAX_PASSES;) {
            r[k] = std::atoi(q);
            sum += r[k++];
            while (*q && *q != ',') q++;
            if (*q == ',') q++;
        }
        if (sum == log_n && k >= 1) {
            *n_pass = k;
            for (int i = 0; i < k; i++) log_radix[i] = r[i];
            return;
        }
    }
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

}  // namespace mzk
