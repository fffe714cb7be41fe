// poly.hip -- host side of the dense-polynomial primitives (poly.cuh).
#include "internal.hpp"
#include "poly.cuh"

namespace mzk {
namespace {

template <class P>
int32_t eval_run(const uint32_t* d_coeffs, uint64_t stride, uint64_t len, uint32_t batch, const uint32_t* x_mont, uint32_t* out_host, hipStream_t st) {
    using F = Fp<P>;
    if (len == 0) { std::memset(out_host, 0, (size_t)batch * 32); return MZK_OK; }
    F x;
    std::memcpy(x.l, x_mont, 32);
    const F y = pow_u64(x, POLY_EVAL_T);
    const int blocks = POLY_EVAL_T / POLY_THREADS;
    MZK_TRY(ws_acquire(st));
    MZK_TRY(g_ws.poly_tmp.reserve((size_t)POLY_EVAL_T * 32 + 64 + (size_t)batch * blocks * 32 + (size_t)batch * 32));
    uint32_t* xpow = g_ws.poly_tmp.as<uint32_t>();
    uint32_t* d_xy = xpow + (size_t)POLY_EVAL_T * 8;               // x, y
    uint32_t* partial = d_xy + 16;
    uint32_t* d_out = partial + (size_t)batch * blocks * 8;
    HIP_TRY(hipMemcpyAsync(d_xy, x.l, 32, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_xy + 8, y.l, 32, hipMemcpyHostToDevice, st));
    const uint64_t tlen = len < POLY_EVAL_T ? len : POLY_EVAL_T;
    hipLaunchKernelGGL((fr_powers_mont_kernel<P>), dim3((unsigned)(((tlen + 15) / 16 + POLY_THREADS - 1) / POLY_THREADS)), dim3(POLY_THREADS), 0, st,
                       d_xy, tlen, xpow);
    hipLaunchKernelGGL((poly_eval_partial_kernel<P>), dim3(blocks, batch), dim3(POLY_THREADS), 0, st, d_coeffs, stride, len, xpow, d_xy + 8, partial);
    hipLaunchKernelGGL((poly_eval_final_kernel<P>), dim3(batch), dim3(POLY_THREADS), 0, st, partial, blocks, d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_host, d_out, (size_t)batch * 32, hipMemcpyDeviceToHost, st));
    MZK_TRY(ws_release(st));
    HIP_TRY(hipStreamSynchronize(st));
    return MZK_OK;
}

template <class P>
int32_t div_run(const uint32_t* d_poly, uint64_t len, const uint32_t* z_mont, uint32_t* d_out, hipStream_t st) {
    using F = Fp<P>;
    if (len <= 1) return MZK_OK;                                   // degree-0 (or empty) dividend: zero quotient, nothing to write
    F z;
    std::memcpy(z.l, z_mont, 32);
    if (z.is_zero()) {                                             // division by X: shift down
        HIP_TRY(hipMemcpyAsync(d_out, d_poly + 8, (len - 1) * 32, hipMemcpyDeviceToDevice, st));
        return MZK_OK;
    }
    const F zi = inv(z);
    const unsigned n_blocks = (unsigned)((len + DIV_BLOCK - 1) / DIV_BLOCK);
    MZK_TRY(ws_acquire(st));
    MZK_TRY(g_ws.poly_tmp.reserve(64 + (size_t)len * 32 * 3 + (size_t)n_blocks * 32));
    uint32_t* d_c = g_ws.poly_tmp.as<uint32_t>();                  // z, z^-1
    uint32_t* zpow = d_c + 16;
    uint32_t* zinvpow = zpow + len * 8;
    uint32_t* t = zinvpow + len * 8;
    uint32_t* totals = t + len * 8;
    HIP_TRY(hipMemcpyAsync(d_c, z.l, 32, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_c + 8, zi.l, 32, hipMemcpyHostToDevice, st));
    const unsigned pg = (unsigned)(((len + 15) / 16 + POLY_THREADS - 1) / POLY_THREADS);
    const unsigned eg = (unsigned)((len + POLY_THREADS - 1) / POLY_THREADS);
    hipLaunchKernelGGL((fr_powers_mont_kernel<P>), dim3(pg), dim3(POLY_THREADS), 0, st, d_c, len, zpow);
    hipLaunchKernelGGL((fr_powers_mont_kernel<P>), dim3(pg), dim3(POLY_THREADS), 0, st, d_c + 8, len, zinvpow);
    hipLaunchKernelGGL((poly_div_scale_kernel<P>), dim3(eg), dim3(POLY_THREADS), 0, st, d_poly, zpow, len, t);
    hipLaunchKernelGGL((fr_suffix_add_block_kernel<P>), dim3(n_blocks), dim3(POLY_THREADS), 0, st, t, len, totals);
    hipLaunchKernelGGL((fr_suffix_add_totals_kernel<P>), dim3(1), dim3(1024), 0, st, totals, n_blocks);
    hipLaunchKernelGGL((poly_div_finish_kernel<P>), dim3(eg), dim3(POLY_THREADS), 0, st, t, totals, zinvpow, len, d_out);
    HIP_TRY(hipGetLastError());
    MZK_TRY(ws_release(st));
    return MZK_OK;
}

template <class P>
int32_t lincomb_run(uint32_t n_terms, const uint32_t* const* d_polys, const uint64_t* lens, const uint32_t* scalars, uint32_t* d_out, uint64_t out_len,
                    hipStream_t st) {
    if (out_len == 0) return MZK_OK;
    LincombArgs a;
    std::memset(&a, 0, sizeof a);
    a.n_terms = (int)n_terms;
    a.out_len = out_len;
    a.out = d_out;
    for (uint32_t k = 0; k < n_terms; k++) {
        a.poly[k] = d_polys[k];
        a.len[k] = lens[k];
        std::memcpy(a.scalar[k], scalars + (size_t)k * 8, 32);
    }
    hipLaunchKernelGGL((poly_lincomb_kernel<P>), dim3((unsigned)((out_len + POLY_THREADS - 1) / POLY_THREADS)), dim3(POLY_THREADS), 0, st, a);
    HIP_TRY(hipGetLastError());
    return MZK_OK;
}

}  // namespace

int32_t poly_eval_dispatch(int curve, const uint32_t* d_coeffs, uint64_t stride, uint64_t len, uint32_t batch, const uint32_t* x_mont, uint32_t* out_host,
                           hipStream_t st) {
    if (curve == 0) return eval_run<BlsFr>(d_coeffs, stride, len, batch, x_mont, out_host, st);
    if (curve == 1) return eval_run<BnFr>(d_coeffs, stride, len, batch, x_mont, out_host, st);
    set_error("unknown curve_id");
    return MZK_ERR_INVALID_ARG;
}
int32_t poly_div_dispatch(int curve, const uint32_t* d_poly, uint64_t len, const uint32_t* z_mont, uint32_t* d_out, hipStream_t st) {
    if (curve == 0) return div_run<BlsFr>(d_poly, len, z_mont, d_out, st);
    if (curve == 1) return div_run<BnFr>(d_poly, len, z_mont, d_out, st);
    set_error("unknown curve_id");
    return MZK_ERR_INVALID_ARG;
}
int32_t poly_lincomb_dispatch(int curve, uint32_t n_terms, const uint32_t* const* d_polys, const uint64_t* lens, const uint32_t* scalars, uint32_t* d_out,
                              uint64_t out_len, hipStream_t st) {
    if (n_terms > POLY_MAX_TERMS) { set_error("too many terms in one linear combination (max 32)"); return MZK_ERR_INVALID_ARG; }
    if (curve == 0) return lincomb_run<BlsFr>(n_terms, d_polys, lens, scalars, d_out, out_len, st);
    if (curve == 1) return lincomb_run<BnFr>(n_terms, d_polys, lens, scalars, d_out, out_len, st);
    set_error("unknown curve_id");
    return MZK_ERR_INVALID_ARG;
}

int32_t poly_mask_dispatch(int curve, uint32_t n_rows, uint32_t* const* d_rows, uint64_t n, uint32_t n_blind, const uint32_t* blind_mont, hipStream_t st) {
    if (n_rows == 0) return MZK_OK;
    if (n_rows > MASK_MAX_ROWS || n_blind == 0 || n_blind > MASK_MAX_BLIND || n_blind > n) { set_error("mask: at most 8 polynomials, 1..4 blinders each"); return MZK_ERR_INVALID_ARG; }
    MaskArgs a;
    a.n = n; a.n_rows = (int)n_rows; a.n_blind = (int)n_blind;
    for (uint32_t r = 0; r < n_rows; r++) {
        a.rows[r] = d_rows[r];
        for (uint32_t j = 0; j < n_blind; j++) std::memcpy(a.blind[r][j], blind_mont + ((size_t)r * n_blind + j) * 8, 32);
    }
    if (curve == 0) hipLaunchKernelGGL((poly_mask_kernel<BlsFr>), dim3(1), dim3(64), 0, st, a);
    else if (curve == 1) hipLaunchKernelGGL((poly_mask_kernel<BnFr>), dim3(1), dim3(64), 0, st, a);
    else { set_error("unknown curve_id"); return MZK_ERR_INVALID_ARG; }
    HIP_TRY(hipGetLastError());
    return MZK_OK;
}

}  // namespace mzk
