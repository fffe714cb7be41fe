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
    // stride T of the strided Horner: a thread's chain is len / T dependent products long and a polynomial gets T / 256 workgroups --
    // at 2^22 coefficients the fixed 16384 meant 256-deep chains on 64 workgroups (170 us for a single polynomial)
    uint64_t T = POLY_EVAL_T;
    while (T * 32 < len && T < (uint64_t)POLY_EVAL_T_MAX) T <<= 1;
    const F y = pow_u64(x, T);
    const int blocks = (int)(T / POLY_THREADS);
    MZK_TRY(ws_acquire(st));
    MZK_TRY(g_ws.poly_tmp.reserve((size_t)T * 32 + (size_t)batch * blocks * 32 + (size_t)batch * 32));
    uint32_t* xpow = g_ws.poly_tmp.as<uint32_t>();
    uint32_t* partial = xpow + (size_t)T * 8;
    uint32_t* d_out = partial + (size_t)batch * blocks * 8;
    const uint64_t tlen = len < T ? len : T;
    launch_powers<P>(st, &x, 1, tlen, &xpow);                        // (x's powers and y travel as kernel arguments)
    hipLaunchKernelGGL((poly_eval_partial_kernel<P>), dim3(blocks, batch), dim3(POLY_THREADS), 0, st, d_coeffs, stride, len, xpow, to_fr_arg<P>(y),
                       (unsigned long long)T, partial);
    hipLaunchKernelGGL((poly_eval_final_kernel<P>), dim3(batch), dim3(POLY_THREADS), 0, st, partial, blocks, d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_host, d_out, (size_t)batch * 32, hipMemcpyDeviceToHost, st));
    MZK_TRY(ws_release(st));
    HIP_TRY(hipStreamSynchronize(st));
    return MZK_OK;
}

// several evaluation jobs at up to two points: the two power tables once, a partial launch per job, ONE final launch, ONE copy and wait
template <class P>
int32_t eval_many_run(const EvalJob* jobs, uint32_t n_jobs, const uint32_t* x_mont, uint32_t* out_host, hipStream_t st) {
    using F = Fp<P>;
    uint64_t max_len = 0, total = 0;
    bool used[2] = {false, false};
    for (uint32_t j = 0; j < n_jobs; j++) { max_len = std::max(max_len, jobs[j].len); total += jobs[j].batch; used[jobs[j].which_x & 1] = true; }
    if (total == 0) return MZK_OK;
    if (max_len == 0) { std::memset(out_host, 0, (size_t)total * 32); return MZK_OK; }
    uint64_t T = POLY_EVAL_T;
    while (T * 32 < max_len && T < (uint64_t)POLY_EVAL_T_MAX) T <<= 1;
    const int blocks = (int)(T / POLY_THREADS);
    F x[2], y[2];
    for (int q = 0; q < 2; q++) { std::memcpy(x[q].l, x_mont + 8 * q, 32); y[q] = pow_u64(x[q], T); }
    MZK_TRY(ws_acquire(st));
    MZK_TRY(g_ws.poly_tmp.reserve(2 * (size_t)T * 32 + (size_t)total * blocks * 32 + (size_t)total * 32));
    uint32_t* xpow = g_ws.poly_tmp.as<uint32_t>();
    uint32_t* partial = xpow + 2 * (size_t)T * 8;
    uint32_t* d_out = partial + (size_t)total * blocks * 8;
    const uint64_t tlen = max_len < T ? max_len : T;
    {                                                              // the power tables of the points in use: one launch
        F bases[2];
        uint32_t* tabs[2];
        int nb = 0;
        for (int q = 0; q < 2; q++)
            if (used[q]) { bases[nb] = x[q]; tabs[nb] = xpow + (size_t)q * T * 8; nb++; }
        launch_powers<P>(st, bases, nb, tlen, tabs);
    }
    uint64_t at = 0;
    for (uint32_t j = 0; j < n_jobs; j++) {
        const EvalJob& jb = jobs[j];
        if (jb.batch == 0) continue;
        const int q = (int)(jb.which_x & 1);
        if (jb.len == 0) HIP_TRY(hipMemsetAsync(partial + at * blocks * 8, 0, (size_t)jb.batch * blocks * 32, st));
        else
            hipLaunchKernelGGL((poly_eval_partial_kernel<P>), dim3(blocks, jb.batch), dim3(POLY_THREADS), 0, st, jb.d, jb.stride, jb.len, xpow + (size_t)q * T * 8,
                               to_fr_arg<P>(y[q]), (unsigned long long)T, partial + at * blocks * 8);
        at += jb.batch;
    }
    hipLaunchKernelGGL((poly_eval_final_kernel<P>), dim3((unsigned)total), dim3(POLY_THREADS), 0, st, partial, blocks, d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_host, d_out, (size_t)total * 32, hipMemcpyDeviceToHost, st));
    MZK_TRY(ws_release(st));
    HIP_TRY(hipStreamSynchronize(st));
    return MZK_OK;
}

template <class P>
int32_t div_run(const uint32_t* d_poly, uint64_t len, const uint32_t* z_mont, uint32_t* d_out, uint32_t* d_rem, hipStream_t st) {
    using F = Fp<P>;
    if (len == 0 && d_rem) HIP_TRY(hipMemsetAsync(d_rem, 0, 32, st));
    if (len == 1 && d_rem) HIP_TRY(hipMemcpyAsync(d_rem, d_poly, 32, hipMemcpyDeviceToDevice, st));
    if (len <= 1) return MZK_OK;                                   // degree-0 (or empty) dividend: zero quotient, nothing to write
    F z;
    std::memcpy(z.l, z_mont, 32);
    if (z.is_zero()) {                                             // division by X: shift down
        HIP_TRY(hipMemcpyAsync(d_out, d_poly + 8, (len - 1) * 32, hipMemcpyDeviceToDevice, st));
        if (d_rem) HIP_TRY(hipMemcpyAsync(d_rem, d_poly, 32, hipMemcpyDeviceToDevice, st));
        return MZK_OK;
    }
    const F zi = inv(z);
    const unsigned n_blocks = (unsigned)((len + DIV_BLOCK - 1) / DIV_BLOCK);
    MZK_TRY(ws_acquire(st));
    MZK_TRY(g_ws.poly_tmp.reserve((size_t)len * 32 * 3 + (size_t)n_blocks * 32));
    uint32_t* zpow = g_ws.poly_tmp.as<uint32_t>();
    uint32_t* zinvpow = zpow + len * 8;
    uint32_t* t = zinvpow + len * 8;
    uint32_t* totals = t + len * 8;
    const unsigned eg = (unsigned)((len + POLY_THREADS - 1) / POLY_THREADS);
    {
        const F bases[2] = {z, zi};
        uint32_t* const tabs[2] = {zpow, zinvpow};
        launch_powers<P>(st, bases, 2, len, tabs);
    }
    hipLaunchKernelGGL((poly_div_scale_kernel<P>), dim3(eg), dim3(POLY_THREADS), 0, st, d_poly, zpow, len, t);
    hipLaunchKernelGGL((fr_suffix_add_block_kernel<P>), dim3(n_blocks), dim3(POLY_THREADS), 0, st, t, len, totals);
    hipLaunchKernelGGL((fr_suffix_add_totals_kernel<P>), dim3(1), dim3(1024), 0, st, totals, n_blocks);
    hipLaunchKernelGGL((poly_div_finish_kernel<P>), dim3(eg), dim3(POLY_THREADS), 0, st, t, totals, zinvpow, len, d_out, d_rem);
    HIP_TRY(hipGetLastError());
    MZK_TRY(ws_release(st));
    return MZK_OK;
}

template <class P> struct FrXOf;
template <> struct FrXOf<BlsFr> { using type = BlsFrX; };
template <> struct FrXOf<BnFr> { using type = BnFrX; };

// floor(p / prod_{i<count} (X - w^(first+i))), w the primitive 2^log_order-th root of unity.  Fast path (the roots are distinct
// and p vanishes on all of them, i.e. the remainder is zero): coset NTT, pointwise 1/Z_D, inverse coset NTT.  Otherwise the
// linear factors are divided out one at a time -- floor division by a product is the composition of the floor divisions.
template <class P>
int32_t div_roots_run(int curve, const uint32_t* d_poly, uint64_t len, uint32_t log_order, uint64_t first, uint64_t count, uint32_t* d_out, hipStream_t st) {
    using F = Fp<P>;
    if (log_order > (uint32_t)P::TWO_ADICITY) { set_error("log_order exceeds the two-adicity of the scalar field"); return MZK_ERR_INVALID_ARG; }
    if (len <= count) return MZK_OK;                               // quotient is the zero polynomial: nothing to write
    const uint64_t out_len = len - count;
    if (count == 0) {
        HIP_TRY(hipMemcpyAsync(d_out, d_poly, len * 32, hipMemcpyDeviceToDevice, st));
        return MZK_OK;
    }
    F g = F::from_const(P::ROOT);
    for (int i = (int)log_order; i < P::TWO_ADICITY; i++) g = sqr(g);
    const F root0 = pow_u64(g, first);
    // the quotient has out_len coefficients: a coset of 2^log_q >= out_len points determines it, and p on that coset is
    // (p mod X^N - h^N) on it; the roots of Z_D are 2^log_order-th roots of unity, where p is (p mod X^Ne - 1)
    int log_q = 6;
    while ((1ull << log_q) < out_len) log_q++;
    const int log_e = log_q > (int)log_order ? log_q : (int)log_order;
    bool fast = len >= 64 && count <= (1ull << log_order) && log_e <= P::TWO_ADICITY && log_e <= 27;
    const uint64_t Ne = 1ull << log_e, Nq = 1ull << log_q;
    MZK_TRY(ws_acquire(st));
    MZK_TRY(g_ws.link_tmp.reserve(((fast && Ne > 2 * len) ? Ne : 2 * len) * 32 + 2 * count * 32));     // either path may follow the root check
    uint32_t* T = g_ws.link_tmp.as<uint32_t>();
    uint32_t* d_at_roots = T + (((fast && Ne > 2 * len) ? Ne : 2 * len)) * 8;
    uint32_t* d_roots = d_at_roots + count * 8;
    DivRootsArgs a;
    std::memset(&a, 0, sizeof a);
    std::vector<uint32_t> roots;
    if (fast) {
        roots.resize(count * 8);
        const F to_int = from_u64<P>(32);                              // x * R -> x * R' (the internal form of the fx kernels)
        F cur = root0;
        for (uint64_t i = 0; i < count; i++) { const F ri = cur * to_int; std::memcpy(&roots[i * 8], ri.l, 32); cur = cur * g; }
        HIP_TRY(hipMemcpyAsync(d_roots, roots.data(), count * 32, hipMemcpyHostToDevice, st));
        std::memcpy(a.h, P::R1, 32);
        hipLaunchKernelGGL((poly_fold_kernel<P>), dim3((unsigned)((Ne + POLY_THREADS - 1) / POLY_THREADS)), dim3(POLY_THREADS), 0, st, d_poly, len, Ne, a, T);
        MZK_TRY(ntt_dispatch(curve, T, Ne, log_e, false, nullptr, 1, Ne, st));
        hipLaunchKernelGGL((poly_gather_roots_kernel<P>), dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, T, first, (1ull << log_order) - 1,
                           1ull << (log_e - (int)log_order), count, d_at_roots);
        HIP_TRY(hipGetLastError());
        std::vector<uint32_t> at_roots(count * 8);
        HIP_TRY(hipMemcpyAsync(at_roots.data(), d_at_roots, count * 32, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));                             // (also: the roots table has left the host vector)
        for (uint32_t v : at_roots) if (v) { fast = false; break; }
    }
    if (fast) {
        const F h = F::from_const(P::GENERATOR);
        const F h_n = pow_u64(h, Nq);
        std::memcpy(a.h, h_n.l, 32);
        hipLaunchKernelGGL((poly_fold_kernel<P>), dim3((unsigned)((Nq + POLY_THREADS - 1) / POLY_THREADS)), dim3(POLY_THREADS), 0, st, d_poly, len, Nq, a, T);
        MZK_TRY(ntt_dispatch(curve, T, Nq, log_q, false, P::GENERATOR, 1, Nq, st, 1));      // evaluations left in the internal form
        using X = typename FrXOf<P>::type;
        const int K = log_q >= 19 ? 4 : log_q >= 18 ? 2 : 1;            // points per thread: keep >= 2^17 threads where the domain allows
        a.evals = T;
        a.roots = d_roots;
        a.threads = Nq / K;
        a.count = (unsigned)count;
        F w = F::from_const(P::ROOT);
        for (int i = log_q; i < P::TWO_ADICITY; i++) w = sqr(w);
        const F to_int = from_u64<P>(32);
        const F w_step = pow_u64(w, a.threads) * to_int, h_int = h * to_int, w_int = w * to_int;
        std::memcpy(a.h, h_int.l, 32);
        std::memcpy(a.w, w_int.l, 32);
        std::memcpy(a.w_step, w_step.l, 32);
        const dim3 grid((unsigned)((a.threads + POLY_THREADS - 1) / POLY_THREADS));
        if (K == 4) hipLaunchKernelGGL((poly_div_roots_pointwise_fx_kernel<X, 4>), grid, dim3(POLY_THREADS), 0, st, a);
        else if (K == 2) hipLaunchKernelGGL((poly_div_roots_pointwise_fx_kernel<X, 2>), grid, dim3(POLY_THREADS), 0, st, a);
        else hipLaunchKernelGGL((poly_div_roots_pointwise_fx_kernel<X, 1>), grid, dim3(POLY_THREADS), 0, st, a);
        HIP_TRY(hipGetLastError());
        MZK_TRY(ntt_dispatch(curve, T, Nq, log_q, true, P::GENERATOR, 1, Nq, st, 2));       // internal form in, boundary form out
        HIP_TRY(hipMemcpyAsync(d_out, T, out_len * 32, hipMemcpyDeviceToDevice, st));
    } else {
        uint32_t* buf[2] = {T, T + len * 8};
        const uint32_t* cur = d_poly;
        uint64_t cur_len = len;
        F root = root0;
        for (uint64_t i = 0; i < count; i++) {
            uint32_t* dst = (i + 1 == count) ? d_out : buf[i & 1];
            MZK_TRY((div_run<P>(cur, cur_len, root.l, dst, nullptr, st)));
            cur = dst;
            cur_len--;
            root = root * g;
        }
    }
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
int32_t poly_eval_many_dispatch(int curve, const EvalJob* jobs, uint32_t n_jobs, const uint32_t* x_mont, uint32_t* out_host, hipStream_t st) {
    if (curve == 0) return eval_many_run<BlsFr>(jobs, n_jobs, x_mont, out_host, st);
    if (curve == 1) return eval_many_run<BnFr>(jobs, n_jobs, x_mont, out_host, st);
    set_error("unknown curve_id");
    return MZK_ERR_INVALID_ARG;
}
int32_t poly_div_dispatch(int curve, const uint32_t* d_poly, uint64_t len, const uint32_t* z_mont, uint32_t* d_out, uint32_t* d_rem, hipStream_t st) {
    if (curve == 0) return div_run<BlsFr>(d_poly, len, z_mont, d_out, d_rem, st);
    if (curve == 1) return div_run<BnFr>(d_poly, len, z_mont, d_out, d_rem, st);
    set_error("unknown curve_id");
    return MZK_ERR_INVALID_ARG;
}
int32_t poly_div_roots_dispatch(int curve, const uint32_t* d_poly, uint64_t len, uint32_t log_order, uint64_t first, uint64_t count, uint32_t* d_out, hipStream_t st) {
    if (curve == 0) return div_roots_run<BlsFr>(curve, d_poly, len, log_order, first, count, d_out, st);
    if (curve == 1) return div_roots_run<BnFr>(curve, d_poly, len, log_order, first, count, d_out, st);
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

int32_t poly_degree_dispatch(const uint32_t* d_poly, uint64_t len, unsigned long long* d_out, hipStream_t st) {
    HIP_TRY(hipMemsetAsync(d_out, 0, 8, st));
    if (len) hipLaunchKernelGGL(poly_degree_kernel, dim3((unsigned)((len + POLY_THREADS - 1) / POLY_THREADS)), dim3(POLY_THREADS), 0, st, d_poly, len, d_out);
    HIP_TRY(hipGetLastError());
    return MZK_OK;
}

int32_t wire_gather_dispatch(const uint32_t* d_witness, uint64_t n_vars, const uint32_t* d_vars, uint64_t count, uint32_t* d_out, hipStream_t st) {
    if (count == 0) return MZK_OK;
    hipLaunchKernelGGL(wire_gather_kernel, dim3((unsigned)((2 * count + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const uint4*>(d_witness),
                       (unsigned long long)n_vars, d_vars, (unsigned long long)count, reinterpret_cast<uint4*>(d_out), (unsigned int*)nullptr);
    HIP_TRY(hipGetLastError());
    return MZK_OK;
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

int32_t poly_split_quotient_dispatch(int curve, const uint32_t* d_q, uint64_t n, uint32_t W, const uint32_t* blind_mont, uint32_t* d_out, uint64_t stride, hipStream_t st) {
    if (W < 2 || W > SPLIT_MAX_ROWS || n + 5 <= W || stride < n + 3 || stride >= (1ull << 40)) { set_error("split_quotient: 2..8 parts, rows of at least n + 3 slots"); return MZK_ERR_INVALID_ARG; }
    SplitArgs a;
    a.q = d_q; a.out = d_out; a.n = n; a.stride = stride; a.expected = (uint64_t)W * (n + 1) + 2; a.W = (int)W;
    std::memset(a.blind, 0, sizeof a.blind);
    for (uint32_t i = 0; i + 1 < W; i++) std::memcpy(a.blind[i], blind_mont + (size_t)i * 8, 32);
    const dim3 grid((unsigned)((2 * stride + 255) / 256), W);
    if (curve == 0) hipLaunchKernelGGL((poly_split_quotient_kernel<BlsFr>), grid, dim3(256), 0, st, a);
    else if (curve == 1) hipLaunchKernelGGL((poly_split_quotient_kernel<BnFr>), grid, dim3(256), 0, st, a);
    else { set_error("unknown curve_id"); return MZK_ERR_INVALID_ARG; }
    HIP_TRY(hipGetLastError());
    return MZK_OK;
}

}  // namespace mzk
