// plonk.hip -- host side of the device-resident TurboPlonk quotient round (plonk.cuh).
#include <map>
#include <memory>

#include "internal.hpp"
#include "plonk.cuh"

namespace mzk {
namespace {

struct PlonkPk {
    int curve = 0, log_n = 0, W = 0;
    uint32_t* d_fixed = nullptr;        // [13 + W][m] coset evaluations of selectors then sigmas
    uint32_t* d_xs = nullptr;           // [m]
    uint32_t* d_inv_den = nullptr;      // [m]
    uint32_t* d_sigma_n = nullptr;      // [W][n] sigma_i on the gate domain H (extended permutation values)
    uint32_t* d_omega_n = nullptr;      // [n] w_n^j
    uint32_t k[PLK_WIRES][8];
    uint32_t zh_inv[PLK_RATIO][8];
    uint32_t gen[8];
};
std::map<uint64_t, std::unique_ptr<PlonkPk>> g_pks;
uint64_t g_next_pk = 1;

template <class P>
int32_t pk_build(PlonkPk& pk, const uint32_t* sel_coeffs, const uint32_t* sig_coeffs, uint64_t poly_len) {
    using F = Fp<P>;
    const int log_m = pk.log_n + 3;
    const uint64_t n = 1ull << pk.log_n, m = 1ull << log_m;
    const int nfix = PLK_SELECTORS + pk.W;
    hipStream_t st = nullptr;
    HIP_TRY(hipMalloc((void**)&pk.d_fixed, (size_t)nfix * m * 32));
    HIP_TRY(hipMalloc((void**)&pk.d_xs, m * 32));
    HIP_TRY(hipMalloc((void**)&pk.d_inv_den, m * 32));
    HIP_TRY(hipMemsetAsync(pk.d_fixed, 0, (size_t)nfix * m * 32, st));
    HIP_TRY(hipMemcpy2DAsync(pk.d_fixed, m * 32, sel_coeffs, poly_len * 32, poly_len * 32, PLK_SELECTORS, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpy2DAsync(pk.d_fixed + (size_t)PLK_SELECTORS * m * 8, m * 32, sig_coeffs, poly_len * 32, poly_len * 32, pk.W, hipMemcpyHostToDevice, st));
    for (int i = 0; i < 8; i++) pk.gen[i] = P::GENERATOR[i];
    // coset evaluations of the fixed polynomials, once per proving key (prover.rs:552-558 does it per proof)
    MZK_TRY(ntt_dispatch(pk.curve, pk.d_fixed, poly_len, log_m, false, pk.gen, nfix, m, st));
    // host constants: w_m, n, 1/Z_H on the 8 coset classes
    F w = F::from_const(P::ROOT);
    for (int i = log_m; i < P::TWO_ADICITY; i++) w = sqr(w);
    const F g = F::from_const(P::GENERATOR), nf = from_u64<P>(n);
    for (int i = 0; i < PLK_RATIO; i++) {
        F x = pow_u64(pow_u64(w, (uint64_t)i) * g, n) - F::one();
        F xi = inv(x);
        for (int q = 0; q < 8; q++) pk.zh_inv[i][q] = xi.l[q];
    }
    MZK_TRY(ws_acquire(st));
    MZK_TRY(g_ws.misc.reserve(64));
    uint32_t* d_c = g_ws.misc.as<uint32_t>();
    HIP_TRY(hipMemcpyAsync(d_c, w.l, 32, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_c + 8, nf.l, 32, hipMemcpyHostToDevice, st));
    const uint64_t threads = (m + 15) / 16;
    hipLaunchKernelGGL((plonk_domain_tables_kernel<P>), dim3((unsigned)((threads + PLK_THREADS - 1) / PLK_THREADS)), dim3(PLK_THREADS), 0, st,
                       d_c, d_c + 8, m, pk.d_xs, pk.d_inv_den);
    HIP_TRY(hipGetLastError());
    // gate-domain tables for the permutation product (round 2): sigma_i(w^j) and w^j
    HIP_TRY(hipMalloc((void**)&pk.d_sigma_n, (size_t)pk.W * n * 32));
    HIP_TRY(hipMalloc((void**)&pk.d_omega_n, n * 32));
    HIP_TRY(hipMemsetAsync(pk.d_sigma_n, 0, (size_t)pk.W * n * 32, st));
    const uint64_t sl = poly_len < n ? poly_len : n;
    HIP_TRY(hipMemcpy2DAsync(pk.d_sigma_n, n * 32, sig_coeffs, poly_len * 32, sl * 32, pk.W, hipMemcpyHostToDevice, st));
    F wn = F::from_const(P::ROOT);
    for (int i = pk.log_n; i < P::TWO_ADICITY; i++) wn = sqr(wn);
    HIP_TRY(hipMemcpyAsync(d_c + 8, wn.l, 32, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL((fr_powers_mont_kernel<P>), dim3((unsigned)(((n + 15) / 16 + PLK_THREADS - 1) / PLK_THREADS)), dim3(PLK_THREADS), 0, st,
                       d_c + 8, n, pk.d_omega_n);
    HIP_TRY(hipGetLastError());
    MZK_TRY(ws_release(st));
    MZK_TRY(ntt_dispatch(pk.curve, pk.d_sigma_n, sl, pk.log_n, false, nullptr, pk.W, n, st));
    HIP_TRY(hipStreamSynchronize(st));
    return MZK_OK;
}

// z = ifft(running product), constraint_system.rs:1197-1223; d_out receives the n coefficients
template <class P>
int32_t perm_product_run(const PlonkPk& pk, const uint32_t* d_wires, const uint32_t* beta, const uint32_t* gamma, uint32_t* d_out, hipStream_t st) {
    const uint64_t n = 1ull << pk.log_n;
    ProfScope total("plonk_perm_product", st);
    MZK_TRY(ws_acquire(st));
    const unsigned n_blocks = (unsigned)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
    MZK_TRY(g_ws.io.reserve(n * 32 + (size_t)n_blocks * 32));
    uint32_t* ratio = g_ws.io.as<uint32_t>();
    uint32_t* totals = ratio + n * 8;
    PermArgs a;
    a.wire = d_wires; a.sigma = pk.d_sigma_n; a.omega = pk.d_omega_n; a.ratio = ratio; a.n = n;
    std::memcpy(a.k, pk.k, sizeof a.k);
    std::memcpy(a.beta, beta, 32);
    std::memcpy(a.gamma, gamma, 32);
    const uint64_t rthreads = (n + PERM_B - 1) / PERM_B;
    hipLaunchKernelGGL((plonk_perm_ratio_kernel<P>), dim3((unsigned)((rthreads + PLK_THREADS - 1) / PLK_THREADS)), dim3(PLK_THREADS), 0, st, a);
    hipLaunchKernelGGL((fr_scan_mul_block_kernel<P>), dim3(n_blocks), dim3(SCAN_T), 0, st, ratio, n, totals);
    hipLaunchKernelGGL((fr_scan_mul_totals_kernel<P>), dim3(1), dim3(1024), 0, st, totals, n_blocks);
    hipLaunchKernelGGL((fr_scan_mul_apply_kernel<P>), dim3((unsigned)((n + PLK_THREADS - 1) / PLK_THREADS)), dim3(PLK_THREADS), 0, st, ratio, totals, n, d_out);
    HIP_TRY(hipGetLastError());
    MZK_TRY(ws_release(st));
    return ntt_dispatch(pk.curve, d_out, n, pk.log_n, true, nullptr, 1, n, st);
}

template <class P>
int32_t quotient_run(const PlonkPk& pk, uint32_t* d_polys, uint64_t in_len, const uint32_t* alpha, const uint32_t* beta,
                     const uint32_t* gamma, uint32_t* d_out, hipStream_t st) {
    using F = Fp<P>;
    const int log_m = pk.log_n + 3;
    const uint64_t m = 1ull << log_m;
    ProfScope total("plonk_quotient_total", st);
    // coset FFT of the W wires, z and the public-input polynomial (prover.rs:559-567), in place
    MZK_TRY(ntt_dispatch(pk.curve, d_polys, in_len, log_m, false, pk.gen, pk.W + 2, m, st));
    QuotientArgs a;
    a.sel = pk.d_fixed;
    a.sig = pk.d_fixed + (size_t)PLK_SELECTORS * m * 8;
    a.wire = d_polys;
    a.z = d_polys + (size_t)pk.W * m * 8;
    a.pi = d_polys + (size_t)(pk.W + 1) * m * 8;
    a.xs = pk.d_xs;
    a.inv_den = pk.d_inv_den;
    a.out = d_out;
    a.m = m;
    std::memcpy(a.k, pk.k, sizeof a.k);
    std::memcpy(a.zh_inv, pk.zh_inv, sizeof a.zh_inv);
    F al, a2;
    std::memcpy(al.l, alpha, 32);
    a2 = sqr(al);
    std::memcpy(a.alpha, alpha, 32);
    std::memcpy(a.alpha2, a2.l, 32);
    std::memcpy(a.beta, beta, 32);
    std::memcpy(a.gamma, gamma, 32);
    {
        ProfScope ps("plonk_quotient_kernel", st);
        hipLaunchKernelGGL((plonk_quotient_kernel<P>), dim3((unsigned)((m + PLK_THREADS - 1) / PLK_THREADS)), dim3(PLK_THREADS), 0, st, a);
        HIP_TRY(hipGetLastError());
    }
    // coefficient form: coset.ifft (prover.rs:672)
    MZK_TRY(ntt_dispatch(pk.curve, d_out, m, log_m, true, pk.gen, 1, m, st));
    return MZK_OK;
}

}  // namespace

int32_t plonk_pk_register(int curve, int log_n, int W, const uint32_t* sel, const uint32_t* sig, uint64_t poly_len, const uint32_t* k_mont,
                          uint64_t* out_handle) {
    if ((curve != 0 && curve != 1) || W != PLK_WIRES || log_n < 1 || log_n + 3 > (curve == 0 ? 32 : 28) || log_n + 3 > 30 ||
        poly_len == 0 || poly_len > (8ull << log_n) || !sel || !sig || !k_mont || !out_handle) {
        set_error("bad argument (TurboPlonk: 5 wire types, 13 selectors)");
        return MZK_ERR_INVALID_ARG;
    }
    auto pk = std::make_unique<PlonkPk>();
    pk->curve = curve; pk->log_n = log_n; pk->W = W;
    std::memcpy(pk->k, k_mont, sizeof pk->k);
    int32_t rc = curve == 0 ? pk_build<BlsFr>(*pk, sel, sig, poly_len) : pk_build<BnFr>(*pk, sel, sig, poly_len);
    if (rc != MZK_OK) {
        for (auto* d : {pk->d_fixed, pk->d_xs, pk->d_inv_den, pk->d_sigma_n, pk->d_omega_n}) if (d) (void)hipFree(d);
        return rc;
    }
    *out_handle = g_next_pk++;
    g_pks[*out_handle] = std::move(pk);
    return MZK_OK;
}

int32_t plonk_pk_release(uint64_t handle) {
    auto it = g_pks.find(handle);
    if (it == g_pks.end()) { set_error("unknown proving-key handle"); return MZK_ERR_BAD_HANDLE; }
    HIP_TRY(hipDeviceSynchronize());
    for (auto* d : {it->second->d_fixed, it->second->d_xs, it->second->d_inv_den, it->second->d_sigma_n, it->second->d_omega_n}) if (d) (void)hipFree(d);
    g_pks.erase(it);
    return MZK_OK;
}
void plonk_release_all() {
    for (auto& kv : g_pks)
        for (auto* d : {kv.second->d_fixed, kv.second->d_xs, kv.second->d_inv_den, kv.second->d_sigma_n, kv.second->d_omega_n}) if (d) (void)hipFree(d);
    g_pks.clear();
}

int32_t plonk_quotient_dev(uint64_t handle, uint32_t* d_polys, uint64_t in_len, const uint32_t* alpha, const uint32_t* beta, const uint32_t* gamma,
                           uint32_t* d_out, hipStream_t st) {
    auto it = g_pks.find(handle);
    if (it == g_pks.end()) { set_error("unknown proving-key handle"); return MZK_ERR_BAD_HANDLE; }
    const PlonkPk& pk = *it->second;
    if (!d_polys || !d_out || !alpha || !beta || !gamma || in_len > (8ull << pk.log_n)) { set_error("bad argument"); return MZK_ERR_INVALID_ARG; }
    return pk.curve == 0 ? quotient_run<BlsFr>(pk, d_polys, in_len, alpha, beta, gamma, d_out, st)
                         : quotient_run<BnFr>(pk, d_polys, in_len, alpha, beta, gamma, d_out, st);
}
int32_t plonk_perm_product_dev(uint64_t handle, const uint32_t* d_wires, const uint32_t* beta, const uint32_t* gamma, uint32_t* d_out, hipStream_t st) {
    auto it = g_pks.find(handle);
    if (it == g_pks.end()) { set_error("unknown proving-key handle"); return MZK_ERR_BAD_HANDLE; }
    if (!d_wires || !d_out || !beta || !gamma) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    const PlonkPk& pk = *it->second;
    return pk.curve == 0 ? perm_product_run<BlsFr>(pk, d_wires, beta, gamma, d_out, st) : perm_product_run<BnFr>(pk, d_wires, beta, gamma, d_out, st);
}
int plonk_pk_log_n(uint64_t handle) {
    auto it = g_pks.find(handle);
    return it == g_pks.end() ? -1 : it->second->log_n;
}
int plonk_pk_wires(uint64_t handle) {
    auto it = g_pks.find(handle);
    return it == g_pks.end() ? -1 : it->second->W;
}

}  // namespace mzk
