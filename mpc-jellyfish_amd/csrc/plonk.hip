// plonk.hip -- host side of the device-resident TurboPlonk quotient round (plonk.cuh).
#include <array>
#include <map>
#include <vector>
#include <memory>

#include "internal.hpp"
#include "plonk.cuh"
#include "plookup.cuh"

namespace mzk {
namespace {

struct PlonkPk {
    int curve = 0, log_n = 0, W = 0;
    bool ultra = false;                 // UltraPlonk: 14 selectors (q_lookup last), 6 wire types, 4 table polynomials
    int nsel = PLK_SELECTORS;
    uint32_t* d_fixed = nullptr;        // [nsel + W (+ 4)][m] coset evaluations of selectors, sigmas (, range, key, table_dom_sep, q_dom_sep)
    uint32_t* d_xs = nullptr;           // [m]
    uint32_t* d_inv_den = nullptr;      // [m]  1 / (n (x - 1))
    uint32_t* d_inv_den_n = nullptr;    // [m]  w^-1 / (n (x - w^-1))   (UltraPlonk)
    uint32_t* d_sigma_n = nullptr;      // [W][n] sigma_i on the gate domain H (extended permutation values)
    uint32_t* d_omega_n = nullptr;      // [n] w_n^j
    uint32_t* d_tab_n = nullptr;        // [5][n] range, key, table_dom_sep, q_dom_sep, q_lookup on H (UltraPlonk)
    uint32_t sel_zero = 0;              // bit j: selector polynomial j is identically zero (QuotientArgs::sel_zero)
    uint32_t* d_top_fixed = nullptr;    // [W + 5][8] coefficients n-8 .. n-1 of sigma_0..W-1, q_hash_0..3, q_ecc (chunked keys: plonk_quotient_top_kernel)
    uint32_t k[PLK_MAX_WIRES][8];
    uint32_t zh_inv[PLK_RATIO][8];
    uint32_t gen[8];
    uint32_t w_inv[8];                  // w_n^-1
    // coset-chunked key: only the residue classes mod 8 listed in cls are resident, class-major
    // ([poly][local class][n]); empty = the whole 8n-point domain in natural order
    std::vector<int> cls;
    uint32_t h_cls[PLK_RATIO][8];       // h_k = g * w_8n^k
    uint32_t c_cls[PLK_RATIO][8];       // h_k^n
    std::array<uint32_t*, 8> bufs() const { return {d_fixed, d_xs, d_inv_den, d_inv_den_n, d_sigma_n, d_omega_n, d_tab_n, d_top_fixed}; }
};
std::map<uint64_t, std::unique_ptr<PlonkPk>> g_pks_of[MAX_CTX];              // proving keys per device context; the handle names its context
#define g_pks (g_pks_of[cur().logical])
std::atomic<uint64_t> g_next_pk{1};              // shared by the device threads of one process

// data[i] *= c (boundary form -> internal form x * R' with c = 32: plonk.cuh)
template <class P>
__global__ __launch_bounds__(PLK_THREADS) void fr_scale_kernel(uint32_t* __restrict__ data, unsigned long long n, const uint32_t* __restrict__ c_mont) {
    const unsigned long long i = (unsigned long long)blockIdx.x * PLK_THREADS + threadIdx.x;
    if (i >= n) return;
    store_fp<P>(data + i * 8, load_fp<P>(data + i * 8) * load_fp<P>(c_mont));
}
template <class P>
void to_internal(uint32_t (&dst)[8], const uint32_t* src_mont) {
    Fp<P> v;
    std::memcpy(v.l, src_mont, 32);
    v = v * from_u64<P>(32);
    std::memcpy(dst, v.l, 32);
}

// out[i] = 1 / (scale * (xs[i] - c)), 16 points per thread with one shared inversion
template <class P>
__global__ __launch_bounds__(PLK_THREADS) void plonk_shifted_inverse_kernel(const uint32_t* __restrict__ xs, unsigned long long m, const uint32_t* __restrict__ c_mont,
                                                                            const uint32_t* __restrict__ scale_mont, uint32_t* __restrict__ out) {
    using F = Fp<P>;
    constexpr int B = 16;
    const unsigned long long start = ((unsigned long long)blockIdx.x * PLK_THREADS + threadIdx.x) * B;
    if (start >= m) return;
    const F c = load_fp<P>(c_mont), sc = load_fp<P>(scale_mont);
    const int cnt = (int)(start + B <= m ? B : m - start);
    F pref[B];
    F run = F::one();
    for (int j = 0; j < cnt; j++) {
        const F d = sc * (load_fp<P>(xs + (start + j) * 8) - c);
        pref[j] = run;
        run = run * d;
        store_fp<P>(out + (start + j) * 8, d);
    }
    F inv_run = inv(run);
    for (int j = cnt - 1; j >= 0; j--) {
        const F d = load_fp<P>(out + (start + j) * 8);
        store_fp<P>(out + (start + j) * 8, inv_run * pref[j]);
        inv_run = inv_run * d;
    }
}

template <class P>
int32_t pk_build(PlonkPk& pk, const uint32_t* sel_coeffs, const uint32_t* sig_coeffs, const uint32_t* tab_coeffs, uint64_t poly_len) {
    using F = Fp<P>;
    const int log_m = pk.log_n + 3;
    const uint64_t n = 1ull << pk.log_n, m = 1ull << log_m;
    const int nfix = pk.nsel + pk.W + (pk.ultra ? 4 : 0);
    hipStream_t st = nullptr;
    HIP_TRY(hipMalloc((void**)&pk.d_fixed, (size_t)nfix * m * 32));
    HIP_TRY(hipMalloc((void**)&pk.d_xs, m * 32));
    HIP_TRY(hipMalloc((void**)&pk.d_inv_den, m * 32));
    HIP_TRY(hipMemsetAsync(pk.d_fixed, 0, (size_t)nfix * m * 32, st));
    HIP_TRY(hipMemcpy2DAsync(pk.d_fixed, m * 32, sel_coeffs, poly_len * 32, poly_len * 32, pk.nsel, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpy2DAsync(pk.d_fixed + (size_t)pk.nsel * m * 8, m * 32, sig_coeffs, poly_len * 32, poly_len * 32, pk.W, hipMemcpyHostToDevice, st));
    if (pk.ultra)
        HIP_TRY(hipMemcpy2DAsync(pk.d_fixed + (size_t)(pk.nsel + pk.W) * m * 8, m * 32, tab_coeffs, poly_len * 32, poly_len * 32, 4, hipMemcpyHostToDevice, st));
    for (int i = 0; i < 8; i++) pk.gen[i] = P::GENERATOR[i];
    // coset evaluations of the fixed polynomials, once per proving key (prover.rs:552-558, 577-584 do it per proof)
    // ... and left in the internal form x * R' the quotient kernels compute in (plonk.cuh)
    MZK_TRY(ntt_dispatch(pk.curve, pk.d_fixed, poly_len, log_m, false, pk.gen, nfix, m, st, 1));
    // host constants: w_m, n, 1/Z_H on the 8 coset classes
    F w = F::from_const(P::ROOT);
    for (int i = log_m; i < P::TWO_ADICITY; i++) w = sqr(w);
    const F g = F::from_const(P::GENERATOR), nf = from_u64<P>(n);
    for (int i = 0; i < PLK_RATIO; i++) {
        F x = pow_u64(pow_u64(w, (uint64_t)i) * g, n) - F::one();
        F xi = inv(x);
        for (int q = 0; q < 8; q++) pk.zh_inv[i][q] = xi.l[q];
    }
    F wn = F::from_const(P::ROOT);
    for (int i = pk.log_n; i < P::TWO_ADICITY; i++) wn = sqr(wn);
    const F wn_inv = inv(wn);
    for (int q = 0; q < 8; q++) pk.w_inv[q] = wn_inv.l[q];
    MZK_TRY(ws_acquire(st));
    MZK_TRY(g_ws.misc.reserve(128));
    uint32_t* d_c = g_ws.misc.as<uint32_t>();
    HIP_TRY(hipMemcpyAsync(d_c, w.l, 32, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_c + 8, nf.l, 32, hipMemcpyHostToDevice, st));
    const uint64_t threads = (m + 15) / 16;
    hipLaunchKernelGGL((plonk_domain_tables_kernel<P>), dim3((unsigned)((threads + PLK_THREADS - 1) / PLK_THREADS)), dim3(PLK_THREADS), 0, st,
                       d_c, d_c + 8, m, pk.d_xs, pk.d_inv_den);
    HIP_TRY(hipGetLastError());
    if (pk.ultra) {
        // L_n(x) / Z_H(x) = w^-1 / (n (x - w^-1)) = 1 / (n w (x - w^-1))   (prover.rs:794-795)
        HIP_TRY(hipMalloc((void**)&pk.d_inv_den_n, m * 32));
        const F scale = nf * wn;
        HIP_TRY(hipMemcpyAsync(d_c + 16, wn_inv.l, 32, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_c + 24, scale.l, 32, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL((plonk_shifted_inverse_kernel<P>), dim3((unsigned)((threads + PLK_THREADS - 1) / PLK_THREADS)), dim3(PLK_THREADS), 0, st,
                           pk.d_xs, m, d_c + 16, d_c + 24, pk.d_inv_den_n);
        HIP_TRY(hipGetLastError());
    }
    {   // the per-point tables go to the internal form too
        const F c32 = from_u64<P>(32);
        HIP_TRY(hipMemcpyAsync(d_c + 8, c32.l, 32, hipMemcpyHostToDevice, st));
        const unsigned sg = (unsigned)((m + PLK_THREADS - 1) / PLK_THREADS);
        hipLaunchKernelGGL((fr_scale_kernel<P>), dim3(sg), dim3(PLK_THREADS), 0, st, pk.d_xs, m, d_c + 8);
        hipLaunchKernelGGL((fr_scale_kernel<P>), dim3(sg), dim3(PLK_THREADS), 0, st, pk.d_inv_den, m, d_c + 8);
        if (pk.ultra) hipLaunchKernelGGL((fr_scale_kernel<P>), dim3(sg), dim3(PLK_THREADS), 0, st, pk.d_inv_den_n, m, d_c + 8);
        HIP_TRY(hipGetLastError());
    }
    // gate-domain tables for the grand products (rounds 2, 2.5): sigma_i(w^j), w^j and the table polynomials' values
    HIP_TRY(hipMalloc((void**)&pk.d_sigma_n, (size_t)pk.W * n * 32));
    HIP_TRY(hipMalloc((void**)&pk.d_omega_n, n * 32));
    HIP_TRY(hipMemsetAsync(pk.d_sigma_n, 0, (size_t)pk.W * n * 32, st));
    const uint64_t sl = poly_len < n ? poly_len : n;
    HIP_TRY(hipMemcpy2DAsync(pk.d_sigma_n, n * 32, sig_coeffs, poly_len * 32, sl * 32, pk.W, hipMemcpyHostToDevice, st));
    launch_powers<P>(st, &wn, 1, n, &pk.d_omega_n);
    HIP_TRY(hipGetLastError());
    MZK_TRY(ws_release(st));
    MZK_TRY(ntt_dispatch(pk.curve, pk.d_sigma_n, sl, pk.log_n, false, nullptr, pk.W, n, st));
    if (pk.ultra) {
        HIP_TRY(hipMalloc((void**)&pk.d_tab_n, (size_t)5 * n * 32));
        HIP_TRY(hipMemsetAsync(pk.d_tab_n, 0, (size_t)5 * n * 32, st));
        HIP_TRY(hipMemcpy2DAsync(pk.d_tab_n, n * 32, tab_coeffs, poly_len * 32, sl * 32, 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(pk.d_tab_n + (size_t)4 * n * 8, sel_coeffs + (size_t)13 * poly_len * 8, sl * 32, hipMemcpyHostToDevice, st));
        MZK_TRY(ntt_dispatch(pk.curve, pk.d_tab_n, sl, pk.log_n, false, nullptr, 5, n, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    return MZK_OK;
}

// prefix product of ratio[0..n) into d_out: out[0] = 1, out[j+1] = prod_{i<=j} ratio[i]; then iFFT
template <class P> struct FxOf;
template <> struct FxOf<BlsFr> { using type = BlsFrX; };
template <> struct FxOf<BnFr> { using type = BnFrX; };

template <class P>
int32_t scan_and_interpolate(const PlonkPk& pk, uint32_t* ratio, uint32_t* totals, unsigned n_blocks, bool last_one, uint32_t* d_out, hipStream_t st) {
    const uint64_t n = 1ull << pk.log_n;
    hipLaunchKernelGGL((fr_scan_mul_block_kernel<P>), dim3(n_blocks), dim3(SCAN_T), 0, st, ratio, n, totals);
    hipLaunchKernelGGL((fr_scan_mul_totals_kernel<P>), dim3(1), dim3(1024), 0, st, totals, n_blocks);
    hipLaunchKernelGGL((fr_scan_mul_apply_kernel<P>), dim3((unsigned)((n + PLK_THREADS - 1) / PLK_THREADS)), dim3(PLK_THREADS), 0, st, ratio, totals, n, d_out);
    if (last_one) hipLaunchKernelGGL((plookup_set_last_one_kernel<P>), dim3(1), dim3(64), 0, st, d_out, n);
    HIP_TRY(hipGetLastError());
    return MZK_OK;
}

// z = ifft(running product), constraint_system.rs:1197-1223; d_out receives the n coefficients
template <class P>
int32_t perm_product_run(const PlonkPk& pk, const uint32_t* d_wires, const uint32_t* beta, const uint32_t* gamma, uint32_t* d_out, hipStream_t st) {
    const uint64_t n = 1ull << pk.log_n;
    ProfScope total("plonk_perm_product", st);
    MZK_TRY(ws_acquire(st));
    const unsigned n_blocks = (unsigned)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
    MZK_TRY(g_ws.io.reserve(3 * n * 32 + (size_t)n_blocks * 32));
    uint32_t* ratio = g_ws.io.as<uint32_t>();
    uint32_t* den = ratio + n * 8;
    uint32_t* pref = den + n * 8;
    uint32_t* totals = pref + n * 8;
    PermArgs a;
    a.wire = d_wires; a.sigma = pk.d_sigma_n; a.omega = pk.d_omega_n; a.ratio = ratio; a.den = den; a.n = n; a.W = pk.W;
    std::memcpy(a.k, pk.k, sizeof a.k);
    std::memcpy(a.beta, beta, 32);
    std::memcpy(a.gamma, gamma, 32);
    hipLaunchKernelGGL((plonk_perm_terms_kernel<P>), dim3((unsigned)((n + PLK_THREADS - 1) / PLK_THREADS)), dim3(PLK_THREADS), 0, st, a);
    const unsigned long long T = batch_div_threads(n);
    hipLaunchKernelGGL((fr_batch_div_kernel<P, typename FxOf<P>::type>), dim3((unsigned)((T + PLK_THREADS - 1) / PLK_THREADS)), dim3(PLK_THREADS), 0, st,
                       ratio, den, n, T, pref);
    MZK_TRY(scan_and_interpolate<P>(pk, ratio, totals, n_blocks, false, d_out, st));
    MZK_TRY(ws_release(st));
    return ntt_dispatch(pk.curve, d_out, n, pk.log_n, true, nullptr, 1, n, st);
}

// merged table, merged lookup witness and their sorted concatenation (constraint_system.rs:1290-1309, 1370-1417).
// Synchronises: the "lookup value outside the table" condition is read back.
template <class P>
int32_t sorted_vec_run(const PlonkPk& pk, const uint32_t* d_wires, const uint32_t* tau, uint32_t* d_table, uint32_t* d_lookup, uint32_t* d_sorted, hipStream_t st) {
    const uint64_t n = 1ull << pk.log_n;
    ProfScope total("plookup_sorted_vec", st);
    MergeArgs ma;
    ma.wire = d_wires;
    ma.range = pk.d_tab_n; ma.key = pk.d_tab_n + n * 8; ma.table_dom_sep = pk.d_tab_n + 2 * n * 8; ma.q_dom_sep = pk.d_tab_n + 3 * n * 8;
    ma.q_lookup = pk.d_tab_n + 4 * n * 8;
    ma.table = d_table; ma.lookup = d_lookup; ma.n = n;
    std::memcpy(ma.tau, tau, 32);
    const unsigned gn = (unsigned)((n + PLK_THREADS - 1) / PLK_THREADS);
    hipLaunchKernelGGL((plookup_merge_kernel<P>), dim3(gn), dim3(PLK_THREADS), 0, st, ma);
    uint64_t slots = 4;
    while (slots < 4 * n) slots <<= 1;
    const unsigned n_blocks = (unsigned)((n + PLK_SCAN_BLOCK - 1) / PLK_SCAN_BLOCK);
    MZK_TRY(ws_acquire(st));
    MZK_TRY(g_ws.io.reserve(slots * 4 + n * 4 + n * 4 + (size_t)n_blocks * 4 + 64));
    uint32_t* d_slots = g_ws.io.as<uint32_t>();
    uint32_t* d_count = d_slots + slots;
    uint32_t* d_pos = d_count + n;
    uint32_t* d_totals = d_pos + n;
    uint32_t* d_missing = d_totals + n_blocks;
    HIP_TRY(hipMemsetAsync(d_slots, 0xFF, slots * 4, st));
    HIP_TRY(hipMemsetAsync(d_count, 0, n * 4, st));
    HIP_TRY(hipMemsetAsync(d_missing, 0, 4, st));
    const uint4* t4 = reinterpret_cast<const uint4*>(d_table);
    hipLaunchKernelGGL(plookup_hash_insert_kernel, dim3(gn), dim3(PLK_THREADS), 0, st, t4, n, d_slots, (uint32_t)(slots - 1));
    // only the first n-1 rows are looked up (constraint_system.rs:1383, 1388)
    hipLaunchKernelGGL(plookup_hash_count_kernel, dim3(gn), dim3(PLK_THREADS), 0, st, t4, reinterpret_cast<const uint4*>(d_lookup), n - 1, d_slots,
                       (uint32_t)(slots - 1), d_count, d_missing);
    hipLaunchKernelGGL(plookup_scan_block_kernel, dim3(n_blocks), dim3(PLK_SCAN_T), 0, st, d_count, n, d_pos, d_totals);
    hipLaunchKernelGGL(plookup_scan_totals_kernel, dim3(1), dim3(1024), 0, st, d_totals, n_blocks);
    hipLaunchKernelGGL(plookup_scan_apply_kernel, dim3(gn), dim3(PLK_THREADS), 0, st, d_pos, d_totals, n);
    const uint64_t out_len = 2 * n - 1;
    hipLaunchKernelGGL(plookup_gather_kernel, dim3((unsigned)((out_len + PLK_THREADS - 1) / PLK_THREADS)), dim3(PLK_THREADS), 0, st, t4, d_pos, n, out_len,
                       reinterpret_cast<uint4*>(d_sorted));
    HIP_TRY(hipGetLastError());
    uint32_t missing = 0;
    HIP_TRY(hipMemcpyAsync(&missing, d_missing, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    MZK_TRY(ws_release(st));
    if (missing) {
        set_error("The sorted vector has wrong length, some lookup variables might be outside the table");      // constraint_system.rs:1410-1412
        return MZK_ERR_LOOKUP;
    }
    return MZK_OK;
}

// Plookup product polynomial (constraint_system.rs:1311-1368): n coefficients into d_out
template <class P>
int32_t lookup_product_run(const PlonkPk& pk, const uint32_t* d_table, const uint32_t* d_lookup, const uint32_t* d_sorted, const uint32_t* beta,
                           const uint32_t* gamma, uint32_t* d_out, hipStream_t st) {
    const uint64_t n = 1ull << pk.log_n;
    ProfScope total("plookup_product", st);
    MZK_TRY(ws_acquire(st));
    const unsigned n_blocks = (unsigned)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
    MZK_TRY(g_ws.io.reserve(3 * n * 32 + (size_t)n_blocks * 32));
    uint32_t* ratio = g_ws.io.as<uint32_t>();
    uint32_t* den = ratio + n * 8;
    uint32_t* pref = den + n * 8;
    uint32_t* totals = pref + n * 8;
    LookupProdArgs a;
    a.table = d_table; a.lookup = d_lookup; a.sorted = d_sorted; a.ratio = ratio; a.den = den; a.n = n;
    std::memcpy(a.beta, beta, 32);
    std::memcpy(a.gamma, gamma, 32);
    hipLaunchKernelGGL((plookup_terms_kernel<P>), dim3((unsigned)((n + PLK_THREADS - 1) / PLK_THREADS)), dim3(PLK_THREADS), 0, st, a);
    const unsigned long long T = batch_div_threads(n);
    hipLaunchKernelGGL((fr_batch_div_kernel<P, typename FxOf<P>::type>), dim3((unsigned)((T + PLK_THREADS - 1) / PLK_THREADS)), dim3(PLK_THREADS), 0, st,
                       ratio, den, n, T, pref);
    MZK_TRY(scan_and_interpolate<P>(pk, ratio, totals, n_blocks, true, d_out, st));
    MZK_TRY(ws_release(st));
    return ntt_dispatch(pk.curve, d_out, n, pk.log_n, true, nullptr, 1, n, st);
}

// challenges and per-key constants of QuotientArgs, converted to the internal form x * R' (x32)
template <class P>
void fill_quotient_constants(QuotientArgs& a, const PlonkPk& pk, const uint32_t* tau, const uint32_t* alpha, const uint32_t* beta, const uint32_t* gamma) {
    using F = Fp<P>;
    for (int j = 0; j < PLK_MAX_WIRES; j++) to_internal<P>(a.k[j], pk.k[j]);
    for (int j = 0; j < PLK_RATIO; j++) to_internal<P>(a.zh_inv[j], pk.zh_inv[j]);
    F al, a2;
    std::memcpy(al.l, alpha, 32);
    a2 = sqr(al);
    to_internal<P>(a.alpha, alpha);
    to_internal<P>(a.alpha2, a2.l);
    to_internal<P>(a.beta, beta);
    to_internal<P>(a.gamma, gamma);
    a.sel_zero = pk.sel_zero;
    if (pk.ultra) {
        const F a3 = a2 * al;
        to_internal<P>(a.tau, tau);
        to_internal<P>(a.alpha3, a3.l);
        to_internal<P>(a.w_inv, pk.w_inv);
    }
}

// d_polys rows: W wires, z, public input (, h_1, h_2, Plookup product)
template <class P>
int32_t quotient_run(const PlonkPk& pk, uint32_t* d_polys, uint64_t in_len, const uint32_t* tau, const uint32_t* alpha, const uint32_t* beta,
                     const uint32_t* gamma, uint32_t* d_out, hipStream_t st) {
    using F = Fp<P>;
    const int log_m = pk.log_n + 3;
    const uint64_t m = 1ull << log_m;
    ProfScope total("plonk_quotient_total", st);
    // coset FFT of the W wires, z and the public-input polynomial (prover.rs:559-567; Plookup oracles :585-590), in place
    // -- evaluations left in the internal form x * R' of the kernels (plonk.cuh)
    MZK_TRY(ntt_dispatch(pk.curve, d_polys, in_len, log_m, false, pk.gen, pk.W + 2 + (pk.ultra ? 3 : 0), m, st, 1));
    QuotientArgs a;
    a.n_cls = 0;
    a.sel = pk.d_fixed;
    a.sig = pk.d_fixed + (size_t)pk.nsel * m * 8;
    a.wire = d_polys;
    a.z = d_polys + (size_t)pk.W * m * 8;
    a.pi = d_polys + (size_t)(pk.W + 1) * m * 8;
    a.xs = pk.d_xs;
    a.inv_den = pk.d_inv_den;
    a.out = d_out;
    a.m = m; a.fstride = m; a.ostride = m; a.next_off = PLK_RATIO; a.zh_class = -1;
    fill_quotient_constants<P>(a, pk, tau, alpha, beta, gamma);
    a.tab = a.h = a.pl = a.inv_den_n = nullptr;
    if (pk.ultra) {
        a.tab = pk.d_fixed + (size_t)(pk.nsel + pk.W) * m * 8;
        a.h = d_polys + (size_t)(pk.W + 2) * m * 8;
        a.pl = d_polys + (size_t)(pk.W + 4) * m * 8;
        a.inv_den_n = pk.d_inv_den_n;
    }
    {
        ProfScope ps("plonk_quotient_kernel", st);
        const dim3 grid((unsigned)((m + PLK_THREADS - 1) / PLK_THREADS));
        if (pk.ultra) {
            hipLaunchKernelGGL((plonk_quotient_kernel<typename FxOf<P>::type, true>), grid, dim3(PLK_THREADS), 0, st, a);
            hipLaunchKernelGGL((plonk_quotient_lookup_kernel<typename FxOf<P>::type>), grid, dim3(PLK_THREADS), 0, st, a);
        } else {
            hipLaunchKernelGGL((plonk_quotient_kernel<typename FxOf<P>::type, false>), grid, dim3(PLK_THREADS), 0, st, a);
        }
        HIP_TRY(hipGetLastError());
    }
    // coefficient form: coset.ifft (prover.rs:672)
    MZK_TRY(ntt_dispatch(pk.curve, d_out, m, log_m, true, pk.gen, 1, m, st, 2));          // internal form in, boundary form out
    return MZK_OK;
}

// xs[j] = h * w^j, 16 points per thread
template <class P>
__global__ __launch_bounds__(PLK_THREADS) void plonk_class_points_kernel(const uint32_t* __restrict__ w_mont, const uint32_t* __restrict__ h_mont,
                                                                         unsigned long long n, uint32_t* __restrict__ xs) {
    using F = Fp<P>;
    const unsigned long long start = ((unsigned long long)blockIdx.x * PLK_THREADS + threadIdx.x) * 16;
    if (start >= n) return;
    const F w = load_fp<P>(w_mont);
    F x = pow_u64(w, start) * load_fp<P>(h_mont);
    for (int j = 0; j < 16 && start + j < n; j++) {
        store_fp<P>(xs + (start + j) * 8, x);
        x = x * w;
    }
}

// proving key holding only the residue classes `pk.cls` of the quotient domain (each class = the coset h_k * H_n)
template <class P>
int32_t pk_build_chunked(PlonkPk& pk, const uint32_t* sel_coeffs, const uint32_t* sig_coeffs, const uint32_t* tab_coeffs, uint64_t poly_len) {
    using F = Fp<P>;
    const int log_m = pk.log_n + 3;
    const uint64_t n = 1ull << pk.log_n;
    const int nfix = pk.nsel + pk.W + (pk.ultra ? 4 : 0);
    const size_t ncl = pk.cls.size();
    hipStream_t st = nullptr;
    for (int i = 0; i < 8; i++) pk.gen[i] = P::GENERATOR[i];
    F w = F::from_const(P::ROOT);
    for (int i = log_m; i < P::TWO_ADICITY; i++) w = sqr(w);
    F wn = F::from_const(P::ROOT);
    for (int i = pk.log_n; i < P::TWO_ADICITY; i++) wn = sqr(wn);
    const F g = F::from_const(P::GENERATOR), nf = from_u64<P>(n), wn_inv = inv(wn);
    for (int q = 0; q < 8; q++) pk.w_inv[q] = wn_inv.l[q];
    for (int i = 0; i < PLK_RATIO; i++) {
        const F h = pow_u64(w, (uint64_t)i) * g, c = pow_u64(h, n);
        const F zi = inv(c - F::one());
        for (int q = 0; q < 8; q++) { pk.h_cls[i][q] = h.l[q]; pk.c_cls[i][q] = c.l[q]; pk.zh_inv[i][q] = zi.l[q]; }
    }
    HIP_TRY(hipMalloc((void**)&pk.d_fixed, (size_t)nfix * ncl * n * 32));
    HIP_TRY(hipMalloc((void**)&pk.d_xs, ncl * n * 32));
    HIP_TRY(hipMalloc((void**)&pk.d_inv_den, ncl * n * 32));
    if (pk.ultra) HIP_TRY(hipMalloc((void**)&pk.d_inv_den_n, ncl * n * 32));
    HIP_TRY(hipMemsetAsync(pk.d_fixed, 0, (size_t)nfix * ncl * n * 32, st));
    const uint64_t sl = poly_len < n ? poly_len : n;               // fixed polynomials have degree < n
    MZK_TRY(ws_acquire(st));
    MZK_TRY(g_ws.misc.reserve(512));
    uint32_t* d_c = g_ws.misc.as<uint32_t>();
    const F one = F::one(), scale_n = nf * wn;
    HIP_TRY(hipMemcpyAsync(d_c, wn.l, 32, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_c + 8, nf.l, 32, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_c + 16, one.l, 32, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_c + 24, wn_inv.l, 32, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_c + 32, scale_n.l, 32, hipMemcpyHostToDevice, st));
    const unsigned tg = (unsigned)(((n + 15) / 16 + PLK_THREADS - 1) / PLK_THREADS);
    for (size_t lc = 0; lc < ncl; lc++) {
        const int k = pk.cls[lc];
        uint32_t* base = pk.d_fixed + lc * n * 8;
        const size_t row = ncl * n * 32;                              // bytes between consecutive polynomials
        HIP_TRY(hipMemcpy2DAsync(base, row, sel_coeffs, poly_len * 32, sl * 32, pk.nsel, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpy2DAsync(base + (size_t)pk.nsel * ncl * n * 8, row, sig_coeffs, poly_len * 32, sl * 32, pk.W, hipMemcpyHostToDevice, st));
        if (pk.ultra)
            HIP_TRY(hipMemcpy2DAsync(base + (size_t)(pk.nsel + pk.W) * ncl * n * 8, row, tab_coeffs, poly_len * 32, sl * 32, 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_c + 40, pk.h_cls[k], 32, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL((plonk_class_points_kernel<P>), dim3(tg), dim3(PLK_THREADS), 0, st, d_c, d_c + 40, n, pk.d_xs + lc * n * 8);
        hipLaunchKernelGGL((plonk_shifted_inverse_kernel<P>), dim3(tg), dim3(PLK_THREADS), 0, st, pk.d_xs + lc * n * 8, n, d_c + 16, d_c + 8,
                           pk.d_inv_den + lc * n * 8);
        if (pk.ultra)
            hipLaunchKernelGGL((plonk_shifted_inverse_kernel<P>), dim3(tg), dim3(PLK_THREADS), 0, st, pk.d_xs + lc * n * 8, n, d_c + 24, d_c + 32,
                               pk.d_inv_den_n + lc * n * 8);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));                            // d_c + 40 is rewritten for the next class
    }
    {   // per-point tables in the internal form x * R' (plonk.cuh)
        const F c32 = from_u64<P>(32);
        HIP_TRY(hipMemcpyAsync(d_c + 48, c32.l, 32, hipMemcpyHostToDevice, st));
        const unsigned sg = (unsigned)((ncl * n + PLK_THREADS - 1) / PLK_THREADS);
        hipLaunchKernelGGL((fr_scale_kernel<P>), dim3(sg), dim3(PLK_THREADS), 0, st, pk.d_xs, ncl * n, d_c + 48);
        hipLaunchKernelGGL((fr_scale_kernel<P>), dim3(sg), dim3(PLK_THREADS), 0, st, pk.d_inv_den, ncl * n, d_c + 48);
        if (pk.ultra) hipLaunchKernelGGL((fr_scale_kernel<P>), dim3(sg), dim3(PLK_THREADS), 0, st, pk.d_inv_den_n, ncl * n, d_c + 48);
        HIP_TRY(hipGetLastError());
    }
    // gate-domain tables for the grand products (replicated on every rank)
    HIP_TRY(hipMalloc((void**)&pk.d_sigma_n, (size_t)pk.W * n * 32));
    HIP_TRY(hipMalloc((void**)&pk.d_omega_n, n * 32));
    HIP_TRY(hipMemsetAsync(pk.d_sigma_n, 0, (size_t)pk.W * n * 32, st));
    HIP_TRY(hipMemcpy2DAsync(pk.d_sigma_n, n * 32, sig_coeffs, poly_len * 32, sl * 32, pk.W, hipMemcpyHostToDevice, st));
    launch_powers<P>(st, &wn, 1, n, &pk.d_omega_n);
    HIP_TRY(hipGetLastError());
    MZK_TRY(ws_release(st));
    // class evaluations of the fixed polynomials: size-n coset NTTs with offset h_k, all polynomials of a class in one batch
    for (size_t lc = 0; lc < ncl; lc++)
        MZK_TRY(ntt_dispatch(pk.curve, pk.d_fixed + lc * n * 8, sl, pk.log_n, false, pk.h_cls[pk.cls[lc]], nfix, ncl * n, st, 1));     // internal form
    MZK_TRY(ntt_dispatch(pk.curve, pk.d_sigma_n, sl, pk.log_n, false, nullptr, pk.W, n, st));
    if (pk.ultra) {
        HIP_TRY(hipMalloc((void**)&pk.d_tab_n, (size_t)5 * n * 32));
        HIP_TRY(hipMemsetAsync(pk.d_tab_n, 0, (size_t)5 * n * 32, st));
        HIP_TRY(hipMemcpy2DAsync(pk.d_tab_n, n * 32, tab_coeffs, poly_len * 32, sl * 32, 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(pk.d_tab_n + (size_t)4 * n * 8, sel_coeffs + (size_t)13 * poly_len * 8, sl * 32, hipMemcpyHostToDevice, st));
        MZK_TRY(ntt_dispatch(pk.curve, pk.d_tab_n, sl, pk.log_n, false, nullptr, 5, n, st));
    }
    {   // top coefficients of the fixed polynomials that reach the top of the quotient's numerator (plonk_quotient_top_kernel)
        std::vector<uint32_t> top((size_t)(pk.W + 5) * 8 * 8, 0u);
        auto fill = [&](int row, const uint32_t* coeffs) {
            for (int t = 0; t < 8; t++) {
                const long long idx = (long long)n - 8 + t;
                if (idx >= 0 && (uint64_t)idx < sl) std::memcpy(&top[((size_t)row * 8 + t) * 8], coeffs + (size_t)idx * 8, 32);
            }
        };
        for (int j = 0; j < pk.W; j++) fill(j, sig_coeffs + (size_t)j * poly_len * 8);
        for (int j = 0; j < 4; j++) fill(pk.W + j, sel_coeffs + (size_t)(6 + j) * poly_len * 8);
        fill(pk.W + 4, sel_coeffs + (size_t)12 * poly_len * 8);
        HIP_TRY(hipMalloc((void**)&pk.d_top_fixed, top.size() * 4));
        HIP_TRY(hipMemcpyAsync(pk.d_top_fixed, top.data(), top.size() * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));                                // `top` leaves scope
    }
    HIP_TRY(hipStreamSynchronize(st));
    return MZK_OK;
}

// the W + 3 coefficients of the quotient from X^(Wn) on (plonk.cuh, plonk_quotient_top_kernel)
template <class P>
int32_t quotient_top_run(const PlonkPk& pk, const uint32_t* d_polys, uint64_t in_stride, uint64_t in_len, const uint32_t* alpha, const uint32_t* beta,
                         const uint32_t* gamma, uint32_t* d_top, hipStream_t st) {
    using F = Fp<P>;
    const uint64_t n = 1ull << pk.log_n;
    TopArgs a;
    std::memset(&a, 0, sizeof a);
    a.polys = d_polys; a.stride = in_stride; a.in_len = in_len; a.n = n; a.top_fixed = pk.d_top_fixed; a.W = pk.W; a.K = pk.W + 3;
    const long long shift = ((long long)(pk.W + 1) * (long long)n + pk.W + 2) - (6ll * (long long)n + 4);      // D - (6n + 4)
    a.gate_shift = shift < a.K ? (int)shift : a.K;
    std::memcpy(a.alpha, alpha, 32); std::memcpy(a.beta, beta, 32); std::memcpy(a.gamma, gamma, 32);
    F b;
    std::memcpy(b.l, beta, 32);
    for (int j = 0; j < pk.W; j++) {
        F kj;
        std::memcpy(kj.l, pk.k[j], 32);
        const F v = b * kj;
        std::memcpy(a.bk[j], v.l, 32);
    }
    F wn = F::from_const(P::ROOT);
    for (int i = pk.log_n; i < P::TWO_ADICITY; i++) wn = sqr(wn);
    F wi;
    std::memcpy(wi.l, pk.w_inv, 32);
    F cur = wn * wn;                                                       // w^2, then times w^-1 per step
    for (int rho = 0; rho < a.K; rho++) { std::memcpy(a.wpow[rho], cur.l, 32); cur = cur * wi; }
    a.out = d_top;
    hipLaunchKernelGGL((plonk_quotient_top_kernel<P>), dim3(1), dim3(PLK_TOP_THREADS), 0, st, a);
    HIP_TRY(hipGetLastError());
    return MZK_OK;
}

// per resident class: fold the online polynomials mod (X^n - h_k^n), size-n coset NTTs, the fused kernel on n points,
// size-n inverse coset NTT.  d_out[lc] = t mod (X^n - h_k^n), n coefficients per class.
template <class P>
int32_t quotient_chunked_run(const PlonkPk& pk, const uint32_t* d_polys, uint64_t in_stride, uint64_t in_len, uint32_t flags, const uint32_t* tau,
                             const uint32_t* alpha, const uint32_t* beta, const uint32_t* gamma, uint32_t* d_out, hipStream_t st) {
    using F = Fp<P>;
    const uint64_t n = 1ull << pk.log_n;
    const int rows = pk.W + 2 + (pk.ultra ? 3 : 0);
    const size_t ncl = pk.cls.size();
    ProfScope total("plonk_quotient_chunked_total", st);
    MZK_TRY(ws_acquire(st));
    MZK_TRY(g_ws.plonk_polys.reserve((size_t)rows * n * 32 + 64 + (size_t)rows * 4 * 32));
    uint32_t* work = g_ws.plonk_polys.as<uint32_t>();
    uint32_t* patch = work + ((size_t)rows * n + 2) * 8;
    const bool pi_zero = (flags & MZK_QUOTIENT_PI_ZERO) != 0;              // the caller knows its public-input polynomial is zero: row W + 1 is neither
                                                                           // transformed nor read
    const bool patched = in_len <= n + 4 && n >= 4;                        // p mod (X^n - c) differs from p's first n coefficients in <= 4 places
    QuotientArgs a;
    a.n_cls = 0;
    a.m = n; a.fstride = ncl * n; a.ostride = n; a.next_off = 1;
    fill_quotient_constants<P>(a, pk, tau, alpha, beta, gamma);
    const unsigned long long fold_threads = n * (unsigned long long)rows;
    // Small circuits: all classes in ONE launch per step (round 5).  A class's transforms are 192 workgroups at 2^15 gates and the round
    // was a chain of 5 (6) x 5 launches, 0.64 ms of a 3.7-ms proof; with the class as a grid dimension (ntt_fx.cuh
    // nttx_pass_classes_kernel, quotient_class_shift) it is 2 + 2 + 1 + 2 launches that fill the chip.  Needs the in-place patched form,
    // a work buffer per class (rows x n x 32 B each: up to 2^18 gates, 280 MB) and multi-pass transforms.  MZK_QUOTIENT_NO_CLASS_BATCH=1: A/B.
    static const bool no_class_batch = std::getenv("MZK_QUOTIENT_NO_CLASS_BATCH") != nullptr;
    static const int class_batch_max_log = std::getenv("MZK_QUOTIENT_CLASS_BATCH_MAX_LOG") ? std::atoi(std::getenv("MZK_QUOTIENT_CLASS_BATCH_MAX_LOG")) : 18;      // (tuning switch; measured: 2^18 gates 1.97 -> 1.60 ms, 2^19 3.47 -> 3.28, 2^20 no gain for 1.1 GB more workspace)
    if (patched && ncl > 1 && ncl <= (size_t)PLK_RATIO && pk.log_n >= 10 && pk.log_n <= class_batch_max_log && !no_class_batch) {
        MZK_TRY(g_ws.plonk_polys.reserve(ncl * (size_t)rows * n * 32 + 64 + ncl * (size_t)rows * 4 * 32));
        work = g_ws.plonk_polys.as<uint32_t>();
        patch = work + (ncl * (size_t)rows * n + 2) * 8;
        FoldClasses fc;
        const uint32_t* cosets[PLK_RATIO];                          // (= NTTX_MAX_CLASSES of ntt_fx.cuh)
        a.n_cls = (int)ncl;
        for (size_t lc = 0; lc < ncl; lc++) {
            std::memcpy(fc.c[lc].l, pk.c_cls[pk.cls[lc]], 32);
            cosets[lc] = pk.h_cls[pk.cls[lc]];
            a.zh_cls[lc] = pk.cls[lc];
        }
        hipLaunchKernelGGL((plonk_fold_patch_classes_kernel<P>), dim3((rows * 4 + 63) / 64, (unsigned)ncl), dim3(64), 0, st, d_polys, in_stride, in_len, n, rows, fc, patch);
        HIP_TRY(hipGetLastError());
        // every row of every class; the workgroups of a zero public-input row (row W + 1) exit at once
        MZK_TRY(ntt_classes_dispatch(pk.curve, work, n, pk.log_n, false, cosets, (int)ncl, (uint32_t)rows, n, st, 1, d_polys, in_stride, patch, pi_zero ? pk.W + 1 : -1));
        a.sel = pk.d_fixed;
        a.sig = pk.d_fixed + (size_t)pk.nsel * ncl * n * 8;
        a.tab = pk.ultra ? pk.d_fixed + (size_t)(pk.nsel + pk.W) * ncl * n * 8 : nullptr;
        a.wire = work;
        a.z = work + (size_t)pk.W * n * 8;
        a.pi = pi_zero ? nullptr : work + (size_t)(pk.W + 1) * n * 8;
        a.h = pk.ultra ? work + (size_t)(pk.W + 2) * n * 8 : nullptr;
        a.pl = pk.ultra ? work + (size_t)(pk.W + 4) * n * 8 : nullptr;
        a.xs = pk.d_xs;
        a.inv_den = pk.d_inv_den;
        a.inv_den_n = pk.ultra ? pk.d_inv_den_n : nullptr;
        a.out = d_out;
        a.cls_fixed = n * 8;
        a.cls_online = (unsigned long long)rows * n * 8;
        a.zh_class = a.zh_cls[0];
        const dim3 grid((unsigned)((n + PLK_THREADS - 1) / PLK_THREADS), (unsigned)ncl);
        {
            ProfScope ps("plonk_quotient_kernel", st);
            if (pk.ultra) {
                hipLaunchKernelGGL((plonk_quotient_kernel<typename FxOf<P>::type, true>), grid, dim3(PLK_THREADS), 0, st, a);
                hipLaunchKernelGGL((plonk_quotient_lookup_kernel<typename FxOf<P>::type>), grid, dim3(PLK_THREADS), 0, st, a);
            } else {
                hipLaunchKernelGGL((plonk_quotient_kernel<typename FxOf<P>::type, false>), grid, dim3(PLK_THREADS), 0, st, a);
            }
        }
        HIP_TRY(hipGetLastError());
        MZK_TRY(ntt_classes_dispatch(pk.curve, d_out, n, pk.log_n, true, cosets, (int)ncl, 1, n, st, 2, nullptr, 0, nullptr, -1));     // back to the boundary form
        MZK_TRY(ws_release(st));
        return MZK_OK;
    }
    for (size_t lc = 0; lc < ncl; lc++) {
        const int k = pk.cls[lc];
        FrArg c_k;
        std::memcpy(c_k.l, pk.c_cls[k], 32);
        // evaluations on the class, in the internal form: size-n coset NTTs of p mod (X^n - c_k)
        auto transform = [&](int first, int count, int skip) -> int32_t {   // rows first .. first + count - 1, row `skip` left out
            if (patched)                                                    // read in place (not overwritten), elements 0..3 from the patch
                return ntt_dispatch(pk.curve, work + (size_t)first * n * 8, n, pk.log_n, false, pk.h_cls[k], (uint32_t)count, n, st, 1,
                                    d_polys + (size_t)first * in_stride * 8, in_stride, patch + (size_t)first * 4 * 8, skip);
            return ntt_dispatch(pk.curve, work + (size_t)first * n * 8, n, pk.log_n, false, pk.h_cls[k], (uint32_t)count, n, st, 1, nullptr, 0, nullptr, skip);
        };
        if (patched) hipLaunchKernelGGL((plonk_fold_patch_kernel<P>), dim3((rows * 4 + 63) / 64), dim3(64), 0, st, d_polys, in_stride, in_len, n, rows, c_k, patch);
        else hipLaunchKernelGGL((plonk_fold_kernel<P>), dim3((unsigned)((fold_threads + PLK_THREADS - 1) / PLK_THREADS)), dim3(PLK_THREADS), 0, st,
                                d_polys, in_stride, in_len, n, rows, c_k, work);
        HIP_TRY(hipGetLastError());
        if (!pi_zero) MZK_TRY(transform(0, rows, -1));
        else if (pk.ultra) MZK_TRY(transform(0, rows, pk.W + 1));          // one batch; the workgroups of the public-input row exit at once
        else MZK_TRY(transform(0, pk.W + 1, -1));
        a.sel = pk.d_fixed + lc * n * 8;
        a.sig = pk.d_fixed + ((size_t)pk.nsel * ncl + lc) * n * 8;
        a.tab = pk.ultra ? pk.d_fixed + ((size_t)(pk.nsel + pk.W) * ncl + lc) * n * 8 : nullptr;
        a.wire = work;
        a.z = work + (size_t)pk.W * n * 8;
        a.pi = pi_zero ? nullptr : work + (size_t)(pk.W + 1) * n * 8;
        a.h = pk.ultra ? work + (size_t)(pk.W + 2) * n * 8 : nullptr;
        a.pl = pk.ultra ? work + (size_t)(pk.W + 4) * n * 8 : nullptr;
        a.xs = pk.d_xs + lc * n * 8;
        a.inv_den = pk.d_inv_den + lc * n * 8;
        a.inv_den_n = pk.ultra ? pk.d_inv_den_n + lc * n * 8 : nullptr;
        a.out = d_out + lc * n * 8;
        a.zh_class = k;
        const dim3 grid((unsigned)((n + PLK_THREADS - 1) / PLK_THREADS));
        {
            ProfScope ps("plonk_quotient_kernel", st);
            if (pk.ultra) {
                hipLaunchKernelGGL((plonk_quotient_kernel<typename FxOf<P>::type, true>), grid, dim3(PLK_THREADS), 0, st, a);
                hipLaunchKernelGGL((plonk_quotient_lookup_kernel<typename FxOf<P>::type>), grid, dim3(PLK_THREADS), 0, st, a);
            } else {
                hipLaunchKernelGGL((plonk_quotient_kernel<typename FxOf<P>::type, false>), grid, dim3(PLK_THREADS), 0, st, a);
            }
        }
        HIP_TRY(hipGetLastError());
        MZK_TRY(ntt_dispatch(pk.curve, d_out + lc * n * 8, n, pk.log_n, true, pk.h_cls[k], 1, n, st, 2));  // back to the boundary form
    }
    MZK_TRY(ws_release(st));
    return MZK_OK;
}

// the class remainders (class-major in the order of `classes`, all resident here after the exchange) -> the 8n quotient
// coefficients (slabs above the number of classes are zero: the classes given must determine t, i.e. deg t < ncl * n)
template <class P>
int32_t quotient_combine_run(int log_n, const uint32_t* classes, int ncl, const uint32_t* d_r, const uint32_t* d_top, int n_top, uint32_t* d_out, hipStream_t st) {
    using F = Fp<P>;
    const uint64_t n = 1ull << log_n;
    F w8 = F::from_const(P::ROOT);
    for (int i = 3; i < P::TWO_ADICITY; i++) w8 = sqr(w8);                   // primitive 8th root of unity = w_8n^n
    const F gn = pow_u64(F::from_const(P::GENERATOR), n);
    F c[8];
    for (int k = 0; k < ncl; k++) c[k] = gn * pow_u64(w8, (uint64_t)classes[k]);      // c_k = h_k^n
    CombineArgs a;
    std::memset(&a, 0, sizeof a);
    a.r = d_r; a.out = d_out; a.n = n; a.ncl = ncl;
    a.top = d_top; a.n_top = d_top ? n_top : 0;
    for (int k = 0; k < ncl; k++) {
        const F e = pow_u64(c[k], (uint64_t)ncl);
        std::memcpy(a.ctop[k], e.l, 32);
    }
    // inverse Vandermonde by Lagrange: column k of V^-1 holds the coefficients of L_k(X) = prod_{m != k} (X - c_m) / (c_k - c_m)
    for (int k = 0; k < ncl; k++) {
        F poly[9];
        poly[0] = F::one();
        int deg = 0;
        F den = F::one();
        for (int m = 0; m < ncl; m++) {
            if (m == k) continue;
            poly[deg + 1] = F::zero();
            for (int d = deg + 1; d >= 1; d--) poly[d] = poly[d - 1] - c[m] * poly[d];        // times (X - c_m)
            poly[0] = F::zero() - c[m] * poly[0];
            deg++;
            den = den * (c[k] - c[m]);
        }
        const F dinv = inv(den);
        for (int q = 0; q < ncl; q++) {
            const F e = poly[q] * dinv;
            std::memcpy(a.mat[q][k], e.l, 32);
        }
    }
    if (ncl < PLK_RATIO) HIP_TRY(hipMemsetAsync(d_out + (size_t)ncl * n * 8, 0, (size_t)(PLK_RATIO - ncl) * n * 32, st));
    hipLaunchKernelGGL((plonk_combine_kernel<P>), dim3((unsigned)((n + PLK_THREADS - 1) / PLK_THREADS)), dim3(PLK_THREADS), 0, st, a);
    HIP_TRY(hipGetLastError());
    return MZK_OK;
}

const PlonkPk* find_pk(uint64_t handle) {
    auto it = g_pks.find(handle);
    if (it == g_pks.end()) { set_error("unknown proving-key handle"); return nullptr; }
    return it->second.get();
}

}  // namespace

int32_t plonk_pk_register(int curve, int log_n, int W, const uint32_t* sel, const uint32_t* sig, const uint32_t* tab, uint64_t poly_len,
                          const uint32_t* k_mont, const uint32_t* classes, uint32_t n_classes, uint64_t* out_handle) {
    const bool ultra = tab != nullptr;
    if (classes) {
        bool ok = n_classes >= 1 && n_classes <= PLK_RATIO && poly_len <= (1ull << log_n);
        for (uint32_t i = 0; ok && i < n_classes; i++) ok = classes[i] < PLK_RATIO && (i == 0 || classes[i] > classes[i - 1]);
        if (!ok) { set_error("chunked proving key: 1..8 strictly increasing residue classes < 8, fixed polynomials of degree < n"); return MZK_ERR_INVALID_ARG; }
    }
    if ((curve != 0 && curve != 1) || W != (ultra ? PLK_MAX_WIRES : PLK_WIRES) || log_n < 1 || log_n + 3 > (curve == 0 ? 32 : 28) || log_n + 3 > 30 ||
        poly_len == 0 || poly_len > (8ull << log_n) || !sel || !sig || !k_mont || !out_handle) {
        set_error("bad argument (TurboPlonk: 5 wire types, 13 selectors; UltraPlonk: 6 wire types, 14 selectors, 4 table polynomials)");
        return MZK_ERR_INVALID_ARG;
    }
    auto pk = std::make_unique<PlonkPk>();
    pk->curve = curve; pk->log_n = log_n; pk->W = W; pk->ultra = ultra; pk->nsel = PLK_SELECTORS + (ultra ? 1 : 0);
    std::memset(pk->k, 0, sizeof pk->k);
    std::memcpy(pk->k, k_mont, (size_t)W * 32);
    {   // selectors that are the zero polynomial: their gate terms are skipped (plonk.cuh)
        const uint64_t words = poly_len * 8;
        for (int j = 0; j < PLK_SELECTORS; j++) {
            const uint32_t* c = sel + (size_t)j * words;
            bool zero = true;
            for (uint64_t i = 0; i < words && zero; i++) zero = c[i] == 0;
            if (zero) pk->sel_zero |= 1u << j;
        }
    }
    int32_t rc;
    if (classes) {
        pk->cls.assign(classes, classes + n_classes);
        rc = curve == 0 ? pk_build_chunked<BlsFr>(*pk, sel, sig, tab, poly_len) : pk_build_chunked<BnFr>(*pk, sel, sig, tab, poly_len);
    } else {
        rc = curve == 0 ? pk_build<BlsFr>(*pk, sel, sig, tab, poly_len) : pk_build<BnFr>(*pk, sel, sig, tab, poly_len);
    }
    if (rc != MZK_OK) {
        for (auto* d : pk->bufs()) if (d) (void)hipFree(d);
        return rc;
    }
    *out_handle = handle_make(cur().logical, g_next_pk++);
    g_pks[*out_handle] = std::move(pk);
    return MZK_OK;
}

int32_t plonk_pk_release(uint64_t handle) {
    auto it = g_pks.find(handle);
    if (it == g_pks.end()) { set_error("unknown proving-key handle"); return MZK_ERR_BAD_HANDLE; }
    HIP_TRY(hipDeviceSynchronize());
    for (auto* d : it->second->bufs()) if (d) (void)hipFree(d);
    g_pks.erase(it);
    return MZK_OK;
}
// HBM held by a proving key: the resident evaluations of the fixed polynomials and the per-point tables
uint64_t plonk_pk_bytes(uint64_t handle) {
    const PlonkPk* pk = find_pk(handle);
    if (!pk) return 0;
    const uint64_t n = 1ull << pk->log_n, pts = pk->cls.empty() ? (uint64_t)PLK_RATIO * n : (uint64_t)pk->cls.size() * n;
    const uint64_t nfix = (uint64_t)pk->nsel + pk->W + (pk->ultra ? 4 : 0);
    uint64_t b = (nfix + 2 + (pk->ultra ? 1 : 0)) * pts * 32;              // d_fixed, d_xs, d_inv_den (, d_inv_den_n)
    b += ((uint64_t)pk->W + 1 + (pk->ultra ? 5 : 0)) * n * 32;             // d_sigma_n, d_omega_n (, d_tab_n)
    if (pk->d_top_fixed) b += ((uint64_t)pk->W + 5) * 8 * 32;
    return b;
}
void plonk_release_all() {
    for (auto& kv : g_pks)
        for (auto* d : kv.second->bufs()) if (d) (void)hipFree(d);
    g_pks.clear();
}

int32_t plonk_quotient_dev(uint64_t handle, uint32_t* d_polys, uint64_t in_len, const uint32_t* tau, const uint32_t* alpha, const uint32_t* beta,
                           const uint32_t* gamma, uint32_t* d_out, hipStream_t st) {
    const PlonkPk* pk = find_pk(handle);
    if (!pk) return MZK_ERR_BAD_HANDLE;
    if (!d_polys || !d_out || !alpha || !beta || !gamma || (pk->ultra && !tau) || in_len > (8ull << pk->log_n)) { set_error("bad argument"); return MZK_ERR_INVALID_ARG; }
    if (!pk->cls.empty()) { set_error("chunked proving key: use mzk_plonk_quotient_chunked_dev"); return MZK_ERR_INVALID_ARG; }
    return pk->curve == 0 ? quotient_run<BlsFr>(*pk, d_polys, in_len, tau, alpha, beta, gamma, d_out, st)
                          : quotient_run<BnFr>(*pk, d_polys, in_len, tau, alpha, beta, gamma, d_out, st);
}
int32_t plonk_perm_product_dev(uint64_t handle, const uint32_t* d_wires, const uint32_t* beta, const uint32_t* gamma, uint32_t* d_out, hipStream_t st) {
    const PlonkPk* pk = find_pk(handle);
    if (!pk) return MZK_ERR_BAD_HANDLE;
    if (!d_wires || !d_out || !beta || !gamma) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    return pk->curve == 0 ? perm_product_run<BlsFr>(*pk, d_wires, beta, gamma, d_out, st) : perm_product_run<BnFr>(*pk, d_wires, beta, gamma, d_out, st);
}
int32_t plookup_sorted_vec_dev(uint64_t handle, const uint32_t* d_wires, const uint32_t* tau, uint32_t* d_table, uint32_t* d_lookup, uint32_t* d_sorted,
                               hipStream_t st) {
    const PlonkPk* pk = find_pk(handle);
    if (!pk) return MZK_ERR_BAD_HANDLE;
    if (!pk->ultra || !d_wires || !tau || !d_table || !d_lookup || !d_sorted || pk->log_n < 2) {
        set_error("bad argument (needs an UltraPlonk proving key)");
        return MZK_ERR_INVALID_ARG;
    }
    return pk->curve == 0 ? sorted_vec_run<BlsFr>(*pk, d_wires, tau, d_table, d_lookup, d_sorted, st)
                          : sorted_vec_run<BnFr>(*pk, d_wires, tau, d_table, d_lookup, d_sorted, st);
}
int32_t plookup_product_dev(uint64_t handle, const uint32_t* d_table, const uint32_t* d_lookup, const uint32_t* d_sorted, const uint32_t* beta,
                            const uint32_t* gamma, uint32_t* d_out, hipStream_t st) {
    const PlonkPk* pk = find_pk(handle);
    if (!pk) return MZK_ERR_BAD_HANDLE;
    if (!pk->ultra || !d_table || !d_lookup || !d_sorted || !beta || !gamma || !d_out || pk->log_n < 2) {
        set_error("bad argument (needs an UltraPlonk proving key)");
        return MZK_ERR_INVALID_ARG;
    }
    return pk->curve == 0 ? lookup_product_run<BlsFr>(*pk, d_table, d_lookup, d_sorted, beta, gamma, d_out, st)
                          : lookup_product_run<BnFr>(*pk, d_table, d_lookup, d_sorted, beta, gamma, d_out, st);
}
int32_t plonk_quotient_chunked_dev(uint64_t handle, const uint32_t* d_polys, uint64_t in_stride, uint64_t in_len, uint32_t flags, const uint32_t* tau,
                                   const uint32_t* alpha, const uint32_t* beta, const uint32_t* gamma, uint32_t* d_out, hipStream_t st) {
    const PlonkPk* pk = find_pk(handle);
    if (!pk) return MZK_ERR_BAD_HANDLE;
    if (pk->cls.empty()) { set_error("not a chunked proving key"); return MZK_ERR_INVALID_ARG; }
    if (!d_polys || !d_out || !alpha || !beta || !gamma || (pk->ultra && !tau) || in_len > (2ull << pk->log_n) || in_len > in_stride) {
        set_error("bad argument (online polynomials have degree < 2n)");
        return MZK_ERR_INVALID_ARG;
    }
    return pk->curve == 0 ? quotient_chunked_run<BlsFr>(*pk, d_polys, in_stride, in_len, flags, tau, alpha, beta, gamma, d_out, st)
                          : quotient_chunked_run<BnFr>(*pk, d_polys, in_stride, in_len, flags, tau, alpha, beta, gamma, d_out, st);
}
int32_t plonk_quotient_top_dev(uint64_t handle, const uint32_t* d_polys, uint64_t in_stride, uint64_t in_len, const uint32_t* alpha, const uint32_t* beta,
                               const uint32_t* gamma, uint32_t* d_top, uint32_t* out_n_top, hipStream_t st) {
    const PlonkPk* pk = find_pk(handle);
    if (!pk) return MZK_ERR_BAD_HANDLE;
    const uint64_t n = 1ull << pk->log_n;
    if (!pk->d_top_fixed) { set_error("not a chunked proving key"); return MZK_ERR_INVALID_ARG; }
    if (!d_polys || !d_top || !alpha || !beta || !gamma || in_len > in_stride || in_len < n + 3) {
        set_error("bad argument (rows of n + 3 coefficient slots: W wire polynomials, then the permutation product)");
        return MZK_ERR_INVALID_ARG;
    }
    if (n <= (uint64_t)pk->W + 2 || n < 8) { set_error("domain too small: the top W + 3 coefficients of the quotient are those of its numerator only for n > W + 2"); return MZK_ERR_UNSUPPORTED; }
    if (out_n_top) *out_n_top = (uint32_t)pk->W + 3;
    return pk->curve == 0 ? quotient_top_run<BlsFr>(*pk, d_polys, in_stride, in_len, alpha, beta, gamma, d_top, st)
                          : quotient_top_run<BnFr>(*pk, d_polys, in_stride, in_len, alpha, beta, gamma, d_top, st);
}
int32_t plonk_quotient_combine_dev(int curve, int log_n, const uint32_t* classes, uint32_t n_classes, const uint32_t* d_r, const uint32_t* d_top, uint32_t n_top,
                                   uint32_t* d_out, hipStream_t st) {
    if (curve != 0 && curve != 1) { set_error("unknown curve_id"); return MZK_ERR_INVALID_ARG; }
    if (d_top && (n_top == 0 || n_top > PLK_TOP_MAX || n_top > (1ull << log_n) || (classes ? n_classes : 8u) >= PLK_RATIO)) {
        set_error("combine: 1..9 top coefficients above at most 7 classes");
        return MZK_ERR_INVALID_ARG;
    }
    static const uint32_t all8[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    if (!classes) { classes = all8; n_classes = 8; }
    bool ok = n_classes >= 1 && n_classes <= PLK_RATIO;
    for (uint32_t i = 0; ok && i < n_classes; i++) ok = classes[i] < PLK_RATIO && (i == 0 || classes[i] > classes[i - 1]);
    if (!ok) { set_error("combine: 1..8 strictly increasing residue classes < 8"); return MZK_ERR_INVALID_ARG; }
    return curve == 0 ? quotient_combine_run<BlsFr>(log_n, classes, (int)n_classes, d_r, d_top, (int)n_top, d_out, st)
                      : quotient_combine_run<BnFr>(log_n, classes, (int)n_classes, d_r, d_top, (int)n_top, d_out, st);
}
// number of resident residue classes of a chunked key (0: whole-domain key, -1: unknown handle); `out` receives them
int plonk_pk_classes(uint64_t handle, uint32_t* out /* 8 slots, nullable */) {
    auto it = g_pks.find(handle);
    if (it == g_pks.end()) return -1;
    if (out) for (size_t i = 0; i < it->second->cls.size(); i++) out[i] = (uint32_t)it->second->cls[i];
    return (int)it->second->cls.size();
}
int plonk_pk_curve(uint64_t handle) {
    auto it = g_pks.find(handle);
    return it == g_pks.end() ? -1 : it->second->curve;
}
int plonk_pk_log_n(uint64_t handle) {
    auto it = g_pks.find(handle);
    return it == g_pks.end() ? -1 : it->second->log_n;
}
int plonk_pk_wires(uint64_t handle) {
    auto it = g_pks.find(handle);
    return it == g_pks.end() ? -1 : it->second->W;
}
int plonk_pk_is_ultra(uint64_t handle) {
    auto it = g_pks.find(handle);
    return it == g_pks.end() ? -1 : (it->second->ultra ? 1 : 0);
}

}  // namespace mzk
