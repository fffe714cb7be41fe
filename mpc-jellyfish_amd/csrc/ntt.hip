// ntt.hip -- host side of the NTT: plan cache and pass launches.  The passes run on the reduced-radix
// kernel of ntt_fx.cuh; ntt.cuh keeps the 32-bit-limb kernel it was derived from (same decomposition).
#include <cstdlib>
#include <map>
#include <memory>

#include "internal.hpp"
#include "ntt_fx.cuh"

namespace mzk {
namespace {

// ---- NTT plans -----------------------------------------------------------------------------------
struct NttPlanDev {
    NttxPlanHost h;
    uint32_t* d_stage[NTT_MAX_PASSES] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t *d_flo = nullptr, *d_fhi = nullptr, *d_fone = nullptr;
    uint64_t last_used = 0;
    uint32_t* d_tfull[NTT_MAX_PASSES] = {nullptr, nullptr, nullptr, nullptr};
};
struct PlanKey {
    int curve, log_n, inverse, scale;
    uint32_t coset[8];
    bool has_coset;
    bool operator<(const PlanKey& o) const {
        if (curve != o.curve) return curve < o.curve;
        if (log_n != o.log_n) return log_n < o.log_n;
        if (inverse != o.inverse) return inverse < o.inverse;
        if (scale != o.scale) return scale < o.scale;
        if (has_coset != o.has_coset) return has_coset < o.has_coset;
        return has_coset && std::memcmp(coset, o.coset, sizeof coset) < 0;
    }
};
std::map<PlanKey, std::unique_ptr<NttPlanDev>> g_plans_of[MAX_CTX];      // one plan cache per device context
#define g_plans (g_plans_of[cur().logical])
std::atomic<uint64_t> g_plan_clock{0};           // LRU clock shared by the device threads of one process
// A plan holds device tables (up to tens of MB for the largest domains) and is keyed by the coset offset: a caller sweeping
// offsets (per-proof random cosets, the multiprover's public-polynomial FFTs) must not grow the cache without bound.
constexpr size_t NTT_MAX_PLANS = 48;

void free_plan(NttPlanDev* p) {
    for (auto* d : p->d_stage) if (d) (void)hipFree(d);
    for (auto* d : {p->d_flo, p->d_fhi, p->d_fone}) if (d) (void)hipFree(d);
    for (auto* d : p->d_tfull) if (d) (void)hipFree(d);
}

int32_t upload_words(uint32_t** d, const std::vector<uint32_t>& v) {
    if (v.empty()) { *d = nullptr; return MZK_OK; }
    HIP_TRY(hipMalloc((void**)d, v.size() * 4));
    HIP_TRY(hipMemcpy(*d, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    return MZK_OK;
}

template <class X>
int32_t get_plan(int curve, int log_n, bool inverse, const uint32_t* coset, int scale, NttPlanDev** out) {
    PlanKey key;
    std::memset(&key, 0, sizeof key);
    key.curve = curve; key.log_n = log_n; key.inverse = inverse ? 1 : 0; key.scale = scale;
    key.has_coset = coset != nullptr;
    if (coset) std::memcpy(key.coset, coset, 32);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) { it->second->last_used = ++g_plan_clock; *out = it->second.get(); return MZK_OK; }
    if (g_plans.size() >= NTT_MAX_PLANS) {                       // evict the least recently used plan (its kernels may still be in flight)
        auto victim = g_plans.begin();
        for (auto p = g_plans.begin(); p != g_plans.end(); ++p)
            if (p->second->last_used < victim->second->last_used) victim = p;
        HIP_TRY(hipDeviceSynchronize());
        free_plan(victim->second.get());
        g_plans.erase(victim);
    }
    auto pl = std::make_unique<NttPlanDev>();
    nttx_build_plan<X>(pl->h, log_n, inverse, coset, scale);
    if (!nttx_growth_ok<X>(pl->h)) { set_error("NTT plan violates the lazy-value bound"); return MZK_ERR_UNSUPPORTED; }
    for (int k = 0; k < pl->h.n_pass; k++) MZK_TRY(upload_words(&pl->d_stage[k], pl->h.stage_tw[k]));
    for (int k = 0; k < pl->h.n_pass; k++) MZK_TRY(upload_words(&pl->d_tfull[k], pl->h.t_full[k]));
    MZK_TRY(upload_words(&pl->d_flo, pl->h.f_lo));
    MZK_TRY(upload_words(&pl->d_fhi, pl->h.f_hi));
    MZK_TRY(upload_words(&pl->d_fone, pl->h.f_one));
    for (int k = 0; k < NTT_MAX_PASSES; k++) { pl->h.stage_tw[k] = {}; pl->h.t_full[k] = {}; }     // the host copies are not needed again
    pl->h.f_lo = {}; pl->h.f_hi = {};
    pl->last_used = ++g_plan_clock;
    *out = pl.get();
    g_plans[key] = std::move(pl);
    return MZK_OK;
}

// how many workgroups of the persistent pass one launch keeps resident (per device context; the occupancy query costs ~50 us)
template <class X, bool TW_LDS>
int32_t persistent_grid(size_t lds, unsigned* out) {
    static unsigned cached[MAX_CTX][4] = {};                             // by LDS size class
    const int cls = lds >= 48 * 1024 ? 0 : (lds >= 36 * 1024 ? 1 : (lds >= 18 * 1024 ? 2 : 3));
    unsigned& slot = cached[cur().logical][cls];
    if (!slot) {
        int per_cu = 0, dev = 0, cus = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)nttx_pass_persistent_kernel<X, TW_LDS>, NTTX_THREADS, lds));
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (per_cu < 1) per_cu = 1;
        slot = (unsigned)(per_cu * cus);
    }
    *out = slot;
    return MZK_OK;
}

template <class X>
int32_t launch_pass(const NttxPassArgs& a, unsigned long long n_tiles, uint32_t batch, hipStream_t st, bool r4) {
    const size_t tile = (size_t)1 << (a.log_r + a.log_c);
    const size_t lds = 2 * tile * 16 + tile * 4;
    static bool attr_set[MAX_CTX] = {};                                  // per device (function attributes live in the device's code object)
    if (!attr_set[cur().logical]) {
        HIP_TRY(hipFuncSetAttribute((const void*)nttx_pass_kernel<X, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        HIP_TRY(hipFuncSetAttribute((const void*)nttx_pass_kernel<X, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        HIP_TRY(hipFuncSetAttribute((const void*)nttx_pass_persistent_kernel<X, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        HIP_TRY(hipFuncSetAttribute((const void*)nttx_pass_persistent_kernel<X, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        attr_set[cur().logical] = true;
    }
    ProfScope ps("ntt_pass", st);
    // MZK_NTT_PERSISTENT = 1 / 2: the persistent pass with / without LDS-staged twiddles (ntt_fx.cuh) -- built and MEASURED in round 3
    // (profiles/r03_ntt_experiments.txt): 0.647 / 0.679 ms per 2^22 transform against 0.603 ms for one tile per workgroup, which
    // therefore stays the default.  Its 113 VGPRs (prefetched elements live across the stages) leave 2 workgroups per CU instead of 4.
    static const int mode = std::getenv("MZK_NTT_PERSISTENT") ? std::atoi(std::getenv("MZK_NTT_PERSISTENT")) : 0;
    const unsigned long long total = n_tiles * batch;
    unsigned resident = 0;
    const bool tw_lds = !a.is_final && mode != 2;                         // (mode 2: persistent without LDS-staged twiddles)
    const size_t lds_p = lds + (tw_lds ? (((size_t)1 << a.log_r) - 1) * FS_TW_WORDS * 4 : 0);
    if (mode) MZK_TRY((tw_lds ? persistent_grid<X, true>(lds_p, &resident) : persistent_grid<X, false>(lds_p, &resident)));
    if (mode && total >= 3ull * resident && tile <= 2 * NTTX_THREADS && !a.patch && a.skip_batch < 0) {
        int log_tiles = 0;
        while ((1ull << log_tiles) < n_tiles) log_tiles++;
        if (tw_lds) hipLaunchKernelGGL((nttx_pass_persistent_kernel<X, true>), dim3(resident), dim3(NTTX_THREADS), lds_p, st, a, log_tiles, total);
        else hipLaunchKernelGGL((nttx_pass_persistent_kernel<X, false>), dim3(resident), dim3(NTTX_THREADS), lds_p, st, a, log_tiles, total);
    } else if (r4) {
        hipLaunchKernelGGL((nttx_pass_kernel<X, true>), dim3((unsigned)n_tiles, batch), dim3(NTTX_THREADS), lds, st, a);
    } else {
        hipLaunchKernelGGL((nttx_pass_kernel<X, false>), dim3((unsigned)n_tiles, batch), dim3(NTTX_THREADS), lds, st, a);
    }
    HIP_TRY(hipGetLastError());
    return MZK_OK;
}

// d_data: batch polynomials, `stride` elements apart, transformed in place (async on st)
// d_src (nullable): the first pass reads the batch from there (src_stride elements apart, not overwritten) instead of d_data, which
// then only receives the result; d_patch (nullable, needs d_src): 4 replacement elements per batch entry for the input indices 0..3
template <class X>
int32_t ntt_dev(int curve, uint32_t* d_data, uint64_t in_len, int log_n, bool inverse, const uint32_t* coset,
                uint32_t batch, uint64_t stride, hipStream_t st, int scale, const uint32_t* d_src, uint64_t src_stride, const uint32_t* d_patch,
                int skip_batch) {
    if (log_n < 0 || log_n > X::TWO_ADICITY || log_n > 30) { set_error("log_n out of range"); return MZK_ERR_INVALID_ARG; }
    const uint64_t N = 1ull << log_n;
    if (batch == 0) return MZK_OK;
    if (stride < N || batch > 65535 || (d_src && src_stride < in_len) || (d_patch && (!d_src || N < 4))) { set_error("bad batch/stride"); return MZK_ERR_INVALID_ARG; }
    if (in_len > N) in_len = N;
    if (log_n == 0 && scale == 0) {
        // size-1 transform is the identity (offset^0 = 1, N^-1 = 1) -- of the zero-padded input: an EMPTY input gives 0, not what the buffer
        // held (found by tools/soak.py in round 5), and a separate source is copied
        if (in_len == 0) HIP_TRY(hipMemset2DAsync(d_data, stride * 32, 0, 32, batch, st));
        else if (d_src) HIP_TRY(hipMemcpy2DAsync(d_data, stride * 32, d_src, src_stride * 32, 32, batch, hipMemcpyDeviceToDevice, st));
        return MZK_OK;
    }
    if (log_n == 0) { set_error("scaled size-1 transform"); return MZK_ERR_UNSUPPORTED; }
    NttPlanDev* pl;
    MZK_TRY(get_plan<X>(curve, log_n, inverse, coset, scale, &pl));
    MZK_TRY(ws_acquire(st));
    MZK_TRY(g_ws.ntt_scratch.reserve((size_t)batch * N * 36));          // 9-limb planes between passes
    uint32_t* scratch = g_ws.ntt_scratch.as<uint32_t>();
    ProfScope total("ntt_total", st);
    const int K = pl->h.n_pass;
    int log_p = 0;
    // transforms of at least 2^21 points take 2048-element tiles and stage pairs in registers (ntt_fx.cuh, R4): -4 % at 2^22, -7 % at 2^24;
    // smaller ones the 1024-element radix-2 form, which fills the chip with twice as many workgroups -- also in batches (the seven
    // 2^20-point class transforms of the quotient round measured 5.56-5.72 ms without and 5.75-5.78 ms with R4: profiles/r04_ntt_radix4.txt).
    // MZK_NTT_NO_RADIX4=1: the round-3 form everywhere (A/B).
    static const bool no_r4 = std::getenv("MZK_NTT_NO_RADIX4") != nullptr;
    const bool r4 = !no_r4 && N >= (1ull << 21);
    static const int tile_rt = std::getenv("MZK_NTT_TILE_LOG_RT") ? std::atoi(std::getenv("MZK_NTT_TILE_LOG_RT")) : 0;      // (experiment switch, see ntt.cuh)
    const int tile_log = r4 ? (tile_rt >= 11 && tile_rt <= 12 ? tile_rt : 11) : NTT_TILE_LOG;
    for (int k = 0; k < K; k++) {
        NttxPassArgs a;
        std::memset(&a, 0, sizeof a);
        const int lr = pl->h.log_radix[k];
        a.log_n = log_n; a.log_r = lr; a.log_p = log_p; a.log_s = log_n - log_p - lr;
        a.n_pass = K; a.is_first = k == 0; a.is_final = k == K - 1;
        for (int q = 0; q < K; q++) a.log_radix[q] = pl->h.log_radix[q];
        a.log_lb = pl->h.log_lb;
        a.stage_tw = pl->d_stage[k];
        a.t_full = pl->d_tfull[k];
        a.f_lo = pl->d_flo;             // non-null only for an inverse coset transform
        a.f_hi = pl->d_fhi;
        a.f_one = pl->h.final_factor ? pl->d_fone : nullptr;
        a.in_len = in_len;
        a.n = N;
        if (a.is_first && !a.is_final && !d_patch)                      // zero-padded input: leading stages of pass 1 are copies
            while (a.skip < lr && in_len <= (N >> (a.skip + 1))) a.skip++;
        int lc = tile_log - lr;
        if (lc < 0) lc = 0;
        if (a.is_final) lc = K == 1 ? 0 : (lc < pl->h.log_radix[0] ? lc : pl->h.log_radix[0]);
        else lc = lc < a.log_s ? lc : a.log_s;
        a.log_c = lc;
        // pass 1 reads the caller's buffer, middle passes run in place on scratch, the last pass
        // writes back to the caller's buffer (K = 1: data -> scratch, copied back below)
        const bool from_data = k == 0, to_data = (k == K - 1) && K > 1;
        a.in = from_data ? (d_src ? const_cast<uint32_t*>(d_src) : d_data) : scratch;
        a.in_stride = from_data ? (d_src ? src_stride : stride) : N;
        a.patch = from_data ? d_patch : nullptr;
        a.skip_batch = skip_batch;
        a.in_planes = from_data ? 0 : 1;
        a.out = to_data ? d_data : scratch;
        a.out_stride = to_data ? stride : N;
        a.out_planes = (to_data || K == 1) ? 0 : 1;
        const unsigned long long n_tiles = N >> (lr + lc);
        MZK_TRY((launch_pass<X>(a, n_tiles, batch, st, r4)));
        log_p += lr;
    }
    if (K == 1)
        HIP_TRY(hipMemcpy2DAsync(d_data, stride * 32, scratch, N * 32, N * 32, batch, hipMemcpyDeviceToDevice, st));
    MZK_TRY(ws_release(st));
    return MZK_OK;
}

// several cosets, one launch per pass (internal.hpp ntt_classes_dispatch)
template <class X>
int32_t ntt_classes_dev(int curve, uint32_t* d_data, uint64_t in_len, int log_n, bool inverse, const uint32_t* const* cosets, int n_classes, uint32_t rows,
                        uint64_t stride, hipStream_t st, int scale, const uint32_t* d_src, uint64_t src_stride, const uint32_t* d_patch, int skip_batch) {
    const uint64_t N = 1ull << log_n;
    const uint64_t batch = (uint64_t)rows * (uint64_t)n_classes;
    if (log_n < 10 || log_n > X::TWO_ADICITY || log_n > 30 || n_classes < 1 || n_classes > NTTX_MAX_CLASSES || rows == 0 || batch > 65535 || stride < N ||
        (d_src && src_stride < in_len) || (d_patch && !d_src)) {
        set_error("ntt_classes: bad shape");
        return MZK_ERR_INVALID_ARG;
    }
    if (in_len > N) in_len = N;
    NttPlanDev* pl[NTTX_MAX_CLASSES];
    for (int c = 0; c < n_classes; c++) MZK_TRY(get_plan<X>(curve, log_n, inverse, cosets[c], scale, &pl[c]));     // (the most recently used plans: none evicts another)
    const int K = pl[0]->h.n_pass;
    for (int c = 1; c < n_classes; c++)
        if (pl[c]->h.n_pass != K || pl[c]->h.log_lb != pl[0]->h.log_lb || pl[c]->h.final_factor != pl[0]->h.final_factor || (pl[c]->d_flo != nullptr) != (pl[0]->d_flo != nullptr)) {
            set_error("ntt_classes: the classes' plans differ in shape");
            return MZK_ERR_UNSUPPORTED;
        }
    if (K < 2) { set_error("ntt_classes: single-pass transform"); return MZK_ERR_UNSUPPORTED; }
    MZK_TRY(ws_acquire(st));
    MZK_TRY(g_ws.ntt_scratch.reserve((size_t)batch * N * 36));
    uint32_t* scratch = g_ws.ntt_scratch.as<uint32_t>();
    ProfScope total("ntt_total", st);
    int log_p = 0;
    for (int k = 0; k < K; k++) {
        NttxPassArgs a;
        std::memset(&a, 0, sizeof a);
        NttxClasses mc;
        std::memset(&mc, 0, sizeof mc);
        const int lr = pl[0]->h.log_radix[k];
        a.log_n = log_n; a.log_r = lr; a.log_p = log_p; a.log_s = log_n - log_p - lr;
        a.n_pass = K; a.is_first = k == 0; a.is_final = k == K - 1;
        for (int q = 0; q < K; q++) a.log_radix[q] = pl[0]->h.log_radix[q];
        a.log_lb = pl[0]->h.log_lb;
        mc.rows = rows;
        mc.shared_in = d_src ? 1 : 0;
        for (int c = 0; c < n_classes; c++) {
            mc.stage_tw[c] = pl[c]->d_stage[k];
            mc.t_full[c] = pl[c]->d_tfull[k];
            mc.f_lo[c] = pl[c]->d_flo;
            mc.f_hi[c] = pl[c]->d_fhi;
            mc.f_one[c] = pl[c]->h.final_factor ? pl[c]->d_fone : nullptr;
        }
        a.in_len = in_len;
        a.n = N;
        if (a.is_first && !d_patch)
            while (a.skip < lr && in_len <= (N >> (a.skip + 1))) a.skip++;
        int lc = NTT_TILE_LOG - lr;
        if (lc < 0) lc = 0;
        if (a.is_final) lc = lc < pl[0]->h.log_radix[0] ? lc : pl[0]->h.log_radix[0];
        else lc = lc < a.log_s ? lc : a.log_s;
        a.log_c = lc;
        const bool from_data = k == 0, to_data = k == K - 1;
        a.in = from_data ? (d_src ? const_cast<uint32_t*>(d_src) : d_data) : scratch;
        a.in_stride = from_data ? (d_src ? src_stride : stride) : N;
        a.patch = from_data ? d_patch : nullptr;
        a.skip_batch = skip_batch;
        a.in_planes = from_data ? 0 : 1;
        a.out = to_data ? d_data : scratch;
        a.out_stride = to_data ? stride : N;
        a.out_planes = to_data ? 0 : 1;
        const unsigned long long n_tiles = N >> (lr + lc);
        const size_t tile = (size_t)1 << (lr + lc);
        ProfScope ps("ntt_pass", st);
        hipLaunchKernelGGL((nttx_pass_classes_kernel<X>), dim3((unsigned)n_tiles, (unsigned)batch), dim3(NTTX_THREADS), 2 * tile * 16 + tile * 4, st, a, mc);
        HIP_TRY(hipGetLastError());
        log_p += lr;
    }
    MZK_TRY(ws_release(st));
    return MZK_OK;
}

}  // namespace

int32_t ntt_classes_dispatch(int curve, uint32_t* d_data, uint64_t in_len, int log_n, bool inverse, const uint32_t* const* cosets, int n_classes, uint32_t rows,
                             uint64_t stride, hipStream_t st, int scale, const uint32_t* d_src, uint64_t src_stride, const uint32_t* d_patch, int skip_batch) {
    if (curve == MZK_CURVE_BLS12_381) return ntt_classes_dev<BlsFrX>(curve, d_data, in_len, log_n, inverse, cosets, n_classes, rows, stride, st, scale, d_src, src_stride, d_patch, skip_batch);
    if (curve == MZK_CURVE_BN254) return ntt_classes_dev<BnFrX>(curve, d_data, in_len, log_n, inverse, cosets, n_classes, rows, stride, st, scale, d_src, src_stride, d_patch, skip_batch);
    set_error("unknown curve_id");
    return MZK_ERR_INVALID_ARG;
}

int32_t ntt_dispatch(int curve, uint32_t* d_data, uint64_t in_len, int log_n, bool inverse, const uint32_t* coset,
                     uint32_t batch, uint64_t stride, hipStream_t st, int scale, const uint32_t* d_src, uint64_t src_stride, const uint32_t* d_patch,
                     int skip_batch) {
    if (curve == MZK_CURVE_BLS12_381) return ntt_dev<BlsFrX>(curve, d_data, in_len, log_n, inverse, coset, batch, stride, st, scale, d_src, src_stride, d_patch, skip_batch);
    if (curve == MZK_CURVE_BN254) return ntt_dev<BnFrX>(curve, d_data, in_len, log_n, inverse, coset, batch, stride, st, scale, d_src, src_stride, d_patch, skip_batch);
    set_error("unknown curve_id");
    return MZK_ERR_INVALID_ARG;
}

void ntt_release_plans() {
    for (auto& kv : g_plans) free_plan(kv.second.get());
    g_plans.clear();
}

}  // namespace mzk
