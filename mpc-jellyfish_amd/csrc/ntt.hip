// ntt.hip -- host side of the NTT: plan cache and pass launches (kernels in ntt.cuh).
#include <map>
#include <memory>

#include "internal.hpp"
#include "ntt.cuh"

namespace mzk {
namespace {

// ---- NTT plans -----------------------------------------------------------------------------------
struct NttPlanDev {
    NttPlanHost h;
    uint32_t* d_stage[NTT_MAX_PASSES] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t *d_tlo = nullptr, *d_thi = nullptr, *d_thi_scaled = nullptr, *d_ninv = nullptr;
};
struct PlanKey {
    int curve, log_n, inverse;
    uint32_t coset[8];
    bool has_coset;
    bool operator<(const PlanKey& o) const { return std::memcmp(this, &o, sizeof(PlanKey)) < 0; }
};
std::map<PlanKey, std::unique_ptr<NttPlanDev>> g_plans;

int32_t upload_words(uint32_t** d, const std::vector<uint32_t>& v) {
    HIP_TRY(hipMalloc((void**)d, v.size() * 4));
    HIP_TRY(hipMemcpy(*d, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    return MZK_OK;
}

template <class P>
int32_t get_plan(int curve, int log_n, bool inverse, const uint32_t* coset, NttPlanDev** out) {
    PlanKey key;
    std::memset(&key, 0, sizeof key);
    key.curve = curve; key.log_n = log_n; key.inverse = inverse ? 1 : 0;
    key.has_coset = coset != nullptr;
    if (coset) std::memcpy(key.coset, coset, 32);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) { *out = it->second.get(); return MZK_OK; }
    auto pl = std::make_unique<NttPlanDev>();
    ntt_build_plan<P>(pl->h, log_n, inverse, coset);
    for (int k = 0; k < pl->h.n_pass; k++) MZK_TRY(upload_words(&pl->d_stage[k], pl->h.stage_tw[k]));
    MZK_TRY(upload_words(&pl->d_tlo, pl->h.t_lo));
    MZK_TRY(upload_words(&pl->d_thi, pl->h.t_hi));
    MZK_TRY(upload_words(&pl->d_thi_scaled, pl->h.t_hi_scaled));
    MZK_TRY(upload_words(&pl->d_ninv, pl->h.n_inv));
    *out = pl.get();
    g_plans[key] = std::move(pl);
    return MZK_OK;
}

template <class P, bool INV>
int32_t launch_pass(const NttPassArgs& a, unsigned long long n_tiles, uint32_t batch, hipStream_t st) {
    const size_t lds = ((size_t)2 * (1u << (a.log_r + a.log_c)) + 2 * (1u << a.log_r)) * 16;
    static size_t lds_max_set = 0;
    if (lds > lds_max_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)ntt_pass_kernel<P, INV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        lds_max_set = 160 * 1024;
    }
    ProfScope ps("ntt_pass", st);
    hipLaunchKernelGGL((ntt_pass_kernel<P, INV>), dim3((unsigned)n_tiles, batch), dim3(NTT_THREADS), lds, st, a);
    HIP_TRY(hipGetLastError());
    return MZK_OK;
}

// d_data: batch polynomials, `stride` elements apart, transformed in place (async on st)
template <class P>
int32_t ntt_dev(int curve, uint32_t* d_data, uint64_t in_len, int log_n, bool inverse, const uint32_t* coset,
                uint32_t batch, uint64_t stride, hipStream_t st) {
    if (log_n < 0 || log_n > P::TWO_ADICITY || log_n > 30) { set_error("log_n out of range"); return MZK_ERR_INVALID_ARG; }
    const uint64_t N = 1ull << log_n;
    if (batch == 0) return MZK_OK;
    if (stride < N || batch > 65535) { set_error("bad batch/stride"); return MZK_ERR_INVALID_ARG; }
    if (in_len > N) in_len = N;
    if (log_n == 0) return MZK_OK;   // size-1 transform is the identity (offset^0 = 1, N^-1 = 1)
    NttPlanDev* pl;
    MZK_TRY(get_plan<P>(curve, log_n, inverse, coset, &pl));
    MZK_TRY(ws_acquire(st));
    MZK_TRY(g_ws.ntt_scratch.reserve((size_t)batch * N * 32));
    uint32_t* scratch = g_ws.ntt_scratch.as<uint32_t>();
    ProfScope total("ntt_total", st);
    const int K = pl->h.n_pass;
    int log_p = 0;
    for (int k = 0; k < K; k++) {
        NttPassArgs a;
        std::memset(&a, 0, sizeof a);
        const int lr = pl->h.log_radix[k];
        a.log_n = log_n; a.log_r = lr; a.log_p = log_p; a.log_s = log_n - log_p - lr;
        a.n_pass = K; a.is_first = k == 0; a.is_final = k == K - 1;
        for (int q = 0; q < K; q++) a.log_radix[q] = pl->h.log_radix[q];
        a.log_lb = pl->h.log_lb;
        a.stage_tw = pl->d_stage[k];
        a.t_lo = pl->d_tlo;
        a.t_hi = (inverse && k == 0) ? pl->d_thi_scaled : pl->d_thi;
        a.scale = (inverse && K == 1) ? pl->d_ninv : nullptr;
        a.in_len = in_len;
        int lc = NTT_TILE_LOG - lr;
        if (lc < 0) lc = 0;
        if (a.is_final) lc = K == 1 ? 0 : (lc < pl->h.log_radix[0] ? lc : pl->h.log_radix[0]);
        else lc = lc < a.log_s ? lc : a.log_s;
        a.log_c = lc;
        // pass 1 reads the caller's buffer, middle passes run in place on scratch, the last pass
        // writes back to the caller's buffer (K = 1: data -> scratch, copied back below)
        const bool from_data = k == 0, to_data = (k == K - 1) && K > 1;
        a.in = from_data ? d_data : scratch;
        a.in_stride = from_data ? stride : N;
        a.out = to_data ? d_data : scratch;
        a.out_stride = to_data ? stride : N;
        const unsigned long long n_tiles = N >> (lr + lc);
        if (inverse) MZK_TRY((launch_pass<P, true>(a, n_tiles, batch, st)));
        else MZK_TRY((launch_pass<P, false>(a, n_tiles, batch, st)));
        log_p += lr;
    }
    if (K == 1)
        HIP_TRY(hipMemcpy2DAsync(d_data, stride * 32, scratch, N * 32, N * 32, batch, hipMemcpyDeviceToDevice, st));
    MZK_TRY(ws_release(st));
    return MZK_OK;
}

}  // namespace

int32_t ntt_dispatch(int curve, uint32_t* d_data, uint64_t in_len, int log_n, bool inverse, const uint32_t* coset,
                     uint32_t batch, uint64_t stride, hipStream_t st) {
    if (curve == MZK_CURVE_BLS12_381) return ntt_dev<BlsFr>(curve, d_data, in_len, log_n, inverse, coset, batch, stride, st);
    if (curve == MZK_CURVE_BN254) return ntt_dev<BnFr>(curve, d_data, in_len, log_n, inverse, coset, batch, stride, st);
    set_error("unknown curve_id");
    return MZK_ERR_INVALID_ARG;
}


void ntt_release_plans() {
    for (auto& kv : g_plans) {
        NttPlanDev* p = kv.second.get();
        for (auto* d : p->d_stage) if (d) (void)hipFree(d);
        for (auto* d : {p->d_tlo, p->d_thi, p->d_thi_scaled, p->d_ninv}) if (d) (void)hipFree(d);
    }
    g_plans.clear();
}

}  // namespace mzk
