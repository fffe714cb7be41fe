// mzk.hip -- libmi355zk: the C ABI (include/mzk.h) and the state shared by ntt.hip / msm.hip / plonk.hip / poly.hip.
// One context per (logical) device; an entry point binds the calling thread to the context of its handle -- or, for calls on
// bare device pointers, to the thread's current device -- and enqueues under that context's lock (internal.hpp).
#include <condition_variable>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "internal.hpp"

namespace mzk {

thread_local std::string g_last_error;
std::string g_last_error_global;
std::mutex g_last_error_lock;                 // the process-wide copy is written by any failing thread (the thread-local one needs no lock)
void set_error(const std::string& s) {
    g_last_error = s;
    std::lock_guard<std::mutex> g(g_last_error_lock);
    g_last_error_global = s;
}

int32_t DevBuf::reserve(size_t bytes) {
    if (bytes <= cap) return MZK_OK;
    if (p) HIP_TRY(hipFree(p));
    p = nullptr;
    cap = 0;
    HIP_TRY(hipMalloc(&p, bytes));
    cap = bytes;
    return MZK_OK;
}
void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}
void Workspace::release() {
    for (DevBuf* b : {&ntt_scratch, &scalars, &hist, &offs, &cursor, &sorted, &buckets, &collect, &io, &misc, &digits, &long_desc, &long_parts, &plonk_polys, &plonk_out, &pre_cnt, &pre_off, &pre_ce, &pre_cb, &poly_tmp, &split, &link_tmp, &occ}) b->release();
    if (h_collect) (void)hipHostFree(h_collect);
    h_collect = nullptr;
    h_collect_cap = 0;
    if (last_use) (void)hipEventDestroy(last_use);
    last_use = nullptr;
    heavy_frac = 0.125;
}

size_t Workspace::bytes() {
    size_t b = 0;
    for (DevBuf* d : {&ntt_scratch, &scalars, &hist, &offs, &cursor, &sorted, &buckets, &collect, &io, &misc, &digits, &long_desc, &long_parts, &plonk_polys, &plonk_out, &pre_cnt, &pre_off, &pre_ce, &pre_cb, &poly_tmp, &split, &link_tmp, &occ}) b += d->cap;
    return b;
}

int32_t ws_acquire(hipStream_t st) {
    if (!g_ws.last_use) HIP_TRY(hipEventCreateWithFlags(&g_ws.last_use, hipEventDisableTiming));
    else HIP_TRY(hipStreamWaitEvent(st, g_ws.last_use, 0));
    return MZK_OK;
}
int32_t ws_release(hipStream_t st) {
    HIP_TRY(hipEventRecord(g_ws.last_use, st));
    return MZK_OK;
}

std::atomic<bool> g_prof{false};
std::atomic<uint64_t> g_launches{0};
ProfScope::ProfScope(const char* n, hipStream_t s) : st(s), name(n) {
    if (!g_prof) return;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
    (void)hipEventRecord(a, st);
}
ProfScope::~ProfScope() {
    if (!a) return;
    (void)hipEventRecord(b, st);
    cur().prof_recs.push_back({name, a, b});
}

// ---- contexts ------------------------------------------------------------------------------------------
Ctx g_ctx[MAX_CTX];
std::mutex g_ctx_lock;                        // guards `init` / `device` of the table and g_default
std::atomic<int> g_default{-1};               // first context initialised: what threads without a binding of their own use
thread_local int t_dev = -1;                  // logical device this thread is bound to (mzk_init / mzk_set_device)
thread_local Ctx* t_ctx = nullptr;            // context of the entry point now running on this thread
Ctx& cur() { return *t_ctx; }

}  // namespace mzk

using namespace mzk;

namespace {

Ctx* ctx_for_thread() {
    const int d = t_dev >= 0 ? t_dev : g_default.load();
    if (d < 0 || !g_ctx[d].init) { set_error("mzk_init has not been called"); return nullptr; }
    return &g_ctx[d];
}
Ctx* ctx_for_handle(uint64_t h) {
    const int d = handle_ctx(h);
    if (d < 0 || d >= MAX_CTX || !g_ctx[d].init) { set_error("unknown handle"); return nullptr; }
    return &g_ctx[d];
}
// binds the calling thread to a context for the duration of an entry point (nested entry points restore the outer one)
struct CtxBind {
    Ctx* prev;
    int32_t rc = MZK_OK;
    explicit CtxBind(Ctx* c) : prev(t_ctx) {
        t_ctx = c;
        const hipError_t e = hipSetDevice(c->device);
        if (e != hipSuccess) { set_error(std::string("hipSetDevice: ") + hipGetErrorString(e)); rc = MZK_ERR_HIP; }
    }
    ~CtxBind() { t_ctx = prev; }
};
#define BIND_CUR()                               \
    Ctx* cx_ = ctx_for_thread();                 \
    if (!cx_) return MZK_ERR_NOT_INIT;           \
    CtxBind bind_(cx_);                          \
    MZK_TRY(bind_.rc)
#define BIND_HANDLE(h)                           \
    Ctx* cx_ = ctx_for_handle(h);                \
    if (!cx_) return MZK_ERR_BAD_HANDLE;         \
    CtxBind bind_(cx_);                          \
    MZK_TRY(bind_.rc)
#define ENTER_CUR() \
    BIND_CUR();     \
    std::lock_guard<std::mutex> lk(cx_->lock)
#define ENTER_HANDLE(h) \
    BIND_HANDLE(h);     \
    std::lock_guard<std::mutex> lk(cx_->lock)

// ---- host-pointer I/O slots ------------------------------------------------------------------------
// The host-pointer entry points (mzk_ntt, mzk_ntt_batch, mzk_msm, mzk_msm_batch: what a shim that swaps only the two
// third-party call sites uses) move their operands over PCIe.  Each call -- or each polynomial of a
// batch -- takes one of the context's IO_SLOTS slots: a non-blocking stream plus a device buffer.  Upload, kernels and download of
// one polynomial are enqueued on its slot's stream; the context's lock is held only while the kernels are ENQUEUED (plan cache and
// shared workspace; ws_acquire / ws_release order the kernels of different streams on the shared scratch), never across a
// transfer or a wait.  So the upload of polynomial k+1 and the download of k-1 overlap the transform of k -- inside one batch
// call and between concurrent callers (the reference commits and transforms from a Rayon par_iter, prover.rs:552-562,
// univariate_kzg/mod.rs:125-127).  Transfers are asynchronous when the host memory is page-locked (mzk_host_alloc /
// mzk_host_register); from pageable memory the runtime stages them and the enqueue blocks, which is still correct.

// block: wait for a free slot; !block: take one only if one is free right now (a caller that already holds a slot must never wait
// for another: two batch calls doing so would deadlock), *out_idx = -1 otherwise
int32_t io_acquire(Ctx& cx, int* out_idx, bool block = true) {
    std::unique_lock<std::mutex> lk(cx.io_lock);
    int idx = -1;
    auto find = [&] {
        for (int i = 0; i < IO_SLOTS; i++)
            if (!cx.io[i].busy) { idx = i; return true; }
        return false;
    };
    if (block) cx.io_cv.wait(lk, find);
    else if (!find()) { *out_idx = -1; return MZK_OK; }
    cx.io[idx].busy = true;
    lk.unlock();
    if (!cx.io[idx].st) {
        hipError_t e = hipStreamCreateWithFlags(&cx.io[idx].st, hipStreamNonBlocking);
        if (e != hipSuccess) {
            { std::lock_guard<std::mutex> g(cx.io_lock); cx.io[idx].busy = false; }
            cx.io_cv.notify_one();
            set_error(std::string("hipStreamCreate: ") + hipGetErrorString(e));
            return MZK_ERR_HIP;
        }
    }
    *out_idx = idx;
    return MZK_OK;
}
void io_release(Ctx& cx, int idx) {
    { std::lock_guard<std::mutex> g(cx.io_lock); cx.io[idx].busy = false; }
    cx.io_cv.notify_one();
}
struct IoGuard {                      // releases the slot on every return path
    Ctx* cx = nullptr;
    int idx = -1;
    ~IoGuard() { if (idx >= 0) io_release(*cx, idx); }
};
void io_release_all(Ctx& cx) {
    for (auto& sl : cx.io) {
        if (sl.st) { (void)hipStreamSynchronize(sl.st); (void)hipStreamDestroy(sl.st); sl.st = nullptr; }
        sl.buf.release();
        sl.busy = false;
    }
}

// logical -> physical device.  MZK_VIRTUAL_DEVICES=G: logical devices 0..G-1 exist whatever the box has, spread round-robin
int virtual_devices() {
    const char* v = std::getenv("MZK_VIRTUAL_DEVICES");
    const int g = v ? std::atoi(v) : 0;
    return g > 0 ? (g < MAX_CTX ? g : MAX_CTX) : 0;
}

}  // namespace

// =================================================================================================
extern "C" {

int32_t mzk_init(int32_t device) {
    std::lock_guard<std::mutex> lk(g_ctx_lock);
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        set_error("no HIP device visible (libmi355zk has no CPU fallback)");
        return MZK_ERR_NO_DEVICE;
    }
    const int virt = virtual_devices();
    if (device < 0) {
        if (t_dev >= 0 || g_default >= 0) return MZK_OK;     // already bound: idempotent
        int curdev = 0;
        HIP_TRY(hipGetDevice(&curdev));
        device = curdev;
    }
    if (device >= (virt ? virt : count) || device >= MAX_CTX) { set_error("device index out of range"); return MZK_ERR_INVALID_ARG; }
    Ctx& cx = g_ctx[device];
    if (!cx.init) {
        cx.logical = device;
        cx.device = virt ? device % count : device;
        HIP_TRY(hipSetDevice(cx.device));
        for (int o = 0; o < MAX_CTX; o++)                        // peer access both ways where the hardware offers it (xGMI): the
            if (g_ctx[o].init && g_ctx[o].device != cx.device) {  // class exchange of a multi-GPU proof is a device-to-device copy
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, cx.device, g_ctx[o].device) == hipSuccess && can) {
                    (void)hipDeviceEnablePeerAccess(g_ctx[o].device, 0);
                    (void)hipSetDevice(g_ctx[o].device);
                    (void)hipDeviceEnablePeerAccess(cx.device, 0);
                    (void)hipSetDevice(cx.device);
                }
            }
        (void)hipGetLastError();                                 // "peer access already enabled" is not an error
        cx.init = true;
    }
    HIP_TRY(hipSetDevice(cx.device));
    t_dev = device;
    if (g_default < 0) g_default = device;
    return MZK_OK;
}

int32_t mzk_set_device(int32_t device) {
    std::lock_guard<std::mutex> lk(g_ctx_lock);
    if (device < 0 || device >= MAX_CTX || !g_ctx[device].init) { set_error("mzk_set_device: mzk_init(device) has not been called"); return MZK_ERR_NOT_INIT; }
    HIP_TRY(hipSetDevice(g_ctx[device].device));
    t_dev = device;
    return MZK_OK;
}

int32_t mzk_get_device(int32_t* out_device) {
    if (!out_device) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    *out_device = t_dev >= 0 ? t_dev : g_default.load();
    return *out_device >= 0 ? MZK_OK : MZK_ERR_NOT_INIT;
}

int32_t mzk_device_count(int32_t* out_count) {
    if (!out_count) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { set_error("no HIP device visible"); return MZK_ERR_NO_DEVICE; }
    const int virt = virtual_devices();
    *out_count = virt ? virt : (count < MAX_CTX ? count : MAX_CTX);
    return MZK_OK;
}

int32_t mzk_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_ctx_lock);
    for (int d = 0; d < MAX_CTX; d++) {
        Ctx& cx = g_ctx[d];
        if (!cx.init) continue;
        CtxBind bind(&cx);
        std::lock_guard<std::mutex> lk2(cx.lock);
        (void)hipDeviceSynchronize();
        for (auto& kv : cx.srs) { (void)hipFree(kv.second.d_xy); if (kv.second.d_int) (void)hipFree(kv.second.d_int); if (kv.second.d_pre) (void)hipFree(kv.second.d_pre); }
        cx.srs.clear();
        io_release_all(cx);
        ntt_release_plans();
        msm_release_streams();
        plonk_release_all();
        for (auto& r : cx.prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        cx.prof_recs.clear();
        cx.ws.release();
        cx.init = false;
    }
    g_default = -1;
    t_dev = -1;
    return MZK_OK;
}

const char* mzk_strerror(int32_t code) {
    switch (code) {
        case MZK_OK: return "ok";
        case MZK_ERR_INVALID_ARG: return "invalid argument";
        case MZK_ERR_HIP: return "HIP runtime error";
        case MZK_ERR_NO_DEVICE: return "no HIP device";
        case MZK_ERR_BAD_HANDLE: return "unknown handle";
        case MZK_ERR_UNSUPPORTED: return "unsupported";
        case MZK_ERR_OOM: return "out of device memory";
        case MZK_ERR_NOT_INIT: return "mzk_init not called";
        case MZK_ERR_LOOKUP: return "Plookup: lookup value outside the table";
        case MZK_ERR_WRONG_QUOTIENT_DEGREE: return "WrongQuotientPolyDegree";
        case MZK_ERR_STATE: return "prover rounds called out of order";
        default: return "unknown error";
    }
}
const char* mzk_last_error(void) {
    if (!mzk::g_last_error.empty()) return mzk::g_last_error.c_str();
    // no error on this thread yet: the latest one of the process, copied under its lock into this thread's own buffer
    thread_local std::string copy;
    { std::lock_guard<std::mutex> g(mzk::g_last_error_lock); copy = mzk::g_last_error_global; }
    return copy.c_str();
}
const char* mzk_version(void) { return "libmi355zk 0.3 (gfx950)"; }

// ---- SRS -----------------------------------------------------------------------------------------
int32_t mzk_srs_register(int32_t curve_id, const uint64_t* xy_mont, uint64_t n_points, uint64_t* out_handle) {
    ENTER_CUR();
    if ((curve_id != 0 && curve_id != 1) || !out_handle || (!xy_mont && n_points)) { set_error("bad argument"); return MZK_ERR_INVALID_ARG; }
    Srs s{curve_id, n_points, nullptr, nullptr, nullptr, 0};
    const size_t bytes = (size_t)n_points * 2 * fq_words(curve_id) * 4;
    HIP_TRY(hipMalloc((void**)&s.d_xy, bytes ? bytes : 4));
    if (bytes) HIP_TRY(hipMemcpy(s.d_xy, xy_mont, bytes, hipMemcpyHostToDevice));
    MZK_TRY(srs_build_internal(s, nullptr));
    *out_handle = handle_make(cx_->logical, cx_->next_handle++);
    cx_->srs[*out_handle] = s;
    return MZK_OK;
}
int32_t mzk_srs_register_dev(int32_t curve_id, const void* d_xy_mont, uint64_t n_points, uint64_t* out_handle, void* stream) {
    ENTER_CUR();
    if ((curve_id != 0 && curve_id != 1) || !out_handle || (!d_xy_mont && n_points)) { set_error("bad argument"); return MZK_ERR_INVALID_ARG; }
    Srs s{curve_id, n_points, nullptr, nullptr, nullptr, 0};
    const size_t bytes = (size_t)n_points * 2 * fq_words(curve_id) * 4;
    HIP_TRY(hipMalloc((void**)&s.d_xy, bytes ? bytes : 4));
    if (bytes) {
        HIP_TRY(hipMemcpyAsync(s.d_xy, d_xy_mont, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    }
    MZK_TRY(srs_build_internal(s, (hipStream_t)stream));
    *out_handle = handle_make(cx_->logical, cx_->next_handle++);
    cx_->srs[*out_handle] = s;
    return MZK_OK;
}
// a new SRS holding the points [first, first + n_points) of a registered one (on its device): a rank of a multi-GPU prover keeps only
// the range it commits over -- 1 / G of the points and of the fixed-base table, whose window then follows the slice's size
int32_t mzk_srs_slice(uint64_t handle, uint64_t first, uint64_t n_points, uint64_t* out_handle) {
    ENTER_HANDLE(handle);
    auto it = cx_->srs.find(handle);
    if (it == cx_->srs.end()) { set_error("unknown SRS handle"); return MZK_ERR_BAD_HANDLE; }
    if (!out_handle || first > it->second.n || n_points > it->second.n - first) { set_error("slice outside the SRS"); return MZK_ERR_INVALID_ARG; }   // (no u64 wrap)
    const Srs& src = it->second;
    Srs s{src.curve, n_points, nullptr, nullptr, nullptr, 0};
    const size_t pt = (size_t)2 * fq_words(src.curve) * 4, bytes = (size_t)n_points * pt;
    HIP_TRY(hipMalloc((void**)&s.d_xy, bytes ? bytes : 4));
    int32_t rc = MZK_OK;
    if (bytes && hipMemcpy(s.d_xy, reinterpret_cast<const uint8_t*>(src.d_xy) + first * pt, bytes, hipMemcpyDeviceToDevice) != hipSuccess) {
        set_error("hipMemcpy of the SRS slice failed");
        rc = MZK_ERR_HIP;
    }
    if (rc == MZK_OK) rc = srs_build_internal(s, nullptr);
    if (rc != MZK_OK) {                                               // nothing of a failed slice stays allocated
        (void)hipFree(s.d_xy);
        if (s.d_int) (void)hipFree(s.d_int);
        return rc;
    }
    *out_handle = handle_make(cx_->logical, cx_->next_handle++);
    cx_->srs[*out_handle] = s;
    return MZK_OK;
}
int32_t mzk_srs_release(uint64_t handle) {
    ENTER_HANDLE(handle);
    auto it = cx_->srs.find(handle);
    if (it == cx_->srs.end()) { set_error("unknown SRS handle"); return MZK_ERR_BAD_HANDLE; }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipFree(it->second.d_xy));
    if (it->second.d_int) HIP_TRY(hipFree(it->second.d_int));
    if (it->second.d_pre) HIP_TRY(hipFree(it->second.d_pre));
    cx_->srs.erase(it);
    return MZK_OK;
}
int32_t mzk_srs_generate_for_testing_g(int32_t curve_id, const uint64_t* beta_canonical, const uint64_t* g_xy_mont, uint64_t n_points, uint64_t* out_handle) {
    ENTER_CUR();
    if ((curve_id != 0 && curve_id != 1) || !out_handle || !beta_canonical) { set_error("bad argument"); return MZK_ERR_INVALID_ARG; }
    Srs s{curve_id, n_points, nullptr, nullptr, nullptr, 0};
    const size_t bytes = (size_t)n_points * 2 * fq_words(curve_id) * 4;
    HIP_TRY(hipMalloc((void**)&s.d_xy, bytes ? bytes : 4));
    int32_t rc = MZK_OK;
    if (n_points) {
        const uint32_t* beta = reinterpret_cast<const uint32_t*>(beta_canonical);
        rc = srs_generate_dispatch(curve_id, beta, reinterpret_cast<const uint32_t*>(g_xy_mont), n_points, s.d_xy);
    }
    if (rc == MZK_OK) rc = srs_build_internal(s, nullptr);
    if (rc != MZK_OK) { (void)hipFree(s.d_xy); return rc; }
    *out_handle = handle_make(cx_->logical, cx_->next_handle++);
    cx_->srs[*out_handle] = s;
    return MZK_OK;
}
int32_t mzk_srs_generate_lagrange_for_testing(int32_t curve_id, const uint64_t* beta_canonical, const uint64_t* g_xy_mont, uint32_t log_n, uint32_t n_extra,
                                              uint64_t* out_handle) {
    ENTER_CUR();
    if ((curve_id != 0 && curve_id != 1) || !out_handle || !beta_canonical || log_n > 27 || n_extra > 16) { set_error("bad argument"); return MZK_ERR_INVALID_ARG; }
    const uint64_t n_points = (1ull << log_n) + n_extra;
    Srs s{curve_id, n_points, nullptr, nullptr, nullptr, 0};
    HIP_TRY(hipMalloc((void**)&s.d_xy, (size_t)n_points * 2 * fq_words(curve_id) * 4));
    int32_t rc = srs_lagrange_generate_dispatch(curve_id, reinterpret_cast<const uint32_t*>(beta_canonical), reinterpret_cast<const uint32_t*>(g_xy_mont), (int)log_n,
                                                n_extra, s.d_xy);
    if (rc == MZK_OK) rc = srs_build_internal(s, nullptr);
    if (rc != MZK_OK) { (void)hipFree(s.d_xy); return rc; }
    *out_handle = handle_make(cx_->logical, cx_->next_handle++);
    cx_->srs[*out_handle] = s;
    return MZK_OK;
}
int32_t mzk_srs_lagrange_from_srs(uint64_t srs_handle, uint32_t log_n, uint32_t n_extra, uint64_t* out_handle) {
    ENTER_HANDLE(srs_handle);
    auto it = cx_->srs.find(srs_handle);
    if (it == cx_->srs.end()) { set_error("unknown SRS handle"); return MZK_ERR_BAD_HANDLE; }
    const Srs src = it->second;
    if (!out_handle || log_n > 27 || n_extra > 16 || (1ull << log_n) + n_extra > src.n) {
        set_error("bad argument (the SRS must hold 2^log_n + n_extra points)");
        return MZK_ERR_INVALID_ARG;
    }
    const uint64_t n_points = (1ull << log_n) + n_extra;
    Srs s{src.curve, n_points, nullptr, nullptr, nullptr, 0};
    HIP_TRY(hipMalloc((void**)&s.d_xy, (size_t)n_points * 2 * fq_words(src.curve) * 4));
    int32_t rc = srs_lagrange_from_points_dispatch(src.curve, src.d_xy, (int)log_n, n_extra, s.d_xy);
    if (rc == MZK_OK) rc = srs_build_internal(s, nullptr);
    if (rc != MZK_OK) { (void)hipFree(s.d_xy); return rc; }
    *out_handle = handle_make(cx_->logical, cx_->next_handle++);
    cx_->srs[*out_handle] = s;
    return MZK_OK;
}
int32_t mzk_srs_generate_for_testing(int32_t curve_id, const uint64_t* beta_canonical, uint64_t n_points, uint64_t* out_handle) {
    return mzk_srs_generate_for_testing_g(curve_id, beta_canonical, nullptr, n_points, out_handle);
}
int32_t mzk_srs_download(uint64_t handle, uint64_t first, uint64_t n_points, uint64_t* out_xy_mont) {
    ENTER_HANDLE(handle);
    auto it = cx_->srs.find(handle);
    if (it == cx_->srs.end()) { set_error("unknown SRS handle"); return MZK_ERR_BAD_HANDLE; }
    const Srs& s = it->second;
    if (first > s.n || n_points > s.n - first) { set_error("range outside the SRS"); return MZK_ERR_INVALID_ARG; }
    const size_t pw = (size_t)2 * fq_words(s.curve);
    HIP_TRY(hipDeviceSynchronize());
    if (n_points) HIP_TRY(hipMemcpy(out_xy_mont, s.d_xy + first * pw, n_points * pw * 4, hipMemcpyDeviceToHost));
    return MZK_OK;
}
int32_t mzk_srs_len(uint64_t handle, uint64_t* out_n_points) {
    ENTER_HANDLE(handle);
    auto it = cx_->srs.find(handle);
    if (it == cx_->srs.end() || !out_n_points) { set_error("unknown SRS handle"); return MZK_ERR_BAD_HANDLE; }
    *out_n_points = it->second.n;
    return MZK_OK;
}
// HBM a registered SRS holds: its points (boundary form + the MSM's internal reduced-radix form) and its fixed-base table
int32_t mzk_srs_hbm_bytes(uint64_t handle, uint64_t* out_points_bytes, uint64_t* out_table_bytes) {
    ENTER_HANDLE(handle);
    auto it = cx_->srs.find(handle);
    if (it == cx_->srs.end()) { set_error("unknown SRS handle"); return MZK_ERR_BAD_HANDLE; }
    const Srs& s = it->second;
    const uint64_t aff_int = s.curve == MZK_CURVE_BLS12_381 ? 2 * 14 * 4 : 2 * 9 * 4;       // 29-bit limbs: ecx.cuh
    if (out_points_bytes) *out_points_bytes = s.n * (uint64_t)(2 * fq_words(s.curve) * 4) + (s.d_int ? s.n * aff_int : 0);
    if (out_table_bytes) *out_table_bytes = s.d_pre ? (uint64_t)s.pre_levels * s.n * aff_int : 0;
    return MZK_OK;
}
int32_t mzk_launch_count(uint64_t* out_launches) {
    if (!out_launches) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    *out_launches = g_launches.load(std::memory_order_relaxed);
    return MZK_OK;
}
// releases the grow-only scratch of the calling thread's device context (it is re-acquired on demand): a process that has run its
// largest problem and goes on with smaller ones, or bench.py before it reports what ONE kind of proof holds
int32_t mzk_workspace_release(void) {
    ENTER_CUR();
    HIP_TRY(hipDeviceSynchronize());
    cx_->ws.release();
    for (auto& slot : cx_->io) slot.buf.release();
    return MZK_OK;
}
// scratch memory of the calling thread's device context: the shared workspace of the NTT / MSM / quotient kernels (grow-only) and the
// device buffers of the host-pointer I/O slots
int32_t mzk_workspace_hbm_bytes(uint64_t* out_bytes) {
    ENTER_CUR();
    if (!out_bytes) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    uint64_t b = cx_->ws.bytes();
    for (auto& slot : cx_->io) b += slot.buf.cap;
    *out_bytes = b;
    if (std::getenv("MZK_WS_DEBUG")) {                                // one line per buffer of the grow-only scratch (tools/hbm_report.py)
        Workspace& w = cx_->ws;
        const char* names[] = {"ntt_scratch", "scalars", "hist", "offs", "cursor", "sorted", "buckets", "collect", "io", "misc", "digits", "long_desc", "long_parts",
                               "plonk_polys", "plonk_out", "pre_cnt", "pre_off", "pre_ce", "pre_cb", "poly_tmp", "split", "link_tmp", "occ"};
        int i = 0;
        for (DevBuf* d : {&w.ntt_scratch, &w.scalars, &w.hist, &w.offs, &w.cursor, &w.sorted, &w.buckets, &w.collect, &w.io, &w.misc, &w.digits, &w.long_desc,
                          &w.long_parts, &w.plonk_polys, &w.plonk_out, &w.pre_cnt, &w.pre_off, &w.pre_ce, &w.pre_cb, &w.poly_tmp, &w.split, &w.link_tmp, &w.occ}) {
            if (d->cap) std::fprintf(stderr, "[mzk ws] %-12s %10.1f MB\n", names[i], d->cap / 1e6);
            i++;
        }
        for (auto& slot : cx_->io)
            if (slot.buf.cap) std::fprintf(stderr, "[mzk ws] io slot      %10.1f MB\n", slot.buf.cap / 1e6);
    }
    return MZK_OK;
}

// ---- MSM -----------------------------------------------------------------------------------------
int32_t mzk_msm_dev(uint64_t srs_handle, uint64_t base_offset, const void* d_scalars, uint64_t n, int32_t scalars_are_mont,
                    uint64_t* out_xyz_mont, void* stream) {
    ENTER_HANDLE(srs_handle);
    auto it = cx_->srs.find(srs_handle);
    if (it == cx_->srs.end()) { set_error("unknown SRS handle"); return MZK_ERR_BAD_HANDLE; }
    if (!out_xyz_mont || (!d_scalars && n)) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    return msm_dispatch(it->second, base_offset, reinterpret_cast<const uint32_t*>(d_scalars), n, scalars_are_mont != 0,
                        reinterpret_cast<uint32_t*>(out_xyz_mont), (hipStream_t)stream);
}

// host scalars: upload on an I/O slot outside the lock (it overlaps whatever MSM or NTT another caller is running), compute under it
static int32_t msm_host(uint64_t srs_handle, uint64_t base_offset, const uint64_t* scalars, uint64_t n, int32_t is_mont, uint64_t* out, int* out_curve) {
    BIND_HANDLE(srs_handle);
    if (!out || (!scalars && n)) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    {   // nothing is read from `scalars` before the range has been checked against the SRS
        std::lock_guard<std::mutex> lk(cx_->lock);
        auto it = cx_->srs.find(srs_handle);
        if (it == cx_->srs.end()) { set_error("unknown SRS handle"); return MZK_ERR_BAD_HANDLE; }
        if (base_offset > it->second.n || n > it->second.n - base_offset) {
            set_error("MSM longer than the registered SRS (poly degree larger than allowed)");
            return MZK_ERR_INVALID_ARG;
        }
        if (n >= (1ull << 27)) { set_error("MSM size must be < 2^27"); return MZK_ERR_INVALID_ARG; }
    }
    IoGuard slot;
    slot.cx = cx_;
    MZK_TRY(io_acquire(*cx_, &slot.idx));
    IoSlot& io = cx_->io[slot.idx];
    if (n) {
        MZK_TRY(io.buf.reserve(n * 32));
        HIP_TRY(hipMemcpyAsync(io.buf.p, scalars, n * 32, hipMemcpyHostToDevice, io.st));
    }
    std::lock_guard<std::mutex> lk(cx_->lock);
    auto it = cx_->srs.find(srs_handle);                                // released by another thread meanwhile?
    if (it == cx_->srs.end()) { (void)hipStreamSynchronize(io.st); set_error("unknown SRS handle"); return MZK_ERR_BAD_HANDLE; }
    if (out_curve) *out_curve = it->second.curve;
    const int32_t rc = msm_dispatch(it->second, base_offset, io.buf.as<uint32_t>(), n, is_mont != 0, reinterpret_cast<uint32_t*>(out), io.st);
    if (rc != MZK_OK) (void)hipStreamSynchronize(io.st);          // the slot's buffer must be idle before it is handed on
    return rc;
}

int32_t mzk_msm(uint64_t srs_handle, uint64_t base_offset, const uint64_t* scalars, uint64_t n, int32_t scalars_are_mont, uint64_t* out_xyz_mont) {
    return msm_host(srs_handle, base_offset, scalars, n, scalars_are_mont, out_xyz_mont, nullptr);
}

int32_t mzk_msm_batch_dev(uint64_t srs_handle, uint32_t n_polys, const void* const* d_scalars, const uint64_t* lens, const uint64_t* base_offsets,
                          int32_t scalars_are_mont, uint64_t* out_xyz_mont, void* stream) {
    ENTER_HANDLE(srs_handle);
    auto it = cx_->srs.find(srs_handle);
    if (it == cx_->srs.end()) { set_error("unknown SRS handle"); return MZK_ERR_BAD_HANDLE; }
    if (n_polys && (!d_scalars || !lens || !out_xyz_mont)) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    return msm_batch_dispatch(it->second, n_polys, reinterpret_cast<const uint32_t* const*>(d_scalars), lens, base_offsets, scalars_are_mont != 0,
                              reinterpret_cast<uint32_t*>(out_xyz_mont), (hipStream_t)stream);
}

int32_t mzk_msm_batch(uint64_t srs_handle, uint32_t n_polys, const uint64_t* const* scalars, const uint64_t* lens, const uint64_t* base_offsets,
                      int32_t scalars_are_mont, uint64_t* out_xyz_mont) {
    BIND_HANDLE(srs_handle);
    if (n_polys && (!scalars || !lens || !out_xyz_mont)) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_polys; i++) {
        if (lens[i] && !scalars[i]) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
        if (lens[i] >= (1ull << 27)) { set_error("MSM size must be < 2^27"); return MZK_ERR_INVALID_ARG; }
        total += lens[i];
    }
    {   // nothing is read from the scalar arrays before the ranges have been checked against the SRS
        std::lock_guard<std::mutex> lk(cx_->lock);
        auto it = cx_->srs.find(srs_handle);
        if (it == cx_->srs.end()) { set_error("unknown SRS handle"); return MZK_ERR_BAD_HANDLE; }
        for (uint32_t i = 0; i < n_polys; i++) {
            const uint64_t off = base_offsets ? base_offsets[i] : 0;
            if (off > it->second.n || lens[i] > it->second.n - off) {
                set_error("MSM longer than the registered SRS (poly degree larger than allowed)");
                return MZK_ERR_INVALID_ARG;
            }
        }
    }
    // one upload slab on an I/O slot (outside the lock), then the fused batch
    IoGuard slot;
    slot.cx = cx_;
    MZK_TRY(io_acquire(*cx_, &slot.idx));
    IoSlot& io = cx_->io[slot.idx];
    MZK_TRY(io.buf.reserve((total ? total : 1) * 32));
    std::vector<const uint32_t*> dptr(n_polys);
    uint64_t off = 0;
    for (uint32_t i = 0; i < n_polys; i++) {
        dptr[i] = io.buf.as<uint32_t>() + off * 8;
        if (lens[i]) HIP_TRY(hipMemcpyAsync(const_cast<uint32_t*>(dptr[i]), scalars[i], lens[i] * 32, hipMemcpyHostToDevice, io.st));
        off += lens[i];
    }
    std::lock_guard<std::mutex> lk(cx_->lock);
    auto it = cx_->srs.find(srs_handle);
    if (it == cx_->srs.end()) { (void)hipStreamSynchronize(io.st); set_error("unknown SRS handle"); return MZK_ERR_BAD_HANDLE; }
    const int32_t rc = msm_batch_dispatch(it->second, n_polys, dptr.data(), lens, base_offsets, scalars_are_mont != 0, reinterpret_cast<uint32_t*>(out_xyz_mont), io.st);
    if (rc != MZK_OK) (void)hipStreamSynchronize(io.st);
    return rc;
}

int32_t mzk_msm_affine(uint64_t srs_handle, uint64_t base_offset, const uint64_t* scalars, uint64_t n, int32_t scalars_are_mont, uint64_t* out_xy_mont) {
    uint64_t xyz[18];
    int curve = 0;
    if (!out_xy_mont) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    MZK_TRY(msm_host(srs_handle, base_offset, scalars, n, scalars_are_mont, xyz, &curve));
    jac_to_affine_host_dispatch(curve, xyz, 1, out_xy_mont);
    return MZK_OK;
}

int32_t mzk_g1_sum_jacobian(int32_t curve_id, const uint64_t* xyz_mont, uint64_t n, uint64_t* out_xyz_mont) {
    if ((curve_id != 0 && curve_id != 1) || !out_xyz_mont || (!xyz_mont && n)) { set_error("bad argument"); return MZK_ERR_INVALID_ARG; }
    jac_sum_host_dispatch(curve_id, xyz_mont, n, out_xyz_mont);
    return MZK_OK;
}

int32_t mzk_g1_jacobian_to_affine(int32_t curve_id, const uint64_t* xyz_mont, uint64_t n, uint64_t* out_xy_mont) {
    if ((curve_id != 0 && curve_id != 1) || ((!xyz_mont || !out_xy_mont) && n)) { set_error("bad argument"); return MZK_ERR_INVALID_ARG; }
    const int L = fq_words(curve_id) / 2;
    (void)L;
    jac_to_affine_host_dispatch(curve_id, xyz_mont, n, out_xy_mont);
    return MZK_OK;
}

// ---- NTT -----------------------------------------------------------------------------------------
int32_t mzk_ntt_dev(int32_t curve_id, void* d_data_mont, uint64_t in_len, uint32_t log_n, int32_t inverse, const uint64_t* coset_offset_mont,
                    uint32_t batch, uint64_t batch_stride, void* stream) {
    ENTER_CUR();
    if (!d_data_mont) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    return ntt_dispatch(curve_id, reinterpret_cast<uint32_t*>(d_data_mont), in_len, (int)log_n, inverse != 0,
                        reinterpret_cast<const uint32_t*>(coset_offset_mont), batch, batch_stride, (hipStream_t)stream);
}

int32_t mzk_ntt_batch(int32_t curve_id, uint32_t n_polys, uint64_t* const* data_mont, const uint64_t* in_lens, uint32_t log_n, int32_t inverse,
                      const uint64_t* coset_offset_mont) {
    BIND_CUR();
    if (log_n > 30) { set_error("log_n out of range"); return MZK_ERR_INVALID_ARG; }
    if (n_polys && (!data_mont || !in_lens)) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    for (uint32_t i = 0; i < n_polys; i++)
        if (!data_mont[i]) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    const uint64_t N = 1ull << log_n;
    // polynomial i runs on slot i mod (slots this call could get): upload, passes and download are enqueued on the slot's stream,
    // the lock is taken for the enqueue of the passes only, and a slot is waited for only when it comes round again
    constexpr int PIPE = 3;
    IoGuard slot[PIPE];
    const int want = (int)(n_polys < (uint32_t)PIPE ? n_polys : (uint32_t)PIPE);
    int n_slots = 0;
    for (int k = 0; k < want; k++) {
        slot[n_slots].cx = cx_;
        MZK_TRY(io_acquire(*cx_, &slot[n_slots].idx, /*block=*/k == 0));    // wait for the first slot only; take more if they are free
        if (slot[n_slots].idx < 0) break;
        MZK_TRY(cx_->io[slot[n_slots].idx].buf.reserve(N * 32));
        n_slots++;
    }
    if (n_polys == 0) return MZK_OK;
    // (a failed HIP call must not hand a slot on while its stream still owns the buffer: every exit path drains the slots first)
    auto hip_ok = [](hipError_t e, const char* what) -> int32_t {
        if (e == hipSuccess) return MZK_OK;
        set_error(std::string(what) + ": " + hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? MZK_ERR_OOM : MZK_ERR_HIP;
    };
    int32_t rc = MZK_OK;
    for (uint32_t i = 0; i < n_polys && rc == MZK_OK; i++) {
        IoSlot& io = cx_->io[slot[i % n_slots].idx];
        if (i >= (uint32_t)n_slots) rc = hip_ok(hipStreamSynchronize(io.st), "hipStreamSynchronize");   // its previous polynomial has left the device
        const uint64_t len = in_lens[i] < N ? in_lens[i] : N;
        if (rc == MZK_OK && len) rc = hip_ok(hipMemcpyAsync(io.buf.p, data_mont[i], len * 32, hipMemcpyHostToDevice, io.st), "hipMemcpyAsync");
        if (rc == MZK_OK) {
            std::lock_guard<std::mutex> lk(cx_->lock);
            rc = ntt_dispatch(curve_id, io.buf.as<uint32_t>(), len, (int)log_n, inverse != 0, reinterpret_cast<const uint32_t*>(coset_offset_mont), 1, N, io.st);
        }
        if (rc == MZK_OK) rc = hip_ok(hipMemcpyAsync(data_mont[i], io.buf.p, N * 32, hipMemcpyDeviceToHost, io.st), "hipMemcpyAsync");
    }
    for (int k = 0; k < n_slots; k++) {
        const int32_t r2 = hip_ok(hipStreamSynchronize(cx_->io[slot[k].idx].st), "hipStreamSynchronize");
        if (rc == MZK_OK) rc = r2;
    }
    return rc;
}

int32_t mzk_ntt(int32_t curve_id, uint64_t* data_mont, uint64_t in_len, uint32_t log_n, int32_t inverse, const uint64_t* coset_offset_mont) {
    uint64_t* ptrs[1] = {data_mont};
    uint64_t lens[1] = {in_len};
    if (!data_mont) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    return mzk_ntt_batch(curve_id, 1, ptrs, lens, log_n, inverse, coset_offset_mont);
}

// ---- host-only: Keccak-f[1600] for the Merlin/STROBE transcript mirror (plonk/src/transcript/standard.rs) ----------------
int32_t mzk_chacha_blocks(const uint32_t key[8], uint64_t counter, uint32_t rounds, uint32_t n_blocks, uint32_t* out_words) {
    if (!key || !out_words || (rounds != 8 && rounds != 12 && rounds != 20)) { mzk::set_error("mzk_chacha_blocks: bad argument"); return MZK_ERR_INVALID_ARG; }
    auto rotl = [](uint32_t v, int n) { return (v << n) | (v >> (32 - n)); };
    for (uint32_t b = 0; b < n_blocks; b++, counter++) {
        uint32_t init[16] = {0x61707865u, 0x3320646Eu, 0x79622D32u, 0x6B206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                             (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
        uint32_t x[16];
        for (int i = 0; i < 16; i++) x[i] = init[i];
        auto qr = [&](int a, int bb, int c, int d) {
            x[a] += x[bb]; x[d] = rotl(x[d] ^ x[a], 16);
            x[c] += x[d]; x[bb] = rotl(x[bb] ^ x[c], 12);
            x[a] += x[bb]; x[d] = rotl(x[d] ^ x[a], 8);
            x[c] += x[d]; x[bb] = rotl(x[bb] ^ x[c], 7);
        };
        for (uint32_t r = 0; r < rounds / 2; r++) {
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
        }
        for (int i = 0; i < 16; i++) out_words[16 * b + i] = x[i] + init[i];
    }
    return MZK_OK;
}

int32_t mzk_keccak_f1600(uint8_t* state200) {
    if (!state200) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    static const uint64_t RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull, 0x000000000000808Bull,
                                    0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008Aull, 0x0000000000000088ull,
                                    0x0000000080008009ull, 0x000000008000000Aull, 0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull,
                                    0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
                                    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};     // lane x + 5y
    uint64_t a[25], b[25], c[5];
    std::memcpy(a, state200, 200);                                   // little-endian host
    auto rol = [](uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; };
    for (int r = 0; r < 24; r++) {
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) {
            const uint64_t d = c[(x + 4) % 5] ^ rol(c[(x + 1) % 5], 1);
            for (int y = 0; y < 5; y++) a[x + 5 * y] ^= d;
        }
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol(a[x + 5 * y], ROT[x + 5 * y]);
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= RC[r];
    }
    std::memcpy(state200, a, 200);
    return MZK_OK;
}

// ---- TurboPlonk quotient round ------------------------------------------------------------------------
int32_t mzk_plonk_pk_register(int32_t curve_id, uint32_t log_n, uint32_t num_wire_types, const uint64_t* selector_coeffs,
                              const uint64_t* sigma_coeffs, uint64_t poly_len, const uint64_t* k_mont, uint64_t* out_handle) {
    ENTER_CUR();
    if (num_wire_types != 5) { set_error("TurboPlonk proving key: 5 wire types (UltraPlonk: mzk_plonk_pk_register_ultra)"); return MZK_ERR_INVALID_ARG; }
    return plonk_pk_register(curve_id, (int)log_n, (int)num_wire_types, reinterpret_cast<const uint32_t*>(selector_coeffs),
                             reinterpret_cast<const uint32_t*>(sigma_coeffs), nullptr, poly_len, reinterpret_cast<const uint32_t*>(k_mont), nullptr, 0,
                             out_handle);
}
int32_t mzk_plonk_pk_register_ultra(int32_t curve_id, uint32_t log_n, const uint64_t* selector_coeffs, const uint64_t* sigma_coeffs,
                                    const uint64_t* table_coeffs, uint64_t poly_len, const uint64_t* k_mont, uint64_t* out_handle) {
    ENTER_CUR();
    if (!table_coeffs) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    return plonk_pk_register(curve_id, (int)log_n, 6, reinterpret_cast<const uint32_t*>(selector_coeffs), reinterpret_cast<const uint32_t*>(sigma_coeffs),
                             reinterpret_cast<const uint32_t*>(table_coeffs), poly_len, reinterpret_cast<const uint32_t*>(k_mont), nullptr, 0, out_handle);
}
int32_t mzk_plonk_pk_register_chunked(int32_t curve_id, uint32_t log_n, uint32_t num_wire_types, const uint64_t* selector_coeffs,
                                      const uint64_t* sigma_coeffs, const uint64_t* table_coeffs, uint64_t poly_len, const uint64_t* k_mont,
                                      const uint32_t* classes, uint32_t n_classes, uint64_t* out_handle) {
    ENTER_CUR();
    if (!classes || (num_wire_types == 6) != (table_coeffs != nullptr)) { set_error("bad argument"); return MZK_ERR_INVALID_ARG; }
    return plonk_pk_register(curve_id, (int)log_n, (int)num_wire_types, reinterpret_cast<const uint32_t*>(selector_coeffs),
                             reinterpret_cast<const uint32_t*>(sigma_coeffs), reinterpret_cast<const uint32_t*>(table_coeffs), poly_len,
                             reinterpret_cast<const uint32_t*>(k_mont), classes, n_classes, out_handle);
}
int32_t mzk_plonk_pk_hbm_bytes(uint64_t pk_handle, uint64_t* out_bytes) {
    ENTER_HANDLE(pk_handle);
    if (!out_bytes) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    *out_bytes = plonk_pk_bytes(pk_handle);
    return *out_bytes ? MZK_OK : MZK_ERR_BAD_HANDLE;
}
int32_t mzk_plonk_quotient_chunked_dev(uint64_t pk_handle, const void* d_polys, uint64_t in_stride, uint64_t in_len, const uint64_t* tau_mont,
                                       const uint64_t* alpha_mont, const uint64_t* beta_mont, const uint64_t* gamma_mont, void* d_out, void* stream) {
    return mzk_plonk_quotient_chunked_flags_dev(pk_handle, d_polys, in_stride, in_len, 0, tau_mont, alpha_mont, beta_mont, gamma_mont, d_out, stream);
}
int32_t mzk_plonk_quotient_chunked_flags_dev(uint64_t pk_handle, const void* d_polys, uint64_t in_stride, uint64_t in_len, uint32_t flags, const uint64_t* tau_mont,
                                             const uint64_t* alpha_mont, const uint64_t* beta_mont, const uint64_t* gamma_mont, void* d_out, void* stream) {
    ENTER_HANDLE(pk_handle);
    if (flags & ~(uint32_t)MZK_QUOTIENT_PI_ZERO) { set_error("unknown flag"); return MZK_ERR_INVALID_ARG; }
    return plonk_quotient_chunked_dev(pk_handle, reinterpret_cast<const uint32_t*>(d_polys), in_stride, in_len, flags, reinterpret_cast<const uint32_t*>(tau_mont),
                                      reinterpret_cast<const uint32_t*>(alpha_mont), reinterpret_cast<const uint32_t*>(beta_mont),
                                      reinterpret_cast<const uint32_t*>(gamma_mont), reinterpret_cast<uint32_t*>(d_out), (hipStream_t)stream);
}
int32_t mzk_plonk_quotient_combine_dev(int32_t curve_id, uint32_t log_n, const void* d_class_remainders, void* d_out, void* stream) {
    return mzk_plonk_quotient_combine_classes_dev(curve_id, log_n, nullptr, 0, d_class_remainders, d_out, stream);
}
int32_t mzk_plonk_quotient_combine_classes_dev(int32_t curve_id, uint32_t log_n, const uint32_t* classes, uint32_t n_classes, const void* d_class_remainders,
                                               void* d_out, void* stream) {
    ENTER_CUR();
    if (!d_class_remainders || !d_out || log_n > 27) { set_error("bad argument"); return MZK_ERR_INVALID_ARG; }
    return plonk_quotient_combine_dev(curve_id, (int)log_n, classes, n_classes, reinterpret_cast<const uint32_t*>(d_class_remainders), nullptr, 0,
                                      reinterpret_cast<uint32_t*>(d_out), (hipStream_t)stream);
}
int32_t mzk_plonk_quotient_top_dev(uint64_t pk_handle, const void* d_polys, uint64_t in_stride, uint64_t in_len, const uint64_t* alpha_mont,
                                   const uint64_t* beta_mont, const uint64_t* gamma_mont, void* d_top, uint32_t* out_n_top, void* stream) {
    ENTER_HANDLE(pk_handle);
    return plonk_quotient_top_dev(pk_handle, reinterpret_cast<const uint32_t*>(d_polys), in_stride, in_len, reinterpret_cast<const uint32_t*>(alpha_mont),
                                  reinterpret_cast<const uint32_t*>(beta_mont), reinterpret_cast<const uint32_t*>(gamma_mont),
                                  reinterpret_cast<uint32_t*>(d_top), out_n_top, (hipStream_t)stream);
}
int32_t mzk_plonk_quotient_combine_top_dev(int32_t curve_id, uint32_t log_n, const uint32_t* classes, uint32_t n_classes, const void* d_class_remainders,
                                           const void* d_top, uint32_t n_top, void* d_out, void* stream) {
    ENTER_CUR();
    if (!d_class_remainders || !d_out || !d_top || !classes || log_n > 27) { set_error("bad argument"); return MZK_ERR_INVALID_ARG; }
    return plonk_quotient_combine_dev(curve_id, (int)log_n, classes, n_classes, reinterpret_cast<const uint32_t*>(d_class_remainders),
                                      reinterpret_cast<const uint32_t*>(d_top), n_top, reinterpret_cast<uint32_t*>(d_out), (hipStream_t)stream);
}
int32_t mzk_plonk_pk_release(uint64_t pk_handle) {
    ENTER_HANDLE(pk_handle);
    return plonk_pk_release(pk_handle);
}
int32_t mzk_plonk_quotient_dev(uint64_t pk_handle, void* d_polys, uint64_t in_len, const uint64_t* alpha_mont, const uint64_t* beta_mont,
                               const uint64_t* gamma_mont, void* d_out, void* stream) {
    ENTER_HANDLE(pk_handle);
    if (plonk_pk_is_ultra(pk_handle) == 1) { set_error("UltraPlonk proving key: use mzk_plonk_quotient_ultra_dev"); return MZK_ERR_INVALID_ARG; }
    return plonk_quotient_dev(pk_handle, reinterpret_cast<uint32_t*>(d_polys), in_len, nullptr, reinterpret_cast<const uint32_t*>(alpha_mont),
                              reinterpret_cast<const uint32_t*>(beta_mont), reinterpret_cast<const uint32_t*>(gamma_mont),
                              reinterpret_cast<uint32_t*>(d_out), (hipStream_t)stream);
}
int32_t mzk_plonk_quotient_ultra_dev(uint64_t pk_handle, void* d_polys, uint64_t in_len, const uint64_t* tau_mont, const uint64_t* alpha_mont,
                                     const uint64_t* beta_mont, const uint64_t* gamma_mont, void* d_out, void* stream) {
    ENTER_HANDLE(pk_handle);
    if (plonk_pk_is_ultra(pk_handle) == 0) { set_error("TurboPlonk proving key: use mzk_plonk_quotient_dev"); return MZK_ERR_INVALID_ARG; }
    return plonk_quotient_dev(pk_handle, reinterpret_cast<uint32_t*>(d_polys), in_len, reinterpret_cast<const uint32_t*>(tau_mont),
                              reinterpret_cast<const uint32_t*>(alpha_mont), reinterpret_cast<const uint32_t*>(beta_mont),
                              reinterpret_cast<const uint32_t*>(gamma_mont), reinterpret_cast<uint32_t*>(d_out), (hipStream_t)stream);
}
int32_t mzk_plookup_sorted_vec_dev(uint64_t pk_handle, const void* d_wire_values, const uint64_t* tau_mont, void* d_merged_table, void* d_merged_lookup,
                                   void* d_sorted, void* stream) {
    ENTER_HANDLE(pk_handle);
    return plookup_sorted_vec_dev(pk_handle, reinterpret_cast<const uint32_t*>(d_wire_values), reinterpret_cast<const uint32_t*>(tau_mont),
                                  reinterpret_cast<uint32_t*>(d_merged_table), reinterpret_cast<uint32_t*>(d_merged_lookup),
                                  reinterpret_cast<uint32_t*>(d_sorted), (hipStream_t)stream);
}
int32_t mzk_plookup_product_dev(uint64_t pk_handle, const void* d_merged_table, const void* d_merged_lookup, const void* d_sorted, const uint64_t* beta_mont,
                                const uint64_t* gamma_mont, void* d_out, void* stream) {
    ENTER_HANDLE(pk_handle);
    return plookup_product_dev(pk_handle, reinterpret_cast<const uint32_t*>(d_merged_table), reinterpret_cast<const uint32_t*>(d_merged_lookup),
                               reinterpret_cast<const uint32_t*>(d_sorted), reinterpret_cast<const uint32_t*>(beta_mont),
                               reinterpret_cast<const uint32_t*>(gamma_mont), reinterpret_cast<uint32_t*>(d_out), (hipStream_t)stream);
}
int32_t mzk_plonk_quotient(uint64_t pk_handle, const uint64_t* polys, uint64_t in_len, const uint64_t* alpha_mont, const uint64_t* beta_mont,
                           const uint64_t* gamma_mont, uint64_t* out) {
    ENTER_HANDLE(pk_handle);
    const int log_n = plonk_pk_log_n(pk_handle), W = plonk_pk_wires(pk_handle);
    if (log_n < 0) { set_error("unknown proving-key handle"); return MZK_ERR_BAD_HANDLE; }
    if (plonk_pk_is_ultra(pk_handle) == 1) { set_error("UltraPlonk proving key: use mzk_plonk_quotient_ultra_dev"); return MZK_ERR_INVALID_ARG; }
    if (!polys || !out || in_len == 0 || in_len > (8ull << log_n)) { set_error("bad argument"); return MZK_ERR_INVALID_ARG; }
    const uint64_t n = 1ull << log_n, m = 8 * n;
    hipStream_t st = nullptr;
    MZK_TRY(g_ws.plonk_out.reserve(m * 32));
    uint32_t classes[8];
    const int ncl = plonk_pk_classes(pk_handle, classes);
    if (ncl > 0) {
        // a key holding residue classes of the quotient domain (the default of both hosts): the rows are read in place (stride in_len),
        // class remainders, then the inverse Vandermonde -- valid when the classes determine the quotient with one coefficient to spare
        // (deg t < ncl * n - 1: the caller's degree check must be able to fail for an unsatisfied witness)
        if ((uint64_t)W * (n + 1) + 2 >= (uint64_t)ncl * n - 1 || in_len > 2 * n) {
            set_error("chunked proving key: its classes do not determine the quotient (or a polynomial of degree >= 2n)");
            return MZK_ERR_INVALID_ARG;
        }
        MZK_TRY(g_ws.io.reserve((size_t)(W + 2) * in_len * 32));
        MZK_TRY(g_ws.link_tmp.reserve((size_t)ncl * n * 32));
        HIP_TRY(hipMemcpyAsync(g_ws.io.p, polys, (size_t)(W + 2) * in_len * 32, hipMemcpyHostToDevice, st));
        MZK_TRY(plonk_quotient_chunked_dev(pk_handle, g_ws.io.as<uint32_t>(), in_len, in_len, 0, nullptr, reinterpret_cast<const uint32_t*>(alpha_mont),
                                           reinterpret_cast<const uint32_t*>(beta_mont), reinterpret_cast<const uint32_t*>(gamma_mont),
                                           g_ws.link_tmp.as<uint32_t>(), st));
        MZK_TRY(plonk_quotient_combine_dev(plonk_pk_curve(pk_handle), log_n, classes, (uint32_t)ncl, g_ws.link_tmp.as<uint32_t>(), nullptr, 0,
                                           g_ws.plonk_out.as<uint32_t>(), st));
    } else {
        MZK_TRY(g_ws.plonk_polys.reserve((size_t)(W + 2) * m * 32));
        HIP_TRY(hipMemcpy2DAsync(g_ws.plonk_polys.p, m * 32, polys, in_len * 32, in_len * 32, W + 2, hipMemcpyHostToDevice, st));
        MZK_TRY(plonk_quotient_dev(pk_handle, g_ws.plonk_polys.as<uint32_t>(), in_len, nullptr, reinterpret_cast<const uint32_t*>(alpha_mont),
                                   reinterpret_cast<const uint32_t*>(beta_mont), reinterpret_cast<const uint32_t*>(gamma_mont),
                                   g_ws.plonk_out.as<uint32_t>(), st));
    }
    HIP_TRY(hipMemcpyAsync(out, g_ws.plonk_out.p, m * 32, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return MZK_OK;
}

int32_t mzk_plonk_perm_product_dev(uint64_t pk_handle, const void* d_wire_values, const uint64_t* beta_mont, const uint64_t* gamma_mont, void* d_out,
                                   void* stream) {
    ENTER_HANDLE(pk_handle);
    return plonk_perm_product_dev(pk_handle, reinterpret_cast<const uint32_t*>(d_wire_values), reinterpret_cast<const uint32_t*>(beta_mont),
                                  reinterpret_cast<const uint32_t*>(gamma_mont), reinterpret_cast<uint32_t*>(d_out), (hipStream_t)stream);
}
int32_t mzk_plonk_perm_product(uint64_t pk_handle, const uint64_t* wire_values, const uint64_t* beta_mont, const uint64_t* gamma_mont, uint64_t* out) {
    ENTER_HANDLE(pk_handle);
    const int log_n = plonk_pk_log_n(pk_handle), W = plonk_pk_wires(pk_handle);
    if (log_n < 0) { set_error("unknown proving-key handle"); return MZK_ERR_BAD_HANDLE; }
    if (!wire_values || !out) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    const uint64_t n = 1ull << log_n;
    hipStream_t st = nullptr;
    MZK_TRY(g_ws.plonk_polys.reserve((size_t)W * n * 32));
    MZK_TRY(g_ws.plonk_out.reserve(n * 32));
    HIP_TRY(hipMemcpyAsync(g_ws.plonk_polys.p, wire_values, (size_t)W * n * 32, hipMemcpyHostToDevice, st));
    MZK_TRY(plonk_perm_product_dev(pk_handle, g_ws.plonk_polys.as<uint32_t>(), reinterpret_cast<const uint32_t*>(beta_mont),
                                   reinterpret_cast<const uint32_t*>(gamma_mont), g_ws.plonk_out.as<uint32_t>(), st));
    HIP_TRY(hipMemcpyAsync(out, g_ws.plonk_out.p, n * 32, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return MZK_OK;
}

// ---- dense-polynomial primitives (prover rounds 4 and 5) ---------------------------------------------
int32_t mzk_poly_eval_dev(int32_t curve_id, const void* d_coeffs, uint64_t len, uint32_t batch, uint64_t batch_stride, const uint64_t* x_mont,
                          uint64_t* out_mont, void* stream) {
    ENTER_CUR();
    if ((!d_coeffs && len) || !x_mont || !out_mont || (batch > 1 && batch_stride < len) || batch > 65535) { set_error("bad argument"); return MZK_ERR_INVALID_ARG; }
    if (batch == 0) return MZK_OK;
    return poly_eval_dispatch(curve_id, reinterpret_cast<const uint32_t*>(d_coeffs), batch_stride, len, batch, reinterpret_cast<const uint32_t*>(x_mont),
                              reinterpret_cast<uint32_t*>(out_mont), (hipStream_t)stream);
}
int32_t mzk_poly_eval_many_dev(int32_t curve_id, uint32_t n_jobs, const void* const* d_coeffs, const uint64_t* lens, const uint32_t* batches,
                               const uint64_t* strides, const uint32_t* which_x, const uint64_t* x_mont, uint64_t* out_mont, void* stream) {
    ENTER_CUR();
    if (n_jobs == 0) return MZK_OK;
    if (!d_coeffs || !lens || !batches || !strides || !which_x || !x_mont || !out_mont || n_jobs > 64) { set_error("bad argument"); return MZK_ERR_INVALID_ARG; }
    EvalJob jobs[64];
    for (uint32_t j = 0; j < n_jobs; j++) {
        if ((!d_coeffs[j] && lens[j] && batches[j]) || (batches[j] > 1 && strides[j] < lens[j]) || batches[j] > 65535 || which_x[j] > 1) {
            set_error("bad argument");
            return MZK_ERR_INVALID_ARG;
        }
        jobs[j] = {reinterpret_cast<const uint32_t*>(d_coeffs[j]), lens[j], strides[j], batches[j], which_x[j]};
    }
    return poly_eval_many_dispatch(curve_id, jobs, n_jobs, reinterpret_cast<const uint32_t*>(x_mont), reinterpret_cast<uint32_t*>(out_mont), (hipStream_t)stream);
}
int32_t mzk_poly_lincomb_dev(int32_t curve_id, uint32_t n_terms, const void* const* d_polys, const uint64_t* lens, const uint64_t* scalars_mont,
                             void* d_out, uint64_t out_len, void* stream) {
    ENTER_CUR();
    if (n_terms && (!d_polys || !lens || !scalars_mont)) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    if (!d_out && out_len) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    return poly_lincomb_dispatch(curve_id, n_terms, reinterpret_cast<const uint32_t* const*>(d_polys), lens, reinterpret_cast<const uint32_t*>(scalars_mont),
                                 reinterpret_cast<uint32_t*>(d_out), out_len, (hipStream_t)stream);
}
int32_t mzk_poly_mask_dev(int32_t curve_id, uint32_t n_polys, void* const* d_polys, uint64_t n, uint32_t n_blinders, const uint64_t* blinders_mont, void* stream) {
    ENTER_CUR();
    if (n_polys && (!d_polys || !blinders_mont)) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    return poly_mask_dispatch(curve_id, n_polys, reinterpret_cast<uint32_t* const*>(d_polys), n, n_blinders, reinterpret_cast<const uint32_t*>(blinders_mont),
                              (hipStream_t)stream);
}
int32_t mzk_poly_split_quotient_dev(int32_t curve_id, const void* d_quot, uint64_t n, uint32_t n_parts, const uint64_t* blinders_mont, void* d_out, uint64_t out_stride, void* stream) {
    ENTER_CUR();
    if (!d_quot || !d_out || !blinders_mont) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    return poly_split_quotient_dispatch(curve_id, reinterpret_cast<const uint32_t*>(d_quot), n, n_parts, reinterpret_cast<const uint32_t*>(blinders_mont),
                                        reinterpret_cast<uint32_t*>(d_out), out_stride, (hipStream_t)stream);
}
int32_t mzk_plonk_gather_witness_dev(const void* d_witness, uint64_t n_vars, const void* d_wire_variables, uint64_t count, void* d_out, void* stream) {
    ENTER_CUR();
    if (count && (!d_witness || !d_wire_variables || !d_out)) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    if (n_vars >= (1ull << 32) || count >= (1ull << 40)) { set_error("witness too large"); return MZK_ERR_INVALID_ARG; }
    return wire_gather_dispatch(reinterpret_cast<const uint32_t*>(d_witness), n_vars, reinterpret_cast<const uint32_t*>(d_wire_variables), count,
                                reinterpret_cast<uint32_t*>(d_out), (hipStream_t)stream);
}
int32_t mzk_poly_div_linear_dev(int32_t curve_id, const void* d_poly, uint64_t len, const uint64_t* z_mont, void* d_out, void* stream) {
    ENTER_CUR();
    if ((!d_poly || !d_out) && len > 1) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    if (!z_mont) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    return poly_div_dispatch(curve_id, reinterpret_cast<const uint32_t*>(d_poly), len, reinterpret_cast<const uint32_t*>(z_mont),
                             reinterpret_cast<uint32_t*>(d_out), nullptr, (hipStream_t)stream);
}
int32_t mzk_poly_div_linear_rem_dev(int32_t curve_id, const void* d_poly, uint64_t len, const uint64_t* z_mont, void* d_out, void* d_rem, void* stream) {
    ENTER_CUR();
    if ((!d_poly || !d_out) && len > 1) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    if (!z_mont || !d_rem || (!d_poly && len)) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    return poly_div_dispatch(curve_id, reinterpret_cast<const uint32_t*>(d_poly), len, reinterpret_cast<const uint32_t*>(z_mont),
                             reinterpret_cast<uint32_t*>(d_out), reinterpret_cast<uint32_t*>(d_rem), (hipStream_t)stream);
}
int32_t mzk_poly_degree_dev(const void* d_poly, uint64_t len, uint64_t* d_out_len, void* stream) {
    ENTER_CUR();
    if ((!d_poly && len) || !d_out_len) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    if (len >= (1ull << 40)) { set_error("polynomial too long"); return MZK_ERR_INVALID_ARG; }
    return poly_degree_dispatch(reinterpret_cast<const uint32_t*>(d_poly), len, reinterpret_cast<unsigned long long*>(d_out_len), (hipStream_t)stream);
}
int32_t mzk_poly_div_roots_dev(int32_t curve_id, const void* d_poly, uint64_t len, uint32_t log_order, uint64_t first, uint64_t count, void* d_out,
                               void* stream) {
    ENTER_CUR();
    if ((!d_poly || !d_out) && len > count) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    return poly_div_roots_dispatch(curve_id, reinterpret_cast<const uint32_t*>(d_poly), len, log_order, first, count, reinterpret_cast<uint32_t*>(d_out),
                                   (hipStream_t)stream);
}

// ---- page-locked host memory for the host-pointer entry points ------------------------------------------
int32_t mzk_host_alloc(uint64_t bytes, void** out_ptr) {
    BIND_CUR();
    if (!out_ptr) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    HIP_TRY(hipHostMalloc(out_ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return MZK_OK;
}
int32_t mzk_host_free(void* ptr) {
    BIND_CUR();
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return MZK_OK;
}
int32_t mzk_host_register(void* ptr, uint64_t bytes) {
    BIND_CUR();
    if (!ptr || !bytes) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return MZK_OK;
}
int32_t mzk_host_unregister(void* ptr) {
    BIND_CUR();
    if (!ptr) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    HIP_TRY(hipHostUnregister(ptr));
    return MZK_OK;
}

// ---- device memory helpers --------------------------------------------------------------------------
int32_t mzk_dev_alloc(uint64_t bytes, void** out_dptr) {
    ENTER_CUR();
    if (!out_dptr) return MZK_ERR_INVALID_ARG;
    HIP_TRY(hipMalloc(out_dptr, bytes ? bytes : 4));
    return MZK_OK;
}
int32_t mzk_dev_free(void* dptr) {
    ENTER_CUR();
    HIP_TRY(hipFree(dptr));
    return MZK_OK;
}
int32_t mzk_dev_upload(void* dptr, const void* host, uint64_t bytes) {
    ENTER_CUR();
    HIP_TRY(hipMemcpy(dptr, host, bytes, hipMemcpyHostToDevice));
    return MZK_OK;
}
int32_t mzk_dev_download(void* host, const void* dptr, uint64_t bytes) {
    ENTER_CUR();
    HIP_TRY(hipMemcpy(host, dptr, bytes, hipMemcpyDeviceToHost));
    return MZK_OK;
}
// streams of the calling thread's device, for hosts that overlap transfers with kernels without HIP of their own
int32_t mzk_stream_create(void** out_stream) {
    BIND_CUR();
    if (!out_stream) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    hipStream_t st = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *out_stream = st;
    return MZK_OK;
}
// (internal to the library's own translation units: csrc/prover.hip) the context's stream for prover handle k
int32_t mzk_ctx_prover_stream(uint32_t k, void** out_stream) {
    ENTER_CUR();
    if (!out_stream) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    hipStream_t st = nullptr;
    MZK_TRY(ctx_prover_stream(k, &st));
    *out_stream = st;
    return MZK_OK;
}
int32_t mzk_stream_destroy(void* stream) {
    BIND_CUR();
    if (stream) HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return MZK_OK;
}
int32_t mzk_stream_sync(void* stream) {
    BIND_CUR();
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return MZK_OK;
}
int32_t mzk_stream_wait_stream(void* waiter, void* signaller) {
    BIND_CUR();
    hipEvent_t ev = nullptr;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, (hipStream_t)signaller);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)waiter, ev, 0);
    (void)hipEventDestroy(ev);                                      // released by the runtime once the recorded work has completed
    if (e != hipSuccess) { set_error(std::string("mzk_stream_wait_stream: ") + hipGetErrorString(e)); return MZK_ERR_HIP; }
    return MZK_OK;
}
int32_t mzk_dev_upload_async(void* dptr, const void* host, uint64_t bytes, void* stream) {
    BIND_CUR();
    if (bytes && (!dptr || !host)) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    if (bytes) HIP_TRY(hipMemcpyAsync(dptr, host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return MZK_OK;
}
int32_t mzk_dev_download_async(void* host, const void* dptr, uint64_t bytes, void* stream) {
    BIND_CUR();
    if (bytes && (!dptr || !host)) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    if (bytes) HIP_TRY(hipMemcpyAsync(host, dptr, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return MZK_OK;
}
int32_t mzk_dev_sync(void) {
    ENTER_CUR();
    HIP_TRY(hipDeviceSynchronize());
    return MZK_OK;
}
int32_t mzk_dev_copy(void* dst, const void* src, uint64_t bytes, void* stream) {
    ENTER_CUR();
    if (bytes && (!dst || !src)) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    if (bytes) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return MZK_OK;
}
int32_t mzk_dev_copy_peer(void* dst, int32_t dst_device, const void* src, int32_t src_device, uint64_t bytes, void* stream) {
    if (dst_device < 0 || dst_device >= MAX_CTX || src_device < 0 || src_device >= MAX_CTX || !g_ctx[dst_device].init || !g_ctx[src_device].init) {
        set_error("mzk_dev_copy_peer: both devices must have been initialised (mzk_init)");
        return MZK_ERR_NOT_INIT;
    }
    if (bytes && (!dst || !src)) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    Ctx* cx_ = &g_ctx[src_device];                                  // enqueued from the source device's side, on a stream of that device
    CtxBind bind_(cx_);
    MZK_TRY(bind_.rc);
    if (!bytes) return MZK_OK;
    const int pd = g_ctx[dst_device].device, ps = cx_->device;
    if (pd == ps) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));     // virtual devices of one card
    else HIP_TRY(hipMemcpyPeerAsync(dst, pd, src, ps, bytes, (hipStream_t)stream));                            // xGMI when peer access is on
    return MZK_OK;
}
int32_t mzk_dev_copy2d(void* dst, uint64_t dst_pitch, const void* src, uint64_t src_pitch, uint64_t width, uint64_t height, void* stream) {
    ENTER_CUR();
    if (width && height && (!dst || !src || width > dst_pitch || width > src_pitch)) { set_error("bad argument"); return MZK_ERR_INVALID_ARG; }
    if (width && height) HIP_TRY(hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width, height, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return MZK_OK;
}
int32_t mzk_dev_memset2d(void* dptr, uint64_t pitch, int32_t value, uint64_t width, uint64_t height, void* stream) {
    ENTER_CUR();
    if (width && height && (!dptr || width > pitch)) { set_error("bad argument"); return MZK_ERR_INVALID_ARG; }
    if (width && height) HIP_TRY(hipMemset2DAsync(dptr, pitch, value, width, height, (hipStream_t)stream));
    return MZK_OK;
}
int32_t mzk_dev_memset(void* dptr, int32_t value, uint64_t bytes, void* stream) {
    ENTER_CUR();
    if (bytes && !dptr) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    if (bytes) HIP_TRY(hipMemsetAsync(dptr, value, bytes, (hipStream_t)stream));
    return MZK_OK;
}

// ---- profiling ---------------------------------------------------------------------------------------
int32_t mzk_profile_enable(int32_t on) {
    g_prof = on != 0;
    return MZK_OK;
}
int32_t mzk_profile_get(const char* name, double* out_ms, uint64_t* out_count) {
    ENTER_CUR();
    if (!name || !out_ms || !out_count) return MZK_ERR_INVALID_ARG;
    double ms = 0;
    uint64_t cnt = 0;
    for (auto& r : cx_->prof_recs) {
        if (r.name != name) continue;
        HIP_TRY(hipEventSynchronize(r.b));
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, r.a, r.b));
        ms += t;
        cnt++;
    }
    *out_ms = ms;
    *out_count = cnt;
    return MZK_OK;
}
int32_t mzk_profile_reset(void) {
    std::lock_guard<std::mutex> lk0(g_ctx_lock);
    for (auto& cx : g_ctx) {
        if (!cx.init) continue;
        std::lock_guard<std::mutex> lk(cx.lock);
        for (auto& r : cx.prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        cx.prof_recs.clear();
    }
    return MZK_OK;
}
int32_t mzk_msm_set_precompute(int32_t on) {
    g_msm_precompute = on != 0;
    return MZK_OK;
}
int32_t mzk_srs_precompute(uint64_t srs_handle, uint32_t* out_window_bits, uint32_t* out_levels, uint64_t* out_table_bytes, double* out_build_ms) {
    ENTER_HANDLE(srs_handle);
    auto it = cx_->srs.find(srs_handle);
    if (it == cx_->srs.end()) { set_error("unknown SRS handle"); return MZK_ERR_BAD_HANDLE; }
    Srs& s = it->second;
    MZK_TRY(srs_build_pre(s, nullptr));
    const bool have = s.d_pre != nullptr && s.pre_c > 0;
    const uint64_t aff_bytes = (s.curve == MZK_CURVE_BLS12_381 ? 28u : 18u) * 4u;       // EcFx::AFF_WORDS: 2 x 14 / 2 x 9 limbs of 29 bits
    if (out_window_bits) *out_window_bits = have ? (uint32_t)s.pre_c : 0u;
    if (out_levels) *out_levels = have ? (uint32_t)s.pre_levels : 0u;
    if (out_table_bytes) *out_table_bytes = have ? (uint64_t)s.pre_levels * s.n * aff_bytes : 0u;
    if (out_build_ms) *out_build_ms = have ? s.pre_build_ms : 0.0;
    return MZK_OK;
}
int32_t mzk_msm_last_shape(uint32_t* out_window_bits, uint32_t* out_windows, uint32_t* out_buckets) {
    ENTER_CUR();
    if (out_window_bits) *out_window_bits = cx_->last_c;
    if (out_windows) *out_windows = cx_->last_w;
    if (out_buckets) *out_buckets = cx_->last_m;
    return MZK_OK;
}

}  // extern "C"
