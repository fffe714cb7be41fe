// internal.hpp -- state and helpers shared by the translation units of libmi355zk.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mzk.h"

// library-internal entry point (not in include/mzk.h, not exported): the calling thread's context hands a prover handle its stream
extern "C" int32_t mzk_ctx_prover_stream(uint32_t k, void** out_stream);

namespace mzk {

void set_error(const std::string& s);

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ::mzk::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));              \
            return _e == hipErrorOutOfMemory ? MZK_ERR_OOM : MZK_ERR_HIP;                     \
        }                                                                                     \
    } while (0)

#define MZK_TRY(expr)                \
    do {                             \
        int32_t _r = (expr);         \
        if (_r != MZK_OK) return _r; \
    } while (0)

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int32_t reserve(size_t bytes);
    void release();
    template <class T> T* as() { return reinterpret_cast<T*>(p); }
};

struct Workspace {
    DevBuf ntt_scratch, scalars, hist, offs, cursor, sorted, buckets, collect, io, misc, digits, long_desc, long_parts, plonk_polys, plonk_out, pre_cnt, pre_off, pre_ce, pre_cb, poly_tmp, split, link_tmp, occ;
    void* h_collect = nullptr;
    size_t h_collect_cap = 0;
    double heavy_frac = 0.125;          // msm.hip: share of the worst case the heavy buckets' level-1 sums are sized for (grows on overflow, with a re-run)
    hipEvent_t last_use = nullptr;
    void release();
    size_t bytes();
};

// every kernel launch of the library is counted (mzk_launch_count)
extern std::atomic<uint64_t> g_launches;
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, ...)                                  \
    do {                                                                     \
        ::mzk::g_launches.fetch_add(1, std::memory_order_relaxed);           \
        hipLaunchKernelGGLInternal((kernelName), __VA_ARGS__);               \
    } while (0)

// HIP-event timing of named regions (mzk_profile_*): one switch for the library, records per device context
extern std::atomic<bool> g_prof;                  // (switches shared by the device threads of one process: atomics)
struct ProfRec { std::string name; hipEvent_t a, b; };
struct ProfScope {
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t st;
    const char* name;
    ProfScope(const char* n, hipStream_t s);
    ~ProfScope();
};
extern std::atomic<bool> g_msm_precompute;

struct Srs {
    int curve;
    uint64_t n;
    uint32_t* d_xy;   // n * 2 * fq words, boundary form (what mzk_srs_download returns)
    uint32_t* d_int;  // internal reduced-radix table used by the MSM (29-bit limbs, R'-Montgomery form: ecx.cuh)
    uint32_t* d_pre = nullptr;  // [W][n] precomputed multiples 2^(c*w) P_i (msm_pre.cuh), built on first large MSM
    int pre_c = 0;              // window bits of d_pre; -1 = do not build
    int pre_levels = 0;         // W: levels of d_pre
    double pre_build_ms = 0;    // wall time of the build (pre_next_level launches, synchronised)
};
inline int fq_words(int curve) { return curve == MZK_CURVE_BLS12_381 ? 12 : 8; }

// ---- device contexts ---------------------------------------------------------------------------------
// One context per LOGICAL device: the HIP device it runs on, its lock, workspace, I/O slots, SRS registry; the NTT plan cache,
// the MSM sort streams and the proving keys are kept per context by their translation units (arrays indexed by Ctx::logical).
// A process may drive several devices, one host thread each (mzk_init(device) binds the calling thread; mzk_set_device rebinds
// it): calls on different contexts share nothing and run concurrently.  SRS / proving-key handles carry their context in the
// top 16 bits, so a handle-taking entry point needs no current device; entry points that take bare device pointers run on the
// calling thread's context (threads that never bound one use the first context initialised: the single-GPU case, Rayon workers
// included).  MZK_VIRTUAL_DEVICES=G maps logical devices 0..G-1 onto the physical ones round-robin -- G contexts on ONE card
// rehearse the multi-GPU code paths of a compiled host (mpc-jellyfish_amd/host/) on a one-GPU box.
constexpr int MAX_CTX = 16;
constexpr int IO_SLOTS = 4;
struct IoSlot {
    hipStream_t st = nullptr;
    DevBuf buf;
    bool busy = false;
};
struct Ctx {
    int logical = -1, device = -1;
    std::atomic<bool> init{false};                  // read without g_ctx_lock by the threads of other devices (handle look-ups)
    std::mutex lock;
    Workspace ws;
    std::vector<ProfRec> prof_recs;
    uint32_t last_c = 0, last_w = 0, last_m = 0;   // shape of the context's last MSM (mzk_msm_last_shape)
    std::map<uint64_t, Srs> srs;
    uint64_t next_handle = 1;
    IoSlot io[IO_SLOTS];
    std::mutex io_lock;
    std::condition_variable io_cv;
};
Ctx& cur();                                       // the context the running entry point is bound to (thread-local)
#define g_ws (::mzk::cur().ws)
// a call on `st` must not start before the previous user of the context's shared workspace is done
int32_t ws_acquire(hipStream_t st);
int32_t ws_release(hipStream_t st);
inline uint64_t handle_make(int logical, uint64_t counter) { return ((uint64_t)(logical + 1) << 48) | counter; }
inline int handle_ctx(uint64_t h) { return (int)(h >> 48) - 1; }

// ntt.hip
// scale: 0 = boundary form in and out; 1 = output left in the internal form x * R' (R' = 2^261 = 32 R) of the quotient kernels;
// 2 = input in that form, output back in the boundary form (the factor rides in the final pass's multiplication)
// d_src (nullable): out of place -- the first pass reads the batch there (src_stride elements apart) and d_data only receives the result;
// d_patch (with d_src): 4 elements per batch entry that replace the input indices 0..3 (ntt_fx.cuh)
int32_t ntt_dispatch(int curve, uint32_t* d_data, uint64_t in_len, int log_n, bool inverse, const uint32_t* coset,
                     uint32_t batch, uint64_t stride, hipStream_t st, int scale = 0, const uint32_t* d_src = nullptr, uint64_t src_stride = 0,
                     const uint32_t* d_patch = nullptr, int skip_batch = -1 /* a batch entry that is not transformed */);
// The same transform over n_classes different cosets in ONE launch per pass (ntt_fx.cuh nttx_pass_classes_kernel): d_data holds
// n_classes * rows polynomials class-major, `stride` elements apart; with d_src every class reads the SAME rows of d_src (src_stride
// apart; d_patch class-major, 4 elements per entry).  2^10 <= N, n_classes <= 8; small transforms (the radix-2 pass form).
int32_t ntt_classes_dispatch(int curve, uint32_t* d_data, uint64_t in_len, int log_n, bool inverse, const uint32_t* const* cosets, int n_classes, uint32_t rows,
                             uint64_t stride, hipStream_t st, int scale, const uint32_t* d_src, uint64_t src_stride, const uint32_t* d_patch, int skip_batch);
void ntt_release_plans();
void msm_release_streams();
int32_t ctx_prover_stream(unsigned k, hipStream_t* out);      // msm.hip: a context-owned stream for a prover handle (never destroyed by the handle)
// msm.hip
int32_t msm_dispatch(const Srs& s, uint64_t base_offset, const uint32_t* d_scalars, uint64_t n, int is_mont, uint32_t* out, hipStream_t st);
int32_t msm_batch_dispatch(const Srs& s, uint32_t n_polys, const uint32_t* const* d_scalars, const uint64_t* lens, const uint64_t* base_offsets,
                           int is_mont, uint32_t* out_xyz, hipStream_t st);
int32_t srs_build_internal(Srs& s, hipStream_t st);
int32_t srs_build_pre(Srs& s, hipStream_t st);
int32_t srs_generate_dispatch(int curve, const uint32_t* beta_canon, const uint32_t* g_xy_mont /* NULL: the standard generator */, uint64_t n, uint32_t* d_out);
int32_t srs_lagrange_from_points_dispatch(int curve, const uint32_t* d_xy, int log_n, uint32_t n_extra, uint32_t* d_out);
int32_t srs_lagrange_generate_dispatch(int curve, const uint32_t* beta_canon, const uint32_t* g_xy_mont /* nullable */, int log_n, uint32_t n_extra, uint32_t* d_out);
void jac_to_affine_host_dispatch(int curve, const uint64_t* xyz, uint64_t n, uint64_t* xy);
void jac_sum_host_dispatch(int curve, const uint64_t* xyz, uint64_t n, uint64_t* out);

// poly.hip
int32_t poly_eval_dispatch(int curve, const uint32_t* d_coeffs, uint64_t stride, uint64_t len, uint32_t batch, const uint32_t* x_mont, uint32_t* out_host,
                           hipStream_t st);
struct EvalJob { const uint32_t* d; uint64_t len, stride; uint32_t batch, which_x; };
int32_t poly_eval_many_dispatch(int curve, const EvalJob* jobs, uint32_t n_jobs, const uint32_t* x_mont /* 2 x 8 words */, uint32_t* out_host, hipStream_t st);
int32_t poly_div_dispatch(int curve, const uint32_t* d_poly, uint64_t len, const uint32_t* z_mont, uint32_t* d_out, uint32_t* d_rem /* nullable: p(z) */, hipStream_t st);
int32_t poly_div_roots_dispatch(int curve, const uint32_t* d_poly, uint64_t len, uint32_t log_order, uint64_t first, uint64_t count, uint32_t* d_out,
                                hipStream_t st);
int32_t poly_lincomb_dispatch(int curve, uint32_t n_terms, const uint32_t* const* d_polys, const uint64_t* lens, const uint32_t* scalars, uint32_t* d_out,
                              uint64_t out_len, hipStream_t st);
int32_t poly_degree_dispatch(const uint32_t* d_poly, uint64_t len, unsigned long long* d_out, hipStream_t st);
int32_t wire_gather_dispatch(const uint32_t* d_witness, uint64_t n_vars, const uint32_t* d_vars, uint64_t count, uint32_t* d_out, hipStream_t st);
int32_t poly_split_quotient_dispatch(int curve, const uint32_t* d_q, uint64_t n, uint32_t W, const uint32_t* blind_mont, uint32_t* d_out, uint64_t stride, hipStream_t st);
int32_t poly_mask_dispatch(int curve, uint32_t n_rows, uint32_t* const* d_rows, uint64_t n, uint32_t n_blind, const uint32_t* blind_mont, hipStream_t st);
// plonk.hip
int32_t plonk_pk_register(int curve, int log_n, int W, const uint32_t* sel, const uint32_t* sig, const uint32_t* tab /* NULL: TurboPlonk */,
                          uint64_t poly_len, const uint32_t* k_mont, const uint32_t* classes /* NULL: whole domain */, uint32_t n_classes,
                          uint64_t* out_handle);
int32_t plonk_quotient_chunked_dev(uint64_t handle, const uint32_t* d_polys, uint64_t in_stride, uint64_t in_len, uint32_t flags, const uint32_t* tau,
                                   const uint32_t* alpha, const uint32_t* beta, const uint32_t* gamma, uint32_t* d_out, hipStream_t st);
int32_t plonk_quotient_top_dev(uint64_t handle, const uint32_t* d_polys, uint64_t in_stride, uint64_t in_len, const uint32_t* alpha, const uint32_t* beta,
                               const uint32_t* gamma, uint32_t* d_top, uint32_t* out_n_top, hipStream_t st);
int32_t plonk_quotient_combine_dev(int curve, int log_n, const uint32_t* classes /* NULL: all 8 */, uint32_t n_classes, const uint32_t* d_r,
                                   const uint32_t* d_top /* nullable */, uint32_t n_top, uint32_t* d_out, hipStream_t st);
int32_t plonk_pk_release(uint64_t handle);
void plonk_release_all();
int32_t plonk_quotient_dev(uint64_t handle, uint32_t* d_polys, uint64_t in_len, const uint32_t* tau /* NULL: TurboPlonk */, const uint32_t* alpha,
                           const uint32_t* beta, const uint32_t* gamma, uint32_t* d_out, hipStream_t st);
int32_t plookup_sorted_vec_dev(uint64_t handle, const uint32_t* d_wires, const uint32_t* tau, uint32_t* d_table, uint32_t* d_lookup, uint32_t* d_sorted,
                               hipStream_t st);
int32_t plookup_product_dev(uint64_t handle, const uint32_t* d_table, const uint32_t* d_lookup, const uint32_t* d_sorted, const uint32_t* beta,
                            const uint32_t* gamma, uint32_t* d_out, hipStream_t st);
int plonk_pk_is_ultra(uint64_t handle);
int32_t plonk_perm_product_dev(uint64_t handle, const uint32_t* d_wires, const uint32_t* beta, const uint32_t* gamma, uint32_t* d_out, hipStream_t st);
int plonk_pk_log_n(uint64_t handle);
int plonk_pk_curve(uint64_t handle);
int plonk_pk_classes(uint64_t handle, uint32_t* out);
int plonk_pk_wires(uint64_t handle);
uint64_t plonk_pk_bytes(uint64_t handle);

}  // namespace mzk
