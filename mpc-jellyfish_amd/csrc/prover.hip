// prover.hip -- the prover's rounds behind the C ABI (mzk_prover_*, include/mzk.h): the bodies of Prover::run_1st_round ..
// compute_opening_proofs (plonk/src/proof_system/prover.rs:72-419) over device-resident vectors, for ANY circuit (public inputs,
// every gate type, copy constraints, lookups), one or several instances, one or several ranks.  No kernels in this translation
// unit: it sequences the library's own entry points (mzk_ntt_dev, mzk_msm_batch_dev, mzk_plonk_*, mzk_poly_*) on the null stream of
// the prover's device and keeps the scalar bookkeeping of the linearisation polynomial (prover.rs:302-358, 963-1112) on the host in
// 64-bit-limb field arithmetic (hostfp.hpp).  Transcript, rng and `Proof` assembly stay with the caller, as in snark.rs:263-431.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "hostfp.hpp"
#include "internal.hpp"

namespace mzk {
namespace {

using h64::from_u64;
using h64::inv;
using h64::pow_u64;
using h64::root_of_unity;

struct Fail { int32_t rc; };                                           // a library call failed: its code travels up to the entry point
inline void ck(int32_t rc) { if (rc != MZK_OK) throw Fail{rc}; }
[[noreturn]] inline void fail(int32_t rc, const std::string& msg) { set_error(msg); throw Fail{rc}; }

constexpr size_t EL = 32;                                              // bytes per scalar-field element

struct Buf {                                                           // device memory of the prover's device
    void* p = nullptr;
    size_t elems = 0;
    Buf() = default;
    Buf(const Buf&) = delete;
    Buf& operator=(const Buf&) = delete;
    ~Buf() { if (p) (void)mzk_dev_free(p); }
    void alloc(size_t n_elems) {
        if (p) (void)mzk_dev_free(p);
        p = nullptr;
        elems = n_elems;
        ck(mzk_dev_alloc((n_elems ? n_elems : 1) * EL, &p));
    }
    void* at(size_t idx) const { return static_cast<uint8_t*>(p) + idx * EL; }
    size_t bytes() const { return p ? (elems ? elems : 1) * EL : 0; }
};
struct Pinned {                                                        // page-locked staging memory
    void* p = nullptr;
    size_t cap = 0;
    ~Pinned() { if (p) (void)mzk_host_free(p); }
    void* reserve(size_t bytes) {
        if (bytes > cap) {
            if (p) (void)mzk_host_free(p);
            p = nullptr;
            cap = 0;
            ck(mzk_host_alloc(bytes, &p));
            cap = bytes;
        }
        return p;
    }
};

inline std::pair<uint64_t, uint64_t> shard_range(uint64_t n, int rank, int world) {      // contiguous share of n points; the first n % world ranks take one more
    const uint64_t base = n / world, extra = n % world, r = (uint64_t)rank;
    const uint64_t lo = r * base + std::min(r, extra);
    return {lo, lo + base + (r < extra ? 1 : 0)};
}
inline std::vector<uint32_t> class_range(int rank, int world, uint32_t n_classes) {      // residue classes of the quotient domain owned by `rank`
    const uint32_t per = (n_classes + world - 1) / world;
    std::vector<uint32_t> out;
    for (uint32_t k = std::min<uint32_t>(rank * per, n_classes); k < std::min<uint32_t>((rank + 1) * per, n_classes); k++) out.push_back(k);
    return out;
}

// indices into the Plookup evaluations (declaration order of PlookupEvaluations, structs.rs:496-541)
enum PlookupEval { RANGE_TABLE, KEY_TABLE, TABLE_DOM_SEP, Q_DOM_SEP, H_1, Q_LOOKUP, PROD_NEXT, RANGE_TABLE_NEXT, KEY_TABLE_NEXT, TABLE_DOM_SEP_NEXT,
                   H_1_NEXT, H_2_NEXT, Q_LOOKUP_NEXT, W_3_NEXT, W_4_NEXT, N_PLOOKUP_EVALS };

enum Stage { CREATED = 0, R1 = 10, R1_5 = 15, R2 = 20, R2_5 = 25, R3 = 30, R4 = 40 };

struct ProverBase {
    int device = 0, curve = 0, log_n = 0, W = 5, nsel = 13, stage = CREATED;
    bool ultra = false;
    uint64_t n = 0;
    bool profile = false;
    // Every handle runs its rounds on a stream of its own (non-blocking): two provers of one card -- two device contexts, MZK_VIRTUAL_DEVICES,
    // or two handles of one context -- then only meet where they share hardware, and the latency-bound tails of one proof (narrow
    // reduction levels, host Horner, transcript) run under the kernels of the other.  MZK_PROVER_NULL_STREAM=1: the device's null
    // stream, as until round 4 (A/B).  What the caller hands over in device memory must be complete on the null stream (or synchronised)
    // when a round is called: round 1 makes S wait for it; every round returns with S idle as far as its outputs are concerned.
    void* S = nullptr;
    std::map<std::string, double> timings_ms;
    void sync_stream() { if (mzk_stream_sync(S) != MZK_OK) throw Fail{MZK_ERR_HIP}; }
    virtual ~ProverBase() {}
    virtual void vk_commitments(uint64_t* out_xy, uint64_t* out_plookup_xy) = 0;
    virtual void set_wire_variables(const uint32_t* vars, uint64_t n_vars) = 0;
    virtual void round1(int kind, const void* witness, uint64_t witness_len, const uint64_t* pi_rows, const uint64_t* pi, uint64_t n_pi,
                        const uint64_t* blinders, uint64_t* out) = 0;
    virtual void round1_5(const uint64_t* tau, const uint64_t* blinders, uint64_t* out) = 0;
    virtual void round2(const uint64_t* beta, const uint64_t* gamma, const uint64_t* blinders, uint64_t* out) = 0;
    virtual void round2_5(const uint64_t* blinders, uint64_t* out) = 0;
    virtual void round3(const std::vector<ProverBase*>& inst, const uint64_t* alpha, const uint64_t* blinders, uint64_t* out) = 0;
    virtual void round4(const uint64_t* zeta, uint64_t* out_evals) = 0;
    virtual void round5(const std::vector<ProverBase*>& inst, const uint64_t* v, uint64_t* out) = 0;
    virtual void exchange_buffer(void** out_p, uint64_t* out_bytes) = 0;
    virtual void set_peer_buffers(void* const* ptrs, const int32_t* devices) = 0;
    virtual void poly_dev(uint32_t which, const void** out_p, uint64_t* out_len) = 0;
    virtual void hbm_bytes(uint64_t* fixed_b, uint64_t* pk_b, uint64_t* ws_b) = 0;
};

template <class FrP, int CURVE>
struct ProverT final : ProverBase {
    using Fr = Fp64<FrP>;
    static constexpr int QL = CURVE == MZK_CURVE_BLS12_381 ? 6 : 4;   // u64 limbs of Fq
    static constexpr size_t PT = 2 * QL;                               // u64 words of an affine point

    int rows = 0;
    uint64_t m = 0;
    std::vector<Fr> k;
    uint64_t srs = 0, srs_lagrange = 0, pk = 0;
    // several ranks: this prover commits over the SRS points [lo, hi) of every polynomial (one fixed partition of the
    // n + 3 powers), owns the residue classes `own` of the quotient domain, and runs rounds 4-5 on its coefficient range
    int rank = 0, world = 1;
    mzk_comm comm{};
    uint64_t lo = 0, hi = 0;
    uint64_t key_first = 0;                                            // SRS index of the commit key's first point: lo when the key is this rank's slice
    Buf fixed;                                                         // (nsel + W [+ 4]) x n coefficient forms
    Buf slab, quot, coeff, split, lin, batch, opening, shifted, hh, table, lookup, sorted, tmp, deg, rem, wv, wit, top, vals_ext, qsum, vars;
    Pinned stage_pi;
    Pinned stage_back;                                                 // 64 B: the words a round reads back AFTER its commitments (quotient degree, batch value at zeta)
                                                                       // are copied here asynchronously BEFORE them: no second device round trip at the end of the round
    uint64_t n_vars = 0;
    bool use_top = false;                                              // W classes + mzk_plonk_quotient_top_dev
    std::vector<uint32_t> classes, own;                                // the classes that determine the quotient; this rank's share of them
    std::vector<void*> peer_rem;
    std::vector<int32_t> peer_dev;
    void* copy_stream = nullptr;
    bool one_ready = false;
    Fr w_n, gen;

    // what Oracles + Challenges carry for the proof in flight
    struct State {
        const void* wire_values = nullptr;                             // W x n wire evaluations on this device
        bool pi_zero = true;
        Fr tau, beta, gamma, alpha, zeta;
        std::vector<Fr> wires_evals, wire_sigma_evals, plookup_evals;
        Fr perm_next_eval, pi_eval;
        std::vector<uint64_t> split_len;                               // round 3 (first instance)
        std::vector<Fr> bases;                                         // alpha_base per instance (first instance)
        Fr batch_at_zeta;
    } st;

    struct Tick {
        ProverBase& P; std::chrono::steady_clock::time_point t0;
        explicit Tick(ProverBase& p) : P(p) { reset(); }
        void reset() { if (P.profile) { (void)mzk_stream_sync(P.S); t0 = std::chrono::steady_clock::now(); } }
        void mark(const char* name) {
            if (!P.profile) return;
            (void)mzk_stream_sync(P.S);
            P.timings_ms[name] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            reset();
        }
    };

    int rowZ() const { return W; }
    int rowPI() const { return W + 1; }
    int rowH1() const { return W + 2; }
    int rowPL() const { return W + 4; }
    // one slab: rows 0..W-1 wires, W z, W+1 public input (, h_1, h_2, Plookup product), n + 3 coefficient slots each; the rounds read
    // their polynomials from here from round 1 to the openings
    void* row(int r) const { return slab.at((size_t)r * (n + 3)); }
    void* fix(int r) const { return fixed.at((size_t)r * n); }
    static Fr load(const uint64_t* p) { Fr v; std::memcpy(v.l, p, EL); return v; }

    ProverT(int log_n_, int W_, const uint64_t* sel, const uint64_t* sig, const uint64_t* tab, uint64_t poly_len, const uint64_t* k_mont,
        uint64_t commit_key, uint64_t lagrange_key, const mzk_comm* cm);   // prover_setup.inc
    ~ProverT() override;   // prover_setup.inc
    void hbm_bytes(uint64_t* fixed_b, uint64_t* pk_b, uint64_t* ws_b) override;   // prover_setup.inc

    // ---- commitments ---------------------------------------------------------------------------------------------------------
    std::vector<uint64_t> msm_partials(const std::vector<const void*>& polys, const std::vector<uint64_t>& lens, uint64_t a, uint64_t b,
        uint64_t key = 0);   // prover_setup.inc
    void combine_partials(const std::vector<uint64_t>& part, uint64_t* out_xy);   // prover_setup.inc
    // UnivariateKzgPCS::batch_commit (mod.rs:119-131) on device-resident coefficient vectors; over several ranks every MSM is
    // sharded by point range (this rank: [lo, hi))
    void commit(const std::vector<const void*>& polys, const std::vector<uint64_t>& lens, uint64_t* out_xy, uint64_t key = 0) {
        combine_partials(msm_partials(polys, lens, lo, hi, key), out_xy);
    }
    void commit_slices(const std::vector<const void*>& slices, const std::vector<uint64_t>& lens, uint64_t* out_xy);   // prover_setup.inc
    void vk_commitments(uint64_t* out_xy, uint64_t* out_plookup_xy) override;   // prover_setup.inc

    // ---- small helpers ---------------------------------------------------------------------------------------------------------
    std::vector<Fr> evaluate(const void* d, uint64_t len, uint32_t batch_n, uint64_t stride, const Fr& x);   // prover_setup.inc
    // evaluations of one round, collected and finished together.  Over several ranks every rank evaluates its coefficient range
    // [lo, hi) of each polynomial -- sum_{j in range} c_j x^j = x^lo * (the range read as a polynomial of its own) -- and ONE
    // all-gather of the partial values (32 bytes each) at the end of the round gives every rank all the sums.
    struct EvalBatch {                                                 // (at most two distinct points per batch: zeta and zeta * omega)
        ProverT& P;
        std::vector<Fr> vals;
        std::vector<Fr> scale;                                         // per value: x^lo of a ranged job (ranks > 1), one otherwise
        std::vector<const void*> ptrs;
        std::vector<uint64_t> lens, strides;
        std::vector<uint32_t> batches, which;
        Fr xs[2];
        int n_x = 0;
        explicit EvalBatch(ProverT& p) : P(p) {}
        uint32_t point(const Fr& x) {
            for (int q = 0; q < n_x; q++) if (xs[q] == x) return (uint32_t)q;
            if (n_x == 2) fail(MZK_ERR_INVALID_ARG, "EvalBatch: more than two points");
            xs[n_x] = x;
            return (uint32_t)n_x++;
        }
        size_t add(const void* d, uint64_t len, uint32_t batch_n, uint64_t stride, const Fr& x) {
            const size_t at = scale.size();
            uint64_t a = 0, b = len;
            if (P.world > 1) { a = std::min(P.lo, len); b = std::min(P.hi, len); }
            const Fr xa = a ? pow_u64(x, a) : Fr::one();
            ptrs.push_back(static_cast<const uint8_t*>(d) + a * EL);
            lens.push_back(b > a ? b - a : 0);
            strides.push_back(stride);
            batches.push_back(batch_n);
            which.push_back(point(x));
            for (uint32_t i = 0; i < batch_n; i++) scale.push_back(xa);
            return at;
        }
        void finish() {                                                // ONE library call and one wait for the whole round
            vals.assign(scale.size(), Fr::zero());
            if (n_x == 1) xs[1] = xs[0];
            if (!scale.empty())
                ck(mzk_poly_eval_many_dev(CURVE, (uint32_t)ptrs.size(), ptrs.data(), lens.data(), batches.data(), strides.data(), which.data(), xs[0].l,
                                          reinterpret_cast<uint64_t*>(vals.data()), P.S));
            if (P.world == 1) return;
            const size_t cnt = vals.size();
            for (size_t i = 0; i < cnt; i++) vals[i] = vals[i] * scale[i];
            std::vector<Fr> all((size_t)P.world * cnt);
            if (P.comm.all_gather(P.comm.ctx, vals.data(), cnt * sizeof(Fr), all.data())) fail(MZK_ERR_INVALID_ARG, "mzk_comm.all_gather failed");
            for (size_t i = 0; i < cnt; i++) {
                Fr sum = Fr::zero();
                for (int g = 0; g < P.world; g++) sum = sum + all[(size_t)g * cnt + i];
                vals[i] = sum;
            }
        }
    };
    struct Term { Fr s; const void* p; uint64_t len; };
    void lincomb(const std::vector<Term>& terms, void* out, uint64_t out_len);   // prover_setup.inc
    void lincomb_many(const std::vector<Term>& terms, void* out, uint64_t out_len);   // prover_setup.inc
    void mask(const std::vector<int>& slab_rows, const uint64_t* blinders, uint32_t n_blind);   // prover_setup.inc
    const void* one_dev();   // prover_setup.inc
    // a scalar into device memory as a kernel argument (times the resident one): no host-to-device copy, hence no stream synchronisation
    void put_scalar(const Fr& v, void* d) { lincomb({{v, one_dev(), 1}}, d, 1); }
    void* rem_dev() const { return tmp.at(2); }                        // where the opening division leaves the batch polynomial's value at zeta
    Fr download_fr(const void* d);   // prover_setup.inc
    void need(int at_least, int below, const char* what) const {
        if (stage < at_least || stage >= below) fail(MZK_ERR_STATE, std::string("prover rounds out of order: ") + what);
    }

    // ---- round 1 (prover.rs:72-87; constraint_system.rs:1225-1259) -------------------------------------------------------------
    void set_wire_variables(const uint32_t* v, uint64_t nv) override;   // prover_setup.inc
    void public_input_row(const uint64_t* pi_rows, const uint64_t* pi, uint64_t n_pi);   // prover_round1.inc
    void round1(int kind, const void* witness, uint64_t witness_len, const uint64_t* pi_rows, const uint64_t* pi, uint64_t n_pi,
        const uint64_t* blinders, uint64_t* out) override;   // prover_round1.inc
    // ---- round 1.5 (prover.rs:89-118), UltraPlonk only --------------------------------------------------------------------------
    void round1_5(const uint64_t* tau, const uint64_t* blinders, uint64_t* out) override;   // prover_round1.inc
    // ---- round 2 (prover.rs:125-141) -------------------------------------------------------------------------------------------
    void round2(const uint64_t* beta, const uint64_t* gamma, const uint64_t* blinders, uint64_t* out) override;   // prover_round2.inc
    // ---- round 2.5 (prover.rs:143-183), UltraPlonk only --------------------------------------------------------------------------
    void round2_5(const uint64_t* blinders, uint64_t* out) override;   // prover_round2.inc
    // ---- round 3 (prover.rs:192-209, 512-673, 902-960) ----------------------------------------------------------------------------
    void quotient(const Fr& alpha);   // prover_round3.inc
    void split_quotient(const void* q, const uint64_t* b_quot);   // prover_round3.inc
    void round3(const std::vector<ProverBase*>& inst, const uint64_t* alpha_p, const uint64_t* blinders, uint64_t* out) override;   // prover_round3.inc
    // ---- round 4: compute_evaluations / compute_plookup_evaluations (prover.rs:216-299) -------------------------------------------
    void round4(const uint64_t* zeta_p, uint64_t* out_evals) override;   // prover_round4.inc
    // ---- round 5 ---------------------------------------------------------------------------------------------------------------
    std::vector<Term> lin_poly_terms(const Fr& alpha_base) const;   // prover_round5.inc
    Fr lin_poly_constant(const Fr& alpha_base) const;   // prover_round5.inc
    void opened_evals(std::vector<Fr>& out) const;   // prover_round5.inc
    std::vector<Term> quotient_lin_terms(const Fr& zeta) const;   // prover_round5.inc
    void open_lists(std::vector<Term>& open_polys, std::vector<Term>& shifted_polys) const;   // prover_round5.inc
    void batched_witness(const std::vector<Term>& polys, const Fr& v, const Fr& point, Buf& out, void* d_rem = nullptr);   // prover_round5.inc
    void openings_ranged(const std::vector<Term>& lin_terms, const std::vector<Term>& open_polys, const std::vector<Term>& shifted_polys,
        const Fr& v, const Fr& zeta, Tick& tick, uint64_t* out);   // prover_round5.inc
    void round5(const std::vector<ProverBase*>& inst, const uint64_t* v_p, uint64_t* out) override;   // prover_round5.inc

    void exchange_buffer(void** out_p, uint64_t* out_bytes) override {
        if (out_p) *out_p = rem.p;
        if (out_bytes) *out_bytes = classes.size() * n * EL;
    }
    void set_peer_buffers(void* const* ptrs, const int32_t* devices) override {
        peer_rem.assign(ptrs, ptrs + world);
        peer_dev.assign(devices, devices + world);
    }
    void poly_dev(uint32_t which, const void** out_p, uint64_t* out_len) override;   // prover_setup.inc
};

#include "prover_setup.inc"
#include "prover_round1.inc"
#include "prover_round2.inc"
#include "prover_round3.inc"
#include "prover_round4.inc"
#include "prover_round5.inc"


// ---- registry --------------------------------------------------------------------------------------------------------------------
std::mutex g_reg_lock;
std::map<uint64_t, std::shared_ptr<ProverBase>> g_provers;
std::atomic<uint64_t> g_next_prover{1};

std::shared_ptr<ProverBase> find(uint64_t h) {
    std::lock_guard<std::mutex> lk(g_reg_lock);
    auto it = g_provers.find(h);
    if (it == g_provers.end()) { set_error("unknown prover handle"); return nullptr; }
    return it->second;
}

// the calling thread works on the prover's device for the duration of an entry point
struct DeviceGuard {
    int32_t prev = -1, rc = MZK_OK;
    bool switched = false;
    explicit DeviceGuard(int device) {
        if (mzk_get_device(&prev) != MZK_OK) prev = -1;
        if (prev != device) { rc = mzk_set_device(device); switched = rc == MZK_OK && prev >= 0; }
    }
    ~DeviceGuard() { if (switched) (void)mzk_set_device(prev); }
};

template <class F>
int32_t guarded(ProverBase& p, F&& f) {
    DeviceGuard g(p.device);
    if (g.rc != MZK_OK) return g.rc;
    try {
        f();
    } catch (const Fail& e) {
        return e.rc;
    } catch (const std::bad_alloc&) {
        set_error("out of host memory");
        return MZK_ERR_OOM;
    } catch (const std::exception& e) {
        set_error(e.what());
        return MZK_ERR_INVALID_ARG;
    }
    return MZK_OK;
}

int32_t collect(const uint64_t* handles, uint32_t cnt, std::vector<std::shared_ptr<ProverBase>>& keep, std::vector<ProverBase*>& inst) {
    if (!handles || cnt == 0) { set_error("zero number of circuits/proving keys"); return MZK_ERR_INVALID_ARG; }
    for (uint32_t i = 0; i < cnt; i++) {
        auto p = find(handles[i]);
        if (!p) return MZK_ERR_BAD_HANDLE;
        for (uint32_t j = 0; j < i; j++)
            if (handles[j] == handles[i]) { set_error("one prover handle per instance: the device workspace belongs to the handle"); return MZK_ERR_INVALID_ARG; }
        if (i && (p->curve != keep[0]->curve || p->n != keep[0]->n || p->device != keep[0]->device)) { set_error("instances of one proof share curve, domain size and device"); return MZK_ERR_INVALID_ARG; }
        if (i && p->W != keep[0]->W) { set_error("inconsistent plonk circuit types"); return MZK_ERR_INVALID_ARG; }
        keep.push_back(p);
        inst.push_back(p.get());
    }
    return MZK_OK;
}

}  // namespace
}  // namespace mzk

using namespace mzk;

extern "C" {

int32_t mzk_prover_create(int32_t curve_id, uint32_t log_n, uint32_t num_wire_types, const uint64_t* selector_coeffs, const uint64_t* sigma_coeffs,
                          const uint64_t* table_coeffs, uint64_t poly_len, const uint64_t* k_mont, uint64_t commit_key, uint64_t lagrange_key,
                          const mzk_comm* comm, uint64_t* out_prover) {
    if ((curve_id != 0 && curve_id != 1) || (num_wire_types != 5 && num_wire_types != 6) || !selector_coeffs || !sigma_coeffs || !k_mont || !out_prover ||
        (num_wire_types == 6) != (table_coeffs != nullptr) || log_n < 1 || log_n > 27 || poly_len == 0 || poly_len > (1ull << log_n)) {
        set_error("bad argument (TurboPlonk: 5 wire types, 13 selectors; UltraPlonk: 6 wire types, 14 selectors, 4 table polynomials; coefficient vectors of at most 2^log_n)");
        return MZK_ERR_INVALID_ARG;
    }
    // the quotient lives on the 8n-point domain: its W (n + 1) + 3 coefficients (prover.rs:916-919) must fit below 8n.  The reference sizes
    // that domain from the degree (n = 2 with five or six wire types and n = 4 with six would take 16n); such domains are refused here
    if ((uint64_t)num_wire_types * ((1ull << log_n) + 1) + 2 >= (8ull << log_n)) {
        set_error("domain too small for the 8n-point quotient domain: need num_wire_types * (n + 1) + 2 < 8 n (n >= 4 for TurboPlonk, n >= 8 for UltraPlonk)");
        return MZK_ERR_UNSUPPORTED;
    }
    int32_t device = -1;
    MZK_TRY(mzk_get_device(&device));
    if (device < 0) { set_error("mzk_init has not been called"); return MZK_ERR_NOT_INIT; }
    std::shared_ptr<ProverBase> p;
    try {
        if (curve_id == 0) p = std::make_shared<ProverT<BlsFr, MZK_CURVE_BLS12_381>>((int)log_n, (int)num_wire_types, selector_coeffs, sigma_coeffs, table_coeffs, poly_len, k_mont, commit_key, lagrange_key, comm);
        else p = std::make_shared<ProverT<BnFr, MZK_CURVE_BN254>>((int)log_n, (int)num_wire_types, selector_coeffs, sigma_coeffs, table_coeffs, poly_len, k_mont, commit_key, lagrange_key, comm);
    } catch (const Fail& e) {
        return e.rc;
    } catch (const std::bad_alloc&) {
        set_error("out of host memory");
        return MZK_ERR_OOM;
    }
    p->device = device;
    const uint64_t h = handle_make(device, g_next_prover++);
    std::lock_guard<std::mutex> lk(g_reg_lock);
    g_provers[h] = std::move(p);
    *out_prover = h;
    return MZK_OK;
}

int32_t mzk_prover_destroy(uint64_t prover) {
    std::shared_ptr<ProverBase> p;
    {
        std::lock_guard<std::mutex> lk(g_reg_lock);
        auto it = g_provers.find(prover);
        if (it == g_provers.end()) { set_error("unknown prover handle"); return MZK_ERR_BAD_HANDLE; }
        p = std::move(it->second);
        g_provers.erase(it);
    }
    DeviceGuard g(p->device);
    p.reset();                                                         // device memory is released on the prover's device
    return MZK_OK;
}

#define PROVER(h)                        \
    auto p_ = find(h);                   \
    if (!p_) return MZK_ERR_BAD_HANDLE

int32_t mzk_prover_vk_commitments(uint64_t prover, uint64_t* out_xy_mont, uint64_t* out_plookup_xy_mont) {
    PROVER(prover);
    if (!out_xy_mont) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    return guarded(*p_, [&] { p_->vk_commitments(out_xy_mont, out_plookup_xy_mont); });
}
int32_t mzk_prover_set_wire_variables(uint64_t prover, const uint32_t* wire_variables, uint64_t n_vars) {
    PROVER(prover);
    if (!wire_variables || n_vars == 0 || n_vars >= (1ull << 32)) { set_error("bad argument"); return MZK_ERR_INVALID_ARG; }
    return guarded(*p_, [&] { p_->set_wire_variables(wire_variables, n_vars); });
}
int32_t mzk_prover_round1(uint64_t prover, int32_t witness_kind, const void* witness, uint64_t witness_len, const uint64_t* pub_input_rows,
                          const uint64_t* pub_input_mont, uint64_t n_pub, const uint64_t* blinders_mont, uint64_t* out_comms_xy) {
    PROVER(prover);
    return guarded(*p_, [&] { p_->round1(witness_kind, witness, witness_len, pub_input_rows, pub_input_mont, n_pub, blinders_mont, out_comms_xy); });
}
int32_t mzk_prover_round1_5(uint64_t prover, const uint64_t* tau_mont, const uint64_t* blinders_mont, uint64_t* out_comms_xy) {
    PROVER(prover);
    if (!tau_mont || !blinders_mont || !out_comms_xy) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    return guarded(*p_, [&] { p_->round1_5(tau_mont, blinders_mont, out_comms_xy); });
}
int32_t mzk_prover_round2(uint64_t prover, const uint64_t* beta_mont, const uint64_t* gamma_mont, const uint64_t* blinders_mont, uint64_t* out_comm_xy) {
    PROVER(prover);
    if (!beta_mont || !gamma_mont || !blinders_mont || !out_comm_xy) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    return guarded(*p_, [&] { p_->round2(beta_mont, gamma_mont, blinders_mont, out_comm_xy); });
}
int32_t mzk_prover_round2_5(uint64_t prover, const uint64_t* blinders_mont, uint64_t* out_comm_xy) {
    PROVER(prover);
    if (!blinders_mont || !out_comm_xy) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    return guarded(*p_, [&] { p_->round2_5(blinders_mont, out_comm_xy); });
}
int32_t mzk_prover_round3(const uint64_t* provers, uint32_t n_instances, const uint64_t* alpha_mont, const uint64_t* blinders_mont, uint64_t* out_comms_xy) {
    std::vector<std::shared_ptr<ProverBase>> keep;
    std::vector<ProverBase*> inst;
    MZK_TRY(collect(provers, n_instances, keep, inst));
    if (!alpha_mont || !blinders_mont || !out_comms_xy) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    return guarded(*inst[0], [&] { inst[0]->round3(inst, alpha_mont, blinders_mont, out_comms_xy); });
}
int32_t mzk_prover_round4(uint64_t prover, const uint64_t* zeta_mont, uint64_t* out_evals_mont) {
    PROVER(prover);
    if (!zeta_mont || !out_evals_mont) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    return guarded(*p_, [&] { p_->round4(zeta_mont, out_evals_mont); });
}
int32_t mzk_prover_round5(const uint64_t* provers, uint32_t n_instances, const uint64_t* v_mont, uint64_t* out_comms_xy) {
    std::vector<std::shared_ptr<ProverBase>> keep;
    std::vector<ProverBase*> inst;
    MZK_TRY(collect(provers, n_instances, keep, inst));
    if (!v_mont || !out_comms_xy) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    return guarded(*inst[0], [&] { inst[0]->round5(inst, v_mont, out_comms_xy); });
}
int32_t mzk_prover_exchange_buffer(uint64_t prover, void** out_dptr, uint64_t* out_bytes) {
    PROVER(prover);
    p_->exchange_buffer(out_dptr, out_bytes);
    return MZK_OK;
}
int32_t mzk_prover_set_peer_buffers(uint64_t prover, void* const* peer_dptrs, const int32_t* peer_devices) {
    PROVER(prover);
    if (!peer_dptrs || !peer_devices) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    p_->set_peer_buffers(peer_dptrs, peer_devices);
    return MZK_OK;
}
int32_t mzk_prover_poly_dev(uint64_t prover, uint32_t which, const void** out_dptr, uint64_t* out_len) {
    PROVER(prover);
    if (!out_dptr || !out_len) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    return guarded(*p_, [&] { p_->poly_dev(which, out_dptr, out_len); });
}
int32_t mzk_prover_profile(uint64_t prover, int32_t on) {
    PROVER(prover);
    p_->profile = on != 0;
    return MZK_OK;
}
int32_t mzk_prover_timings(uint64_t prover, char* buf, uint64_t cap) {
    PROVER(prover);
    if (!buf || cap == 0) { set_error("null pointer"); return MZK_ERR_INVALID_ARG; }
    std::string s = "{";
    bool first = true;
    for (auto& kv : p_->timings_ms) {
        char t[96];
        std::snprintf(t, sizeof t, "%s\"%s\": %.3f", first ? "" : ", ", kv.first.c_str(), kv.second);
        s += t;
        first = false;
    }
    s += "}";
    std::snprintf(buf, cap, "%s", s.c_str());
    return MZK_OK;
}
int32_t mzk_prover_hbm_bytes(uint64_t prover, uint64_t* out_fixed, uint64_t* out_proving_key, uint64_t* out_workspace) {
    PROVER(prover);
    return guarded(*p_, [&] { p_->hbm_bytes(out_fixed, out_proving_key, out_workspace); });
}

}  // extern "C"
