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

    // PlonkKzgSnark::preprocess's device half (snark.rs:529-617): coefficient forms resident, evaluations on the needed classes
    ProverT(int log_n_, int W_, const uint64_t* sel, const uint64_t* sig, const uint64_t* tab, uint64_t poly_len, const uint64_t* k_mont,
            uint64_t commit_key, uint64_t lagrange_key, const mzk_comm* cm) {
        curve = CURVE; log_n = log_n_; W = W_; ultra = W_ == 6; nsel = ultra ? 14 : 13;
        rows = W + 2 + (ultra ? 3 : 0);
        n = 1ull << log_n; m = 8 * n;
        srs = commit_key; srs_lagrange = lagrange_key;
        if (cm) { comm = *cm; rank = cm->rank; world = cm->world; }
        if (world < 1 || rank < 0 || rank >= world || (world > 1 && (!comm.all_gather || !comm.barrier)))
            fail(MZK_ERR_INVALID_ARG, "mzk_comm: 0 <= rank < world, all_gather and barrier callbacks required");
        for (int i = 0; i < W; i++) k.push_back(load(k_mont + 4 * i));
        std::tie(lo, hi) = shard_range(n + 3, rank, world);            // the proving key keeps trim(n + 2) = n + 3 powers (snark.rs:535, 561)
        uint64_t srs_len = 0;
        ck(mzk_srs_len(srs, &srs_len));
        const bool sliced = world > 1 && srs_len == hi - lo;           // the key IS this rank's range of the SRS (mzk_srs_slice)
        if (sliced) key_first = lo;
        else if (srs_len < n + 3) fail(MZK_ERR_INVALID_ARG, "commit key too small: need domain size + 3 powers (srs.rs:88), or exactly this rank's point range");
        if (srs_lagrange) {
            ck(mzk_srs_len(srs_lagrange, &srs_len));
            if (sliced ? srs_len != hi - lo : srs_len < n + 3)
                fail(MZK_ERR_INVALID_ARG, "Lagrange-basis key: 2^log_n + 3 points (mzk_srs_lagrange_from_srs(.., log_n, 3)), sliced like the commit key");
        }
        const int nfix = nsel + W + (ultra ? 4 : 0);
        if (!std::getenv("MZK_PROVER_NULL_STREAM")) ck(mzk_stream_create(&S));
        fixed.alloc((size_t)nfix * n);
        ck(mzk_dev_memset(fixed.p, 0, (size_t)nfix * n * EL, S));
        sync_stream();                                                 // the uploads below are synchronous copies, not ordered after S
        auto up_rows = [&](int first, int cnt, const uint64_t* src) {
            if (poly_len == n) ck(mzk_dev_upload(fix(first), src, (size_t)cnt * n * EL));
            else for (int i = 0; i < cnt; i++) ck(mzk_dev_upload(fix(first + i), src + (size_t)i * poly_len * 4, poly_len * EL));
        };
        up_rows(0, nsel, sel);
        up_rows(nsel, W, sig);
        if (ultra) up_rows(nsel + W, 4, tab);
        // The quotient has degree W (n + 1) + 2 (prover.rs:916-919).  Its W + 3 coefficients from X^(Wn) on are the top coefficients of
        // its numerator (mzk_plonk_quotient_top_dev, n > W + 2), so W of the 8 residue classes of the quotient domain determine the
        // rest -- 5 for TurboPlonk, 6 for UltraPlonk -- and only those are resident and evaluated; the polynomial so recovered has the
        // expected degree whatever the witness, which is why round 5 checks the quotient identity at zeta.  Tiny domains: W + 1 classes
        // with one spare coefficient above the expected degree (or an unsatisfied witness could not trip WrongQuotientPolyDegree), else all 8.
        use_top = n > (uint64_t)W + 2 && n >= 8;
        const uint32_t needed = use_top ? (uint32_t)W : (((uint64_t)W * (n + 1) + 2 < (uint64_t)(W + 1) * n - 1 && W + 1 <= 8) ? (uint32_t)W + 1 : 8u);
        for (uint32_t kc = 0; kc < needed; kc++) classes.push_back(kc);
        own = class_range(rank, world, needed);                        // contiguous blocks of ceil(needed / world); the last ranks may own none
        // a rank that owns no class still registers one (a key cannot be empty); it is never evaluated
        const std::vector<uint32_t> resident = own.empty() ? std::vector<uint32_t>{classes.back()} : own;
        rem.alloc(classes.size() * n);                                 // the remainders of ALL needed classes: own ones computed here, the others received
        top.alloc(16);
        slab.alloc((size_t)rows * (n + 3)); quot.alloc(m); coeff.alloc((size_t)(W + 1) * n);
        split.alloc((size_t)W * (n + 3)); lin.alloc(n + 4); batch.alloc(n + 4); opening.alloc(n + 3); shifted.alloc(n + 3); tmp.alloc(64); deg.alloc(1);
        if (ultra) { hh.alloc(2 * n); table.alloc(n); lookup.alloc(n); sorted.alloc(2 * n); }
        // (last: a constructor that throws runs no destructor, and the key is the one resource the members do not release themselves)
        ck(mzk_plonk_pk_register_chunked(CURVE, log_n, W, sel, sig, tab, poly_len, k_mont, resident.data(), (uint32_t)resident.size(), &pk));
        w_n = root_of_unity<FrP>(log_n);
        gen = Fr::from_words(FrP::GENERATOR);
    }
    ~ProverT() override {
        (void)mzk_dev_sync();
        if (pk) (void)mzk_plonk_pk_release(pk);
        if (copy_stream) (void)mzk_stream_destroy(copy_stream);
        if (S) (void)mzk_stream_destroy(S);
    }
    void hbm_bytes(uint64_t* fixed_b, uint64_t* pk_b, uint64_t* ws_b) override {
        uint64_t ws = 0;
        for (const Buf* b : {&slab, &quot, &coeff, &split, &lin, &batch, &opening, &shifted, &hh, &table, &lookup, &sorted, &tmp, &deg, &rem, &wv, &wit, &top,
                             &vals_ext, &qsum, &vars})
            ws += b->bytes();
        if (fixed_b) *fixed_b = fixed.bytes();
        if (ws_b) *ws_b = ws;
        if (pk_b) ck(mzk_plonk_pk_hbm_bytes(pk, pk_b));
    }

    // ---- commitments ---------------------------------------------------------------------------------------------------------
    // Jacobian sums of the coefficients [a, b) of every polynomial over the SRS points of the same indices
    std::vector<uint64_t> msm_partials(const std::vector<const void*>& polys, const std::vector<uint64_t>& lens, uint64_t a, uint64_t b, uint64_t key = 0) {
        const uint32_t kp = (uint32_t)polys.size();
        std::vector<const void*> p(kp);
        std::vector<uint64_t> l(kp), off(kp), xyz((size_t)kp * 3 * QL);
        for (uint32_t i = 0; i < kp; i++) {
            const uint64_t s0 = std::min(a, lens[i]), s1 = std::min(b, lens[i]);
            p[i] = static_cast<const uint8_t*>(polys[i]) + s0 * EL;
            l[i] = s1 - s0;
            off[i] = (s1 > s0 ? s0 : a) - key_first;                     // (a sliced key starts at this rank's lo: a >= lo there)
        }
        ck(mzk_msm_batch_dev(key ? key : srs, kp, p.data(), l.data(), off.data(), 1, xyz.data(), S));
        return xyz;
    }
    // the ranks' partial sums -> the commitments, identical on every rank: all-gather of k x 144 B (96 B on BN254) through host
    // memory and <= 8 EC additions per commitment on the host (mzk_g1_sum_jacobian) -- the "all-reduce of partial EC sums"
    void combine_partials(const std::vector<uint64_t>& part, uint64_t* out_xy) {
        const size_t kp = part.size() / (3 * QL), one = 3 * QL;
        if (world == 1) { ck(mzk_g1_jacobian_to_affine(CURVE, part.data(), kp, out_xy)); return; }
        std::vector<uint64_t> all((size_t)world * part.size());
        if (comm.all_gather(comm.ctx, part.data(), part.size() * 8, all.data())) fail(MZK_ERR_INVALID_ARG, "mzk_comm.all_gather failed");
        std::vector<uint64_t> sum(kp * one), col((size_t)world * one);
        for (size_t i = 0; i < kp; i++) {
            for (int g = 0; g < world; g++) std::memcpy(&col[g * one], &all[((size_t)g * kp + i) * one], one * 8);
            ck(mzk_g1_sum_jacobian(CURVE, col.data(), world, &sum[i * one]));
        }
        ck(mzk_g1_jacobian_to_affine(CURVE, sum.data(), kp, out_xy));
    }
    // UnivariateKzgPCS::batch_commit (mod.rs:119-131) on device-resident coefficient vectors; over several ranks every MSM is
    // sharded by point range (this rank: [lo, hi))
    void commit(const std::vector<const void*>& polys, const std::vector<uint64_t>& lens, uint64_t* out_xy, uint64_t key = 0) {
        combine_partials(msm_partials(polys, lens, lo, hi, key), out_xy);
    }
    // ... of polynomials of which this rank holds ONLY the coefficients [lo, lo + lens[i])
    void commit_slices(const std::vector<const void*>& slices, const std::vector<uint64_t>& lens, uint64_t* out_xy) {
        const uint32_t kp = (uint32_t)slices.size();
        std::vector<uint64_t> off(kp, lo - key_first), xyz((size_t)kp * 3 * QL);
        ck(mzk_msm_batch_dev(srs, kp, slices.data(), lens.data(), off.data(), 1, xyz.data(), S));
        combine_partials(xyz, out_xy);
    }
    void vk_commitments(uint64_t* out_xy, uint64_t* out_plookup_xy) override {   // set-up work: over several ranks sharded by point range like every commitment
        std::vector<const void*> ptrs;
        std::vector<uint64_t> lens;
        const int cnt = nsel + W + (ultra && out_plookup_xy ? 4 : 0);
        for (int i = 0; i < cnt; i++) { ptrs.push_back(fix(i)); lens.push_back(n); }
        std::vector<uint64_t> xy((size_t)cnt * PT);
        for (int i = 0; i < cnt; i += W) {                                       // W at a time, like a round's commitments: set-up does not enlarge the MSM scratch the proofs need
            const int k = std::min(W, cnt - i);
            commit(std::vector<const void*>(ptrs.begin() + i, ptrs.begin() + i + k), std::vector<uint64_t>(lens.begin() + i, lens.begin() + i + k), &xy[(size_t)i * PT]);
        }
        std::memcpy(out_xy, xy.data(), (size_t)(nsel + W) * PT * 8);
        if (ultra && out_plookup_xy) std::memcpy(out_plookup_xy, &xy[(size_t)(nsel + W) * PT], 4 * PT * 8);
    }

    // ---- small helpers ---------------------------------------------------------------------------------------------------------
    std::vector<Fr> evaluate(const void* d, uint64_t len, uint32_t batch_n, uint64_t stride, const Fr& x) {
        std::vector<Fr> out(batch_n);
        ck(mzk_poly_eval_dev(CURVE, d, len, batch_n, stride, x.l, reinterpret_cast<uint64_t*>(out.data()), S));
        return out;
    }
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
    void lincomb(const std::vector<Term>& terms, void* out, uint64_t out_len) {
        std::vector<const void*> ptrs;
        std::vector<uint64_t> lens, sc;
        for (auto& t : terms) { ptrs.push_back(t.p); lens.push_back(t.len); for (int i = 0; i < 4; i++) sc.push_back(t.s.l[i]); }
        ck(mzk_poly_lincomb_dev(CURVE, (uint32_t)terms.size(), ptrs.data(), lens.data(), sc.data(), out, out_len, S));
    }
    // sum of any number of terms into `out` (one launch takes 32)
    void lincomb_many(const std::vector<Term>& terms, void* out, uint64_t out_len) {
        constexpr size_t MAXT = 32;
        if (terms.size() <= MAXT) { lincomb(terms, out, out_len); return; }
        lincomb(std::vector<Term>(terms.begin(), terms.begin() + MAXT), out, out_len);
        for (size_t i = MAXT; i < terms.size(); i += MAXT - 1) {
            std::vector<Term> chunk{{Fr::one(), out, out_len}};
            chunk.insert(chunk.end(), terms.begin() + i, terms.begin() + std::min(terms.size(), i + MAXT - 1));
            lincomb(chunk, out, out_len);                              // elementwise: reading out[j] before writing it is safe
        }
    }
    void mask(const std::vector<int>& slab_rows, const uint64_t* blinders, uint32_t n_blind) {      // prover.rs:463-486
        std::vector<void*> ptrs;
        for (int r : slab_rows) ptrs.push_back(row(r));
        ck(mzk_poly_mask_dev(CURVE, (uint32_t)ptrs.size(), ptrs.data(), n, n_blind, blinders, S));
    }
    const void* one_dev() {                                            // the field's one (Montgomery), resident: tmp[1]
        if (!one_ready) {
            const Fr one = Fr::one();
            ck(mzk_dev_upload(tmp.at(1), one.l, EL));
            one_ready = true;
        }
        return tmp.at(1);
    }
    // a scalar into device memory as a kernel argument (times the resident one): no host-to-device copy, hence no stream synchronisation
    void put_scalar(const Fr& v, void* d) { lincomb({{v, one_dev(), 1}}, d, 1); }
    void* rem_dev() const { return tmp.at(2); }                        // where the opening division leaves the batch polynomial's value at zeta
    Fr download_fr(const void* d) {
        Fr v;
        sync_stream();                                                 // (a synchronous copy is ordered after the null stream, not after S)
        ck(mzk_dev_download(v.l, d, EL));
        return v;
    }
    void need(int at_least, int below, const char* what) const {
        if (stage < at_least || stage >= below) fail(MZK_ERR_STATE, std::string("prover rounds out of order: ") + what);
    }

    // ---- round 1 (prover.rs:72-87; constraint_system.rs:1225-1259) -------------------------------------------------------------
    void set_wire_variables(const uint32_t* v, uint64_t nv) override {
        const size_t cnt = (size_t)W * n;
        for (size_t i = 0; i < cnt; i++)
            if (v[i] >= nv) fail(MZK_ERR_INVALID_ARG, "wire_variables: variable index " + std::to_string(v[i]) + " >= number of variables " + std::to_string(nv));
        vars.alloc((cnt * 4 + EL - 1) / EL);
        ck(mzk_dev_upload(vars.p, v, cnt * 4));
        n_vars = nv;
    }
    void public_input_row(const uint64_t* pi_rows, const uint64_t* pi, uint64_t n_pi) {
        void* d = coeff.at((size_t)W * n);
        ck(mzk_dev_memset(d, 0, n * EL, S));
        st.pi_zero = true;
        for (uint64_t i = 0; i < n_pi * 4 && st.pi_zero; i++) st.pi_zero = pi[i] == 0;
        if (st.pi_zero) return;
        if (n_pi > n) fail(MZK_ERR_INVALID_ARG, "more public inputs than rows");
        // staged through page-locked memory: asynchronous on the null stream, the caller's buffer is free when the call returns
        uint64_t* h = static_cast<uint64_t*>(stage_pi.reserve(n_pi * EL));
        std::memcpy(h, pi, n_pi * EL);
        if (!pi_rows) {
            ck(mzk_dev_upload_async(d, h, n_pi * EL, S));
        } else {
            for (uint64_t i = 0; i < n_pi; i++) {
                if (pi_rows[i] >= n) fail(MZK_ERR_INVALID_ARG, "public-input row outside the domain");
                ck(mzk_dev_upload_async(static_cast<uint8_t*>(d) + pi_rows[i] * EL, h + 4 * i, EL, S));
            }
        }
        ck(mzk_ntt_dev(CURVE, d, n, log_n, 1, nullptr, 1, n, S));      // compute_pub_input_polynomial (:1249-1259)
    }
    void round1(int kind, const void* witness, uint64_t witness_len, const uint64_t* pi_rows, const uint64_t* pi, uint64_t n_pi, const uint64_t* blinders,
                uint64_t* out) override {
        Tick tick(*this);
        st = State();
        stage = CREATED;
        timings_ms.clear();
        if (!witness || !blinders || !out || (n_pi && !pi)) fail(MZK_ERR_INVALID_ARG, "null pointer");
        if (S) ck(mzk_stream_wait_stream(S, nullptr));                 // what the caller wrote on the null stream (a device-resident witness) is complete first
        public_input_row(pi_rows, pi, n_pi);
        const size_t cells = (size_t)W * n;
        if (kind == MZK_WITNESS_DEV_WIRES) {
            if (witness_len != cells) fail(MZK_ERR_INVALID_ARG, "witness_len != num_wire_types * domain size");
            st.wire_values = witness;
            ck(mzk_dev_copy(coeff.p, witness, cells * EL, S));
            ck(mzk_ntt_dev(CURVE, coeff.p, n, log_n, 1, nullptr, W, n, S));
        } else if (kind == MZK_WITNESS_HOST_VECTOR || kind == MZK_WITNESS_DEV_VECTOR) {
            // the witness vector crosses PCIe; `witness[wire_variable(i, j)]` (constraint_system.rs:1239) is gathered on the device
            if (!vars.p) fail(MZK_ERR_STATE, "mzk_prover_set_wire_variables has not been called");
            if (witness_len != n_vars) fail(MZK_ERR_INVALID_ARG, "witness_len != the n_vars of mzk_prover_set_wire_variables");
            if (!wv.p) wv.alloc(cells);
            st.wire_values = wv.p;
            const void* d_wit = witness;
            if (kind == MZK_WITNESS_HOST_VECTOR) {
                if (wit.elems < n_vars) wit.alloc(n_vars);
                if (!copy_stream) ck(mzk_stream_create(&copy_stream));
                ck(mzk_stream_wait_stream(copy_stream, S));                                       // the previous proof is done with `wit`
                ck(mzk_dev_upload_async(wit.p, witness, n_vars * EL, copy_stream));
                ck(mzk_stream_wait_stream(S, copy_stream));
                d_wit = wit.p;
            }
            ck(mzk_plonk_gather_witness_dev(d_wit, n_vars, vars.p, cells, wv.p, S));
            ck(mzk_dev_copy(coeff.p, wv.p, cells * EL, S));
            ck(mzk_ntt_dev(CURVE, coeff.p, n, log_n, 1, nullptr, W, n, S));
        } else if (kind == MZK_WITNESS_HOST_WIRES) {
            // host-resident witness (constraint_system.rs:1225-1247 gathers it on the host): column i + 1 crosses PCIe on a copy
            // stream while column i is transformed on the null stream
            if (witness_len != cells) fail(MZK_ERR_INVALID_ARG, "witness_len != num_wire_types * domain size");
            if (!wv.p) wv.alloc(cells);
            if (!copy_stream) ck(mzk_stream_create(&copy_stream));
            st.wire_values = wv.p;
            ck(mzk_stream_wait_stream(copy_stream, S));                                           // the previous proof is done with `wv`
            for (int i = 0; i < W; i++) {
                ck(mzk_dev_upload_async(wv.at((size_t)i * n), static_cast<const uint8_t*>(witness) + (size_t)i * n * EL, n * EL, copy_stream));
                ck(mzk_stream_wait_stream(S, copy_stream));                                       // columns 0..i have arrived
                ck(mzk_dev_copy(coeff.at((size_t)i * n), wv.at((size_t)i * n), n * EL, S));
                ck(mzk_ntt_dev(CURVE, coeff.at((size_t)i * n), n, log_n, 1, nullptr, 1, n, S));
            }
        } else {
            fail(MZK_ERR_INVALID_ARG, "unknown witness_kind");
        }
        for (int r = 0; r < rows; r++) ck(mzk_dev_memset(static_cast<uint8_t*>(row(r)) + n * EL, 0, 3 * EL, S));
        ck(mzk_dev_copy2d(slab.p, (n + 3) * EL, coeff.p, n * EL, n * EL, W, S));
        ck(mzk_dev_copy(row(rowPI()), coeff.at((size_t)W * n), n * EL, S));
        { std::vector<int> rs; for (int i = 0; i < W; i++) rs.push_back(i); mask(rs, blinders, 2); }
        tick.mark("r1_ntt_mask");
        std::vector<const void*> p; std::vector<uint64_t> l;
        if (srs_lagrange) {
            // sum_i v_i [L_i(beta)]g + b_0 [Z_H(beta)]g + b_1 [beta Z_H(beta)]g: rows of n + 3 slots, the values, then the blinders
            if (!vals_ext.p) vals_ext.alloc((size_t)W * (n + 3));
            ck(mzk_dev_copy2d(vals_ext.p, (n + 3) * EL, st.wire_values, n * EL, n * EL, W, S));
            for (int i = 0; i < W; i++)
                for (int j = 0; j < 2; j++) put_scalar(load(blinders + (size_t)(2 * i + j) * 4), vals_ext.at((size_t)i * (n + 3) + n + j));
            for (int i = 0; i < W; i++) { p.push_back(vals_ext.at((size_t)i * (n + 3))); l.push_back(n + 2); }
        } else {
            for (int i = 0; i < W; i++) { p.push_back(row(i)); l.push_back(n + 2); }
        }
        commit(p, l, out, srs_lagrange);                                // (synchronises the null stream, which has waited for the copy stream: the
        tick.mark("r1_commit");                                         // caller's host buffers are free again when this returns)
        stage = R1;
    }
    // ---- round 1.5 (prover.rs:89-118), UltraPlonk only --------------------------------------------------------------------------
    void round1_5(const uint64_t* tau, const uint64_t* blinders, uint64_t* out) override {
        if (!ultra) fail(MZK_ERR_UNSUPPORTED, "round 1.5 exists for UltraPlonk only");
        need(R1, R1_5, "round 1.5 follows round 1");
        Tick tick(*this);
        st.tau = load(tau);
        const int H1 = rowH1();
        ck(mzk_plookup_sorted_vec_dev(pk, st.wire_values, st.tau.l, table.p, lookup.p, sorted.p, S));
        ck(mzk_dev_copy(hh.p, sorted.p, n * EL, S));
        ck(mzk_dev_copy(hh.at(n), sorted.at(n - 1), n * EL, S));
        if (srs_lagrange) {
            // h_1, h_2 are committed from the sorted vector's VALUES (table entries and looked-up values: small numbers unless the circuit
            // looks up keyed tables) plus their three blinders, over the Lagrange-basis key -- as the wires in round 1
            if (!vals_ext.p) vals_ext.alloc((size_t)W * (n + 3));
            ck(mzk_dev_copy2d(vals_ext.p, (n + 3) * EL, hh.p, n * EL, n * EL, 2, S));
            for (int i = 0; i < 2; i++)
                for (int j = 0; j < 3; j++) put_scalar(load(blinders + (size_t)(3 * i + j) * 4), vals_ext.at((size_t)i * (n + 3) + n + j));
        }
        ck(mzk_ntt_dev(CURVE, hh.p, n, log_n, 1, nullptr, 2, n, S));
        ck(mzk_dev_copy2d(row(H1), (n + 3) * EL, hh.p, n * EL, n * EL, 2, S));
        mask({H1, H1 + 1}, blinders, 3);
        tick.mark("r1_5_sorted_vec");
        if (srs_lagrange) commit({vals_ext.p, vals_ext.at(n + 3)}, {n + 3, n + 3}, out, srs_lagrange);
        else commit({row(H1), row(H1 + 1)}, {n + 3, n + 3}, out);
        tick.mark("r1_5_commit");
        stage = R1_5;
    }
    // ---- round 2 (prover.rs:125-141) -------------------------------------------------------------------------------------------
    void round2(const uint64_t* beta, const uint64_t* gamma, const uint64_t* blinders, uint64_t* out) override {
        need(ultra ? R1_5 : R1, R2, ultra ? "round 2 follows round 1.5" : "round 2 follows round 1");
        Tick tick(*this);
        st.beta = load(beta); st.gamma = load(gamma);
        ck(mzk_plonk_perm_product_dev(pk, st.wire_values, st.beta.l, st.gamma.l, coeff.p, S));
        ck(mzk_dev_copy(row(rowZ()), coeff.p, n * EL, S));
        mask({rowZ()}, blinders, 3);
        tick.mark("r2_product");
        commit({row(rowZ())}, {n + 3}, out);
        tick.mark("r2_commit");
        stage = R2;
    }
    // ---- round 2.5 (prover.rs:143-183), UltraPlonk only --------------------------------------------------------------------------
    void round2_5(const uint64_t* blinders, uint64_t* out) override {
        if (!ultra) fail(MZK_ERR_UNSUPPORTED, "round 2.5 exists for UltraPlonk only");
        need(R2, R2_5, "round 2.5 follows round 2");
        Tick tick(*this);
        ck(mzk_plookup_product_dev(pk, table.p, lookup.p, sorted.p, st.beta.l, st.gamma.l, coeff.p, S));
        ck(mzk_dev_copy(row(rowPL()), coeff.p, n * EL, S));
        mask({rowPL()}, blinders, 3);
        tick.mark("r2_5_product");
        commit({row(rowPL())}, {n + 3}, out);
        tick.mark("r2_5_commit");
        stage = R2_5;
    }
    // ---- round 3 (prover.rs:192-209, 512-673, 902-960) ----------------------------------------------------------------------------
    // this instance's quotient polynomial, 8n coefficients into `quot` (prover.rs:512-673 without the sum over instances)
    void quotient(const Fr& alpha) {
        st.alpha = alpha;
        // per OWN class: fold mod X^n - h_k^n, size-n coset NTTs, the fused kernel, size-n inverse coset NTT -> t mod (X^n - h_k^n), straight
        // into this class's slot of `rem` (the rows of the slab are read, not overwritten)
        if (!own.empty())
            ck(mzk_plonk_quotient_chunked_flags_dev(pk, slab.p, n + 3, n + 3, st.pi_zero ? MZK_QUOTIENT_PI_ZERO : 0u, ultra ? st.tau.l : nullptr, alpha.l,
                                                    st.beta.l, st.gamma.l, rem.at((size_t)own[0] * n), S));
        if (world > 1) {
            // THE one exchange
            const uint32_t first = own.empty() ? 0u : own[0];
            if (comm.exchange_classes) {
                ck(mzk_dev_sync());
                if (comm.exchange_classes(comm.ctx, rem.p, n * EL, first, (uint32_t)own.size(), (uint32_t)classes.size())) fail(MZK_ERR_INVALID_ARG, "mzk_comm.exchange_classes failed");
            } else {
                // every rank pushes its class remainders into the same slots of every other rank's `rem`, device to device (xGMI peer
                // copies; n x 32 B per class and peer), then all ranks meet
                if ((int)peer_rem.size() != world) fail(MZK_ERR_STATE, "mzk_prover_set_peer_buffers has not been called");
                if (!own.empty())
                    for (int q = 0; q < world; q++)
                        if (q != rank)
                            ck(mzk_dev_copy_peer(static_cast<uint8_t*>(peer_rem[q]) + (size_t)own[0] * n * EL, peer_dev[q], rem.at((size_t)own[0] * n), device,
                                                 own.size() * n * EL, S));
                ck(mzk_dev_sync());
                if (comm.barrier(comm.ctx)) fail(MZK_ERR_INVALID_ARG, "mzk_comm.barrier failed");
            }
        }
        // the inverse Vandermonde per coefficient index (replicated: every rank needs the quotient's coefficients for the split)
        if (use_top) {
            ck(mzk_plonk_quotient_top_dev(pk, slab.p, n + 3, n + 3, alpha.l, st.beta.l, st.gamma.l, top.p, nullptr, S));
            ck(mzk_plonk_quotient_combine_top_dev(CURVE, log_n, classes.data(), (uint32_t)classes.size(), rem.p, top.p, (uint32_t)W + 3, quot.p, S));
        } else {
            ck(mzk_plonk_quotient_combine_classes_dev(CURVE, log_n, classes.data(), (uint32_t)classes.size(), rem.p, quot.p, S));
        }
    }
    // split_quotient_polynomial (prover.rs:902-960) of the 8n coefficients at `q` into this->split
    void split_quotient(const void* q, const uint64_t* b_quot) {
        const uint64_t expected = (uint64_t)W * (n + 1) + 2;                                                  // quotient_polynomial_degree
        // only what lies at and above the expected degree is scanned: its length must be exactly 1 (read after the commitments)
        ck(mzk_poly_degree_dev(static_cast<const uint8_t*>(q) + expected * EL, m - expected, static_cast<uint64_t*>(deg.p), S));
        ck(mzk_dev_memset(split.p, 0, (size_t)W * (n + 3) * EL, S));
        st.split_len.assign(W, 0);
        Fr last = Fr::zero();
        for (int i = 0; i < W; i++) {
            const uint64_t a = (uint64_t)i * (n + 2), b = i < W - 1 ? a + n + 2 : expected + 1;
            void* p = split.at((size_t)i * (n + 3));
            ck(mzk_dev_copy(p, static_cast<const uint8_t*>(q) + a * EL, (b - a) * EL, S));
            if (i < W - 1) put_scalar(load(b_quot + 4 * i), static_cast<uint8_t*>(p) + (n + 2) * EL);
            if (i > 0) lincomb({{Fr::one(), p, 1}, {neg(last), one_dev(), 1}}, p, 1);                             // t_i[0] -= b_{i-1}
            if (i < W - 1) last = load(b_quot + 4 * i);
            st.split_len[i] = i < W - 1 ? n + 3 : b - a;
        }
    }
    void round3(const std::vector<ProverBase*>& inst, const uint64_t* alpha_p, const uint64_t* blinders, uint64_t* out) override {
        Tick tick(*this);
        if (world > 1 && inst.size() > 1) fail(MZK_ERR_UNSUPPORTED, "several instances over several ranks: not supported");
        const Fr alpha = load(alpha_p);
        const Fr a3 = alpha * alpha * alpha, a7 = a3 * a3 * alpha;
        std::vector<Term> qterms;
        st.bases.clear();
        Fr base = Fr::one();
        for (ProverBase* b : inst) {
            ProverT* p = static_cast<ProverT*>(b);
            p->need(p->ultra ? R2_5 : R2, R3, "round 3 follows round 2 (2.5 with Plookup) of every instance");
            p->quotient(alpha);
            qterms.push_back({base, p->quot.p, m});
            st.bases.push_back(base);
            base = base * (p->ultra ? a7 : a3);                         // prover.rs:661-669
            if (p != this) ck(mzk_stream_wait_stream(S, p->S));         // its quotient was computed on its own stream
        }
        const void* q = quot.p;
        if (inst.size() > 1) {                                          // the per-instance quotients are combined after their inverse NTTs (linear maps)
            if (!qsum.p) qsum.alloc(m);
            lincomb_many(qterms, qsum.p, m);
            q = qsum.p;
        }
        tick.mark("r3_quotient");
        split_quotient(q, blinders);
        tick.mark("r3_split");
        std::vector<const void*> p;
        for (int i = 0; i < W; i++) p.push_back(split.at((size_t)i * (n + 3)));
        commit(p, st.split_len, out);
        tick.mark("r3_commit");
        // quot_poly.degree() != expected_degree => WrongQuotientPolyDegree (prover.rs:915-918): the reference's only guard against an
        // unsatisfied witness; it can fire on the W + 1 / 8-class paths (tiny domains) only -- with the top coefficients taken from the
        // numerator the degree is right by construction and round 5 checks the identity at zeta.  (The commitments have synchronised the
        // stream; this reads 8 bytes.)
        uint64_t tail = 0;
        sync_stream();
        ck(mzk_dev_download(&tail, deg.p, 8));
        for (ProverBase* b : inst) b->stage = R3;
        if (tail != 1) {
            const uint64_t expected = (uint64_t)W * (n + 1) + 2;
            for (ProverBase* b : inst) b->stage = CREATED;
            fail(MZK_ERR_WRONG_QUOTIENT_DEGREE, "WrongQuotientPolyDegree: quotient polynomial of degree " +
                     (tail ? std::to_string(expected + tail - 1) : "below " + std::to_string(expected)) + ", expected " + std::to_string(expected) +
                     " (the witness does not satisfy the circuit)");
        }
    }
    // ---- round 4: compute_evaluations / compute_plookup_evaluations (prover.rs:216-299) -------------------------------------------
    void round4(const uint64_t* zeta_p, uint64_t* out_evals) override {
        need(R3, R4, "round 4 follows round 3");
        Tick tick(*this);
        const Fr zeta = st.zeta = load(zeta_p);
        const Fr zeta_w = zeta * w_n;
        const int sigma0 = nsel, tab0 = nsel + W, H1 = rowH1(), PL = rowPL();
        EvalBatch ev(*this);
        // the wires and, in the same launch, z and the public-input polynomial (rows W, W + 1; every row is zero above its own length):
        // pi(zeta) is not part of the proof, the identity check of round 5 needs it
        const size_t h_w = ev.add(row(0), n + 3, W + 2, n + 3, zeta);
        const size_t h_s = ev.add(fix(sigma0), n, W - 1, n, zeta);
        const size_t h_z = ev.add(row(rowZ()), n + 3, 1, n + 3, zeta_w);
        size_t h_tz = 0, h_tn = 0, h_h1 = 0, h_ql = 0, h_qln = 0, h_pl = 0, h_hn = 0, h_wn = 0;
        if (ultra) {
            h_tz = ev.add(fix(tab0), n, 4, n, zeta);                                                           // range, key, table_dom_sep, q_dom_sep
            h_tn = ev.add(fix(tab0), n, 3, n, zeta_w);
            h_h1 = ev.add(row(H1), n + 3, 1, n + 3, zeta);
            h_ql = ev.add(fix(13), n, 1, n, zeta);
            h_qln = ev.add(fix(13), n, 1, n, zeta_w);
            h_pl = ev.add(row(PL), n + 3, 1, n + 3, zeta_w);
            h_hn = ev.add(row(H1), n + 3, 2, n + 3, zeta_w);
            h_wn = ev.add(row(3), n + 2, 2, n + 3, zeta_w);
        }
        ev.finish();
        const std::vector<Fr>& v = ev.vals;
        st.wires_evals.assign(v.begin() + h_w, v.begin() + h_w + W);
        st.pi_eval = v[h_w + W + 1];
        st.wire_sigma_evals.assign(v.begin() + h_s, v.begin() + h_s + W - 1);
        st.perm_next_eval = v[h_z];
        std::vector<Fr>& pe = st.plookup_evals;
        pe.clear();
        if (ultra) {
            pe.assign(N_PLOOKUP_EVALS, Fr::zero());
            pe[RANGE_TABLE] = v[h_tz]; pe[KEY_TABLE] = v[h_tz + 1]; pe[TABLE_DOM_SEP] = v[h_tz + 2]; pe[Q_DOM_SEP] = v[h_tz + 3];
            pe[RANGE_TABLE_NEXT] = v[h_tn]; pe[KEY_TABLE_NEXT] = v[h_tn + 1]; pe[TABLE_DOM_SEP_NEXT] = v[h_tn + 2];
            pe[H_1] = v[h_h1];
            pe[Q_LOOKUP] = v[h_ql];
            pe[Q_LOOKUP_NEXT] = v[h_qln];
            pe[PROD_NEXT] = v[h_pl];
            pe[H_1_NEXT] = v[h_hn]; pe[H_2_NEXT] = v[h_hn + 1];
            pe[W_3_NEXT] = v[h_wn]; pe[W_4_NEXT] = v[h_wn + 1];
        }
        uint64_t* o = out_evals;
        auto put = [&](const Fr& x) { std::memcpy(o, x.l, EL); o += 4; };
        for (auto& x : st.wires_evals) put(x);
        for (auto& x : st.wire_sigma_evals) put(x);
        put(st.perm_next_eval);
        for (auto& x : pe) put(x);
        tick.mark("r4_evals");
        stage = R4;
    }
    // ---- round 5 ---------------------------------------------------------------------------------------------------------------
    // compute_non_quotient_component_for_lin_poly (prover.rs:302-337, 963-1112) as terms, every scalar times alpha_base
    std::vector<Term> lin_poly_terms(const Fr& alpha_base) const {
        const std::vector<Fr>& we = st.wires_evals;
        const std::vector<Fr>& pe = st.plookup_evals;
        const Fr &alpha = st.alpha, &beta = st.beta, &gamma = st.gamma, &tau = st.tau, &zeta = st.zeta;
        const int sigma0 = nsel;
        auto pow5 = [](const Fr& x) { const Fr x2 = x * x; return x2 * x2 * x; };
        std::vector<Term> terms;
        for (int j = 0; j < 4; j++) terms.push_back({we[j], fix(j), n});
        terms.push_back({we[0] * we[1], fix(4), n});
        terms.push_back({we[2] * we[3], fix(5), n});
        for (int j = 0; j < 4; j++) terms.push_back({pow5(we[j]), fix(6 + j), n});
        terms.push_back({we[0] * we[1] * we[2] * we[3] * we[4], fix(12), n});
        terms.push_back({neg(we[4]), fix(10), n});
        terms.push_back({Fr::one(), fix(11), n});
        const Fr one = Fr::one(), nf = from_u64<FrP>(n);
        const Fr vanish = pow_u64(zeta, n) - one;
        const Fr lagrange_1 = vanish * inv(nf * (zeta - one));
        Fr cf = alpha;
        for (int j = 0; j < W; j++) cf = cf * (we[j] + beta * k[j] * zeta + gamma);
        terms.push_back({cf + alpha * alpha * lagrange_1, row(rowZ()), n + 3});
        cf = alpha * beta * st.perm_next_eval;
        for (int j = 0; j < W - 1; j++) cf = cf * (we[j] + beta * st.wire_sigma_evals[j] + gamma);
        terms.push_back({neg(cf), fix(sigma0 + W - 1), n});
        if (ultra) {                                                                                          // compute_lin_poly_plookup_contribution
            auto em = [&](const Fr& first, const Fr& ql, const Fr& ds, const Fr& a0, const Fr& a1, const Fr& a2) {
                return first + ql * tau * (ds + tau * (a0 + tau * (a1 + tau * a2)));
            };
            const Fr mt = em(pe[RANGE_TABLE], pe[Q_LOOKUP], pe[TABLE_DOM_SEP], pe[KEY_TABLE], we[3], we[4]);
            const Fr mt_next = em(pe[RANGE_TABLE_NEXT], pe[Q_LOOKUP_NEXT], pe[TABLE_DOM_SEP_NEXT], pe[KEY_TABLE_NEXT], pe[W_3_NEXT], pe[W_4_NEXT]);
            const Fr ml = em(we[5], pe[Q_LOOKUP], pe[Q_DOM_SEP], we[0], we[1], we[2]);
            const Fr w_inv = inv(w_n);
            const Fr lagrange_n = vanish * w_inv * inv(nf * (zeta - w_inv));
            const Fr a2 = alpha * alpha, a4 = a2 * a2, a5 = a4 * alpha, a6 = a4 * a2;
            const Fr b1 = one + beta, g1 = gamma * b1, zmg = zeta - w_inv;
            terms.push_back({a4 * lagrange_1 + a5 * lagrange_n + a6 * zmg * b1 * (gamma + ml) * (g1 + mt + beta * mt_next), row(rowPL()), n + 3});
            terms.push_back({neg(a6 * zmg * pe[PROD_NEXT] * (g1 + pe[H_1] + beta * pe[H_1_NEXT])), row(rowH1() + 1), n + 3});
        }
        if (!(alpha_base == one)) for (auto& t : terms) t.s = t.s * alpha_base;
        return terms;
    }
    // What the verifier takes for -(linearisation polynomial)(zeta): Verifier::compute_lin_poly_constant_term (verifier.rs:340-414) for this
    // instance, times alpha_base.  The prover knows every input: its own evaluations and pi(zeta).
    Fr lin_poly_constant(const Fr& alpha_base) const {
        const std::vector<Fr>& we = st.wires_evals;
        const std::vector<Fr>& pe = st.plookup_evals;
        const Fr &alpha = st.alpha, &beta = st.beta, &gamma = st.gamma, &zeta = st.zeta;
        const Fr one = Fr::one(), nf = from_u64<FrP>(n), a2 = alpha * alpha;
        const Fr vanish = pow_u64(zeta, n) - one;
        const Fr lagrange_1 = vanish * inv(nf * (zeta - one));
        Fr t = st.pi_eval - a2 * lagrange_1;
        Fr acc = alpha * st.perm_next_eval * (gamma + we[W - 1]);
        for (int j = 0; j < W - 1; j++) acc = acc * (gamma + we[j] + beta * st.wire_sigma_evals[j]);
        t = t - acc;
        if (ultra) {
            const Fr a3 = a2 * alpha, w_inv = inv(w_n);
            const Fr lagrange_n = vanish * w_inv * inv(nf * (zeta - w_inv));
            const Fr g1 = gamma * (one + beta);
            const Fr pc = lagrange_n * (pe[H_1] - pe[H_2_NEXT] - a2) - alpha * lagrange_1
                          - a3 * (zeta - w_inv) * pe[PROD_NEXT] * (g1 + pe[H_1] + beta * pe[H_1_NEXT]) * (g1 + beta * pe[H_2_NEXT]);
            t = t + a3 * pc;
        }
        return t * alpha_base;
    }
    // the evaluations at zeta in the order of open_lists' first list (after the linearisation polynomial)
    void opened_evals(std::vector<Fr>& out) const {
        out.insert(out.end(), st.wires_evals.begin(), st.wires_evals.end());
        out.insert(out.end(), st.wire_sigma_evals.begin(), st.wire_sigma_evals.end());
        if (ultra) {
            const std::vector<Fr>& pe = st.plookup_evals;
            for (int i : {RANGE_TABLE, KEY_TABLE, H_1, Q_LOOKUP, TABLE_DOM_SEP, Q_DOM_SEP}) out.push_back(pe[i]);
        }
    }
    // compute_quotient_component_for_lin_poly (prover.rs:343-358) over this->split
    std::vector<Term> quotient_lin_terms(const Fr& zeta) const {
        const Fr one = Fr::one(), vanish = pow_u64(zeta, n) - one, zeta_n2 = (vanish + one) * zeta * zeta;
        std::vector<Term> terms;
        Fr cf = one;
        for (int i = 0; i < W; i++) {
            terms.push_back({neg(vanish) * cf, split.at((size_t)i * (n + 3)), st.split_len[i]});
            cf = cf * zeta_n2;
        }
        return terms;
    }
    // the polynomials this instance opens at zeta (after the linearisation polynomial) and at zeta * w (prover.rs:362-460)
    void open_lists(std::vector<Term>& open_polys, std::vector<Term>& shifted_polys) const {
        const Fr one = Fr::one();
        const int sigma0 = nsel, tab0 = nsel + W, H1 = rowH1(), PL = rowPL();
        for (int i = 0; i < W; i++) open_polys.push_back({one, row(i), n + 2});
        for (int i = 0; i < W - 1; i++) open_polys.push_back({one, fix(sigma0 + i), n});
        shifted_polys.push_back({one, row(rowZ()), n + 3});
        if (ultra) {
            for (const void* p : {(const void*)fix(tab0), (const void*)fix(tab0 + 1)}) open_polys.push_back({one, p, n});
            open_polys.push_back({one, row(H1), n + 3});
            open_polys.push_back({one, fix(13), n});
            open_polys.push_back({one, fix(tab0 + 2), n});
            open_polys.push_back({one, fix(tab0 + 3), n});
            shifted_polys.push_back({one, row(PL), n + 3});
            shifted_polys.push_back({one, fix(tab0), n});
            shifted_polys.push_back({one, fix(tab0 + 1), n});
            shifted_polys.push_back({one, row(H1), n + 3});
            shifted_polys.push_back({one, row(H1 + 1), n + 3});
            shifted_polys.push_back({one, fix(13), n});
            shifted_polys.push_back({one, row(3), n + 2});
            shifted_polys.push_back({one, row(4), n + 2});
            shifted_polys.push_back({one, fix(tab0 + 2), n});
        }
    }
    // compute_batched_witness_polynomial_commitment (prover.rs:490-509) up to the commitment
    void batched_witness(const std::vector<Term>& polys, const Fr& v, const Fr& point, Buf& out, void* d_rem = nullptr) {
        std::vector<Term> t;
        Fr c = Fr::one();
        for (auto& p : polys) { t.push_back({c, p.p, p.len}); c = c * v; }
        lincomb_many(t, batch.p, n + 3);
        if (d_rem) ck(mzk_poly_div_linear_rem_dev(CURVE, batch.p, n + 3, point.l, out.p, d_rem, S));   // remainder = batch(point)
        else ck(mzk_poly_div_linear_dev(CURVE, batch.p, n + 3, point.l, out.p, S));
    }
    // Round 5 over several ranks.  The opening witness of a batch polynomial b at a point z is
    // w_j = sum_{i > j} b_i z^(i-j-1).  A rank needs w on its own coefficient range [lo, hi) only -- that is its MSM shard -- and
    // w_j = (the same sum over i < hi) + z^(hi-1-j) S_hi with S_hi = sum_{i >= hi} b_i z^(i-hi): the higher ranks' contribution enters
    // as ONE field element.  So: linear combinations on the range only (they are pointwise); e = the range read as a polynomial,
    // evaluated at z; one all-gather of the e's; S_hi appended as an extra top coefficient, after which the ordinary division by
    // (X - z) of the extended range returns exactly w on the range; commit over the range.
    void openings_ranged(const std::vector<Term>& lin_terms, const std::vector<Term>& open_polys, const std::vector<Term>& shifted_polys, const Fr& v,
                         const Fr& zeta, Tick& tick, uint64_t* out) {
        const uint64_t hi_c = std::min<uint64_t>(hi, n + 3), width = hi_c > lo ? hi_c - lo : 0;
        auto cut = [&](const std::vector<Term>& terms) {
            std::vector<Term> o;
            for (auto& t : terms) {
                const uint64_t a = std::min(lo, t.len), b = std::min(hi_c, t.len);
                if (b > a) o.push_back({t.s, static_cast<const uint8_t*>(t.p) + a * EL, b - a});
            }
            return o;
        };
        const Fr zw = zeta * w_n;
        std::vector<Term> open_terms = lin_terms, shift_terms;         // 1 * lin + sum_i v^(i+1) p_i
        Fr c = v;
        for (auto& p : open_polys) { open_terms.push_back({c, p.p, p.len}); c = c * v; }
        c = Fr::one();
        for (auto& p : shifted_polys) { shift_terms.push_back({c, p.p, p.len}); c = c * v; }
        // batch.p: the open batch's range, then ONE carried coefficient; lin.p: the same for the shifted batch
        void* bufs[2] = {batch.p, lin.p};
        const std::vector<Term> cuts[2] = {cut(open_terms), cut(shift_terms)};
        const Fr points[2] = {zeta, zw};
        Fr e[2] = {Fr::zero(), Fr::zero()};
        for (int j = 0; j < 2; j++) {
            if (!width) continue;
            if (cuts[j].empty()) ck(mzk_dev_memset(bufs[j], 0, width * EL, S));
            else lincomb_many(cuts[j], bufs[j], width);
            e[j] = evaluate(bufs[j], width, 1, width, points[j])[0];
        }
        std::vector<Fr> every((size_t)2 * world);
        if (comm.all_gather(comm.ctx, e, sizeof e, every.data())) fail(MZK_ERR_INVALID_ARG, "mzk_comm.all_gather failed");
        st.batch_at_zeta = Fr::zero();                                  // the open batch polynomial's value at zeta: every rank's range value times zeta^lo
        for (int q = 0; q < world; q++)
            st.batch_at_zeta = st.batch_at_zeta + pow_u64(zeta, std::min<uint64_t>(shard_range(n + 3, q, world).first, n + 3)) * every[2 * q];
        Fr carry[2] = {Fr::zero(), Fr::zero()};
        for (int q = rank + 1; q < world; q++) {                       // S_hi: the ranges above, shifted down to start at hi
            const uint64_t lo_q = std::min<uint64_t>(shard_range(n + 3, q, world).first, n + 3);
            for (int j = 0; j < 2; j++) carry[j] = carry[j] + pow_u64(points[j], lo_q - hi_c) * every[2 * q + j];
        }
        void* outs[2] = {opening.p, shifted.p};
        for (int j = 0; j < 2 && width; j++) {
            put_scalar(carry[j], static_cast<uint8_t*>(bufs[j]) + width * EL);                                    // the carried coefficient, without a copy
            ck(mzk_poly_div_linear_dev(CURVE, bufs[j], width + 1, points[j].l, outs[j], S));               // width coefficients: w on [lo, hi)
        }
        tick.mark("r5_polys");
        commit_slices({opening.p, shifted.p}, {width, width}, out);
        tick.mark("r5_commit");
    }
    void round5(const std::vector<ProverBase*>& inst, const uint64_t* v_p, uint64_t* out) override {
        Tick tick(*this);
        if (st.bases.size() != inst.size()) fail(MZK_ERR_STATE, "round 5 takes the instances of round 3, first instance first");
        for (ProverBase* b : inst) {
            static_cast<ProverT*>(b)->need(R4, R4 + 1, "round 5 follows round 4 of every instance");
            if (b != this) ck(mzk_stream_wait_stream(S, b->S));         // the other instances' polynomials are read on this stream
        }
        const Fr v = load(v_p), zeta = st.zeta;
        std::vector<Term> terms = quotient_lin_terms(zeta);
        for (size_t i = 0; i < inst.size(); i++) {
            ProverT* p = static_cast<ProverT*>(inst[i]);
            if (!(p->st.zeta == zeta)) fail(MZK_ERR_STATE, "round 4 of the instances used different evaluation challenges");
            const std::vector<Term> t = p->lin_poly_terms(st.bases[i]);
            terms.insert(terms.end(), t.begin(), t.end());
        }
        std::vector<Term> open_polys, shifted_polys;
        if (world > 1) {
            open_lists(open_polys, shifted_polys);
            openings_ranged(terms, open_polys, shifted_polys, v, zeta, tick, out);
        } else {
            lincomb_many(terms, lin.p, n + 3);
            open_polys.push_back({Fr::one(), lin.p, n + 3});
            for (ProverBase* b : inst) static_cast<ProverT*>(b)->open_lists(open_polys, shifted_polys);
            batched_witness(open_polys, v, zeta, opening, rem_dev());
            batched_witness(shifted_polys, v, zeta * w_n, shifted);
            tick.mark("r5_polys");
            commit({opening.p, shifted.p}, {n + 2, n + 2}, out);
            tick.mark("r5_commit");
            st.batch_at_zeta = download_fr(rem_dev());                  // (the commitments have synchronised the stream)
        }
        for (ProverBase* b : inst) b->stage = CREATED;
        // t(X) Z_H(X) = numerator(X), checked at the evaluation challenge the way the verifier will check it (verifier.rs:186-231, 340-414):
        // the opening proof's batch polynomial lin + sum_i v^i p_i must take the value -r_0 + sum_i v^i p_i(zeta) at zeta, and its value there
        // is the remainder its division by (X - zeta) leaves: one 32-byte read.  The guard against an unsatisfied witness where the top
        // coefficients of the quotient come from its numerator -- reported under the reference's error name (prover.rs:915-918).
        Fr lin_constant = Fr::zero();
        std::vector<Fr> opened;
        for (size_t i = 0; i < inst.size(); i++) {
            ProverT* p = static_cast<ProverT*>(inst[i]);
            lin_constant = lin_constant + p->lin_poly_constant(st.bases[i]);
            p->opened_evals(opened);
        }
        Fr want = neg(lin_constant), c = Fr::one();
        for (auto& e : opened) { c = c * v; want = want + c * e; }
        if (!(st.batch_at_zeta == want))
            fail(MZK_ERR_WRONG_QUOTIENT_DEGREE, "WrongQuotientPolyDegree: the quotient identity t(X) Z_H(X) = numerator(X) does not hold at the evaluation "
                                                "challenge (the witness does not satisfy the circuit)");
    }

    void exchange_buffer(void** out_p, uint64_t* out_bytes) override {
        if (out_p) *out_p = rem.p;
        if (out_bytes) *out_bytes = classes.size() * n * EL;
    }
    void set_peer_buffers(void* const* ptrs, const int32_t* devices) override {
        peer_rem.assign(ptrs, ptrs + world);
        peer_dev.assign(devices, devices + world);
    }
    void poly_dev(uint32_t which, const void** out_p, uint64_t* out_len) override {
        if (which > (uint32_t)W) fail(MZK_ERR_INVALID_ARG, "which: 0 .. W - 1 wire polynomials, W the permutation product");
        if (stage != CREATED && stage < R1) fail(MZK_ERR_STATE, "no proof in flight");
        *out_p = row((int)which);
        *out_len = which < (uint32_t)W ? n + 2 : n + 3;
    }
};

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
