// mzk_prover.hpp -- the reference's bench circuit, PlonkKzgSnark::preprocess and ::prove in C++ above the C ABI
// (see mzk_host.hpp for the map to the reference).  One instance, TurboPlonk or UltraPlonk.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <tuple>

#include "mzk_host.hpp"

namespace mzk_host {

constexpr size_t EL = 32;                                              // bytes per scalar-field element

struct DevBuf {                                                        // memory of the device the allocating thread is bound to
    void* p = nullptr;
    size_t elems = 0;
    DevBuf() = default;
    explicit DevBuf(size_t n_elems) { alloc(n_elems); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), elems(o.elems) { o.p = nullptr; o.elems = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { if (p) (void)mzk_dev_free(p); p = o.p; elems = o.elems; o.p = nullptr; o.elems = 0; }
        return *this;
    }
    ~DevBuf() { if (p) (void)mzk_dev_free(p); }
    void alloc(size_t n_elems) {
        if (p) (void)mzk_dev_free(p);
        p = nullptr;
        elems = n_elems;
        check(mzk_dev_alloc((n_elems ? n_elems : 1) * EL, &p), "mzk_dev_alloc");
    }
    void* at(size_t idx) const { return static_cast<uint8_t*>(p) + idx * EL; }
};
struct PinnedBuf {                                                     // page-locked host memory (mzk_host_alloc): asynchronous DMA
    void* p = nullptr;
    size_t bytes = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    PinnedBuf(PinnedBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    ~PinnedBuf() { if (p) (void)mzk_host_free(p); }
    void alloc(size_t b) { if (p) (void)mzk_host_free(p); p = nullptr; bytes = b; check(mzk_host_alloc(b ? b : 1, &p), "mzk_host_alloc"); }
};

// ---- several GPUs from one process: one host thread per device, SPMD ---------------------------------
// The reference is ONE process calling `prove` once, with Rayon inside (univariate_kzg/mod.rs:125-127, prover.rs:552-562).  Here
// the G device threads of a ShardedProver all run the same `prove`, each on its own device context of libmi355zk, and meet in
// this communicator -- barriers and all-gathers of a few hundred bytes through host memory (Jacobian partials of 144 B, field
// elements of 32 B: no collective library needed); bulk data moves device to device (mzk_dev_copy_peer).
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
struct LocalComm {
    const int G;
    std::atomic<int> arrived{0};
    std::atomic<uint64_t> gen{0};
    std::atomic<bool> aborted{false};
    std::vector<std::vector<uint8_t>> slot;
    explicit LocalComm(int g) : G(g), slot(g) {}
    void reset() { arrived = 0; aborted = false; }
    void abort() { aborted = true; }
    void barrier() {
        if (G == 1) return;
        const uint64_t my = gen.load();
        if (arrived.fetch_add(1) + 1 == G) { arrived = 0; gen.fetch_add(1); return; }
        for (int spins = 0; gen.load() == my; spins++) {
            if (aborted.load()) throw std::runtime_error("another device thread failed");
            if (spins > 4000) std::this_thread::yield();
        }
    }
    // every rank's `bytes` bytes, concatenated in rank order
    std::vector<uint8_t> all_gather(int rank, const void* data, size_t bytes) {
        slot[rank].assign(static_cast<const uint8_t*>(data), static_cast<const uint8_t*>(data) + bytes);
        barrier();
        std::vector<uint8_t> out;
        out.reserve(bytes * G);
        for (int g = 0; g < G; g++) out.insert(out.end(), slot[g].begin(), slot[g].end());
        barrier();                                                     // nobody overwrites a slot before everybody has read it
        return out;
    }
};

template <class C>
struct BenchCircuitHost {                                              // the finalised circuit in HOST memory (what Arithmetization holds)
    using Fr = Fp64<typename C::Fr>;
    bool ultra = false;
    int log_n = 0, W = 5, nsel = 13;
    uint64_t n = 0;
    std::vector<Fr> k, wires, selectors, sigmas, tables;              // W x n, nsel x n, W x n, 4 x n (UltraPlonk)
    std::vector<Fr> witness;                                           // the witness vector ...
    std::vector<uint32_t> wire_variables;                              // ... and the variable on every (wire, row): wires[i] = witness[wire_variables[i]]

    // plonk/benches/bench.rs:29-46 through PlonkCircuit::new, Circuit::add and finalize_for_arithmetization
    static BenchCircuitHost generate(uint64_t num_gates, bool ultra, int range_bit_len = 8) {
        BenchCircuitHost cs;
        cs.ultra = ultra;
        cs.W = ultra ? 6 : 5;
        cs.nsel = ultra ? 14 : 13;
        const uint64_t n_add = num_gates - 10, used = 2 + n_add;
        const uint64_t need = ultra ? std::max<uint64_t>(used, (1ull << range_bit_len) + 1) : used;
        cs.log_n = 0;
        while ((1ull << cs.log_n) < need) cs.log_n++;
        const uint64_t n = cs.n = 1ull << cs.log_n;
        const int W = cs.W;
        // variable on every (wire, row): 0 = zero, 1 = one, 2 + t = output of the t-th addition
        std::vector<uint32_t> var((size_t)W * n, 0);
        for (uint64_t t = 0; t < n_add; t++) {
            const uint64_t row = 2 + t;
            var[0 * n + row] = t == 0 ? 0 : (uint32_t)(1 + t);            // a: the previous sum (the first one is `zero`)
            var[1 * n + row] = 1;                                         // b = one
            var[4 * n + row] = (uint32_t)(2 + t);                         // c
        }
        var[4 * n + 1] = 1;                                               // constant gate of `one`
        const size_t n_vars = 2 + n_add;
        std::vector<Fr> witness(n_vars);
        const Fr one = Fr::one();
        witness[0] = Fr::zero();                                          // 0, 1, then the running sums 1, 2, 3, ...: witness[2 + t] = t + 1
        witness[1] = one;
        if (n_vars > 2) witness[2] = one;
        for (size_t v = 3; v < n_vars; v++) witness[v] = witness[v - 1] + one;
        cs.wires.resize((size_t)W * n);
        for (size_t i = 0; i < cs.wires.size(); i++) cs.wires[i] = witness[var[i]];
        cs.witness = witness;
        cs.wire_variables = var;
        // selectors: AdditionGate q_lc = [1,1,0,0], q_o = 1; ConstantGate q_c = value, q_o = 1; PaddingGate all zero
        std::vector<Fr>& sel = cs.selectors;
        sel.assign((size_t)cs.nsel * n, Fr::zero());
        for (uint64_t row = 2; row < 2 + n_add; row++) sel[0 * n + row] = sel[1 * n + row] = one;
        for (uint64_t row = 0; row < 2 + n_add; row++) sel[10 * n + row] = one;
        sel[11 * n + 1] = one;
        // wire permutation (constraint_system.rs:743-778): the occurrences of a variable, in (wire, row) order, form a cycle
        std::vector<uint32_t> cnt(n_vars + 1, 0);
        for (uint32_t v : var) cnt[v + 1]++;
        for (size_t v = 0; v < n_vars; v++) cnt[v + 1] += cnt[v];
        std::vector<uint32_t> cells(var.size()), fill(cnt.begin(), cnt.end() - 1);
        for (uint32_t cell = 0; cell < var.size(); cell++) cells[fill[var[cell]]++] = cell;
        std::vector<uint32_t> perm(var.size());
        for (size_t v = 0; v < n_vars; v++)
            for (uint32_t i = cnt[v]; i < cnt[v + 1]; i++) perm[cells[i]] = cells[i + 1 < cnt[v + 1] ? i + 1 : cnt[v]];
        // extended identity k_i * w^j (:913-931) and sigma = id o perm
        cs.k = compute_coset_representatives<typename C::Fr>(W, n);
        const Fr w = root_of_unity<typename C::Fr>(cs.log_n);
        std::vector<Fr> ext((size_t)W * n);
        Fr cur = Fr::one();
        for (uint64_t j = 0; j < n; j++) { ext[j] = cur; cur = cur * w; }
        for (int i = 1; i < W; i++)
            for (uint64_t j = 0; j < n; j++) ext[(size_t)i * n + j] = cs.k[i] * ext[j];
        cs.sigmas.resize((size_t)W * n);
        for (size_t i = 0; i < cs.sigmas.size(); i++) cs.sigmas[i] = ext[perm[i]];
        if (ultra) {
            cs.tables.assign((size_t)4 * n, Fr::zero());
            Fr v = Fr::zero();
            for (uint64_t i = 0; i < (1ull << range_bit_len); i++) { cs.tables[i] = v; v = v + one; }   // compute_range_table (:1423-1438)
        }
        return cs;
    }
};

template <class C>
struct BenchCircuit {                                                  // ... and on one device (the thread's current one)
    using Fr = Fp64<typename C::Fr>;
    bool ultra = false;
    int log_n = 0, W = 5, nsel = 13;
    uint64_t n = 0;
    std::vector<Fr> k;
    DevBuf wire_values, selector_values, sigma_values, table_values;   // W x n, nsel x n, W x n, 4 x n (UltraPlonk)
    // host_witness: every proof starts from HOST memory (page-locked), as the reference holds its witness (constraint_system.rs:1225-1247).
    //   1: the gathered wire table W x n (what a host that gathers itself hands over): round 1 uploads column k + 1 under the iNTT of column k
    //   2: the witness VECTOR (n_vars x 32 B); the variable-index table is resident circuit structure and the gather runs on the device
    PinnedBuf host_wires, host_vars;
    DevBuf wire_variables;                                             // W x n u32
    uint64_t n_vars = 0;
    int host_witness = 0;

    static BenchCircuit upload(const BenchCircuitHost<C>& h, int host_witness = 0) {
        BenchCircuit cs;
        cs.ultra = h.ultra; cs.log_n = h.log_n; cs.W = h.W; cs.nsel = h.nsel; cs.n = h.n; cs.k = h.k;
        auto up = [](DevBuf& d, const std::vector<Fr>& v, const char* what) {
            d.alloc(v.size());
            check(mzk_dev_upload(d.p, v.data(), v.size() * EL), what);
        };
        up(cs.wire_values, h.wires, "upload wires");
        up(cs.selector_values, h.selectors, "upload selectors");
        up(cs.sigma_values, h.sigmas, "upload sigma");
        if (h.ultra) up(cs.table_values, h.tables, "upload tables");
        cs.host_witness = host_witness;
        if (host_witness == 1) {
            cs.host_wires.alloc(h.wires.size() * EL);
            std::memcpy(cs.host_wires.p, h.wires.data(), h.wires.size() * EL);
        } else if (host_witness == 2) {
            cs.n_vars = h.witness.size();
            cs.host_vars.alloc(h.witness.size() * EL);
            std::memcpy(cs.host_vars.p, h.witness.data(), h.witness.size() * EL);
            cs.wire_variables.alloc((h.wire_variables.size() * 4 + EL - 1) / EL);
            check(mzk_dev_upload(cs.wire_variables.p, h.wire_variables.data(), h.wire_variables.size() * 4), "upload wire variables");
        }
        return cs;
    }
    static BenchCircuit generate(uint64_t num_gates, bool ultra, int range_bit_len = 8) {
        return upload(BenchCircuitHost<C>::generate(num_gates, ultra, range_bit_len));
    }
};

template <class C>
struct Proof {                                                         // structs.rs:59-84 (+ PlookupProof :208-222)
    using E = Encoding<C>;
    using Fr = typename E::Fr;
    using Affine = typename E::Affine;
    std::vector<Affine> wires_poly_comms, split_quot_poly_comms, h_poly_comms;
    Affine prod_perm_poly_comm, opening_proof, shifted_opening_proof, prod_lookup_poly_comm;
    std::vector<Fr> wires_evals, wire_sigma_evals, plookup_evals;      // plookup_evals in the field order of PlookupEvaluations (:496-541)
    Fr perm_next_eval;
    bool has_plookup = false;

    std::vector<uint8_t> serialize_compressed() const {
        std::vector<uint8_t> out;
        auto u64le = [&](uint64_t v) { for (int i = 0; i < 8; i++) out.push_back((uint8_t)(v >> (8 * i))); };
        auto g1 = [&](const Affine& p) { uint8_t b[48]; E::g1_bytes(p, b); out.insert(out.end(), b, b + C::G1_BYTES); };
        auto fr = [&](const Fr& v) { uint8_t b[32]; E::fr_bytes(v, b); out.insert(out.end(), b, b + 32); };
        u64le(wires_poly_comms.size()); for (auto& p : wires_poly_comms) g1(p);
        g1(prod_perm_poly_comm);
        u64le(split_quot_poly_comms.size()); for (auto& p : split_quot_poly_comms) g1(p);
        g1(opening_proof); g1(shifted_opening_proof);
        u64le(wires_evals.size()); for (auto& v : wires_evals) fr(v);
        u64le(wire_sigma_evals.size()); for (auto& v : wire_sigma_evals) fr(v);
        fr(perm_next_eval);
        out.push_back(has_plookup ? 1 : 0);
        if (has_plookup) {
            u64le(h_poly_comms.size()); for (auto& p : h_poly_comms) g1(p);
            g1(prod_lookup_poly_comm);
            for (auto& v : plookup_evals) fr(v);
        }
        return out;
    }
};

// indices into Proof::plookup_evals (declaration order of PlookupEvaluations)
enum PlookupEval { RANGE_TABLE, KEY_TABLE, TABLE_DOM_SEP, Q_DOM_SEP, H_1, Q_LOOKUP, PROD_NEXT, RANGE_TABLE_NEXT, KEY_TABLE_NEXT, TABLE_DOM_SEP_NEXT,
                   H_1_NEXT, H_2_NEXT, Q_LOOKUP_NEXT, W_3_NEXT, W_4_NEXT, N_PLOOKUP_EVALS };

template <class C>
struct Prover {                                                        // ProvingKey on ONE device + Prover of prover.rs
    using E = Encoding<C>;
    using Fr = typename E::Fr;
    using Affine = typename E::Affine;
    using FrP = typename C::Fr;
    static constexpr int QL = E::QL;

    bool ultra;
    int log_n, W, nsel, rows;
    uint64_t n, m;
    std::vector<Fr> k;
    uint64_t srs = 0, pk = 0;
    // optional: a commit key over the Lagrange basis of the gate domain, n + 3 points (mzk_srs_generate_lagrange_for_testing /
    // mzk_srs_lagrange_from_srs): round 1 then commits the wires from their VALUES (plus the two blinders), the same group elements
    // as the commitments of the masked coefficient forms, with scalars that are mostly small numbers
    uint64_t srs_lagrange = 0;
    DevBuf vals_ext;
    // several devices: this prover is rank `rank` of `world`; it commits over the SRS points [lo, hi) of every
    // polynomial (one fixed partition of the n + 3 powers), owns the residue classes `own` of the quotient domain, and runs rounds
    // 4-5 on its coefficient range.  world == 1: lo = 0, hi = n + 3, every class.
    int rank = 0, world = 1;
    LocalComm* comm = nullptr;
    uint64_t lo = 0, hi = 0;
    DevBuf fixed;                                                      // (nsel + W [+ 4]) x n coefficient forms
    DevBuf slab, quot, coeff, split, lin, batch, opening, shifted, hh, table, lookup, sorted, tmp, deg, rem, wv, wit, top;
    bool use_top = false;                                              // W classes + mzk_plonk_quotient_top_dev (setup)
    Fr batch_at_zeta;                                                  // value of the opening batch polynomial at zeta (check_quotient_identity)
    std::vector<uint32_t> classes, own;                                // the classes that determine the quotient; this rank's share of them
    std::vector<void*> peer_rem;                                       // `rem` of every rank (device pointers), for the one exchange
    void* copy_stream = nullptr;
    bool one_ready = false;
    std::vector<Affine> selector_comms, sigma_comms;
    std::map<std::string, double> timings_ms;
    Fr w_n, gen;

    // PlonkKzgSnark::preprocess (snark.rs:529-617)
    Prover(uint64_t srs_handle, const BenchCircuit<C>& cs, int rank_ = 0, int world_ = 1, LocalComm* comm_ = nullptr)
        : ultra(cs.ultra), log_n(cs.log_n), W(cs.W), nsel(cs.nsel), rows(cs.W + 2 + (cs.ultra ? 3 : 0)), n(cs.n), m(8 * cs.n), k(cs.k), srs(srs_handle),
          rank(rank_), world(world_), comm(comm_) {
        uint64_t srs_len = 0;
        check(mzk_srs_len(srs, &srs_len), "mzk_srs_len");
        if (srs_len < n + 3) throw std::runtime_error("SRS too small: need domain size + 3 powers (srs.rs:88)");
        std::tie(lo, hi) = shard_range(n + 3, rank, world);            // the proving key keeps trim(n + 2) = n + 3 powers (snark.rs:535, 561)
        const int nfix = nsel + W + (ultra ? 4 : 0);
        fixed.alloc((size_t)nfix * n);
        check(mzk_dev_copy(fixed.p, cs.selector_values.p, (size_t)nsel * n * EL, nullptr), "copy");
        check(mzk_dev_copy(fixed.at((size_t)nsel * n), cs.sigma_values.p, (size_t)W * n * EL, nullptr), "copy");
        if (ultra) check(mzk_dev_copy(fixed.at((size_t)(nsel + W) * n), cs.table_values.p, (size_t)4 * n * EL, nullptr), "copy");
        check(mzk_ntt_dev(C::ID, fixed.p, n, log_n, 1, nullptr, nfix, n, nullptr), "mzk_ntt_dev");        // selector / sigma / table polynomials
        std::vector<uint64_t> host((size_t)nfix * n * 4);
        check(mzk_dev_download(host.data(), fixed.p, host.size() * 8), "download");
        std::vector<uint64_t> kk((size_t)W * 4);
        for (int i = 0; i < W; i++) std::memcpy(&kk[4 * i], k[i].l, 32);
        const uint64_t* sel = host.data();
        const uint64_t* sig = sel + (size_t)nsel * n * 4;
        // The quotient has degree W (n + 1) + 2 (prover.rs:916-919).  Its W + 3 coefficients from X^(Wn) on are the top coefficients of
        // its numerator (mzk_plonk_quotient_top_dev, n > W + 2), so W of the 8 residue classes of the quotient domain determine the
        // rest -- 5 for TurboPlonk, 6 for UltraPlonk -- and only those are resident and evaluated; the polynomial so recovered has the
        // expected degree whatever the witness, which is why check_quotient_identity exists.  Tiny domains: W + 1 classes with one spare
        // coefficient above the expected degree (or an unsatisfied witness could not trip WrongQuotientPolyDegree), else all 8.
        classes.clear();
        use_top = n > (uint64_t)W + 2 && n >= 8;
        const uint32_t needed = use_top ? (uint32_t)W : (((uint64_t)W * (n + 1) + 2 < (uint64_t)(W + 1) * n - 1 && W + 1 <= 8) ? (uint32_t)W + 1 : 8u);
        for (uint32_t kcl = 0; kcl < needed; kcl++) classes.push_back(kcl);
        own = class_range(rank, world, needed);                          // contiguous blocks of ceil(needed / world); the last ranks may own none
        // a rank that owns no class still registers one (a key cannot be empty); it is never evaluated
        const std::vector<uint32_t> resident = own.empty() ? std::vector<uint32_t>{classes.back()} : own;
        check(mzk_plonk_pk_register_chunked(C::ID, log_n, W, sel, sig, ultra ? sig + (size_t)W * n * 4 : nullptr, n, kk.data(), resident.data(),
                                            (uint32_t)resident.size(), &pk), "mzk_plonk_pk_register_chunked");
        rem.alloc(classes.size() * n);                                   // the remainders of ALL needed classes: own ones computed here, the others received
        top.alloc(16);                                                   // the quotient's top W + 3 coefficients (use_top)
        // the class-wise quotient reads the coefficient rows without overwriting them: n + 3 columns per row, no second copy
        slab.alloc((size_t)rows * (n + 3)); quot.alloc(m); coeff.alloc((size_t)(W + 1) * n);
        split.alloc((size_t)W * (n + 3)); lin.alloc(n + 3); batch.alloc(n + 4); opening.alloc(n + 3); shifted.alloc(n + 3); tmp.alloc(64);
        if (ultra) { hh.alloc(2 * n); table.alloc(n); lookup.alloc(n); sorted.alloc(2 * n); }
        w_n = root_of_unity<FrP>(log_n);
        gen = Fr::from_words(FrP::GENERATOR);
        // verifying-key commitments (snark.rs:562-594): whole MSMs on this device (set-up work, replicated on every rank)
        std::vector<const void*> ptrs;
        std::vector<uint64_t> lens;
        for (int i = 0; i < nsel + W; i++) { ptrs.push_back(fixed.at((size_t)i * n)); lens.push_back(n); }
        auto comms = to_affine(msm_partials(ptrs, lens, 0, n + 3));
        selector_comms.assign(comms.begin(), comms.begin() + nsel);
        sigma_comms.assign(comms.begin() + nsel, comms.end());
    }
    ~Prover() {
        if (pk) (void)mzk_plonk_pk_release(pk);
        if (copy_stream) (void)mzk_stream_destroy(copy_stream);
    }

    // Jacobian sums of the coefficients [a, b) of every polynomial over the SRS points of the same indices
    std::vector<uint64_t> msm_partials(const std::vector<const void*>& polys, const std::vector<uint64_t>& lens, uint64_t a, uint64_t b, uint64_t key = 0) {
        const uint32_t kpolys = (uint32_t)polys.size();
        std::vector<const void*> p(kpolys);
        std::vector<uint64_t> l(kpolys), off(kpolys), xyz((size_t)kpolys * 3 * QL);
        for (uint32_t i = 0; i < kpolys; i++) {
            const uint64_t s0 = std::min(a, lens[i]), s1 = std::min(b, lens[i]);
            p[i] = static_cast<const uint8_t*>(polys[i]) + s0 * EL;
            l[i] = s1 - s0;
            off[i] = s1 > s0 ? s0 : a;
        }
        check(mzk_msm_batch_dev(key ? key : srs, kpolys, p.data(), l.data(), off.data(), 1, xyz.data(), nullptr), "mzk_msm_batch_dev");
        return xyz;
    }
    std::vector<Affine> to_affine(const std::vector<uint64_t>& xyz) const {
        const size_t kpolys = xyz.size() / (3 * QL);
        std::vector<uint64_t> xy(kpolys * 2 * QL);
        check(mzk_g1_jacobian_to_affine(C::ID, xyz.data(), kpolys, xy.data()), "mzk_g1_jacobian_to_affine");
        std::vector<Affine> out(kpolys);
        for (size_t i = 0; i < kpolys; i++) std::memcpy(out[i].data(), &xy[i * 2 * QL], sizeof(Affine));
        return out;
    }
    // the ranks' partial sums -> the commitments, identical on every rank: all-gather of k x 144 B (96 B on BN254) through host
    // memory and <= 8 EC additions per commitment on the host (mzk_g1_sum_jacobian) -- the "all-reduce of partial EC sums"
    std::vector<Affine> combine_partials(const std::vector<uint64_t>& part) {
        if (world == 1) return to_affine(part);
        const size_t kpolys = part.size() / (3 * QL), one = 3 * QL;
        const std::vector<uint8_t> all = comm->all_gather(rank, part.data(), part.size() * 8);
        const uint64_t* a = reinterpret_cast<const uint64_t*>(all.data());
        std::vector<uint64_t> sum(kpolys * one), col((size_t)world * one);
        for (size_t i = 0; i < kpolys; i++) {
            for (int g = 0; g < world; g++) std::memcpy(&col[g * one], a + ((size_t)g * kpolys + i) * one, one * 8);
            check(mzk_g1_sum_jacobian(C::ID, col.data(), world, &sum[i * one]), "mzk_g1_sum_jacobian");
        }
        return to_affine(sum);
    }
    // UnivariateKzgPCS::batch_commit (mod.rs:119-131) on device-resident coefficient vectors; over several ranks every MSM is
    // sharded by point range (this rank: [lo, hi))
    std::vector<Affine> commit(const std::vector<const void*>& polys, const std::vector<uint64_t>& lens, uint64_t key = 0) {
        return combine_partials(msm_partials(polys, lens, lo, hi, key));
    }
    // ... of polynomials of which this rank holds ONLY the coefficients [lo, lo + lens[i])
    std::vector<Affine> commit_slices(const std::vector<const void*>& slices, const std::vector<uint64_t>& lens) {
        const uint32_t kpolys = (uint32_t)slices.size();
        std::vector<uint64_t> off(kpolys, lo), xyz((size_t)kpolys * 3 * QL);
        check(mzk_msm_batch_dev(srs, kpolys, slices.data(), lens.data(), off.data(), 1, xyz.data(), nullptr), "mzk_msm_batch_dev");
        return combine_partials(xyz);
    }
    std::vector<Fr> evaluate(const void* d, uint64_t len, uint32_t batch_n, uint64_t stride, const Fr& x) {
        std::vector<Fr> out(batch_n);
        check(mzk_poly_eval_dev(C::ID, d, len, batch_n, stride, x.l, reinterpret_cast<uint64_t*>(out.data()), nullptr), "mzk_poly_eval_dev");
        return out;
    }
    // evaluations of one round, collected and finished together.  Over several ranks every rank evaluates its coefficient range
    // [lo, hi) of each polynomial -- sum_{j in range} c_j x^j = x^lo * (the range read as a polynomial of its own) -- and ONE
    // all-gather of the partial values (32 bytes each) at the end of the round gives every rank all the sums.
    struct EvalBatch {
        Prover& P;
        std::vector<Fr> vals;
        explicit EvalBatch(Prover& p) : P(p) {}
        size_t add(const void* d, uint64_t len, uint32_t batch_n, uint64_t stride, const Fr& x) {
            const size_t at = vals.size();
            if (P.world == 1) {
                for (auto& v : P.evaluate(d, len, batch_n, stride, x)) vals.push_back(v);
                return at;
            }
            const uint64_t a = std::min(P.lo, len), b = std::min(P.hi, len);
            if (b > a) {
                const Fr xa = pow_u64(x, a);
                for (auto& v : P.evaluate(static_cast<const uint8_t*>(d) + a * EL, b - a, batch_n, stride, x)) vals.push_back(v * xa);
            } else {
                for (uint32_t i = 0; i < batch_n; i++) vals.push_back(Fr::zero());
            }
            return at;
        }
        void finish() {
            if (P.world == 1) return;
            const std::vector<uint8_t> all = P.comm->all_gather(P.rank, vals.data(), vals.size() * sizeof(Fr));
            const Fr* a = reinterpret_cast<const Fr*>(all.data());
            const size_t cnt = vals.size();
            for (size_t i = 0; i < cnt; i++) {
                Fr sum = Fr::zero();
                for (int g = 0; g < P.world; g++) sum = sum + a[(size_t)g * cnt + i];
                vals[i] = sum;
            }
        }
    };
    struct Term { Fr s; const void* p; uint64_t len; };
    void lincomb(const std::vector<Term>& terms, void* out, uint64_t out_len) {
        std::vector<const void*> ptrs;
        std::vector<uint64_t> lens, sc;
        for (auto& t : terms) { ptrs.push_back(t.p); lens.push_back(t.len); for (int i = 0; i < 4; i++) sc.push_back(t.s.l[i]); }
        check(mzk_poly_lincomb_dev(C::ID, (uint32_t)terms.size(), ptrs.data(), lens.data(), sc.data(), out, out_len, nullptr), "mzk_poly_lincomb_dev");
    }
    void mask(const std::vector<int>& slab_rows, const std::vector<std::vector<Fr>>& blinders) {     // prover.rs:463-486
        std::vector<void*> ptrs;
        std::vector<uint64_t> b;
        for (size_t i = 0; i < slab_rows.size(); i++) {
            ptrs.push_back(row(slab_rows[i]));
            for (auto& v : blinders[i]) for (int q = 0; q < 4; q++) b.push_back(v.l[q]);
        }
        check(mzk_poly_mask_dev(C::ID, (uint32_t)ptrs.size(), ptrs.data(), n, (uint32_t)blinders[0].size(), b.data(), nullptr), "mzk_poly_mask_dev");
    }
    struct Tick {
        std::map<std::string, double>& t; bool on; std::chrono::steady_clock::time_point t0;
        Tick(std::map<std::string, double>& tt, bool o) : t(tt), on(o) { reset(); }
        void reset() { if (on) { (void)mzk_dev_sync(); t0 = std::chrono::steady_clock::now(); } }
        void mark(const char* name) { if (on) { (void)mzk_dev_sync(); t[name] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); reset(); } }
    };

    // ---- the rounds of one instance as stages, so that batch_prove can interleave several instances the way
    // ---- batch_prove_internal does (snark.rs:263-431).  `st` carries what Oracles + Challenges carry for the instance in flight.
    struct Blinds { std::vector<std::vector<Fr>> wires, h; std::vector<Fr> z, pl; };
    struct State {
        Blinds b;
        const BenchCircuit<C>* cs = nullptr;
        const void* wire_values = nullptr;                               // W x n wire evaluations on this device
        Fr tau, beta, gamma, alpha, zeta;
        std::vector<Fr> wires_evals, wire_sigma_evals, plookup_evals;
        Fr perm_next_eval, pi_eval;
    } st;
    int rowZ() const { return W; }
    int rowPI() const { return W + 1; }
    int rowH1() const { return W + 2; }
    int rowPL() const { return W + 4; }
    // one slab: rows 0..W-1 wires, W z, W+1 public input (, h_1, h_2, Plookup product), n + 3 coefficient slots each; the rounds read
    // their polynomials from here from round 1 to the openings
    void* row(int r) const { return slab.at((size_t)r * (n + 3)); }
    void* krow(int r) const { return row(r); }
    void* fix(int r) const { return fixed.at((size_t)r * n); }

    void append_vk_and_pub_input(StandardTranscript<C>& tr) const {       // transcript/mod.rs:45-104; the bench circuit has no public input
        tr.append_u32("field size in bits", (uint32_t)FrP::BITS);
        tr.append_u64("domain size", n);
        tr.append_u64("input size", 0);
        for (auto& ki : k) tr.append_fr("wire subsets separators", ki);
        for (auto& cm : selector_comms) tr.append_commitment("selector commitments", cm);
        for (auto& cm : sigma_comms) tr.append_commitment("sigma commitments", cm);
    }
    // round 1 (prover.rs:72-87)
    std::vector<Affine> round1(const BenchCircuit<C>& cs, Blinds blinds, Tick& tick) {
        st = State();
        st.cs = &cs;
        st.b = std::move(blinds);
        check(mzk_dev_memset(coeff.at((size_t)W * n), 0, n * EL, nullptr), "memset");                        // the bench circuit has no public input
        if (!cs.host_witness) {
            st.wire_values = cs.wire_values.p;
            check(mzk_dev_copy(coeff.p, cs.wire_values.p, (size_t)W * n * EL, nullptr), "copy");
            check(mzk_ntt_dev(C::ID, coeff.p, n, log_n, 1, nullptr, W, n, nullptr), "mzk_ntt_dev");   // (the zero public-input row needs no transform)
        } else if (cs.host_witness == 2) {
            // the witness vector crosses PCIe; `witness[wire_variable(i, j)]` (constraint_system.rs:1239) is gathered on the device
            if (!wv.p) { wv.alloc((size_t)W * n); wit.alloc(cs.n_vars); check(mzk_stream_create(&copy_stream), "mzk_stream_create"); }
            st.wire_values = wv.p;
            check(mzk_stream_wait_stream(copy_stream, nullptr), "wait");                                     // the previous proof is done with `wit`
            check(mzk_dev_upload_async(wit.p, cs.host_vars.p, cs.n_vars * EL, copy_stream), "upload");
            check(mzk_stream_wait_stream(nullptr, copy_stream), "wait");
            check(mzk_plonk_gather_witness_dev(wit.p, cs.n_vars, cs.wire_variables.p, (uint64_t)W * n, wv.p, nullptr), "mzk_plonk_gather_witness_dev");
            check(mzk_dev_copy(coeff.p, wv.p, (size_t)W * n * EL, nullptr), "copy");
            check(mzk_ntt_dev(C::ID, coeff.p, n, log_n, 1, nullptr, W, n, nullptr), "mzk_ntt_dev");   // (the zero public-input row needs no transform)
        } else {
            // host-resident witness (constraint_system.rs:1225-1247 gathers it on the host): column i + 1 crosses PCIe on a copy
            // stream while column i is transformed on the null stream
            if (!wv.p) { wv.alloc((size_t)W * n); check(mzk_stream_create(&copy_stream), "mzk_stream_create"); }
            st.wire_values = wv.p;
            check(mzk_stream_wait_stream(copy_stream, nullptr), "wait");                                     // the previous proof is done with `wv`
            for (int i = 0; i < W; i++) {
                check(mzk_dev_upload_async(wv.at((size_t)i * n), static_cast<const uint8_t*>(cs.host_wires.p) + (size_t)i * n * EL, n * EL, copy_stream), "upload");
                check(mzk_stream_wait_stream(nullptr, copy_stream), "wait");                                 // columns 0..i have arrived
                check(mzk_dev_copy(coeff.at((size_t)i * n), wv.at((size_t)i * n), n * EL, nullptr), "copy");
                check(mzk_ntt_dev(C::ID, coeff.at((size_t)i * n), n, log_n, 1, nullptr, 1, n, nullptr), "mzk_ntt_dev");
            }
        }
        for (int r = 0; r < rows; r++) check(mzk_dev_memset(static_cast<uint8_t*>(row(r)) + n * EL, 0, 3 * EL, nullptr), "memset");
        check(mzk_dev_copy2d(slab.p, (n + 3) * EL, coeff.p, n * EL, n * EL, W, nullptr), "copy2d");
        check(mzk_dev_copy(row(rowPI()), coeff.at((size_t)W * n), n * EL, nullptr), "copy");
        { std::vector<int> rs; for (int i = 0; i < W; i++) rs.push_back(i); mask(rs, st.b.wires); }
        tick.mark("r1_ntt_mask");
        std::vector<const void*> p; std::vector<uint64_t> l;
        if (srs_lagrange) {
            // sum_i v_i [L_i(beta)]g + b_0 [Z_H(beta)]g + b_1 [beta Z_H(beta)]g: rows of n + 3 slots, the values, then the blinders
            if (!vals_ext.p) vals_ext.alloc((size_t)W * (n + 3));
            check(mzk_dev_copy2d(vals_ext.p, (n + 3) * EL, st.wire_values, n * EL, n * EL, W, nullptr), "copy2d");
            for (int i = 0; i < W; i++)
                for (int j = 0; j < 2; j++) lincomb({{st.b.wires[i][j], one_dev(), 1}}, vals_ext.at((size_t)i * (n + 3) + n + j), 1);
            for (int i = 0; i < W; i++) { p.push_back(vals_ext.at((size_t)i * (n + 3))); l.push_back(n + 2); }
        } else {
            for (int i = 0; i < W; i++) { p.push_back(row(i)); l.push_back(n + 2); }
        }
        auto comms = commit(p, l, srs_lagrange);
        tick.mark("r1_commit");
        return comms;
    }
    // round 1.5 (prover.rs:89-118), UltraPlonk only
    std::vector<Affine> round1_5(const Fr& tau, Tick& tick) {
        st.tau = tau;
        if (!ultra) return {};
        const int H1 = rowH1();
        check(mzk_plookup_sorted_vec_dev(pk, st.wire_values, tau.l, table.p, lookup.p, sorted.p, nullptr), "mzk_plookup_sorted_vec_dev");
        check(mzk_dev_copy(hh.p, sorted.p, n * EL, nullptr), "copy");
        check(mzk_dev_copy(hh.at(n), sorted.at(n - 1), n * EL, nullptr), "copy");
        if (srs_lagrange) {
            // h_1, h_2 are committed from the sorted vector's VALUES (table entries and looked-up values: small numbers unless the circuit
            // looks up keyed tables) plus their three blinders, over the Lagrange-basis key -- as the wires in round 1
            if (!vals_ext.p) vals_ext.alloc((size_t)W * (n + 3));
            check(mzk_dev_copy2d(vals_ext.p, (n + 3) * EL, hh.p, n * EL, n * EL, 2, nullptr), "copy2d");
            for (int i = 0; i < 2; i++)
                for (int j = 0; j < 3; j++) lincomb({{st.b.h[i][j], one_dev(), 1}}, vals_ext.at((size_t)i * (n + 3) + n + j), 1);
        }
        check(mzk_ntt_dev(C::ID, hh.p, n, log_n, 1, nullptr, 2, n, nullptr), "mzk_ntt_dev");
        check(mzk_dev_copy2d(row(H1), (n + 3) * EL, hh.p, n * EL, n * EL, 2, nullptr), "copy2d");
        mask({H1, H1 + 1}, st.b.h);
        tick.mark("r1_5_sorted_vec");
        auto comms = srs_lagrange ? commit({vals_ext.p, vals_ext.at(n + 3)}, {n + 3, n + 3}, srs_lagrange) : commit({row(H1), row(H1 + 1)}, {n + 3, n + 3});
        tick.mark("r1_5_commit");
        return comms;
    }
    // round 2 (prover.rs:125-141)
    Affine round2(const Fr& beta, const Fr& gamma, Tick& tick) {
        st.beta = beta; st.gamma = gamma;
        check(mzk_plonk_perm_product_dev(pk, st.wire_values, beta.l, gamma.l, coeff.p, nullptr), "mzk_plonk_perm_product_dev");
        check(mzk_dev_copy(row(rowZ()), coeff.p, n * EL, nullptr), "copy");
        mask({rowZ()}, {st.b.z});
        tick.mark("r2_product");
        const Affine cm = commit({row(rowZ())}, {n + 3})[0];
        tick.mark("r2_commit");
        return cm;
    }
    // round 2.5 (prover.rs:143-183), UltraPlonk only
    Affine round2_5(Tick& tick) {
        check(mzk_plookup_product_dev(pk, table.p, lookup.p, sorted.p, st.beta.l, st.gamma.l, coeff.p, nullptr), "mzk_plookup_product_dev");
        check(mzk_dev_copy(row(rowPL()), coeff.p, n * EL, nullptr), "copy");
        mask({rowPL()}, {st.b.pl});
        tick.mark("r2_5_product");
        const Affine cm = commit({row(rowPL())}, {n + 3})[0];
        tick.mark("r2_5_commit");
        return cm;
    }
    // this instance's quotient polynomial, 8n coefficients into `quot` (prover.rs:512-673 without the sum over instances)
    void quotient(const Fr& alpha, Tick& tick) {
        st.alpha = alpha;
        // per OWN class: fold mod X^n - h_k^n, size-n coset NTTs, the fused kernel, size-n inverse coset NTT -> t mod (X^n - h_k^n), straight
        // into this class's slot of `rem` (the rows of the slab are read, not overwritten)
        if (!own.empty())
            check(mzk_plonk_quotient_chunked_flags_dev(pk, slab.p, n + 3, n + 3, MZK_QUOTIENT_PI_ZERO /* the bench circuit has no public input: round1 */,
                                                       ultra ? st.tau.l : nullptr, alpha.l, st.beta.l, st.gamma.l, rem.at((size_t)own[0] * n), nullptr),
                  "mzk_plonk_quotient_chunked_flags_dev");
        if (world > 1) {
            // THE one exchange: every rank pushes its class remainders into the same slots of every other rank's
            // `rem`, device to device (xGMI peer copies; n x 32 B per class and peer), then all ranks meet
            if (!own.empty())
                for (int q = 0; q < world; q++)
                    if (q != rank)
                        check(mzk_dev_copy_peer(static_cast<uint8_t*>(peer_rem[q]) + (size_t)own[0] * n * EL, q, rem.at((size_t)own[0] * n), rank,
                                                own.size() * n * EL, nullptr), "mzk_dev_copy_peer");
            check(mzk_dev_sync(), "mzk_dev_sync");
            comm->barrier();
        }
        // the inverse Vandermonde per coefficient index (replicated: every rank needs the quotient's coefficients for the split)
        if (use_top) {
            check(mzk_plonk_quotient_top_dev(pk, slab.p, n + 3, n + 3, alpha.l, st.beta.l, st.gamma.l, top.p, nullptr, nullptr), "mzk_plonk_quotient_top_dev");
            check(mzk_plonk_quotient_combine_top_dev(C::ID, log_n, classes.data(), (uint32_t)classes.size(), rem.p, top.p, (uint32_t)W + 3, quot.p, nullptr),
                  "mzk_plonk_quotient_combine_top_dev");
        } else {
            check(mzk_plonk_quotient_combine_classes_dev(C::ID, log_n, classes.data(), (uint32_t)classes.size(), rem.p, quot.p, nullptr),
                  "mzk_plonk_quotient_combine_classes_dev");
        }
        tick.mark("r3_quotient");
    }
    // split_quotient_polynomial (prover.rs:902-960) of the 8n coefficients at `q` into this->split; returns the W lengths
    std::vector<uint64_t> split_quotient(const void* q, const std::vector<Fr>& b_quot) {
        const uint64_t expected = (uint64_t)W * (n + 1) + 2;                                                  // quotient_polynomial_degree
        if (!deg.p) deg.alloc(1);
        // only what lies at and above the expected degree is scanned: its length must be exactly 1 (read in check_quotient_degree)
        check(mzk_poly_degree_dev(static_cast<const uint8_t*>(q) + expected * EL, m - expected, static_cast<uint64_t*>(deg.p), nullptr), "mzk_poly_degree_dev");
        check(mzk_dev_memset(split.p, 0, (size_t)W * (n + 3) * EL, nullptr), "memset");
        std::vector<uint64_t> split_len(W);
        Fr last = Fr::zero();
        for (int i = 0; i < W; i++) {
            const uint64_t lo = (uint64_t)i * (n + 2), hi = i < W - 1 ? lo + n + 2 : expected + 1;
            void* p = split.at((size_t)i * (n + 3));
            check(mzk_dev_copy(p, static_cast<const uint8_t*>(q) + lo * EL, (hi - lo) * EL, nullptr), "copy");
            // the masking scalars travel as kernel arguments of mzk_poly_lincomb_dev (times the resident constant one): no host-to-device
            // copy, hence no stream synchronisation between the quotient kernels and the commitments
            if (i < W - 1) lincomb({{b_quot[i], one_dev(), 1}}, static_cast<uint8_t*>(p) + (n + 2) * EL, 1);
            if (i > 0) lincomb({{Fr::one(), p, 1}, {mzk::neg(last), one_dev(), 1}}, p, 1);                       // t_i[0] -= b_{i-1}
            if (i < W - 1) last = b_quot[i];
            split_len[i] = i < W - 1 ? n + 3 : hi - lo;
        }
        return split_len;
    }
    // quot_poly.degree() != expected_degree => WrongQuotientPolyDegree (prover.rs:915-918): the reference's only guard against an
    // unsatisfied witness.  Call after commit_split (the commitments have synchronised the stream; this reads 8 bytes).
    void check_quotient_degree() {
        uint64_t tail = 0;                                                                                   // length of quot[expected..]
        check(mzk_dev_download(&tail, deg.p, 8), "download");
        const uint64_t expected = (uint64_t)W * (n + 1) + 2;
        if (tail != 1)
            throw std::runtime_error("WrongQuotientPolyDegree: quotient polynomial of degree " +
                                     (tail ? std::to_string(expected + tail - 1) : "below " + std::to_string(expected)) + ", expected " +
                                     std::to_string(expected) + " (the witness does not satisfy the circuit)");
    }
    const void* one_dev() {                                               // the field's one (Montgomery), resident: tmp[1]
        if (!one_ready) {
            const Fr one = Fr::one();
            check(mzk_dev_upload(tmp.at(1), one.l, EL), "upload");
            one_ready = true;
        }
        return tmp.at(1);
    }
    std::vector<Affine> commit_split(const std::vector<uint64_t>& split_len) {
        std::vector<const void*> p;
        for (int i = 0; i < W; i++) p.push_back(split.at((size_t)i * (n + 3)));
        return commit(p, split_len);
    }
    // compute_evaluations / compute_plookup_evaluations (prover.rs:216-299) into st
    void round4(const Fr& zeta, Tick& tick) {
        st.zeta = zeta;
        const Fr zeta_w = zeta * w_n;
        const int sigma0 = nsel, tab0 = nsel + W, H1 = rowH1(), PL = rowPL();
        EvalBatch ev(*this);
        // the wires and, in the same launch, z and the public-input polynomial (rows W, W + 1; every row is zero above its own length):
        // pi(zeta) is not part of the proof, check_quotient_identity needs it
        const size_t h_w = ev.add(krow(0), n + 3, W + 2, n + 3, zeta);
        const size_t h_s = ev.add(fix(sigma0), n, W - 1, n, zeta);
        const size_t h_z = ev.add(krow(rowZ()), n + 3, 1, n + 3, zeta_w);
        size_t h_tz = 0, h_tn = 0, h_h1 = 0, h_ql = 0, h_qln = 0, h_pl = 0, h_hn = 0, h_wn = 0;
        if (ultra) {
            h_tz = ev.add(fix(tab0), n, 4, n, zeta);                                                           // range, key, table_dom_sep, q_dom_sep
            h_tn = ev.add(fix(tab0), n, 3, n, zeta_w);
            h_h1 = ev.add(krow(H1), n + 3, 1, n + 3, zeta);
            h_ql = ev.add(fix(13), n, 1, n, zeta);
            h_qln = ev.add(fix(13), n, 1, n, zeta_w);
            h_pl = ev.add(krow(PL), n + 3, 1, n + 3, zeta_w);
            h_hn = ev.add(krow(H1), n + 3, 2, n + 3, zeta_w);
            h_wn = ev.add(krow(3), n + 2, 2, n + 3, zeta_w);
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
        tick.mark("r4_evals");
    }
    void append_proof_evaluations(StandardTranscript<C>& tr) const {       // transcript/mod.rs:140-163
        for (auto& v : st.wires_evals) tr.append_fr("wire_evals", v);
        for (auto& v : st.wire_sigma_evals) tr.append_fr("wire_sigma_evals", v);
        tr.append_fr("perm_next_eval", st.perm_next_eval);
    }
    void append_plookup_evaluations(StandardTranscript<C>& tr) const {     // transcript/mod.rs:165-202
        if (!ultra) return;
        const std::vector<Fr>& pe = st.plookup_evals;
        tr.append_fr("lookup_table_eval", pe[RANGE_TABLE]); tr.append_fr("h_1_eval", pe[H_1]); tr.append_fr("prod_next_eval", pe[PROD_NEXT]);
        tr.append_fr("lookup_table_next_eval", pe[RANGE_TABLE_NEXT]); tr.append_fr("h_1_next_eval", pe[H_1_NEXT]);
        tr.append_fr("h_2_next_eval", pe[H_2_NEXT]);
    }
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
        terms.push_back({mzk::neg(we[4]), fix(10), n});
        terms.push_back({Fr::one(), fix(11), n});
        const Fr one = Fr::one(), nf = from_u64<FrP>(n);
        const Fr vanish = pow_u64(zeta, n) - one;
        const Fr lagrange_1 = vanish * inv(nf * (zeta - one));
        Fr cf = alpha;
        for (int j = 0; j < W; j++) cf = cf * (we[j] + beta * k[j] * zeta + gamma);
        terms.push_back({cf + alpha * alpha * lagrange_1, krow(rowZ()), n + 3});
        cf = alpha * beta * st.perm_next_eval;
        for (int j = 0; j < W - 1; j++) cf = cf * (we[j] + beta * st.wire_sigma_evals[j] + gamma);
        terms.push_back({mzk::neg(cf), fix(sigma0 + W - 1), n});
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
            terms.push_back({a4 * lagrange_1 + a5 * lagrange_n + a6 * zmg * b1 * (gamma + ml) * (g1 + mt + beta * mt_next), krow(rowPL()), n + 3});
            terms.push_back({mzk::neg(a6 * zmg * pe[PROD_NEXT] * (g1 + pe[H_1] + beta * pe[H_1_NEXT])), krow(rowH1() + 1), n + 3});
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
        Fr tmp = st.pi_eval - a2 * lagrange_1;
        Fr acc = alpha * st.perm_next_eval * (gamma + we[W - 1]);
        for (int j = 0; j < W - 1; j++) acc = acc * (gamma + we[j] + beta * st.wire_sigma_evals[j]);
        tmp = tmp - acc;
        if (ultra) {
            const Fr a3 = a2 * alpha, w_inv = inv(w_n);
            const Fr lagrange_n = vanish * w_inv * inv(nf * (zeta - w_inv));
            const Fr g1 = gamma * (one + beta);
            const Fr pc = lagrange_n * (pe[H_1] - pe[H_2_NEXT] - a2) - alpha * lagrange_1
                          - a3 * (zeta - w_inv) * pe[PROD_NEXT] * (g1 + pe[H_1] + beta * pe[H_1_NEXT]) * (g1 + beta * pe[H_2_NEXT]);
            tmp = tmp + a3 * pc;
        }
        return tmp * alpha_base;
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
    // t(X) Z_H(X) = numerator(X), checked at the evaluation challenge the way the verifier will check it (verifier.rs:186-231, 340-414): the
    // opening proof's batch polynomial lin + sum_i v^i p_i must take the value -r_0 + sum_i v^i p_i(zeta) at zeta, and its value there is
    // the remainder its division by (X - zeta) leaves (mzk_poly_div_linear_rem_dev): one 32-byte read.  The guard against an unsatisfied
    // witness where the top coefficients of the quotient come from its numerator (use_top): `WrongQuotientPolyDegree` (prover.rs:915-918)
    // cannot fire there, the recovered polynomial having the expected degree by construction -- reported under the same name.
    static void check_quotient_identity(const Fr& batch_at_zeta, const Fr& lin_constant, const std::vector<Fr>& opened, const Fr& v) {
        Fr want = mzk::neg(lin_constant), c = Fr::one();
        for (auto& e : opened) { c = c * v; want = want + c * e; }
        if (!(batch_at_zeta == want))
            throw std::runtime_error("WrongQuotientPolyDegree: the quotient identity t(X) Z_H(X) = numerator(X) does not hold at the evaluation "
                                     "challenge (the witness does not satisfy the circuit)");
    }
    // compute_quotient_component_for_lin_poly (prover.rs:343-358) over this->split
    std::vector<Term> quotient_lin_terms(const Fr& zeta, const std::vector<uint64_t>& split_len) const {
        const Fr one = Fr::one(), vanish = pow_u64(zeta, n) - one, zeta_n2 = (vanish + one) * zeta * zeta;
        std::vector<Term> terms;
        Fr cf = one;
        for (int i = 0; i < W; i++) {
            terms.push_back({mzk::neg(vanish) * cf, split.at((size_t)i * (n + 3)), split_len[i]});
            cf = cf * zeta_n2;
        }
        return terms;
    }
    // the polynomials this instance opens at zeta (after the linearisation polynomial) and at zeta * w (prover.rs:362-460)
    void open_lists(std::vector<Term>& open_polys, std::vector<Term>& shifted_polys) const {
        const Fr one = Fr::one();
        const int sigma0 = nsel, tab0 = nsel + W, H1 = rowH1(), PL = rowPL();
        for (int i = 0; i < W; i++) open_polys.push_back({one, krow(i), n + 2});
        for (int i = 0; i < W - 1; i++) open_polys.push_back({one, fix(sigma0 + i), n});
        shifted_polys.push_back({one, krow(rowZ()), n + 3});
        if (ultra) {
            for (const void* p : {(const void*)fix(tab0), (const void*)fix(tab0 + 1)}) open_polys.push_back({one, p, n});
            open_polys.push_back({one, krow(H1), n + 3});
            open_polys.push_back({one, fix(13), n});
            open_polys.push_back({one, fix(tab0 + 2), n});
            open_polys.push_back({one, fix(tab0 + 3), n});
            shifted_polys.push_back({one, krow(PL), n + 3});
            shifted_polys.push_back({one, fix(tab0), n});
            shifted_polys.push_back({one, fix(tab0 + 1), n});
            shifted_polys.push_back({one, krow(H1), n + 3});
            shifted_polys.push_back({one, krow(H1 + 1), n + 3});
            shifted_polys.push_back({one, fix(13), n});
            shifted_polys.push_back({one, krow(3), n + 2});
            shifted_polys.push_back({one, krow(4), n + 2});
            shifted_polys.push_back({one, fix(tab0 + 2), n});
        }
    }
    // sum of any number of terms into `out` (one launch takes 32)
    void lincomb_many(const std::vector<Term>& terms, void* out, uint64_t out_len) {
        constexpr size_t MAXT = 32;
        if (terms.size() <= MAXT) { lincomb(terms, out, out_len); return; }
        lincomb(std::vector<Term>(terms.begin(), terms.begin() + MAXT), out, out_len);
        for (size_t i = MAXT; i < terms.size(); i += MAXT - 1) {
            std::vector<Term> chunk{{Fr::one(), out, out_len}};
            chunk.insert(chunk.end(), terms.begin() + i, terms.begin() + std::min(terms.size(), i + MAXT - 1));
            lincomb(chunk, out, out_len);                                    // elementwise: reading out[j] before writing it is safe
        }
    }
    // compute_batched_witness_polynomial_commitment (prover.rs:490-509) up to the commitment
    void batched_witness(const std::vector<Term>& polys, const Fr& v, const Fr& point, DevBuf& out, void* d_rem = nullptr) {
        std::vector<Term> t;
        Fr c = Fr::one();
        for (auto& p : polys) { t.push_back({c, p.p, p.len}); c = c * v; }
        lincomb_many(t, batch.p, n + 3);
        if (d_rem) check(mzk_poly_div_linear_rem_dev(C::ID, batch.p, n + 3, point.l, out.p, d_rem, nullptr), "mzk_poly_div_linear_rem_dev");   // remainder = batch(point)
        else check(mzk_poly_div_linear_dev(C::ID, batch.p, n + 3, point.l, out.p, nullptr), "mzk_poly_div_linear_dev");
    }
    void* rem_dev() const { return tmp.at(2); }                          // where the opening division leaves the batch polynomial's value at zeta
    Fr download_fr(const void* d) const {
        Fr v;
        check(mzk_dev_download(v.l, d, EL), "download");
        return v;
    }
    // Round 5 over several ranks.  The opening witness of a batch polynomial b at a point z is
    // w_j = sum_{i > j} b_i z^(i-j-1).  A rank needs w on its own coefficient range [lo, hi) only -- that is its MSM shard -- and
    // w_j = (the same sum over i < hi) + z^(hi-1-j) S_hi with S_hi = sum_{i >= hi} b_i z^(i-hi): the higher ranks' contribution enters
    // as ONE field element.  So: linear combinations on the range only (they are pointwise); e = the range read as a polynomial,
    // evaluated at z; one all-gather of the e's; S_hi appended as an extra top coefficient, after which the ordinary division by
    // (X - z) of the extended range returns exactly w on the range; commit over the range.
    std::vector<Affine> openings_ranged(const std::vector<Term>& lin_terms, const std::vector<Term>& open_polys, const std::vector<Term>& shifted_polys,
                                        const Fr& v, const Fr& zeta, Tick& tick) {
        const uint64_t hi_c = std::min<uint64_t>(hi, n + 3), width = hi_c > lo ? hi_c - lo : 0;
        auto cut = [&](const std::vector<Term>& terms) {
            std::vector<Term> out;
            for (auto& t : terms) {
                const uint64_t a = std::min(lo, t.len), b = std::min(hi_c, t.len);
                if (b > a) out.push_back({t.s, static_cast<const uint8_t*>(t.p) + a * EL, b - a});
            }
            return out;
        };
        const Fr zw = zeta * w_n;
        std::vector<Term> open_terms = lin_terms, shift_terms;           // 1 * lin + sum_i v^(i+1) p_i
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
            if (cuts[j].empty()) check(mzk_dev_memset(bufs[j], 0, width * EL, nullptr), "memset");
            else lincomb_many(cuts[j], bufs[j], width);
            e[j] = evaluate(bufs[j], width, 1, width, points[j])[0];
        }
        const std::vector<uint8_t> all = comm->all_gather(rank, e, sizeof e);
        const Fr* every = reinterpret_cast<const Fr*>(all.data());
        batch_at_zeta = Fr::zero();                                      // the open batch polynomial's value at zeta: every rank's range value times zeta^lo
        for (int q = 0; q < world; q++)
            batch_at_zeta = batch_at_zeta + pow_u64(zeta, std::min<uint64_t>(shard_range(n + 3, q, world).first, n + 3)) * every[2 * q];
        Fr carry[2] = {Fr::zero(), Fr::zero()};
        for (int q = rank + 1; q < world; q++) {                         // S_hi: the ranges above, shifted down to start at hi
            const uint64_t lo_q = std::min<uint64_t>(shard_range(n + 3, q, world).first, n + 3);
            for (int j = 0; j < 2; j++) carry[j] = carry[j] + pow_u64(points[j], lo_q - hi_c) * every[2 * q + j];
        }
        void* outs[2] = {opening.p, shifted.p};
        for (int j = 0; j < 2 && width; j++) {
            lincomb({{carry[j], one_dev(), 1}}, static_cast<uint8_t*>(bufs[j]) + width * EL, 1);                 // the carried coefficient, without a copy
            check(mzk_poly_div_linear_dev(C::ID, bufs[j], width + 1, points[j].l, outs[j], nullptr), "mzk_poly_div_linear_dev");   // width coefficients: w on [lo, hi)
        }
        tick.mark("r5_polys");
        const auto oc = commit_slices({opening.p, shifted.p}, {width, width});
        tick.mark("r5_commit");
        return oc;
    }
    static Blinds draw_blinds(ChaChaRng& rng, int W, bool ultra) {         // one instance, draw order of prover.rs:79-83, 113-114, 133-138, 169-180
        auto draw = [&](int cnt) { std::vector<Fr> v; for (int i = 0; i < cnt; i++) v.push_back(fr_rand<FrP>(rng)); return v; };
        Blinds b;
        for (int i = 0; i < W; i++) b.wires.push_back(draw(2));
        if (ultra) { b.h.push_back(draw(3)); b.h.push_back(draw(3)); }
        b.z = draw(3);
        if (ultra) b.pl = draw(3);
        return b;
    }

    // PlonkKzgSnark::prove (snark.rs:624-651) -> batch_prove_internal (:201-469), one instance
    Proof<C> prove(ChaChaRng& rng, const BenchCircuit<C>& cs, bool profile = false) {
        Blinds blinds = draw_blinds(rng, W, ultra);
        std::vector<Fr> b_quot;
        for (int i = 0; i < W - 1; i++) b_quot.push_back(fr_rand<FrP>(rng));                                   // prover.rs:947-955
        return prove_with(std::move(blinds), b_quot, cs, profile);
    }
    // ... with the masking draws made by the caller: over several devices every rank runs this with the SAME draws (ShardedProver)
    Proof<C> prove_with(Blinds blinds, const std::vector<Fr>& b_quot, const BenchCircuit<C>& cs, bool profile = false) {
        Tick tick(timings_ms, profile);
        StandardTranscript<C> tr;
        append_vk_and_pub_input(tr);
        Proof<C> proof;
        proof.has_plookup = ultra;
        proof.wires_poly_comms = round1(cs, std::move(blinds), tick);
        for (auto& cm : proof.wires_poly_comms) tr.append_commitment("witness_poly_comms", cm);
        const Fr tau = tr.get_and_append_challenge("tau");
        proof.h_poly_comms = round1_5(tau, tick);
        for (auto& cm : proof.h_poly_comms) tr.append_commitment("h_poly_comms", cm);
        const Fr beta = tr.get_and_append_challenge("beta"), gamma = tr.get_and_append_challenge("gamma");
        proof.prod_perm_poly_comm = round2(beta, gamma, tick);
        tr.append_commitment("perm_poly_comms", proof.prod_perm_poly_comm);
        if (ultra) {
            proof.prod_lookup_poly_comm = round2_5(tick);
            tr.append_commitment("plookup_poly_comms", proof.prod_lookup_poly_comm);
        }
        const Fr alpha = tr.get_and_append_challenge("alpha");
        quotient(alpha, tick);
        const std::vector<uint64_t> split_len = split_quotient(quot.p, b_quot);
        tick.mark("r3_split");
        proof.split_quot_poly_comms = commit_split(split_len);
        tick.mark("r3_commit");
        check_quotient_degree();
        for (auto& cm : proof.split_quot_poly_comms) tr.append_commitment("quot_poly_comms", cm);
        const Fr zeta = tr.get_and_append_challenge("zeta");
        round4(zeta, tick);
        append_proof_evaluations(tr);
        append_plookup_evaluations(tr);
        proof.wires_evals = st.wires_evals;
        proof.wire_sigma_evals = st.wire_sigma_evals;
        proof.perm_next_eval = st.perm_next_eval;
        proof.plookup_evals = st.plookup_evals;
        const Fr v = tr.get_and_append_challenge("v");
        std::vector<Term> terms = lin_poly_terms(Fr::one());
        const std::vector<Term> qt = quotient_lin_terms(zeta, split_len);
        terms.insert(terms.end(), qt.begin(), qt.end());
        std::vector<Term> open_polys, shifted_polys;
        std::vector<Affine> oc;
        if (world > 1) {
            open_lists(open_polys, shifted_polys);
            oc = openings_ranged(terms, open_polys, shifted_polys, v, zeta, tick);
        } else {
            lincomb_many(terms, lin.p, n + 3);
            open_polys.push_back({Fr::one(), lin.p, n + 3});
            open_lists(open_polys, shifted_polys);
            batched_witness(open_polys, v, zeta, opening, rem_dev());
            batched_witness(shifted_polys, v, zeta * w_n, shifted);
            tick.mark("r5_polys");
            oc = commit({opening.p, shifted.p}, {n + 2, n + 2});
            tick.mark("r5_commit");
            batch_at_zeta = download_fr(rem_dev());                         // (the commitments have synchronised the stream)
        }
        {
            std::vector<Fr> opened;
            opened_evals(opened);
            check_quotient_identity(batch_at_zeta, lin_poly_constant(Fr::one()), opened, v);
        }
        proof.opening_proof = oc[0];
        proof.shifted_opening_proof = oc[1];
        return proof;
    }
};

// ---- G devices from one process: G host threads, each bound to its device's context of libmi355zk, all running Prover::prove_with
// ---- on the same draws (SPMD); they end with identical proofs.  Every device-side object of rank g is created, used and destroyed
// ---- on thread g.  world == 1 runs on the calling thread.
template <class C>
struct ShardedProver {
    using P = Prover<C>;
    using Fr = typename P::Fr;
    const int G;
    LocalComm comm;
    std::vector<std::unique_ptr<BenchCircuit<C>>> circuit;             // per device
    std::vector<std::unique_ptr<P>> prover;
    std::vector<uint64_t> srs, srs_lagrange;
    double lagrange_key_s = 0;                                         // wall time of mzk_srs_lagrange_from_srs on rank 0 (set-up, once per SRS and domain)
    // worker threads
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::function<void(int)> job;
    uint64_t job_gen = 0;
    int pending = 0;
    bool stop = false;
    std::vector<std::exception_ptr> errors;

    explicit ShardedProver(int g) : G(g), comm(g), circuit(g), prover(g), srs(g, 0), srs_lagrange(g, 0), errors(g) {
        if (G > 1)
            for (int r = 0; r < G; r++) threads.emplace_back([this, r] { worker(r); });
    }
    ~ShardedProver() {
        try { each([&](int r) { prover[r].reset(); circuit[r].reset(); if (srs[r]) (void)mzk_srs_release(srs[r]); srs[r] = 0;
                                    if (srs_lagrange[r]) (void)mzk_srs_release(srs_lagrange[r]); srs_lagrange[r] = 0; }); } catch (...) {}
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_job.notify_all();
        for (auto& t : threads) t.join();
    }
    void worker(int r) {
        bool bound = false;
        uint64_t seen = 0;
        for (;;) {
            std::function<void(int)> f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [&] { return stop || job_gen != seen; });
                if (stop) return;
                seen = job_gen;
                f = job;
            }
            try {
                if (!bound) { check(mzk_init(r), "mzk_init"); bound = true; }    // this thread drives device r from now on
                f(r);
            } catch (...) {
                errors[r] = std::current_exception();
                comm.abort();                                                 // ranks waiting in a barrier give up too
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--pending == 0) cv_done.notify_all();
            }
        }
    }
    // f(rank) on every device thread; returns when all are done; the first failure is rethrown
    void each(const std::function<void(int)>& f) {
        if (G == 1) { f(0); return; }
        comm.reset();
        for (auto& e : errors) e = nullptr;
        {
            std::lock_guard<std::mutex> lk(mu);
            job = f;
            pending = G;
            job_gen++;
        }
        cv_job.notify_all();
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_done.wait(lk, [&] { return pending == 0; });
        }
        std::exception_ptr other = nullptr;
        for (auto& e : errors) {                                          // prefer the root cause over "another device thread failed"
            if (!e) continue;
            try { std::rethrow_exception(e); }
            catch (const std::runtime_error& x) { if (std::string(x.what()) != "another device thread failed") std::rethrow_exception(e); other = e; }
            catch (...) { std::rethrow_exception(e); }
        }
        if (other) std::rethrow_exception(other);
    }
    // the testing SRS [beta^i] G on every device, the circuit uploaded to every device, PlonkKzgSnark::preprocess per device
    // lagrange: also the key over the Lagrange basis of the gate domain -- round 1 then commits from the wire values
    void setup(const BenchCircuitHost<C>& host, const std::array<uint64_t, 4>& beta_canonical, int host_witness = 0, bool lagrange = true) {
        each([&](int r) {
            check(mzk_srs_generate_for_testing(C::ID, beta_canonical.data(), host.n + 3, &srs[r]), "mzk_srs_generate_for_testing");
            if (lagrange) {                                               // from the SRS's points alone (no trapdoor): an inverse NTT over the group
                const auto t0 = std::chrono::steady_clock::now();
                check(mzk_srs_lagrange_from_srs(srs[r], (uint32_t)host.log_n, 3, &srs_lagrange[r]), "mzk_srs_lagrange_from_srs");
                if (r == 0) lagrange_key_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            }
            circuit[r] = std::make_unique<BenchCircuit<C>>(BenchCircuit<C>::upload(host, host_witness));
            prover[r] = std::make_unique<P>(srs[r], *circuit[r], r, G, &comm);
            prover[r]->srs_lagrange = srs_lagrange[r];
        });
        for (int r = 0; r < G; r++) {                                     // where every rank receives the class remainders of the others
            prover[r]->peer_rem.resize(G);
            for (int q = 0; q < G; q++) prover[r]->peer_rem[q] = prover[q]->rem.p;
        }
    }
    // PlonkKzgSnark::prove: the draws are made once, every rank proves with them; `check_agree`: all ranks must hold the same bytes
    Proof<C> prove(ChaChaRng& rng, bool profile = false, bool check_agree = false) {
        const typename P::Blinds blinds = P::draw_blinds(rng, prover[0]->W, prover[0]->ultra);
        std::vector<Fr> b_quot;
        for (int i = 0; i < prover[0]->W - 1; i++) b_quot.push_back(fr_rand<typename C::Fr>(rng));
        std::vector<Proof<C>> proofs(check_agree ? G : 1);
        each([&](int r) {
            Proof<C> pr = prover[r]->prove_with(blinds, b_quot, *circuit[r], profile);
            if (r == 0 || check_agree) proofs[check_agree ? r : 0] = std::move(pr);
        });
        if (check_agree)
            for (int r = 1; r < G; r++)
                if (proofs[r].serialize_compressed() != proofs[0].serialize_compressed()) throw std::runtime_error("the ranks disagree on the proof");
        return std::move(proofs[0]);
    }
    void sync() { each([&](int) { check(mzk_dev_sync(), "sync"); }); }
};

// ---- aggregated proofs over several instances (snark.rs:64-78, 201-469; structs.rs:266-291) -----------------------------
template <class C>
struct BatchProof {
    using E = Encoding<C>;
    using Fr = typename E::Fr;
    using Affine = typename E::Affine;
    struct Evals { std::vector<Fr> wires_evals, wire_sigma_evals; Fr perm_next_eval; };
    struct Plookup { bool some = false; std::vector<Affine> h_poly_comms; Affine prod_lookup_poly_comm; std::vector<Fr> poly_evals; };
    std::vector<std::vector<Affine>> wires_poly_comms_vec;
    std::vector<Affine> prod_perm_poly_comms_vec;
    std::vector<Evals> poly_evals_vec;
    std::vector<Plookup> plookup_proofs_vec;
    std::vector<Affine> split_quot_poly_comms;
    Affine opening_proof, shifted_opening_proof;

    std::vector<uint8_t> serialize_compressed() const {
        std::vector<uint8_t> out;
        auto u64le = [&](uint64_t v) { for (int i = 0; i < 8; i++) out.push_back((uint8_t)(v >> (8 * i))); };
        auto g1 = [&](const Affine& p) { uint8_t b[48]; E::g1_bytes(p, b); out.insert(out.end(), b, b + C::G1_BYTES); };
        auto fr = [&](const Fr& v) { uint8_t b[32]; E::fr_bytes(v, b); out.insert(out.end(), b, b + 32); };
        u64le(wires_poly_comms_vec.size());
        for (auto& comms : wires_poly_comms_vec) { u64le(comms.size()); for (auto& p : comms) g1(p); }
        u64le(prod_perm_poly_comms_vec.size()); for (auto& p : prod_perm_poly_comms_vec) g1(p);
        u64le(poly_evals_vec.size());
        for (auto& ev : poly_evals_vec) {
            u64le(ev.wires_evals.size()); for (auto& v : ev.wires_evals) fr(v);
            u64le(ev.wire_sigma_evals.size()); for (auto& v : ev.wire_sigma_evals) fr(v);
            fr(ev.perm_next_eval);
        }
        u64le(plookup_proofs_vec.size());
        for (auto& pl : plookup_proofs_vec) {
            out.push_back(pl.some ? 1 : 0);
            if (!pl.some) continue;
            u64le(pl.h_poly_comms.size()); for (auto& p : pl.h_poly_comms) g1(p);
            g1(pl.prod_lookup_poly_comm);
            for (auto& v : pl.poly_evals) fr(v);
        }
        u64le(split_quot_poly_comms.size()); for (auto& p : split_quot_poly_comms) g1(p);
        g1(opening_proof); g1(shifted_opening_proof);
        return out;
    }
};

// PlonkKzgSnark::batch_prove: round k of every instance, then one challenge; one quotient t = sum_k alpha_base_k t_k
// (alpha_base_{k+1} = alpha_base_k alpha^3, alpha^7 with Plookup: prover.rs:661-669) -- the per-instance quotients are
// combined after their inverse NTTs, which is the same polynomial because both maps are linear --, one split (first key's
// commit key), one linearisation polynomial and the two opening proofs over the concatenated lists (prover.rs:362-419).
template <class C>
BatchProof<C> batch_prove(ChaChaRng& rng, const std::vector<Prover<C>*>& provers, const std::vector<const BenchCircuit<C>*>& circuits) {
    using P = Prover<C>;
    using Fr = typename P::Fr;
    using FrP = typename C::Fr;
    using Term = typename P::Term;
    if (provers.empty()) throw std::runtime_error("zero number of circuits/proving keys");
    if (provers.size() != circuits.size()) throw std::runtime_error("the number of circuits != the number of proving keys");
    for (size_t i = 0; i < provers.size(); i++)
        for (size_t j = 0; j < i; j++)
            if (provers[i] == provers[j]) throw std::runtime_error("one Prover per instance: the device workspace belongs to the Prover");
    P& p0 = *provers[0];
    const uint64_t n = p0.n;
    const int W = p0.W;
    for (size_t i = 0; i < provers.size(); i++) {
        if (circuits[i]->n != n) throw std::runtime_error("circuit domain size != expected domain size");
        if (provers[i]->n != n) throw std::runtime_error("proving key domain size != expected domain size");
        if (circuits[i]->ultra != provers[i]->ultra) throw std::runtime_error("Mismatched Plonk types between the proving key and the circuit");
        if (provers[i]->W != W) throw std::runtime_error("inconsistent plonk circuit types");
    }
    const size_t K = provers.size();
    std::map<std::string, double> unused;
    typename P::Tick tick(unused, false);
    // prng draws in the reference's order (snark.rs:277-360)
    auto draw = [&](int cnt) { std::vector<Fr> v; for (int i = 0; i < cnt; i++) v.push_back(fr_rand<FrP>(rng)); return v; };
    std::vector<typename P::Blinds> blinds(K);
    for (size_t i = 0; i < K; i++) for (int j = 0; j < W; j++) blinds[i].wires.push_back(draw(2));
    for (size_t i = 0; i < K; i++) if (provers[i]->ultra) { blinds[i].h.push_back(draw(3)); blinds[i].h.push_back(draw(3)); }
    for (size_t i = 0; i < K; i++) blinds[i].z = draw(3);
    for (size_t i = 0; i < K; i++) if (provers[i]->ultra) blinds[i].pl = draw(3);
    const std::vector<Fr> b_quot = draw(W - 1);
    StandardTranscript<C> tr;
    for (auto* p : provers) p->append_vk_and_pub_input(tr);
    BatchProof<C> proof;
    proof.plookup_proofs_vec.resize(K);
    for (size_t i = 0; i < K; i++) {
        proof.wires_poly_comms_vec.push_back(provers[i]->round1(*circuits[i], std::move(blinds[i]), tick));
        for (auto& cm : proof.wires_poly_comms_vec.back()) tr.append_commitment("witness_poly_comms", cm);
    }
    const Fr tau = tr.get_and_append_challenge("tau");
    for (size_t i = 0; i < K; i++) {
        auto h = provers[i]->round1_5(tau, tick);
        for (auto& cm : h) tr.append_commitment("h_poly_comms", cm);
        if (provers[i]->ultra) { proof.plookup_proofs_vec[i].some = true; proof.plookup_proofs_vec[i].h_poly_comms = h; }
    }
    const Fr beta = tr.get_and_append_challenge("beta"), gamma = tr.get_and_append_challenge("gamma");
    for (size_t i = 0; i < K; i++) {
        proof.prod_perm_poly_comms_vec.push_back(provers[i]->round2(beta, gamma, tick));
        tr.append_commitment("perm_poly_comms", proof.prod_perm_poly_comms_vec.back());
    }
    for (size_t i = 0; i < K; i++)
        if (provers[i]->ultra) {
            proof.plookup_proofs_vec[i].prod_lookup_poly_comm = provers[i]->round2_5(tick);
            tr.append_commitment("plookup_poly_comms", proof.plookup_proofs_vec[i].prod_lookup_poly_comm);
        }
    const Fr alpha = tr.get_and_append_challenge("alpha");
    const Fr a3 = alpha * alpha * alpha, a7 = a3 * a3 * alpha;
    std::vector<Fr> bases;
    std::vector<Term> qterms;
    Fr base = Fr::one();
    for (size_t i = 0; i < K; i++) {
        provers[i]->quotient(alpha, tick);
        qterms.push_back({base, provers[i]->quot.p, p0.m});
        bases.push_back(base);
        base = base * (provers[i]->ultra ? a7 : a3);
    }
    DevBuf qsum;
    const void* q = p0.quot.p;
    if (K > 1) { qsum.alloc(p0.m); p0.lincomb_many(qterms, qsum.p, p0.m); q = qsum.p; }
    const std::vector<uint64_t> split_len = p0.split_quotient(q, b_quot);
    proof.split_quot_poly_comms = p0.commit_split(split_len);
    p0.check_quotient_degree();
    for (auto& cm : proof.split_quot_poly_comms) tr.append_commitment("quot_poly_comms", cm);
    const Fr zeta = tr.get_and_append_challenge("zeta");
    for (size_t i = 0; i < K; i++) {
        provers[i]->round4(zeta, tick);
        provers[i]->append_proof_evaluations(tr);
        proof.poly_evals_vec.push_back({provers[i]->st.wires_evals, provers[i]->st.wire_sigma_evals, provers[i]->st.perm_next_eval});
    }
    for (size_t i = 0; i < K; i++) {
        provers[i]->append_plookup_evaluations(tr);
        if (provers[i]->ultra) proof.plookup_proofs_vec[i].poly_evals = provers[i]->st.plookup_evals;
    }
    std::vector<Term> lin_terms = p0.quotient_lin_terms(zeta, split_len);
    for (size_t i = 0; i < K; i++) {
        const std::vector<Term> t = provers[i]->lin_poly_terms(bases[i]);
        lin_terms.insert(lin_terms.end(), t.begin(), t.end());
    }
    p0.lincomb_many(lin_terms, p0.lin.p, n + 3);
    const Fr v = tr.get_and_append_challenge("v");
    std::vector<Term> open_polys{{Fr::one(), p0.lin.p, n + 3}}, shifted_polys;
    for (auto* p : provers) p->open_lists(open_polys, shifted_polys);
    p0.batched_witness(open_polys, v, zeta, p0.opening, p0.rem_dev());
    p0.batched_witness(shifted_polys, v, zeta * p0.w_n, p0.shifted);
    const auto oc = p0.commit({p0.opening.p, p0.shifted.p}, {n + 2, n + 2});
    {   // the quotient identity at zeta over all instances (Prover::check_quotient_identity): the guard against an unsatisfied witness
        Fr lin_constant = Fr::zero();
        std::vector<Fr> opened;
        for (size_t i = 0; i < K; i++) { lin_constant = lin_constant + provers[i]->lin_poly_constant(bases[i]); provers[i]->opened_evals(opened); }
        P::check_quotient_identity(p0.download_fr(p0.rem_dev()), lin_constant, opened, v);
    }
    proof.opening_proof = oc[0];
    proof.shifted_opening_proof = oc[1];
    return proof;
}

// ---- proof linking (plonk/src/proof_system/proof_linking.rs) -----------------------------------------------------------
struct GroupLayout {                                                   // relation/src/proof_linking/mod.rs:16-54
    uint32_t alignment;
    uint64_t offset, size;
};
template <class C>
struct LinkingHint {                                                   // structs.rs:88-97: the masked a(X) on the device + its commitment
    DevBuf linking_wire_poly;
    uint64_t len = 0;
    typename Encoding<C>::Affine linking_wire_comm;
};
template <class C>
struct LinkingProof {                                                  // proof_linking.rs:33-39
    typename Encoding<C>::Affine quotient_commitment, opening_proof;
    std::vector<uint8_t> serialize_compressed() const {
        std::vector<uint8_t> out(2 * C::G1_BYTES);
        Encoding<C>::g1_bytes(quotient_commitment, out.data());
        Encoding<C>::g1_bytes(opening_proof, out.data() + C::G1_BYTES);
        return out;
    }
};

// the hint of PlonkKzgSnark::prove_with_link_hint (snark.rs:81-119), taken from the prover right after `prove`
template <class C>
LinkingHint<C> link_hint(const Prover<C>& prover, const Proof<C>& proof) {
    LinkingHint<C> h;
    h.len = prover.n + 2;
    h.linking_wire_poly.alloc(h.len);
    check(mzk_dev_copy(h.linking_wire_poly.p, prover.krow(0), h.len * EL, nullptr), "copy");          // row 0 = wire polynomial 0
    h.linking_wire_comm = proof.wires_poly_comms[0];
    return h;
}

// PlonkKzgSnark::link_proofs (proof_linking.rs:80-111)
template <class C>
LinkingProof<C> link_proofs(uint64_t srs, const LinkingHint<C>& lhs, const LinkingHint<C>& rhs, const GroupLayout& layout) {
    using E = Encoding<C>;
    using Fr = typename E::Fr;
    using FrP = typename C::Fr;
    using Affine = typename E::Affine;
    constexpr int QL = E::QL;
    if ((int)layout.alignment > FrP::TWO_ADICITY) throw std::runtime_error("field 2-adicity too small for layout");
    const uint64_t len = std::max(lhs.len, rhs.len);
    auto lincomb = [&](const std::vector<std::tuple<Fr, const void*, uint64_t>>& terms, void* out, uint64_t out_len) {
        std::vector<const void*> ptrs;
        std::vector<uint64_t> lens, sc;
        for (auto& t : terms) { ptrs.push_back(std::get<1>(t)); lens.push_back(std::get<2>(t)); for (int i = 0; i < 4; i++) sc.push_back(std::get<0>(t).l[i]); }
        check(mzk_poly_lincomb_dev(C::ID, (uint32_t)terms.size(), ptrs.data(), lens.data(), sc.data(), out, out_len, nullptr), "mzk_poly_lincomb_dev");
    };
    auto commit = [&](const void* poly, uint64_t n_coeffs) {
        Affine out;
        std::memset(out.data(), 0, sizeof(Affine));
        if (n_coeffs == 0) return out;
        uint64_t xyz[3 * QL];
        const void* ptrs[1] = {poly};
        const uint64_t lens[1] = {n_coeffs};
        check(mzk_msm_batch_dev(srs, 1, ptrs, lens, nullptr, 1, xyz, nullptr), "mzk_msm_batch_dev");
        check(mzk_g1_jacobian_to_affine(C::ID, xyz, 1, reinterpret_cast<uint64_t*>(out.data())), "mzk_g1_jacobian_to_affine");
        return out;
    };
    const Fr one = Fr::one(), minus_one = Fr::zero() - Fr::one();
    // quotient (a_1 - a_2) / Z_D (proof_linking.rs:119-134)
    DevBuf diff(len), quotient(len > layout.size ? len - layout.size : 0), identity(len), witness(len - 1);
    lincomb({{one, lhs.linking_wire_poly.p, lhs.len}, {minus_one, rhs.linking_wire_poly.p, rhs.len}}, diff.p, len);
    const uint64_t q_len = quotient.elems;
    if (q_len) check(mzk_poly_div_roots_dev(C::ID, diff.p, len, layout.alignment, layout.offset, layout.size, quotient.p, nullptr), "mzk_poly_div_roots_dev");
    LinkingProof<C> proof;
    proof.quotient_commitment = commit(quotient.p, q_len);
    // eta (proof_linking.rs:185-197)
    StandardTranscript<C> tr("PlonkLinkingProof");
    tr.append_commitment("linking_wire_comms", lhs.linking_wire_comm);
    tr.append_commitment("linking_wire_comms", rhs.linking_wire_comm);
    tr.append_commitment("quotient_comm", proof.quotient_commitment);
    const Fr eta = tr.get_and_append_challenge("eta");
    // Z_D(eta) (proof_linking.rs:162-176)
    const Fr g = root_of_unity<FrP>((int)layout.alignment);
    Fr root = pow_u64(g, layout.offset), z_eta = Fr::one();
    for (uint64_t i = 0; i < layout.size; i++) { z_eta = z_eta * (eta - root); root = root * g; }
    // identity polynomial a_1 - a_2 - q Z_D(eta), opened at eta (proof_linking.rs:204-221, univariate_kzg/mod.rs:135-161)
    if (q_len) lincomb({{one, diff.p, len}, {Fr::zero() - z_eta, quotient.p, q_len}}, identity.p, len);
    else lincomb({{one, diff.p, len}}, identity.p, len);
    check(mzk_poly_div_linear_dev(C::ID, identity.p, len, eta.l, witness.p, nullptr), "mzk_poly_div_linear_dev");
    proof.opening_proof = commit(witness.p, len - 1);
    return proof;
}

}  // namespace mzk_host
