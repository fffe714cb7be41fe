// mzk_prover.hpp -- PlonkKzgSnark::preprocess / ::prove / ::batch_prove / ::link_proofs in C++ above the C ABI (see mzk_host.hpp for the
// map to the reference): circuits (the reference's bench circuit, or any finalised circuit read from a file), transcript, rng draws and
// Proof assembly here; the rounds themselves run inside the library (mzk_prover_*, csrc/prover.hip).  TurboPlonk or UltraPlonk.
#include <fstream>
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
    // public input (`Arithmetization::public_input`): value i sits on row pub_rows[i] of the public-input vector
    // (relation/src/constraint_system.rs:1249-1259); pub_rows empty = rows 0 .. len - 1, where finalisation puts the IO gates
    std::vector<Fr> pub_input;
    std::vector<uint64_t> pub_rows;

    // Does round 1 gain from the Lagrange-basis key (snark.py witness_is_small, the same rule)?  Small = below 2^64 for at least half of 2048
    // strided wire values; a dense witness gains nothing from the key and its table (1.75 GB and 0.5 s at 2^20 gates on BLS12-381).
    bool witness_is_small() const {
        const size_t total = wires.size();
        if (!total) return false;
        const size_t step = std::max<size_t>(1, total / 2048);
        size_t seen = 0, small = 0;
        for (size_t i = 0; i < total && seen < 2048; i += step, seen++) {
            const auto c = canonical(wires[i]);
            small += (c[1] | c[2] | c[3]) == 0;
        }
        return 2 * small >= seen;
    }

    // A finalised circuit from a file (mpc-jellyfish_amd/circuit_io.py writes it), everything little-endian, field elements as 4 x u64
    // Montgomery limbs, vectors as VALUES on the gate domain H (what `Arithmetization` exposes before the iFFTs of preprocess / round 1):
    //   "MZKCIRC1" | u32 curve_id | u32 num_wire_types (5 | 6) | u32 log_n | u32 n_pub
    //   k[W] | selectors[nsel][n] | sigmas[W][n] | (W == 6: range, key, table_dom_sep, q_dom_sep tables [4][n]) | wires[W][n]
    //   pub_rows[n_pub] (u64) | pub_values[n_pub]
    static BenchCircuitHost read(const std::string& path) {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("cannot open circuit file " + path);
        char magic[8];
        uint32_t hdr[4];
        f.read(magic, 8);
        f.read(reinterpret_cast<char*>(hdr), sizeof hdr);
        if (!f || std::memcmp(magic, "MZKCIRC1", 8) != 0) throw std::runtime_error("not a circuit file (magic MZKCIRC1)");
        if ((int)hdr[0] != C::ID) throw std::runtime_error("circuit file is over the other curve");
        if ((hdr[1] != 5 && hdr[1] != 6) || hdr[2] < 1 || hdr[2] > 27) throw std::runtime_error("circuit file: 5 or 6 wire types, 1 <= log_n <= 27");
        BenchCircuitHost cs;
        cs.W = (int)hdr[1]; cs.ultra = cs.W == 6; cs.nsel = cs.ultra ? 14 : 13; cs.log_n = (int)hdr[2]; cs.n = 1ull << cs.log_n;
        const uint64_t n = cs.n, n_pub = hdr[3];
        if (n_pub > n) throw std::runtime_error("circuit file: more public inputs than rows");
        auto vec = [&](std::vector<Fr>& v, size_t cnt) { v.resize(cnt); f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(cnt * EL)); };
        vec(cs.k, cs.W); vec(cs.selectors, (size_t)cs.nsel * n); vec(cs.sigmas, (size_t)cs.W * n);
        if (cs.ultra) vec(cs.tables, (size_t)4 * n);
        vec(cs.wires, (size_t)cs.W * n);
        cs.pub_rows.resize(n_pub);
        f.read(reinterpret_cast<char*>(cs.pub_rows.data()), (std::streamsize)(n_pub * 8));
        vec(cs.pub_input, n_pub);
        if (!f) throw std::runtime_error("circuit file truncated");
        for (uint64_t r : cs.pub_rows) if (r >= n) throw std::runtime_error("circuit file: public-input row outside the domain");
        return cs;
    }

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
    std::vector<uint32_t> host_wire_variables;                         // W x n: handed to the prover once (mzk_prover_set_wire_variables)
    uint64_t n_vars = 0;
    int host_witness = 0;
    std::vector<Fr> pub_input;
    std::vector<uint64_t> pub_rows;

    static BenchCircuit upload(const BenchCircuitHost<C>& h, int host_witness = 0) {
        BenchCircuit cs;
        cs.ultra = h.ultra; cs.log_n = h.log_n; cs.W = h.W; cs.nsel = h.nsel; cs.n = h.n; cs.k = h.k;
        cs.pub_input = h.pub_input; cs.pub_rows = h.pub_rows;
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
            if (h.witness.empty()) throw std::runtime_error("this circuit carries no witness vector / wire variables (--host-witness-vars)");
            cs.n_vars = h.witness.size();
            cs.host_vars.alloc(h.witness.size() * EL);
            std::memcpy(cs.host_vars.p, h.witness.data(), h.witness.size() * EL);
            cs.host_wire_variables = h.wire_variables;
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

// ---- PlonkKzgSnark::{preprocess, prove} as a CLIENT of the library's round-level ABI (mzk_prover_*, include/mzk.h) ----------------------
// The rounds of prover.rs:72-419 run inside libmi355zk (csrc/prover.hip); this struct keeps what snark.rs keeps: the transcript, the
// rng draws, the Proof.  It is the C++ twin of what a Rust caller writes.
template <class C>
struct Prover {
    using E = Encoding<C>;
    using Fr = typename E::Fr;
    using Affine = typename E::Affine;
    using FrP = typename C::Fr;
    static constexpr int QL = E::QL;

    bool ultra;
    int log_n, W, nsel;
    uint64_t n;
    std::vector<Fr> k;
    uint64_t srs = 0, handle = 0;
    std::vector<Affine> selector_comms, sigma_comms;
    std::map<std::string, double> timings_ms;
    struct Blinds { std::vector<std::vector<Fr>> wires, h; std::vector<Fr> z, pl; };
    // per-instance output of the rounds (what Proof / BatchProof carry)
    std::vector<Affine> wires_comms, h_comms;
    Affine z_comm, pl_comm;
    std::vector<Fr> wires_evals, wire_sigma_evals, plookup_evals;
    Fr perm_next_eval;

    // PlonkKzgSnark::preprocess (snark.rs:529-617): selector / sigma / table polynomials by iNTT on the device, handed to the library as
    // `ProvingKey{selectors, sigmas, plookup_pk}` holds them (coefficient vectors); verifying-key commitments from the resident forms.
    // lagrange_key: a handle from mzk_srs_lagrange_from_srs (0: round 1 commits the masked coefficient forms, as the reference does).
    Prover(uint64_t srs_handle, const BenchCircuit<C>& cs, uint64_t lagrange_key = 0, const mzk_comm* comm = nullptr)
        : ultra(cs.ultra), log_n(cs.log_n), W(cs.W), nsel(cs.nsel), n(cs.n), k(cs.k), srs(srs_handle) {
        const int nfix = nsel + W + (ultra ? 4 : 0);
        DevBuf fixed((size_t)nfix * n);
        check(mzk_dev_copy(fixed.p, cs.selector_values.p, (size_t)nsel * n * EL, nullptr), "copy");
        check(mzk_dev_copy(fixed.at((size_t)nsel * n), cs.sigma_values.p, (size_t)W * n * EL, nullptr), "copy");
        if (ultra) check(mzk_dev_copy(fixed.at((size_t)(nsel + W) * n), cs.table_values.p, (size_t)4 * n * EL, nullptr), "copy");
        check(mzk_ntt_dev(C::ID, fixed.p, n, log_n, 1, nullptr, nfix, n, nullptr), "mzk_ntt_dev");
        std::vector<uint64_t> host((size_t)nfix * n * 4);
        check(mzk_dev_download(host.data(), fixed.p, host.size() * 8), "download");
        std::vector<uint64_t> kk((size_t)W * 4);
        for (int i = 0; i < W; i++) std::memcpy(&kk[4 * i], k[i].l, 32);
        const uint64_t* sel = host.data();
        const uint64_t* sig = sel + (size_t)nsel * n * 4;
        check(mzk_prover_create(C::ID, log_n, W, sel, sig, ultra ? sig + (size_t)W * n * 4 : nullptr, n, kk.data(), srs, lagrange_key, comm, &handle), "mzk_prover_create");
        std::vector<uint64_t> xy((size_t)(nsel + W) * 2 * QL);
        check(mzk_prover_vk_commitments(handle, xy.data(), nullptr), "mzk_prover_vk_commitments");
        auto pt = [&](size_t i) { Affine a; std::memcpy(a.data(), &xy[i * 2 * QL], sizeof(Affine)); return a; };
        for (int i = 0; i < nsel; i++) selector_comms.push_back(pt(i));
        for (int i = 0; i < W; i++) sigma_comms.push_back(pt(nsel + i));
        if (cs.host_witness == 2) check(mzk_prover_set_wire_variables(handle, cs.host_wire_variables.data(), cs.n_vars), "mzk_prover_set_wire_variables");
    }
    Prover(const Prover&) = delete;
    Prover& operator=(const Prover&) = delete;
    ~Prover() { if (handle) (void)mzk_prover_destroy(handle); }

    static std::vector<Affine> points(const std::vector<uint64_t>& xy) {
        std::vector<Affine> out(xy.size() / (2 * QL));
        for (size_t i = 0; i < out.size(); i++) std::memcpy(out[i].data(), &xy[i * 2 * QL], sizeof(Affine));
        return out;
    }
    static std::vector<uint64_t> flat(const std::vector<Fr>& v) {
        std::vector<uint64_t> out(v.size() * 4);
        for (size_t i = 0; i < v.size(); i++) std::memcpy(&out[4 * i], v[i].l, 32);
        return out;
    }
    static std::vector<uint64_t> flat2(const std::vector<std::vector<Fr>>& v) {
        std::vector<uint64_t> out;
        for (auto& row : v) { auto f = flat(row); out.insert(out.end(), f.begin(), f.end()); }
        return out;
    }
    void append_vk_and_pub_input(StandardTranscript<C>& tr, const BenchCircuit<C>& cs) const {       // transcript/mod.rs:45-104
        tr.append_u32("field size in bits", (uint32_t)FrP::BITS);
        tr.append_u64("domain size", n);
        tr.append_u64("input size", cs.pub_input.size());
        for (auto& ki : k) tr.append_fr("wire subsets separators", ki);
        for (auto& cm : selector_comms) tr.append_commitment("selector commitments", cm);
        for (auto& cm : sigma_comms) tr.append_commitment("sigma commitments", cm);
        for (auto& x : cs.pub_input) tr.append_fr("public input", x);
    }
    // round 1 (prover.rs:72-87): the witness as the circuit holds it -- on the device, in page-locked host memory (W x n), or as the
    // witness vector gathered on the device
    void round1(const BenchCircuit<C>& cs, const Blinds& b) {
        const void* wit = cs.host_witness == 0 ? cs.wire_values.p : (cs.host_witness == 1 ? cs.host_wires.p : cs.host_vars.p);
        const int32_t kind = cs.host_witness == 0 ? MZK_WITNESS_DEV_WIRES : (cs.host_witness == 1 ? MZK_WITNESS_HOST_WIRES : MZK_WITNESS_HOST_VECTOR);
        const uint64_t len = cs.host_witness == 2 ? cs.n_vars : (uint64_t)W * n;
        std::vector<uint64_t> out((size_t)W * 2 * QL), bl = flat2(b.wires), pi = flat(cs.pub_input);
        check(mzk_prover_round1(handle, kind, wit, len, cs.pub_rows.empty() ? nullptr : cs.pub_rows.data(), pi.data(), cs.pub_input.size(), bl.data(), out.data()),
              "mzk_prover_round1");
        wires_comms = points(out);
    }
    void round1_5(const Fr& tau, const Blinds& b) {                          // prover.rs:89-118
        h_comms.clear();
        if (!ultra) return;
        std::vector<uint64_t> out((size_t)2 * 2 * QL), bl = flat2(b.h);
        check(mzk_prover_round1_5(handle, tau.l, bl.data(), out.data()), "mzk_prover_round1_5");
        h_comms = points(out);
    }
    void round2(const Fr& beta, const Fr& gamma, const Blinds& b) {          // prover.rs:125-141
        std::vector<uint64_t> out((size_t)2 * QL), bl = flat(b.z);
        check(mzk_prover_round2(handle, beta.l, gamma.l, bl.data(), out.data()), "mzk_prover_round2");
        z_comm = points(out)[0];
    }
    void round2_5(const Blinds& b) {                                          // prover.rs:143-183
        if (!ultra) return;
        std::vector<uint64_t> out((size_t)2 * QL), bl = flat(b.pl);
        check(mzk_prover_round2_5(handle, bl.data(), out.data()), "mzk_prover_round2_5");
        pl_comm = points(out)[0];
    }
    static std::vector<Affine> round3(const std::vector<Prover*>& inst, const Fr& alpha, const std::vector<Fr>& b_quot) {      // prover.rs:192-209
        std::vector<uint64_t> hs, out((size_t)inst[0]->W * 2 * QL), bl = flat(b_quot);
        for (auto* p : inst) hs.push_back(p->handle);
        check(mzk_prover_round3(hs.data(), (uint32_t)hs.size(), alpha.l, bl.data(), out.data()), "mzk_prover_round3");
        return points(out);
    }
    void round4(const Fr& zeta) {                                             // prover.rs:216-299
        std::vector<Fr> ev((size_t)2 * W + (ultra ? N_PLOOKUP_EVALS : 0));
        check(mzk_prover_round4(handle, zeta.l, reinterpret_cast<uint64_t*>(ev.data())), "mzk_prover_round4");
        wires_evals.assign(ev.begin(), ev.begin() + W);
        wire_sigma_evals.assign(ev.begin() + W, ev.begin() + 2 * W - 1);
        perm_next_eval = ev[2 * W - 1];
        plookup_evals.assign(ev.begin() + 2 * W, ev.end());
    }
    static std::vector<Affine> round5(const std::vector<Prover*>& inst, const Fr& v) {      // prover.rs:302-460
        std::vector<uint64_t> hs, out((size_t)2 * 2 * QL);
        for (auto* p : inst) hs.push_back(p->handle);
        check(mzk_prover_round5(hs.data(), (uint32_t)hs.size(), v.l, out.data()), "mzk_prover_round5");
        return points(out);
    }
    void append_proof_evaluations(StandardTranscript<C>& tr) const {       // transcript/mod.rs:140-163
        for (auto& v : wires_evals) tr.append_fr("wire_evals", v);
        for (auto& v : wire_sigma_evals) tr.append_fr("wire_sigma_evals", v);
        tr.append_fr("perm_next_eval", perm_next_eval);
    }
    void append_plookup_evaluations(StandardTranscript<C>& tr) const {     // transcript/mod.rs:165-202
        if (!ultra) return;
        const std::vector<Fr>& pe = plookup_evals;
        tr.append_fr("lookup_table_eval", pe[RANGE_TABLE]); tr.append_fr("h_1_eval", pe[H_1]); tr.append_fr("prod_next_eval", pe[PROD_NEXT]);
        tr.append_fr("lookup_table_next_eval", pe[RANGE_TABLE_NEXT]); tr.append_fr("h_1_next_eval", pe[H_1_NEXT]);
        tr.append_fr("h_2_next_eval", pe[H_2_NEXT]);
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
    void fetch_timings() {                                                 // {"r1_ntt_mask": 1.234, ...} as written by mzk_prover_timings
        char buf[2048];
        check(mzk_prover_timings(handle, buf, sizeof buf), "mzk_prover_timings");
        timings_ms.clear();
        for (const char* p = buf; (p = std::strchr(p, '"')) != nullptr;) {
            const char* q = std::strchr(p + 1, '"');
            if (!q) break;
            timings_ms[std::string(p + 1, q)] = std::atof(q + 2);
            p = q + 1;
        }
    }

    // PlonkKzgSnark::prove (snark.rs:624-651) -> batch_prove_internal (:201-469), one instance
    Proof<C> prove(ChaChaRng& rng, const BenchCircuit<C>& cs, bool profile = false) {
        Blinds blinds = draw_blinds(rng, W, ultra);
        std::vector<Fr> b_quot;
        for (int i = 0; i < W - 1; i++) b_quot.push_back(fr_rand<FrP>(rng));                                   // prover.rs:947-955
        return prove_with(blinds, b_quot, cs, profile);
    }
    // ... with the masking draws made by the caller: over several devices every rank runs this with the SAME draws (ShardedProver)
    Proof<C> prove_with(const Blinds& blinds, const std::vector<Fr>& b_quot, const BenchCircuit<C>& cs, bool profile = false) {
        check(mzk_prover_profile(handle, profile ? 1 : 0), "mzk_prover_profile");
        StandardTranscript<C> tr;
        append_vk_and_pub_input(tr, cs);
        Proof<C> proof;
        proof.has_plookup = ultra;
        round1(cs, blinds);
        proof.wires_poly_comms = wires_comms;
        for (auto& cm : proof.wires_poly_comms) tr.append_commitment("witness_poly_comms", cm);
        const Fr tau = tr.get_and_append_challenge("tau");                    // squeezed even without Plookup (snark.rs:293)
        round1_5(tau, blinds);
        proof.h_poly_comms = h_comms;
        for (auto& cm : proof.h_poly_comms) tr.append_commitment("h_poly_comms", cm);
        const Fr beta = tr.get_and_append_challenge("beta"), gamma = tr.get_and_append_challenge("gamma");
        round2(beta, gamma, blinds);
        proof.prod_perm_poly_comm = z_comm;
        tr.append_commitment("perm_poly_comms", proof.prod_perm_poly_comm);
        if (ultra) {
            round2_5(blinds);
            proof.prod_lookup_poly_comm = pl_comm;
            tr.append_commitment("plookup_poly_comms", proof.prod_lookup_poly_comm);
        }
        const Fr alpha = tr.get_and_append_challenge("alpha");
        proof.split_quot_poly_comms = round3({this}, alpha, b_quot);
        for (auto& cm : proof.split_quot_poly_comms) tr.append_commitment("quot_poly_comms", cm);
        const Fr zeta = tr.get_and_append_challenge("zeta");
        round4(zeta);
        append_proof_evaluations(tr);
        append_plookup_evaluations(tr);
        proof.wires_evals = wires_evals;
        proof.wire_sigma_evals = wire_sigma_evals;
        proof.perm_next_eval = perm_next_eval;
        proof.plookup_evals = plookup_evals;
        const Fr v = tr.get_and_append_challenge("v");
        const auto oc = round5({this}, v);
        proof.opening_proof = oc[0];
        proof.shifted_opening_proof = oc[1];
        if (profile) fetch_timings();
        return proof;
    }
};

// ---- G devices from one process: G host threads, each bound to its device's context of libmi355zk, all running Prover::prove_with
// ---- on the same draws (SPMD); they end with identical proofs.  Every device-side object of rank g is created, used and destroyed
// ---- on thread g.  world == 1 runs on the calling thread.  The library's rounds reach the other ranks through the mzk_comm callbacks
// ---- below: all-gathers of a few hundred bytes and barriers through LocalComm; the one bulk exchange (class remainders of round 3)
// ---- is pushed device to device by the library itself into the peers' buffers (mzk_prover_set_peer_buffers).
template <class C>
struct ShardedProver {
    using P = Prover<C>;
    using Fr = typename P::Fr;
    const int G;
    LocalComm comm;
    struct RankCtx { LocalComm* comm; int rank; };
    std::vector<RankCtx> rank_ctx;
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

    static int32_t cb_all_gather(void* ctx, const void* send, uint64_t bytes, void* recv) {
        auto* c = static_cast<RankCtx*>(ctx);
        try {
            const std::vector<uint8_t> all = c->comm->all_gather(c->rank, send, bytes);
            std::memcpy(recv, all.data(), all.size());
            return 0;
        } catch (...) { return 1; }
    }
    static int32_t cb_barrier(void* ctx) {
        auto* c = static_cast<RankCtx*>(ctx);
        try { c->comm->barrier(); return 0; } catch (...) { return 1; }
    }

    explicit ShardedProver(int g) : G(g), comm(g), rank_ctx(g), circuit(g), prover(g), srs(g, 0), srs_lagrange(g, 0), errors(g) {
        for (int r = 0; r < G; r++) rank_ctx[r] = {&comm, r};
        if (G > 1)
            for (int r = 0; r < G; r++) threads.emplace_back([this, r] { worker(r); });
    }
    ~ShardedProver() {
        try {
            each([&](int r) {
                prover[r].reset();
                circuit[r].reset();
                if (srs[r]) (void)mzk_srs_release(srs[r]);
                srs[r] = 0;
                if (srs_lagrange[r]) (void)mzk_srs_release(srs_lagrange[r]);
                srs_lagrange[r] = 0;
            });
        } catch (...) {}
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
        for (auto& e : errors) {                                          // prefer the root cause over a rank that merely gave up in a collective
            if (!e) continue;
            try { std::rethrow_exception(e); }
            catch (const std::runtime_error& x) { if (std::string(x.what()).find("mzk_comm.") == std::string::npos) std::rethrow_exception(e); other = e; }
            catch (...) { std::rethrow_exception(e); }
        }
        if (other) std::rethrow_exception(other);
    }
    // the testing SRS [beta^i] G on every device, the circuit uploaded to every device, PlonkKzgSnark::preprocess per device
    // lagrange: also the key over the Lagrange basis of the gate domain -- round 1 then commits from the wire values
    // slice_srs (several devices): every rank keeps ONLY its point range of the commit key(s) -- mzk_srs_slice -- i.e. 1 / G of the SRS and of
    // its fixed-base table, built with the window that suits the slice (2^17-point shards at G = 8: window 16 and fused small batches)
    void setup(const BenchCircuitHost<C>& host, const std::array<uint64_t, 4>& beta_canonical, int host_witness = 0, bool lagrange = true, bool slice_srs = true) {
        each([&](int r) {
            check(mzk_srs_generate_for_testing(C::ID, beta_canonical.data(), host.n + 3, &srs[r]), "mzk_srs_generate_for_testing");
            if (lagrange) {                                               // from the SRS's points alone (no trapdoor): an inverse NTT over the group
                const auto t0 = std::chrono::steady_clock::now();
                check(mzk_srs_lagrange_from_srs(srs[r], (uint32_t)host.log_n, 3, &srs_lagrange[r]), "mzk_srs_lagrange_from_srs");
                if (r == 0) lagrange_key_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            }
            if (G > 1 && slice_srs) {
                const auto range = shard_range(host.n + 3, r, G);
                for (uint64_t* key : {&srs[r], &srs_lagrange[r]}) {
                    if (!*key) continue;
                    uint64_t part = 0;
                    check(mzk_srs_slice(*key, range.first, range.second - range.first, &part), "mzk_srs_slice");
                    check(mzk_srs_release(*key), "mzk_srs_release");
                    *key = part;
                }
            }
            circuit[r] = std::make_unique<BenchCircuit<C>>(BenchCircuit<C>::upload(host, host_witness));
            mzk_comm cm{&rank_ctx[r], r, G, &cb_all_gather, &cb_barrier, nullptr};
            prover[r] = std::make_unique<P>(srs[r], *circuit[r], srs_lagrange[r], G > 1 ? &cm : nullptr);
        });
        if (G > 1) {                                                      // where every rank receives the class remainders of the others
            std::vector<void*> rem(G);
            std::vector<int32_t> dev(G);
            for (int q = 0; q < G; q++) { check(mzk_prover_exchange_buffer(prover[q]->handle, &rem[q], nullptr), "mzk_prover_exchange_buffer"); dev[q] = q; }
            for (int r = 0; r < G; r++) check(mzk_prover_set_peer_buffers(prover[r]->handle, rem.data(), dev.data()), "mzk_prover_set_peer_buffers");
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

// PlonkKzgSnark::batch_prove: round k of every instance, then one challenge; rounds 3 and 5 ONCE over all handles -- one quotient
// t = sum_k alpha_base_k t_k (alpha_base_{k+1} = alpha_base_k alpha^3, alpha^7 with Plookup: prover.rs:661-669), one split (first key's
// commit key), one linearisation polynomial and the two opening proofs over the concatenated lists (prover.rs:362-419): all inside
// mzk_prover_round3 / mzk_prover_round5.
template <class C>
BatchProof<C> batch_prove(ChaChaRng& rng, const std::vector<Prover<C>*>& provers, const std::vector<const BenchCircuit<C>*>& circuits) {
    using P = Prover<C>;
    using Fr = typename P::Fr;
    using FrP = typename C::Fr;
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
    // prng draws in the reference's order (snark.rs:277-360)
    auto draw = [&](int cnt) { std::vector<Fr> v; for (int i = 0; i < cnt; i++) v.push_back(fr_rand<FrP>(rng)); return v; };
    std::vector<typename P::Blinds> blinds(K);
    for (size_t i = 0; i < K; i++) for (int j = 0; j < W; j++) blinds[i].wires.push_back(draw(2));
    for (size_t i = 0; i < K; i++) if (provers[i]->ultra) { blinds[i].h.push_back(draw(3)); blinds[i].h.push_back(draw(3)); }
    for (size_t i = 0; i < K; i++) blinds[i].z = draw(3);
    for (size_t i = 0; i < K; i++) if (provers[i]->ultra) blinds[i].pl = draw(3);
    const std::vector<Fr> b_quot = draw(W - 1);
    StandardTranscript<C> tr;
    for (size_t i = 0; i < K; i++) provers[i]->append_vk_and_pub_input(tr, *circuits[i]);
    BatchProof<C> proof;
    proof.plookup_proofs_vec.resize(K);
    for (size_t i = 0; i < K; i++) {
        provers[i]->round1(*circuits[i], blinds[i]);
        proof.wires_poly_comms_vec.push_back(provers[i]->wires_comms);
        for (auto& cm : proof.wires_poly_comms_vec.back()) tr.append_commitment("witness_poly_comms", cm);
    }
    const Fr tau = tr.get_and_append_challenge("tau");
    for (size_t i = 0; i < K; i++) {
        provers[i]->round1_5(tau, blinds[i]);
        for (auto& cm : provers[i]->h_comms) tr.append_commitment("h_poly_comms", cm);
        if (provers[i]->ultra) { proof.plookup_proofs_vec[i].some = true; proof.plookup_proofs_vec[i].h_poly_comms = provers[i]->h_comms; }
    }
    const Fr beta = tr.get_and_append_challenge("beta"), gamma = tr.get_and_append_challenge("gamma");
    for (size_t i = 0; i < K; i++) {
        provers[i]->round2(beta, gamma, blinds[i]);
        proof.prod_perm_poly_comms_vec.push_back(provers[i]->z_comm);
        tr.append_commitment("perm_poly_comms", proof.prod_perm_poly_comms_vec.back());
    }
    for (size_t i = 0; i < K; i++)
        if (provers[i]->ultra) {
            provers[i]->round2_5(blinds[i]);
            proof.plookup_proofs_vec[i].prod_lookup_poly_comm = provers[i]->pl_comm;
            tr.append_commitment("plookup_poly_comms", proof.plookup_proofs_vec[i].prod_lookup_poly_comm);
        }
    const Fr alpha = tr.get_and_append_challenge("alpha");
    proof.split_quot_poly_comms = P::round3(provers, alpha, b_quot);
    for (auto& cm : proof.split_quot_poly_comms) tr.append_commitment("quot_poly_comms", cm);
    const Fr zeta = tr.get_and_append_challenge("zeta");
    for (size_t i = 0; i < K; i++) {
        provers[i]->round4(zeta);
        provers[i]->append_proof_evaluations(tr);
        proof.poly_evals_vec.push_back({provers[i]->wires_evals, provers[i]->wire_sigma_evals, provers[i]->perm_next_eval});
    }
    for (size_t i = 0; i < K; i++) {
        provers[i]->append_plookup_evaluations(tr);
        if (provers[i]->ultra) proof.plookup_proofs_vec[i].poly_evals = provers[i]->plookup_evals;
    }
    const Fr v = tr.get_and_append_challenge("v");
    const auto oc = P::round5(provers, v);
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
    const void* wire0 = nullptr;
    check(mzk_prover_poly_dev(prover.handle, 0, &wire0, &h.len), "mzk_prover_poly_dev");             // the masked wire polynomial 0, n + 2 coefficients
    h.linking_wire_poly.alloc(h.len);
    check(mzk_dev_copy(h.linking_wire_poly.p, wire0, h.len * EL, nullptr), "copy");
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
