// mzk_prove -- PlonkKzgSnark::prove on the reference's bench circuit from a compiled host: C++ above the C ABI of
// include/mzk.h, no Python, no HIP in this translation unit (g++ builds it).
//   mzk_prove <curve: 0 BLS12-381 | 1 BN254> <turbo|ultra> <num_gates> [reps] [range_bit_len] [--gpus G] [--host-witness | --host-witness-vars] [--check-agree]
//   mzk_prove <curve> file <circuit file> [reps] [--gpus G] [--host-witness] ...
// `file`: ANY finalised circuit -- public inputs, every gate type, copy constraints, lookups -- as the arrays `Arithmetization` exposes
// (format: BenchCircuitHost::read in mzk_prover.hpp; mpc-jellyfish_amd/circuit_io.py writes it).
// Prints one JSON line: proof bytes (hex), wall time per proof, per-round times of one profiled proof.
// --gpus G: G devices driven from this ONE process, one host thread per device (ShardedProver, mzk_prover.hpp): commitments sharded by
// point range, the quotient by residue class with one device-to-device exchange, rounds 4-5 by coefficient range; same proof bytes.
// With MZK_VIRTUAL_DEVICES=G in the environment the G device contexts share one card (rehearsal on a one-GPU box).
//   mzk_prove <curve> link <num_gates_1> <num_gates_2> <alignment> <offset> <size> [reps]
//   mzk_prove <curve> batch <turbo|ultra> <range_bit_len> <num_gates_1> <num_gates_2> ...
// PlonkKzgSnark::batch_prove: one aggregated BatchProof over bench circuits of one domain size; prints its bytes.
// (link:) proves two TurboPlonk bench circuits of one domain size (wire 0 of the bench circuit holds 0, 1, 2, .. so any rows below both
// gate counts are shared witnesses), then PlonkKzgSnark::link_proofs on their hints; prints both proofs and the LinkingProof.
#include <chrono>
#include <cstdlib>
#include <memory>

#include <algorithm>
#include "mzk_prover.hpp"

using namespace mzk_host;

struct Options {
    int gpus = 1;                 // --gpus G: G devices from this one process, one host thread each (MZK_VIRTUAL_DEVICES=G: all on one card)
    int host_witness = 0;         // --host-witness: every proof uploads its W x n wire values from page-locked host memory;
                                  // --host-witness-vars: only the witness vector, gathered per wire on the device
    bool check_agree = false;     // --check-agree: every rank's proof bytes are compared (tests)
    bool slice_srs = true;        // --no-slice: with --gpus G every rank keeps the whole commit key (and its table) instead of its point range
    int lagrange = -1;            // round 1 commits the wires from their VALUES over the Lagrange-basis key derived from the SRS (same proof
                                  // bytes): -1 = from 2^18 gates on (below, the heavy-bucket paths of small scalars cost more than they save: 2^15 gates 3.93 against 3.73 ms) when a sample of the witness
                                  // shows small values (a dense witness gains nothing from the key), --lagrange = always,
                                  // --no-lagrange = never (from the masked coefficient forms, as the reference does)
};

template <class C>
int run(bool ultra, uint64_t num_gates, int reps, int range_bits, const Options& opt, const char* circuit_file = nullptr) {
    using Fr = Fp64<typename C::Fr>;
    if (opt.gpus == 1) check(mzk_init(-1), "mzk_init");                  // (with several devices each worker thread binds its own)
    auto t0 = std::chrono::steady_clock::now();
    BenchCircuitHost<C> host = circuit_file ? BenchCircuitHost<C>::read(circuit_file) : BenchCircuitHost<C>::generate(num_gates, ultra, range_bits);
    ultra = host.ultra;
    if (std::getenv("MZK_PROVE_CORRUPT_WITNESS"))                       // test hook: wire 0 of row 5 takes the value of row 6 -> gate 5 no longer holds
    {
        host.wires[5] = host.wires[6];
        if (!host.witness.empty()) host.witness[host.wire_variables[5]] = host.witness[host.wire_variables[6]];
    }
    ChaChaRng rng = test_rng();
    const Fr beta = fr_rand<typename C::Fr>(rng);                       // the SRS trapdoor: first draw of the bench's rng (bench.rs:50-54)
    const auto beta_c = canonical(beta);
    ShardedProver<C> sp(opt.gpus);
    double circuit_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    t0 = std::chrono::steady_clock::now();
    const bool lagrange = opt.lagrange < 0 ? (host.log_n >= 18 && host.witness_is_small()) : opt.lagrange != 0;
    sp.setup(host, beta_c, opt.host_witness, lagrange, opt.slice_srs);                           // SRS, circuit upload and PlonkKzgSnark::preprocess on every device
    const double preprocess_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    Proof<C> proof = sp.prove(rng, false, opt.check_agree);             // the proof whose bytes are printed (and warm-up)
    const std::vector<uint8_t> bytes = proof.serialize_compressed();
    double ms = 0, median_ms = 0, min_ms = 0, max_ms = 0;
    if (reps > 0) {
        for (int i = 0; i < 2; i++) sp.prove(rng);
        sp.sync();
        // prove() returns the finished proof, so every repetition can be read off the clock on its own: the mean is what the figures of
        // rounds 1-5 are; the median says what a proof takes when nothing else happens on the host (one in a few dozen takes milliseconds longer)
        std::vector<double> each;
        t0 = std::chrono::steady_clock::now();
        auto t1 = t0;
        for (int i = 0; i < reps; i++) {
            sp.prove(rng);
            const auto t2 = std::chrono::steady_clock::now();
            each.push_back(std::chrono::duration<double, std::milli>(t2 - t1).count());
            t1 = t2;
        }
        sp.sync();
        ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
        std::sort(each.begin(), each.end());
        median_ms = each[each.size() / 2]; min_ms = each.front(); max_ms = each.back();
        sp.prove(rng, true);
    }
    Prover<C>& prover = *sp.prover[0];
    std::string hex;
    static const char* d = "0123456789abcdef";
    for (uint8_t b : bytes) { hex.push_back(d[b >> 4]); hex.push_back(d[b & 15]); }
    std::printf("{\"curve\": %d, \"plonk_type\": \"%s\", \"num_gates\": %llu, \"log_n\": %d, \"gpus\": %d, \"host_witness\": %s, \"lagrange_round1\": %s, \"proof_bytes\": %zu, "
                "\"prove_ms\": %.3f, \"prove_median_ms\": %.3f, \"prove_min_ms\": %.3f, \"prove_max_ms\": %.3f, \"circuit_build_s\": %.3f, \"preprocess_s\": %.3f, \"lagrange_key_s\": %.3f, \"rounds_ms\": {",
                C::ID, ultra ? "UltraPlonk" : "TurboPlonk", (unsigned long long)num_gates, host.log_n, opt.gpus, opt.host_witness == 0 ? "false" : (opt.host_witness == 1 ? "\"wire table\"" : "\"witness vector\""), lagrange ? "true" : "false",
                bytes.size(), ms, median_ms, min_ms, max_ms, circuit_s, preprocess_s, sp.lagrange_key_s);
    bool first = true;
    for (auto& kv : prover.timings_ms) { std::printf("%s\"%s\": %.3f", first ? "" : ", ", kv.first.c_str(), kv.second); first = false; }
    std::vector<uint8_t> vk_bytes;                                      // VerifyingKey commitments (selectors, then sigmas), compressed
    for (const auto* v : {&prover.selector_comms, &prover.sigma_comms})
        for (const auto& cm : *v) { uint8_t b[48]; Encoding<C>::g1_bytes(cm, b); vk_bytes.insert(vk_bytes.end(), b, b + C::G1_BYTES); }
    std::string vk_hex;
    for (uint8_t b : vk_bytes) { vk_hex.push_back(d[b >> 4]); vk_hex.push_back(d[b & 15]); }
    std::printf("}, \"vk_hex\": \"%s\", \"proof_hex\": \"%s\"}\n", vk_hex.c_str(), hex.c_str());
    return 0;
}

static std::string to_hex(const std::vector<uint8_t>& bytes) {
    std::string hex;
    static const char* d = "0123456789abcdef";
    for (uint8_t b : bytes) { hex.push_back(d[b >> 4]); hex.push_back(d[b & 15]); }
    return hex;
}

template <class C>
int run_link(uint64_t gates1, uint64_t gates2, const GroupLayout& layout, int reps) {
    using Fr = Fp64<typename C::Fr>;
    check(mzk_init(-1), "mzk_init");
    BenchCircuit<C> cs1 = BenchCircuit<C>::generate(gates1, false, 8), cs2 = BenchCircuit<C>::generate(gates2, false, 8);
    if (cs1.n != cs2.n) throw std::runtime_error("the two bench circuits must share one domain size");
    ChaChaRng rng = test_rng();
    const Fr beta = fr_rand<typename C::Fr>(rng);
    const auto beta_c = canonical(beta);
    uint64_t srs = 0;
    check(mzk_srs_generate_for_testing(C::ID, beta_c.data(), cs1.n + 3, &srs), "mzk_srs_generate_for_testing");
    Prover<C> p1(srs, cs1), p2(srs, cs2);
    Proof<C> proof1 = p1.prove(rng, cs1);
    LinkingHint<C> h1 = link_hint(p1, proof1);
    Proof<C> proof2 = p2.prove(rng, cs2);
    LinkingHint<C> h2 = link_hint(p2, proof2);
    LinkingProof<C> link = link_proofs<C>(srs, h1, h2, layout);
    double ms = 0;
    if (reps > 0) {
        check(mzk_dev_sync(), "sync");
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; i++) link_proofs<C>(srs, h1, h2, layout);
        check(mzk_dev_sync(), "sync");
        ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
    }
    std::printf("{\"curve\": %d, \"log_n\": %d, \"link_ms\": %.3f, \"proof1_hex\": \"%s\", \"proof2_hex\": \"%s\", \"link_proof_hex\": \"%s\"}\n", C::ID, cs1.log_n, ms,
                to_hex(proof1.serialize_compressed()).c_str(), to_hex(proof2.serialize_compressed()).c_str(), to_hex(link.serialize_compressed()).c_str());
    (void)mzk_srs_release(srs);
    return 0;
}

template <class C>
int run_batch(bool ultra, int range_bits, const std::vector<uint64_t>& gates) {
    using Fr = Fp64<typename C::Fr>;
    check(mzk_init(-1), "mzk_init");
    std::vector<BenchCircuit<C>> circuits;
    for (uint64_t g : gates) circuits.push_back(BenchCircuit<C>::generate(g, ultra, range_bits));
    ChaChaRng rng = test_rng();
    const Fr beta = fr_rand<typename C::Fr>(rng);
    const auto beta_c = canonical(beta);
    uint64_t srs = 0;
    check(mzk_srs_generate_for_testing(C::ID, beta_c.data(), circuits[0].n + 3, &srs), "mzk_srs_generate_for_testing");
    std::vector<std::unique_ptr<Prover<C>>> owned;
    std::vector<Prover<C>*> provers;
    std::vector<const BenchCircuit<C>*> cs;
    for (auto& c : circuits) {
        if (c.n != circuits[0].n) throw std::runtime_error("circuit domain size != expected domain size");
        owned.push_back(std::make_unique<Prover<C>>(srs, c));
        provers.push_back(owned.back().get());
        cs.push_back(&c);
    }
    const BatchProof<C> proof = batch_prove<C>(rng, provers, cs);
    std::printf("{\"curve\": %d, \"log_n\": %d, \"instances\": %zu, \"batch_proof_hex\": \"%s\"}\n", C::ID, circuits[0].log_n, gates.size(),
                to_hex(proof.serialize_compressed()).c_str());
    owned.clear();
    (void)mzk_srs_release(srs);
    return 0;
}

int main(int argc_in, char** argv_in) {
    // options anywhere on the line; what is left is positional
    Options opt;
    std::vector<char*> args;
    for (int i = 0; i < argc_in; i++) {
        const std::string a = argv_in[i];
        if (a == "--gpus" && i + 1 < argc_in) opt.gpus = std::atoi(argv_in[++i]);
        else if (a == "--host-witness") opt.host_witness = 1;
        else if (a == "--host-witness-vars") opt.host_witness = 2;
        else if (a == "--check-agree") opt.check_agree = true;
        else if (a == "--lagrange") opt.lagrange = 1;
        else if (a == "--no-lagrange") opt.lagrange = 0;
        else if (a == "--no-slice") opt.slice_srs = false;
        else args.push_back(argv_in[i]);
    }
    const int argc = (int)args.size();
    char** argv = args.data();
    if (opt.gpus < 1 || opt.gpus > 16) { std::fprintf(stderr, "mzk_prove: --gpus 1..16\n"); return 2; }
    if (argc < 4) { std::fprintf(stderr, "usage: %s <curve 0|1> <turbo|ultra> <num_gates> [reps] [range_bit_len] [--gpus G] [--host-witness | --host-witness-vars] [--check-agree] [--no-lagrange]\n", argv[0]); return 2; }
    const int curve = std::atoi(argv[1]);
    if (std::string(argv[2]) == "link") {
        if (argc < 8) { std::fprintf(stderr, "usage: %s <curve 0|1> link <num_gates_1> <num_gates_2> <alignment> <offset> <size> [reps]\n", argv[0]); return 2; }
        const GroupLayout layout{(uint32_t)std::atoi(argv[5]), std::strtoull(argv[6], nullptr, 10), std::strtoull(argv[7], nullptr, 10)};
        const uint64_t g1 = std::strtoull(argv[3], nullptr, 10), g2 = std::strtoull(argv[4], nullptr, 10);
        const int reps = argc > 8 ? std::atoi(argv[8]) : 0;
        try {
            return curve == 0 ? run_link<Bls12_381>(g1, g2, layout, reps) : run_link<Bn254>(g1, g2, layout, reps);
        } catch (const std::exception& e) {
            std::fprintf(stderr, "mzk_prove: %s\n", e.what());
            return 1;
        }
    }
    if (std::string(argv[2]) == "batch") {
        if (argc < 6) { std::fprintf(stderr, "usage: %s <curve 0|1> batch <turbo|ultra> <range_bit_len> <num_gates> ...\n", argv[0]); return 2; }
        std::vector<uint64_t> gates;
        for (int i = 5; i < argc; i++) gates.push_back(std::strtoull(argv[i], nullptr, 10));
        try {
            const bool u = std::string(argv[3]) == "ultra";
            return curve == 0 ? run_batch<Bls12_381>(u, std::atoi(argv[4]), gates) : run_batch<Bn254>(u, std::atoi(argv[4]), gates);
        } catch (const std::exception& e) {
            std::fprintf(stderr, "mzk_prove: %s\n", e.what());
            return 1;
        }
    }
    if (std::string(argv[2]) == "file") {
        try {
            const int reps = argc > 4 ? std::atoi(argv[4]) : 0;
            return curve == 0 ? run<Bls12_381>(false, 0, reps, 8, opt, argv[3]) : run<Bn254>(false, 0, reps, 8, opt, argv[3]);
        } catch (const std::exception& e) {
            std::fprintf(stderr, "mzk_prove: %s\n", e.what());
            return 1;
        }
    }
    const bool ultra = std::string(argv[2]) == "ultra";
    const uint64_t gates = std::strtoull(argv[3], nullptr, 10);
    const int reps = argc > 4 ? std::atoi(argv[4]) : 0, range_bits = argc > 5 ? std::atoi(argv[5]) : 8;
    try {
        return curve == 0 ? run<Bls12_381>(ultra, gates, reps, range_bits, opt) : run<Bn254>(ultra, gates, reps, range_bits, opt);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "mzk_prove: %s\n", e.what());
        return 1;
    }
}
