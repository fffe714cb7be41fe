// mzk_prove -- PlonkKzgSnark::prove on the reference's bench circuit from a compiled host: C++ above the C ABI of
// include/mzk.h, no Python, no HIP in this translation unit (g++ builds it).
//   mzk_prove <curve: 0 BLS12-381 | 1 BN254> <turbo|ultra> <num_gates> [reps] [range_bit_len]
// Prints one JSON line: proof bytes (hex), wall time per proof, per-round times of one profiled proof.
#include <chrono>
#include <cstdlib>

#include "mzk_prover.hpp"

using namespace mzk_host;

template <class C>
int run(bool ultra, uint64_t num_gates, int reps, int range_bits) {
    using Fr = Fp64<typename C::Fr>;
    check(mzk_init(-1), "mzk_init");
    auto t0 = std::chrono::steady_clock::now();
    BenchCircuit<C> cs = BenchCircuit<C>::generate(num_gates, ultra, range_bits);
    const double circuit_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ChaChaRng rng = test_rng();
    const Fr beta = fr_rand<typename C::Fr>(rng);                       // the SRS trapdoor: first draw of the bench's rng (bench.rs:50-54)
    const auto beta_c = canonical(beta);
    uint64_t srs = 0;
    check(mzk_srs_generate_for_testing(C::ID, beta_c.data(), cs.n + 3, &srs), "mzk_srs_generate_for_testing");
    t0 = std::chrono::steady_clock::now();
    Prover<C> prover(srs, cs);
    const double preprocess_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    Proof<C> proof = prover.prove(rng, cs);                              // the proof whose bytes are printed (and warm-up)
    const std::vector<uint8_t> bytes = proof.serialize_compressed();
    double ms = 0;
    if (reps > 0) {
        for (int i = 0; i < 2; i++) prover.prove(rng, cs);
        check(mzk_dev_sync(), "sync");
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; i++) prover.prove(rng, cs);
        check(mzk_dev_sync(), "sync");
        ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
        prover.prove(rng, cs, true);
    }
    std::string hex;
    static const char* d = "0123456789abcdef";
    for (uint8_t b : bytes) { hex.push_back(d[b >> 4]); hex.push_back(d[b & 15]); }
    std::printf("{\"curve\": %d, \"plonk_type\": \"%s\", \"num_gates\": %llu, \"log_n\": %d, \"proof_bytes\": %zu, \"prove_ms\": %.3f, "
                "\"circuit_build_s\": %.3f, \"preprocess_s\": %.3f, \"rounds_ms\": {",
                C::ID, ultra ? "UltraPlonk" : "TurboPlonk", (unsigned long long)num_gates, cs.log_n, bytes.size(), ms, circuit_s, preprocess_s);
    bool first = true;
    for (auto& kv : prover.timings_ms) { std::printf("%s\"%s\": %.3f", first ? "" : ", ", kv.first.c_str(), kv.second); first = false; }
    std::printf("}, \"proof_hex\": \"%s\"}\n", hex.c_str());
    (void)mzk_srs_release(srs);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: %s <curve 0|1> <turbo|ultra> <num_gates> [reps] [range_bit_len]\n", argv[0]); return 2; }
    const int curve = std::atoi(argv[1]);
    const bool ultra = std::string(argv[2]) == "ultra";
    const uint64_t gates = std::strtoull(argv[3], nullptr, 10);
    const int reps = argc > 4 ? std::atoi(argv[4]) : 0, range_bits = argc > 5 ? std::atoi(argv[5]) : 8;
    try {
        return curve == 0 ? run<Bls12_381>(ultra, gates, reps, range_bits) : run<Bn254>(ultra, gates, reps, range_bits);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "mzk_prove: %s\n", e.what());
        return 1;
    }
}
