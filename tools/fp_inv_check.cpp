// fp_inv_check.cpp -- csrc/fp_inv.cuh (inversion by Bernstein-Yang division steps) against the Fermat power (fp.cuh inv) on the host,
// both scalar fields: edge values (0 -> 0, 1, -1, 2, integers whose image is tiny or huge: 1, 2^30, 2^255, p - 1 as raw words) and
// seeded random elements; every inverse is also multiplied back.  No GPU (what tests/test_fp_inv.py builds and runs).
//   g++ -O2 -std=c++17 -I../mpc-jellyfish_amd/csrc -o fp_inv_check fp_inv_check.cpp && ./fp_inv_check
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <initializer_list>

#include "constants.cuh"
#include "fp.cuh"
#include "fp_inv.cuh"

using namespace mzk;

static uint64_t rng_state = 0x6d7a6b5f32303236ull;
static uint64_t next64() {                                   // splitmix64
    uint64_t z = (rng_state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

template <class P>
static bool below_mod(const uint32_t* w) {
    for (int i = 7; i >= 0; i--) {
        if (w[i] < P::MOD[i]) return true;
        if (w[i] > P::MOD[i]) return false;
    }
    return false;
}

template <class P>
static int check(const char* name, int n_random) {
    using F = Fp<P>;
    int bad = 0, done = 0;
    auto one_case = [&](const F& a) {
        const F want = inv(a), got = inv_safegcd(a);
        bool ok = std::memcmp(want.l, got.l, sizeof want.l) == 0;
        if (!a.is_zero()) { const F prod = a * got, one = F::one(); ok = ok && std::memcmp(prod.l, one.l, sizeof one.l) == 0; }
        else ok = ok && got.is_zero();
        if (!ok) { bad++; std::printf("%s: MISMATCH on %08x..%08x\n", name, a.l[7], a.l[0]); }
        done++;
    };
    F z = F::zero(), o = F::one();
    one_case(z); one_case(o); one_case(neg(o)); one_case(o + o);
    for (int k : {0, 1, 29, 30, 31, 59, 60, 61, 224, 239, 240, 241, 253}) {          // raw images 2^k (k < bits of p): limb boundaries of the 30-bit form
        F a = F::zero();
        a.l[k >> 5] = 1u << (k & 31);
        if (below_mod<P>(a.l)) one_case(a);
    }
    {   // raw image p - 1 and (p - 1) / 2, (p + 1) / 2
        F a;
        for (int i = 0; i < 8; i++) a.l[i] = P::MOD[i];
        a.l[0] -= 1;
        one_case(a);
        F h;
        for (int i = 0; i < 8; i++) h.l[i] = (a.l[i] >> 1) | (i < 7 ? a.l[i + 1] << 31 : 0);
        one_case(h);
        one_case(h + o);
    }
    for (int r = 0; r < n_random; r++) {
        F a;
        do {
            for (int i = 0; i < 8; i += 2) { const uint64_t v = next64(); a.l[i] = (uint32_t)v; a.l[i + 1] = (uint32_t)(v >> 32); }
            a.l[7] &= (r & 1) ? 0xffffffffu : 0x3fffffffu;                               // (odd rounds: near the top too)
        } while (!below_mod<P>(a.l));
        one_case(a);
    }
    std::printf("%s: %d cases, %d mismatches\n", name, done, bad);
    return bad;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 2000;
    const int bad = check<BlsFr>("BLS12-381 Fr", n) + check<BnFr>("BN254 Fr", n);
    std::printf("%s\n", bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
