// host_inv_bench.cpp -- csrc/hostinv.hpp (the host's inversion by Bernstein-Yang division steps, h64::inv since round 5) against the Fermat
// power (h64::inv_fermat) in all four fields: edge values (0 -> 0, +-1, 2, raw images 2^k at the 30-bit limb boundaries, p - 1,
// (p +- 1) / 2) and seeded random elements, every inverse multiplied back; then the time of both.  No GPU (tests/test_host_inv.py builds
// and runs it under UBSan).
//   g++ -O2 -std=c++17 -I../mpc-jellyfish_amd/csrc -o host_inv_bench host_inv_bench.cpp && ./host_inv_bench [n_random] [time]
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "hostfp.hpp"

using namespace mzk;

static uint64_t rng_state = 0x6d7a6b5f32303236ull;
static uint64_t next64() {                                   // splitmix64
    uint64_t z = (rng_state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

template <class P>
static Fp64<P> random_element(bool near_top) {
    using F = Fp64<P>;
    F a;
    do {
        for (int i = 0; i < F::N; i++) a.l[i] = next64();
        a.l[F::N - 1] &= near_top ? ~0ull >> (64 - (P::BITS - 64 * (F::N - 1))) : ~0ull >> (66 - (P::BITS - 64 * (F::N - 1)));
    } while (F::geq_mod(a.l));
    return a;
}

template <class P>
static int check(const char* name, int n_random, bool timing) {
    using F = Fp64<P>;
    int bad = 0, done = 0, max_batches = 0;
    long sum_batches = 0;
    auto one_case = [&](const F& a) {
        const F want = h64::inv_fermat(a), got = h64::inv(a);
        bool ok = want == got;
        if (!a.is_zero()) {
            ok = ok && a * got == F::one();
            uint32_t x[P::N], y[P::N];
            int b = 0;
            a.to_words(x);
            if (!hinv::inverse_words<P>(x, y, &b)) { ok = false; std::printf("%s: step cap reached\n", name); }
            if (b > max_batches) max_batches = b;
            sum_batches += b;
        } else ok = ok && got.is_zero();
        if (!ok) { bad++; std::printf("%s: MISMATCH on %016llx..%016llx\n", name, (unsigned long long)a.l[F::N - 1], (unsigned long long)a.l[0]); }
        done++;
    };
    const F z = F::zero(), o = F::one();
    one_case(z); one_case(o); one_case(neg(o)); one_case(o + o);
    for (int k = 0; k < P::BITS; k++) {                     // raw images 2^k: every bit position, so every limb boundary of the 30-bit form
        F a = F::zero();
        a.l[k >> 6] = 1ull << (k & 63);
        if (!F::geq_mod(a.l)) one_case(a);
    }
    {   // raw images p - 1, (p - 1) / 2, (p + 1) / 2
        F a;
        for (int i = 0; i < F::N; i++) a.l[i] = F::mod(i);
        a.l[0] -= 1;
        one_case(a);
        F h;
        for (int i = 0; i < F::N; i++) h.l[i] = (a.l[i] >> 1) | (i < F::N - 1 ? a.l[i + 1] << 63 : 0);
        one_case(h);
        F h1 = h;
        h1.l[0] += 1;                                       // (p - 1) / 2 is even in its low limb or not: no carry either way for these moduli
        if (h1.l[0] != 0) one_case(h1);
    }
    for (int r = 0; r < n_random; r++) one_case(random_element<P>(r & 1));
    std::printf("%s: %d cases, %d mismatches; batches of 30 division steps: mean %.1f, max %d\n", name, done, bad, (double)sum_batches / done, max_batches);
    if (timing) {
        std::vector<F> xs;
        for (int i = 0; i < 2000; i++) xs.push_back(random_element<P>(true));
        F acc = F::zero();
        auto t0 = std::chrono::steady_clock::now();
        for (const F& x : xs) acc = acc + h64::inv_fermat(x);
        auto t1 = std::chrono::steady_clock::now();
        for (const F& x : xs) acc = acc - h64::inv(x);
        auto t2 = std::chrono::steady_clock::now();
        const double fm = std::chrono::duration<double, std::micro>(t1 - t0).count() / xs.size(), ds = std::chrono::duration<double, std::micro>(t2 - t1).count() / xs.size();
        std::printf("%s: Fermat power %.2f us, division steps %.2f us per inversion (%s)\n", name, fm, ds, acc.is_zero() ? "same sums" : "SUMS DIFFER");
        if (!acc.is_zero()) bad++;
    }
    return bad;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 2000;
    const bool timing = argc > 2;
    const int bad = check<BlsFr>("BLS12-381 Fr", n, timing) + check<BnFr>("BN254 Fr", n, timing) + check<BlsFq>("BLS12-381 Fq", n, timing) +
                    check<BnFq>("BN254 Fq", n, timing);
    std::printf("%s\n", bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
