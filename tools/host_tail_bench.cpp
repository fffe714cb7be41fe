// host_tail_bench.cpp -- times the MSM's host Horner tail (csrc/msm.hip host_horner) on the CPU it runs on, for the shapes the
// library meets: the plain path's 16 windows x 16 bit-sums (bench.py's headline) and the table path's single bucket set.
//   g++ -O3 -std=c++17 -pthread -I../mpc-jellyfish_amd/csrc -o host_tail_bench host_tail_bench.cpp && ./host_tail_bench
// No GPU, no library: includes the product's host headers only (hostfp.hpp, ec.cuh, host_tail.hpp).
#include <chrono>
#include <cstdio>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "hostfp.hpp"
#include "ec.cuh"
#include "host_tail.hpp"

using namespace mzk;

template <class FQ>
static void run(const char* name, int n_win, int c, int reps) {
    using F = Fp64<FQ>;
    const int per = c, W4 = 4 * FQ::N;
    std::vector<uint32_t> pts((size_t)n_win * per * W4);
    Affine<F> g;
    g.x = F::from_words(FQ::GEN_X); g.y = F::from_words(FQ::GEN_Y);
    XYZZ<F> p = XYZZ<F>::from_affine(g);
    for (int i = 0; i < n_win * per; i++) {                       // distinct, non-trivial XYZZ points (zz != 1)
        p = xyzz_dbl(p);
        p = xyzz_madd(p, g);
        uint32_t* d = pts.data() + (size_t)i * W4;
        p.x.to_words(d); p.y.to_words(d + FQ::N); p.zz.to_words(d + 2 * FQ::N); p.zzz.to_words(d + 3 * FQ::N);
    }
    std::vector<uint32_t> a(3 * FQ::N), b(3 * FQ::N);
    auto time_it = [&](auto fn) {
        fn();
        double best = 1e30;
        for (int r = 0; r < reps; r++) {
            auto t0 = std::chrono::steady_clock::now();
            fn();
            best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
        return best;
    };
    const double t_serial = time_it([&] { host_horner_serial<FQ>(pts.data(), n_win, c, a.data()); });
    const double t_par = time_it([&] { host_horner<FQ>(pts.data(), n_win, c, b.data()); });
    // same POINT (the Jacobian representatives differ with the order of operations): compare x/z^2, y/z^3 cross-multiplied
    auto same = [&] {
        F X1 = F::from_words(a.data()), Y1 = F::from_words(a.data() + FQ::N), Z1 = F::from_words(a.data() + 2 * FQ::N);
        F X2 = F::from_words(b.data()), Y2 = F::from_words(b.data() + FQ::N), Z2 = F::from_words(b.data() + 2 * FQ::N);
        F z1s = sqr(Z1), z2s = sqr(Z2);
        return X1 * z2s == X2 * z1s && Y1 * (z2s * Z2) == Y2 * (z1s * Z1);
    };
    std::printf("%-10s n_win %2d c %2d: serial %7.1f us, host_horner (pool of %d) %7.1f us, same point: %s\n", name, n_win, c, t_serial, host_tail_pool_size(), t_par,
                same() ? "yes" : "NO");
}

// a commit group: `count` MSMs of one bucket set each (what a round of a proof ends with)
template <class FQ>
static void run_group(const char* name, int count, int c, int reps) {
    using F = Fp64<FQ>;
    const int W4 = 4 * FQ::N;
    const size_t per = (size_t)c * W4;
    std::vector<uint32_t> pts(count * per);
    Affine<F> g;
    g.x = F::from_words(FQ::GEN_X); g.y = F::from_words(FQ::GEN_Y);
    XYZZ<F> p = XYZZ<F>::from_affine(g);
    for (int i = 0; i < count * c; i++) {
        p = xyzz_dbl(p);
        p = xyzz_madd(p, g);
        uint32_t* d = pts.data() + (size_t)i * W4;
        p.x.to_words(d); p.y.to_words(d + FQ::N); p.zz.to_words(d + 2 * FQ::N); p.zzz.to_words(d + 3 * FQ::N);
    }
    std::vector<uint32_t> a(count * 3 * FQ::N), b(count * 3 * FQ::N);
    std::vector<uint32_t*> oa(count), ob(count);
    for (int q = 0; q < count; q++) { oa[q] = a.data() + (size_t)q * 3 * FQ::N; ob[q] = b.data() + (size_t)q * 3 * FQ::N; }
    auto time_it = [&](auto fn) {
        fn();
        double best = 1e30;
        for (int r = 0; r < reps; r++) {
            auto t0 = std::chrono::steady_clock::now();
            fn();
            best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
        return best;
    };
    const double t_serial = time_it([&] { for (int q = 0; q < count; q++) host_horner_serial<FQ>(pts.data() + q * per, 1, c, oa[q]); });
    const double t_par = time_it([&] { host_horner_batch<FQ>(pts.data(), per, count, 1, c, ob.data()); });
    std::printf("%-10s group of %d, c %2d: serial %7.1f us, host_horner_batch (pool of %d) %7.1f us, same point: %s\n", name, count, c, t_serial, host_tail_pool_size(),
                t_par, a == b ? "yes" : "NO");      // (a single set's tail is the same sequence of operations on either path: same representative)
}

// `callers` threads ask for the same tail at once, again and again: one of them gets the pool, the others fall back to the serial
// order -- every answer must be one of the two representatives of the same point (what tests/test_host_tail.py runs under TSan)
template <class FQ>
static bool stress(int n_win, int c, int callers, int rounds) {
    using F = Fp64<FQ>;
    const int W4 = 4 * FQ::N;
    std::vector<uint32_t> pts((size_t)n_win * c * W4);
    Affine<F> g;
    g.x = F::from_words(FQ::GEN_X); g.y = F::from_words(FQ::GEN_Y);
    XYZZ<F> p = XYZZ<F>::from_affine(g);
    for (int i = 0; i < n_win * c; i++) {
        p = xyzz_dbl(p);
        if (i % 7 != 3) p = xyzz_madd(p, g);
        uint32_t* d = pts.data() + (size_t)i * W4;
        if (i % 11 == 5) { XYZZ<F> z = XYZZ<F>::inf(); z.x.to_words(d); z.y.to_words(d + FQ::N); z.zz.to_words(d + 2 * FQ::N); z.zzz.to_words(d + 3 * FQ::N); continue; }   // an empty bit-sum
        p.x.to_words(d); p.y.to_words(d + FQ::N); p.zz.to_words(d + 2 * FQ::N); p.zzz.to_words(d + 3 * FQ::N);
    }
    std::vector<uint32_t> ser(3 * FQ::N), par(3 * FQ::N);
    host_horner_serial<FQ>(pts.data(), n_win, c, ser.data());
    host_horner<FQ>(pts.data(), n_win, c, par.data());
    std::atomic<int> bad{0};
    std::vector<std::thread> th;
    for (int t = 0; t < callers; t++)
        th.emplace_back([&] {
            std::vector<uint32_t> o(3 * FQ::N);
            for (int r = 0; r < rounds; r++) {
                host_horner<FQ>(pts.data(), n_win, c, o.data());
                if (o != ser && o != par) bad++;
            }
        });
    // ... and groups of whole tails (host_horner_batch) beside them
    const size_t per = (size_t)n_win * c * W4;
    std::vector<uint32_t> grp(3 * per);
    for (int q = 0; q < 3; q++) std::memcpy(grp.data() + q * per, pts.data(), per * 4);
    for (int t = 0; t < 2; t++)
        th.emplace_back([&] {
            std::vector<uint32_t> o(3 * 3 * FQ::N);
            uint32_t* outs[3] = {o.data(), o.data() + 3 * FQ::N, o.data() + 6 * FQ::N};
            for (int r = 0; r < rounds; r++) {
                host_horner_batch<FQ>(grp.data(), per, 3, n_win, c, outs);
                for (int q = 0; q < 3; q++)
                    if (std::memcmp(outs[q], ser.data(), 3 * FQ::N * 4) != 0) bad++;        // (whole tails of a group run in the serial order)
            }
        });
    for (auto& t : th) t.join();
    return bad == 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::strcmp(argv[1], "--stress") == 0) {
        const bool ok = stress<BlsFq>(16, 16, 4, 40) && stress<BnFq>(17, 15, 3, 40) && stress<BlsFq>(5, 20, 2, 40);
        std::printf("stress: %s\n", ok ? "ok" : "MISMATCH");
        return ok ? 0 : 1;
    }
    run<BlsFq>("BLS12-381", 16, 16, 200);
    run<BlsFq>("BLS12-381", 1, 20, 200);
    run<BlsFq>("BLS12-381", 5, 20, 200);
    run<BnFq>("BN254", 16, 16, 200);
    run<BnFq>("BN254", 1, 20, 200);
    run_group<BlsFq>("BLS12-381", 5, 20, 200);
    run_group<BlsFq>("BLS12-381", 2, 20, 200);
    run_group<BnFq>("BN254", 6, 20, 200);
    return 0;
}
