// tools/fs_check.cpp -- host-side check of fs.cuh (signed lazy limbs, constant-operand Barrett product): prints test cases as
// lines of integers for tools/fs_check.py, which verifies them with Python big-ints.  g++ -O2 -std=c++17 -I mpc-jellyfish_amd/csrc
#include <cstdio>
#include <cstdint>
#include <random>
#include "fs.cuh"
using namespace mzk;

template <class X>
void run(const char* tag, int cases) {
    std::mt19937_64 rng(12345);
    for (int c = 0; c < cases; c++) {
        Fp<X> a;                                                      // a random canonical constant (as a Montgomery image: any value < p works)
        for (int i = 0; i < 8; i++) a.l[i] = (uint32_t)rng();
        a.l[7] &= X::MOD[7] >> 1;
        if (c == 0) for (int i = 0; i < 8; i++) a.l[i] = 0;                                   // w = 0
        if (c == 1) { for (int i = 0; i < 8; i++) a.l[i] = X::MOD[i]; a.l[0] -= 1; }          // image p - 1
        if (c == 2) for (int i = 0; i < 8; i++) a.l[i] = X::R1[i];                            // w = 1
        const FsTw t = fs_make_tw<X>(a);
        Fs<X> x;                                                       // a lazy multiplicand: limbs up to +-2^30, |value| < 2^261
        const int mode = c % 4;
        for (int i = 0; i < FS_N; i++) {
            int64_t v;
            if (mode == 0) v = (int64_t)(rng() & XMASK);                                      // class C, non-negative
            else if (mode == 1) v = (int64_t)(rng() % ((1ull << 31) + 1)) - (1ll << 30);      // anything in [-2^30, 2^30]
            else if (mode == 2) v = (rng() & 1) ? (1ll << 30) : -(1ll << 30);                 // extremes
            else v = (int64_t)(rng() & XMASK) - (int64_t)(rng() & XMASK);
            x.l[i] = (int32_t)v;
        }
        // keep |value| < 2^261: the top limb decides
        x.l[FS_N - 1] = (int32_t)((int64_t)(rng() % ((1ull << 29) - 8)) * ((rng() & 1) ? 1 : -1));
        if (mode == 2) x.l[FS_N - 1] = (c & 4) ? ((1 << 29) - 9) : -((1 << 29) - 9);
        const Fs<X> r = fs_mulc<X>(x, t);
        const Fx<X> cx = fs_canonical<X>(x), cr = fs_canonical<X>(r);
        std::printf("%s", tag);
        for (int i = 0; i < FS_N; i++) std::printf(" %d", t.w[i]);
        for (int i = 0; i < FS_N; i++) std::printf(" %d", t.q[i]);
        for (int i = 0; i < FS_N; i++) std::printf(" %d", x.l[i]);
        for (int i = 0; i < FS_N; i++) std::printf(" %d", r.l[i]);
        for (int i = 0; i < FS_N; i++) std::printf(" %u", cx.l[i]);
        for (int i = 0; i < FS_N; i++) std::printf(" %u", cr.l[i]);
        std::printf("\n");
    }
}

// fx_inv (windowed Fermat inverse on the reduced-radix product, fx.cuh) against the bitwise power on 32-bit limbs (fp.cuh inv)
template <class X>
int check_inv(const char* tag, int cases) {
    std::mt19937_64 rng(777);
    int bad = 0;
    for (int c = 0; c < cases; c++) {
        Fp<X> a;
        for (int i = 0; i < 8; i++) a.l[i] = (uint32_t)rng();
        a.l[7] &= X::MOD[7] >> 1;
        if (c == 0) for (int i = 0; i < 8; i++) a.l[i] = 0;                                   // inv(0) = 0
        if (c == 1) for (int i = 0; i < 8; i++) a.l[i] = X::R1[i];                            // 1
        if (c == 2) { for (int i = 0; i < 8; i++) a.l[i] = X::MOD[i]; a.l[0] -= 1; }          // image p - 1
        const Fp<X> want = inv(a);
        const Fp<X> got = fx_to_boundary<X>(fx_inv<X>(fx_from_boundary<X>(a)));
        for (int i = 0; i < 8; i++) if (want.l[i] != got.l[i]) { bad++; break; }
    }
    if (bad) std::fprintf(stderr, "%s: fx_inv differs from inv in %d of %d cases\n", tag, bad, cases);
    return bad;
}

int main(int argc, char** argv) {
    const int cases = argc > 1 ? std::atoi(argv[1]) : 20000;
    if (check_inv<BlsFrX>("bls", 300) + check_inv<BnFrX>("bn", 300)) return 1;
    run<BlsFrX>("bls", cases);
    run<BnFrX>("bn", cases);
    return 0;
}
