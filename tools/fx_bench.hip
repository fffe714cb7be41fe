// tools/fx_bench.hip -- reduced-radix (29-bit limb) field arithmetic: bit-exactness against the
// 32-bit-limb Montgomery code and throughput.  hipcc -O3 -std=c++17 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../mpc-jellyfish_amd/csrc/fx.cuh"
using namespace mzk;

template <class X>
__global__ void kcheck(const uint32_t* in, uint32_t* bad) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    constexpr int W = X::N;
    Fp<X> a = load_fp<X>(in + (t * 3) * W), b = load_fp<X>(in + (t * 3 + 1) * W), c = load_fp<X>(in + (t * 3 + 2) * W);
    Fx<X> A = fx_unpack<X>(a.l), B = fx_unpack<X>(b.l), C = fx_unpack<X>(c.l);
    Fx<X> Bi = fx_mul(B, Fx<X>::from_const(X::XTO)), Ci = fx_mul(C, Fx<X>::from_const(X::XTO));
    uint32_t err = 0;
    auto cmp = [&](const Fx<X>& lazy, const Fp<X>& want, uint32_t bit) {
        Fp<X> got;
        fx_pack<X>(got.l, fx_canonical(lazy));
        if (got != want) err |= bit;
    };
    cmp(fx_mul(A, Bi), a * b, 1);
    cmp(fx_mul(fx_norm(fx_add(A, B)), Ci), (a + b) * c, 2);
    cmp(fx_mul(fx_norm(fx_sub2(A, B)), Ci), (a - b) * c, 4);
    cmp(fx_mul(Bi, Fx<X>::from_const(X::XFROM)), b, 8);
    cmp(fx_mul(fx_norm(fx_sub8(fx_add(A, B), fx_add(B, C))), Fx<X>::one()), a - c, 16);      // (a+b)-(b+c)
    Fx<X> s = fx_add(fx_add(A, B), fx_add(C, A));                                              // lazy sum < 4p
    cmp(fx_mul(fx_norm(fx_sub32(fx_norm(fx_add(s, s)), s)), Fx<X>::one()), a + a + b + c, 32);
    cmp(fx_mul(fx_norm(fx_mul(A, Bi)), Ci), (a * b) * c, 64);
    cmp(fx_mul(fx_sqr(fx_norm(fx_add(A, B))), Fx<X>::from_const(X::XTO)), (a + b) * (a + b), 128);          // (a+b)^2 R: sqr gives (a+b)^2 R^2/R'
    cmp(fx_mul(fx_sqr(Bi), Fx<X>::from_const(X::XFROM)), b * b, 256);
    bad[t] = err;
}

template <class X, int VARIANT, int ITERS>
__global__ __launch_bounds__(256) void kbench(const uint32_t* in, uint32_t* out) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    constexpr int W = X::N;
    Fp<X> a = load_fp<X>(in + (t % 4096) * 3 * W), b = load_fp<X>(in + ((t % 4096) * 3 + 1) * W);
    if (VARIANT == 0) {
#pragma unroll 1
        for (int i = 0; i < ITERS; i++) { a = a * b; b = b * a; }
        store_fp<X>(out + t * W, a + b);
    } else {
        Fx<X> A = fx_unpack<X>(a.l), B = fx_unpack<X>(b.l);
#pragma unroll 1
        for (int i = 0; i < ITERS; i++) { A = fx_mul(A, B); B = fx_mul(B, A); }
        Fp<X> r;
        fx_pack<X>(r.l, fx_canonical(fx_mul(fx_norm(fx_add(A, B)), Fx<X>::one())));
        store_fp<X>(out + t * W, r);
    }
}

template <class X, int VARIANT>
void run(const char* name, const uint32_t* d_in, uint32_t* d_out, int blocks) {
    constexpr int ITERS = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kbench<X, VARIANT, ITERS><<<blocks, 256>>>(d_in, d_out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    kbench<X, VARIANT, ITERS><<<blocks, 256>>>(d_in, d_out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s blocks=%5d  %8.3f ms  %8.2f Gmul/s\n", name, blocks, ms, (double)blocks * 256 * ITERS * 2 / ms * 1e-6);
}

template <class X>
void suite(const char* tag) {
    constexpr int W = X::N;
    const int NT = 64 * 256;
    std::vector<uint32_t> h((size_t)NT * 3 * W);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (auto& w : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)(s >> 16); }
    for (int i = 0; i < NT * 3; i++) h[(size_t)i * W + W - 1] &= (X::MOD[W - 1] >> 1);     // canonical (< p)
    // edge values in the first triples: 0, 1, p-1
    for (int k = 0; k < W; k++) { h[k] = 0; h[W + k] = X::MOD[k]; h[2 * W + k] = k == 0 ? 1 : 0; }
    h[W] -= 1;                                                                              // p - 1
    uint32_t *d_in, *d_out;
    (void)hipMalloc(&d_in, h.size() * 4);
    (void)hipMalloc(&d_out, (size_t)2048 * 256 * W * 4);
    (void)hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    kcheck<X><<<NT / 256, 256>>>(d_in, d_out);
    std::vector<uint32_t> bad(NT);
    (void)hipMemcpy(bad.data(), d_out, NT * 4, hipMemcpyDeviceToHost);
    uint32_t any = 0; int cnt = 0;
    for (auto v : bad) { any |= v; cnt += v != 0; }
    printf("%s: fx vs 32-bit-limb Montgomery on %d triples: %s (mask 0x%x, %d bad)\n", tag, NT, any ? "MISMATCH" : "bit-exact", any, cnt);
    char name[64];
    for (int blocks : {256, 1024, 2048}) {
        snprintf(name, sizeof name, "%s fips32(asm)", tag); run<X, 0>(name, d_in, d_out, blocks);
        snprintf(name, sizeof name, "%s fx29", tag); run<X, 1>(name, d_in, d_out, blocks);
    }
    (void)hipFree(d_in); (void)hipFree(d_out);
}

int main() {
    suite<BlsFrX>("BlsFr");
    suite<BnFrX>("BnFr");
    suite<BlsFqX>("BlsFq");
    suite<BnFqX>("BnFq");
    return 0;
}
