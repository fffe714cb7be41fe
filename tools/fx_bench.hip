// tools/fx_bench.hip -- reduced-radix (29-bit limb) field arithmetic: bit-exactness against the
// 32-bit-limb Montgomery code and throughput.  hipcc -O3 -std=c++17 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../mpc-jellyfish_amd/csrc/fs.cuh"
using namespace mzk;

// ---- constant-operand Barrett product (fs.cuh) against the Montgomery product it replaces in the NTT -------------------------
// in: triples (a, b, c) of canonical values; tw: fs_make_tw(b) built on the host (b read as the Montgomery image of w = b / R)
template <class X>
__global__ void kcheck_fs(const uint32_t* in, const uint32_t* tw, uint32_t* bad) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    constexpr int W = X::N;
    const Fp<X> a = load_fp<X>(in + (t * 3) * W), b = load_fp<X>(in + (t * 3 + 1) * W), c = load_fp<X>(in + (t * 3 + 2) * W);
    FsTw k;
    for (int i = 0; i < FS_N; i++) { k.w[i] = (int32_t)tw[t * FS_TW_WORDS + i]; k.q[i] = (int32_t)tw[t * FS_TW_WORDS + FS_N + i]; }
    const Fs<X> A = fs_unpack<X>(a.l), C = fs_unpack<X>(c.l);
    uint32_t err = 0;
    auto cmp = [&](const Fs<X>& lazy, const Fp<X>& want, uint32_t bit) {
        Fp<X> got;
        fx_pack<X>(got.l, fs_canonical<X>(lazy));
        if (got != want) err |= bit;
    };
    cmp(fs_mulc<X>(A, k), a * b, 1);                                        // a * w  (a (x) b = a b / R = a w)
    cmp(fs_mulc<X>(fs_add(A, C), k), (a + c) * b, 2);                       // lazy sum as the multiplicand
    cmp(fs_mulc<X>(fs_sub(A, C), k), (a - c) * b, 4);                       // negative values
    const Fs<X> t1 = fs_mulc<X>(A, k);
    cmp(fs_add(C, t1), c + a * b, 8);                                       // a butterfly's two outputs
    cmp(fs_sub(C, t1), c - a * b, 16);
    cmp(fs_mulc<X>(fs_sub(fs_norm(fs_sub(C, t1)), t1), k), (c - a * b - a * b) * b, 32);   // two stages deep, then multiplied again
    cmp(A, a, 64);
    bad[t] = err;
}
template <class X, int ITERS>
__global__ __launch_bounds__(256) void kbench_fs(const uint32_t* in, const uint32_t* tw, uint32_t* out) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    constexpr int W = X::N;
    const Fp<X> a = load_fp<X>(in + (t % 4096) * 3 * W);
    FsTw k;
    for (int i = 0; i < FS_N; i++) { k.w[i] = (int32_t)tw[(t % 4096) * FS_TW_WORDS + i]; k.q[i] = (int32_t)tw[(t % 4096) * FS_TW_WORDS + FS_N + i]; }
    Fs<X> A = fs_unpack<X>(a.l), B = A;
#pragma unroll 1
    for (int i = 0; i < ITERS; i++) { A = fs_mulc<X>(A, k); B = fs_mulc<X>(fs_add(B, A), k); }
    Fp<X> r;
    fx_pack<X>(r.l, fs_canonical<X>(fs_add(A, B)));
    store_fp<X>(out + t * W, r);
}

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
    Fx<X> s = fx_norm(fx_add(fx_add(A, B), fx_add(C, A)));                                     // lazy sum < 4p, limbs < 2^29 + 8
    cmp(fx_mul(fx_norm(fx_sub32(fx_add(s, s), s)), Fx<X>::one()), a + a + b + c, 32);          // (2s + 32p) - s
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
    if constexpr (X::XN == FS_N && X::N == 8) {
        std::vector<uint32_t> htw((size_t)NT * FS_TW_WORDS, 0);
        for (int i = 0; i < NT; i++) {
            Fp<X> b;
            for (int k = 0; k < W; k++) b.l[k] = h[((size_t)i * 3 + 1) * W + k];
            const FsTw t = fs_make_tw<X>(b);
            for (int k = 0; k < FS_N; k++) { htw[(size_t)i * FS_TW_WORDS + k] = (uint32_t)t.w[k]; htw[(size_t)i * FS_TW_WORDS + FS_N + k] = (uint32_t)t.q[k]; }
        }
        uint32_t* d_tw;
        (void)hipMalloc(&d_tw, htw.size() * 4);
        (void)hipMemcpy(d_tw, htw.data(), htw.size() * 4, hipMemcpyHostToDevice);
        kcheck_fs<X><<<NT / 256, 256>>>(d_in, d_tw, d_out);
        (void)hipMemcpy(bad.data(), d_out, NT * 4, hipMemcpyDeviceToHost);
        any = 0; cnt = 0;
        for (auto v : bad) { any |= v; cnt += v != 0; }
        printf("%s: fs_mulc (constant-operand Barrett, signed lazy limbs) vs Montgomery on %d triples: %s (mask 0x%x, %d bad)\n", tag, NT,
               any ? "MISMATCH" : "bit-exact", any, cnt);
        for (int blocks : {1024, 2048}) {
            constexpr int ITERS = 256;
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            kbench_fs<X, ITERS><<<blocks, 256>>>(d_in, d_tw, d_out);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            kbench_fs<X, ITERS><<<blocks, 256>>>(d_in, d_tw, d_out);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%s fs_mulc                   blocks=%5d  %8.3f ms  %8.2f Gmul/s\n", tag, blocks, ms, (double)blocks * 256 * ITERS * 2 / ms * 1e-6);
        }
        (void)hipFree(d_tw);
    }
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
