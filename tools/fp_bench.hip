// tools/fp_bench.hip -- throughput + bit-exactness of the Montgomery multiplication variants on gfx950.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/fp_bench.hip -o tools/fp_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../mpc-jellyfish_amd/csrc/fp.cuh"
using namespace mzk;

template <class P, int VARIANT>
__device__ __forceinline__ Fp<P> mulv(const Fp<P>& a, const Fp<P>& b) {
    if (VARIANT == 0) return mont_mul_cios(a, b);
    if (VARIANT == 2) return sqr(a) + b;        // squaring path (different function: only timed)
    return a * b;
}

template <class P, int VARIANT, int ITERS>
__global__ __launch_bounds__(256) void kbench(const uint32_t* in, uint32_t* out) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    Fp<P> a = load_fp<P>(in + (t % 4096) * 2 * P::N), b = load_fp<P>(in + (t % 4096) * 2 * P::N + P::N);
#pragma unroll 1
    for (int i = 0; i < ITERS; i++) {
        a = mulv<P, VARIANT>(a, b);
        b = mulv<P, VARIANT>(b, a);
    }
    store_fp<P>(out + t * P::N, a + b);
}

template <class P>
__global__ void ksqrcheck(const uint32_t* in, uint32_t* out) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    Fp<P> a = load_fp<P>(in + (t % 8192) * P::N);
    Fp<P> d = sqr(a) - mont_mul_cios(a, a);
    Fp<P> e = (a * a) - mont_mul_cios(a, a);
    uint32_t bad = 0;
    for (int i = 0; i < P::N; i++) bad |= d.l[i] | e.l[i];
    out[t] = bad;
}

// same work, loop body unrolled UNR times: code size UNR x larger (instruction-cache pressure test)
template <class P, int UNR, int ITERS>
__global__ __launch_bounds__(256) void kbench_unr(const uint32_t* in, uint32_t* out) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    Fp<P> a = load_fp<P>(in + (t % 4096) * 2 * P::N), b = load_fp<P>(in + (t % 4096) * 2 * P::N + P::N);
#pragma unroll 1
    for (int i = 0; i < ITERS / UNR; i++) {
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            a = a * b;
            b = b * a;
        }
    }
    store_fp<P>(out + t * P::N, a + b);
}
template <class P, int UNR>
void run_unr(const uint32_t* d_in, uint32_t* d_out, int blocks) {
    constexpr int ITERS = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kbench_unr<P, UNR, ITERS><<<blocks, 256>>>(d_in, d_out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    kbench_unr<P, UNR, ITERS><<<blocks, 256>>>(d_in, d_out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("  unroll %2d muls/iter (%3d KB of code) blocks=%5d  %8.3f ms  %8.2f Gmul/s\n", 2 * UNR, 2 * UNR * (P::N == 12 ? 844 : 427) * 8 / 1024, blocks, ms,
           (double)blocks * 256 * ITERS * 2 / ms * 1e-6);
}

template <class P, int VARIANT>
double run(const char* name, const uint32_t* d_in, uint32_t* d_out, int blocks, std::vector<uint32_t>* result) {
    constexpr int ITERS = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kbench<P, VARIANT, ITERS><<<blocks, 256>>>(d_in, d_out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    kbench<P, VARIANT, ITERS><<<blocks, 256>>>(d_in, d_out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double muls = (double)blocks * 256 * ITERS * 2;
    printf("%-24s blocks=%5d  %8.3f ms  %8.2f Gmul/s\n", name, blocks, ms, muls / ms * 1e-6);
    if (result) {
        result->resize((size_t)blocks * 256 * P::N);
        (void)hipMemcpy(result->data(), d_out, result->size() * 4, hipMemcpyDeviceToHost);
    }
    return muls / ms * 1e-6;
}

template <class P>
void suite(const char* tag) {
    std::vector<uint32_t> h(4096 * 2 * P::N);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (auto& w : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)(s >> 16); }
    for (int i = 0; i < 4096 * 2; i++) h[(size_t)i * P::N + P::N - 1] &= (P::MOD[P::N - 1] >> 1);  // < p
    uint32_t *d_in, *d_out;
    const int maxb = 256 * 8;
    (void)hipMalloc(&d_in, h.size() * 4);
    (void)hipMalloc(&d_out, (size_t)maxb * 256 * P::N * 4);
    (void)hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<uint32_t> r0, r1;
    char name[64];
    for (int blocks : {256, 512, 1024, 2048}) {
        snprintf(name, sizeof name, "%s cios(C)", tag);
        run<P, 0>(name, d_in, d_out, blocks, blocks == 2048 ? &r0 : nullptr);
        snprintf(name, sizeof name, "%s fips(asm)", tag);
        run<P, 1>(name, d_in, d_out, blocks, blocks == 2048 ? &r1 : nullptr);
        snprintf(name, sizeof name, "%s fips sqr(+add)", tag);
        run<P, 2>(name, d_in, d_out, blocks, nullptr);
    }
    printf("%s mul bit-exact vs CIOS: %s\n", tag, r0 == r1 ? "YES" : "NO");
    for (int blocks : {768, 2048}) {
        run_unr<P, 1>(d_in, d_out, blocks);
        run_unr<P, 2>(d_in, d_out, blocks);
        run_unr<P, 4>(d_in, d_out, blocks);
        run_unr<P, 8>(d_in, d_out, blocks);
    }
    ksqrcheck<P><<<32, 256>>>(d_in, d_out);
    std::vector<uint32_t> bad(32 * 256);
    (void)hipMemcpy(bad.data(), d_out, bad.size() * 4, hipMemcpyDeviceToHost);
    uint32_t any = 0;
    for (auto v : bad) any |= v;
    printf("%s sqr/mul vs CIOS on 8192 inputs: %s\n", tag, any ? "MISMATCH" : "exact");
    (void)hipFree(d_in); (void)hipFree(d_out);
}

int main() {
    suite<BlsFr>("Fr(8)");
    suite<BlsFq>("Fq(12)");
    suite<BnFq>("BnFq(8)");
    return 0;
}
