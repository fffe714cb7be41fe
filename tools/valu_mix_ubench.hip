// tools/valu_mix_ubench.hip -- do simple VALU instructions cost issue time BESIDE v_mad_u64_u32?  Four independent multiply-add chains
// per lane (as clang schedules fx_mul), with 0, 1 or 2 simple 32-bit instructions (v_and / v_add) or one 64-bit add (v_lshl_add_u64) after
// every multiply-add; 1, 2, 4 waves per SIMD.  Time per multiply-add tells whether the extras ride for free.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/valu_mix_ubench.hip -o tools/valu_mix_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
template <int EXTRA>
__global__ __launch_bounds__(256) void k(unsigned long long* out, unsigned seed, int iters) {
    unsigned long long a0 = threadIdx.x, a1 = seed, a2 = blockIdx.x, a3 = 7;
    unsigned x = threadIdx.x * 2654435761u + seed, y = x ^ 0x9e3779b9u, e0 = x, e1 = y;
    unsigned long long w = seed;
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y) : "vcc");
            if (EXTRA == 1 || EXTRA == 2) asm volatile("v_and_b32 %0, %1, %0" : "+v"(e0) : "v"(y));
            if (EXTRA == 3) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w) : "v"(a3));
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a1) : "v"(y), "v"(x) : "vcc");
            if (EXTRA == 2) asm volatile("v_add_u32 %0, %1, %0" : "+v"(e1) : "v"(x));
            if (EXTRA == 3) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w) : "v"(a2));
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a2) : "v"(x), "v"(x) : "vcc");
            if (EXTRA == 1 || EXTRA == 2) asm volatile("v_and_b32 %0, %1, %0" : "+v"(e0) : "v"(x));
            if (EXTRA == 3) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w) : "v"(a1));
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a3) : "v"(y), "v"(y) : "vcc");
            if (EXTRA == 2) asm volatile("v_add_u32 %0, %1, %0" : "+v"(e1) : "v"(y));
            if (EXTRA == 3) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w) : "v"(a0));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + e0 + e1 + w;
}
int main() {
    unsigned long long* d;
    (void)hipMalloc(&d, 8ull * 256 * 4096 * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000;
    const char* names[4] = {"mads only", "+1 simple per 2 mads", "+1 simple per mad", "+1 v_lshl_add_u64 per mad"};
    for (int wps : {1, 2, 4}) {                       // waves per SIMD: blocks of 256 threads = 4 waves = one per SIMD of a CU
        const int blocks = 256 * wps;
        for (int v = 0; v < 4; v++) {
            float ms = 0;
            for (int r = 0; r < 2; r++) {
                (void)hipEventRecord(e0);
                if (v == 0) k<0><<<blocks, 256>>>(d, 1, iters);
                if (v == 1) k<1><<<blocks, 256>>>(d, 1, iters);
                if (v == 2) k<2><<<blocks, 256>>>(d, 1, iters);
                if (v == 3) k<3><<<blocks, 256>>>(d, 1, iters);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
            }
            const double mads_per_simd = (double)wps * iters * 64;
            printf("%d wave(s) per SIMD, %-28s %8.3f ms  %6.2f ns per multiply-add per SIMD\n", wps, names[v], ms, ms * 1e6 / mads_per_simd);
        }
    }
    return 0;
}
