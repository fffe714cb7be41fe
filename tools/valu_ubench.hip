// tools/valu_ubench.hip -- integer/fp64 VALU issue-rate microbenchmark for gfx950.
// Decides the limb width / multiplier instruction for the 256/384-bit Montgomery kernels.
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_ubench.hip -o tools/valu_ubench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHAINS 8
#define ITERS 4096

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
    uint32_t a[CHAINS], b[CHAINS];
    uint64_t c[CHAINS];
    double d[CHAINS], e[CHAINS];
    for (int i = 0; i < CHAINS; i++) {
        a[i] = seed * (threadIdx.x + 1 + i) + 12345u;
        b[i] = seed ^ (0x9E3779B9u * (i + 1 + blockIdx.x));
        c[i] = ((uint64_t)a[i] << 20) ^ b[i];
        d[i] = 1.0 + 1e-9 * a[i];
        e[i] = 1e-3 * b[i];
    }
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
            if (OP == 0) {
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
            } else if (OP == 1) {
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            } else if (OP == 2) {
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            } else if (OP == 3) {
                asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            } else if (OP == 4) {
                asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            } else if (OP == 5) {
                asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(e[i]));
            } else if (OP == 6) {
                asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a[i]), "+v"(b[i]) : "v"(a[(i+1)%CHAINS]), "v"(b[(i+1)%CHAINS]) : "vcc");
            } else if (OP == 7) {
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            } else if (OP == 8) {
                asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
            } else if (OP == 9) {
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_add_co_u32 %1, vcc, %1, %2\n\tv_addc_co_u32 %2, vcc, 0, %2, vcc" : "+v"(c[i]), "+v"(a[i]), "+v"(b[i]) : : "vcc");
            } else if (OP == 10) {
                asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            } else if (OP == 11) {
                asm volatile("v_pk_mad_u16 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            } else if (OP == 12) {
                asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            } else if (OP == 13) {
                asm volatile("v_mad_u64_u32 %0, %3, %1, %2, %0" : "+v"(c[i]) : "v"(a[i]), "v"(b[i]), "s"(0ull) : );
            }
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < CHAINS; i++) r ^= a[i] ^ (uint32_t)c[i] ^ (uint32_t)(c[i] >> 32) ^ (uint32_t)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
void run(const char* name, uint32_t* d_out, int blocks) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d_out, 7u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d_out, 11u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 256 * ITERS * CHAINS;
    // per CU per cycle @2.4 GHz, 256 CUs
    printf("%-28s %8.3f ms  %8.2f Gop/s   %6.2f lane-ops/clk/CU(@2.4GHz)  wave-instr cycles/SIMD ~ %.2f\n", name, ms,
           ops / ms * 1e-6, ops / (ms * 1e-3) / 256 / 2.4e9, 64.0 / (ops / (ms * 1e-3) / 256 / 2.4e9 / 4));
}

int main() {
    int blocks = 256 * 8;
    uint32_t* d_out;
    hipMalloc(&d_out, (size_t)blocks * 256 * 4);
    run<0>("v_mad_u64_u32", d_out, blocks);
    run<1>("v_mul_lo_u32", d_out, blocks);
    run<2>("v_mul_hi_u32", d_out, blocks);
    run<3>("v_mad_u32_u24", d_out, blocks);
    run<4>("v_mul_hi_u32_u24", d_out, blocks);
    run<5>("v_fma_f64", d_out, blocks);
    run<6>("add64 (add_co+addc)", d_out, blocks);
    run<7>("v_add_u32", d_out, blocks);
    run<8>("v_add3_u32", d_out, blocks);
    run<9>("mad64+carry-add (CIOS step)", d_out, blocks);
    run<10>("v_fma_f32", d_out, blocks);
    run<11>("v_pk_mad_u16", d_out, blocks);
    run<12>("v_mul_u32_u24", d_out, blocks);
    hipFree(d_out);
    return 0;
}
