// tools/valu_ubench2.hip -- second round: carry ops with private SGPR carries, 64-bit helpers.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHAINS 8
#define ITERS 4096
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
    uint32_t a[CHAINS], b[CHAINS];
    uint64_t c[CHAINS];
    for (int i = 0; i < CHAINS; i++) {
        a[i] = seed * (threadIdx.x + 1 + i) + 12345u;
        b[i] = seed ^ (0x9E3779B9u * (i + 1 + blockIdx.x));
        c[i] = ((uint64_t)a[i] << 20) ^ b[i];
    }
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
        if (OP == 0) {  // v_add_co_u32 VOP3, private carry regs, independent
            asm volatile("v_add_co_u32 %0, s[20:21], %0, %8\n\tv_add_co_u32 %1, s[22:23], %1, %9\n\tv_add_co_u32 %2, s[24:25], %2, %10\n\tv_add_co_u32 %3, s[26:27], %3, %11\n\t"
                         "v_add_co_u32 %4, s[28:29], %4, %12\n\tv_add_co_u32 %5, s[30:31], %5, %13\n\tv_add_co_u32 %6, s[32:33], %6, %14\n\tv_add_co_u32 %7, s[34:35], %7, %15"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                         : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7])
                         : "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","s32","s33","s34","s35");
        } else if (OP == 1) {  // v_addc_co_u32 VOP3 reading/writing private carry regs (independent chains)
            asm volatile("v_addc_co_u32 %0, s[20:21], %0, %8, s[20:21]\n\tv_addc_co_u32 %1, s[22:23], %1, %9, s[22:23]\n\tv_addc_co_u32 %2, s[24:25], %2, %10, s[24:25]\n\tv_addc_co_u32 %3, s[26:27], %3, %11, s[26:27]\n\t"
                         "v_addc_co_u32 %4, s[28:29], %4, %12, s[28:29]\n\tv_addc_co_u32 %5, s[30:31], %5, %13, s[30:31]\n\tv_addc_co_u32 %6, s[32:33], %6, %14, s[32:33]\n\tv_addc_co_u32 %7, s[34:35], %7, %15, s[34:35]"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                         : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7])
                         : "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","s32","s33","s34","s35");
        } else if (OP == 2) {  // v_lshl_add_u64
            for (int i = 0; i < CHAINS; i++) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c[i]) : "v"(c[(i + 1) % CHAINS]));
        } else if (OP == 3) {  // mad with private carry sgprs, independent
            asm volatile("v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n\tv_mad_u64_u32 %1, s[22:23], %10, %11, %1\n\tv_mad_u64_u32 %2, s[24:25], %12, %13, %2\n\tv_mad_u64_u32 %3, s[26:27], %14, %15, %3\n\t"
                         "v_mad_u64_u32 %4, s[28:29], %8, %11, %4\n\tv_mad_u64_u32 %5, s[30:31], %10, %13, %5\n\tv_mad_u64_u32 %6, s[32:33], %12, %15, %6\n\tv_mad_u64_u32 %7, s[34:35], %14, %9, %7"
                         : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7])
                         : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[1]), "v"(a[2]), "v"(b[2]), "v"(a[3]), "v"(b[3])
                         : "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","s32","s33","s34","s35");
        } else if (OP == 4) {  // interleaved mac96 on 4 independent accumulators with private carries: 4 mads then 4 addcs
            asm volatile("v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n\tv_mad_u64_u32 %1, s[22:23], %10, %11, %1\n\tv_mad_u64_u32 %2, s[24:25], %12, %13, %2\n\tv_mad_u64_u32 %3, s[26:27], %14, %15, %3\n\t"
                         "v_addc_co_u32 %4, s[20:21], 0, %4, s[20:21]\n\tv_addc_co_u32 %5, s[22:23], 0, %5, s[22:23]\n\tv_addc_co_u32 %6, s[24:25], 0, %6, s[24:25]\n\tv_addc_co_u32 %7, s[26:27], 0, %7, s[26:27]"
                         : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                         : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[1]), "v"(a[2]), "v"(b[2]), "v"(a[3]), "v"(b[3])
                         : "s20","s21","s22","s23","s24","s25","s26","s27");
        } else if (OP == 5) {  // serial mac96 on ONE accumulator via vcc: 4 x (mad, addc)
            asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
                         "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
                         : "+v"(c[0]), "+v"(a[4])
                         : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[1]), "v"(a[2]), "v"(b[2]), "v"(a[3]), "v"(b[3]) : "vcc");
        } else if (OP == 6) {  // v_add_u32 VOP2 baseline, 8 independent
            for (int i = 0; i < CHAINS; i++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        } else if (OP == 7) {  // v_and_b32
            for (int i = 0; i < CHAINS; i++) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        } else if (OP == 8) {  // v_lshrrev_b64
            for (int i = 0; i < CHAINS; i++) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(c[i]));
        } else if (OP == 9) {  // v_alignbit_b32
            for (int i = 0; i < CHAINS; i++) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(b[i]));
        } else if (OP == 10) {  // v_mad_u64_u32 with vcc dst but independent data (8 chains)
            for (int i = 0; i < CHAINS; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
        } else if (OP == 11) {  // v_mul_lo_u32 + v_mul_hi_u32 pair
            for (int i = 0; i < CHAINS; i += 2) asm volatile("v_mul_lo_u32 %0, %2, %3\n\tv_mul_hi_u32 %1, %2, %3" : "+v"(a[i]), "+v"(a[i + 1]) : "v"(b[i]), "v"(b[i + 1]));
        } else if (OP == 12) {  // v_bfe_u32
            for (int i = 0; i < CHAINS; i++) asm volatile("v_bfe_u32 %0, %0, 3, 29" : "+v"(a[i]));
        } else if (OP == 13) {  // v_mad_u32_u24 -> used for 24-bit limb ideas
            for (int i = 0; i < CHAINS; i++) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
        } else if (OP == 14) {  // v_add_co_u32 VOP2 via vcc, 8 in a row (independent data, shared vcc)
            for (int i = 0; i < CHAINS; i++) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b[i]) : "vcc");
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < CHAINS; i++) r ^= a[i] ^ b[i] ^ (uint32_t)c[i] ^ (uint32_t)(c[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP>
void run(const char* name, uint32_t* d_out, int blocks, int instr_per_iter) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d_out, 7u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; r++) k<OP><<<blocks, 256>>>(d_out, 11u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    double winstr = (double)blocks * 4 * ITERS * instr_per_iter;      // wave-instructions
    printf("%-46s %8.3f ms   %.3f ns per wave-instr per SIMD\n", name, ms, ms * 1e6 / (winstr / 1024));
}
int main() {
    int blocks = 256 * 8;
    uint32_t* d_out;
    (void)hipMalloc(&d_out, (size_t)blocks * 256 * 4);
    run<6>("v_add_u32 (VOP2)", d_out, blocks, 8);
    run<7>("v_and_b32 (VOP2)", d_out, blocks, 8);
    run<14>("v_add_co_u32 via vcc", d_out, blocks, 8);
    run<0>("v_add_co_u32 e64, private sgpr carry", d_out, blocks, 8);
    run<1>("v_addc_co_u32 e64, private sgpr carry", d_out, blocks, 8);
    run<2>("v_lshl_add_u64", d_out, blocks, 8);
    run<8>("v_lshrrev_b64", d_out, blocks, 8);
    run<9>("v_alignbit_b32", d_out, blocks, 8);
    run<12>("v_bfe_u32", d_out, blocks, 8);
    run<10>("v_mad_u64_u32 (vcc dst, independent)", d_out, blocks, 8);
    run<3>("v_mad_u64_u32 (private sgpr dst)", d_out, blocks, 8);
    run<4>("4x mad + 4x addc interleaved (private carries)", d_out, blocks, 8);
    run<5>("4x (mad, addc) serial on one acc via vcc", d_out, blocks, 8);
    run<11>("v_mul_lo_u32 + v_mul_hi_u32", d_out, blocks, 8);
    run<13>("v_mad_u32_u24", d_out, blocks, 8);
    return 0;
}
