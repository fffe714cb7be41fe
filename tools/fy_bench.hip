// tools/fy_bench.hip -- probe for round 4: BLS12-381 Fq on 13 SIGNED limbs of 30 bits (390 bits) instead of 14 unsigned limbs of
// 29 (fx.cuh, 406 bits).  With balanced digits |l| <= 2^29 a column of 13 + 13 products stays below 2^63, so the Montgomery product
// needs 2 * 13^2 = 338 multiply-adds (v_mad_i64_i32) instead of 2 * 14^2 = 392 -- if extracting BALANCED output digits (one
// sign-extension + subtract + shift per limb instead of and + shift) does not eat the difference.  Measures both products in the
// loop shape of tools/fx_bench.hip and checks fy_mul against the 32-bit-limb Montgomery code.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/fy_bench.hip -o tools/fy_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../mpc-jellyfish_amd/csrc/fx.cuh"
using namespace mzk;

constexpr int YN = 13, YL = 30;
constexpr int32_t YP[YN] = {-21845, -402915328, 356515836, -352321620, -252304353, 55215067, 288093811, 316751073, -321428361, 517541167, -375082566, -91332614, 1704210};
constexpr uint32_t YPINV = 0x3ffcfffdu;                     // -p^-1 mod 2^30
constexpr int32_t YR2[YN] = {84936463, -82245875, 20063291, -375672600, -184045713, -75371400, -508475920, 172522421, -150322876, 98350284, 415856896, -132992156, 1010031};

struct Fy { int32_t l[YN]; };

__device__ __forceinline__ int32_t sext30(uint32_t v) { return ((int32_t)(v << 2)) >> 2; }

// a * b / 2^390 mod p, balanced digits in and out (|l_i| <= 2^29; the top limb carries the sign of the value, |value| < 2p)
__device__ __forceinline__ Fy fy_mul(const Fy& a, const Fy& b) {
    int32_t m[YN];
    Fy t;
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < YN; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (int64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (int64_t)m[i] * YP[k - i];
        m[k] = sext30(((uint32_t)acc * YPINV) & ((1u << YL) - 1));
        acc += (int64_t)m[k] * YP[0];
        acc >>= YL;                                          // exact: the low 30 bits are zero
    }
#pragma unroll
    for (int k = YN; k < 2 * YN - 1; k++) {
#pragma unroll
        for (int i = k - YN + 1; i < YN; i++) {
            acc += (int64_t)a.l[i] * b.l[k - i];
            acc += (int64_t)m[i] * YP[k - i];
        }
        const int32_t d = sext30((uint32_t)acc);
        t.l[k - YN] = d;
        acc = (acc - d) >> YL;
    }
    t.l[YN - 1] = (int32_t)acc;
    return t;
}

// canonical integer (12 x 32-bit words) -> balanced digits
__device__ __forceinline__ Fy fy_from_words(const uint32_t* w) {
    Fy r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < YN; i++) {
        const int bit = YL * i, wi = bit >> 5, s = bit & 31;
        uint32_t v = wi < 12 ? (w[wi] >> s) : 0u;
        if (s > 32 - YL && wi + 1 < 12) v |= w[wi + 1] << (32 - s);
        int32_t d = (int32_t)(v & ((1u << YL) - 1)) + c;
        c = 0;
        if (i < YN - 1 && d >= (1 << (YL - 1))) { d -= 1 << YL; c = 1; }
        r.l[i] = d;
    }
    return r;
}

// value of balanced digits as an Fp<BlsFq> element (Montgomery form), by Horner with field operations
__device__ Fp<BlsFq> fy_value(const Fy& a) {
    using F = Fp<BlsFq>;
    const F two30 = from_u64<BlsFq>(1ull << YL);
    F acc = F::zero();
    for (int i = YN - 1; i >= 0; i--) {
        acc = acc * two30;
        const int32_t d = a.l[i];
        acc = d >= 0 ? acc + from_u64<BlsFq>((uint64_t)d) : acc - from_u64<BlsFq>((uint64_t)(-(int64_t)d));
    }
    return acc;
}

__global__ void kcheck(const uint32_t* in, uint32_t* bad) {
    using F = Fp<BlsFq>;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const F a = load_fp<BlsFq>(in + (t * 2) * 12), b = load_fp<BlsFq>(in + (t * 2 + 1) * 12);       // canonical integers < p
    const Fy A = fy_from_words(a.l), B = fy_from_words(b.l);
    uint32_t err = 0;
    // one product, and a chain: ((a b / R') b / R') a / R' with the intermediate digits fed back unnormalised
    const Fy ab = fy_mul(A, B);
    const Fy chain = fy_mul(fy_mul(ab, B), A);
    F rinv = F::one();                                       // 2^-390 as a field element: (2^-30)^13
    {
        const F inv30 = inv(from_u64<BlsFq>(1ull << YL));
        for (int i = 0; i < YN; i++) rinv = rinv * inv30;
    }
    const F am = to_mont(a), bm = to_mont(b);
    if (fy_value(A) != am) err |= 1;
    if (fy_value(ab) != am * bm * rinv) err |= 2;
    if (fy_value(chain) != am * bm * bm * am * rinv * rinv * rinv) err |= 4;
    for (int i = 0; i < YN - 1; i++)
        if (ab.l[i] < -(1 << 29) || ab.l[i] >= (1 << 29) || chain.l[i] < -(1 << 29) || chain.l[i] >= (1 << 29)) err |= 8;
    bad[t] = err;
}

template <int ITERS>
__global__ __launch_bounds__(256) void kbench_fy(const uint32_t* in, uint32_t* out) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const Fp<BlsFq> a = load_fp<BlsFq>(in + ((t % 4096) * 2) * 12), b = load_fp<BlsFq>(in + ((t % 4096) * 2 + 1) * 12);
    Fy A = fy_from_words(a.l), B = fy_from_words(b.l);
#pragma unroll 1
    for (int i = 0; i < ITERS; i++) { A = fy_mul(A, B); B = fy_mul(B, A); }
#pragma unroll
    for (int i = 0; i < YN; i++) out[t * YN + i] = (uint32_t)(A.l[i] + B.l[i]);
}
template <int ITERS>
__global__ __launch_bounds__(256) void kbench_fx(const uint32_t* in, uint32_t* out) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const Fp<BlsFqX> a = load_fp<BlsFqX>(in + ((t % 4096) * 2) * 12), b = load_fp<BlsFqX>(in + ((t % 4096) * 2 + 1) * 12);
    Fx<BlsFqX> A = fx_unpack<BlsFqX>(a.l), B = fx_unpack<BlsFqX>(b.l);
#pragma unroll 1
    for (int i = 0; i < ITERS; i++) { A = fx_mul(A, B); B = fx_mul(B, A); }
#pragma unroll
    for (int i = 0; i < 14; i++) out[t * 14 + i] = A.l[i] + B.l[i];
}

int main() {
    const int NT = 64 * 256;
    std::vector<uint32_t> h((size_t)NT * 2 * 12);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (auto& w : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)(s >> 16); }
    for (int i = 0; i < NT * 2; i++) h[(size_t)i * 12 + 11] &= (BlsFq::MOD[11] >> 1);              // < p
    for (int k = 0; k < 12; k++) { h[k] = 0; h[12 + k] = BlsFq::MOD[k] - (k == 0 ? 1 : 0); h[24 + k] = k == 0 ? 1 : 0; h[36 + k] = BlsFq::MOD[k] - (k == 0 ? 1 : 0); }
    uint32_t *d_in, *d_bad, *d_out;
    (void)hipMalloc(&d_in, h.size() * 4); (void)hipMalloc(&d_bad, NT * 4); (void)hipMalloc(&d_out, (size_t)2048 * 256 * 14 * 4);
    (void)hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    kcheck<<<NT / 256, 256>>>(d_in, d_bad);
    std::vector<uint32_t> bad(NT);
    (void)hipMemcpy(bad.data(), d_bad, NT * 4, hipMemcpyDeviceToHost);
    uint32_t mask = 0, nbad = 0;
    for (uint32_t b : bad) { mask |= b; nbad += b != 0; }
    printf("BlsFq fy30 (13 signed 30-bit limbs) vs 32-bit-limb Montgomery on %d pairs: %s (mask 0x%x, %u bad)\n", NT, mask ? "MISMATCH" : "bit-exact", mask, nbad);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    constexpr int ITERS = 256;
    for (int blocks : {256, 1024, 2048}) {
        for (int v = 0; v < 2; v++) {
            float ms = 0;
            for (int r = 0; r < 2; r++) {
                (void)hipEventRecord(e0);
                if (v == 0) kbench_fx<ITERS><<<blocks, 256>>>(d_in, d_out); else kbench_fy<ITERS><<<blocks, 256>>>(d_in, d_out);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            printf("BlsFq %-22s blocks=%5d  %8.3f ms  %8.2f Gmul/s\n", v == 0 ? "fx29 (14 x 29, ships)" : "fy30 (13 x 30 signed)", blocks, ms, (double)blocks * 256 * ITERS * 2 / ms * 1e-6);
        }
    }
    return 0;
}
