// tools/affine_bench.hip -- VERDICT r2 #6: what batched-AFFINE bucket accumulation would cost on gfx950, measured in the form of
// tools/madd_bench.hip (cache-resident points, no sort, no divergence), against the XYZZ mixed add the library runs.
//
// Affine P + Q: lambda = (y2 - y1) / (x2 - x1), x3 = lambda^2 - x1 - x2, y3 = lambda (x1 - x3) - y1.  With Montgomery's trick over a run
// of B additions the division becomes 3 products (prefix product, two on the way back) and a share of ONE inversion:
//     5 products + 1 squaring per addition  (XYZZ mixed add: 7 products + 2 squarings + one double product)
// plus fx_inv / B.  Each thread here owns a run of B independent additions:
//   pass 1  d_j = x2_j - x1_j,  pref_j = pref_(j-1) d_j  -> scratch in HBM (56 B per addition: a thread cannot hold B of them)
//   inv     = 1 / pref_B        (fx_inv: Fermat, 4-bit windows; INV = 0 skips it -- the floor a free inversion would give)
//   pass 2  backwards: 1/d_j = inv pref_(j-1);  inv *= d_j;  the addition itself; result stored (112 B)
// Reported: additions per ms for several B with and without the inversion, the XYZZ mixed add on the same points, and fx_inv alone.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/affine_bench.hip -o tools/affine_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../mpc-jellyfish_amd/csrc/msm.cuh"
using namespace mzk;

template <class X>
__device__ __forceinline__ void load2(const uint32_t* __restrict__ pts, uint32_t idx, Fx<X>& x, Fx<X>& y) {
    const AffineX<X> a = EcFx<X>::load_aff(pts, idx);
    x = a.x; y = a.y;
}
template <class X>
__device__ __forceinline__ void put(uint32_t* __restrict__ dst, const Fx<X>& v) {
#pragma unroll
    for (int i = 0; i < X::XN; i++) dst[i] = v.l[i];
}
template <class X>
__device__ __forceinline__ Fx<X> get(const uint32_t* __restrict__ src) {
    Fx<X> v;
#pragma unroll
    for (int i = 0; i < X::XN; i++) v.l[i] = src[i];
    return v;
}

// run of B additions pts[i1(j)] + pts[i2(j)] per thread
template <class X, bool INV>
__global__ __launch_bounds__(128) void kaffine(const uint32_t* __restrict__ pts, int npts, int B, uint32_t* __restrict__ scratch, uint32_t* __restrict__ out) {
    constexpr int N = X::XN;
    const size_t t = (size_t)blockIdx.x * 128 + threadIdx.x;
    uint32_t* pref = scratch + t * (size_t)B * N;
    const uint32_t seed = (uint32_t)(t * 7) % npts;
    Fx<X> run = Fx<X>::one();
#pragma unroll 1
    for (int j = 0; j < B; j++) {
        const uint32_t i1 = (seed + 13u * (uint32_t)j) % npts, i2 = (i1 + 1 + (uint32_t)j % 7) % npts;   // i1 != i2: distinct points, x2 != x1
        Fx<X> x1, y1, x2, y2;
        load2<X>(pts, i1, x1, y1);
        load2<X>(pts, i2, x2, y2);
        const Fx<X> d = fx_norm(fx_sub2(x2, x1));
        put<X>(pref + (size_t)j * N, run);                                 // pref_(j-1)
        run = fx_mul(run, d);
    }
    Fx<X> inv = INV ? fx_inv(run) : run;                                   // INV = 0: stand-in (results are then not points; timing only)
#pragma unroll 1
    for (int j = B - 1; j >= 0; j--) {
        const uint32_t i1 = (seed + 13u * (uint32_t)j) % npts, i2 = (i1 + 1 + (uint32_t)j % 7) % npts;
        Fx<X> x1, y1, x2, y2;
        load2<X>(pts, i1, x1, y1);
        load2<X>(pts, i2, x2, y2);
        const Fx<X> d = fx_norm(fx_sub2(x2, x1));
        const Fx<X> dinv = fx_mul(inv, get<X>(pref + (size_t)j * N));      // 1 / d_j
        inv = fx_mul(inv, d);
        const Fx<X> lam = fx_mul(fx_norm(fx_sub2(y2, y1)), dinv);
        const Fx<X> x3 = fx_norm(fx_sub8(fx_sqr(lam), fx_add(x1, x2)));      // < 10p
        const Fx<X> y3 = fx_norm(fx_sub2(fx_mul(lam, fx_norm(fx_sub_pad<X>(x1, x3, X::XSUB64))), y1));
        uint32_t* o = out + (t * (size_t)B + j) * 2 * N;
        put<X>(o, x3);
        put<X>(o + N, y3);
    }
}

// y^2 == x^3 + 4 for the stored sums (the INV = 1 run): count of points off the curve
template <class X>
__global__ void kcheck(const uint32_t* __restrict__ out, size_t n, unsigned long long* __restrict__ bad) {
    constexpr int N = X::XN;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fx<X> x = get<X>(out + i * 2 * N), y = get<X>(out + i * 2 * N + N);
    const Fx<X> one = Fx<X>::one();
    const Fx<X> four = fx_norm(fx_add(fx_add(one, one), fx_add(one, one)));
    const Fx<X> lhs = fx_sqr(y), rhs = fx_norm(fx_add(fx_mul(fx_sqr(x), x), four));
    const Fx<X> diff = fx_canonical(fx_mul(fx_norm(fx_sub8(lhs, rhs)), one));
    uint32_t z = 0;
    for (int k = 0; k < N; k++) z |= diff.l[k];
    if (z) atomicAdd(bad, 1ull);
}

template <class X>
__global__ __launch_bounds__(128) void kinv(const uint32_t* __restrict__ pts, int npts, uint32_t* __restrict__ out) {
    const size_t t = (size_t)blockIdx.x * 128 + threadIdx.x;
    const AffineX<X> a = EcFx<X>::load_aff(pts, (uint32_t)(t % npts));
    put<X>(out + t * X::XN, fx_inv(a.x));
}

template <class EC, int ADDS>
__global__ __launch_bounds__(128) void kmadd(const uint32_t* __restrict__ pts, uint32_t* __restrict__ out, int npts) {
    const size_t t = (size_t)blockIdx.x * 128 + threadIdx.x;
    typename EC::Pt acc = EC::inf();
    uint32_t idx = (uint32_t)(t * 7) % npts;
    typename EC::Aff p = EC::load_aff(pts, idx);
#pragma unroll 1
    for (int k = 0; k < ADDS; k++) {
        idx = (idx * 5 + 1) % npts;
        typename EC::Aff pn = EC::load_aff(pts, idx);
        acc = EC::madd(acc, p, (k & 3) == 3);
        p = pn;
    }
    EC::store_pt(out, t, acc);
}

int main() {
    using X = BlsFqX;
    using EC = EcFx<X>;
    constexpr int N = X::XN;
    const int npts = 4096;
    uint32_t *d_tab_xyzz, *d_tab, *d_scal, *d_xy, *d_int, *d_out, *d_scratch;
    unsigned long long* d_bad;
    (void)hipMalloc(&d_tab_xyzz, 256 * 4 * 12 * 4); (void)hipMalloc(&d_tab, 256 * 2 * 12 * 4);
    (void)hipMalloc(&d_scal, npts * 32); (void)hipMalloc(&d_xy, npts * 96); (void)hipMalloc(&d_int, npts * EC::AFF_WORDS * 4);
    const size_t total_adds = (size_t)1 << 24;                                  // the same number of additions in every configuration
    (void)hipMalloc(&d_out, total_adds * 2 * N * 4);
    (void)hipMalloc(&d_scratch, total_adds * N * 4);
    (void)hipMalloc(&d_bad, 8);
    std::vector<uint32_t> sc(npts * 8, 0);
    for (int i = 0; i < npts; i++) { sc[i * 8] = 0x9E3779B9u * (i + 1); sc[i * 8 + 1] = i + 1; }
    (void)hipMemcpy(d_scal, sc.data(), sc.size() * 4, hipMemcpyHostToDevice);
    g1_pow2_table_kernel<BlsFq><<<1, 64>>>(d_tab_xyzz);
    g1_table_to_affine_kernel<BlsFq><<<4, 64>>>(d_tab_xyzz, d_tab, 256);
    g1_fixed_base_kernel<BlsFq><<<npts / 128, 128>>>(d_tab, d_scal, npts, d_xy);
    srs_to_internal_kernel<X><<<npts / 256, 256>>>(d_xy, npts, d_int);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms;
    auto timeit = [&](auto launch) {
        launch();
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        return ms;
    };
    {
        const int threads = 1 << 19, ADDS = 32;
        const float t = timeit([&] { kmadd<EC, ADDS><<<threads / 128, 128>>>(d_int, d_out, npts); });
        printf("XYZZ mixed add (the library's): %d threads x %d: %.3f ms  -> %.2f M adds/ms\n", threads, ADDS, t, (double)threads * ADDS / t * 1e-6);
    }
    for (int B : {64, 256, 1024}) {
        const int threads = (int)(total_adds / B);
        const float t0 = timeit([&] { kaffine<X, false><<<threads / 128, 128>>>(d_int, npts, B, d_scratch, d_out); });
        const float t1 = timeit([&] { kaffine<X, true><<<threads / 128, 128>>>(d_int, npts, B, d_scratch, d_out); });
        (void)hipMemset(d_bad, 0, 8);
        kcheck<X><<<(unsigned)((total_adds + 255) / 256), 256>>>(d_out, total_adds, d_bad);
        unsigned long long bad = 0;
        (void)hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost);
        printf("batched affine, runs of %4d per thread (%7d threads = %5d waves): no inversion %.3f ms -> %.2f M adds/ms;  with fx_inv per run %.3f ms -> %.2f M adds/ms;"
               "  sums off the curve: %llu of %zu\n", B, threads, threads / 64, t0, total_adds / t0 * 1e-6, t1, total_adds / t1 * 1e-6, bad, total_adds);
    }
    for (int threads : {1 << 16, 1 << 18}) {
        const float t = timeit([&] { kinv<X><<<threads / 128, 128>>>(d_int, npts, d_out); });
        printf("fx_inv alone: %d threads (%d waves) %.3f ms -> %.1f us per wave-inversion at this occupancy\n", threads, threads / 64, t, t * 1e3 / ((threads / 64 + 1023) / 1024));
    }
    return 0;
}
