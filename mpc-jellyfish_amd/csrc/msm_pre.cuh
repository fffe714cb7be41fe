// msm_pre.cuh -- MSM over a precomputed SRS: table[w][i] = 2^(c*w) * P_i kept in HBM per SRS
// (KZG commits always use the same bases: primitives/src/pcs/univariate_kzg/mod.rs:109-111).
// With the window shifts folded into the bases, all windows share ONE set of 2^(c-1) buckets:
//   * the bucket reduction runs over one window instead of W (so c can grow to ~log2 n: fewer adds),
//   * no cross-window doublings remain, so the host tail shrinks to c doublings.
// 13 x 2^20 x 112 B = 1.5 GB for the 2^20-point BLS12-381 SRS -- the kind of trade 288 GB of HBM allows.
// Sorting 13.6 M (entry, bucket) pairs into 2^19 buckets takes two LDS levels: a coarse partition on
// the top bits (one workgroup per chunk of scalars), then one workgroup per coarse bin.
#pragma once
#include "msm.cuh"

namespace mzk {

constexpr int PRE_CHUNK = 4096;              // scalars per coarse-partition workgroup at large sizes; small MSMs take smaller chunks (pre_chunk_of)
// A workgroup of the coarse level owns `chunk` scalars (a multiple of 4: rows of digits are read 16 bytes at a time).  4096 of them give
// 256 workgroups at 2^20 scalars; a 2^15-scalar MSM would run on 8 (measured: 27 + 38 us for the two coarse kernels of a 2^15-pair MSM,
// all latency), so small sizes halve the chunk until about 128 workgroups exist.
inline uint32_t pre_chunk_of(unsigned long long n) {
    uint32_t chunk = PRE_CHUNK;
    while (chunk > 256 && n / chunk < 128) chunk >>= 1;
    return chunk;
}
constexpr int PRE_CTHREADS = 1024;
constexpr int PRE_FINE_LOG = 11;             // buckets per fine workgroup (2^19 buckets -> 256 workgroups)
constexpr uint32_t PRE_EMPTY = 0xFFFFFFFFu;

// Coarse bins.  Buckets below `low` (the range of the short top digit, which doubles or triples their load) are
// binned 2^low_log at a time, the rest 2^PRE_FINE_LOG at a time, so that every fine workgroup gets about the same
// number of entries.  low == 0: uniform bins.
struct PreBins {
    uint32_t low, low_log, low_bins;
    uint32_t hi_log;                            // buckets per bin above `low`: 2^hi_log <= 2^PRE_FINE_LOG (what the kernels' LDS arrays hold); narrower for
                                                // large MSMs (msm.hip: a fine workgroup re-reads its bin once per PRE_STAGE entries, and a bin of more than
                                                // PRE_HUGE entries leaves the regular path altogether -- at 2^22 pairs every bin of 2^11 buckets did)
    __host__ __device__ uint32_t bin_of(uint32_t b) const { return b < low ? b >> low_log : low_bins + ((b - low) >> hi_log); }
    __host__ __device__ uint32_t first_bucket(uint32_t bin) const { return bin < low_bins ? bin << low_log : low + ((bin - low_bins) << hi_log); }
    __host__ __device__ uint32_t size_of(uint32_t bin, uint32_t M) const {
        const uint32_t full = bin < low_bins ? 1u << low_log : 1u << hi_log;
        const uint32_t left = M - first_bucket(bin);
        return full < left ? full : left;
    }
    __host__ __device__ uint32_t count(uint32_t M) const { return low_bins + ((M - low + (1u << hi_log) - 1) >> hi_log); }
};

// A FUSED batch of small table-path MSMs (msm.hip, msm_group_dev): the `count` MSMs are sorted, accumulated and reduced as ONE problem whose
// bucket range is the concatenation of their bucket sets -- MSM q owns the buckets [q * set_stride, (q + 1) * set_stride) -- the way the
// plain path concatenates the bucket sets of its windows.  Only the digit -> (bucket, table row) mapping of the coarse level knows about
// it: blockIdx.y = q, the digits of MSM q start at digits + q * n_win * stride, its scalars number n[q], its first SRS point is base_off[q].
constexpr int PRE_FUSE_MAX = 8;
struct PreMulti {
    uint32_t count, set_stride;                 // count == 0: one MSM (n, base_off from the kernel arguments)
    unsigned long long n[PRE_FUSE_MAX], base_off[PRE_FUSE_MAX];
};

// digits[w*stride + i] = sign<<31 | (magnitude-1), PRE_EMPTY for a zero digit
template <class FR>
__global__ __launch_bounds__(MSM_THREADS) void pre_digits_kernel(const uint32_t* __restrict__ scalars, unsigned long long n, int is_mont,
                                                                  int c, int n_win, uint32_t* __restrict__ digits, unsigned long long stride) {
    const unsigned long long i = (unsigned long long)blockIdx.x * MSM_THREADS + threadIdx.x;
    if (i >= n) return;
    DigitIter it;
    load_scalar<FR>(it, scalars, i, is_mont);
    for (int w = 0; w < n_win; w++) {
        uint32_t mag, ng;
        it.next(w, c, mag, ng);
        digits[(size_t)w * stride + i] = mag ? ((ng << 31) | (mag - 1)) : PRE_EMPTY;
    }
}

// the digits of all MSMs of a FUSED batch in one launch (blockIdx.y = MSM q: its scalars, its length, its rows of `digits`): the five or
// six launches of a commit group were 7 us each on a corner of the chip, one after the other
struct PreDigitsMulti {
    const uint32_t* scalars[PRE_FUSE_MAX];
    unsigned long long n[PRE_FUSE_MAX];
};
template <class FR>
__global__ __launch_bounds__(MSM_THREADS) void pre_digits_multi_kernel(PreDigitsMulti m, int is_mont, int c, int n_win, uint32_t* __restrict__ digits,
                                                                        unsigned long long stride) {
    const uint32_t q = blockIdx.y;
    const unsigned long long i = (unsigned long long)blockIdx.x * MSM_THREADS + threadIdx.x;
    if (i >= m.n[q]) return;
    DigitIter it;
    load_scalar<FR>(it, m.scalars[q], i, is_mont);
    uint32_t* out = digits + (size_t)q * n_win * stride;
    for (int w = 0; w < n_win; w++) {
        uint32_t mag, ng;
        it.next(w, c, mag, ng);
        out[(size_t)w * stride + i] = mag ? ((ng << 31) | (mag - 1)) : PRE_EMPTY;
    }
}

// Coarse partition on the top bucket bits.  A workgroup owns PRE_CHUNK scalars (all their windows).
//   pre_coarse_count:   bin_total[bin] += entries of this chunk in that bin          (bin_total zeroed first)
//   pre_bin_scan:       bin_start = exclusive scan of bin_total (n_bins + 1 entries), bin_cursor = copy
//   pre_coarse_scatter: counts its chunk again in LDS, reserves a range per bin with ONE global atomic
//                       per bin, then writes (entry, low bucket bits) into it.
// entry = w*tab_stride + base_off + i (row of the precomputed table), sign in bit 31.
// bucket_stride: 0 on the table path (all windows share one bucket set); M on the plain path, whose window w owns the buckets
// [w M, (w + 1) M) of one combined bucket range -- the same two-level sort then serves both paths.
__global__ __launch_bounds__(PRE_CTHREADS) void pre_coarse_count_kernel(const uint32_t* __restrict__ digits, unsigned long long n, unsigned long long stride,
                                                                int n_win, int n_bins, PreBins pb, uint32_t bucket_stride, uint32_t chunk,
                                                                PreMulti multi, uint32_t* __restrict__ bin_total) {
    __shared__ uint32_t bins[1024];
    const uint32_t tid = threadIdx.x;
    for (int j = tid; j < n_bins; j += PRE_CTHREADS) bins[j] = 0u;
    __syncthreads();
    const uint32_t q = blockIdx.y, set_base = q * multi.set_stride;
    if (multi.count) { n = multi.n[q]; digits += (size_t)q * n_win * stride; }
    const unsigned long long lo = (unsigned long long)blockIdx.x * chunk, hi = max(lo, min(n, lo + chunk));
    // the chunk's (window, four scalars) items spread over the threads: a small chunk still keeps the whole workgroup busy
    const uint32_t quads = (uint32_t)((hi - lo + 3) / 4), items = quads * (uint32_t)n_win;
    for (uint32_t t = tid; t < items; t += PRE_CTHREADS) {
        const uint32_t w = t / quads;
        const unsigned long long i = lo + 4ull * (t - w * quads);                                 // stride rows are 16-B aligned, lo is too
        const uint4 v = *reinterpret_cast<const uint4*>(digits + (size_t)w * stride + i);
        const uint32_t d4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (i + k < hi && d4[k] != PRE_EMPTY) atomicAdd(&bins[pb.bin_of((d4[k] & 0x7FFFFFFFu) + w * bucket_stride + set_base)], 1u);
    }
    __syncthreads();
    for (int j = tid; j < n_bins; j += PRE_CTHREADS)
        if (bins[j]) atomicAdd(&bin_total[j], bins[j]);
}

// Bins of the fine level are sized for the expected load (PreBins).  Skewed scalars can put most entries into one bin -- small witness
// values leave every digit of a level below 2^12, flags put them into ONE bucket -- and pre_fine_kernel, one workgroup per bin that
// re-reads the bin once per staged run, then takes milliseconds (8 ms for 2^20 8-bit scalars).  A bin with more than PRE_HUGE entries is
// HUGE: pre_fine skips it and pre_huge_{count,scan,scatter} sort it with as many workgroups as it has slices of `slice` records
// (LDS histogram per slice, one global atomic per slice and non-empty bucket).  No huge bin (uniform scalars): three empty launches.
constexpr uint32_t PRE_HUGE = 131072;
constexpr int PRE_HUGE_MAX = 128;             // huge bins per sort (more: the rest stays with pre_fine_kernel -- slow, still correct)
constexpr int PRE_SLICE_CAP = 2816;           // slice descriptors per sort
// layout of the per-sort words at `huge` (after bin_start in the pre_off block): [0] huge bins, [1] slices, [2 ..] bins, [256 ..] slices (bin << 16 | j)
// zero_words (nullable): n_zero words this one-workgroup launch clears on the way -- the over-long / heavy bucket counters of the MSM
// (msm.cuh), which msm_order_place_kernel counts into later in the same sort
__global__ __launch_bounds__(1024) void pre_bin_scan_kernel(const uint32_t* __restrict__ bin_total, int n_bins, uint32_t* __restrict__ bin_start,
                                                            uint32_t* __restrict__ bin_cursor, uint32_t slice, uint32_t* __restrict__ huge,
                                                            uint32_t* __restrict__ zero_words, uint32_t n_zero) {
    __shared__ uint32_t part[1024];
    const int t = threadIdx.x;
    if (zero_words)
        for (uint32_t i = (uint32_t)t; i < n_zero; i += 1024) zero_words[i] = 0u;
    const uint32_t mine = t < n_bins ? bin_total[t] : 0u;
    part[t] = mine;
    if (t < 2) huge[t] = 0u;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        uint32_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    if (t < n_bins) { bin_start[t] = part[t] - mine; bin_cursor[t] = part[t] - mine; }
    if (t == n_bins - 1) bin_start[n_bins] = part[t];                 // total
    if (mine > PRE_HUGE) {
        const uint32_t h = atomicAdd(&huge[0], 1u);
        if (h < (uint32_t)PRE_HUGE_MAX) {
            const uint32_t ns = (mine + slice - 1) / slice;
            const uint32_t sb = atomicAdd(&huge[1], ns);
            huge[2 + h] = (uint32_t)t;
            for (uint32_t j = 0; j < ns && sb + j < (uint32_t)PRE_SLICE_CAP; j++) huge[256 + sb + j] = ((uint32_t)t << 16) | j;
        }
    }
}
// is `bin` one of the (at most PRE_HUGE_MAX) registered huge bins?  (a bin beyond the cap stays with pre_fine_kernel)
__device__ __forceinline__ bool pre_is_huge(const uint32_t* __restrict__ huge, uint32_t bin) {
    const uint32_t nh = min(huge[0], (uint32_t)PRE_HUGE_MAX);
    for (uint32_t h = 0; h < nh; h++)
        if (huge[2 + h] == bin) return true;
    return false;
}
// one workgroup per slice of a huge bin: bucket histogram of the slice -> hist (zeroed by the bin's own pre_fine workgroup, which leaves
// the bin to these kernels)
__global__ __launch_bounds__(1024) void pre_huge_count_kernel(const uint32_t* __restrict__ huge, const uint32_t* __restrict__ bin_start,
                                                              const unsigned long long* __restrict__ coarse, uint32_t M, PreBins pb, uint32_t slice,
                                                              uint32_t* __restrict__ hist) {
    __shared__ uint32_t bins[1 << PRE_FINE_LOG];
    const uint32_t s = blockIdx.x, tid = threadIdx.x;
    if (s >= min(huge[1], (uint32_t)PRE_SLICE_CAP)) return;
    const uint32_t desc = huge[256 + s], bin = desc >> 16, j = desc & 0xFFFFu;
    const uint32_t rsize = pb.size_of(bin, M), bucket0 = pb.first_bucket(bin);
    const uint32_t start = bin_start[bin] + j * slice, end = min(bin_start[bin + 1], start + slice);
    for (uint32_t q = tid; q < rsize; q += 1024) bins[q] = 0u;
    __syncthreads();
    for (uint32_t k = start + tid; k < end; k += 1024) atomicAdd(&bins[(uint32_t)(coarse[k] >> 32)], 1u);
    __syncthreads();
    for (uint32_t q = tid; q < rsize; q += 1024)
        if (bins[q]) atomicAdd(&hist[bucket0 + q], bins[q]);
}
// one workgroup per huge bin: offs = bin start + exclusive scan of hist; cursor = offs (what pre_huge_scatter reserves from)
__global__ __launch_bounds__(1024) void pre_huge_scan_kernel(const uint32_t* __restrict__ huge, const uint32_t* __restrict__ bin_start, uint32_t M, PreBins pb,
                                                             const uint32_t* __restrict__ hist, uint32_t* __restrict__ offs, uint32_t* __restrict__ cursor,
                                                             uint32_t set_size, uint32_t* __restrict__ keycnt /* nullable: [set][1024] order keys */) {
    __shared__ uint32_t part[1024];
    const uint32_t h = blockIdx.x, tid = threadIdx.x;
    if (h >= min(huge[0], (uint32_t)PRE_HUGE_MAX)) return;
    const uint32_t bin = huge[2 + h], rsize = pb.size_of(bin, M), bucket0 = pb.first_bucket(bin), start = bin_start[bin];
    const uint32_t per = (rsize + 1023) / 1024;
    uint32_t sum = 0;
    for (uint32_t j = tid * per; j < min(rsize, (tid + 1) * per); j++) sum += hist[bucket0 + j];
    part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        uint32_t v = tid >= (uint32_t)d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = start + part[tid] - sum;
    __syncthreads();
    part[tid] = 0u;                                                    // now the counts of the order keys (msm.cuh order_key) of this bin's buckets
    __syncthreads();
    for (uint32_t j = tid * per; j < min(rsize, (tid + 1) * per); j++) {
        const uint32_t c = hist[bucket0 + j];
        offs[bucket0 + j] = run;
        cursor[bucket0 + j] = run;
        run += c;
        if (keycnt) atomicAdd(&part[order_key(c)], 1u);
    }
    __syncthreads();
    if (keycnt && part[tid]) atomicAdd(&keycnt[(size_t)(bucket0 / set_size) * 1024 + tid], part[tid]);
}
// one workgroup per slice: counts again, reserves its range of every bucket with one global atomic, writes the entries
__global__ __launch_bounds__(1024) void pre_huge_scatter_kernel(const uint32_t* __restrict__ huge, const uint32_t* __restrict__ bin_start,
                                                                const unsigned long long* __restrict__ coarse, uint32_t M, PreBins pb, uint32_t slice,
                                                                uint32_t* __restrict__ cursor, uint32_t* __restrict__ sorted) {
    __shared__ uint32_t bins[1 << PRE_FINE_LOG];
    const uint32_t s = blockIdx.x, tid = threadIdx.x;
    if (s >= min(huge[1], (uint32_t)PRE_SLICE_CAP)) return;
    const uint32_t desc = huge[256 + s], bin = desc >> 16, j = desc & 0xFFFFu;
    const uint32_t rsize = pb.size_of(bin, M), bucket0 = pb.first_bucket(bin);
    const uint32_t start = bin_start[bin] + j * slice, end = min(bin_start[bin + 1], start + slice);
    for (uint32_t q = tid; q < rsize; q += 1024) bins[q] = 0u;
    __syncthreads();
    for (uint32_t k = start + tid; k < end; k += 1024) atomicAdd(&bins[(uint32_t)(coarse[k] >> 32)], 1u);
    __syncthreads();
    for (uint32_t q = tid; q < rsize; q += 1024) {
        const uint32_t c = bins[q];
        bins[q] = c ? atomicAdd(&cursor[bucket0 + q], c) : 0u;          // this slice's range of bucket q starts here (absolute index)
    }
    __syncthreads();
    for (uint32_t k = start + tid; k < end; k += 1024) {
        const unsigned long long r = coarse[k];
        sorted[atomicAdd(&bins[(uint32_t)(r >> 32)], 1u)] = (uint32_t)r;
    }
}

__global__ __launch_bounds__(PRE_CTHREADS) void pre_coarse_scatter_kernel(const uint32_t* __restrict__ digits, unsigned long long n, unsigned long long stride,
                                                                  int n_win, int n_bins, PreBins pb, uint32_t bucket_stride, uint32_t chunk, unsigned long long tab_stride,
                                                                  unsigned long long base_off, PreMulti multi, uint32_t* __restrict__ bin_cursor,
                                                                  unsigned long long* __restrict__ coarse) {
    __shared__ uint32_t bins[1024];
    const uint32_t tid = threadIdx.x;
    for (int j = tid; j < n_bins; j += PRE_CTHREADS) bins[j] = 0u;
    __syncthreads();
    const uint32_t q = blockIdx.y, set_base = q * multi.set_stride;
    if (multi.count) { n = multi.n[q]; base_off = multi.base_off[q]; digits += (size_t)q * n_win * stride; }
    const unsigned long long lo = (unsigned long long)blockIdx.x * chunk, hi = max(lo, min(n, lo + chunk));
    const uint32_t quads = (uint32_t)((hi - lo + 3) / 4), items = quads * (uint32_t)n_win;       // as in pre_coarse_count_kernel
    for (uint32_t t = tid; t < items; t += PRE_CTHREADS) {
        const uint32_t w = t / quads;
        const unsigned long long i = lo + 4ull * (t - w * quads);
        const uint4 v = *reinterpret_cast<const uint4*>(digits + (size_t)w * stride + i);
        const uint32_t d4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (i + k < hi && d4[k] != PRE_EMPTY) atomicAdd(&bins[pb.bin_of((d4[k] & 0x7FFFFFFFu) + w * bucket_stride + set_base)], 1u);
    }
    __syncthreads();
    for (int j = tid; j < n_bins; j += PRE_CTHREADS) {
        const uint32_t c = bins[j];
        bins[j] = c ? atomicAdd(&bin_cursor[j], c) : 0u;              // this chunk's range in bin j starts here
    }
    __syncthreads();
    for (uint32_t t = tid; t < items; t += PRE_CTHREADS) {
        const uint32_t w = t / quads;
        const unsigned long long i = lo + 4ull * (t - w * quads);
        const uint4 v = *reinterpret_cast<const uint4*>(digits + (size_t)w * stride + i);
        const uint32_t d4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t d = d4[k];
            if (i + k >= hi || d == PRE_EMPTY) continue;
            const uint32_t b = (d & 0x7FFFFFFFu) + w * bucket_stride + set_base;
            const uint32_t bin = pb.bin_of(b);
            const uint32_t pos = atomicAdd(&bins[bin], 1u);
            const uint32_t e = (uint32_t)((unsigned long long)w * tab_stride + base_off + i + k) | (d & 0x80000000u);
            coarse[pos] = ((unsigned long long)(b - pb.first_bucket(bin)) << 32) | e;                   // one 8-byte record: bucket inside the bin, entry
        }
    }
}

// one workgroup per coarse bin: counts its buckets, scans them, and writes the bucket-sorted entries plus the
// per-bucket hist / offs that the accumulation kernels read.  The scatter goes through LDS: a 4-byte write per
// entry at a random place of the bin's 200 KB output range costs a whole HBM sector each (13.6 M of them at 2^20
// pairs); instead the bin is emitted in runs of consecutive buckets that fit PRE_STAGE entries of LDS -- the
// records are re-read once per run (sequential, L2-resident) and every run leaves as one coalesced stream.
constexpr uint32_t PRE_STAGE = 24576;        // entries staged per run (96 KiB)

__global__ __launch_bounds__(1024) void pre_fine_kernel(const uint32_t* __restrict__ bin_start, const unsigned long long* __restrict__ coarse,
                                                        uint32_t M, PreBins pb, uint32_t* __restrict__ hist, uint32_t* __restrict__ offs,
                                                        uint32_t* __restrict__ sorted, const uint32_t* __restrict__ huge,
                                                        uint32_t set_size, uint32_t* __restrict__ keycnt /* nullable: [set][1024] order keys, zeroed */) {
    __shared__ uint32_t bins[1 << PRE_FINE_LOG];          // counts, then scatter cursors (relative to the bin)
    __shared__ uint32_t loc[(1 << PRE_FINE_LOG) + 1];     // exclusive offsets of the buckets inside the bin
    __shared__ uint32_t part[1024];
    __shared__ uint32_t stage[PRE_STAGE];
    __shared__ uint32_t run_hi;
    const uint32_t bin = blockIdx.x, tid = threadIdx.x;
    const uint32_t rsize = pb.size_of(bin, M), bucket0 = pb.first_bucket(bin);
    const uint32_t start = bin_start[bin], end = bin_start[bin + 1];
    if (end - start > PRE_HUGE && pre_is_huge(huge, bin)) {             // (uniform over the workgroup) sorted by pre_huge_* kernels:
        for (uint32_t j = tid; j < rsize; j += 1024) hist[bucket0 + j] = 0u;        // their slices count into hist
        return;
    }
    for (uint32_t j = tid; j < rsize; j += 1024) bins[j] = 0;
    __syncthreads();
    for (uint32_t k = start + tid; k < end; k += 4096) {
        unsigned long long r[4];
#pragma unroll
        for (int q = 0; q < 4; q++) r[q] = k + q * 1024 < end ? coarse[k + q * 1024] : ~0ull;
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (k + q * 1024 < end) atomicAdd(&bins[(uint32_t)(r[q] >> 32)], 1u);
    }
    __syncthreads();
    const uint32_t per = (rsize + 1023) / 1024;
    uint32_t sum = 0;
    for (uint32_t j = tid * per; j < min(rsize, (tid + 1) * per); j++) sum += bins[j];
    part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        uint32_t v = tid >= (uint32_t)d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;                     // offset inside the bin
    __syncthreads();
    part[tid] = 0u;                                     // now the counts of the order keys of this bin's buckets (what msm_order_hist_kernel
    __syncthreads();                                    // computed in a launch of its own: the ranking of the buckets by load, msm.cuh)
    for (uint32_t j = tid * per; j < min(rsize, (tid + 1) * per); j++) {
        const uint32_t c = bins[j];
        hist[bucket0 + j] = c;
        offs[bucket0 + j] = start + run;
        loc[j] = run;
        bins[j] = run;                                  // becomes the scatter cursor
        run += c;
        if (keycnt) atomicAdd(&part[order_key(c)], 1u);
    }
    if (tid == 1023) loc[rsize] = end - start;
    __syncthreads();
    if (keycnt && part[tid]) atomicAdd(&keycnt[(size_t)(bucket0 / set_size) * 1024 + tid], part[tid]);
    uint32_t lo = 0;                                    // uniform across the workgroup
    while (lo < rsize) {
        // the longest run of buckets [lo, hi) with at most PRE_STAGE entries; one over-long bucket forms a run of its own
        if (tid == 0) {
            uint32_t a = lo + 1, b = rsize;             // hi in [lo + 1, rsize]
            const uint32_t base = loc[lo];
            while (a < b) {
                const uint32_t mid = (a + b + 1) >> 1;
                if (loc[mid] - base <= PRE_STAGE) a = mid; else b = mid - 1;
            }
            run_hi = a;
        }
        __syncthreads();
        const uint32_t hi = run_hi, base = loc[lo], cnt = loc[hi] - base;
        const bool staged = cnt <= PRE_STAGE;
        if (cnt) {
            for (uint32_t k = start + tid; k < end; k += 4096) {
                unsigned long long r[4];
#pragma unroll
                for (int q = 0; q < 4; q++) r[q] = k + q * 1024 < end ? coarse[k + q * 1024] : ~0ull;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t bk = (uint32_t)(r[q] >> 32);
                    if (k + q * 1024 < end && bk >= lo && bk < hi) {
                        const uint32_t pos = atomicAdd(&bins[bk], 1u);
                        if (staged) stage[pos - base] = (uint32_t)r[q];
                        else sorted[start + pos] = (uint32_t)r[q];      // a single bucket longer than the stage: its range is contiguous anyway
                    }
                }
            }
            __syncthreads();
            if (staged)
                for (uint32_t k = tid; k < cnt; k += 1024) sorted[start + base + k] = stage[k];
        }
        __syncthreads();
        lo = hi;
    }
}

// ---- table construction: next[i] = 2^c * cur[i], affine, 8 points per thread share one inversion ----
template <class X>
__global__ __launch_bounds__(128) void pre_next_level_kernel(const uint32_t* __restrict__ cur, uint32_t* __restrict__ next, unsigned long long n, int c) {
    constexpr int B = 4;
    using EC = EcFx<X>;
    const unsigned long long t = (unsigned long long)blockIdx.x * 128 + threadIdx.x;
    const unsigned long long start = t * B;
    if (start >= n) return;
    XYZZX<X> pt[B];
    Fx<X> pref[B];
    Fx<X> run = Fx<X>::one();
#pragma unroll
    for (int j = 0; j < B; j++) {
        XYZZX<X> p = XYZZX<X>::inf();                                // slots past the end stay out of the shared inversion
        if (start + j < n) {
            p = XYZZX<X>::from_affine(EC::load_aff(cur, start + j));
            for (int k = 0; k < c; k++) p = xyzzx_dbl(p);
        }
        pt[j] = p;
        pref[j] = run;
        if (!p.is_inf()) run = fx_mul(run, p.zzz);                   // zzz is class M
    }
    Fx<X> inv_run = fx_inv(run);
#pragma unroll
    for (int j = B - 1; j >= 0; j--) {
        if (start + j >= n) continue;
        uint32_t* dst = next + (start + j) * EC::AFF_WORDS;
        if (pt[j].is_inf()) {
            for (int k = 0; k < EC::AFF_WORDS; k++) dst[k] = 0;
            continue;
        }
        const Fx<X> zi = fx_mul(inv_run, pref[j]);                   // 1 / ZZZ_j
        inv_run = fx_mul(inv_run, pt[j].zzz);
        const Fx<X> zzi = fx_sqr(fx_mul(zi, pt[j].zz));              // (ZZ/ZZZ)^2 = 1/ZZ
        const Fx<X> x = fx_canonical(fx_mul(pt[j].x, zzi)), y = fx_canonical(fx_mul(pt[j].y, zi));
#pragma unroll
        for (int k = 0; k < X::XN; k++) { dst[k] = x.l[k]; dst[X::XN + k] = y.l[k]; }
    }
}

}  // namespace mzk
