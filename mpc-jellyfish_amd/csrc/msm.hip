// msm.hip -- host side of the MSM: pipeline launches, the host Horner tail, testing SRS.
#include <algorithm>
#include <chrono>
#include <thread>
#include <vector>

#include <cmath>
#include <cstdlib>

#include "internal.hpp"
#include "hostfp.hpp"
#include "host_tail.hpp"
#include "msm.cuh"
#include "msm_pre.cuh"

namespace mzk {
std::atomic<bool> g_msm_precompute{true};
namespace {

// ---- MSM (the host Horner tail over the device's bit-sums: host_tail.hpp) ---------------------------------------
struct MsmItem {
    const uint32_t* d_bases;     // plain path: first base of this MSM; pre path: the precomputed table
    const uint32_t* d_scalars;
    uint64_t n;
    uint32_t* out_xyz;           // host, Jacobian
    uint64_t base_off;           // pre path: index of this MSM's first SRS point
};
constexpr unsigned long long MSM_MIN_CAP = 48;
constexpr int32_t MSM_RETRY = 1;                 // internal: msm_group_dev wants to run again (never leaves this file)
struct PreInfo { int c = 0; uint64_t tab_stride = 0; };      // c == 0: plain path

// A batch of MSMs sorts on a second stream: the sort of MSM p + 1 (memory- and LDS-bound, few registers) runs under the
// accumulation of MSM p (VALU-bound at two waves per SIMD, which leaves register file and LDS for it).  SORT_SETS sets of sort buffers (~220 MB each at 2^20).
constexpr int SORT_SETS = 6;                                       // sorts run up to five MSMs ahead of the accumulation (3 sets left the sort of a dense MSM that follows two sparse ones exposed: round 1 of the bench circuit)
struct SortStreams {                                               // one set per device context
    hipStream_t stream = nullptr;                                  // (non-null once the set is initialised)
    hipStream_t streams[4] = {};                                   // sorts go round these: the ~14 short launches of a sort are a latency chain, chains side by side cost one
    hipStream_t prover[2] = {};                                    // the streams the context's mzk_prover handles run on (created in the SAME burst: below)
    hipEvent_t ev_start = nullptr, ev_sorted[SORT_SETS] = {}, ev_acc[SORT_SETS] = {};
};
SortStreams g_sort[MAX_CTX];
inline int n_sort_streams() {                                     // (A/B switch MZK_MSM_SORT_STREAMS: 1 .. 4, default 2)
    static const int n = std::min(4, std::max(1, std::getenv("MZK_MSM_SORT_STREAMS") ? std::atoi(std::getenv("MZK_MSM_SORT_STREAMS")) : 2));
    return n;
}
int32_t sort_stream_init(SortStreams& ss) {
    if (ss.stream) return MZK_OK;
    int prio_least = 0, prio_greatest = 0;                               // the short sort kernels go first whenever a slot frees up
    HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    // only as many streams as the sorts alternate between (two): HIP multiplexes a process's streams onto a handful of hardware queues
    // (four by default) in creation order, and a stream that lands on the queue of another is serialised with it -- four sort streams
    // created BEFORE a prover's own stream put that stream on the queue of sort stream 0: its accumulations then waited for the sorts
    // they were meant to overlap (+5 ms per 2^20-gate proof, round 5: profiles/r05_stream_queue_aliasing.txt)
    // several contexts on ONE card (MZK_VIRTUAL_DEVICES: proofs in flight) cannot all have queues of their own: context j starts its burst
    // j * MZK_STREAM_ROT queues further on (placeholder streams), which decides WHICH of its streams meet which of its neighbour's
    static const int rot = std::getenv("MZK_STREAM_ROT") ? std::atoi(std::getenv("MZK_STREAM_ROT")) : 1;      // (measured: profiles/r05_proofs_in_flight.txt)
    for (int q = 0; q < (cur().logical * rot) % 4; q++) {
        hipStream_t pad = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&pad, hipStreamNonBlocking));       // (kept: destroying it would give its place in the rotation back)
    }
    for (int q = 0; q < n_sort_streams(); q++)
        HIP_TRY(hipStreamCreateWithPriority(&ss.streams[q], hipStreamNonBlocking, std::getenv("MZK_MSM_SORT_PRIO_LOW") ? prio_least : prio_greatest));
    // ... and the prover streams right behind them: the runtime hands out its hardware queues round-robin in creation order, so streams
    // created in one burst sit on DIFFERENT queues wherever the burst starts -- a handle's stream created later, on its own, shares a
    // sort stream's queue whenever the number of streams the process made in between happens to be 2 mod 4 (measured with torch side
    // streams and the host-pointer I/O slots: tools/stream_alias_probe.py extra_streams:k -- 42.5 / 47.6 / 43.0 ms for k = 1 / 2 / 4)
    for (auto& q : ss.prover) HIP_TRY(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    ss.stream = ss.streams[0];
    HIP_TRY(hipEventCreateWithFlags(&ss.ev_start, hipEventDisableTiming));
    for (int i = 0; i < SORT_SETS; i++) {
        HIP_TRY(hipEventCreateWithFlags(&ss.ev_sorted[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ss.ev_acc[i], hipEventDisableTiming));
    }
    return MZK_OK;
}

template <class FQ>
void write_infinity(uint32_t* out_xyz) {
    using F64 = Fp64<FQ>;
    F64 one = F64::one(), z = F64::zero();
    one.to_words(out_xyz); one.to_words(out_xyz + FQ::N); z.to_words(out_xyz + 2 * FQ::N);
}

// the rare paths of `n_jobs` MSMs (over-long buckets: chunk sums + their combination; heavy buckets: level-1 sums + trees A, B[, C]) -- with
// `diet` the leaf sums of both paths share one launch (msm_rare_leaf_kernel)
template <class EC>
int32_t launch_rare(const HeavyJobs& jobs, unsigned n_jobs, int sets, uint32_t desc_cap, uint32_t run_cap, bool level_c, bool diet, hipStream_t st) {
    const uint32_t long_blocks = (desc_cap + MSM_ACC_THREADS - 1) / MSM_ACC_THREADS, heavy_blocks = std::min<uint32_t>(run_cap, 2048u);
    const dim3 hg(heavy_blocks, sets, n_jobs);
    if (diet) {
        hipLaunchKernelGGL((msm_rare_leaf_kernel<EC>), dim3(long_blocks + heavy_blocks, sets, n_jobs), dim3(MSM_ACC_THREADS), 0, st, jobs, long_blocks);
    } else {
        for (unsigned q = 0; q < n_jobs; q++) {
            const HeavyJob& jb = jobs.j[q];
            hipLaunchKernelGGL((msm_long_chunk_kernel<EC>), dim3((jb.desc_cap + MSM_ACC_THREADS - 1) / MSM_ACC_THREADS, sets), dim3(MSM_ACC_THREADS), 0, st,
                               jb.bases, jb.n, jb.sorted, jb.desc, jb.desc_count, jb.desc_cap, jb.parts);
        }
        hipLaunchKernelGGL((msm_heavy_chunk_kernel<EC>), hg, dim3(MSM_ACC_THREADS), 0, st, jobs);
    }
    hipLaunchKernelGGL((msm_long_combine_kernel<EC>), dim3(std::min<uint32_t>(long_blocks, 1024u), sets, n_jobs), dim3(MSM_ACC_THREADS), 0, st, jobs);
    hipLaunchKernelGGL((msm_heavy_reduce_kernel<EC>), hg, dim3(MSM_ACC_THREADS), 0, st, jobs, 0);
    hipLaunchKernelGGL((msm_heavy_reduce_kernel<EC>), hg, dim3(MSM_ACC_THREADS), 0, st, jobs, 1);
    if (level_c) hipLaunchKernelGGL((msm_heavy_reduce_kernel<EC>), dim3(64, sets, n_jobs), dim3(MSM_ACC_THREADS), 0, st, jobs, 2);
    HIP_TRY(hipGetLastError());
    return MZK_OK;
}

// `count` MSMs that share the window size: sort + accumulate run one after the other (they fill the
// chip on their own), the latency-bound recursive halving runs ONCE over all bucket sets, one copy
// brings every partial sum to the host, and the host Horner tails run there in sequence (threads only when there are many).
// Plain path: n_dig windows, each with its own 2^(c-1) buckets.  Pre path (msm_pre.cuh): the digits of
// all windows index rows of the precomputed table and share one bucket set.
template <class FR, class EC>
int32_t msm_group_dev(const MsmItem* items, int count, int c, int is_mont, const PreInfo& pre, hipStream_t st) {
    using FQ = typename EC::Field;
    const uint32_t M = 1u << (c - 1);
    const int log_m = c - 1;
    const int n_dig = msm_num_windows(is_mont ? FR::BITS : 256, c);      // digits per scalar
    const int n_win = pre.c ? 1 : n_dig;                                 // bucket sets per MSM
    cur().last_c = c; cur().last_w = n_dig; cur().last_m = M;
    uint64_t n_max = 0, n_min = ~0ull;
    for (int p = 0; p < count; p++) { n_max = std::max<uint64_t>(n_max, items[p].n); n_min = std::min<uint64_t>(n_min, items[p].n); }
    // A batch of SMALL table-path MSMs is FUSED (msm_pre.cuh, PreMulti): one sort, one accumulation, one pass over the over-long buckets
    // for all of them, over the concatenation of their bucket sets -- at these sizes every launch is a chain of latencies on a corner of
    // the chip (a 2^15-pair MSM: ~25 launches, 0.26 ms of them before the reduction), and k chains side by side in ONE launch cost about
    // one.  Downstream of the coarse sort level the fused batch IS a plain-path problem with `count` bucket sets.  Needs uniform coarse
    // bins (no short top digit: top_bits >= c - 1).
    static const bool no_fuse = std::getenv("MZK_MSM_NO_FUSE") != nullptr;                       // (A/B switch)
    const int top_bits_all = (is_mont ? FR::BITS : 256) - c * (n_dig - 1);
    const bool fuse = pre.c && count > 1 && count <= PRE_FUSE_MAX && n_max <= (1ull << 17) && top_bits_all >= c - 1 && !no_fuse;
    const int sets = fuse ? count : n_win;                               // bucket sets of one pass of the pipeline below
    const int passes = fuse ? 1 : count;
    const size_t wm = (size_t)sets * M;
    // Large plain-path MSMs (no fixed-base table: what bench.py's headline runs) sort with the table path's two-level LDS sort over
    // the COMBINED bucket range of their windows (window w owns buckets [w M, (w + 1) M)): one coalesced pass over the digits per
    // level instead of msm_sort_kernel's one scan of a window's digits per 2048-bucket range.
    static const bool no_sort2 = std::getenv("MZK_MSM_PLAIN_SORT1") != nullptr;                   // (A/B switch)
    const bool sort2 = !pre.c && !no_sort2 && n_min >= (1ull << 16) && wm >= (1u << 14) && (wm >> PRE_FINE_LOG) <= 1024;
    const uint64_t sorted_max = pre.c ? n_max * n_dig : n_max;           // entries per bucket set
    MZK_TRY(ws_acquire(st));
    // (Tried in round 4 and dropped: batches of SMALL MSMs in "lanes", one stream per MSM with all of its kernels on it -- cross-stream
    // event waits and queue switches cost more than the latency chains they overlap: profiles/r04_small_msm_lanes_experiment.txt.)
    const bool overlap = passes > 1 && std::getenv("MZK_MSM_NO_OVERLAP") == nullptr;
    const size_t nb = overlap ? (size_t)std::min(passes, SORT_SETS) : 1;  // sets of sort buffers
    MZK_TRY(g_ws.hist.reserve(nb * wm * 4));
    MZK_TRY(g_ws.offs.reserve(nb * wm * 4));
    MZK_TRY(g_ws.cursor.reserve(nb * wm * 4));                           // bucket order by load
    const uint32_t desc_cap_worst = (uint32_t)(sorted_max / MSM_MIN_CAP + 1);        // cap >= MSM_MIN_CAP below
    uint32_t desc_cap_max = desc_cap_worst;                                          // (optimistic with a slot per MSM, like h1_cap below)
    // heavy buckets (msm.cuh): per window <= entries / MSM_HEAVY_RUN full level-1 runs plus one partial run per heavy bucket
    const uint32_t run_cap_max = (uint32_t)(2 * (sorted_max / MSM_HEAVY_RUN) + 2);
    // level-1 sums of the heavy buckets: one per 16 entries (+ slack per run) in the WORST case, every entry in a heavy bucket -- 197 MB
    // per 2^20-pair table-path MSM, which almost no MSM needs.  With a slot per MSM (defer_heavy, below) the arrays are sized for a share
    // of that, remembered per device context; the counters the kernels keep say what was really needed, the host reads them with the
    // results, and a group that overflowed runs again with a larger share (msm.cuh, msm_heavy_push).
    auto h1_cap_of = [](uint64_t entries, uint32_t run_cap_, double frac) {
        const uint64_t worst = entries / MSM_HEAVY_PER_THREAD + 2 * (uint64_t)run_cap_ + 2;
        return (uint32_t)std::min<uint64_t>(worst, (uint64_t)((double)worst * frac) + 2 * (uint64_t)run_cap_ + 4096);
    };
    const uint32_t h1_cap_worst = h1_cap_of(sorted_max, run_cap_max, 1.0);
    uint32_t h1_cap_max = h1_cap_worst;
    const size_t heavy_runs_bytes = (size_t)sets * 3 * run_cap_max * sizeof(HeavyRun);
    // In a batch of at most SORT_SETS (and MSM_HEAVY_JOBS) MSMs every MSM keeps its own sorted list until the end, so their heavy
    // kernels are deferred and run as ONE launch per level over all of them (msm.cuh, HeavyJobs): each MSM then needs its own
    // descriptors, counters and partial sums ("slot").
    bool defer_heavy = passes > 1 && (size_t)passes <= nb && passes <= MSM_HEAVY_JOBS;
    const double heavy_frac = defer_heavy ? g_ws.heavy_frac : 1.0;        // (one shared slot: its counters are overwritten MSM after MSM -- worst case there)
    auto desc_cap_of = [](uint64_t entries, uint64_t cap_, double frac) {
        const uint64_t worst = entries / cap_ + 1;
        return (uint32_t)std::min<uint64_t>(worst, (uint64_t)((double)worst * frac) + 1024);
    };
    if (defer_heavy) { h1_cap_max = h1_cap_of(sorted_max, run_cap_max, heavy_frac); desc_cap_max = desc_cap_of(sorted_max, MSM_MIN_CAP, heavy_frac); }
    size_t desc_slot_bytes = (((size_t)sets * desc_cap_max * sizeof(LongDesc) + (size_t)sets * 4 * (1 + MSM_HEAVY_COUNTERS) + 16 + heavy_runs_bytes) + 255) & ~(size_t)255;
    size_t parts_slot_words = (size_t)sets * desc_cap_max * EC::PT_WORDS + (size_t)sets * ((size_t)h1_cap_max + 2 * (size_t)run_cap_max) * EC::PT_WORDS;
    // ... while the slots fit: the scratch is sized for the worst case (every entry in heavy buckets: ~3.5 x the sorted list per slot,
    // 2.3 GB for five 2^20-pair MSMs) and scales with n.  A batch whose slots would need more than what is reserved already AND more than
    // a quarter of the free HBM (batches of 2^24 pairs and up) runs its heavy kernels per MSM out of one slot instead.
    if (defer_heavy && (size_t)passes * parts_slot_words * 4 > g_ws.long_parts.cap) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        if ((size_t)passes * parts_slot_words * 4 > free_b / 4) {
            defer_heavy = false;
            h1_cap_max = h1_cap_worst;
            desc_cap_max = desc_cap_worst;
            desc_slot_bytes = (((size_t)sets * desc_cap_max * sizeof(LongDesc) + (size_t)sets * 4 * (1 + MSM_HEAVY_COUNTERS) + 16 + heavy_runs_bytes) + 255) & ~(size_t)255;
            parts_slot_words = (size_t)sets * desc_cap_max * EC::PT_WORDS + (size_t)sets * ((size_t)h1_cap_max + 2 * (size_t)run_cap_max) * EC::PT_WORDS;
        }
    }
    const double h1_frac = defer_heavy ? heavy_frac : 1.0;
    const size_t slots = defer_heavy ? (size_t)passes : 1;
    MZK_TRY(g_ws.long_desc.reserve(slots * desc_slot_bytes));
    MZK_TRY(g_ws.long_parts.reserve(slots * parts_slot_words * 4));
    const unsigned long long dstride_max = (n_max + 7) & ~7ull;
    const size_t fused_k = fuse ? (size_t)count : 1;                      // MSMs per pass
    const size_t digits_bytes = fused_k * n_dig * dstride_max * ((pre.c || sort2) ? 4 : 2), sorted_words = (fused_k * n_dig * n_max + 3) & ~(size_t)3;
    MZK_TRY(g_ws.digits.reserve(nb * digits_bytes));
    MZK_TRY(g_ws.sorted.reserve(nb * sorted_words * 4));
    // coarse bins of the table path: the low 2^top_bits buckets also receive the short top digit of every scalar, so they
    // are binned finer by the density ratio 1 + M / (2^top_bits (n_dig - 1)) (msm_pre.cuh PreBins)
    PreBins pb{0, PRE_FINE_LOG, 0, PRE_FINE_LOG};
    if (pre.c && n_dig > 1) {
        const int top_bits = (is_mont ? FR::BITS : 256) - c * (n_dig - 1);
        if (top_bits >= PRE_FINE_LOG && top_bits < c - 1) {
            const double rho = 1.0 + (double)M / ((double)(1ull << top_bits) * (n_dig - 1));
            int shrink = (int)std::lround(std::log2(rho));
            if (shrink > 3) shrink = 3;
            if (shrink > 0) { pb.low = 1u << top_bits; pb.low_log = PRE_FINE_LOG - shrink; pb.low_bins = pb.low >> pb.low_log; }
        }
    }
    // Small bucket ranges: the fine level is one workgroup per bin of 2^PRE_FINE_LOG buckets -- 16 workgroups for the 2^15 buckets of a
    // 2^15-pair MSM (29 us of latency per sort).  With uniform bins narrowed to 2^L buckets about 256 workgroups share the same entries.
    if ((pre.c || sort2) && pb.low == 0 && (wm >> PRE_FINE_LOG) < 128) {
        int lg = 0;
        while ((2ull << lg) <= wm) lg++;
        const int L = std::max(5, lg - 8);
        if (L < PRE_FINE_LOG && (wm >> L) >= 2 && (wm >> L) <= 1024 && wm % (1ull << L) == 0) { pb.low = (uint32_t)wm; pb.low_log = (uint32_t)L; pb.low_bins = (uint32_t)(wm >> L); }
    }
    // Large MSMs: a fine workgroup re-reads its bin once per PRE_STAGE entries and a bin of more than PRE_HUGE entries goes to the
    // pre_huge_* kernels (meant for skewed scalars) -- at 2^22 pairs EVERY bin of 2^11 buckets held 200 K entries and the sort took 1.5 ms
    // where four times the 2^20 sort is 1.1.  Bins are halved while they would hold more than ~32 K entries (about one staged run) and fit the 1024 the coarse level can count.
    if (pre.c || sort2) {
        const uint64_t rec_max = n_max * (uint64_t)n_dig * (fuse ? (uint64_t)count : 1);
        static const uint64_t per_bin = std::getenv("MZK_PRE_BIN_RECORDS") ? std::strtoull(std::getenv("MZK_PRE_BIN_RECORDS"), nullptr, 10) : 32768;     // (tuning switch; 65536: the 2^20 sort 0.27 instead of 0.245 ms)
        while (rec_max / std::max<uint32_t>(1u, pb.count((uint32_t)wm)) > per_bin) {
            PreBins t = pb;
            if (t.low) {
                if (t.low_log <= 5) break;
                t.low_log--; t.low_bins = t.low >> t.low_log;
            }
            if (t.low != (uint32_t)wm) {
                if (t.hi_log <= 5) break;
                t.hi_log--;
            }
            if (t.count((uint32_t)wm) > 1024) break;
            pb = t;
        }
    }
    const uint32_t n_bins = (pre.c || sort2) ? std::max<uint32_t>(1u, pb.count((uint32_t)wm)) : 0u;      // (table path, one MSM: wm == M)
    // Round 5, the launch diet of the two-level sort's paths: the order keys are counted by the sort's own workgroups, ONE kernel ranks the
    // buckets AND registers the over-long / heavy ones (msm_order_place_kernel), the leaf sums of both rare paths share a launch, the
    // reduction halves twice per launch and its last launch also collects the results: 32 -> 21 launches per variable-base MSM of 2^20
    // pairs, bit-identical.  MZK_MSM_LEGACY_LAUNCHES=1: the round-4 sequence (A/B).
    static const bool legacy_launches = std::getenv("MZK_MSM_LEGACY_LAUNCHES") != nullptr;
    const bool diet = (pre.c || sort2) && !legacy_launches;
    const size_t cnt_words = 2048 + (size_t)sets * 1024 * (diet ? 2 : 1);  // bin totals, bin cursors, order keys (diet: and their cursors)
    // the over-long buckets are registered inside the sort only when the sort may write this MSM's descriptors while the stream `st` still
    // works on the previous MSM of the batch: every MSM has a slot of its own (defer_heavy), or there is no second stream
    const bool find_in_sort = diet && (!overlap || defer_heavy);
    MZK_TRY(g_ws.pre_cnt.reserve(nb * cnt_words * 4));
    if (pre.c || sort2) {
        MZK_TRY(g_ws.pre_off.reserve(nb * 8192 * 4));                      // bin_start [n_bins + 1 <= 1025], then the huge-bin words (msm_pre.cuh)
        MZK_TRY(g_ws.pre_ce.reserve(nb * sorted_words * 8));
    }
    MZK_TRY(g_ws.buckets.reserve((size_t)passes * wm * EC::PT_WORDS * 4));
    MZK_TRY(g_ws.occ.reserve((size_t)passes * wm));                       // one byte per bucket slot: does it hold a point?
    const int n_out_one = n_win * (log_m + 1);
    const int n_out = n_out_one * count;
    const size_t out_bytes = (size_t)n_out * 4 * FQ::N * 4;
    const size_t counts_words = (size_t)sets * (1 + MSM_HEAVY_COUNTERS), host_bytes = out_bytes + (size_t)passes * counts_words * 4;   // results, then every MSM's counters
    MZK_TRY(g_ws.collect.reserve(out_bytes));
    if (g_ws.h_collect_cap < host_bytes) {
        if (g_ws.h_collect) HIP_TRY(hipHostFree(g_ws.h_collect));
        g_ws.h_collect = nullptr;
        HIP_TRY(hipHostMalloc(&g_ws.h_collect, host_bytes, hipHostMallocDefault));
        g_ws.h_collect_cap = host_bytes;
    }
    std::vector<uint32_t> h1_caps(passes, 0), desc_caps(passes, 0);
    std::vector<const uint32_t*> count_ptrs(passes, nullptr);
    uint32_t* collect = g_ws.collect.as<uint32_t>();
    HeavyJobs jobs;
    std::memset(&jobs, 0, sizeof jobs);
    uint32_t heavy_run_cap_max = 0, long_desc_cap_max = 0;
    bool heavy_level_c = false;
    SortStreams& ss = g_sort[cur().logical];
    if (overlap) {
        MZK_TRY(sort_stream_init(ss));
        HIP_TRY(hipEventRecord(ss.ev_start, st));                        // scalars and workspace are ready on st
        for (int q = 0; q < n_sort_streams(); q++) HIP_TRY(hipStreamWaitEvent(ss.streams[q], ss.ev_start, 0));
    }

    {
        ProfScope total("msm_total", st);
        const unsigned n_ranges = M >> MSM_RANGE_LOG ? M >> MSM_RANGE_LOG : 1u;
        for (int p = 0; p < passes; p++) {
            const size_t b = overlap ? (size_t)p % nb : 0;               // this MSM's set of sort buffers
            const hipStream_t sst = overlap ? ss.streams[p % n_sort_streams()] : st;      // the stream this MSM's sort runs on
            uint32_t* hist = g_ws.hist.as<uint32_t>() + b * wm;
            uint32_t* offs = g_ws.offs.as<uint32_t>() + b * wm;
            uint32_t* order = g_ws.cursor.as<uint32_t>() + b * wm;
            uint32_t* sorted = g_ws.sorted.as<uint32_t>() + b * sorted_words;
            const uint64_t n = fuse ? n_max : items[p].n;                // fused: the longest MSM of the batch sizes grids and caps
            const uint32_t* d_scalars = items[p].d_scalars;
            const uint32_t* d_bases = items[p].d_bases;
            uint32_t* buckets = g_ws.buckets.as<uint32_t>() + (size_t)p * wm * EC::PT_WORDS;
            uint8_t* occ = g_ws.occ.as<uint8_t>() + (size_t)p * wm;
            const uint64_t n_sorted = pre.c ? n * (uint64_t)n_dig : n;
            // per-thread cap on a bucket's run (a chain of dependent mixed adds, ~5 us each when a wave runs alone): the
            // expected peak load plus six standard deviations.  On the table path the short top digit (scalar bits above
            // c * (n_dig - 1)) lands in the low 2^top_bits buckets only, which therefore carry n / 2^top_bits more points
            // than the mean -- folded into the cap while moderate.  What exceeds the cap (skewed scalars; a short top
            // window) goes to the chunked path below.
            unsigned long long peak = n_sorted / M + 1;
            if (pre.c) {
                // the top digit of a scalar < r takes the values 0 .. r >> (c (n_dig - 1)) only -- 12 389 of the 2^14 a 14-bit digit
                // could take on BN254, 29 678 of 2^15 on BLS12-381 -- and that many buckets share the n top digits
                const int top_bits = (is_mont ? FR::BITS : 256) - c * (n_dig - 1);
                unsigned long long top_range = 1ull << (top_bits > 0 ? top_bits : 0);
                if (top_bits > 0 && top_bits < 32) {
                    const int sh = c * (n_dig - 1), wi = sh >> 5, bo = sh & 31;
                    unsigned long long v = FR::MOD[wi] >> bo;
                    if (bo && wi + 1 < FR::N) v |= (unsigned long long)FR::MOD[wi + 1] << (32 - bo);
                    v &= (1ull << top_bits) - 1;
                    if (v + 1 < top_range) top_range = v + 1;
                }
                const unsigned long long extra = top_bits < c - 1 ? n / top_range : 0;
                if (extra <= 4 * peak) peak += extra;      // a very short top digit (few buckets, huge runs) is left to the chunked path
            }
            const uint32_t cap = (uint32_t)std::max<unsigned long long>(MSM_MIN_CAP, peak + 6 * (unsigned long long)std::sqrt((double)peak) + 8);
            const uint32_t desc_cap = std::min(desc_cap_max, desc_cap_of(n_sorted, cap, h1_frac));
            desc_caps[p] = desc_cap;
            const size_t slot = defer_heavy ? (size_t)p : 0;
            LongDesc* desc = reinterpret_cast<LongDesc*>(g_ws.long_desc.as<char>() + slot * desc_slot_bytes);
            uint32_t* parts = g_ws.long_parts.as<uint32_t>() + slot * parts_slot_words;
            uint32_t* desc_count = reinterpret_cast<uint32_t*>(desc + (size_t)sets * desc_cap);        // `sets` words, then the heavy counters
            const uint32_t* heavy_count = desc_count + sets;
            const uint32_t run_cap = (uint32_t)(2 * (n_sorted / MSM_HEAVY_RUN) + 2);
            const uint32_t h1_cap = std::min(h1_cap_max, h1_cap_of(n_sorted, run_cap, h1_frac));
            h1_caps[p] = h1_cap;
            count_ptrs[p] = desc_count;
            HeavyRun* heavy_runs = reinterpret_cast<HeavyRun*>(desc_count + (((size_t)sets * (1 + MSM_HEAVY_COUNTERS) + 3) & ~(size_t)3));
            uint32_t* h1 = parts + (size_t)sets * desc_cap * EC::PT_WORDS;                           // level-1 sums: one per MSM_HEAVY_PER_THREAD entries of a run
            uint32_t* h2 = h1 + (size_t)sets * h1_cap * EC::PT_WORDS;                                 // one per level-1 run
            uint32_t* h3 = h2 + (size_t)sets * run_cap * EC::PT_WORDS;                                // one per level-B run
            const unsigned long long dstride = (n + 7) & ~7ull;
            const unsigned gs = (unsigned)((n + MSM_THREADS - 1) / MSM_THREADS);
            if (overlap && (size_t)p >= nb) HIP_TRY(hipStreamWaitEvent(sst, ss.ev_acc[b], 0));     // MSM p - nb has read this set
            const unsigned long long list_stride = (pre.c || sort2) ? 0ull : n;     // window w's entries start at sorted + w * list_stride (+ offs)
            if (!pre.c && !sort2) {
                ProfScope ps("msm_sort", sst);
                uint16_t* digits = reinterpret_cast<uint16_t*>(g_ws.digits.as<char>() + b * digits_bytes);
                hipLaunchKernelGGL((msm_digits_kernel<FR>), dim3(gs), dim3(MSM_THREADS), 0, sst, d_scalars, n, is_mont, c, sets, digits, dstride);
                hipLaunchKernelGGL((msm_sort_kernel<false>), dim3(n_ranges, sets), dim3(MSM_SORT_THREADS), 0, sst, digits, n, dstride, M, hist, offs, sorted);
                hipLaunchKernelGGL(msm_scan_kernel, dim3(sets), dim3(1024), 0, sst, hist, offs, M);
                hipLaunchKernelGGL((msm_sort_kernel<true>), dim3(n_ranges, sets), dim3(MSM_SORT_THREADS), 0, sst, digits, n, dstride, M, hist, offs, sorted);
            } else {
                ProfScope ps("msm_sort", sst);
                uint32_t* dig32 = reinterpret_cast<uint32_t*>(g_ws.digits.as<char>() + b * digits_bytes);
                uint32_t* cnt = g_ws.pre_cnt.as<uint32_t>() + b * cnt_words;
                uint32_t* coff = g_ws.pre_off.as<uint32_t>() + b * 8192;
                unsigned long long* coarse = g_ws.pre_ce.as<unsigned long long>() + b * sorted_words;
                const uint32_t chunk = pre_chunk_of(n);
                const uint32_t n_chunks = (uint32_t)((n + chunk - 1) / chunk);
                uint32_t* bin_total = cnt;                       // [n_bins]
                uint32_t* bin_cursor = cnt + 1024;               // [n_bins]
                uint32_t* bin_start = coff;                      // [n_bins + 1]
                const uint32_t bstride = pre.c ? 0u : M;                 // plain path: window w sorts into buckets [w M, (w + 1) M)
                PreMulti multi;
                std::memset(&multi, 0, sizeof multi);
                if (fuse) {                                              // the digits of MSM q: n_dig rows of dstride words from dig32 + q * n_dig * dstride
                    multi.count = (uint32_t)count; multi.set_stride = M;
                    PreDigitsMulti dm;
                    std::memset(&dm, 0, sizeof dm);
                    unsigned long long n_max = 0;
                    for (int q = 0; q < count; q++) {
                        multi.n[q] = items[q].n; multi.base_off[q] = items[q].base_off;
                        dm.scalars[q] = items[q].d_scalars; dm.n[q] = items[q].n;
                        n_max = std::max<unsigned long long>(n_max, items[q].n);
                    }
                    static const bool digits_per_msm = std::getenv("MZK_MSM_DIGITS_PER_MSM") != nullptr;      // (A/B switch: the launches of rounds 4-5)
                    if (digits_per_msm) {
                        for (int q = 0; q < count; q++)
                            hipLaunchKernelGGL((pre_digits_kernel<FR>), dim3((unsigned)((items[q].n + MSM_THREADS - 1) / MSM_THREADS)), dim3(MSM_THREADS), 0, sst,
                                               items[q].d_scalars, items[q].n, is_mont, c, n_dig, dig32 + (size_t)q * n_dig * dstride, dstride);
                    } else if (n_max) {
                        hipLaunchKernelGGL((pre_digits_multi_kernel<FR>), dim3((unsigned)((n_max + MSM_THREADS - 1) / MSM_THREADS), (unsigned)count), dim3(MSM_THREADS), 0, sst,
                                           dm, is_mont, c, n_dig, dig32, dstride);
                    }
                } else {
                    hipLaunchKernelGGL((pre_digits_kernel<FR>), dim3(gs), dim3(MSM_THREADS), 0, sst, d_scalars, n, is_mont, c, n_dig, dig32, dstride);
                }
                const dim3 coarse_grid(n_chunks, (unsigned)fused_k);
                HIP_TRY(hipMemsetAsync(cnt, 0, cnt_words * 4, sst));              // bin totals and, further down, the order keys: one fill
                hipLaunchKernelGGL(pre_coarse_count_kernel, coarse_grid, dim3(PRE_CTHREADS), 0, sst, dig32, n, dstride, n_dig, (int)n_bins, pb, bstride, chunk, multi, bin_total);
                // a bin with more than PRE_HUGE entries (skewed scalars) is sorted by the pre_huge_* kernels in slices of `slice` records
                uint32_t* huge = coff + 1088;
                const uint64_t records = n * (uint64_t)n_dig * fused_k;   // what the coarse level holds (both paths)
                const uint32_t slice = (uint32_t)std::max<uint64_t>(16384, (records + 2047) / 2048);
                const uint32_t slice_grid = (uint32_t)std::min<uint64_t>(PRE_SLICE_CAP, records / slice + PRE_HUGE_MAX + 1);
                const uint32_t huge_grid = (uint32_t)std::min<uint64_t>(PRE_HUGE_MAX, records / PRE_HUGE + 1);
                hipLaunchKernelGGL(pre_bin_scan_kernel, dim3(1), dim3(1024), 0, sst, bin_total, (int)n_bins, bin_start, bin_cursor, slice, huge,
                                   find_in_sort ? desc_count : nullptr, (uint32_t)sets * (1 + MSM_HEAVY_COUNTERS));
                hipLaunchKernelGGL(pre_coarse_scatter_kernel, coarse_grid, dim3(PRE_CTHREADS), 0, sst, dig32, n, dstride, n_dig, (int)n_bins, pb, bstride, chunk,
                                   pre.c ? pre.tab_stride : 0ull, pre.c ? items[p].base_off : 0ull, multi, bin_cursor, coarse);
                uint32_t* keycnt_sort = diet ? cnt + 2048 : nullptr;     // [sets][1024] order keys, counted by the sort's own workgroups
                hipLaunchKernelGGL(pre_fine_kernel, dim3(n_bins), dim3(1024), 0, sst, bin_start, coarse, (uint32_t)wm, pb, hist, offs, sorted, huge, M, keycnt_sort);
                if (records > PRE_HUGE) {                                // (no bin of a sort with at most PRE_HUGE records can be huge: three launches less for small MSMs)
                    hipLaunchKernelGGL(pre_huge_count_kernel, dim3(slice_grid), dim3(1024), 0, sst, huge, bin_start, coarse, (uint32_t)wm, pb, slice, hist);
                    hipLaunchKernelGGL(pre_huge_scan_kernel, dim3(huge_grid), dim3(1024), 0, sst, huge, bin_start, (uint32_t)wm, pb, hist, offs, order, M, keycnt_sort);
                    hipLaunchKernelGGL(pre_huge_scatter_kernel, dim3(slice_grid), dim3(1024), 0, sst, huge, bin_start, coarse, (uint32_t)wm, pb, slice, order, sorted);
                }
            }
            {
                // buckets ranked by load within each bucket set
                ProfScope ps("msm_sort", sst);
                uint32_t* keycnt = g_ws.pre_cnt.as<uint32_t>() + b * cnt_words + 2048;       // [sets][1024]
                const unsigned slices = (M + MSM_ORDER_SLICE - 1) / MSM_ORDER_SLICE;
                if (!pre.c && !sort2) HIP_TRY(hipMemsetAsync(keycnt, 0, (size_t)sets * 1024 * 4, sst));    // (two-level sort: zeroed with the bin totals above)
                if (diet) {
                    hipLaunchKernelGGL(msm_order_place_kernel, dim3(slices, sets), dim3(1024), 0, sst, hist, offs, M, keycnt, keycnt + (size_t)sets * 1024, order,
                                       sets, cap, desc_cap, desc, find_in_sort ? desc_count : nullptr, run_cap, h1_cap, heavy_runs);
                } else {
                    hipLaunchKernelGGL(msm_order_hist_kernel, dim3(slices, sets), dim3(1024), 0, sst, hist, M, keycnt);
                    hipLaunchKernelGGL(msm_order_scan_kernel, dim3(sets), dim3(1024), 0, sst, keycnt);
                    hipLaunchKernelGGL(msm_order_scatter_kernel, dim3(slices, sets), dim3(1024), 0, sst, hist, M, keycnt, order);
                }
            }
            if (overlap) {
                HIP_TRY(hipEventRecord(ss.ev_sorted[b], sst));
                HIP_TRY(hipStreamWaitEvent(st, ss.ev_sorted[b], 0));
            }
            {
                // latency-bound sizes: split every bucket over 2^log_split threads (msm.cuh), aiming at chains of ~4 adds
                // -- as long as the grid stays below what fills the chip (2^18 threads); beyond that only while the (S - 1) M extra
                // additions stay under ~5 % of the work (S - 1 <= mean / 32)
                int log_split = 0;
                const unsigned long long mean = n_sorted / M;
                static const int max_split = std::getenv("MZK_MSM_MAX_SPLIT") ? std::atoi(std::getenv("MZK_MSM_MAX_SPLIT")) : 3;   // (tuning switch)
                // Below 2^18 buckets one thread per bucket is less than two rounds of waves (the chip holds 2^17 threads at the kernel's two
                // waves per SIMD): the chains are exposed in full and a partial last round idles up to half the chip -- split until the grid
                // is ~2^20 threads or chains drop to 2 adds (8 threads per bucket only up to 2^15 buckets and while chains keep 4: measured worse beyond).  Re-swept in round 5 over one
                // MSM, pairs and fused batches of five / six (tools/msm_split_sweep.py): the round-4 rule stopped at 2^18 THREADS, which left
                // the fused commits of a proof at one or two threads per bucket -- batch of five, BLS12-381: 2^13 pairs 0.59 -> 0.53 ms,
                // 2^15 0.96 -> 0.85; BN254: 2^13 0.37 -> 0.32, 2^15 0.63 -> 0.48.
                const bool few_rounds = wm < (1ull << 18);
                while (log_split < max_split) {
                    const int nx = log_split + 1;
                    bool stop;
                    if (few_rounds) stop = nx < 3 ? ((wm << nx) > (1ull << 20) || (mean >> nx) < 2) : ((wm << nx) > (1ull << 18) || (mean >> 5) == 0);
                    else stop = ((1ull << nx) - 1) * 32 > mean;               // beyond: only while the (S - 1) M extra additions stay under ~5 % of the work
                    if (stop) break;
                    log_split = nx;
                }
                static const int force_split = std::getenv("MZK_MSM_FORCE_SPLIT") ? std::atoi(std::getenv("MZK_MSM_FORCE_SPLIT")) : -1;      // (tuning switch: tools/msm_split_sweep.py)
                if (force_split >= 0 && force_split <= 3) log_split = force_split;
                // LARGE plain-path MSMs (several bucket sets, >= 2^18 buckets: several rounds of waves) that the rule above halves: whole-bucket
                // threads for all but the last 1/8 of the ranks, four threads per bucket there (msm.cuh, msm_accumulate_split_kernel).  Headline
                // step, same box, alternating: 3.30-3.31 against 3.33-3.37 ms (profiles/r05_d_tail_split.txt).  The table path's ONE bucket set is
                // ranked as a whole and ends on its lightest buckets: no gain there (measured), not used.  MZK_MSM_NO_TAIL_SPLIT=1: the rule alone
                // (A/B); MZK_MSM_TAIL_FRAC_LOG / MZK_MSM_TAIL_SPLIT: tuning switches.
                static const bool no_tail = std::getenv("MZK_MSM_NO_TAIL_SPLIT") != nullptr;
                static const int tail_frac_log = std::getenv("MZK_MSM_TAIL_FRAC_LOG") ? std::atoi(std::getenv("MZK_MSM_TAIL_FRAC_LOG")) : 3;
                static const int tail_split = std::getenv("MZK_MSM_TAIL_SPLIT") ? std::atoi(std::getenv("MZK_MSM_TAIL_SPLIT")) : 2;
                unsigned long long rank0 = 0;
                if (!few_rounds && sets > 1 && !no_tail && force_split < 0 && log_split == 1 && mean >= 8 && tail_frac_log >= 1 && tail_frac_log <= 6 && tail_split >= 1 && tail_split <= 3) {
                    rank0 = (wm - (wm >> tail_frac_log)) & ~(unsigned long long)(MSM_ACC_THREADS - 1);
                    log_split = tail_split;
                }
                if (log_split == 0) {
                    ProfScope ps("msm_accumulate", st);
                    hipLaunchKernelGGL((msm_accumulate_kernel<EC>), dim3((unsigned)((wm + MSM_ACC_THREADS - 1) / MSM_ACC_THREADS)), dim3(MSM_ACC_THREADS), 0, st,
                                       d_bases, list_stride, offs, hist, sorted, order, M, sets, cap, find_in_sort ? nullptr : desc_count, buckets, occ);
                } else {
                    const size_t n_split = (size_t)(wm - rank0), split_threads = n_split << log_split, threads = (size_t)rank0 + split_threads;
                    MZK_TRY(g_ws.split.reserve(split_threads * EC::PT_WORDS * 4));
                    uint32_t* sub = g_ws.split.as<uint32_t>();
                    const dim3 acc_grid((unsigned)((threads + MSM_ACC_THREADS - 1) / MSM_ACC_THREADS)), cmb_grid((unsigned)((split_threads + 2 * MSM_ACC_THREADS - 1) / (2 * MSM_ACC_THREADS)));
                    {
                        ProfScope ps("msm_accumulate", st);               // the dominant launch alone: what rocprofv3 --stats averages
                        if (sets == 1)
                            hipLaunchKernelGGL((msm_accumulate_split_kernel<EC, true>), acc_grid, dim3(MSM_ACC_THREADS), 0, st, d_bases, list_stride, offs, hist, sorted, order, M, sets,
                                               cap, log_split, find_in_sort ? nullptr : desc_count, sub, rank0, buckets, occ);
                        else
                            hipLaunchKernelGGL((msm_accumulate_split_kernel<EC, false>), acc_grid, dim3(MSM_ACC_THREADS), 0, st, d_bases, list_stride, offs, hist, sorted, order, M, sets,
                                               cap, log_split, find_in_sort ? nullptr : desc_count, sub, rank0, buckets, occ);
                    }
                    ProfScope pc("msm_split_combine", st);
                    if (sets == 1)
                        hipLaunchKernelGGL((msm_split_combine_kernel<EC, true>), cmb_grid, dim3(MSM_ACC_THREADS), 0, st, sub, (unsigned long long)n_split, log_split, order, M, rank0, buckets, occ);
                    else
                        hipLaunchKernelGGL((msm_split_combine_kernel<EC, false>), cmb_grid, dim3(MSM_ACC_THREADS), 0, st, sub, (unsigned long long)n_split, log_split, order, M, rank0, buckets, occ);
                }
            }
            {
                // over-long buckets (skewed scalars); no-ops for uniformly random scalars
                ProfScope ps("msm_long", st);
                if (!find_in_sort)
                    hipLaunchKernelGGL(msm_long_find_kernel, dim3((unsigned)((wm + 255) / 256)), dim3(256), 0, st, hist, offs, M, sets, cap, desc_cap, desc, desc_count,
                                       run_cap, h1_cap, heavy_runs);
                // heavy buckets: a workgroup per run of MSM_HEAVY_RUN entries, then workgroup trees (levels A, B; C only when a bucket can hold
                // more than MSM_HEAVY_RUN * MSM_HEAVY_FANIN entries).  Every workgroup exits at once when there is no heavy bucket.
                HeavyJob& jb = jobs.j[defer_heavy ? p : 0];
                jb.bases = d_bases; jb.sorted = sorted; jb.n = list_stride; jb.runs = heavy_runs; jb.count = heavy_count;
                jb.h1 = h1; jb.h2 = h2; jb.h3 = h3; jb.buckets = buckets; jb.occ = occ; jb.run_cap = run_cap; jb.h1_cap = h1_cap; jb.M = M;
                jb.desc = desc; jb.desc_count = desc_count; jb.parts = parts; jb.desc_cap = desc_cap;
                heavy_run_cap_max = std::max(heavy_run_cap_max, run_cap);
                long_desc_cap_max = std::max(long_desc_cap_max, desc_cap);
                heavy_level_c = heavy_level_c || n_sorted > (uint64_t)MSM_HEAVY_RUN * MSM_HEAVY_FANIN;
                if (!defer_heavy) MZK_TRY((launch_rare<EC>(jobs, 1, sets, desc_cap, run_cap, n_sorted > (uint64_t)MSM_HEAVY_RUN * MSM_HEAVY_FANIN, diet, st)));
            }
            if (overlap) HIP_TRY(hipEventRecord(ss.ev_acc[b], st));
        }
        if (defer_heavy) {                                          // the over-long and heavy buckets of all MSMs of the batch, one launch per level
            ProfScope ps("msm_long", st);
            MZK_TRY((launch_rare<EC>(jobs, (unsigned)passes, sets, long_desc_cap_max, heavy_run_cap_max, heavy_level_c, diet, st)));
        }
        {
            ProfScope ps("msm_reduce", st);
            uint32_t* buckets = g_ws.buckets.as<uint32_t>();
            uint8_t* occ = g_ws.occ.as<uint8_t>();
            const int nw_all = sets * passes;                   // every bucket set folds independently (= n_win * count)
            // wide levels: one launch each over all bucket sets; narrow levels (<= 256 adds per set): one launch in all
            int first_tail = 1;
            while (first_tail <= log_m && (size_t)first_tail * (M >> first_tail) > 256) first_tail++;
            // a level with few additions lasts as long as ONE of them: four lanes per addition there (msm.cuh, fold_one_quad)
            static const size_t quad_max = std::getenv("MZK_MSM_QUAD_MAX") ? (size_t)std::atoll(std::getenv("MZK_MSM_QUAD_MAX")) : 65536;      // (A/B switch; 0 = off.  Measured: a level of 49 K additions 32 -> 23 us, of 82 K the same either way)
            for (int lvl = 1; lvl < first_tail; lvl++) {
                const uint32_t h = M >> lvl;
                const size_t threads = (size_t)nw_all * lvl * h;
                // (two levels per launch -- msm.cuh, msm_fold2_kernel -- measured SLOWER, profiles/r05_msm_launch_diet.txt: the second level's
                // operands are the first level's results, a store -> load round trip per addition that two launches overlap across threads;
                // MZK_MSM_FOLD2=1 switches it on for A/B)
                static const bool fold2 = std::getenv("MZK_MSM_FOLD2") != nullptr;
                if (fold2 && diet && lvl + 1 < first_tail && h >= 2) {
                    const size_t t2 = (size_t)nw_all * lvl * (h >> 1);
                    if (threads <= quad_max)
                        hipLaunchKernelGGL((msm_fold2_quad_kernel<EC>), dim3((unsigned)((4 * t2 + MSM_ACC_THREADS - 1) / MSM_ACC_THREADS)), dim3(MSM_ACC_THREADS), 0, st,
                                           buckets, occ, M, h, lvl, nw_all);
                    else
                        hipLaunchKernelGGL((msm_fold2_kernel<EC>), dim3((unsigned)((t2 + MSM_ACC_THREADS - 1) / MSM_ACC_THREADS)), dim3(MSM_ACC_THREADS), 0, st,
                                           buckets, occ, M, h, lvl, nw_all);
                    lvl++;
                    continue;
                }
                if (threads <= quad_max)
                    hipLaunchKernelGGL((msm_fold_quad_kernel<EC>), dim3((unsigned)((4 * threads + MSM_ACC_THREADS - 1) / MSM_ACC_THREADS)), dim3(MSM_ACC_THREADS), 0, st,
                                       buckets, occ, M, h, lvl, nw_all);
                else
                    hipLaunchKernelGGL((msm_fold_kernel<EC>), dim3((unsigned)((threads + MSM_ACC_THREADS - 1) / MSM_ACC_THREADS)), dim3(MSM_ACC_THREADS), 0, st,
                                       buckets, occ, M, h, lvl, nw_all);
            }
            uint32_t* tail_collect = diet ? collect : nullptr;          // the last launch of the halving also writes the results
            if (first_tail <= log_m) {
                if (quad_max) hipLaunchKernelGGL((msm_fold_tail_quad_kernel<EC>), dim3(nw_all), dim3(MSM_TAIL_QUAD_THREADS), 0, st, buckets, occ, M, log_m, first_tail, tail_collect);
                else hipLaunchKernelGGL((msm_fold_tail_kernel<EC>), dim3(nw_all), dim3(256), 0, st, buckets, occ, M, log_m, first_tail, tail_collect);
            }
            if (!tail_collect || first_tail > log_m)
                hipLaunchKernelGGL((msm_collect_kernel<EC>), dim3((n_out + 63) / 64), dim3(64), 0, st, buckets, occ, M, log_m, nw_all, collect);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(g_ws.h_collect, collect, out_bytes, hipMemcpyDeviceToHost, st));
        if (defer_heavy)                                             // what the heavy buckets of every MSM really needed (their slots keep the counters)
            for (int p = 0; p < passes; p++)
                HIP_TRY(hipMemcpyAsync(static_cast<uint8_t*>(g_ws.h_collect) + out_bytes + (size_t)p * counts_words * 4, count_ptrs[p], counts_words * 4,
                                       hipMemcpyDeviceToHost, st));
    }
    MZK_TRY(ws_release(st));
    HIP_TRY(hipStreamSynchronize(st));
    if (defer_heavy) {
        const uint32_t* cw = reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(g_ws.h_collect) + out_bytes);
        double grow = 0;
        for (int p = 0; p < passes; p++)
            for (int w = 0; w < sets; w++) {
                const uint32_t need = cw[(size_t)p * counts_words + sets + (size_t)w * MSM_HEAVY_COUNTERS + 5];
                if (need > h1_caps[p]) grow = std::max(grow, (double)need / (double)h1_cap_worst);
                const uint32_t need_desc = cw[(size_t)p * counts_words + w];                     // chunk descriptors of the over-long buckets
                if (need_desc > desc_caps[p]) grow = std::max(grow, (double)need_desc / (double)desc_cap_worst);
            }
        if (grow > 0) {                                              // some heavy bucket found no room: its MSM's result is incomplete -- larger arrays, same group again
            g_ws.heavy_frac = std::min(1.0, std::max(g_ws.heavy_frac * 2, grow * 1.25));
            return MSM_RETRY;
        }
    }
    const uint32_t* h = reinterpret_cast<const uint32_t*>(g_ws.h_collect);
    const size_t per = (size_t)n_out_one * 4 * FQ::N;
    // a Horner tail is c doublings + c additions on one core, ~17 us on the table path (one bucket set): starting a thread costs more
    // than running it (measured: 5 tails on 5 fresh threads 130-250 us, in sequence 70 us) -- the tails of a group, and the bucket
    // sets of ONE plain-path MSM, run side by side on the sleeping workers of host_tail.hpp instead (round 5)
    std::vector<uint32_t*> outs(count);
    for (int p = 0; p < count; p++) outs[p] = items[p].out_xyz;
    host_horner_batch<FQ>(h, per, count, n_win, c, outs.data());
    return MZK_OK;
}

constexpr uint64_t PRE_MIN_N = 1ull << 10;       // smaller MSMs stay on the plain path
constexpr int MSM_GROUP_MAX = 6;

template <class FR, class EC>
int32_t msm_batch_dev(const MsmItem* items, int count, int is_mont, const PreInfo& pre, const uint32_t* plain_table, size_t aff_words,
                      hipStream_t st) {
    using FQ = typename EC::Field;
    // runs of consecutive non-empty MSMs with one window size share a fused reduction
    std::vector<MsmItem> run;
    int i = 0;
    while (i < count) {
        if (items[i].n == 0) { write_infinity<FQ>(items[i].out_xyz); i++; continue; }
        if (items[i].n >= (1ull << 27)) { set_error("MSM size must be < 2^27"); return MZK_ERR_INVALID_ARG; }
        auto use_pre = [&](const MsmItem& it) { return pre.c != 0 && it.n >= PRE_MIN_N; };
        auto win_of = [&](const MsmItem& it) { return use_pre(it) ? pre.c : msm_choose_window(it.n); };
        const bool p0 = use_pre(items[i]);
        const int c = win_of(items[i]);
        int j = i + 1;
        // (at most 6 MSMs per group -- a round of UltraPlonk commits six wires --: every MSM of a group keeps its bucket set -- 117 MB at 2^19 buckets -- until the shared reduction; groups
        // of 16, which only the 18 verifying-key commitments of set-up ever formed, left 1.9 GB in the grow-only scratch: round 5)
        while (j < count && j - i < MSM_GROUP_MAX && items[j].n != 0 && items[j].n < (1ull << 27) && use_pre(items[j]) == p0 && win_of(items[j]) == c) j++;
        run.assign(items + i, items + j);
        if (!p0)
            for (auto& it : run) it.d_bases = plain_table + it.base_off * aff_words;   // plain path: pointer to the first base
        for (int attempt = 0;; attempt++) {                          // (MSM_RETRY: the heavy-bucket scratch was too small; it has been enlarged)
            const int32_t rc = msm_group_dev<FR, EC>(run.data(), j - i, c, is_mont, p0 ? pre : PreInfo{}, st);
            if (rc == MSM_RETRY && attempt < 8) continue;
            if (rc == MSM_RETRY) { set_error("MSM heavy-bucket scratch: no fit after 8 attempts"); return MZK_ERR_UNSUPPORTED; }
            MZK_TRY(rc);
            break;
        }
        i = j;
    }
    return MZK_OK;
}

// table[w][i] = 2^(c*w) * P_i for every SRS point (msm_pre.cuh)
template <class X>
int32_t srs_build_pre_t(Srs& s, hipStream_t st) {
    using EC = EcFx<X>;
    int lg = 0;
    while ((2ull << lg) <= s.n) lg++;
    // Window size of the table.  Only sizes whose TOP digit still spans many buckets are eligible: with all windows sharing
    // one bucket set, a top digit of t bits piles n / 2^t extra points onto each of the lowest 2^t buckets (bits(r) = 255
    // for BLS12-381, 254 for BN254; digits = ceil((bits + 1) / c)):
    //   BLS12-381: c = 16 -> 16 digits, top 15 bits;  c = 20 -> 13 digits, top 15 bits
    //   BN254:     c = 15 -> 17 digits, top 14 bits;  c = 17 -> 15 digits, top 16 bits;  c = 20 -> 13 digits, top 14 bits
    int c;
    // (re-swept in round 4 with the fused small batches, profiles/r04_msm_window_sweep.txt: BLS12-381 at 2^17 points c = 16 0.80 / 2.34 ms
    // (one MSM / batch of five) against 0.92 / 2.99 at c = 20, equal at 2^18; BN254 at 2^19 c = 17 1.04 / 4.24 against 1.10 / 4.70 at c = 20)
    if (s.curve == MZK_CURVE_BLS12_381) c = lg <= 17 ? 16 : 20;
    else c = lg <= 15 ? 15 : (lg <= 19 ? 17 : 20);
    if (const char* force = std::getenv("MZK_PRE_C")) {            // tuning only (tools/msm_window_sweep.py): force the table's window
        const int f = std::atoi(force);
        if (f >= 8 && f <= 22) c = f;
    }
    const int W = msm_num_windows(256, c);
    const size_t level = (size_t)s.n * EC::AFF_WORDS;
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    size_t budget = free_b / 2;                                                    // not worth half the free HBM
    if (const char* b = std::getenv("MZK_MSM_TABLE_BUDGET")) budget = std::min<size_t>(budget, (size_t)std::strtoull(b, nullptr, 10));
    if ((size_t)W * level * 4 > budget) { s.pre_c = -1; return MZK_OK; }           // this SRS commits on the plain path
    HIP_TRY(hipMalloc((void**)&s.d_pre, (size_t)W * level * 4));
    HIP_TRY(hipStreamSynchronize(st));
    const auto t_build = std::chrono::steady_clock::now();
    HIP_TRY(hipMemcpyAsync(s.d_pre, s.d_int, level * 4, hipMemcpyDeviceToDevice, st));
    const unsigned long long threads = (s.n + 3) / 4;
    for (int w = 1; w < W; w++)
        hipLaunchKernelGGL((pre_next_level_kernel<X>), dim3((unsigned)((threads + 127) / 128)), dim3(128), 0, st,
                           s.d_pre + (size_t)(w - 1) * level, s.d_pre + (size_t)w * level, (unsigned long long)s.n, c);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    s.pre_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build).count();
    s.pre_c = c;
    s.pre_levels = W;
    return MZK_OK;
}
}  // namespace
int32_t srs_build_pre(Srs& s, hipStream_t st) {
    if (s.d_pre || s.pre_c < 0 || !s.d_int) return MZK_OK;
    return s.curve == MZK_CURVE_BLS12_381 ? srs_build_pre_t<BlsFqX>(s, st) : srs_build_pre_t<BnFqX>(s, st);
}

int32_t msm_dispatch(const Srs& s, uint64_t base_offset, const uint32_t* d_scalars, uint64_t n, int is_mont, uint32_t* out, hipStream_t st) {
    const uint32_t* sc[1] = {d_scalars};
    return msm_batch_dispatch(s, 1, sc, &n, &base_offset, is_mont, out, st);
}

int32_t msm_batch_dispatch(const Srs& s, uint32_t n_polys, const uint32_t* const* d_scalars, const uint64_t* lens, const uint64_t* base_offsets,
                           int is_mont, uint32_t* out_xyz, hipStream_t st) {
    const int fw = fq_words(s.curve);
    // both curves run on the reduced-radix internal table (EcFx: 14 x 29-bit limbs for BLS12-381 Fq, 10 x 29 for BN254 Fq)
    const bool bls = s.curve == MZK_CURVE_BLS12_381;
    const size_t aff_words = bls ? (size_t)EcFx<BlsFqX>::AFF_WORDS : (size_t)EcFx<BnFqX>::AFF_WORDS;
    const uint32_t* table = s.d_int;
    if (!table) { set_error("SRS has no internal table"); return MZK_ERR_BAD_HANDLE; }
    std::vector<MsmItem> items(n_polys);
    bool want_pre = false;
    for (uint32_t i = 0; i < n_polys; i++) {
        const uint64_t off = base_offsets ? base_offsets[i] : 0;
        if (off > s.n || lens[i] > s.n - off) {
            set_error("MSM longer than the registered SRS (poly degree larger than allowed)");
            return MZK_ERR_INVALID_ARG;
        }
        want_pre |= lens[i] >= PRE_MIN_N;
        items[i] = MsmItem{nullptr, d_scalars[i], lens[i], out_xyz + (size_t)i * 3 * fw, off};
    }
    PreInfo pre;
    if (want_pre && g_msm_precompute) {
        Srs& ms = const_cast<Srs&>(s);
        MZK_TRY(srs_build_pre(ms, st));               // first large MSM on this SRS pays for the table
        if (ms.d_pre && ms.pre_c > 0) { pre.c = ms.pre_c; pre.tab_stride = ms.n; }
    }
    if (pre.c)
        for (auto& it : items) it.d_bases = s.d_pre;
    if (bls) return msm_batch_dev<BlsFr, EcFx<BlsFqX>>(items.data(), (int)n_polys, is_mont, pre, table, aff_words, st);
    return msm_batch_dev<BnFr, EcFx<BnFqX>>(items.data(), (int)n_polys, is_mont, pre, table, aff_words, st);
}

// builds the internal (reduced-radix) copy of a freshly registered SRS
template <class X>
static int32_t srs_build_internal_t(Srs& s, hipStream_t st) {
    using EC = EcFx<X>;
    HIP_TRY(hipMalloc((void**)&s.d_int, (size_t)(s.n ? s.n : 1) * EC::AFF_WORDS * 4));
    if (s.n) {
        hipLaunchKernelGGL((srs_to_internal_kernel<X>), dim3((unsigned)((s.n + MSM_THREADS - 1) / MSM_THREADS)), dim3(MSM_THREADS), 0, st,
                           s.d_xy, s.n, s.d_int);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));
    }
    return MZK_OK;
}
int32_t srs_build_internal(Srs& s, hipStream_t st) {
    s.d_int = nullptr;
    s.d_pre = nullptr;
    s.pre_c = 0;
    return s.curve == MZK_CURVE_BLS12_381 ? srs_build_internal_t<BlsFqX>(s, st) : srs_build_internal_t<BnFqX>(s, st);
}

namespace {
template <class FR, class FQ>
int32_t srs_generate(const uint32_t* beta_canon, const uint32_t* g_xy_mont, uint64_t n, uint32_t* d_out) {
    hipStream_t st = nullptr;
    MZK_TRY(ws_acquire(st));
    MZK_TRY(g_ws.scalars.reserve((n ? n : 1) * 32));
    MZK_TRY(g_ws.misc.reserve(32 + 256 * 4 * FQ::N * 4 + 256 * 2 * FQ::N * 4 + 2 * FQ::N * 4));
    uint32_t* d_beta = g_ws.misc.as<uint32_t>();
    uint32_t* d_tab_xyzz = d_beta + 8;
    uint32_t* d_tab = d_tab_xyzz + 256 * 4 * FQ::N;
    uint32_t* d_g = d_tab + 256 * 2 * FQ::N;
    HIP_TRY(hipMemcpyAsync(d_beta, beta_canon, 32, hipMemcpyHostToDevice, st));
    if (g_xy_mont) HIP_TRY(hipMemcpyAsync(d_g, g_xy_mont, 2 * FQ::N * 4, hipMemcpyHostToDevice, st));
    const unsigned long long chunks = (n + 63) / 64;
    hipLaunchKernelGGL((fr_powers_kernel<FR>), dim3((unsigned)((chunks + MSM_THREADS - 1) / MSM_THREADS)), dim3(MSM_THREADS), 0, st,
                       d_beta, n, g_ws.scalars.as<uint32_t>());
    hipLaunchKernelGGL((g1_pow2_table_kernel<FQ>), dim3(1), dim3(64), 0, st, d_tab_xyzz, g_xy_mont ? d_g : nullptr);
    hipLaunchKernelGGL((g1_table_to_affine_kernel<FQ>), dim3(4), dim3(64), 0, st, d_tab_xyzz, d_tab, 256);
    hipLaunchKernelGGL((g1_fixed_base_kernel<FQ>), dim3((unsigned)((n + MSM_ACC_THREADS - 1) / MSM_ACC_THREADS)), dim3(MSM_ACC_THREADS), 0, st,
                       d_tab, g_ws.scalars.as<uint32_t>(), n, d_out);
    HIP_TRY(hipGetLastError());
    MZK_TRY(ws_release(st));
    HIP_TRY(hipStreamSynchronize(st));
    return MZK_OK;
}

// the same over the Lagrange basis of H = <w>, |H| = 2^log_n: point i = L_i(beta) g, then n_extra points beta^j (beta^n - 1) g
template <class FR, class FQ>
int32_t srs_lagrange_generate(const uint32_t* beta_canon, const uint32_t* g_xy_mont, int log_n, uint32_t n_extra, uint32_t* d_out) {
    hipStream_t st = nullptr;
    const uint64_t n = 1ull << log_n, total = n + n_extra;
    MZK_TRY(ws_acquire(st));
    MZK_TRY(g_ws.scalars.reserve(total * 32));
    MZK_TRY(g_ws.misc.reserve(32 + 256 * 4 * FQ::N * 4 + 256 * 2 * FQ::N * 4 + 2 * FQ::N * 4));
    uint32_t* d_beta = g_ws.misc.as<uint32_t>();
    uint32_t* d_tab_xyzz = d_beta + 8;
    uint32_t* d_tab = d_tab_xyzz + 256 * 4 * FQ::N;
    uint32_t* d_g = d_tab + 256 * 2 * FQ::N;
    HIP_TRY(hipMemcpyAsync(d_beta, beta_canon, 32, hipMemcpyHostToDevice, st));
    if (g_xy_mont) HIP_TRY(hipMemcpyAsync(d_g, g_xy_mont, 2 * FQ::N * 4, hipMemcpyHostToDevice, st));
    const unsigned long long chunks = (n + 63) / 64;
    hipLaunchKernelGGL((fr_lagrange_kernel<FR>), dim3((unsigned)((chunks + MSM_THREADS - 1) / MSM_THREADS)), dim3(MSM_THREADS), 0, st,
                       d_beta, log_n, n_extra, g_ws.scalars.as<uint32_t>());
    hipLaunchKernelGGL((g1_pow2_table_kernel<FQ>), dim3(1), dim3(64), 0, st, d_tab_xyzz, g_xy_mont ? d_g : nullptr);
    hipLaunchKernelGGL((g1_table_to_affine_kernel<FQ>), dim3(4), dim3(64), 0, st, d_tab_xyzz, d_tab, 256);
    hipLaunchKernelGGL((g1_fixed_base_kernel<FQ>), dim3((unsigned)((total + MSM_ACC_THREADS - 1) / MSM_ACC_THREADS)), dim3(MSM_ACC_THREADS), 0, st,
                       d_tab, g_ws.scalars.as<uint32_t>(), total, d_out);
    HIP_TRY(hipGetLastError());
    MZK_TRY(ws_release(st));
    HIP_TRY(hipStreamSynchronize(st));
    return MZK_OK;
}

// ... and from the points of an SRS alone (no trapdoor): the inverse group-NTT of its first 2^log_n points (msm.cuh), then S_(n+j) - S_j
template <class FR, class FQ, class X>
int32_t srs_lagrange_from_points(const uint32_t* d_xy, int log_n, uint32_t n_extra, uint32_t* d_out) {
    using F = Fp<FR>;
    using EC = EcFx<X>;
    hipStream_t st = nullptr;
    const uint64_t n = 1ull << log_n;
    MZK_TRY(ws_acquire(st));
    // the array (n points) and the window tables of the scalar multiplications (8 points for each of the n / 2 threads of a stage):
    // (5n + 8) XYZZ points -- 1.17 GB at 2^20 on BLS12-381, 18.8 GB at 2^24 -- a one-off set-up peak that is handed back below
    // instead of staying in the grow-only workspace (an MSM's over-long-bucket scratch needs a small fraction of it)
    const size_t parts_before = g_ws.long_parts.cap, parts_need = (n * 5 + 8) * EC::PT_WORDS * 4;
    MZK_TRY(g_ws.long_parts.reserve(parts_need));
    MZK_TRY(g_ws.misc.reserve(64));
    uint32_t* a = g_ws.long_parts.as<uint32_t>();
    uint32_t* tab = a + n * EC::PT_WORDS;
    uint32_t* d_c = g_ws.misc.as<uint32_t>();
    F w = F::from_const(FR::ROOT);
    for (int i = log_n; i < FR::TWO_ADICITY; i++) w = sqr(w);
    const F winv = from_mont(inv(w)), ninv = from_mont(inv(from_u64<FR>(n)));
    HIP_TRY(hipMemcpyAsync(d_c, winv.l, 32, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_c + 8, ninv.l, 32, hipMemcpyHostToDevice, st));
    const unsigned gn = (unsigned)((n + MSM_ACC_THREADS - 1) / MSM_ACC_THREADS), gh = (unsigned)((n / 2 + MSM_ACC_THREADS - 1) / MSM_ACC_THREADS);
    hipLaunchKernelGGL((ecx_ntt_load_kernel<X>), dim3(gn), dim3(MSM_ACC_THREADS), 0, st, d_xy, n, a);
    for (uint64_t h = n / 2; h >= 1; h >>= 1)
        hipLaunchKernelGGL((ecx_ntt_stage_kernel<FR, X>), dim3(gh ? gh : 1), dim3(MSM_ACC_THREADS), 0, st, a, n, h, d_c, tab);
    hipLaunchKernelGGL((ecx_ntt_finish_kernel<X>), dim3(gn), dim3(MSM_ACC_THREADS), 0, st, a, n, log_n, d_c + 8, tab, d_out);
    if (n_extra) hipLaunchKernelGGL((ec_ntt_extra_kernel<FQ>), dim3(1), dim3(64), 0, st, d_xy, n, n_extra, d_out + n * 2 * FQ::N);
    HIP_TRY(hipGetLastError());
    MZK_TRY(ws_release(st));
    HIP_TRY(hipStreamSynchronize(st));
    if (parts_before < parts_need) g_ws.long_parts.release();          // the set-up peak does not stay resident
    return MZK_OK;
}

// n Jacobian points -> affine on the host (`normalize_batch`): ONE inversion for all of them (Montgomery's trick)
template <class FQ>
void jac_to_affine_host(const uint64_t* xyz, uint64_t n, uint64_t* xy) {
    using F = Fp64<FQ>;
    constexpr int L = FQ::N / 2;
    std::vector<F> pref(n + 1);
    pref[0] = F::one();
    for (uint64_t i = 0; i < n; i++) {
        const F Z = F::from_words((const uint32_t*)(xyz + i * 3 * L + 2 * L));
        pref[i + 1] = Z.is_zero() ? pref[i] : pref[i] * Z;
    }
    F acc = h64::inv(pref[n]);                     // (prod Z)^-1 by division steps (hostinv.hpp; the Fermat power of rounds 1-4 took ~25 us in BLS12-381's Fq)
    for (uint64_t i = n; i-- > 0;) {
        const uint64_t* p = xyz + i * 3 * L;
        uint64_t* o = xy + i * 2 * L;
        const F Z = F::from_words((const uint32_t*)(p + 2 * L));
        if (Z.is_zero()) { std::memset(o, 0, 2 * L * 8); continue; }
        const F zi = acc * pref[i];               // 1 / Z_i
        acc = acc * Z;
        const F X = F::from_words((const uint32_t*)p), Y = F::from_words((const uint32_t*)(p + L)), zi2 = zi * zi;
        const F x = X * zi2, y = Y * zi2 * zi;
        x.to_words((uint32_t*)o);
        y.to_words((uint32_t*)(o + L));
    }
}

}  // namespace

namespace {
template <class FQ>
void jac_sum_host(const uint64_t* xyz, uint64_t n, uint64_t* out) {
    using F = Fp64<FQ>;
    constexpr int L = FQ::N / 2;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (uint64_t i = 0; i < n; i++) {
        const uint64_t* p = xyz + i * 3 * L;
        F X = F::from_words((const uint32_t*)p), Y = F::from_words((const uint32_t*)(p + L)), Z = F::from_words((const uint32_t*)(p + 2 * L));
        if (Z.is_zero()) continue;
        XYZZ<F> q;                                   // Jacobian (X,Y,Z) == XYZZ (X, Y, Z^2, Z^3)
        q.x = X; q.y = Y; q.zz = Z * Z; q.zzz = q.zz * Z;
        acc = xyzz_add(acc, q);
    }
    F X, Y, Z;
    xyzz_to_jacobian(acc, X, Y, Z);
    X.to_words((uint32_t*)out); Y.to_words((uint32_t*)(out + L)); Z.to_words((uint32_t*)(out + 2 * L));
}
}  // namespace

void jac_sum_host_dispatch(int curve, const uint64_t* xyz, uint64_t n, uint64_t* out) {
    if (curve == 0) jac_sum_host<BlsFq>(xyz, n, out);
    else jac_sum_host<BnFq>(xyz, n, out);
}

int32_t srs_generate_dispatch(int curve, const uint32_t* beta_canon, const uint32_t* g_xy_mont, uint64_t n, uint32_t* d_out) {
    return curve == 0 ? srs_generate<BlsFr, BlsFq>(beta_canon, g_xy_mont, n, d_out) : srs_generate<BnFr, BnFq>(beta_canon, g_xy_mont, n, d_out);
}
int32_t srs_lagrange_generate_dispatch(int curve, const uint32_t* beta_canon, const uint32_t* g_xy_mont, int log_n, uint32_t n_extra, uint32_t* d_out) {
    return curve == 0 ? srs_lagrange_generate<BlsFr, BlsFq>(beta_canon, g_xy_mont, log_n, n_extra, d_out)
                      : srs_lagrange_generate<BnFr, BnFq>(beta_canon, g_xy_mont, log_n, n_extra, d_out);
}
int32_t srs_lagrange_from_points_dispatch(int curve, const uint32_t* d_xy, int log_n, uint32_t n_extra, uint32_t* d_out) {
    return curve == 0 ? srs_lagrange_from_points<BlsFr, BlsFq, BlsFqX>(d_xy, log_n, n_extra, d_out) : srs_lagrange_from_points<BnFr, BnFq, BnFqX>(d_xy, log_n, n_extra, d_out);
}
void jac_to_affine_host_dispatch(int curve, const uint64_t* xyz, uint64_t n, uint64_t* xy) {
    if (curve == 0) jac_to_affine_host<BlsFq>(xyz, n, xy);
    else jac_to_affine_host<BnFq>(xyz, n, xy);
}

// the stream handle `k` of this context runs its rounds on: one of two, made together with the sort streams (sort_stream_init)
int32_t ctx_prover_stream(unsigned k, hipStream_t* out) {
    SortStreams& ss = g_sort[cur().logical];
    MZK_TRY(sort_stream_init(ss));
    *out = ss.prover[k & 1];
    return MZK_OK;
}

void msm_release_streams() {
    SortStreams& ss = g_sort[cur().logical];
    if (!ss.stream) return;
    for (auto& q : ss.streams) if (q) (void)hipStreamDestroy(q);
    for (auto& q : ss.prover) if (q) (void)hipStreamDestroy(q);
    (void)hipEventDestroy(ss.ev_start);
    for (int i = 0; i < SORT_SETS; i++) { (void)hipEventDestroy(ss.ev_sorted[i]); (void)hipEventDestroy(ss.ev_acc[i]); }
    ss = SortStreams();
}

}  // namespace mzk
