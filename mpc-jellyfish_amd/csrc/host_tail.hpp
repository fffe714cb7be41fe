// host_tail.hpp -- the host end of an MSM: Horner over the bit-sums the device's recursive halving leaves (msm.cuh, msm_fold_*):
// per bucket set w the points T_{w,0} = sum of all buckets (bucket i weighs i + 1) and T_{w,j} = sum of the buckets whose index has
// bit log_m - j set, so the set's sum is T_{w,0} + sum_k 2^k T_{w,log_m-k} and the MSM is sum_w 2^(c w) (set w).
//
// One bucket set (the table path: every commit of a proof) is c doublings + c additions, ~14 us on one core.  The plain path's
// n_win sets (bench.py's headline: 16 windows of 16 bits) are 255 doublings + 256 additions in sequence, ~0.2 ms = 6 % of the step;
// only the doublings depend on each other.  host_horner() therefore sums the sets side by side on a small pool of sleeping worker
// threads (host_tail_pool) while the calling thread starts with the top set and then runs the chain of c doublings + ONE addition
// per set: 256 doublings + 31 additions on the critical path.  Same point, another Jacobian representative than the serial order.
// Product code (host side of libmi355zk); no GPU call in here.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>
#if defined(__linux__)
#include <sched.h>
#endif

#include "ec.cuh"
#include "hostfp.hpp"

namespace mzk {

template <class FQ>
inline XYZZ<Fp64<FQ>> host_tail_load(const uint32_t* pts, int per, int w, int j) {
    using F = Fp64<FQ>;
    XYZZ<F> p;
    const uint32_t* s = pts + ((size_t)w * per + j) * 4 * FQ::N;
    p.x = F::from_words(s); p.y = F::from_words(s + FQ::N);
    p.zz = F::from_words(s + 2 * FQ::N); p.zzz = F::from_words(s + 3 * FQ::N);
    return p;
}

// sum of bucket set w: sum_i (i + 1) B_i from its c = log_m + 1 bit-sums
template <class FQ>
inline XYZZ<Fp64<FQ>> host_tail_set_sum(const uint32_t* pts, int c, int w) {
    using F = Fp64<FQ>;
    const int log_m = c - 1, per = c;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int k = c - 2; k >= 0; k--) {
        if (!acc.is_inf()) acc = xyzz_dbl(acc);
        XYZZ<F> t = host_tail_load<FQ>(pts, per, w, log_m - k);       // buckets whose index has bit k set
        if (!t.is_inf()) acc = xyzz_add(acc, t);
    }
    XYZZ<F> t0 = host_tail_load<FQ>(pts, per, w, 0);                  // every bucket once (weights are index + 1)
    if (!t0.is_inf()) acc = xyzz_add(acc, t0);
    return acc;
}

template <class FQ>
inline void host_tail_store(const XYZZ<Fp64<FQ>>& acc, uint32_t* out_xyz) {
    using F = Fp64<FQ>;
    F X, Y, Z;
    xyzz_to_jacobian(acc, X, Y, Z);
    X.to_words(out_xyz); Y.to_words(out_xyz + FQ::N); Z.to_words(out_xyz + 2 * FQ::N);
}

// everything on the calling thread, top window first (rounds 1-4's only form; what a single bucket set still runs)
template <class FQ>
void host_horner_serial(const uint32_t* pts, int n_win, int c, uint32_t* out_xyz) {
    using F = Fp64<FQ>;
    const int log_m = c - 1, per = c;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int w = n_win - 1; w >= 0; w--) {
        for (int k = c - 1; k >= 0; k--) {
            if (!acc.is_inf()) acc = xyzz_dbl(acc);
            if (k <= c - 2) {
                XYZZ<F> t = host_tail_load<FQ>(pts, per, w, log_m - k);
                if (!t.is_inf()) acc = xyzz_add(acc, t);
            }
        }
        XYZZ<F> t0 = host_tail_load<FQ>(pts, per, w, 0);
        if (!t0.is_inf()) acc = xyzz_add(acc, t0);
    }
    host_tail_store<FQ>(acc, out_xyz);
}

// ---- the worker pool ---------------------------------------------------------------------------------------------------------
// A handful of threads asleep on a condition variable; a job is "call fn(ctx, i) for i = count - 1 .. 0", items claimed from an
// atomic counter by the workers AND by the caller, who never waits for a worker it could stand in for.  One job at a time: a caller
// that finds the pool busy (another device context's MSM ends at the same moment) runs its tail serially instead.  The counter
// carries the job's generation, so a worker that wakes up late for job A can never take an item of job B with A's context; a job's
// function must finish with whatever tells the caller that the item is done (nothing of `ctx` is touched after that).
class HostTailPool {
public:
    using Fn = void (*)(void* ctx, int item);
    explicit HostTailPool(int n_threads) {
        for (int i = 0; i < n_threads; i++) workers_.emplace_back([this] { loop(); });
    }
    ~HostTailPool() {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    int size() const { return (int)workers_.size(); }
    // publishes the job and wakes the workers; returns its generation (> 0), or 0 when the pool is in use
    uint32_t try_begin(Fn fn, void* ctx, int count) {
        if (busy_.exchange(true, std::memory_order_acquire)) return 0;
        uint32_t gen;
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = fn; ctx_ = ctx;
            gen = ++gen_ ? gen_ : ++gen_;                           // (never 0)
            next_.store(((uint64_t)gen << 32) | (uint32_t)count, std::memory_order_release);
        }
        cv_.notify_all();
        return gen;
    }
    int claim(uint32_t gen) {                                       // >= 0: an item of job `gen` to run; < 0: none left (or another job's turn)
        uint64_t v = next_.load(std::memory_order_acquire);
        for (;;) {
            if ((uint32_t)(v >> 32) != gen || (uint32_t)v == 0) return -1;
            if (next_.compare_exchange_weak(v, v - 1, std::memory_order_acq_rel, std::memory_order_acquire)) return (int)(uint32_t)v - 1;
        }
    }
    void end() { busy_.store(false, std::memory_order_release); }   // caller: every item has been seen finished
    static void cpu_relax() {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }

private:
    void loop() {
        uint32_t seen = 0;
        for (;;) {
            Fn fn; void* ctx;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_; fn = fn_; ctx = ctx_;                  // (one job's triple, read under the lock it was written under)
            }
            for (int i; (i = claim(seen)) >= 0;) fn(ctx, i);
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_;
    bool stop_ = false;
    uint32_t gen_ = 0;
    Fn fn_ = nullptr;
    void* ctx_ = nullptr;
    std::atomic<uint64_t> next_{0};
    std::atomic<bool> busy_{false};
};

// MZK_HOST_TAIL_THREADS = worker threads of the pool (0: everything on the caller).  Default min(4, usable hardware threads - 1): the
// caller's chain takes a set every c doublings (4.5 us at c = 16 on an EPYC 9575F) and a set sum lasts 10 us, so two or three
// workers keep ahead of it; waking more only lengthens notify_all (measured on the GPU box, tools/host_tail_bench.cpp: 248 us on the
// caller alone, 121 / 124 / 137 us with 3 / 7 / 15 workers).
inline HostTailPool& host_tail_pool() {
    static HostTailPool pool([] {
        if (const char* e = std::getenv("MZK_HOST_TAIL_THREADS")) return std::max(0, std::min(63, std::atoi(e)));
        int hw = (int)std::thread::hardware_concurrency();
#if defined(__linux__)
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) hw = CPU_COUNT(&set);
#endif
        return std::max(0, std::min(4, hw - 1));
    }());
    return pool;
}
inline int host_tail_pool_size() { return host_tail_pool().size(); }

template <class FQ>
struct HostTailJob {
    using F = Fp64<FQ>;
    const uint32_t* pts;
    int c;
    XYZZ<F>* sums;                      // [n_win]
    std::atomic<int>* ready;            // [n_win]
    static void run(void* ctx, int w) {
        auto* j = static_cast<HostTailJob*>(ctx);
        j->sums[w] = host_tail_set_sum<FQ>(j->pts, j->c, w);
        j->ready[w].store(1, std::memory_order_release);
    }
};

// the MSM's result from the n_win * c bit-sums at `pts` (XYZZ, 4 * FQ::N words each): Jacobian (X, Y, Z) into out_xyz
template <class FQ>
void host_horner(const uint32_t* pts, int n_win, int c, uint32_t* out_xyz) {
    using F = Fp64<FQ>;
    constexpr int MAX_WIN = 64;
    if (n_win < 4 || n_win > MAX_WIN) return host_horner_serial<FQ>(pts, n_win, c, out_xyz);
    HostTailPool& pool = host_tail_pool();
    XYZZ<F> sums[MAX_WIN];
    std::atomic<int> ready[MAX_WIN];
    for (int w = 0; w < n_win; w++) ready[w].store(0, std::memory_order_relaxed);
    HostTailJob<FQ> job{pts, c, sums, ready};
    const uint32_t gen = pool.size() ? pool.try_begin(&HostTailJob<FQ>::run, &job, n_win) : 0;
    if (!gen) return host_horner_serial<FQ>(pts, n_win, c, out_xyz);
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int w = n_win - 1; w >= 0; w--) {
        while (!ready[w].load(std::memory_order_acquire)) {           // not there yet: stand in for a worker rather than wait for one
            const int i = pool.claim(gen);
            if (i >= 0) HostTailJob<FQ>::run(&job, i);
            else HostTailPool::cpu_relax();
        }
        if (!acc.is_inf())
            for (int k = 0; k < c; k++) acc = xyzz_dbl(acc);
        if (!sums[w].is_inf()) acc = xyzz_add(acc, sums[w]);
    }
    pool.end();
    host_tail_store<FQ>(acc, out_xyz);
}

// the tails of `count` MSMs of one group (every MSM: n_win sets of c bit-sums, `per_words` words apart; result p to outs[p]).
// Single-set tails -- the five or six commits of a proof's round on the table path, ~17 us each -- run side by side on the pool, the
// caller taking its share; a many-window MSM on its own takes the overlapped form above; batches of many-window MSMs (plain path,
// rare) run whole tails side by side.
template <class FQ>
struct HostTailBatchJob {
    const uint32_t* pts;
    size_t per_words;
    int n_win, c;
    uint32_t* const* outs;
    std::atomic<int>* ready;            // [count]
    static void run(void* ctx, int p) {
        auto* j = static_cast<HostTailBatchJob*>(ctx);
        host_horner_serial<FQ>(j->pts + (size_t)p * j->per_words, j->n_win, j->c, j->outs[p]);
        j->ready[p].store(1, std::memory_order_release);
    }
};
template <class FQ>
void host_horner_batch(const uint32_t* pts, size_t per_words, int count, int n_win, int c, uint32_t* const* outs) {
    constexpr int MAX_COUNT = 64;
    if (count == 1) return host_horner<FQ>(pts, n_win, c, outs[0]);
    HostTailPool& pool = host_tail_pool();
    std::atomic<int> ready[MAX_COUNT];
    HostTailBatchJob<FQ> job{pts, per_words, n_win, c, outs, ready};
    uint32_t gen = 0;
    if (count <= MAX_COUNT && pool.size()) {
        for (int p = 0; p < count; p++) ready[p].store(0, std::memory_order_relaxed);
        gen = pool.try_begin(&HostTailBatchJob<FQ>::run, &job, count);
    }
    if (!gen) {
        for (int p = 0; p < count; p++) host_horner_serial<FQ>(pts + (size_t)p * per_words, n_win, c, outs[p]);
        return;
    }
    for (int i; (i = pool.claim(gen)) >= 0;) HostTailBatchJob<FQ>::run(&job, i);
    for (int p = 0; p < count; p++)
        while (!ready[p].load(std::memory_order_acquire)) HostTailPool::cpu_relax();
    pool.end();
}

}  // namespace mzk
