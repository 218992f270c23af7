// host_pool.hpp -- ranged host work on a persistent pool of threads: the host side of an evaluation (rank_api.cpp: plan, measures) and of
// cmi_set_ratings (schedule construction, tuple stream) is O(tuples) of independent per-range work.  Internal.
#pragma once
#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <exception>
#include <functional>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

#include <unistd.h>

namespace cmi {

// The host side of an evaluation (plan before the device run, measures after it) is O(tuples + queries) of independent per-range
// work: it runs on the host's cores in contiguous index ranges, each range producing its own output that is then concatenated in
// range order -- the result is the serial one, element for element.
inline int host_threads(int64_t work_items) {
    if (const char *e = getenv("CMI_HOST_THREADS")) return std::max(1, std::min(atoi(e), 64)); // tests: force the ranged form on small inputs
    int t = (int)std::thread::hardware_concurrency();
    t = std::max(1, std::min(t, 16)); // measured on the 64-core host of an MI355X box: 16 threads 13 ms, 32 threads 17 ms for the plan
    return (int)std::max<int64_t>(1, std::min<int64_t>(t, work_items / 4096 + 1));
}

// Worker threads that outlive the call: an evaluation runs five ranged phases of about a millisecond each, and creating 16 threads costs
// about as much as one of them.  One job at a time; a caller that finds the pool busy (another fold's evaluation on another host thread)
// or is itself a worker creates its own threads as before.  The pool is never destroyed (workers sleep on the condition variable until
// the process ends).
class HostPool {
  public:
    static HostPool &get() {
        static HostPool *p = new HostPool();
        return *p;
    }
    bool try_run(int nt, const std::function<void(int)> &fn) { // fn(0) runs on the caller
        if (in_job()) return false; // a ranged body that starts ranged work itself: its own threads (the job lock is not recursive)
        std::unique_lock<std::mutex> job(job_mu_, std::try_to_lock);
        if (!job.owns_lock()) return false;
        struct Busy {
            Busy() { in_job() = true; }
            ~Busy() { in_job() = false; }
        } busy;
        int have = 0; // workers that take part in this job
        {
            std::lock_guard<std::mutex> g(mu_);
            if (pid_ != getpid()) { // a forked child has no workers
                workers_ = 0;
                pid_ = getpid();
            }
            while (workers_ < nt - 1) {
                try {
                    std::thread(&HostPool::work, this, workers_ + 1, gen_).detach(); // gen_: the job posted below is the worker's first
                } catch (const std::system_error &) { // the process may not create more threads (a container's pid limit): fewer workers
                    break;
                }
                ++workers_;
            }
            have = std::min(workers_, nt - 1);
            fn_ = &fn;
            want_ = have + 1;
            left_ = have;
            ++gen_;
        }
        cv_.notify_all();
        // A throwing body (std::bad_alloc of a range's vectors) must not unwind this frame while workers still hold `fn`: the caller's
        // own ranges are caught, the workers are ALWAYS waited for, and the first exception of the job -- the caller's or a worker's,
        // recorded in work() -- is rethrown to the caller once nobody references `fn` any more.
        std::exception_ptr mine;
        try {
            fn(0);
            for (int id = have + 1; id < nt; ++id) fn(id); // the ranges no worker exists for
        } catch (...) {
            mine = std::current_exception();
        }
        std::exception_ptr theirs;
        {
            std::unique_lock<std::mutex> g(mu_);
            done_.wait(g, [&] { return left_ == 0; });
            fn_ = nullptr;
            theirs = err_;
            err_ = nullptr;
        }
        if (mine) std::rethrow_exception(mine);
        if (theirs) std::rethrow_exception(theirs);
        return true;
    }

  private:
    static bool &in_job() {
        static thread_local bool b = false;
        return b;
    }
    void work(int id, uint64_t seen) {
        for (;;) {
            const std::function<void(int)> *fn;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return gen_ != seen; });
                seen = gen_;
                if (id >= want_) continue;
                fn = fn_;
            }
            in_job() = true;
            std::exception_ptr e;
            try {
                (*fn)(id);
            } catch (...) { // never std::terminate the host process from a detached worker: hand the exception to the job's caller
                e = std::current_exception();
            }
            in_job() = false;
            std::lock_guard<std::mutex> g(mu_);
            if (e && !err_) err_ = e;
            if (--left_ == 0) done_.notify_one();
        }
    }
    std::mutex job_mu_, mu_;
    std::condition_variable cv_, done_;
    const std::function<void(int)> *fn_ = nullptr;
    std::exception_ptr err_; // first exception a worker's range threw in the current job
    int workers_ = 0, want_ = 0, left_ = 0;
    uint64_t gen_ = 0;
    pid_t pid_ = getpid();
};

template <typename F>
inline void parallel_ranges(int64_t n, int nt, F &&body) { // body(range index, begin, end)
    if (nt <= 1 || n <= 0) {
        body(0, (int64_t)0, n);
        return;
    }
    const int64_t step = (n + nt - 1) / nt;
    const std::function<void(int)> one = [&](int t) {
        const int64_t b = std::min<int64_t>(n, t * step), e = std::min<int64_t>(n, b + step);
        body(t, b, e);
    };
    if (HostPool::get().try_run(nt, one)) return;
    std::vector<std::thread> th;
    std::mutex emu;
    std::exception_ptr first;
    auto guarded = [&](int t) {
        try {
            one(t);
        } catch (...) {
            std::lock_guard<std::mutex> g(emu);
            if (!first) first = std::current_exception();
        }
    };
    int started = 1;
    try {
        for (; started < nt; ++started) th.emplace_back([&guarded, started]() { guarded(started); });
    } catch (const std::system_error &) { // no more threads: the caller takes the rest
    }
    guarded(0);
    for (int t = started; t < nt; ++t) guarded(t);
    for (std::thread &x : th) x.join();
    if (first) std::rethrow_exception(first);
}

} // namespace cmi
