// WorkerPool.h -- the persistent host thread pool of the facade (r05; ADVICE r4: detectTemplatesBatch created up to 32 std::threads per
// call and an exception in one of them ended in std::terminate).
//
// What runs on it: the staging copies of a batch's pageable frames (lm_stage_rows: host memory only), the per-frame grouping of the
// match lists and the reference's depth checks + poses of the independent match groups (PostProcess.cpp finish_group) -- nothing that
// touches the detector, which stays with the thread that owns the HighLevelLineMOD.  Tasks belong to a Group; wait(group) lets the
// caller run queued tasks itself until the group is done, so a pool of n threads is n - 1 workers + the caller.  A task that throws
// is caught and the first message of its group is kept (Group::error): nothing reaches std::terminate.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <cstdlib>
#include <exception>
#include <fstream>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace lmamd {

// CPUs this process may really use: hardware threads, cut by the cgroup CPU quota (cgroup v2 cpu.max / v1 cfs_quota) -- more busy
// threads than the quota only get the whole process throttled.
inline int usable_cpus() {
    unsigned hw = std::thread::hardware_concurrency();
    int n = hw ? (int)hw : 1;
    {
        std::ifstream f("/sys/fs/cgroup/cpu.max");
        std::string q; long long period = 0;
        if (f && (f >> q >> period) && q != "max" && period > 0) {
            const long long quota = std::atoll(q.c_str());
            if (quota > 0) n = std::min<int>(n, (int)std::max<long long>(1, (quota + period / 2) / period));
        }
    }
    {
        std::ifstream fq("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), fp("/sys/fs/cgroup/cpu/cpu.cfs_period_us");
        long long quota = -1, period = 0;
        if (fq && fp && (fq >> quota) && (fp >> period) && quota > 0 && period > 0)
            n = std::min<int>(n, (int)std::max<long long>(1, (quota + period / 2) / period));
    }
    return n < 1 ? 1 : n;
}

class WorkerPool {
public:
    struct Group {
        std::atomic<int> pending{0};
        std::mutex mu;
        std::string error;          // first exception message of the group's tasks ("" = none)
        void fail(const std::string& what) { std::lock_guard<std::mutex> g(mu); if (error.empty()) error = what; }
    };

    explicit WorkerPool(int threads) {
        const int workers = threads > 1 ? threads - 1 : 0;
        for (int i = 0; i < workers; ++i) th.emplace_back([this] { loop(); });
    }
    ~WorkerPool() {
        { std::lock_guard<std::mutex> g(mu); stop = true; }
        cv.notify_all();
        for (std::thread& t : th) t.join();
    }
    WorkerPool(const WorkerPool&) = delete;
    WorkerPool& operator=(const WorkerPool&) = delete;

    int threads() const { return (int)th.size() + 1; }

    // front: ahead of everything queued (a batch's staging copies go before the depth checks of the batch before it)
    void submit(Group& g, std::function<void()> fn, bool front = false) {
        g.pending.fetch_add(1, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(mu);
            if (front) q.push_front(Task{&g, std::move(fn)}); else q.push_back(Task{&g, std::move(fn)});
        }
        cv.notify_one();
        done_cv.notify_one();           // a caller sleeping in wait() helps with the new task
    }

    // Many tasks at once: one lock, one wake-up of everybody (r05: a submit is a lock + two futex wake-ups, ~3 us with 15 workers asleep -- the 155
    // first-wave tasks of a batch's match groups took the submitting thread 450 us one by one).
    void submit_many(Group& g, std::vector<std::function<void()>>&& fns, bool front = false) {
        if (fns.empty()) return;
        g.pending.fetch_add((int)fns.size(), std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(mu);
            if (front) for (size_t i = fns.size(); i-- > 0;) q.push_front(Task{&g, std::move(fns[i])});
            else for (auto& fn : fns) q.push_back(Task{&g, std::move(fn)});
        }
        cv.notify_all();
        done_cv.notify_all();
        fns.clear();
    }

    // Returns when every task of `g` has finished; the caller runs queued tasks (of any group) meanwhile and sleeps when there is
    // nothing to run (woken by a submit or by the group's last task).
    void wait(Group& g) {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lk(mu);
                done_cv.wait(lk, [&] { return !q.empty() || g.pending.load(std::memory_order_acquire) == 0; });
                if (q.empty()) return;                  // (the group is done)
                if (g.pending.load(std::memory_order_acquire) == 0) return;
                t = std::move(q.front()); q.pop_front();
            }
            run(t);
        }
    }

private:
    struct Task { Group* g = nullptr; std::function<void()> fn; };
    void run(Task& t) {
        try { t.fn(); }
        catch (const std::exception& e) { t.g->fail(e.what()); }
        catch (...) { t.g->fail("unknown exception in a pool task"); }
        if (t.g->pending.fetch_sub(1, std::memory_order_acq_rel) == 1) {
            // (through the mutex: a waiter is either before its check of `pending` -- and will see 0 -- or asleep -- and is woken)
            { std::lock_guard<std::mutex> lk(mu); }
            done_cv.notify_all();
        }
    }
    void loop() {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [this] { return stop || !q.empty(); });
                if (q.empty()) return;       // stop
                t = std::move(q.front()); q.pop_front();
            }
            run(t);
        }
    }
    std::vector<std::thread> th;
    std::deque<Task> q;
    std::mutex mu;
    std::condition_variable cv, done_cv;
    bool stop = false;
};

}  // namespace lmamd
