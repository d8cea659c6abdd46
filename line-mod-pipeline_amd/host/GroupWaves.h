// GroupWaves.h -- many dependent walks evaluated in growing waves on the WorkerPool (r05).
//
// The reference's post-processing walks a group's matches IN ORDER -- check, maybe accept, stop once numberWantedPoses poses are
// found (HighLevelLinemod.cpp:165-174, 382-421) -- and every check is a pure function of the frame.  The scheduler below keeps the walk
// sequential and moves the checks ahead of it: a group's items are EVALUATED in waves of 1, 2, 4, .. max_wave on the pool (any
// thread, any order inside a wave), and when the last item of a wave has been evaluated that thread CONSUMES the wave -- the caller's
// `consume(group, from, to)` does the reference's walk over the now known results and says whether the group is finished; if not,
// the next wave starts.  At most the surplus of the wave in which the walk stops is evaluated in vain (nothing when the first item
// ends the walk).
//
// Tokens: start(with_token = true) lets the first wave's evaluations run while something the WALK needs is still missing (the colour
// counts from the GPU): no group consumes before release_tokens() has been called.
//
// Threading contract (verified under ThreadSanitizer, tests/cpp/group_waves_tsan.cpp): evaluate(g, k, early) may run on any thread,
// concurrently for different (g, k); consume(g, from, to) runs on exactly one thread at a time per group, after every evaluate of
// [from, to) has returned and (with tokens) after release_tokens(); start / release_tokens are called by the owner thread; the
// owner then waits for the WorkerPool::Group it handed in.
#pragma once
#include <algorithm>
#include <atomic>
#include <functional>
#include <memory>
#include <vector>

#include "WorkerPool.h"

namespace lmamd {

class GroupWaves {
public:
    // evaluate(group, item, early): early = the wave was started before the tokens were released (the walk's inputs may be missing)
    using Evaluate = std::function<void(size_t, size_t, bool)>;
    // consume(group, from, to) -> true when the group's walk is finished
    using Consume = std::function<bool(size_t, size_t, size_t)>;

    GroupWaves(WorkerPool& pool, WorkerPool::Group& tasks, std::vector<size_t> lengths, Evaluate evaluate, Consume consume, size_t max_wave = 16)
        : pool_(pool), tasks_(tasks), evaluate_(std::move(evaluate)), consume_(std::move(consume)), max_wave_(std::max<size_t>(max_wave, 1)),
          runs_(new Run[lengths.size() ? lengths.size() : 1]), n_runs_(lengths.size()) {
        for (size_t g = 0; g < n_runs_; ++g) runs_[g].n = lengths[g];
    }
    GroupWaves(const GroupWaves&) = delete;
    GroupWaves& operator=(const GroupWaves&) = delete;

    // first wave of every group that has not started yet; with_token: the groups wait for release_tokens() before they consume
    void start(bool with_token) {
        std::vector<std::function<void()>> batch;           // every group's first wave in ONE submission
        for (size_t g = 0; g < n_runs_; ++g)
            if (runs_[g].n && runs_[g].wave == 0) { runs_[g].token = with_token; start_wave(g, with_token, with_token, &batch); }
        pool_.submit_many(tasks_, std::move(batch));
    }
    // the walk's inputs are complete: every group that holds a token may consume (the caller's writes before this call are visible
    // to the consumers: the token is taken off with a read-modify-write on the group's counter)
    // (r05: a group whose first wave is already evaluated is advanced by a pool task, not by the caller -- with 155 groups the caller walked
    // them one after the other for 1.2 ms while the pool slept)
    void release_tokens() {
        std::vector<std::function<void()>> batch;
        for (size_t g = 0; g < n_runs_; ++g)
            if (runs_[g].n && runs_[g].token) { runs_[g].token = false; if (runs_[g].left.fetch_sub(1) == 1) batch.push_back([this, g] { advance(g); }); }
        pool_.submit_many(tasks_, std::move(batch));
    }
    size_t groups() const { return n_runs_; }

private:
    struct Run {
        size_t n = 0, next = 0, from = 0, to = 0, wave = 0;
        bool token = false;
        std::atomic<int> left{0};
    };
    void start_wave(size_t g, bool token, bool early, std::vector<std::function<void()>>* batch = nullptr) {
        Run& r = runs_[g];
        r.wave = r.wave == 0 ? 1 : std::min(2 * r.wave, max_wave_);
        r.from = r.next; r.to = std::min(r.n, r.from + r.wave); r.next = r.to;
        r.left.store((int)(r.to - r.from) + (token ? 1 : 0));
        // (the bounds as locals: once the wave's last task is queued another thread may finish the wave and start the next one, moving r.from / r.to)
        const size_t from = r.from, to = r.to;
        std::vector<std::function<void()>> own;
        std::vector<std::function<void()>>& dst = batch ? *batch : own;
        for (size_t k = from; k < to; ++k)
            dst.push_back([this, g, k, early] {
                evaluate_(g, k, early);
                if (runs_[g].left.fetch_sub(1) == 1) advance(g);
            });
        if (!batch) pool_.submit_many(tasks_, std::move(own));       // a wave's tasks under one lock
    }
    void advance(size_t g) {
        Run& r = runs_[g];
        const bool done = consume_(g, r.from, r.to);
        if (!done && r.next < r.n) start_wave(g, false, false);
    }
    WorkerPool& pool_;
    WorkerPool::Group& tasks_;
    Evaluate evaluate_;
    Consume consume_;
    size_t max_wave_;
    std::unique_ptr<Run[]> runs_;
    size_t n_runs_;
};

}  // namespace lmamd
