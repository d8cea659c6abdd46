// ThreadSanitizer / AddressSanitizer drive of the facade's host concurrency (r05): host/WorkerPool.h and host/GroupWaves.h, without the GPU.
//   * many groups of random length, every item "evaluated" on the pool (a little work + a write to its own slot), the walk consuming
//     wave after wave and stopping at a random item -- the items the walk looked at and their order must be exactly those of the plain
//     sequential walk, every item it looked at must have been evaluated before, no item twice, and nothing past the stopping wave;
//   * with tokens: the first wave runs while the owner is still "computing the colour counts"; no walk starts before the release and
//     every walk sees the owner's writes;
//   * staging-style groups submitted to the FRONT of the queue while walks are running; an exception inside a task ends up in the
//     group's error string, not in std::terminate.
// build: g++ -std=c++17 -O1 -g -fsanitize=thread -pthread tests/cpp/group_waves_tsan.cpp -o t   (CPU only)
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <vector>

#include "../../line-mod-pipeline_amd/host/GroupWaves.h"
#include "../../line-mod-pipeline_amd/host/WorkerPool.h"

using namespace lmamd;

static int failures = 0;
#define CHECK(c) do { if (!(c)) { ++failures; std::fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, #c); } } while (0)

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 200;
    const int threads = argc > 2 ? std::atoi(argv[2]) : 8;
    uint32_t seed = 12345;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    WorkerPool pool(threads);
    CHECK(pool.threads() == threads);
    // ---- r05: submit_many -- every task runs exactly once, a group's error is kept, and with a single-threaded pool (the caller runs everything
    // in wait) the queue order is visible: a batch submitted at the FRONT runs before what was queued earlier, in its own order
    {
        WorkerPool::Group g;
        std::vector<std::atomic<int>> ran(300);
        for (auto& r : ran) r.store(0);
        std::vector<std::function<void()>> fns;
        for (int i = 0; i < 300; ++i) fns.push_back([&ran, i] { ran[(size_t)i].fetch_add(1); if (i == 7) throw std::runtime_error("task seven"); });
        pool.submit_many(g, std::move(fns));
        CHECK(fns.empty());
        pool.wait(g);
        for (auto& r : ran) CHECK(r.load() == 1);
        CHECK(g.error == "task seven");
        WorkerPool one(1);
        WorkerPool::Group h;
        std::vector<int> order;
        std::vector<std::function<void()>> back, front;
        for (int i = 0; i < 4; ++i) back.push_back([&order, i] { order.push_back(100 + i); });
        for (int i = 0; i < 3; ++i) front.push_back([&order, i] { order.push_back(i); });
        one.submit_many(h, std::move(back));
        one.submit_many(h, std::move(front), true);
        one.submit_many(h, std::vector<std::function<void()>>());        // (nothing: no wake-up, no count)
        one.wait(h);
        CHECK((order == std::vector<int>{0, 1, 2, 100, 101, 102, 103}));
    }
    for (int round = 0; round < rounds; ++round) {
        const size_t G = 1 + rnd() % 40;
        std::vector<size_t> len(G), stop(G);
        size_t total = 0;
        for (size_t g = 0; g < G; ++g) { len[g] = rnd() % 70; stop[g] = len[g] ? rnd() % (len[g] + 8) : 0; total += len[g]; }      // stop >= len: the walk runs to the end
        std::vector<size_t> base(G + 1, 0);
        for (size_t g = 0; g < G; ++g) base[g + 1] = base[g] + len[g];
        std::vector<std::atomic<int>> evaluated(total ? total : 1);
        for (auto& e : evaluated) e.store(0);
        std::vector<int> value(total ? total : 1, -1);              // written by evaluate, read by consume (ordering is what TSan checks)
        std::vector<std::vector<size_t>> walked(G);                 // per group: the items the walk looked at, in order
        std::vector<int> early_seen(G, 0);
        int owner_data = 0;                                          // written by the owner before release_tokens, read by every consume
        const bool with_tokens = round % 2 == 0;
        std::atomic<bool> released{!with_tokens};
        WorkerPool::Group tasks;
        GroupWaves waves(pool, tasks, len,
            [&](size_t g, size_t k, bool early) {
                volatile unsigned spin = 0;
                for (unsigned i = 0; i < (unsigned)(k * 37 % 200); ++i) spin += i;
                value[base[g] + k] = (int)(g * 1000 + k);
                if (early) early_seen[g] = 1;                        // (first wave only: one item per group, so one writer)
                evaluated[base[g] + k].fetch_add(1);
            },
            [&](size_t g, size_t from, size_t to) {
                CHECK(released.load());                              // never before the tokens are released
                if (with_tokens) CHECK(owner_data == 4711);
                for (size_t k = from; k < to; ++k) {
                    CHECK(evaluated[base[g] + k].load() == 1);
                    CHECK(value[base[g] + k] == (int)(g * 1000 + k));
                    walked[g].push_back(k);
                    if (k == stop[g]) return true;
                }
                return false;
            }, 1 + rnd() % 16);
        // a "staging" group jumps the queue while the walks run
        WorkerPool::Group staging;
        std::atomic<int> staged{0};
        waves.start(with_tokens);
        for (int i = 0; i < 12; ++i) pool.submit(staging, [&] { staged.fetch_add(1); }, true);
        if (with_tokens) {
            volatile unsigned spin = 0;
            for (unsigned i = 0; i < 20000; ++i) spin += i;          // the owner is busy (the GPU counts)
            owner_data = 4711;
            released.store(true);
            waves.release_tokens();
        }
        pool.wait(staging);
        pool.wait(tasks);
        CHECK(staged.load() == 12 && staging.error.empty() && tasks.error.empty());
        for (size_t g = 0; g < G; ++g) {
            // the plain sequential walk looks at 0 .. min(stop, len - 1)
            const size_t want = len[g] == 0 ? 0 : std::min(stop[g], len[g] - 1) + 1;
            CHECK(walked[g].size() == want);
            for (size_t k = 0; k < walked[g].size(); ++k) CHECK(walked[g][k] == k);
            // evaluated: everything the walk looked at, at most one wave beyond it, nothing twice
            size_t n_eval = 0;
            for (size_t k = 0; k < len[g]; ++k) { const int e = evaluated[base[g] + k].load(); CHECK(e == 0 || e == 1); n_eval += (size_t)e; if (k < want) CHECK(e == 1); }
            CHECK(n_eval <= want + 16);
            if (len[g]) CHECK(early_seen[g] == (with_tokens ? 1 : 0));
        }
    }
    // exceptions stay inside the group
    {
        WorkerPool::Group g;
        for (int i = 0; i < 20; ++i) pool.submit(g, [i] { if (i == 7) throw std::runtime_error("task 7 failed"); });
        pool.wait(g);
        CHECK(g.error == "task 7 failed");
    }
    // a pool of one thread is the caller alone
    {
        WorkerPool solo(1);
        WorkerPool::Group g;
        int n = 0;
        for (int i = 0; i < 100; ++i) solo.submit(g, [&n] { ++n; });
        solo.wait(g);
        CHECK(n == 100);
    }
    CHECK(usable_cpus() >= 1);
    if (failures) { std::printf("FAILED %d\n", failures); return 1; }
    std::printf("OK %d rounds, %d threads\n", rounds, threads);
    return 0;
}
