// Gathering of independent callers on one key (round 5): the queueing half, free of HIP so that it can be run under
// ThreadSanitizer on the CPU (tests/native/coalescer_tsan.cpp, tests/test_host_sanitizer.py).  engine.hip supplies
// the other half: what a leader does with the requests it took (one batch on its own ctx).
//
// Separate launch chains overlap only so far on this device -- the runtime feeds four hardware queues, and more busy
// queues than that are time-sliced: clones with a stream each reach 1.8 x one caller's rate at two callers and 3 x at
// four and beyond (profiles/r05_concurrent.txt) -- but ONE chain of g gates costs little more than a chain of one
// (15.1 ms for 1 gate, 20.8 for 8, 26.5 for 16 at Params(1024)).  So small host-pointer calls that arrive together
// on ctxs sharing a key (sgfhe_ctx_clone) are run as one call: the caller that finds no combined call in flight leads -- it takes every
// request waiting, runs them as one batch on ITS OWN ctx (its lock, lanes and streams) and hands each caller its rows
// -- and callers that arrive meanwhile wait for the next round.
// A row's result does not depend on the rows beside it (src/fhe.jl:579-582 is per bootstrap; tests/test_gpu_golden.py
// batch-position test), so every caller gets the bytes of its call made alone -- in the randomised flatten too:
// there every row of the combined call draws from the stream of the ctx it came in on (that ctx's key, the number
// of its call, the row's index in its call: kernels.h RndRow), and deterministic and randomised requests form
// separate rounds.
#pragma once

#include <stddef.h>
#include <stdint.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace sgfhe {

struct Coalescer {
    struct Req {
        const void *owner = nullptr;   // the ctx the request came in on
        const uint64_t *a1 = nullptr, *b1 = nullptr, *a2 = nullptr, *b2 = nullptr;
        size_t batch = 0;
        uint64_t *out = nullptr;
        uint32_t flags = 0;
        bool rnd = false;              // randomised flatten: the request's own draw stream (key, number of the call)
        uint32_t key[8] = {};
        uint32_t call = 0;
        int32_t rc = 0;
        bool done = false;
        std::string err;
    };
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Req *> pending;
    bool running = false;        // a leader is running a combined call
    // which ctxs had a request served in which round: the callers of the last two rounds are the ones a leader
    // expects back (with eight callers, half of them are still on their way back from the previous round when the
    // other half -- who waited through it -- could already start: two alternating rounds of four, at 3.5 x one
    // caller's rate, where one round of eight gives 5.8 x)
    uint64_t round = 0;
    std::unordered_map<const void *, uint64_t> seen;
    // knobs (sgfhe_set_coalesce): on / off, largest request that is gathered, gates per combined call, how long a
    // leader waits for the callers of the last two rounds to come back
    bool enabled = true;
    uint32_t req_max = 32, gates_max = 256, window_us = 300;
    // statistics (sgfhe_coalesce_stats)
    uint64_t n_calls = 0, n_reqs = 0, n_gates = 0, max_reqs = 0;

    // Is a call of `batch` gates one to gather?  (the knobs may change under a caller's feet: read under the lock)
    bool wants(size_t batch) {
        std::lock_guard<std::mutex> lk(mu);
        return enabled && batch <= req_max;
    }

    // A request arrives.  Returns 0 when another caller's combined call served it (me.rc / me.err hold its outcome);
    // 1 when the caller LEADS a round: `take` holds the requests of the round, `me` among them -- same flags and
    // flatten mode as `me`, at most gates_max gates (always at least `me`); the caller runs them and then calls
    // finish(); between the two calls no other round starts.  -1 when the host is out of memory: the request is
    // in no queue and nothing else has changed (the entry points of the C ABI must not throw).
    int arrive(Req &me, std::vector<Req *> &take, size_t &gates) {
        take.clear();
        gates = 0;
        std::unique_lock<std::mutex> lk(mu);
        try {
            pending.push_back(&me);
        } catch (...) {
            return -1;
        }
        cv.notify_all();                            // a leader gathering its round sees the arrival
        for (;;) {
            if (me.done) return 0;                  // a leader ran it
            if (!running && pending.front() == &me) break;     // nobody is running: the oldest request leads
            cv.wait(lk);
        }
        running = true;
        // The callers of the last two rounds are about to come back (they got their results microseconds ago):
        // wait until as many requests are here as there were callers, a few hundred microseconds at most.  A
        // caller on its own never waits.
        if (window_us) {
            size_t expect = 1;                                     // the leader itself
            for (auto it = seen.begin(); it != seen.end();) {
                if (it->first != me.owner && it->second + 2 > round) expect++;
                if (it->second + 64 <= round) it = seen.erase(it); else ++it;       // long gone
            }
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(window_us);
            while (pending.size() < expect && cv.wait_until(lk, deadline) != std::cv_status::timeout) {}
        }
        try {
            take.reserve(pending.size());           // (the only allocation of the round: nothing below can throw)
        } catch (...) {
            for (auto it = pending.begin(); it != pending.end(); ++it)
                if (*it == &me) { pending.erase(it); break; }
            running = false;
            cv.notify_all();                        // the next oldest request leads
            return -1;
        }
        for (auto it = pending.begin(); it != pending.end();) {
            if ((*it)->flags == me.flags && (*it)->rnd == me.rnd && (take.empty() || gates + (*it)->batch <= gates_max)) {
                take.push_back(*it);
                gates += (*it)->batch;
                it = pending.erase(it);
            } else {
                ++it;
            }
        }
        return 1;
    }

    // The leader's round is over: every request of it gets the outcome, the waiting callers are released, and the
    // oldest request still waiting leads the next round.
    void finish(const std::vector<Req *> &take, size_t gates, int32_t rc, const std::string &err) {
        std::lock_guard<std::mutex> lk(mu);
        for (Req *q : take) {
            q->rc = rc;
            if (rc) q->err = err;
            q->done = true;
            try {
                seen[q->owner] = round;             // (only the next leader's expectation depends on it)
            } catch (...) {
            }
        }
        round++;
        n_calls++;
        n_reqs += take.size();
        n_gates += gates;
        if (take.size() > max_reqs) max_reqs = take.size();
        running = false;
        cv.notify_all();
    }

    // a ctx goes away: the coalescer no longer expects it back
    void forget(const void *owner) {
        std::lock_guard<std::mutex> lk(mu);
        seen.erase(owner);
    }
};

}  // namespace sgfhe
