// The queueing half of the gathering of independent callers (sgfhe.jl_amd/csrc/coalescer.h) on the CPU, under
// ThreadSanitizer and under AddressSanitizer / UBSan (tests/test_host_sanitizer.py): the half that engine.hip adds --
// one batch on the leader's ctx -- is replaced by a stand-in that "computes" out[i] = f(owner's key, call, a1[i]) for
// every row of every request of the round after a short sleep, which is exactly the contract the real one has: a row's
// result depends on its own request only.
// T threads x K calls, mixed batch sizes, flags and flatten modes (requests gather only with their like); checks
//   - every request gets ITS result (nobody else's rows, nothing missing), whoever led its round;
//   - no two rounds run at once (the device-side buffers of a leader are its own, but the round is the unit the
//     `running` flag protects);
//   - a round never mixes flags or modes and never exceeds gates_max unless it is a single oversized request;
//   - the statistics add up, and callers on their own are not delayed by the gathering window;
//   - an error of the leader's run reaches every request of its round and nobody else;
//   - the knobs change under the callers' feet (sgfhe_set_coalesce from another thread) while every caller asks
//     wants() first, as sgfhe_bootstrap_batch does: requests it declines run on their own.
// Prints "ok <rounds> <requests> <requests declined by wants()>".  Any data race, lock-order inversion, use-after-return of a Req (they live on the
// callers' stacks) or leak aborts the run.
#if defined(__SANITIZE_THREAD__)
// ThreadSanitizer of this GCC does not intercept pthread_cond_clockwait, which libstdc++ uses for waits on the steady
// clock: it then misses the unlock / re-lock inside the wait and reports a double lock.  For the sanitizer build only,
// let <condition_variable> take its pthread_cond_timedwait path (the production build keeps the steady clock).
#include <bits/c++config.h>
#undef _GLIBCXX_USE_PTHREAD_COND_CLOCKWAIT
#endif
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <thread>

#include "coalescer.h"

using sgfhe::Coalescer;

static uint64_t f(uint32_t key0, uint32_t call, uint32_t flags, uint64_t a, size_t row) {
    uint64_t x = a * 0x9E3779B97F4A7C15ull + key0;
    x ^= (uint64_t)call << 32 | flags;
    return x * 1099511628211ull + row;
}

static std::atomic<int> in_round{0};
static std::atomic<uint64_t> rounds{0}, violations{0};

// what engine.hip's coalesced_call does around arrive() / finish(), with the stand-in for the combined call
static int32_t call(Coalescer &co, const void *owner, uint32_t key0, uint32_t callno, bool rnd, uint32_t flags,
                    const uint64_t *a1, size_t batch, uint64_t *out, bool fail_if_leading, std::string &err_out) {
    Coalescer::Req me;
    me.owner = owner;
    me.a1 = a1; me.b1 = a1; me.a2 = a1; me.b2 = a1;
    me.batch = batch; me.out = out; me.flags = flags;
    me.rnd = rnd; me.key[0] = key0; me.call = callno;
    std::vector<Coalescer::Req *> take;
    size_t gates = 0;
    const int lead = co.arrive(me, take, gates);
    if (lead < 0) return -6;
    if (lead == 0) {
        err_out = me.err;
        return me.rc;
    }
    if (in_round.fetch_add(1) != 0) violations++;                  // two rounds at once
    size_t sum = 0;
    bool mine = false;
    for (const Coalescer::Req *q : take) {
        sum += q->batch;
        mine |= q == &me;
        if (q->flags != me.flags || q->rnd != me.rnd) violations++;
    }
    if (!mine || sum != gates || (gates > co.gates_max && take.size() != 1)) violations++;
    std::this_thread::sleep_for(std::chrono::microseconds(150 + 20 * gates));
    int32_t rc = 0;
    std::string err;
    if (fail_if_leading) {
        rc = -4;
        err = "stand-in failure";
    } else {
        for (const Coalescer::Req *q : take)
            for (size_t i = 0; i < q->batch; i++) q->out[i] = f(q->key[0], q->call, q->flags, q->a1[i], i);
    }
    rounds++;
    in_round.fetch_sub(1);
    co.finish(take, gates, rc, err);
    err_out = err;
    return rc;
}

int main() {
    Coalescer co;
    co.gates_max = 24;
    co.window_us = 200;
    const int T = 12, K = 60;
    std::atomic<uint64_t> served{0}, failed{0}, wrong{0}, alone{0};
    std::atomic<bool> tuning{true};
    std::thread tuner([&] {                                 // what sgfhe_set_coalesce does, every half millisecond
        for (int i = 0; tuning.load(); i++) {
            {
                std::lock_guard<std::mutex> lk(co.mu);
                co.req_max = (i & 1) ? 4 : 32;
                co.window_us = (i & 2) ? 100 : 200;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(500));
        }
    });
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
            uint64_t s = 1000 + t;
            auto rnd64 = [&] { s = s * 6364136223846793005ull + 1442695040888963407ull; return s >> 17; };
            int owner_tag = t;                              // stands for the ctx
            for (int k = 0; k < K; k++) {
                const size_t batch = 1 + rnd64() % (t == 0 ? 30 : 5);          // thread 0 also sends requests above gates_max
                const bool rnd = (t % 3) == 0;
                const uint32_t flags = (t % 4) == 1 ? 1u : 0u;
                const bool poison = (t == 5 && k % 7 == 3);                     // this caller's run fails when it leads
                std::vector<uint64_t> a(batch), out(batch, 0xDEADull);
                for (auto &x : a) x = rnd64();
                std::string err;
                if (!co.wants(batch)) {                     // above the present req_max: the caller's own call
                    alone++;
                    continue;
                }
                const int32_t rc = call(co, &owner_tag, 77u + t, (uint32_t)k, rnd, flags, a.data(), batch, out.data(), poison, err);
                if (rc) {
                    failed++;
                    if (err != "stand-in failure") wrong++;
                    for (auto v : out) if (v != 0xDEADull) wrong++;           // a failed round writes nothing
                } else {
                    served++;
                    for (size_t i = 0; i < batch; i++)
                        if (out[i] != f(77u + t, (uint32_t)k, flags, a[i], i)) wrong++;
                }
                if (t % 2) std::this_thread::sleep_for(std::chrono::microseconds(rnd64() % 300));
            }
            co.forget(&owner_tag);
        });
    for (auto &x : th) x.join();
    tuning = false;
    tuner.join();
    co.req_max = 32;
    co.window_us = 200;
    // a caller on its own: no other caller seen in the last rounds, so no waiting for anybody
    int solo_tag = 0;
    std::vector<uint64_t> a(3, 5), out(3);
    std::string err;
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < 200; k++) call(co, &solo_tag, 1, k, false, 0, a.data(), 3, out.data(), false, err);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    const bool ok = wrong == 0 && violations == 0 && served + failed + alone == (uint64_t)T * K && alone > 0 &&
                    co.n_reqs == (uint64_t)T * K - alone + 200 &&
                    co.n_calls == rounds.load() && co.max_reqs >= 2 && co.pending.empty() && !co.running &&
                    ms < 200 * (0.15 + 0.06 + 0.25);                // 200 solo rounds of ~0.21 ms of stand-in work, no 0.2 ms windows
    if (!ok) {
        printf("FAILED: alone %llu wrong %llu violations %llu served %llu failed %llu n_reqs %llu n_calls %llu rounds %llu max %llu solo %.1f ms\n",
               (unsigned long long)alone.load(), (unsigned long long)wrong.load(), (unsigned long long)violations.load(),
               (unsigned long long)served.load(),
               (unsigned long long)failed.load(), (unsigned long long)co.n_reqs, (unsigned long long)co.n_calls,
               (unsigned long long)rounds.load(), (unsigned long long)co.max_reqs, ms);
        return 1;
    }
    printf("ok %llu %llu %llu\n", (unsigned long long)co.n_calls, (unsigned long long)co.n_reqs, (unsigned long long)alone.load());
    return 0;
}
