// Independent callers on one key through the C ABI alone (include/sgfhe_hip.h): T host threads, each with a clone
// of its own (sgfhe_ctx_clone), each making calls of G gates with the drop-in host-pointer entry point
// (sgfhe_bootstrap_batch) as fast as it can -- what T Julia tasks on T threads running bootstrap(hkey, nothing, ...)
// amount to, without an interpreter between the calls (tools/callers.py has one: eight Python threads take turns on
// the GIL between calls, and the gathering window then closes on part of them).  Prints aggregate gates per second
// for each T, the ratio to one caller, the mean call latency and how the calls were gathered; every result is
// compared with the same call made alone.
//   g++ -O2 -std=c++17 -pthread -Iinclude -o tools/abl/callers tools/callers.cpp -Lsgfhe.jl_amd/csrc -lsgfhe_hip \
//       -Wl,-rpath,'$ORIGIN/../../sgfhe.jl_amd/csrc'
//   tools/abl/callers [n = 1024] [gates = 1] [seconds = 3] [gather = 1] [window_us = 300] [random = 0]
//                     [req_max = 32] [gates_max = 256] [most callers = 32]      (sgfhe_set_coalesce's limits)
// random = 1: bootstrap(hkey, rng, ...) -- every caller sets a flatten key of its own before every call, as
// julia/SGFHEHip.jl does with 32 bytes of the caller's rng (so every call is call 0 of its stream).
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "sgfhe_hip.h"

// Params(n) of the reference (src/fhe.jl:43-97) for the rings this tool is run on
static bool params_of(uint64_t n, sgfhe_params *p) {
    struct { uint64_t n; const char *Q; } tab[] = {{64, "5494391545392009217"}, {512, "1440321777275241790332929"},
                                                   {1024, "92180593745615474572738561"}};
    for (auto &t : tab)
        if (t.n == n) {
            unsigned __int128 Q = 0;
            for (const char *c = t.Q; *c; c++) Q = Q * 10 + (unsigned)(*c - '0');
            const unsigned __int128 r = 16 * n, B = 35 * r * r * n, D = Q / 8;
            *p = sgfhe_params{n, (uint64_t)r, (uint64_t)(r / 2), 2, {(uint64_t)Q, (uint64_t)(Q >> 64)},
                              {(uint64_t)B, (uint64_t)(B >> 64)}, {(uint64_t)D, (uint64_t)(D >> 64)}};
            return true;
        }
    return false;
}

#define OK(call) do { int32_t rc_ = (call); if (rc_) { printf("%s -> %d: %s\n", #call, rc_, sgfhe_last_error_string(ctx)); exit(1); } } while (0)

int main(int argc, char **argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1024;
    const size_t gates = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1;
    const double seconds = argc > 3 ? atof(argv[3]) : 3.0;
    const int gather = argc > 4 ? atoi(argv[4]) : 1;
    const uint32_t window = argc > 5 ? (uint32_t)atoi(argv[5]) : 300;
    const int rnd = argc > 6 ? atoi(argv[6]) : 0;
    const uint32_t req_max = argc > 7 ? (uint32_t)atoi(argv[7]) : 32;        // the library's defaults
    const uint32_t gates_max = argc > 8 ? (uint32_t)atoi(argv[8]) : 256;
    const int most = argc > 9 ? atoi(argv[9]) : 32;
    sgfhe_params p;
    if (!params_of(n, &p)) { printf("n must be 64, 512 or 1024\n"); return 2; }
    sgfhe_ctx *ctx = nullptr;
    OK(sgfhe_ctx_create(&p, 0, &ctx));
    std::mt19937_64 rng(1);
    std::vector<uint64_t> sk(n);
    for (auto &x : sk) x = rng() & 1;
    uint8_t seed[32] = {2};
    OK(sgfhe_bkey_generate(ctx, sk.data(), n, seed, (uint32_t)n));
    std::vector<int> counts;
    for (int t = 1; t <= most; t *= 2) counts.push_back(t);
    const int tmax = counts.back();
    const size_t row = 3 * (n + 1);
    struct Work { std::vector<uint64_t> a1, b1, a2, b2, ref; };
    std::vector<Work> work(tmax);
    auto fkey = [](int t, uint8_t *k) { for (int i = 0; i < 32; i++) k[i] = (uint8_t)(17 * t + i); };
    for (int t = 0; t < tmax; t++) {
        Work &w = work[t];
        w.a1.resize(gates * n); w.a2.resize(gates * n); w.b1.resize(gates); w.b2.resize(gates); w.ref.resize(gates * row);
        for (auto *v : {&w.a1, &w.a2, &w.b1, &w.b2}) for (auto &x : *v) x = rng() % p.r;
        if (rnd) { uint8_t k[32]; fkey(t, k); OK(sgfhe_set_random_flatten_key(ctx, 1, k)); }
        OK(sgfhe_bootstrap_batch(ctx, w.a1.data(), w.b1.data(), w.a2.data(), w.b2.data(), gates, w.ref.data(), 0));   // alone
    }
    OK(sgfhe_set_random_flatten(ctx, 0, 0));
    std::vector<sgfhe_ctx *> clones(tmax);
    for (auto &c : clones) OK(sgfhe_ctx_clone(ctx, &c));
    OK(sgfhe_set_coalesce(ctx, gather, req_max, gates_max, window));
    printf("Params(%llu), calls of %zu gate(s) through sgfhe_bootstrap_batch, %s flatten, %.1f s per point, gathering %s (window %u us, requests up to %u gates, %u per chain), %s\n",
           (unsigned long long)n, gates, rnd ? "randomised (a key per caller and call)" : "deterministic", seconds,
           gather ? "on" : "off", window, req_max, gates_max, sgfhe_build_id());
    double base = 0;
    for (int T : counts) {
        std::atomic<bool> go{false}, stop{false}, bad{false};
        std::vector<uint64_t> calls(T, 0);
        std::vector<std::thread> th;
        uint64_t st0[4];
        OK(sgfhe_coalesce_stats(ctx, st0, 1));
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                const Work &w = work[t];
                std::vector<uint64_t> out(gates * row);
                uint8_t k[32];
                fkey(t, k);
                if (rnd) sgfhe_set_random_flatten_key(clones[t], 1, k);
                sgfhe_bootstrap_batch(clones[t], w.a1.data(), w.b1.data(), w.a2.data(), w.b2.data(), gates, out.data(), 0);   // warm
                while (!go.load()) std::this_thread::yield();
                while (!stop.load()) {
                    if (rnd) sgfhe_set_random_flatten_key(clones[t], 1, k);     // call 0 of this caller's stream again
                    if (sgfhe_bootstrap_batch(clones[t], w.a1.data(), w.b1.data(), w.a2.data(), w.b2.data(), gates, out.data(), 0) ||
                        ((calls[t]++ & 7) == 0 && memcmp(out.data(), w.ref.data(), out.size() * 8))) { bad = true; return; }
                }
            });
        std::this_thread::sleep_for(std::chrono::milliseconds(300));      // every thread warm
        OK(sgfhe_coalesce_stats(ctx, st0, 1));
        const auto t0 = std::chrono::steady_clock::now();
        go = true;
        std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
        stop = true;
        for (auto &x : th) x.join();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (bad) { printf("a thread got bytes that differ from the call made alone\n"); return 1; }
        uint64_t total = 0, st[4];
        for (auto c : calls) total += c;
        OK(sgfhe_coalesce_stats(ctx, st, 0));
        const double rate = total * gates / dt;
        if (!base) base = rate;
        printf("callers %2d: %8.1f calls/s %9.1f gates/s  x%.2f  %6.2f ms per call", T, total / dt, rate, rate / base, 1e3 * dt * T / (total ? total : 1));
        if (st[0]) printf("   gathered: %.2f requests per launch chain (most %llu)", (double)st[1] / st[0], (unsigned long long)st[3]);
        printf("\n");
        fflush(stdout);
    }
    for (auto c : clones) sgfhe_ctx_destroy(c);
    sgfhe_ctx_destroy(ctx);
    return 0;
}
