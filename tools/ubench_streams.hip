// Micro-benchmark: do dependent launch chains from SEVERAL host threads overlap on one device, and what does it
// depend on?  (round 5, sgfhe_ctx_clone: profiles/r05_concurrent.txt)
// T threads, each with a stream of its own, each queueing the latency form's chain -- n x 3 dependent launches of
// a few microseconds -- and waiting for it, R times.  Stream kinds:
//   plain    hipStreamCreateWithFlags(hipStreamNonBlocking): the runtime multiplexes streams onto its pool of
//            GPU_MAX_HW_QUEUES hardware queues (default 4); two streams on one hardware queue run one after the other
//   prio     hipStreamCreateWithPriority, priorities cycling low / normal / high (a queue pool per priority)
//   cumask   hipExtStreamCreateWithCUMask with every CU enabled: a hardware queue of the stream's own
// and two ways of queueing: launch by launch, or the chain captured once into a hipGraph per thread.
// UB_EXTRA_STREAMS=k in the environment: every thread creates k more (unused) plain streams after its own, as an
// engine ctx does (two lanes + two copy streams = 3 extra): if the runtime deals streams round-robin onto its
// hardware queues, the streams in USE then all sit on one queue and nothing overlaps.  UB_KINDS=plain limits the run.
// Prints aggregate chains per second and the ratio to one thread.
//   hipcc --offload-arch=gfx950 -O3 -pthread -o tools/abl/ubench_streams tools/ubench_streams.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(1024) k_link(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int work) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t v = src[i & 8191u];
    for (int k = 0; k < work; k++) v = v * 2654435761u + 12345u;
    dst[i & 8191u] = v + 1u;
}

static void enqueue_chain(hipStream_t st, uint32_t *a, uint32_t *b, uint32_t *c, int iters, int work) {
    for (int k = 0; k < iters; k++) {   // the quarter form's grids at one gate: 80 / 40 / 64 workgroups of 256 threads
        hipLaunchKernelGGL(k_link, dim3(80), dim3(256), 0, st, a, b, work);
        hipLaunchKernelGGL(k_link, dim3(40), dim3(256), 0, st, b, c, work);
        hipLaunchKernelGGL(k_link, dim3(64), dim3(256), 0, st, c, a, work);
    }
}

struct Worker {
    hipStream_t st = nullptr;
    uint32_t *buf = nullptr;
    hipGraphExec_t exec = nullptr;
    double queue_ms = 0;   // host time spent queueing, per chain
};

int main(int argc, char **argv) {
    const int iters = 1024, reps = 6;
    const int work = argc > 1 ? atoi(argv[1]) : 300;          // ~ 4-5 us per kernel
    int ncu = 0;
    CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    printf("chain = %d x 3 dependent launches, work %d, %d CUs, GPU_MAX_HW_QUEUES=%s\n", iters, work, ncu,
           getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(default)");
    const int extra = getenv("UB_EXTRA_STREAMS") ? atoi(getenv("UB_EXTRA_STREAMS")) : 0;
    const char *only = getenv("UB_KINDS");
    if (extra) printf("every thread creates %d more plain streams after its own and leaves them unused\n", extra);
    for (const char *kind : {"plain", "prio", "cumask"}) {
        if (only && strcmp(only, kind)) continue;
        for (int graph = 0; graph < 2; graph++) {
            double base = 0;
            for (int T : {1, 2, 4, 8, 16}) {
                std::vector<Worker> w(T);
                std::vector<hipStream_t> unused;
                for (int t = 0; t < T; t++) {
                    if (!strcmp(kind, "plain")) {
                        CHECK(hipStreamCreateWithFlags(&w[t].st, hipStreamNonBlocking));
                    } else if (!strcmp(kind, "prio")) {
                        int lo = 0, hi = 0;
                        CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));      // lo = least priority (largest number)
                        const int span = lo - hi + 1;
                        CHECK(hipStreamCreateWithPriority(&w[t].st, hipStreamNonBlocking, hi + (t % span)));
                    } else {
                        std::vector<uint32_t> mask((ncu + 31) / 32, 0xFFFFFFFFu);
                        if (ncu % 32) mask.back() = (1u << (ncu % 32)) - 1u;
                        CHECK(hipExtStreamCreateWithCUMask(&w[t].st, (uint32_t)mask.size(), mask.data()));
                    }
                    for (int x = 0; x < extra; x++) {
                        hipStream_t u;
                        CHECK(hipStreamCreateWithFlags(&u, hipStreamNonBlocking));
                        unused.push_back(u);
                    }
                    CHECK(hipMalloc(&w[t].buf, 3 * 8192 * 4));
                    CHECK(hipMemset(w[t].buf, 0, 3 * 8192 * 4));
                    uint32_t *a = w[t].buf, *b = a + 8192, *c = a + 16384;
                    enqueue_chain(w[t].st, a, b, c, 8, work);                  // warm: code object, queue
                    CHECK(hipStreamSynchronize(w[t].st));
                    if (graph) {
                        hipGraph_t g;
                        CHECK(hipStreamBeginCapture(w[t].st, hipStreamCaptureModeThreadLocal));
                        enqueue_chain(w[t].st, a, b, c, iters, work);
                        CHECK(hipStreamEndCapture(w[t].st, &g));
                        CHECK(hipGraphInstantiate(&w[t].exec, g, nullptr, nullptr, 0));
                        CHECK(hipGraphDestroy(g));
                        CHECK(hipGraphLaunch(w[t].exec, w[t].st));
                        CHECK(hipStreamSynchronize(w[t].st));
                    }
                }
                std::atomic<int> ready{0};
                std::atomic<bool> go{false};
                std::vector<std::thread> th;
                for (int t = 0; t < T; t++)
                    th.emplace_back([&, t] {
                        CHECK(hipSetDevice(0));
                        Worker &W = w[t];
                        uint32_t *a = W.buf, *b = a + 8192, *c = a + 16384;
                        ready++;
                        while (!go.load()) std::this_thread::yield();
                        for (int r = 0; r < reps; r++) {
                            const auto q0 = std::chrono::steady_clock::now();
                            if (W.exec) CHECK(hipGraphLaunch(W.exec, W.st));
                            else enqueue_chain(W.st, a, b, c, iters, work);
                            W.queue_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - q0).count() / reps;
                            CHECK(hipStreamSynchronize(W.st));
                        }
                    });
                while (ready.load() < T) std::this_thread::yield();
                const auto t0 = std::chrono::steady_clock::now();
                go.store(true);
                for (auto &x : th) x.join();
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                const double rate = T * reps / ms * 1e3;
                if (T == 1) base = rate;
                double q = 0;
                for (auto &x : w) q += x.queue_ms / T;
                printf("%-6s %-6s threads %2d: %7.1f chains/s  x%.2f   one chain %.2f ms, of which the host queues for %.2f ms\n", kind,
                       graph ? "graph" : "launch", T, rate, rate / base, ms / reps, q);
                fflush(stdout);
                for (auto u : unused) CHECK(hipStreamDestroy(u));
                for (auto &x : w) {
                    if (x.exec) CHECK(hipGraphExecDestroy(x.exec));
                    CHECK(hipStreamDestroy(x.st));
                    CHECK(hipFree(x.buf));
                }
            }
        }
    }
    return 0;
}
