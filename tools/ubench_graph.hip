// Micro-benchmark: the cost of a chain of dependent kernel launches on gfx950, launched one by one
// on a stream (what the small-batch k-loop does today: 3 x n launches per call)
// against the same chain captured once into a hipGraph and launched as one graph.  The kernels
// mimic the small-batch form's shapes: grids of 20 / 10 / 128 workgroups that read what the
// previous kernel wrote (a real dependency through global memory, a few microseconds of work
// each or none).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_graph tools/ubench_graph.hip && tools/ubench_graph
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// `work` dependent multiply-adds per thread, then one word per thread read from src and written to dst
__global__ void __launch_bounds__(1024) k_link(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int work) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t v = src[i & 8191u];
    for (int k = 0; k < work; k++) v = v * 2654435761u + 12345u;
    dst[i & 8191u] = v + 1u;
}

static int enqueue_chain(hipStream_t st, uint32_t *a, uint32_t *b, uint32_t *c, int iters, int work) {
    for (int k = 0; k < iters; k++) {      // one k-loop iteration of the small-batch form: three launches
        hipLaunchKernelGGL(k_link, dim3(20), dim3(1024), 0, st, a, b, work);
        hipLaunchKernelGGL(k_link, dim3(10), dim3(1024), 0, st, b, c, work);
        hipLaunchKernelGGL(k_link, dim3(128), dim3(256), 0, st, c, a, work / 4);
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int main() {
    uint32_t *buf;
    CHECK(hipMalloc(&buf, 3 * 8192 * 4));
    CHECK(hipMemset(buf, 0, 3 * 8192 * 4));
    uint32_t *a = buf, *b = buf + 8192, *c = buf + 16384;
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 1024;
    for (int work : {0, 100, 200, 400, 700, 2000}) {   // 0: launch floor only; 100 ... 700: kernels of about 1.5 ... 11 us, the small-batch kernels' range
        // (1) stream launches
        if (enqueue_chain(st, a, b, c, 16, work)) return 1;
        CHECK(hipStreamSynchronize(st));
        float ms_stream = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0, st));
            if (enqueue_chain(st, a, b, c, iters, work)) return 1;
            CHECK(hipEventRecord(e1, st));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < ms_stream) ms_stream = ms;
        }
        // (2) the same chain as one graph
        hipGraph_t graph; hipGraphExec_t exec;
        CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        if (enqueue_chain(st, a, b, c, iters, work)) return 1;
        CHECK(hipStreamEndCapture(st, &graph));
        CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        CHECK(hipGraphLaunch(exec, st));
        CHECK(hipStreamSynchronize(st));
        float ms_graph = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0, st));
            CHECK(hipGraphLaunch(exec, st));
            CHECK(hipEventRecord(e1, st));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < ms_graph) ms_graph = ms;
        }
        CHECK(hipGraphExecDestroy(exec));
        CHECK(hipGraphDestroy(graph));
        printf("work %5d: %d x 3 dependent launches  stream %.3f ms (%.2f us per launch)   graph %.3f ms (%.2f us per launch)\n",
               work, iters, ms_stream, ms_stream * 1e3 / (3 * iters), ms_graph, ms_graph * 1e3 / (3 * iters));
    }
    return 0;
}
