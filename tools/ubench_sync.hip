// Micro-benchmark: what a synchronisation between co-resident workgroups through global memory
// costs on gfx950, against the kernel-launch boundary the small-batch form of the k-loop uses today
// (three launches per iteration).  G workgroups of 1024 threads run `iters`
// rounds of: every thread stores a word of its workgroup's page; release; barrier; thread 0 adds 1
// to a counter at agent scope and spins (bounded) until it reads G (k + 1); acquire; barrier; every
// thread loads the word its left neighbour workgroup stored in this round and checks it.
// Prints microseconds per round and the number of stale reads (must be 0).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_sync tools/ubench_sync.hip && tools/ubench_sync
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(1024)
k_sync(uint32_t *pages, uint32_t *counter, uint32_t *bad, uint32_t *gaveup, int iters, int stride) {
    const uint32_t G = gridDim.x, g = blockIdx.x, t = threadIdx.x;
    const uint32_t left = (g + G - stride % G) % G;
    uint32_t nbad = 0;
    for (int k = 0; k < iters; k++) {
        pages[((size_t)(k & 1) * G + g) * 1024 + t] = (uint32_t)k * 2654435761u + g * 1024u + t;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (t == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t target = G * (uint32_t)(k + 1);
            uint32_t polls = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++polls > (1u << 22)) { atomicAdd(gaveup, 1u); break; }   // every wave reaches the exit
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const uint32_t v = __builtin_nontemporal_load(&pages[((size_t)(k & 1) * G + left) * 1024 + t]);
        const uint32_t want = (uint32_t)k * 2654435761u + left * 1024u + t;
        nbad += (v != want);
        // pages are double-buffered by round parity: a page of this parity is written again in round
        // k + 2, which its writer enters only after every workgroup has passed round k + 1's wait,
        // i.e. after all reads of round k
    }
    if (nbad) atomicAdd(bad, nbad);
}

__global__ void k_empty(uint32_t *p) { if (threadIdx.x == 9999) p[0] = 1; }

int main() {
    uint32_t *pages, *ctr;
    CHECK(hipMalloc(&pages, 2 * 256 * 1024 * 4));
    CHECK(hipMalloc(&ctr, 3 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 3000;
    for (int stride : {1, 8}) {           // left neighbour at g - 1 (another XCD) or g - 8 (the same XCD)
        for (int G : {2, 10, 30, 60, 240}) {
            CHECK(hipMemset(ctr, 0, 12));
            hipLaunchKernelGGL(k_sync, dim3(G), dim3(1024), 0, 0, pages, ctr, ctr + 1, ctr + 2, 10, stride);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemset(ctr, 0, 12));
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_sync, dim3(G), dim3(1024), 0, 0, pages, ctr, ctr + 1, ctr + 2, iters, stride);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            uint32_t h[3]; CHECK(hipMemcpy(h, ctr, 12, hipMemcpyDeviceToHost));
            printf("G = %3d workgroups, neighbour at g - %d: %.2f us per round, stale reads %u, gave up %u\n",
                   G, stride, ms * 1e3 / iters, h[1], h[2]);
        }
    }
    // the launch boundary it would replace: dependent empty kernels on one stream
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL(k_empty, dim3(30), dim3(1024), 0, 0, ctr);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("dependent empty launches (30 x 1024 threads): %.2f us per launch\n", ms * 1e3 / iters);
    return 0;
}
