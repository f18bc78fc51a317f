// Micro-benchmark: does data WRITTEN by one kernel stay in the 256 MB Infinity Cache (MALL) for the
// next kernel's reads?  (The question behind keeping k_extprod's residue hand-off off HBM,
// DESIGN.md section 8.)  For buffer sizes from 16 MB to 1 GB: kernel W overwrites the buffer
// (non-temporal or plain stores), kernel R reads it all back; both timed with HIP events, several
// rounds.  If read-after-write of a buffer below ~200 MB runs well above the HBM rate that the
// 1 GB case shows, the cache retains written lines.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_mall tools/ubench_mall.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool NT>
__global__ void __launch_bounds__(256) k_write(uint4 *p, size_t n16, uint32_t tag) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        const v4u v = {(uint32_t)i, tag, (uint32_t)i ^ tag, 7u};
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4u *>(&p[i]));
        else *reinterpret_cast<v4u *>(&p[i]) = v;
    }
}
__global__ void __launch_bounds__(256) k_read(const uint4 *p, size_t n16, uint32_t *out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;  // keeps the loads alive
}

int main() {
    const size_t sizes_mb[] = {16, 32, 64, 128, 192, 256, 384, 512, 1024};
    const size_t maxb = (size_t)1024 << 20;
    uint4 *buf;
    uint32_t *out;
    CHECK(hipMalloc(&buf, maxb));
    CHECK(hipMalloc(&out, 1 << 20));
    hipEvent_t e0, e1, e2;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
    const int grid = 256 * 16;
    for (int nt = 0; nt < 2; nt++) {
        for (size_t s : sizes_mb) {
            const size_t bytes = s << 20, n16 = bytes / 16;
            float tw = 0, tr = 0;
            const int rounds = 10;
            for (int r = 0; r < rounds + 2; r++) {
                CHECK(hipEventRecord(e0));
                if (nt) hipLaunchKernelGGL(k_write<true>, dim3(grid), dim3(256), 0, 0, buf, n16, (uint32_t)r);
                else hipLaunchKernelGGL(k_write<false>, dim3(grid), dim3(256), 0, 0, buf, n16, (uint32_t)r);
                CHECK(hipEventRecord(e1));
                hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, buf, n16, out);
                CHECK(hipEventRecord(e2));
                CHECK(hipEventSynchronize(e2));
                float a, b;
                CHECK(hipEventElapsedTime(&a, e0, e1));
                CHECK(hipEventElapsedTime(&b, e1, e2));
                if (r >= 2) { tw += a; tr += b; }
            }
            printf("%s stores  %5zu MB   write %7.0f GB/s   read-after-write %7.0f GB/s\n",
                   nt ? "non-temporal" : "plain       ", s, bytes / (tw / rounds) / 1e6, bytes / (tr / rounds) / 1e6);
            fflush(stdout);
        }
    }
    return 0;
}
