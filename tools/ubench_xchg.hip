// Micro-benchmark for the one exchange of the LDS-staged NTT that stays inside a wavefront: between
// the layouts S = LOGE and S = 0 (ntt.h exchange_sync) the 16 x 16 points of 16 consecutive threads
// are transposed.  ntt.h does it through LDS (16 ds_write_b32 + wavefront fence + 16 ds_read_b32 on
// XOR-swizzled addresses, no workgroup barrier).  north_star names "wavefront shuffle primitives
// chosen for gfx950" for this step, so here is the same transpose in registers with DPP row
// operations (the 16 threads are exactly one DPP row): four exchange steps, step b swaps register
// bit b against lane bit b; bits 0 / 1 through quad_perm with a lane-mask select, bits 2 / 3
// through row_shl / row_shr / row_ror with the bank mask doing the select.
// Each variant runs between signed 29-bit radix-16 register passes (rns_arith.h arithmetic), i.e.
// under the vector-ALU pressure the exchange sees in k_extprod, at 4 waves per SIMD.
// Prints time per (pass + exchange), and checks that both transposes give the same values.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_xchg tools/ubench_xchg.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

struct SMod { int32_t p, negp; uint32_t pinv; };
__device__ __forceinline__ int32_t smont(int32_t a, int32_t w, const SMod &md) {
    const int64_t T = (int64_t)a * w;
    const int32_t m = (int32_t)((uint32_t)T * md.pinv);
    return (int32_t)(((int64_t)m * md.negp + T) >> 32);
}
__device__ __forceinline__ int32_t sred_floor(int32_t x, const SMod &md) { return x - (x >> 29) * md.p; }
template <int B>
__device__ __forceinline__ void fwd_stage(int32_t (&x)[16], const int32_t (&t)[15], const SMod &md) {
    constexpr int NG = 1 << (3 - B);
#pragma unroll
    for (int g = 0; g < NG; g++)
#pragma unroll
        for (int l = 0; l < (1 << B); l++) {
            const int e0 = (g << (B + 1)) | l, e1 = e0 | (1 << B);
            const int32_t tt = smont(x[e1], t[NG - 1 + g], md);
            const int32_t X = x[e0];
            x[e0] = X + tt;
            x[e1] = X - tt;
        }
}
__device__ __forceinline__ void pass(int32_t (&x)[16], const int32_t (&t)[15], const SMod &md) {
#pragma unroll
    for (int e = 0; e < 8; e++) x[e] = sred_floor(x[e], md);
    fwd_stage<3>(x, t, md); fwd_stage<2>(x, t, md); fwd_stage<1>(x, t, md); fwd_stage<0>(x, t, md);
}

// ---- LDS form (ntt.h: layout S = 4 stores, layout S = 0 loads, E = 16 swizzle) ------------------
__host__ __device__ constexpr uint32_t swz_bits(uint32_t idx) {
    return (((idx >> 5) & 1u) * 0x01u) ^ (((idx >> 6) & 1u) * 0x02u) ^ (((idx >> 7) & 1u) * 0x04u) ^
           (((idx >> 8) & 1u) * 0x18u);
}
__host__ __device__ constexpr uint32_t swz(uint32_t idx) { return idx ^ swz_bits(idx); }
// thread tid = (hi = tid >> 4, lo = tid & 15) holds points idx(e) = hi << 8 | e << 4 | lo (S = 4);
// afterwards thread tid holds idx(e) = tid << 4 | e (S = 0)
__device__ __forceinline__ void xchg_lds(int32_t (&x)[16], uint32_t *lds, int tid) {
    const uint32_t hi = (uint32_t)tid >> 4, lo = (uint32_t)tid & 15u;
    const uint32_t pb = swz((hi << 8) | lo) << 2;
    char *base = reinterpret_cast<char *>(lds);
#pragma unroll
    for (int e = 0; e < 16; e++) *reinterpret_cast<int32_t *>(base + (pb ^ (swz((uint32_t)e << 4) << 2))) = x[e];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    const uint32_t pl = swz((uint32_t)tid << 4) << 2;
#pragma unroll
    for (int e = 0; e < 16; e++) x[e] = *reinterpret_cast<const int32_t *>(base + (pl ^ (swz((uint32_t)e) << 2)));
}

// ---- DPP form ---------------------------------------------------------------------------------------
// new x[e] of lane l = old x[l & 15] of lane (l & ~15) | e: a 16 x 16 transpose inside every row.
template <int CTRL, int BANK>
__device__ __forceinline__ int32_t dpp_into(int32_t old, int32_t src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xF, BANK, false);
}
template <int B>
__device__ __forceinline__ void xstep(int32_t (&x)[16], bool bit) {
#pragma unroll
    for (int e0 = 0; e0 < 16; e0++) {
        if (e0 & (1 << B)) continue;
        const int e1 = e0 | (1 << B);
        const int32_t a = x[e0], b = x[e1];
        if constexpr (B == 0) {         // partner lane = l ^ 1: quad_perm [1, 0, 3, 2]
            const int32_t pa = __builtin_amdgcn_mov_dpp(a, 0xB1, 0xF, 0xF, false);
            const int32_t pb = __builtin_amdgcn_mov_dpp(b, 0xB1, 0xF, 0xF, false);
            x[e1] = bit ? b : pa;
            x[e0] = bit ? pb : a;
        } else if constexpr (B == 1) {  // l ^ 2: quad_perm [2, 3, 0, 1]
            const int32_t pa = __builtin_amdgcn_mov_dpp(a, 0x4E, 0xF, 0xF, false);
            const int32_t pb = __builtin_amdgcn_mov_dpp(b, 0x4E, 0xF, 0xF, false);
            x[e1] = bit ? b : pa;
            x[e0] = bit ? pb : a;
        } else if constexpr (B == 2) {  // l ^ 4: lanes of banks 0, 2 read lane l + 4, banks 1, 3 lane l - 4
            x[e1] = dpp_into<0x104, 0x5>(b, a);   // row_shl:4 into banks 0 and 2
            x[e0] = dpp_into<0x114, 0xA>(a, b);   // row_shr:4 into banks 1 and 3
        } else {                        // l ^ 8 = rotation by 8 inside the row
            x[e1] = dpp_into<0x128, 0x3>(b, a);   // row_ror:8 into banks 0 and 1
            x[e0] = dpp_into<0x128, 0xC>(a, b);   // into banks 2 and 3
        }
    }
}
__device__ __forceinline__ void xchg_dpp(int32_t (&x)[16], int tid) {
    xstep<0>(x, (tid & 1) != 0);
    xstep<1>(x, (tid & 2) != 0);
    xstep<2>(x, false);
    xstep<3>(x, false);
}

#define ITER 128
template <int MODE>  // 0: register passes only, 1: + LDS exchange, 2: + DPP exchange
__global__ void __launch_bounds__(256) k_bench(uint32_t *out, uint32_t seed) {
    __shared__ uint32_t lds[256 * 16];
    const int32_t p = 536608769;
    uint32_t inv = (uint32_t)p;
    for (int i = 0; i < 5; i++) inv *= 2u - (uint32_t)p * inv;
    const SMod md = {p, -p, inv};
    int32_t x[16], t[15];
    for (int e = 0; e < 16; e++) x[e] = (int32_t)((threadIdx.x * 2654435761u + e * 40503u + seed) % (uint32_t)p) - p / 2;
    for (int e = 0; e < 15; e++) t[e] = (int32_t)((threadIdx.x * 97u + e * 7919u + seed * 3u) % (uint32_t)p) - p / 2;
    for (int it = 0; it < ITER; it++) {
        pass(x, t, md);
        if (MODE == 1) xchg_lds(x, lds, threadIdx.x);
        if (MODE == 2) xchg_dpp(x, threadIdx.x);
    }
    uint32_t r = 0;
    for (int e = 0; e < 16; e++) r ^= (uint32_t)x[e];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
// one transpose of known values: out[tid * 16 + e]
template <int MODE>
__global__ void __launch_bounds__(256) k_check(uint32_t *out) {
    __shared__ uint32_t lds[256 * 16];
    int32_t x[16];
    // S = 4 layout: thread (hi, lo) holds point hi << 8 | e << 4 | lo; value = point index
    for (int e = 0; e < 16; e++) x[e] = (int32_t)(((threadIdx.x >> 4) << 8) | (e << 4) | (threadIdx.x & 15));
    if (MODE == 1) xchg_lds(x, lds, threadIdx.x); else xchg_dpp(x, threadIdx.x);
    for (int e = 0; e < 16; e++) out[threadIdx.x * 16 + e] = (uint32_t)x[e];
}

template <typename K>
double run(K kern, uint32_t *d, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 2u + r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5.0;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, cus);
    uint32_t *d;
    hipMalloc(&d, (size_t)cus * 8 * 256 * 4 + 2 * 4096 * 4);
    // correctness: both forms give the S = 0 layout (thread tid holds points tid << 4 | e)
    uint32_t *c1 = d + (size_t)cus * 8 * 256, *c2 = c1 + 4096;
    hipLaunchKernelGGL(k_check<1>, dim3(1), dim3(256), 0, 0, c1);
    hipLaunchKernelGGL(k_check<2>, dim3(1), dim3(256), 0, 0, c2);
    static uint32_t h1[4096], h2[4096];
    hipMemcpy(h1, c1, sizeof h1, hipMemcpyDeviceToHost);
    hipMemcpy(h2, c2, sizeof h2, hipMemcpyDeviceToHost);
    int bad1 = 0, bad2 = 0;
    for (int i = 0; i < 4096; i++) { bad1 += h1[i] != (uint32_t)i; bad2 += h2[i] != (uint32_t)i; }
    printf("transpose check: LDS form %s, DPP form %s\n", bad1 ? "WRONG" : "ok", bad2 ? "WRONG" : "ok");
    for (int wps : {4, 8}) {
        const int blocks = cus * wps;
        const double passes = (double)blocks * 256 * ITER;  // per-thread passes
        const double base = run(k_bench<0>, d, blocks);
        const double lds = run(k_bench<1>, d, blocks);
        const double dpp = run(k_bench<2>, d, blocks);
        printf("%d waves/SIMD  pass only %.3f ms | pass + LDS exchange %.3f ms (exchange %.3f, %.2f ns per thread-pass) | "
               "pass + DPP exchange %.3f ms (exchange %.3f, %.2f ns) | DPP / LDS total %.3f, butterflies/s LDS %.2f T DPP %.2f T\n",
               wps, base, lds, lds - base, (lds - base) * 1e6 / passes * blocks * 256 / (blocks * 256),
               dpp, dpp - base, (dpp - base) * 1e6 / passes * blocks * 256 / (blocks * 256), dpp / lds,
               passes * 32 / (lds * 1e-3) * 1e-12, passes * 32 / (dpp * 1e-3) * 1e-12);
    }
    return bad1 || bad2;
}
