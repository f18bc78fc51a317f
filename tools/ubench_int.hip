// Micro-benchmark: issue rate of the gfx950 integer / fp64 VALU instructions the RNS butterflies
// are built from.  Prints lane-ops per CU per clock-equivalent relative to v_add_u32.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITER 2048

#define KERNEL(NAME, ASM)                                                          \
    __global__ void __launch_bounds__(256) NAME(uint32_t *out, uint32_t seed) {     \
        uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3; \
        uint32_t a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;     \
        uint32_t b = seed | 1;                                                      \
        for (int i = 0; i < ITER; i++) {                                            \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)   \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(b) : "vcc");                                                 \
        }                                                                           \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; \
    }

#define A_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define A_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define A_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define A_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define A_MULHI24(i) "v_mul_hi_u32_u24 %" #i ", %" #i ", %8\n"
#define A_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %" #i "\n"
#define A_MIN(i) "v_min_u32 %" #i ", %" #i ", %8\n"
#define A_SUB(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
#define A_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define A_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 1, %8\n"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %8\n"

KERNEL(k_add, A_ADD)
KERNEL(k_mullo, A_MULLO)
KERNEL(k_mulhi, A_MULHI)
KERNEL(k_mul24, A_MUL24)
KERNEL(k_mulhi24, A_MULHI24)
KERNEL(k_mad24, A_MAD24)
KERNEL(k_min, A_MIN)
KERNEL(k_sub, A_SUB)
KERNEL(k_xor, A_XOR)
KERNEL(k_lshladd, A_LSHLADD)
KERNEL(k_add3, A_ADD3)
#define A_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_SUBCO(i) "v_sub_co_u32 %" #i ", vcc, %" #i ", %8\n"
#define A_CMP(i) "v_cmp_lt_u32 vcc, %" #i ", %8\nv_add_u32 %" #i ", %" #i ", %8\n"
#define A_MAX(i) "v_max_u32 %" #i ", %" #i ", %8\n"
#define A_MINI(i) "v_min_i32 %" #i ", %" #i ", %8\n"
#define A_LSHR(i) "v_lshrrev_b32 %" #i ", 1, %" #i "\n"
#define A_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %8\n"
#define A_ADDCO(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %8\n"
#define A_SUBCND(i) "v_sub_co_u32 %" #i ", vcc, %" #i ", %8\nv_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_SUBMIN(i) "v_sub_u32 %" #i ", %" #i ", %8\nv_min_u32 %" #i ", %" #i ", %8\n"
#define A_MULLOHI(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\nv_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define A_ASHR(i) "v_ashrrev_i32 %" #i ", 1, %" #i "\n"
#define A_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define A_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define A_OR(i) "v_or_b32 %" #i ", %" #i ", %8\n"
#define A_XAD(i) "v_xad_u32 %" #i ", %" #i ", %8, %8\n"
#define A_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %" #i ", 7\n"
#define A_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 1, 8\n"
#define A_MOV(i) "v_mov_b32 %" #i ", %8\n"
KERNEL(k_ashr, A_ASHR)
KERNEL(k_lshl, A_LSHL)
KERNEL(k_and, A_AND)
KERNEL(k_or, A_OR)
KERNEL(k_xad, A_XAD)
KERNEL(k_alignbit, A_ALIGNBIT)
KERNEL(k_bfe, A_BFE)
KERNEL(k_mov, A_MOV)
KERNEL(k_cndmask, A_CNDMASK)
KERNEL(k_subco, A_SUBCO)
KERNEL(k_cmp_add, A_CMP)
KERNEL(k_max, A_MAX)
KERNEL(k_mini, A_MINI)
KERNEL(k_lshr, A_LSHR)
KERNEL(k_andor, A_ANDOR)
KERNEL(k_addco, A_ADDCO)
KERNEL(k_subco_cnd, A_SUBCND)
KERNEL(k_sub_min, A_SUBMIN)
KERNEL(k_mullo_hi, A_MULLOHI)

// 64-bit forms
__global__ void __launch_bounds__(256) k_mad64(uint32_t *out, uint32_t seed) {
    uint64_t a[8];
    for (int j = 0; j < 8; j++) a[j] = threadIdx.x * (2 * j + 3) + seed;
    uint32_t b = seed | 1, c = seed * 7 + 1;
    for (int i = 0; i < ITER; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++)
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n" : "+v"(a[j]) : "v"(b), "v"(c) : "vcc");
    }
    uint64_t r = 0;
    for (int j = 0; j < 8; j++) r ^= a[j];
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32);
}
__global__ void __launch_bounds__(256) k_fma64(uint32_t *out, uint32_t seed) {
    double a[8];
    for (int j = 0; j < 8; j++) a[j] = threadIdx.x * (2 * j + 3) + seed;
    double b = 1.0000001, c = 1e-9 * seed;
    for (int i = 0; i < ITER; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_fma_f64 %0, %0, %1, %2\n" : "+v"(a[j]) : "v"(b), "v"(c));
    }
    double r = 0;
    for (int j = 0; j < 8; j++) r += a[j];
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)r;
}
__global__ void __launch_bounds__(256) k_fma32(uint32_t *out, uint32_t seed) {
    float a[8];
    for (int j = 0; j < 8; j++) a[j] = threadIdx.x * (2 * j + 3) + seed;
    float b = 1.0000001f, c = 1e-9f * seed;
    for (int i = 0; i < ITER; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_fma_f32 %0, %0, %1, %2\n" : "+v"(a[j]) : "v"(b), "v"(c));
    }
    float r = 0;
    for (int j = 0; j < 8; j++) r += a[j];
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)r;
}

template <typename K>
double run(K kern, uint32_t *d, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 2u + r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5.0;
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    int blocks = cus * 8;  // 8 x 256 threads per CU = 8 waves per SIMD
    uint32_t *d; hipMalloc(&d, (size_t)blocks * 256 * 4);
    double lane_ops = (double)blocks * 256 * ITER * 8;
    struct { const char *name; double ms; } res[64]; int n = 0;
#define RUN(K) res[n].name = #K; res[n].ms = run(K, d, blocks); n++;
    RUN(k_add) RUN(k_sub) RUN(k_min) RUN(k_xor) RUN(k_lshladd) RUN(k_add3) RUN(k_mullo) RUN(k_mulhi)
    RUN(k_mul24) RUN(k_mulhi24) RUN(k_mad24) RUN(k_mad64) RUN(k_fma32) RUN(k_fma64)
    RUN(k_cndmask) RUN(k_subco) RUN(k_cmp_add) RUN(k_max) RUN(k_mini) RUN(k_lshr) RUN(k_andor) RUN(k_addco) RUN(k_subco_cnd) RUN(k_sub_min) RUN(k_mullo_hi)
    RUN(k_ashr) RUN(k_lshl) RUN(k_and) RUN(k_or) RUN(k_xad) RUN(k_alignbit) RUN(k_bfe) RUN(k_mov)
    for (int i = 0; i < n; i++) {
        double rate = lane_ops / (res[i].ms * 1e-3);                 // lane-ops / s
        double per_cu_clk = rate / cus / (prop.clockRate * 1e3);    // at nominal clock
        printf("%-10s %8.3f ms  %7.2f Tlane-op/s  %6.1f lanes/CU/clk(nominal)  rel-to-add %.2fx slower\n",
               res[i].name, res[i].ms, rate * 1e-12, per_cu_clk, res[i].ms / res[0].ms);
    }
    return 0;
}
