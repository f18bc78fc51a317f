// Micro-benchmark: cost of one radix-16 register pass (4 butterfly stages on 16 points per
// thread, twiddles in registers) in four arithmetic styles, to decide whether a different
// butterfly is worth a rewrite of ntt.h:
//   u30   the engine's butterflies: unsigned Harvey lazy, p < 2^30, Montgomery twiddles
//         (rns_arith.h bfly_fwd / bfly_inv)
//   s29   signed Montgomery, p < 2^29, values in (-4p, 4p): no conditional subtraction and no
//         +2p offsets inside a stage; the X inputs are range-reduced once per pass (forward) or
//         every second stage (inverse)
//   f64   double precision, p < 2^49: h = y w, l = fma(y, w, -h), q = rint(y w/p),
//         t = fma(-q, p, h) + l; X + t, X - t with no reductions at all
//   u31   (round 4) unsigned, 2^30 < p < 2^31, values in [0, 2p) -- the only lazy
//         window a 32-bit word leaves such a prime: every sum and every difference needs its own
//         conditional correction (10 instructions per butterfly)
//   s30   (round 4) signed Montgomery on a 30-bit prime: the int32 window is (-2p, 2p), so the X
//         input of every butterfly is pulled back to [0, p] in every stage (8 instructions)
// Prints nanoseconds per butterfly per lane-slot at 4 and 8 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_bfly tools/ubench_bfly.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

// (self-contained: the unsigned 30-bit butterflies of round 1 are kept here for the comparison)

// ---- round-1 arithmetic (unsigned Harvey lazy butterflies, p < 2^30) --------------------------
struct Mod { uint32_t p, ninv, p2; };
__device__ __forceinline__ uint32_t condsub(uint32_t x, uint32_t m) {
    uint32_t d;
    const bool borrow = __builtin_usub_overflow(x, m, &d);
    return borrow ? x : d;
}
__device__ __forceinline__ uint32_t mont_lazy(uint32_t y, uint32_t wM, const Mod &md) {
    const uint64_t T = (uint64_t)wM * y;
    const uint32_t mq = (uint32_t)T * md.ninv;
    return (uint32_t)(((uint64_t)mq * md.p + T) >> 32);
}
__device__ __forceinline__ void bfly_fwd(uint32_t &X, uint32_t &Y, uint32_t wM, const Mod &md) {
    const uint32_t x = condsub(X, md.p2);
    const uint32_t t = mont_lazy(Y, wM, md);
    X = x + t;
    Y = x + md.p2 - t;
}
__device__ __forceinline__ void bfly_inv(uint32_t &X, uint32_t &Y, uint32_t wM, const Mod &md) {
    const uint32_t s = X + Y;
    const uint32_t t = X + md.p2 - Y;
    X = condsub(s, md.p2);
    Y = mont_lazy(t, wM, md);
}

#define ITER 256

// ---- unsigned 30-bit (the engine's) -----------------------------------------------------------
template <int B>
__device__ __forceinline__ void u_fwd_stage(uint32_t (&x)[16], const uint32_t (&t)[15], const Mod &md) {
    constexpr int NG = 1 << (3 - B);
#pragma unroll
    for (int g = 0; g < NG; g++)
#pragma unroll
        for (int l = 0; l < (1 << B); l++) {
            const int e0 = (g << (B + 1)) | l;
            bfly_fwd(x[e0], x[e0 | (1 << B)], t[NG - 1 + g], md);
        }
}
template <int B>
__device__ __forceinline__ void u_inv_stage(uint32_t (&x)[16], const uint32_t (&t)[15], const Mod &md) {
    constexpr int NG = 1 << (3 - B);
#pragma unroll
    for (int g = 0; g < NG; g++)
#pragma unroll
        for (int l = 0; l < (1 << B); l++) {
            const int e0 = (g << (B + 1)) | l;
            bfly_inv(x[e0], x[e0 | (1 << B)], t[NG - 1 + g], md);
        }
}
__global__ void __launch_bounds__(256) k_u30_fwd(uint32_t *out, uint32_t seed) {
    const uint32_t p = 1073479681u;  // 2^30 - 2^18 + 1 (shape only; exactness is not checked here)
    uint32_t inv = p;
    for (int i = 0; i < 5; i++) inv *= 2u - p * inv;
    const Mod md = {p, 0u - inv, 2 * p};
    uint32_t x[16], t[15];
    for (int e = 0; e < 16; e++) x[e] = (threadIdx.x * 2654435761u + e * 40503u + seed) % p;
    for (int e = 0; e < 15; e++) t[e] = (threadIdx.x * 97u + e * 7919u + seed * 3u) % p;
    for (int it = 0; it < ITER; it++) {
        u_fwd_stage<3>(x, t, md); u_fwd_stage<2>(x, t, md); u_fwd_stage<1>(x, t, md); u_fwd_stage<0>(x, t, md);
    }
    uint32_t r = 0;
    for (int e = 0; e < 16; e++) r ^= x[e];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
__global__ void __launch_bounds__(256) k_u30_inv(uint32_t *out, uint32_t seed) {
    const uint32_t p = 1073479681u;
    uint32_t inv = p;
    for (int i = 0; i < 5; i++) inv *= 2u - p * inv;
    const Mod md = {p, 0u - inv, 2 * p};
    uint32_t x[16], t[15];
    for (int e = 0; e < 16; e++) x[e] = (threadIdx.x * 2654435761u + e * 40503u + seed) % p;
    for (int e = 0; e < 15; e++) t[e] = (threadIdx.x * 97u + e * 7919u + seed * 3u) % p;
    for (int it = 0; it < ITER; it++) {
        u_inv_stage<0>(x, t, md); u_inv_stage<1>(x, t, md); u_inv_stage<2>(x, t, md); u_inv_stage<3>(x, t, md);
    }
    uint32_t r = 0;
    for (int e = 0; e < 16; e++) r ^= x[e];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

// ---- signed 29-bit ------------------------------------------------------------------------------
struct SMod {
    int32_t p, negp;
    uint32_t pinv;  // p^-1 mod 2^32
};
// a * w * 2^-32 mod p, |result| < |a| |w| / 2^32 + p / 2; any a in int32, |w| <= p / 2
__device__ __forceinline__ int32_t smont(int32_t a, int32_t w, const SMod &md) {
    const int64_t T = (int64_t)a * w;
    const int32_t m = (int32_t)((uint32_t)T * md.pinv);
    const int64_t U = (int64_t)m * md.negp + T;  // low word cancels
    return (int32_t)(U >> 32);
}
// x in (-4p, 4p) -> about (-p/2 - eps, p/2 + eps) for p just below 2^29
__device__ __forceinline__ int32_t sred(int32_t x, const SMod &md) {
    const int32_t q = (x + (1 << 28)) >> 29;
    return x - q * md.p;
}
template <int B>
__device__ __forceinline__ void s_fwd_stage(int32_t (&x)[16], const int32_t (&t)[15], const SMod &md) {
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
template <int B, bool RED>
__device__ __forceinline__ void s_inv_stage(int32_t (&x)[16], const int32_t (&t)[15], const SMod &md) {
    constexpr int NG = 1 << (3 - B);
#pragma unroll
    for (int g = 0; g < NG; g++)
#pragma unroll
        for (int l = 0; l < (1 << B); l++) {
            const int e0 = (g << (B + 1)) | l, e1 = e0 | (1 << B);
            const int32_t s = x[e0] + x[e1], d = x[e0] - x[e1];
            x[e0] = RED ? sred(s, md) : s;
            x[e1] = smont(d, t[NG - 1 + g], md);
        }
}
__global__ void __launch_bounds__(256) k_s29_fwd(uint32_t *out, uint32_t seed) {
    const int32_t p = 536608769;  // just below 2^29
    uint32_t inv = (uint32_t)p;
    for (int i = 0; i < 5; i++) inv *= 2u - (uint32_t)p * inv;
    const SMod md = {p, -p, inv};
    int32_t x[16], t[15];
    for (int e = 0; e < 16; e++) x[e] = (int32_t)((threadIdx.x * 2654435761u + e * 40503u + seed) % (uint32_t)p) - p / 2;
    for (int e = 0; e < 15; e++) t[e] = (int32_t)((threadIdx.x * 97u + e * 7919u + seed * 3u) % (uint32_t)p) - p / 2;
    for (int it = 0; it < ITER; it++) {
        // X inputs of the first stage of the pass are range-reduced (values grow by < p per stage)
#pragma unroll
        for (int e = 0; e < 8; e++) x[e] = sred(x[e], md);
        s_fwd_stage<3>(x, t, md); s_fwd_stage<2>(x, t, md); s_fwd_stage<1>(x, t, md); s_fwd_stage<0>(x, t, md);
    }
    uint32_t r = 0;
    for (int e = 0; e < 16; e++) r ^= (uint32_t)x[e];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
__global__ void __launch_bounds__(256) k_s29_inv(uint32_t *out, uint32_t seed) {
    const int32_t p = 536608769;
    uint32_t inv = (uint32_t)p;
    for (int i = 0; i < 5; i++) inv *= 2u - (uint32_t)p * inv;
    const SMod md = {p, -p, inv};
    int32_t x[16], t[15];
    for (int e = 0; e < 16; e++) x[e] = (int32_t)((threadIdx.x * 2654435761u + e * 40503u + seed) % (uint32_t)p) - p / 2;
    for (int e = 0; e < 15; e++) t[e] = (int32_t)((threadIdx.x * 97u + e * 7919u + seed * 3u) % (uint32_t)p) - p / 2;
    for (int it = 0; it < ITER; it++) {
        s_inv_stage<0, false>(x, t, md); s_inv_stage<1, true>(x, t, md);
        s_inv_stage<2, false>(x, t, md); s_inv_stage<3, true>(x, t, md);
    }
    uint32_t r = 0;
    for (int e = 0; e < 16; e++) r ^= (uint32_t)x[e];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

// ---- unsigned 31-bit: values in [0, 2p), p < 2^31 ----------------------------------------------------
// t = y w R^-1 mod p in [0, 2p) for any 32-bit y and w < p (T + mq p < 2^64)
__device__ __forceinline__ void bfly31_fwd(uint32_t &X, uint32_t &Y, uint32_t wM, const Mod &md) {
    const uint32_t t = mont_lazy(Y, wM, md);
    // X + t - 2p when that is not negative, else X + t:   d = X - (2p - t)
    uint32_t d, e;
    const bool b0 = __builtin_usub_overflow(X, md.p2 - t, &d);
    const bool b1 = __builtin_usub_overflow(X, t, &e);
    X = b0 ? d + md.p2 : d;
    Y = b1 ? e + md.p2 : e;
}
template <int B>
__device__ __forceinline__ void u31_fwd_stage(uint32_t (&x)[16], const uint32_t (&t)[15], const Mod &md) {
    constexpr int NG = 1 << (3 - B);
#pragma unroll
    for (int g = 0; g < NG; g++)
#pragma unroll
        for (int l = 0; l < (1 << B); l++) {
            const int e0 = (g << (B + 1)) | l;
            bfly31_fwd(x[e0], x[e0 | (1 << B)], t[NG - 1 + g], md);
        }
}
__global__ void __launch_bounds__(256) k_u31_fwd(uint32_t *out, uint32_t seed) {
    const uint32_t p = 2147352577u;  // 2^31 - 2^17 + 1 (shape only)
    uint32_t inv = p;
    for (int i = 0; i < 5; i++) inv *= 2u - p * inv;
    const Mod md = {p, 0u - inv, 2 * p};
    uint32_t x[16], t[15];
    for (int e = 0; e < 16; e++) x[e] = (threadIdx.x * 2654435761u + e * 40503u + seed) % p;
    for (int e = 0; e < 15; e++) t[e] = (threadIdx.x * 97u + e * 7919u + seed * 3u) % p;
    for (int it = 0; it < ITER; it++) {
        u31_fwd_stage<3>(x, t, md); u31_fwd_stage<2>(x, t, md); u31_fwd_stage<1>(x, t, md); u31_fwd_stage<0>(x, t, md);
    }
    uint32_t r = 0;
    for (int e = 0; e < 16; e++) r ^= x[e];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

// ---- signed 30-bit: window (-2p, 2p), X reduced in every stage ----------------------------------------
template <int B>
__device__ __forceinline__ void s30_fwd_stage(int32_t (&x)[16], const int32_t (&t)[15], const SMod &md) {
    constexpr int NG = 1 << (3 - B);
#pragma unroll
    for (int g = 0; g < NG; g++)
#pragma unroll
        for (int l = 0; l < (1 << B); l++) {
            const int e0 = (g << (B + 1)) | l, e1 = e0 | (1 << B);
            const int32_t tt = smont(x[e1], t[NG - 1 + g], md);      // |tt| < |y| / 8 + p / 2 < 0.75 p
            const int32_t X = x[e0] - (x[e0] >> 30) * md.p;          // floor form: [0, p] up to a few delta
            x[e0] = X + tt;
            x[e1] = X - tt;
        }
}
__global__ void __launch_bounds__(256) k_s30_fwd(uint32_t *out, uint32_t seed) {
    const int32_t p = 1073479681;  // just below 2^30
    uint32_t inv = (uint32_t)p;
    for (int i = 0; i < 5; i++) inv *= 2u - (uint32_t)p * inv;
    const SMod md = {p, -p, inv};
    int32_t x[16], t[15];
    for (int e = 0; e < 16; e++) x[e] = (int32_t)((threadIdx.x * 2654435761u + e * 40503u + seed) % (uint32_t)p) - p / 2;
    for (int e = 0; e < 15; e++) t[e] = (int32_t)((threadIdx.x * 97u + e * 7919u + seed * 3u) % (uint32_t)p) - p / 2;
    for (int it = 0; it < ITER; it++) {
        s30_fwd_stage<3>(x, t, md); s30_fwd_stage<2>(x, t, md); s30_fwd_stage<1>(x, t, md); s30_fwd_stage<0>(x, t, md);
    }
    uint32_t r = 0;
    for (int e = 0; e < 16; e++) r ^= (uint32_t)x[e];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

// ---- fp64, 49-bit primes -------------------------------------------------------------------------
template <int B>
__device__ __forceinline__ void f_fwd_stage(double (&x)[16], const double (&w)[15], const double (&wp)[15],
                                            double p) {
    constexpr int NG = 1 << (3 - B);
#pragma unroll
    for (int g = 0; g < NG; g++)
#pragma unroll
        for (int l = 0; l < (1 << B); l++) {
            const int e0 = (g << (B + 1)) | l, e1 = e0 | (1 << B);
            const double y = x[e1];
            const double h = y * w[NG - 1 + g];
            const double lo = __builtin_fma(y, w[NG - 1 + g], -h);
            const double q = __builtin_rint(y * wp[NG - 1 + g]);
            const double tt = __builtin_fma(-q, p, h) + lo;
            const double X = x[e0];
            x[e0] = X + tt;
            x[e1] = X - tt;
        }
}
__global__ void __launch_bounds__(256) k_f64_fwd(uint32_t *out, uint32_t seed) {
    const double p = 562949953290241.0 / 2.0 + 0.5;  // about 2^48 (shape only)
    double x[16], w[15], wp[15];
    for (int e = 0; e < 16; e++) x[e] = (double)((threadIdx.x * 2654435761u + e * 40503u + seed) % 1000003u) * 281474976.0;
    for (int e = 0; e < 15; e++) {
        w[e] = (double)((threadIdx.x * 97u + e * 7919u + seed * 3u) % 1000003u) * 140737488.0;
        wp[e] = w[e] / p;
    }
    for (int it = 0; it < ITER; it++) {
        f_fwd_stage<3>(x, w, wp, p); f_fwd_stage<2>(x, w, wp, p); f_fwd_stage<1>(x, w, wp, p); f_fwd_stage<0>(x, w, wp, p);
        // keep the X chain bounded the way a 13-stage transform would: one reduction per 3 passes
        if ((it & 3) == 3) {
#pragma unroll
            for (int e = 0; e < 16; e++) x[e] = x[e] - __builtin_rint(x[e] * (1.0 / p)) * p;
        }
    }
    double r = 0;
    for (int e = 0; e < 16; e++) r += x[e];
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(int64_t)r;
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
    hipMalloc(&d, (size_t)cus * 8 * 256 * 4);
    for (int wps : {4, 8}) {
        const int blocks = cus * wps;  // wps x 256 threads per CU = wps waves per SIMD
        const double bflies = (double)blocks * 256 * ITER * 32;
        struct { const char *name; double ms; } res[10];
        int n = 0;
#define RUN(K) res[n].name = #K; res[n].ms = run(K, d, blocks); n++;
        RUN(k_u30_fwd) RUN(k_s29_fwd) RUN(k_f64_fwd) RUN(k_u30_inv) RUN(k_s29_inv) RUN(k_u31_fwd) RUN(k_s30_fwd)
        for (int i = 0; i < n; i++)
            printf("%d waves/SIMD  %-10s %8.3f ms  %7.3f T butterflies/s  rel-to-u30_fwd %.3f  rel-to-s29_fwd %.3f\n", wps,
                   res[i].name, res[i].ms, bflies / (res[i].ms * 1e-3) * 1e-12, res[i].ms / res[0].ms,
                   res[i].ms / res[1].ms);
    }
    return 0;
}
