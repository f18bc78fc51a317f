// Micro-benchmark (round 4): what DESIGN.md section 9.3's latency idea can buy.
// One bootstrap() call of a few gates is a dependent chain of 3 x n launches (k_fwd_phase,
// k_inv_column, k_crt_lean; 24-25 ms at Params(1024)).  The idea: cut each 8192-point transform of the
// two transform kernels across four workgroups --
//   forward: every workgroup reads all four quarters of the digit plane, forms its own quarter of
//            the first two Cooley-Tukey stages (one radix-4 combination per point) and runs a
//            2048-point transform (256 threads x 8 points), then the products with the key;
//   inverse: (x^j - 1) applied in the NTT domain (one more product per slot), four workgroups each
//            run a 2048-point inverse transform of a quarter of the slots and store a partial
//            polynomial; the CRT kernel finishes with one radix-4 combination per prime and coefficient
//            (four residue loads instead of one).
// This tool prices it WITHOUT building it: timing-only kernels with the engine's real transform code
// (ntt.h, rns_arith.h), the same loads, stores, products and grid shapes, but the twiddle tables of
// the full-size ring (the sub-transform of a quarter needs tables of its own), so the numbers they
// produce are not a transform of anything.  It runs, per variant, a dependent chain of launches on
// one stream -- the shape of the k-loop -- and prints microseconds per launch:
//   fwd_full   the engine's k_fwd_phase shape: npr x 4 workgroups of 1024 threads, m = 8192
//   fwd_split  npr x 4 x 4 workgroups of 256 threads, m = 2048 each, 4 x the digit loads + radix-4 combine
//   inv_full   the engine's k_inv_column shape: npr x 2 workgroups of 1024 threads
//   inv_split  npr x 2 x 4 workgroups of 256 threads + the diagonal product
//   crt_full / crt_split   the CRT kernel's loads: one residue per prime and coefficient against four
//                          plus the radix-4 combination (3 Montgomery products per prime)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench_split tools/ubench_split.hip
#include "../sgfhe.jl_amd/csrc/kernels.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

using namespace sgfhe;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int LOGM = 13, M = 1 << LOGM, LE3 = 3;

// Round 4, second use: where a quarter kernel's microseconds go.  Timing-only switches for the two split
// kernels (the ntt.h ones -- SGFHE_ABL_NO_TW, SGFHE_ABL_NO_LDS, SGFHE_ABL_NO_BARRIER -- apply as well):
//   UB_NO_DIG   no input loads (digit planes / partial products)      UB_NO_KEY  no key / rotation-factor loads
//   UB_CONST_PS the per-prime records from constant memory filled by the host instead of through the pointer
#ifdef UB_CONST_PS
__constant__ PrimeK c_pk[8];
#define UB_NPR(PS) (c_pk[0].npr)
#define UB_PRIME(PS, i) (c_pk[i])
#else
#define UB_NPR(PS) ((PS)[0].npr)
#define UB_PRIME(PS, i) ((PS)[i])
#endif
__global__ void k_empty(uint32_t *p) { if (p && threadIdx.x == 9999) p[0] = 1; }

// ---- forward -----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024, 1)
k_fwd_full(const uint64_t *__restrict__ dig, const int32_t *__restrict__ keyk, int32_t *__restrict__ zpart,
           PrimeSet PS) {
    using G = NttGeom<LOGM, LE3>;
    constexpr int T = G::T, E = G::E;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x;
    const uint32_t npr = PS[0].npr, ph = blockIdx.x & 3u, pi = (blockIdx.x >> 2) % npr;
    const PrimeK P = PS[pi];
    const Mod md = mod_of(P);
    int32_t x[1][E];
    const uint32_t *dl = digit_lo_plane(dig, ph >> 1, M) + (ph & 1) * M;
    const uint16_t *dh = digit_hi_plane(dig, ph >> 1, M) + (ph & 1) * M;
#pragma unroll
    for (int e = 0; e < E; e++)
        x[0][e] = digit_reduce(dl[tid + T * e] | ((uint64_t)dh[tid + T * e] << 32), md, P.sR);
    ntt_forward<LOGM, 1, LE3>(x, lds, P.twf, tid, md);
    const int32_t *kp = keyk + ((size_t)pi * 8 + ph * 2) * M + E * tid;
    int32_t *zp = zpart + (((size_t)pi * 4 + ph) * 2) * M + E * tid;
#pragma unroll
    for (int h = 0; h < E / 4; h++) {
        const int4 a = reinterpret_cast<const int4 *>(kp)[h], bq = reinterpret_cast<const int4 *>(kp + M)[h];
        const int32_t ka[4] = {a.x, a.y, a.z, a.w}, kb[4] = {bq.x, bq.y, bq.z, bq.w};
        int32_t r0[4], r1[4];
#pragma unroll
        for (int t = 0; t < 4; t++) { r0[t] = smont(x[0][4 * h + t], ka[t], md); r1[t] = smont(x[0][4 * h + t], kb[t], md); }
        reinterpret_cast<int4 *>(zp)[h] = make_int4(r0[0], r0[1], r0[2], r0[3]);
        reinterpret_cast<int4 *>(zp + M)[h] = make_int4(r1[0], r1[1], r1[2], r1[3]);
    }
}

__global__ void __launch_bounds__(256, 1)
k_fwd_split(const uint64_t *__restrict__ dig, const int32_t *__restrict__ keyk, int32_t *__restrict__ zpart,
            PrimeSet PS) {
    constexpr int LS = LOGM - 2, MS = 1 << LS;          // the quarter: 2048 points, 256 threads x 8
    using G = NttGeom<LS, LE3>;
    constexpr int T = G::T, E = G::E;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x;
    const uint32_t npr = UB_NPR(PS), q = blockIdx.x & 3u, ph = (blockIdx.x >> 2) & 3u, pi = (blockIdx.x >> 4) % npr;
    const PrimeK P = UB_PRIME(PS, pi);
    const Mod md = mod_of(P);
    int32_t x[1][E];
    const uint32_t *dl = digit_lo_plane(dig, ph >> 1, M) + (ph & 1) * M;
    const uint16_t *dh = digit_hi_plane(dig, ph >> 1, M) + (ph & 1) * M;
    // first two stages of the 8192-point transform, this workgroup's output quarter q: a radix-4
    // combination of the four input quarters (three twiddled products, one of them shared)
    const int32_t wA = P.twf[1], wB = P.twf[2 + (q >> 1)], wP = P.twf[2 * M + 2 + (q >> 1)];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint32_t i = tid + T * e;
        int32_t X[4];
#pragma unroll
        for (int s = 0; s < 4; s++)
#ifdef UB_NO_DIG
            X[s] = digit_reduce((uint64_t)(i + s) * 0x9E3779B97F4Aull, md, P.sR);
#else
            X[s] = digit_reduce(dl[i + s * MS] | ((uint64_t)dh[i + s * MS] << 32), md, P.sR);
#endif
        const int32_t u = smont(X[2], wA, md);
        const int32_t a = (q & 2) ? X[0] - u : X[0] + u;
        const int32_t sgn = sredc((int64_t)X[1] * wB + (int64_t)X[3] * wP, md);
        x[0][e] = (q & 1) ? a - sgn : a + sgn;
    }
    ntt_forward<LS, 1, LE3>(x, lds, P.twf, tid, md);
    const int32_t *kp = keyk + ((size_t)pi * 8 + ph * 2) * M + q * MS + E * tid;
    int32_t *zp = zpart + (((size_t)pi * 4 + ph) * 2) * M + q * MS + E * tid;
#pragma unroll
    for (int h = 0; h < E / 4; h++) {
#ifdef UB_NO_KEY
        const int4 a = make_int4(tid, h, 5, 7), bq = make_int4(h, tid, 3, 9);
#else
        const int4 a = reinterpret_cast<const int4 *>(kp)[h], bq = reinterpret_cast<const int4 *>(kp + M)[h];
#endif
        const int32_t ka[4] = {a.x, a.y, a.z, a.w}, kb[4] = {bq.x, bq.y, bq.z, bq.w};
        int32_t r0[4], r1[4];
#pragma unroll
        for (int t = 0; t < 4; t++) { r0[t] = smont(x[0][4 * h + t], ka[t], md); r1[t] = smont(x[0][4 * h + t], kb[t], md); }
        reinterpret_cast<int4 *>(zp)[h] = make_int4(r0[0], r0[1], r0[2], r0[3]);
        reinterpret_cast<int4 *>(zp + M)[h] = make_int4(r1[0], r1[1], r1[2], r1[3]);
    }
}

// ---- inverse -----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024, 1)
k_inv_full(const int32_t *__restrict__ zpart, uint32_t *__restrict__ yres, PrimeSet PS, uint32_t j) {
    using G = NttGeom<LOGM, LE3>;
    constexpr int T = G::T, E = G::E;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x;
    const uint32_t npr = PS[0].npr, c = blockIdx.x & 1u, pi = (blockIdx.x >> 1) % npr;
    const PrimeK P = PS[pi];
    const Mod md = mod_of(P);
    int32_t z[1][E];
    const int32_t *zp = zpart + (((size_t)pi * 4) * 2 + c) * M + E * tid;
#pragma unroll
    for (int h = 0; h < E / 4; h++) {
        int32_t acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int ph = 0; ph < 4; ph++) {
            const int4 v = reinterpret_cast<const int4 *>(zp + (size_t)ph * 2 * M)[h];
            acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
        }
#pragma unroll
        for (int t = 0; t < 4; t++) z[0][4 * h + t] = sred(acc[t], md);
    }
    ntt_inverse<LOGM, 1, LE3>(z, lds, P.twi, tid, md);
    uint32_t *yb = yres + ((size_t)c * npr + pi) * M;
    const uint32_t yoff = 3u * (uint32_t)P.p + P.hoff;
    lds_store<LOGM, 1, LE3, G::STOP>(z, lds, tid);
    __syncthreads();
    constexpr uint32_t LOWMASK = (1u << G::STOP) - 1u;
    const uint32_t s0 = ((uint32_t)tid - j) & (2 * M - 1);
    const uint32_t lowswz = swz<LE3>(s0 & LOWMASK), h0 = s0 >> G::STOP;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint32_t he = h0 + e, hipart = (he & (E - 1)) << G::STOP;
        const int32_t v = (int32_t)lds[hipart ^ lowswz ^ swz_bits<LE3>(hipart)];
        yb[tid + T * e] = (uint32_t)(((he & E) ? -v : v) - z[0][e]) + yoff;
    }
}

__global__ void __launch_bounds__(256, 1)
k_inv_split(const int32_t *__restrict__ zpart, uint32_t *__restrict__ ypart, PrimeSet PS,
            const int32_t *__restrict__ diag) {
    constexpr int LS = LOGM - 2, MS = 1 << LS;
    using G = NttGeom<LS, LE3>;
    constexpr int T = G::T, E = G::E;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x;
    const uint32_t npr = UB_NPR(PS), q = blockIdx.x & 3u, c = (blockIdx.x >> 2) & 1u, pi = (blockIdx.x >> 3) % npr;
    const PrimeK P = UB_PRIME(PS, pi);
    const Mod md = mod_of(P);
    int32_t z[1][E];
    const int32_t *zp = zpart + (((size_t)pi * 4) * 2 + c) * M + q * MS + E * tid;
    const int32_t *dg = diag + (size_t)pi * M + q * MS + E * tid;     // psi^(j (2 brv(s) + 1)) - 1 per slot
#pragma unroll
    for (int h = 0; h < E / 4; h++) {
        int32_t acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int ph = 0; ph < 4; ph++) {
#ifdef UB_NO_DIG
            const int4 v = make_int4(tid + ph, h, 3, ph);
#else
            const int4 v = reinterpret_cast<const int4 *>(zp + (size_t)ph * 2 * M)[h];
#endif
            acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
        }
#ifdef UB_NO_KEY
        const int4 d = make_int4(tid, h, 11, 13);
#else
        const int4 d = reinterpret_cast<const int4 *>(dg)[h];
#endif
        const int32_t dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int t = 0; t < 4; t++) z[0][4 * h + t] = smont(acc[t], dd[t], md);   // (x^j - 1) in the NTT domain
    }
    ntt_inverse<LS, 1, LE3>(z, lds, P.twi, tid, md);
    uint32_t *yb = ypart + (((size_t)c * npr + pi) * 4 + q) * MS;                    // partial polynomial of quarter q
#pragma unroll
    for (int e = 0; e < E; e++) yb[tid + T * e] = (uint32_t)z[0][e] + 3u * (uint32_t)P.p;
}

// ---- the CRT kernel's side: one residue per prime and coefficient, or four + a radix-4 combination -----
template <int NP, bool SPLIT>
__global__ void __launch_bounds__(256)
k_crt_loads(const uint32_t *__restrict__ y, uint64_t *__restrict__ dig, PrimeSet PS, uint32_t quads) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= quads) return;
    const uint32_t i = (4u * t) & (M - 1), bc = (4u * t) >> LOGM;
    uint32_t acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < NP; q++) {
        const uint4 *src = reinterpret_cast<const uint4 *>(y + ((size_t)bc * NP + q) * M + i);
        if (!SPLIT) {
            const uint4 v = src[0];
            acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
        } else {   // the four partial polynomials of this coefficient + the last two inverse stages
            const PrimeK P = PS[q];
            const Mod md = mod_of(P);
            const uint32_t *base = y + ((size_t)bc * NP + q) * M + (i & (M / 4 - 1));
            const uint4 v0 = *reinterpret_cast<const uint4 *>(base), v1 = *reinterpret_cast<const uint4 *>(base + M / 4),
                        v2 = *reinterpret_cast<const uint4 *>(base + M / 2), v3 = *reinterpret_cast<const uint4 *>(base + 3 * M / 4);
            const uint32_t a0[4] = {v0.x, v0.y, v0.z, v0.w}, a1[4] = {v1.x, v1.y, v1.z, v1.w},
                           a2[4] = {v2.x, v2.y, v2.z, v2.w}, a3[4] = {v3.x, v3.y, v3.z, v3.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int32_t s0 = (int32_t)a0[k] + (int32_t)a1[k], s1 = (int32_t)a2[k] + (int32_t)a3[k];
                const int32_t d0 = (int32_t)a0[k] - (int32_t)a1[k], d1 = (int32_t)a2[k] - (int32_t)a3[k];
                const int32_t sel = (i >> (LOGM - 2)) & 3;
                const int32_t r = sel == 0 ? s0 + s1 : sel == 1 ? sredc((int64_t)d0 * P.r1 + (int64_t)d1 * P.r2, md)
                                : sel == 2 ? smont(s0 - s1, P.r3, md) : sredc((int64_t)d0 * P.r2 - (int64_t)d1 * P.r3, md);
                acc[k] += (uint32_t)r;
            }
        }
    }
    uint4 *o = reinterpret_cast<uint4 *>(dig + (size_t)bc * 2 * M) + (i >> 2);
    *o = make_uint4(acc[0], acc[1], acc[2], acc[3]);
}

template <class F>
static double chain_us(F launch, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 20; i++) launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < iters; i++) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / iters;
}

int main() {
    const uint32_t npr = 5;
    const uint32_t primes[5] = {536608769u, 536215553u, 535822337u, 535756801u, 535658497u};   // shapes only
    std::vector<int32_t> tw((size_t)npr * 4 * M);
    for (size_t i = 0; i < tw.size(); i++) tw[i] = (int32_t)((i * 2654435761u) % 268000000u) - 134000000;
    int32_t *d_tw;
    CHECK(hipMalloc(&d_tw, tw.size() * 4));
    CHECK(hipMemcpy(d_tw, tw.data(), tw.size() * 4, hipMemcpyHostToDevice));
    std::vector<PrimeK> pk(npr);
    for (uint32_t i = 0; i < npr; i++) {
        PrimeK &P = pk[i];
        memset(&P, 0, sizeof P);
        P.p = (int32_t)primes[i];
        uint32_t inv = primes[i];
        for (int it = 0; it < 5; it++) inv *= 2u - primes[i] * inv;
        P.pinv = inv;
        P.sR = 12345; P.r1 = 1111111; P.r2 = -2222222; P.r3 = 3333333;
        P.twf = d_tw + (size_t)(4 * i) * M;
        P.twi = d_tw + (size_t)(4 * i + 1) * M;
        P.npr = npr;
    }
    PrimeK *d_pk;
    CHECK(hipMalloc(&d_pk, npr * sizeof(PrimeK)));
    CHECK(hipMemcpy(d_pk, pk.data(), npr * sizeof(PrimeK), hipMemcpyHostToDevice));
#ifdef UB_CONST_PS
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_pk), pk.data(), npr * sizeof(PrimeK)));
#endif
    uint64_t *d_dig;
    int32_t *d_key, *d_z, *d_diag;
    uint32_t *d_y;
    CHECK(hipMalloc(&d_dig, (size_t)8 * 4 * M * 8));
    CHECK(hipMemset(d_dig, 1, (size_t)8 * 4 * M * 8));
    CHECK(hipMalloc(&d_key, (size_t)npr * 8 * M * 4));
    CHECK(hipMemset(d_key, 3, (size_t)npr * 8 * M * 4));
    CHECK(hipMalloc(&d_z, (size_t)npr * 8 * M * 4));
    CHECK(hipMemset(d_z, 0, (size_t)npr * 8 * M * 4));
    CHECK(hipMalloc(&d_diag, (size_t)npr * M * 4));
    CHECK(hipMemset(d_diag, 5, (size_t)npr * M * 4));
    CHECK(hipMalloc(&d_y, (size_t)2 * npr * M * 4 * 2));
    CHECK(hipMemset(d_y, 0, (size_t)2 * npr * M * 4 * 2));
    CHECK(hipFuncSetAttribute((const void *)k_fwd_full, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * M));
    CHECK(hipFuncSetAttribute((const void *)k_inv_full, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * M));
    const int iters = 2000;
    if (getenv("UB_ANATOMY")) {   // the split kernels only, one line (for the ablation builds)
        const double e0 = chain_us([&] { hipLaunchKernelGGL(k_empty, dim3(80), dim3(256), 0, 0, (uint32_t *)nullptr); }, iters);
        const double fs = chain_us([&] { hipLaunchKernelGGL(k_fwd_split, dim3(npr * 16), dim3(256), M, 0, d_dig, d_key, d_z, d_pk); }, iters);
        const double vs = chain_us([&] { hipLaunchKernelGGL(k_inv_split, dim3(npr * 8), dim3(256), M, 0, d_z, d_y, d_pk, d_diag); }, iters);
        const uint32_t quads = 2 * M / 4;
        const double cs = chain_us([&] { hipLaunchKernelGGL((k_crt_loads<5, true>), dim3((quads + 255) / 256), dim3(256), 0, 0, d_y, d_dig, d_pk, quads); }, iters);
        printf("%-28s empty %.2f   forward quarter %.2f   inverse quarter %.2f   CRT loads %.2f  (us per dependent launch)\n",
               getenv("UB_ANATOMY"), e0, fs, vs, cs);
        return 0;
    }
    const double ff = chain_us([&] { hipLaunchKernelGGL(k_fwd_full, dim3(npr * 4), dim3(1024), 4 * M, 0, d_dig, d_key, d_z, d_pk); }, iters);
    const double fs = chain_us([&] { hipLaunchKernelGGL(k_fwd_split, dim3(npr * 16), dim3(256), M, 0, d_dig, d_key, d_z, d_pk); }, iters);
    const double vf = chain_us([&] { hipLaunchKernelGGL(k_inv_full, dim3(npr * 2), dim3(1024), 4 * M, 0, d_z, d_y, d_pk, 77u); }, iters);
    const double vs = chain_us([&] { hipLaunchKernelGGL(k_inv_split, dim3(npr * 8), dim3(256), M, 0, d_z, d_y, d_pk, d_diag); }, iters);
    const uint32_t quads = 2 * M / 4;
    const double cf = chain_us([&] { hipLaunchKernelGGL((k_crt_loads<5, false>), dim3((quads + 255) / 256), dim3(256), 0, 0, d_y, d_dig, d_pk, quads); }, iters);
    const double cs = chain_us([&] { hipLaunchKernelGGL((k_crt_loads<5, true>), dim3((quads + 255) / 256), dim3(256), 0, 0, d_y, d_dig, d_pk, quads); }, iters);
    // ---- where the in-situ launches lose their time: the engine's k_fwd_phase / k_inv_column take 11.3 /
    // 9.3 us inside the k-loop (rocprofv3) where the same code takes 4-6 us in the chains above.  The
    // chains above repeat ONE kernel on ONE key slice; the k-loop alternates three kernels and walks
    // through a 1.34 GB key.  Mixed chains: (a) the three kernels alternating on one key slice, (b) the
    // same with the key slice advancing by its 1.3 MB every iteration through a 1.34 GB buffer.
    int32_t *d_bigkey = nullptr;
    const size_t slice = (size_t)npr * 8 * M;             // int32 elements per key slice
    const size_t nslices = 1024;
    double mix_hot = 0, mix_cold = 0;
    if (hipMalloc(&d_bigkey, nslices * slice * 4) == hipSuccess) {
        (void)hipMemset(d_bigkey, 3, nslices * slice * 4);
        size_t kk = 0;
        auto iter_hot = [&] {
            hipLaunchKernelGGL(k_fwd_full, dim3(npr * 4), dim3(1024), 4 * M, 0, d_dig, d_key, d_z, d_pk);
            hipLaunchKernelGGL(k_inv_full, dim3(npr * 2), dim3(1024), 4 * M, 0, d_z, d_y, d_pk, 77u);
            hipLaunchKernelGGL((k_crt_loads<5, false>), dim3((quads + 255) / 256), dim3(256), 0, 0, d_y, d_dig, d_pk, quads);
        };
        auto iter_cold = [&] {
            hipLaunchKernelGGL(k_fwd_full, dim3(npr * 4), dim3(1024), 4 * M, 0, d_dig, d_bigkey + (kk++ % nslices) * slice, d_z, d_pk);
            hipLaunchKernelGGL(k_inv_full, dim3(npr * 2), dim3(1024), 4 * M, 0, d_z, d_y, d_pk, 77u);
            hipLaunchKernelGGL((k_crt_loads<5, false>), dim3((quads + 255) / 256), dim3(256), 0, 0, d_y, d_dig, d_pk, quads);
        };
        mix_hot = chain_us(iter_hot, 1024);
        mix_cold = chain_us(iter_cold, 1024);
        (void)hipFree(d_bigkey);
    }
    printf("dependent chains of %d launches, one gate, five primes, m = 8192 (us per launch)\n", iters);
    printf("forward : full (20 x 1024 threads) %.2f   split (80 x 256 threads, 2048 points each) %.2f   saves %.2f\n", ff, fs, ff - fs);
    printf("inverse : full (10 x 1024 threads) %.2f   split (40 x 256 threads + diagonal product) %.2f   saves %.2f\n", vf, vs, vf - vs);
    printf("CRT side: one residue per prime %.2f   four partial residues + radix-4 combination %.2f   costs %.2f\n", cf, cs, cs - cf);
    printf("per k-loop iteration: %.2f us saved of the three launches; x 1024 iterations = %.2f ms of a 24-25 ms call\n",
           (ff - fs) + (vf - vs) - (cs - cf), ((ff - fs) + (vf - vs) - (cs - cf)) * 1.024);
    printf("mixed chain (forward, inverse, CRT alternating), us per iteration: one key slice %.2f (sum of the three "
           "chains above %.2f), key slice advancing through 1.34 GB %.2f\n", mix_hot, ff + vf + cf, mix_cold);
    return 0;
}
