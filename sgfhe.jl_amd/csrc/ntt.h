// LDS-staged negacyclic NTT of length m = 2^LOGM over one 30-bit RNS prime, NP polynomials at
// once (same twiddles, NP-fold instruction-level parallelism).
//
// Geometry: a workgroup of T = m / 8 threads; every thread keeps 8 points of each polynomial in
// registers and performs radix-8 passes (three butterfly stages per pass) on them.  Between
// passes the points are exchanged through LDS.  A pass over index bits [S, S+3) gives thread
// (hi, lo) = (tid >> S, tid & (2^S - 1)) the points  idx(e) = hi << (S+3) | e << S | lo.
// When LOGM is not a multiple of 3 the top LOGM % 3 bits are handled by a partial pass in the
// (hi = 0) layout S = LOGM - 3, which is also the coalesced global-memory layout
// idx(e) = tid + T * e.
//
// Forward: Cooley-Tukey with the psi twist merged into the twiddles (natural order in, slot
// order = bit-reversed evaluation order out); on exit thread tid holds slots 8 tid .. 8 tid + 7.
// Inverse: Gentleman-Sande, slot order in (same ownership), natural order out in the
// idx(e) = tid + T * e layout, not scaled by 1/m (the scale is folded into the key).
//
// LDS addressing: word address = poly * m + swz(idx), where swz XORs the five bank bits with
// index bits 5..7 such that the b32 accesses of every pass (S = 0, 3, 6, 9 and S >= 5 in
// general) are bank-conflict free within each 32-lane group:
//   bank bits (b0..b4) = (a0^a6, a1^a7, a2^a5, a3^a6, a4^a7).
#pragma once

#include "rns_arith.h"

namespace sgfhe {

__host__ __device__ constexpr uint32_t swz_bits(uint32_t idx) {
    return (((idx >> 6) & 1u) * 0x09u) ^ (((idx >> 7) & 1u) * 0x12u) ^ (((idx >> 5) & 1u) * 0x04u);
}
__host__ __device__ constexpr uint32_t swz(uint32_t idx) { return idx ^ swz_bits(idx); }

template <int LOGM>
struct NttGeom {
    static constexpr int M = 1 << LOGM;
    static constexpr int T = M / 8;
    static constexpr int RHO = LOGM % 3;
    static constexpr int STOP = LOGM - 3;                       // layout of the global order
    static constexpr int SFIRST = RHO ? LOGM - RHO - 3 : LOGM - 6;  // first LDS pass (forward)
};

// ---- butterfly stages on the 3 local index bits of e ------------------------------------

template <int NP>
__device__ __forceinline__ void fwd_bit2(uint32_t (&x)[NP][8], uint2 w, uint32_t p, uint32_t p2) {
#pragma unroll
    for (int q = 0; q < NP; q++)
#pragma unroll
        for (int e = 0; e < 4; e++) bfly_fwd(x[q][e], x[q][e + 4], w.x, w.y, p, p2);
}
template <int NP>
__device__ __forceinline__ void fwd_bit1(uint32_t (&x)[NP][8], uint2 wa, uint2 wb, uint32_t p,
                                         uint32_t p2) {
#pragma unroll
    for (int q = 0; q < NP; q++) {
        bfly_fwd(x[q][0], x[q][2], wa.x, wa.y, p, p2);
        bfly_fwd(x[q][1], x[q][3], wa.x, wa.y, p, p2);
        bfly_fwd(x[q][4], x[q][6], wb.x, wb.y, p, p2);
        bfly_fwd(x[q][5], x[q][7], wb.x, wb.y, p, p2);
    }
}
template <int NP>
__device__ __forceinline__ void fwd_bit0(uint32_t (&x)[NP][8], const uint2 (&w)[4], uint32_t p,
                                         uint32_t p2) {
#pragma unroll
    for (int q = 0; q < NP; q++) {
#pragma unroll
        for (int h = 0; h < 4; h++) bfly_fwd(x[q][2 * h], x[q][2 * h + 1], w[h].x, w[h].y, p, p2);
    }
}
template <int NP>
__device__ __forceinline__ void inv_bit2(uint32_t (&x)[NP][8], uint2 w, uint32_t p, uint32_t p2) {
#pragma unroll
    for (int q = 0; q < NP; q++)
#pragma unroll
        for (int e = 0; e < 4; e++) bfly_inv(x[q][e], x[q][e + 4], w.x, w.y, p, p2);
}
template <int NP>
__device__ __forceinline__ void inv_bit1(uint32_t (&x)[NP][8], uint2 wa, uint2 wb, uint32_t p,
                                         uint32_t p2) {
#pragma unroll
    for (int q = 0; q < NP; q++) {
        bfly_inv(x[q][0], x[q][2], wa.x, wa.y, p, p2);
        bfly_inv(x[q][1], x[q][3], wa.x, wa.y, p, p2);
        bfly_inv(x[q][4], x[q][6], wb.x, wb.y, p, p2);
        bfly_inv(x[q][5], x[q][7], wb.x, wb.y, p, p2);
    }
}
template <int NP>
__device__ __forceinline__ void inv_bit0(uint32_t (&x)[NP][8], const uint2 (&w)[4], uint32_t p,
                                         uint32_t p2) {
#pragma unroll
    for (int q = 0; q < NP; q++) {
#pragma unroll
        for (int h = 0; h < 4; h++) bfly_inv(x[q][2 * h], x[q][2 * h + 1], w[h].x, w[h].y, p, p2);
    }
}

// ---- LDS exchange ------------------------------------------------------------------------

template <int LOGM, int S>
__device__ __forceinline__ uint32_t lds_base(int tid) {
    uint32_t lo = (uint32_t)tid & ((1u << S) - 1u);
    uint32_t hi = (uint32_t)tid >> S;
    return swz((hi << (S + 3)) | lo);
}

template <int LOGM, int NP, int S>
__device__ __forceinline__ void lds_store(const uint32_t (&x)[NP][8], uint32_t *lds, int tid) {
    constexpr int M = 1 << LOGM;
    const uint32_t pb = lds_base<LOGM, S>(tid);
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const uint32_t a = pb ^ swz((uint32_t)e << S);
#pragma unroll
        for (int q = 0; q < NP; q++) lds[q * M + a] = x[q][e];
    }
}
template <int LOGM, int NP, int S>
__device__ __forceinline__ void lds_load(uint32_t (&x)[NP][8], const uint32_t *lds, int tid) {
    constexpr int M = 1 << LOGM;
    const uint32_t pb = lds_base<LOGM, S>(tid);
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const uint32_t a = pb ^ swz((uint32_t)e << S);
#pragma unroll
        for (int q = 0; q < NP; q++) x[q][e] = lds[q * M + a];
    }
}

// ---- twiddle tables -------------------------------------------------------------------------
// tw[i] = (w, floor(w 2^32 / p)) with w = psi^bitrev(i) (forward) or psi^-bitrev(i) (inverse),
// i in [1, m): the tables of the merged-twist CT / GS transforms.  A pass over bits [S, S+3)
// of thread-group `hi` needs entries  2^(LOGM-3-S) + hi,  2^(LOGM-2-S) + 2 hi + {0,1},
// 2^(LOGM-1-S) + 4 hi + {0..3}: 1 + 2 + 4 consecutive entries.  (All loads keep the uint2 element
// type: a uint4-typed view of the same table made hipcc -O3's load/store vectorizer emit a
// partial load of one entry for LOGM = 11.)

template <int LOGM, int NP, int S>
__device__ __forceinline__ void fwd_pass_full(uint32_t (&x)[NP][8], const uint2 *tw, int tid,
                                              uint32_t p, uint32_t p2) {
    const uint32_t hi = (uint32_t)tid >> S;
    const uint2 *t1 = tw + (1u << (LOGM - 2 - S)) + 2 * hi;
    const uint2 *t0 = tw + (1u << (LOGM - 1 - S)) + 4 * hi;
    const uint2 w2 = tw[(1u << (LOGM - 3 - S)) + hi];
    const uint2 w1a = t1[0], w1b = t1[1];
    const uint2 w0[4] = {t0[0], t0[1], t0[2], t0[3]};
    fwd_bit2<NP>(x, w2, p, p2);
    fwd_bit1<NP>(x, w1a, w1b, p, p2);
    fwd_bit0<NP>(x, w0, p, p2);
}
template <int LOGM, int NP, int S>
__device__ __forceinline__ void inv_pass_full(uint32_t (&x)[NP][8], const uint2 *tw, int tid,
                                              uint32_t p, uint32_t p2) {
    const uint32_t hi = (uint32_t)tid >> S;
    const uint2 *t1 = tw + (1u << (LOGM - 2 - S)) + 2 * hi;
    const uint2 *t0 = tw + (1u << (LOGM - 1 - S)) + 4 * hi;
    const uint2 w0[4] = {t0[0], t0[1], t0[2], t0[3]};
    const uint2 w1a = t1[0], w1b = t1[1];
    const uint2 w2 = tw[(1u << (LOGM - 3 - S)) + hi];
    inv_bit0<NP>(x, w0, p, p2);
    inv_bit1<NP>(x, w1a, w1b, p, p2);
    inv_bit2<NP>(x, w2, p, p2);
}

// Recursion over the LDS passes S = SCUR, SCUR - 3, ..., 0 (forward).
template <int LOGM, int NP, int SPREV, int SCUR>
struct FwdPasses {
    static __device__ __forceinline__ void run(uint32_t (&x)[NP][8], uint32_t *lds,
                                               const uint2 *tw, int tid, uint32_t p, uint32_t p2) {
        lds_store<LOGM, NP, SPREV>(x, lds, tid);
        __syncthreads();
        lds_load<LOGM, NP, SCUR>(x, lds, tid);
        fwd_pass_full<LOGM, NP, SCUR>(x, tw, tid, p, p2);
        if constexpr (SCUR >= 3) FwdPasses<LOGM, NP, SCUR, SCUR - 3>::run(x, lds, tw, tid, p, p2);
    }
};
// Inverse: passes S = SCUR, SCUR + 3, ... up to SLAST (inclusive), data arrives in registers in
// layout SCUR.
template <int LOGM, int NP, int SCUR, int SLAST>
struct InvPasses {
    static __device__ __forceinline__ void run(uint32_t (&x)[NP][8], uint32_t *lds,
                                               const uint2 *tw, int tid, uint32_t p, uint32_t p2) {
        inv_pass_full<LOGM, NP, SCUR>(x, tw, tid, p, p2);
        if constexpr (SCUR < SLAST) {
            lds_store<LOGM, NP, SCUR>(x, lds, tid);
            __syncthreads();
            lds_load<LOGM, NP, SCUR + 3>(x, lds, tid);
            InvPasses<LOGM, NP, SCUR + 3, SLAST>::run(x, lds, tw, tid, p, p2);
        }
    }
};

// Forward transform.  In: x[q][e] = coefficient tid + T e of polynomial q, in [0, 4p).
// Out: x[q][e] = slot 8 tid + e, in [0, 4p).  `lds` must hold NP * m words.
template <int LOGM, int NP>
__device__ __forceinline__ void ntt_forward(uint32_t (&x)[NP][8], uint32_t *lds, const uint2 *tw,
                                            int tid, uint32_t p) {
    using G = NttGeom<LOGM>;
    const uint32_t p2 = 2 * p;
    if constexpr (G::RHO == 0) {
        fwd_pass_full<LOGM, NP, G::STOP>(x, tw, tid, p, p2);
    } else {
        fwd_bit2<NP>(x, tw[1], p, p2);
        if constexpr (G::RHO == 2) fwd_bit1<NP>(x, tw[2], tw[3], p, p2);
    }
    if constexpr (G::SFIRST >= 0)
        FwdPasses<LOGM, NP, G::STOP, G::SFIRST>::run(x, lds, tw, tid, p, p2);
}

// Inverse transform (unscaled).  In: slots 8 tid + e in [0, 2p).  Out: coefficient tid + T e in
// [0, 2p).  On return every thread has its output both in registers and NOT in LDS.
template <int LOGM, int NP>
__device__ __forceinline__ void ntt_inverse(uint32_t (&x)[NP][8], uint32_t *lds, const uint2 *tw,
                                            int tid, uint32_t p) {
    using G = NttGeom<LOGM>;
    const uint32_t p2 = 2 * p;
    if constexpr (G::RHO == 0) {
        InvPasses<LOGM, NP, 0, G::STOP>::run(x, lds, tw, tid, p, p2);
    } else {
        constexpr int SLAST = LOGM - G::RHO - 3;
        InvPasses<LOGM, NP, 0, SLAST>::run(x, lds, tw, tid, p, p2);
        lds_store<LOGM, NP, SLAST>(x, lds, tid);
        __syncthreads();
        lds_load<LOGM, NP, G::STOP>(x, lds, tid);
        if constexpr (G::RHO == 2) inv_bit1<NP>(x, tw[2], tw[3], p, p2);
        inv_bit2<NP>(x, tw[1], p, p2);
    }
}

}  // namespace sgfhe
