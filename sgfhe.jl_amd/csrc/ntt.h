// LDS-staged negacyclic NTT of length m = 2^LOGM over one 29-bit RNS prime, NP polynomials at
// once (same twiddles, NP-fold instruction-level parallelism).  Values are signed lazy residues
// (rns_arith.h): the forward transform range-reduces the X inputs of the first stage of every LDS
// pass, the inverse transform the sums of every second stage and of the last stage of a pass.
//
// Geometry: a workgroup of T = m / E threads, E = 2^LOGE (8 or 16); every thread keeps E points
// of each polynomial in registers and performs radix-E passes (LOGE butterfly stages per pass)
// on them.  Between passes the points are exchanged through LDS.  A pass over index bits
// [S, S+LOGE) gives thread (hi, lo) = (tid >> S, tid & (2^S - 1)) the points
//     idx(e) = hi << (S+LOGE) | e << S | lo.
// When LOGM is not a multiple of LOGE the top LOGM % LOGE bits are handled by a partial pass in
// the (hi = 0) layout S = LOGM - LOGE, which is also the coalesced global-memory layout
// idx(e) = tid + T * e.
//
// Forward: Cooley-Tukey with the psi twist merged into the twiddles (natural order in, slot
// order = bit-reversed evaluation order out); on exit thread tid holds slots E tid .. E tid + E-1.
// Inverse: Gentleman-Sande, slot order in (same ownership), natural order out in the
// idx(e) = tid + T * e layout, not scaled by 1/m (the scale is folded into the key).
//
// Twiddles: tw[i] = psi^(+-bitrev(i)) * 2^32 mod p (Montgomery form, centred: |tw| <= p/2), i in [1, m).  Stage
// "local bit b" of a pass over [S, S+LOGE) uses the 2^(LOGE-1-b) consecutive entries starting at
// 2^(LOGM-1-S-b) + (hi << (LOGE-1-b)).
//
// LDS addressing: word address = poly * m + swz(idx); swz XORs the five bank bits with higher
// index bits such that the b32 accesses of every pass are bank-conflict free within each
// 32-lane group (checked exhaustively in tests/test_rns_model.py):
//   E = 8 :  (b0..b4) ^= (a6, a7, a5, a6, a7)
//   E = 16:  (b0..b4) ^= (a5, a6, a7, a8, a8)
#pragma once

#include "rns_arith.h"

#ifdef SGFHE_ABL_NO_BARRIER
#define SGFHE_SYNC() ((void)0)  // timing-only build
#else
#define SGFHE_SYNC() __syncthreads()
#endif

#ifndef SGFHE_FWD_RADIX4
#define SGFHE_FWD_RADIX4 1
#endif
#ifndef SGFHE_INV_RADIX4
#define SGFHE_INV_RADIX4 1
#endif
// Forward passes with per-lane twiddles take the radix-4 form (two more vector registers) only where
// k_extprod keeps its registers without spilling: hipcc spills 16-20 bytes at m = 4096 and 16384.
#ifndef SGFHE_FWD_VEC4
#define SGFHE_FWD_VEC4(LOGM) ((LOGM) != 12 && (LOGM) != 14)
#endif
// The inverse keeps its radix-2 form (with the searched reduction pattern and the un-reduced entry
// of column 0) at m = 4096, where the radix-4 steps cost k_extprod 20-24 bytes of scratch.
#ifndef SGFHE_INV_R4
#define SGFHE_INV_R4(LOGM) ((LOGM) != 12)
#endif

namespace sgfhe {

template <int LOGE>
__host__ __device__ constexpr uint32_t swz_bits(uint32_t idx) {
    if constexpr (LOGE == 3)
        return (((idx >> 6) & 1u) * 0x09u) ^ (((idx >> 7) & 1u) * 0x12u) ^ (((idx >> 5) & 1u) * 0x04u);
    else
        return (((idx >> 5) & 1u) * 0x01u) ^ (((idx >> 6) & 1u) * 0x02u) ^
               (((idx >> 7) & 1u) * 0x04u) ^ (((idx >> 8) & 1u) * 0x18u);
}
template <int LOGE>
__host__ __device__ constexpr uint32_t swz(uint32_t idx) { return idx ^ swz_bits<LOGE>(idx); }

template <int LOGM, int LOGE>
struct NttGeom {
    static constexpr int M = 1 << LOGM;
    static constexpr int E = 1 << LOGE;
    static constexpr int T = M / E;
    static constexpr int RHO = LOGM % LOGE;
    static constexpr int STOP = LOGM - LOGE;                           // layout of the global order
    static constexpr int SFIRST = RHO ? LOGM - RHO - LOGE : LOGM - 2 * LOGE;  // first LDS pass
    static constexpr int SLAST_INV = RHO ? LOGM - RHO - LOGE : STOP;   // last full inverse pass
};

// A zero the compiler cannot see through.  Added to the twiddle pointer once per pass so that the
// (read-only, hence freely hoistable) twiddle loads stay one pass ahead of their use and no
// further: hoisting all ~46 loads of an NTT to its top costs that many registers.
__device__ __forceinline__ uint32_t opaque_zero() {
    uint32_t z = 0;
    asm volatile("" : "+v"(z));
    return z;
}
// The same in a scalar register, for addresses that are uniform over the wavefront.
__device__ __forceinline__ uint32_t opaque_zero_s() {
    uint32_t z = 0;
    asm volatile("" : "+s"(z));
    return z;
}

typedef __attribute__((address_space(4))) int32_t gmem_i32;  // a word of read-only global memory ("constant")

// ---- twiddles of one pass ---------------------------------------------------------------------
// The E - 1 twiddles of a radix-E pass sit in registers in heap order: stage "local bit B" owns
// entries [NG - 1, 2 NG - 1), NG = 2^(LOGE-1-B).  They are loaded (from L1/L2) BEFORE the LDS
// exchange that precedes the pass, so the load latency overlaps the exchange and its barrier.

// Pass S reads the table at an index built from tid >> S: for 2^S >= 64 that is the same for
// every lane of a wavefront, so the loads are scalar (s_load into SGPRs: no VGPRs, no VMEM
// traffic, no address arithmetic on the vector ALU).
template <int LOGM, int LOGE, int S, int BHI, int BLO>
__device__ __forceinline__ void load_twiddles_at(int32_t (&t)[(1 << LOGE) - 1], const gmem_i32 *tw,
                                                 uint32_t hi) {
    constexpr int NG = 1 << (LOGE - 1 - BHI);
    const gmem_i32 *w = tw + (1u << (LOGM - 1 - S - BHI)) + (hi << (LOGE - 1 - BHI));
#pragma unroll
    for (int g = 0; g < NG; g++)
#ifdef SGFHE_ABL_NO_TW
        t[NG - 1 + g] = (int32_t)(hi + g + 12345u);  // timing-only build: no twiddle loads (wrong results)
#else
        t[NG - 1 + g] = w[g];
#endif
    if constexpr (BHI > BLO) load_twiddles_at<LOGM, LOGE, S, BHI - 1, BLO>(t, tw, hi);
}
template <int LOGM, int LOGE, int S, int BHI, int BLO>
__device__ __forceinline__ void load_twiddles(int32_t (&t)[(1 << LOGE) - 1], const int32_t *tw,
                                              uint32_t hi) {
    // The table pointer comes out of a PrimeK record in memory, so the compiler only knows it as
    // a generic pointer and would emit flat_load, which counts on lgkmcnt as well as vmcnt: every
    // wait for an LDS exchange would then also wait for the twiddles prefetched across it.
    // The opaque zero keeps the loads one pass ahead of their use and no further.
    if constexpr ((1 << S) >= 64) {
        const uint32_t hu = __builtin_amdgcn_readfirstlane(hi);
        load_twiddles_at<LOGM, LOGE, S, BHI, BLO>(t, (const gmem_i32 *)tw + opaque_zero_s(), hu);
    } else {
        load_twiddles_at<LOGM, LOGE, S, BHI, BLO>(t, (const gmem_i32 *)tw, hi);
    }
}

// ---- one butterfly stage on local bit B of the register index e -----------------------------

template <int NP, int LOGE, int B>
__device__ __forceinline__ void fwd_stage(int32_t (&x)[NP][1 << LOGE],
                                          const int32_t (&t)[(1 << LOGE) - 1], const Mod &md) {
    constexpr int NG = 1 << (LOGE - 1 - B);
#pragma unroll
    for (int q = 0; q < NP; q++)
#pragma unroll
        for (int g = 0; g < NG; g++)
#pragma unroll
            for (int l = 0; l < (1 << B); l++) {
                const int e0 = (g << (B + 1)) | l;
                bfly_fwd(x[q][e0], x[q][e0 | (1 << B)], t[NG - 1 + g], md);
            }
}
// Two forward stages (local bits BH and BH - 1) as one radix-4 step with the second stage's
// products deferred: with X0..X3 = x[e0], x[e0 | lo], x[e0 | hi], x[e0 | hi | lo] and the twiddles
// wA (stage BH), wB0 / wB1 (stage BH - 1, the two children of wA in the table's binary tree),
//   u = wA X2,  a = X0 + u,  b = X0 - u,
//   s = wB0 X1 + (wA wB0) X3,   r = wB1 X1 - (wA wB1) X3      (one Montgomery reduction each: the two
//                                                               64-bit products are summed first)
//   outputs a + s, a - s, b + r, b - r
// -- 17 instructions where four radix-2 butterflies take 20.  tp holds the product twiddles
// wA wB0 and -(wA wB1) (table tw + 2 m, same indexing as tw).  Any int32 X1..X3; the sums stay
// below 2^60.  |u| <= |X2| / 16 + p / 2,  |s|, |r| <= (|X1| + |X3|) / 16 + p / 2.
template <int NP, int LOGE, int BH>
__device__ __forceinline__ void fwd_step4(int32_t (&x)[NP][1 << LOGE], const int32_t (&t)[(1 << LOGE) - 1],
                                          const int32_t (&tp)[(1 << LOGE) - 1], const Mod &md) {
    static_assert(BH >= 1, "a radix-4 step covers local bits BH and BH - 1");
    constexpr int NGA = 1 << (LOGE - 1 - BH);
    constexpr int LO = 1 << (BH - 1), HI = 1 << BH;
#pragma unroll
    for (int q = 0; q < NP; q++)
#pragma unroll
        for (int g = 0; g < NGA; g++)
#pragma unroll
            for (int l = 0; l < LO; l++) {
                const int e0 = (g << (BH + 1)) | l;
                const int32_t wA = t[NGA - 1 + g];
                const int32_t wB0 = t[2 * NGA - 1 + 2 * g], wB1 = t[2 * NGA - 1 + 2 * g + 1];
                const int32_t P0 = tp[2 * NGA - 1 + 2 * g], P1 = tp[2 * NGA - 1 + 2 * g + 1];
                const int32_t X0 = x[q][e0], X1 = x[q][e0 | LO], X2 = x[q][e0 | HI], X3 = x[q][e0 | HI | LO];
                const int32_t u = smont(X2, wA, md);
                const int32_t a = X0 + u, b = X0 - u;
                const int32_t s_ = sredc((int64_t)X1 * wB0 + (int64_t)X3 * P0, md);
                const int32_t r_ = sredc((int64_t)X1 * wB1 + (int64_t)X3 * P1, md);
                x[q][e0] = a + s_;
                x[q][e0 | LO] = a - s_;
                x[q][e0 | HI] = b + r_;
                x[q][e0 | HI | LO] = b - r_;
            }
}
// registers whose index has the local bits BH and BH - 1 clear: the X0 inputs of fwd_step4<BH>
template <int NP, int LOGE, int BH>
__device__ __forceinline__ void fwd_reduce_x0(int32_t (&x)[NP][1 << LOGE], const Mod &md) {
#pragma unroll
    for (int q = 0; q < NP; q++)
#pragma unroll
        for (int e = 0; e < (1 << LOGE); e++)
            if ((e & (3 << (BH - 1))) == 0) x[q][e] = sred_floor(x[q][e], md);
}

// Two inverse (Gentleman-Sande) stages, local bits B then B + 1, as one radix-4 step.  With
// x0..x3 = x[e0], x[e0 | lo], x[e0 | hi], x[e0 | hi | lo], the stage-B twiddles wB0 / wB1 and the
// stage-(B+1) twiddle wA (their parent in the table's tree):
//   s0 = x0 + x1, s1 = x2 + x3, d0 = x0 - x1, d1 = x2 - x3
//   y0 = s0 + s1                       -> x[e0]            (a sum of four: reduced by the caller)
//   y2 = wA (s0 - s1)                  -> x[e0 | hi]
//   y1 = wB0 d0 + wB1 d1               -> x[e0 | lo]       (two 64-bit products summed, one reduction)
//   y3 = (wA wB0) d0 - (wA wB1) d1     -> x[e0 | hi | lo]  (product twiddles from the table tw + 2 m)
// 17 instructions where four radix-2 butterflies take 20, and no intermediate sum needs a range
// reduction.  Inputs below 0.875 * 2^29 keep y0 inside `sred`'s precondition (3.5 * 2^29).
template <int NP, int LOGE, int B>
__device__ __forceinline__ void inv_step4(int32_t (&x)[NP][1 << LOGE], const int32_t (&t)[(1 << LOGE) - 1],
                                          const int32_t (&tp)[(1 << LOGE) - 1], const Mod &md) {
    constexpr int NGA = 1 << (LOGE - 2 - B);          // groups of stage B + 1
    constexpr int LO = 1 << B, HI = 1 << (B + 1);
#pragma unroll
    for (int q = 0; q < NP; q++)
#pragma unroll
        for (int g = 0; g < NGA; g++)
#pragma unroll
            for (int l = 0; l < LO; l++) {
                const int e0 = (g << (B + 2)) | l;
                const int32_t wA = t[NGA - 1 + g];
                const int32_t wB0 = t[2 * NGA - 1 + 2 * g], wB1 = t[2 * NGA - 1 + 2 * g + 1];
                const int32_t P0 = tp[2 * NGA - 1 + 2 * g], P1 = tp[2 * NGA - 1 + 2 * g + 1];
                const int32_t x0 = x[q][e0], x1 = x[q][e0 | LO], x2 = x[q][e0 | HI], x3 = x[q][e0 | HI | LO];
                const int32_t s0 = x0 + x1, s1 = x2 + x3, d0 = x0 - x1, d1 = x2 - x3;
                x[q][e0] = s0 + s1;
                x[q][e0 | HI] = smont(s0 - s1, wA, md);
                x[q][e0 | LO] = sredc((int64_t)d0 * wB0 + (int64_t)d1 * wB1, md);
                x[q][e0 | HI | LO] = sredc((int64_t)d0 * P0 + (int64_t)d1 * P1, md);
            }
}
// `sred` of the registers whose index has the local bits B and B + 1 clear: the y0 outputs of inv_step4<B>
template <int NP, int LOGE, int B>
__device__ __forceinline__ void inv_reduce_y0(int32_t (&x)[NP][1 << LOGE], const Mod &md) {
#pragma unroll
    for (int q = 0; q < NP; q++)
#pragma unroll
        for (int e = 0; e < (1 << LOGE); e++)
            if ((e & (3 << B)) == 0) x[q][e] = sred(x[q][e], md);
}

// REDMASK: bit e0 set = the sum X' of the butterfly whose X sits in register e0 is range-reduced
// (REDMASK0: the same for polynomial 0, which may arrive with a different bound)
template <int NP, int LOGE, int B, uint32_t REDMASK, uint32_t REDMASK0>
__device__ __forceinline__ void inv_stage(int32_t (&x)[NP][1 << LOGE],
                                          const int32_t (&t)[(1 << LOGE) - 1], const Mod &md) {
    constexpr int NG = 1 << (LOGE - 1 - B);
#pragma unroll
    for (int q = 0; q < NP; q++)
#pragma unroll
        for (int g = 0; g < NG; g++)
#pragma unroll
            for (int l = 0; l < (1 << B); l++) {
                const int e0 = (g << (B + 1)) | l;
                bfly_inv(x[q][e0], x[q][e0 | (1 << B)], t[NG - 1 + g], md,
                         (((q == 0 ? REDMASK0 : REDMASK) >> e0) & 1u) != 0);
            }
}

// stages B = BHI, BHI-1, ..., BLO (forward order)
template <int NP, int LOGE, int BHI, int BLO>
__device__ __forceinline__ void fwd_stages(int32_t (&x)[NP][1 << LOGE],
                                           const int32_t (&t)[(1 << LOGE) - 1], const Mod &md) {
    fwd_stage<NP, LOGE, BHI>(x, t, md);
    if constexpr (BHI > BLO) fwd_stages<NP, LOGE, BHI - 1, BLO>(x, t, md);
}
// The X inputs of the first stage of a full pass (register index bit LOGE-1 clear) are pulled back
// to [0, 2^29] (`sred_floor`: any int32 in, three instructions); the stages of the pass then add at
// most 0.73 * 2^29 each, so a radix-2 pass hands out |x| < 3.7 * 2^29; with the radix-4 steps of
// FwdPasses a transform hands out |x| < 3.95 * 2^29 (tests/rns_model.py RangeModel).
template <int NP, int LOGE>
__device__ __forceinline__ void fwd_reduce_x(int32_t (&x)[NP][1 << LOGE], const Mod &md) {
#pragma unroll
    for (int q = 0; q < NP; q++)
#pragma unroll
        for (int e = 0; e < (1 << (LOGE - 1)); e++) x[q][e] = sred_floor(x[q][e], md);
}
// stages B = BLO, BLO+1, ..., BHI (inverse order).  The sums of a Gentleman-Sande stage double in
// size while the twiddled differences come back below 0.75 * 2^29, so which sums need a range
// reduction depends on the history of the register: bit k of its index says whether it left
// stage k as a sum (0) or as a product (1).
//  * a full radix-16 pass (LOGE = 4, stages 0..3, inputs <= 0.75 * 2^29) reduces 14 of its 32 sums --
//    stage 1: the sums of sums (bit 0 clear); stage 2: the registers that were products in stage 0
//    and skipped in stage 1 (bit 0 set, bit 1 clear); stage 3: all -- and hands at most
//    0.68 * 2^29 to the next pass.  (Exhaustive search over the 2^15 reduction patterns:
//    14 is the fewest that avoids int32 overflow, keeps every `sred` input below 3.5 * 2^29
//    and returns to the input bound; tests/rns_model.py RangeModel re-derives the bounds.)
//  * the same pass as the LAST step of a transform (LASTRED = 2: its outputs only have to stay
//    below 1.4 * 2^29 for the epilogues) reduces 9: in stage 3 only registers 0, 2 and 3.
//  * the same pass as the FIRST step with inputs up to 1.5 * 2^29 (LASTRED = 3: k_extprod's
//    column 0 straight out of its Montgomery reduction, without the input `sred`) reduces 20:
//    stage 0 all, stage 1 none, stage 2 the registers with bit 1 clear, stage 3 all.
//  * any other run of stages reduces every sum of every second stage (counted from its first
//    stage BFIRST) and, when LASTRED, of its last stage.
template <int LOGE, int B, int BLO, int BHI, int BFIRST, int LASTRED>
constexpr uint32_t inv_red_mask() {
    if (LOGE == 4 && BFIRST == 0 && BHI == 3 && LASTRED) {
        uint32_t m = 0;
        for (int e0 = 0; e0 < 16; e0++) {
            if (e0 & (1 << B)) continue;  // not an X register of this stage
            bool red = false;
            if (LASTRED == 3)
                red = B == 0 || B == 3 || (B == 2 && (e0 & 2) == 0);
            else
                red = B == 1 ? (e0 & 1) == 0
                    : B == 2 ? (e0 & 3) == 1
                    : B == 3 ? (LASTRED == 2 ? (e0 == 0 || e0 == 2 || e0 == 3) : true) : false;
            if (red) m |= 1u << e0;
        }
        return m;
    }
    return ((((B - BFIRST) & 1) != 0) || (LASTRED && B == BHI)) ? 0xFFFFFFFFu : 0u;
}
template <int NP, int LOGE, int BLO, int BHI, int BFIRST, int LASTRED, int LASTRED0 = LASTRED>
__device__ __forceinline__ void inv_stages(int32_t (&x)[NP][1 << LOGE],
                                           const int32_t (&t)[(1 << LOGE) - 1], const Mod &md) {
    inv_stage<NP, LOGE, BLO, inv_red_mask<LOGE, BLO, BLO, BHI, BFIRST, LASTRED>(),
              inv_red_mask<LOGE, BLO, BLO, BHI, BFIRST, LASTRED0>()>(x, t, md);
    if constexpr (BLO < BHI) inv_stages<NP, LOGE, BLO + 1, BHI, BFIRST, LASTRED, LASTRED0>(x, t, md);
}

// ---- LDS exchange ------------------------------------------------------------------------

template <int LOGE, int S>
__device__ __forceinline__ uint32_t lds_base(int tid) {
    const uint32_t lo = (uint32_t)tid & ((1u << S) - 1u);
    const uint32_t hi = (uint32_t)tid >> S;
    return swz<LOGE>((hi << (S + LOGE)) | lo);
}
// Addresses are formed in bytes: the thread's swizzled base is shifted once and every element
// costs one XOR with a compile-time constant (an index-then-scale form costs a shift per element).
template <int LOGM, int NP, int LOGE, int S>
__device__ __forceinline__ void lds_store(const int32_t (&x)[NP][1 << LOGE], uint32_t *lds, int tid) {
#ifdef SGFHE_ABL_NO_LDS
    return;  // timing-only build: no LDS exchange (wrong results)
#endif
    constexpr int M = 1 << LOGM;
    const uint32_t pb = lds_base<LOGE, S>(tid) << 2;
    char *const base = reinterpret_cast<char *>(lds);
#pragma unroll
    for (int e = 0; e < (1 << LOGE); e++) {
        const uint32_t a = pb ^ (swz<LOGE>((uint32_t)e << S) << 2);
#pragma unroll
        for (int q = 0; q < NP; q++) *reinterpret_cast<int32_t *>(base + q * M * 4 + a) = x[q][e];
    }
}
template <int LOGM, int NP, int LOGE, int S>
__device__ __forceinline__ void lds_load(int32_t (&x)[NP][1 << LOGE], const uint32_t *lds, int tid) {
#ifdef SGFHE_ABL_NO_LDS
    return;
#endif
    constexpr int M = 1 << LOGM;
    const uint32_t pb = lds_base<LOGE, S>(tid) << 2;
    const char *const base = reinterpret_cast<const char *>(lds);
#pragma unroll
    for (int e = 0; e < (1 << LOGE); e++) {
        const uint32_t a = pb ^ (swz<LOGE>((uint32_t)e << S) << 2);
#pragma unroll
        for (int q = 0; q < NP; q++) x[q][e] = *reinterpret_cast<const int32_t *>(base + q * M * 4 + a);
    }
}

// ---- pass recursions ---------------------------------------------------------------------------

// Synchronisation of the exchange between the layouts S = SLOW + LOGE and S = SLOW.  The exchange
// between S = LOGE and S = 0 moves data only inside aligned groups of 2^LOGE consecutive threads
// (thread (hi, lo) <-> thread (hi, e)), i.e. inside one wavefront: the LDS queue of a wave is
// in order, so no workgroup barrier is needed, only that the compiler keeps the stores ahead of
// the loads (they may alias, so it does).
template <int LOGE, int SLOW>
__device__ __forceinline__ void exchange_sync() {
#ifdef SGFHE_ABL_ONEBAR  // timing-only build: the S = LOGE <-> 2 LOGE exchange without a barrier (wrong results)
    if constexpr ((SLOW == 0 || SLOW == LOGE) && (1 << LOGE) <= 64) {
#else
    if constexpr (SLOW == 0 && (1 << LOGE) <= 64) {
#endif
#ifndef SGFHE_ABL_NO_BARRIER
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#endif
    } else {
        SGFHE_SYNC();
    }
}

// forward LDS passes S = SCUR, SCUR - LOGE, ..., 0; data arrives in registers in layout SPREV
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};
// `before_last` runs just before the exchange into the last pass: the caller's place to issue
// loads it needs right after the transform, so that their latency overlaps that pass.
template <int LOGM, int NP, int LOGE, int SPREV, int SCUR>
struct FwdPasses {
    template <class F>
    static __device__ __forceinline__ void run(int32_t (&x)[NP][1 << LOGE], uint32_t *lds,
                                               const int32_t *tw, int tid, const Mod &md,
                                               const F &before_last) {
        int32_t t[(1 << LOGE) - 1], tp[(1 << LOGE) - 1];
        load_twiddles<LOGM, LOGE, SCUR, LOGE - 1, 0>(t, tw, (uint32_t)tid >> SCUR);
        // Product twiddles of the radix-4 steps (table tw + 2 m).  A pass whose twiddles are
        // wave-uniform keeps them in scalar registers and runs both of its step pairs as radix-4
        // (LOGE = 4: ten products); a pass with per-lane twiddles only its first pair (two more
        // vector registers).
        constexpr bool UNIFORM = (1 << SCUR) >= 64;
        constexpr bool TWO_STEPS = SGFHE_FWD_RADIX4 && LOGE == 4 && UNIFORM;
        if constexpr (SGFHE_FWD_RADIX4 && (UNIFORM || SGFHE_FWD_VEC4(LOGM))) {
            load_twiddles<LOGM, LOGE, SCUR, LOGE - 2, LOGE - 2>(tp, tw + (2 << LOGM), (uint32_t)tid >> SCUR);
            if constexpr (TWO_STEPS) load_twiddles<LOGM, LOGE, SCUR, 0, 0>(tp, tw + (2 << LOGM), (uint32_t)tid >> SCUR);
        }
        if constexpr (SCUR < LOGE) before_last();
        lds_store<LOGM, NP, LOGE, SPREV>(x, lds, tid);
        exchange_sync<LOGE, SCUR>();
        lds_load<LOGM, NP, LOGE, SCUR>(x, lds, tid);
        // -p re-enters every pass through an opaque scalar move: hipcc otherwise hoists its 64-bit
        // sign extension out of the caller's loop, and instruction selection (per basic block) then
        // no longer sees the 32 x 32 -> 64 multiply-add of `sredc` and emulates a 64 x 64 one.
        Mod mdl = md;
        mdl.negp += (int32_t)opaque_zero_s();
        if constexpr (TWO_STEPS) {
            // X0 inputs of each step pulled back to [0, 2^29]: every value of the pass stays below
            // 2.5 * 2^29 (tests/rns_model.py RangeModel)
            fwd_reduce_x0<NP, LOGE, 3>(x, mdl);
            fwd_step4<NP, LOGE, 3>(x, t, tp, mdl);
            fwd_reduce_x0<NP, LOGE, 1>(x, mdl);
            fwd_step4<NP, LOGE, 1>(x, t, tp, mdl);
        } else if constexpr (SGFHE_FWD_RADIX4 && (UNIFORM || SGFHE_FWD_VEC4(LOGM))) {
            fwd_reduce_x<NP, LOGE>(x, mdl);
            fwd_step4<NP, LOGE, LOGE - 1>(x, t, tp, mdl);
            fwd_stages<NP, LOGE, LOGE - 3, 0>(x, t, mdl);
        } else {
            fwd_reduce_x<NP, LOGE>(x, mdl);
            fwd_stages<NP, LOGE, LOGE - 1, 0>(x, t, mdl);
        }
        if constexpr (SCUR >= LOGE)
            FwdPasses<LOGM, NP, LOGE, SCUR, SCUR - LOGE>::run(x, lds, tw, tid, md, before_last);
    }
};
// inverse passes S = SCUR, SCUR + LOGE, ..., SLAST; data arrives in registers in layout SCUR and
// the twiddles of pass SCUR in t
// FINAL: the pass S = SLAST is the last step of the transform (no partial pass follows);
// WIDE0: polynomial 0 enters the first pass with |x| <= 1.5 * 2^29 instead of 0.75 (LOGE = 4 only)
// product twiddles of inverse pass S (table tw + 2 m): for the step on local bits 0 / 1, and on 2 / 3
// as well where the pass runs both steps (LOGE = 4, wave-uniform twiddles)
template <int LOGM, int S>
constexpr bool inv_pass_radix4() { return SGFHE_INV_RADIX4 && SGFHE_INV_R4(LOGM); }
template <int LOGM, int LOGE, int S>
__device__ __forceinline__ void load_inv_products(int32_t (&tp)[(1 << LOGE) - 1], const int32_t *tw, int tid) {
    if constexpr (inv_pass_radix4<LOGM, S>()) {
        load_twiddles<LOGM, LOGE, S, 0, 0>(tp, tw + (2 << LOGM), (uint32_t)tid >> S);
        if constexpr (LOGE == 4 && (1 << S) >= 64)
            load_twiddles<LOGM, LOGE, S, 2, 2>(tp, tw + (2 << LOGM), (uint32_t)tid >> S);
    }
}
template <int LOGM, int NP, int LOGE, int SCUR, int SLAST, bool FINAL = false, bool WIDE0 = false>
struct InvPasses {
    static __device__ __forceinline__ void run(int32_t (&x)[NP][1 << LOGE], uint32_t *lds,
                                               const int32_t *tw, int tid, const Mod &md,
                                               const int32_t (&t)[(1 << LOGE) - 1],
                                               const int32_t (&tp)[(1 << LOGE) - 1]) {
        constexpr int MODE = (FINAL && SCUR == SLAST) ? 2 : 1;
#if SGFHE_INV_RADIX4
        if constexpr (!inv_pass_radix4<LOGM, SCUR>()) {
            // radix-2 stages with the searched reduction pattern (inputs and outputs below 0.75 * 2^29)
            static_assert(!WIDE0 || (LOGE == 4 && !(FINAL && SLAST == 0)), "wide first pass: radix 16, not the final pass");
            inv_stages<NP, LOGE, 0, LOGE - 1, 0, MODE, (WIDE0 && SCUR == 0) ? 3 : MODE>(x, t, md);
        } else
        // Radix-4 steps with deferred reductions (inv_step4).  Every pass opens with one on local
        // bits 0 and 1; a pass whose twiddles are wave-uniform (scalar registers) runs bits 2 and 3
        // the same way, the others as radix-2 stages.  Reductions: the four-fold sums y0 after each
        // step; in the radix-2 tail the sums of the last stage (in the final pass only those that
        // can exceed 1.4 * 2^29).  tests/rns_model.py NttModel.inv_pass / RangeModel.inverse.
        {
            static_assert(!WIDE0, "the radix-4 inverse takes every polynomial below 0.75 * 2^29");
            constexpr bool UNIFORM = (1 << SCUR) >= 64;
            inv_step4<NP, LOGE, 0>(x, t, tp, md);
            inv_reduce_y0<NP, LOGE, 0>(x, md);
            if constexpr (LOGE == 4 && UNIFORM) {
                inv_step4<NP, LOGE, 2>(x, t, tp, md);
                inv_reduce_y0<NP, LOGE, 2>(x, md);
            } else if constexpr (LOGE == 4) {
                inv_stage<NP, LOGE, 2, 0u, 0u>(x, t, md);
                // last stage: all eight sums, or in the final pass the sums of sums (registers 0..3)
                inv_stage<NP, LOGE, 3, MODE == 2 ? 0x000Fu : 0x00FFu, MODE == 2 ? 0x000Fu : 0x00FFu>(x, t, md);
            } else {
                static_assert(LOGE == 3, "8 or 16 points per thread");
                inv_stage<NP, LOGE, 2, MODE == 2 ? 0u : 0x0Fu, MODE == 2 ? 0u : 0x0Fu>(x, t, md);
            }
        }
#else
        static_assert(!WIDE0 || (LOGE == 4 && !(FINAL && SLAST == 0)), "wide first pass: radix 16, not the final pass");
        inv_stages<NP, LOGE, 0, LOGE - 1, 0, MODE, (WIDE0 && SCUR == 0) ? 3 : MODE>(x, t, md);
#endif
        if constexpr (SCUR < SLAST) {
            int32_t tn[(1 << LOGE) - 1], tpn[(1 << LOGE) - 1];
            load_twiddles<LOGM, LOGE, SCUR + LOGE, LOGE - 1, 0>(tn, tw,
                                                                (uint32_t)tid >> (SCUR + LOGE));
            load_inv_products<LOGM, LOGE, SCUR + LOGE>(tpn, tw, tid);
            lds_store<LOGM, NP, LOGE, SCUR>(x, lds, tid);
            exchange_sync<LOGE, SCUR>();
            lds_load<LOGM, NP, LOGE, SCUR + LOGE>(x, lds, tid);
            InvPasses<LOGM, NP, LOGE, SCUR + LOGE, SLAST, FINAL, WIDE0>::run(x, lds, tw, tid, md, tn, tpn);
        }
    }
};

// Forward transform.  In: x[q][e] = coefficient tid + T e of polynomial q, |x| <= 1.01 * 2^29.
// Out: x[q][e] = slot E tid + e, |x| < 3.95 * 2^29.  `lds` must hold NP * m words.
template <int LOGM, int NP, int LOGE, class F = NoHook>
__device__ __forceinline__ void ntt_forward(int32_t (&x)[NP][1 << LOGE], uint32_t *lds,
                                            const int32_t *tw, int tid, const Mod &md,
                                            const F &before_last = F()) {
    using G = NttGeom<LOGM, LOGE>;
    constexpr int BLO = G::RHO == 0 ? 0 : LOGE - G::RHO;
    int32_t t[(1 << LOGE) - 1];
    load_twiddles<LOGM, LOGE, G::STOP, LOGE - 1, BLO>(t, tw, 0u);
    fwd_stages<NP, LOGE, LOGE - 1, BLO>(x, t, md);
    if constexpr (G::SFIRST >= 0)
        FwdPasses<LOGM, NP, LOGE, G::STOP, G::SFIRST>::run(x, lds, tw, tid, md, before_last);
    else
        before_last();
}

// Inverse transform (unscaled).  In: slots E tid + e, |x| <= 0.75 * 2^29 (WIDE0: polynomial 0 up
// to 1.5 * 2^29; needs LOGE = 4 and at least one full pass).  Out: coefficient
// tid + T e, |x| < 1.4 * 2^29, in registers; the last LDS accesses of every thread were loads in
// layout STOP.
template <int LOGM, int NP, int LOGE, bool WIDE0 = false>
__device__ __forceinline__ void ntt_inverse(int32_t (&x)[NP][1 << LOGE], uint32_t *lds,
                                            const int32_t *tw, int tid, const Mod &md) {
    using G = NttGeom<LOGM, LOGE>;
    if constexpr (G::RHO == 0) {
        int32_t t[(1 << LOGE) - 1], tp[(1 << LOGE) - 1];
        load_twiddles<LOGM, LOGE, 0, LOGE - 1, 0>(t, tw, (uint32_t)tid);
        load_inv_products<LOGM, LOGE, 0>(tp, tw, tid);
        InvPasses<LOGM, NP, LOGE, 0, G::STOP, true, WIDE0>::run(x, lds, tw, tid, md, t, tp);
    } else {
        int32_t tp[(1 << LOGE) - 1];
        if constexpr (G::SLAST_INV >= 0) {
            int32_t t[(1 << LOGE) - 1], tq[(1 << LOGE) - 1];   // (tp is the partial pass's twiddle array)
            load_twiddles<LOGM, LOGE, 0, LOGE - 1, 0>(t, tw, (uint32_t)tid);
            load_inv_products<LOGM, LOGE, 0>(tq, tw, tid);
            InvPasses<LOGM, NP, LOGE, 0, G::SLAST_INV, false, WIDE0>::run(x, lds, tw, tid, md, t, tq);
            load_twiddles<LOGM, LOGE, G::STOP, LOGE - 1, LOGE - G::RHO>(tp, tw, 0u);
            lds_store<LOGM, NP, LOGE, G::SLAST_INV>(x, lds, tid);
            SGFHE_SYNC();
            lds_load<LOGM, NP, LOGE, G::STOP>(x, lds, tid);
        } else {
            load_twiddles<LOGM, LOGE, G::STOP, LOGE - 1, LOGE - G::RHO>(tp, tw, 0u);
        }
        inv_stages<NP, LOGE, LOGE - G::RHO, LOGE - 1, LOGE - G::RHO, 0>(x, tp, md);
    }
}

}  // namespace sgfhe
