// HIP kernels of the gate-bootstrap engine (gfx950).  Hot path of nucypher/SGFHE.jl:
// _bootstrap_internal / external_product / flatten / reduce_modulus
// (/root/reference/src/fhe.jl:519-621, src/utils.jl:78-189), restructured as in DESIGN.md:
//
//   acc <- acc + (x^j - 1) * sum_row u_row (*) C_k[row]          (CMux form of fhe.jl:580-581)
//
// State per bootstrap: the *flattened* accumulators, i.e. for every coefficient of (a, b) the
// two base-B digits (lo, hi) of x' = (acc + off) mod Q with off = (1 + B) s (utils.jl:162-181).
// Per iteration k two kernels run over a chunk of bootstraps in lock-step:
//   k_extprod  one workgroup per (bootstrap, RNS prime): digits -> 4 forward NTTs -> 8 pointwise
//              multiply-accumulates against the NTT-domain key slice -> 2 inverse NTTs ->
//              negacyclic rotate-and-subtract (x^j - 1) -> residues y mod p_i
//   k_crt_acc  per coefficient: CRT of the residues to the exact integer, reduce mod Q, add to
//              the accumulator, flatten again (divide by B).
#pragma once

#include <type_traits>

#include "ntt.h"

namespace sgfhe {

typedef unsigned __int128 u128;

constexpr int NPR_MAX = 7;  // at most this many RNS primes (p_i < 2^29); a ctx uses the fewest whose
                            // product covers 5 m B Q (4 at Params(64), 5 at Params(512/1024))
#ifndef SGFHE_LOGE
#define SGFHE_LOGE 4
#endif
constexpr int LOGE = SGFHE_LOGE;  // points per thread (2^LOGE) in every NTT of the engine

struct PrimeK {
    int32_t p;        // prime, < 2^29
    uint32_t pinv;    // p^-1 mod 2^32
    int32_t sR;       // -(s R^-1) mod p, centred (digit offset s of utils.jl:162-166, R = 2^32)
    int32_t sRr;      // the same for the randomised flatten: offset s + xmax (utils.jl:198-241)
    uint32_t hoff;    // offset added to the output residue: (p-1)/2 for the last prime, else 0
    int32_t r1, r2, r3;  // R, R^2, R^3 mod p, centred
    int32_t qmodp;    // Q mod p, centred
    int32_t kappaR;   // key scale kappa * R mod p (centred), kappa = R^2 m^-1 (M/p)^-1 mod p
    int32_t minvR;    // m^-1 * R mod p, centred (debug inverse NTT scaling)
    float invp;       // 1 / p
    const int32_t *twf;  // forward twiddles psi^bitrev(i) * R mod p (Montgomery form, centred)
    const int32_t *twi;  // inverse twiddles psi^-bitrev(i) * R mod p
    uint32_t npr;     // number of primes of the ctx (the same in every record)
    // quarter form of the latency kernels (k_fwd_quarter / k_inv_quarter / k_crt_lean1q):
    int32_t f1, f2, f3, fp2, fp3;   // forward twiddles of the first two stages: twf[1..3], and the products twf[2 m + 2], [2 m + 3]
    int32_t v1, v2, v3;             // inverse twiddles of the last two stages: twi[1..3]
    const int32_t *twq;  // twiddle tables of the four quarter transforms: block q = [f_q | v_q | fp_q | vp_q], m / 4 words each
    const int32_t *pw;   // psi^e * R mod p, centred, e in [0, 2 m): (x^j - 1) in the NTT domain is psi^(j (2 brv(s) + 1)) - 1 at slot s
};
__device__ __forceinline__ Mod mod_of(const PrimeK &P) { return Mod{P.p, -P.p, P.pinv}; }

// The npr PrimeK records live in device memory and are indexed by the (wave-uniform) prime
// index of the workgroup.
typedef const PrimeK *__restrict__ PrimeSet;

constexpr int CRT_ALPHA_MAX = 6 * NPR_MAX + 2;  // entries of the alpha-indexed CRT correction table

// CRT / flatten constants, resident in device memory.
struct CrtConst {
    u128 Q, B;
    u128 c[NPR_MAX];     // (M / p_i) mod Q
    u128 T[CRT_ALPHA_MAX];   // (-alpha * M - H') mod Q,  H' = (M / p_last) * (p_last - 1) / 2;
                             // alpha < 6 npr + 1: the residues are non-negative representatives
                             // below 5.7 p (6.2 p for the last prime), not canonical ones
    u128 offneg;         // (Q - off) mod Q, off = (1 + B) s mod Q
    u128 offneg_rnd;     // the same with s + xmax in place of s (randomised flatten)
    uint64_t xmax;       // v_i uniform in [-xmax, xmax], xmax = 3 (B / 2) (utils.jl:210-214)
    u128 DQ;             // DQ_tilde mod Q
    u128 halfQ;          // Q / 2 (centred lift of key residues)
    u128 roundthr;       // Q / 2 + (Q odd)        (utils.jl:84)
    double invQ, invB;
    // 32-bit limb / double views of the same constants for k_crt_acc's 96-bit arithmetic
    uint32_t c32[NPR_MAX][3];
    uint32_t Q32[3];
    alignas(16) uint32_t T32[CRT_ALPHA_MAX][4];
    double cd[NPR_MAX], Td[CRT_ALPHA_MAX], Bd;
    float invp[NPR_MAX];
    uint32_t npr;        // number of primes in use
    uint32_t logr;
    ulonglong2 dig0, digP, digN;  // (lo, hi) digits of x' for acc = 0, DQ_tilde, Q - DQ_tilde
};

enum : uint32_t {
    MODE_PLAIN = 1u,  // k_extprod: no (x^j - 1) factor (external_product debug hook)
    MODE_NOACC = 2u,  // k_crt_acc: do not add the previous accumulator
    MODE_CANON = 4u,  // k_crt_acc: write canonical residues instead of digits
    MODE_RANDOM = 16u,// randomised flatten (rng != nothing, utils.jl:198-241): digits e_i hold
                      // u_i + s + xmax, u_i in (-2B, 2B]
    MODE_WIDE = 32u   // with MODE_RANDOM and B >= 2^46: stored digits reach 4 B >= 2^48, bits 48..55
                      // live in the record's third plane (Params(2048): B = 35 * 2^41)
};

// ---- small 128-bit helpers -----------------------------------------------------------------

__device__ __forceinline__ double u128_to_double(u128 x) {
    return (double)(uint64_t)(x >> 64) * 18446744073709551616.0 + (double)(uint64_t)x;
}
__device__ __forceinline__ u128 mul_u64_u128(uint64_t a, u128 b) {
    // a * b mod 2^128
    return (u128)a * (uint64_t)b + (((u128)(a * (uint64_t)(b >> 64))) << 64);
}
// x mod d for x < 2^127, d < 2^94, quotient < 2^62; inv = 1.0 / d.
__device__ __forceinline__ u128 mod_wide(u128 x, u128 d, double inv, uint64_t *quot) {
    uint64_t q = (uint64_t)(u128_to_double(x) * inv);
    u128 r = x - mul_u64_u128(q, d);
#pragma unroll 1
    while ((__int128)r < 0) { r += d; q--; }
#pragma unroll 1
    while (r >= d) { r -= d; q++; }
    if (quot) *quot = q;
    return r;
}

// ---- digit planes ---------------------------------------------------------------------------------
// Per (bootstrap, c) a record of 16 m bytes (the size of the canonical residues it replaces):
//   [lo words: digit 0 | digit 1] 2 m x uint32      bits 0..31 of the digits
//   [hi words: digit 0 | digit 1] 2 m x uint16      bits 32..47          (12 m bytes in use)
//   [top bytes: digit 0 | digit 1] 2 m x uint8      bits 48..55, MODE_WIDE only (14 m bytes in use)
// digit 0 = lo, digit 1 = hi of x' for acc_a (c = 0) and acc_b (c = 1); every stored digit is
// below 2^48 (B < 2^46 is checked at ctx creation; the randomised mode stores up to 4 B).  Plane
// p = 2 c + digit is the p-th row of u = [a_lo, a_hi, b_lo, b_hi] (fhe.jl:524-526) and multiplies
// key row p.  6 bytes per digit instead of 8: the CRT kernel is bound by these bytes.
// Byte offsets inside a chunk's buffers fit 32 bits (chunk <= 8192, buffers below 4 GB): a uniform
// base plus a 32-bit byte offset takes the SGPR-base addressing mode, with no 64-bit address
// arithmetic on the vector ALU.
template <class T>
__device__ __forceinline__ T ld_off(const void *base, uint32_t byte_off) {
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}
template <class T>
__device__ __forceinline__ void st_off(void *base, uint32_t byte_off, T v) {
    *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byte_off) = v;
}
// Buffer-descriptor addressing (MI355X guide, T8 / T20): the 128-bit resource lives in SGPRs, a
// load or store takes a 32-bit per-lane byte offset (VGPR), a wave-uniform byte offset (SGPR,
// formed on the scalar ALU) and a 12-bit immediate -- no 64-bit address arithmetic on the vector
// ALU, which is what bounds k_extprod.  Everything that builds a descriptor or an soffset must be
// provably wave-uniform (kernel arguments, blockIdx, loop counters), or hipcc wraps the access in
// a waterfall loop.  Accesses beyond `bytes` read 0 / are dropped by the hardware range check.
typedef __amdgpu_buffer_rsrc_t BufRsrc;
__device__ __forceinline__ BufRsrc make_rsrc(const void *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint32_t buf_ld_u32(BufRsrc r, uint32_t voff, uint32_t soff) {
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ uint32_t buf_ld_u16(BufRsrc r, uint32_t voff, uint32_t soff) {
    return (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ uint32_t buf_ld_u8(BufRsrc r, uint32_t voff, uint32_t soff) {
    return (uint8_t)__builtin_amdgcn_raw_buffer_load_b8(r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ int4 buf_ld_i4(BufRsrc r, uint32_t voff, uint32_t soff) {
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return make_int4((int)v.x, (int)v.y, (int)v.z, (int)v.w);
}
// Cache policy of the residue stores (aux bits: 1 = sc0, 2 = nt, 16 = sc1), measured in the
// pipeline (profiles/r02_exp_plain_vs_nt_stores_final.txt, r02_exp_store_policy.txt): non-temporal
// 1863, plain 1893 (the Infinity Cache keeps up to 256 MB of written data for the next kernel's
// reads, and a non-temporal store writes more slowly: tools/ubench_mall.hip, 4.0 against
// 5.9 TB/s); on another box plain 1922, sc0 1924, sc1 1936, sc0 sc1 1935, sc0 nt 1903.  sc1
// (write-through) it is.
#ifndef SGFHE_YRES_AUX
#define SGFHE_YRES_AUX 16
#endif
__device__ __forceinline__ void buf_st_u32(BufRsrc r, uint32_t voff, uint32_t soff, uint32_t v) {
    __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)voff, (int)soff, SGFHE_YRES_AUX);
}

__device__ __forceinline__ const uint32_t *digit_lo_plane(const uint64_t *dig, size_t bc, uint32_t M) {
    return reinterpret_cast<const uint32_t *>(dig + bc * 2 * M);
}
__device__ __forceinline__ const uint16_t *digit_hi_plane(const uint64_t *dig, size_t bc, uint32_t M) {
    return reinterpret_cast<const uint16_t *>(dig + bc * 2 * M + M);
}
__device__ __forceinline__ ulonglong2 load_digits(const uint64_t *__restrict__ dig, size_t bc,
                                                  uint32_t i, uint32_t M, bool wide = false) {
    const uint32_t rec = (uint32_t)bc * 16u * M;           // byte offset of the record
    const uint32_t lo = rec + 4u * i, hi = rec + 8u * M + 2u * i;
    ulonglong2 d = make_ulonglong2(
        ld_off<uint32_t>(dig, lo) | ((uint64_t)ld_off<uint16_t>(dig, hi) << 32),
        ld_off<uint32_t>(dig, lo + 4u * M) | ((uint64_t)ld_off<uint16_t>(dig, hi + 2u * M) << 32));
    if (wide) {
        const uint32_t top = rec + 12u * M + i;
        d.x |= (uint64_t)ld_off<uint8_t>(dig, top) << 48;
        d.y |= (uint64_t)ld_off<uint8_t>(dig, top + M) << 48;
    }
    return d;
}
__device__ __forceinline__ void store_digits(uint64_t *__restrict__ dig, size_t bc, uint32_t i,
                                             uint32_t M, uint64_t lo, uint64_t hi, bool wide = false) {
    const uint32_t rec = (uint32_t)bc * 16u * M;
    const uint32_t ol = rec + 4u * i, oh = rec + 8u * M + 2u * i;
    st_off<uint32_t>(dig, ol, (uint32_t)lo);
    st_off<uint32_t>(dig, ol + 4u * M, (uint32_t)hi);
    st_off<uint16_t>(dig, oh, (uint16_t)(lo >> 32));
    st_off<uint16_t>(dig, oh + 2u * M, (uint16_t)(hi >> 32));
    if (wide) {
        const uint32_t top = rec + 12u * M + i;
        st_off<uint8_t>(dig, top, (uint8_t)(lo >> 48));
        st_off<uint8_t>(dig, top + M, (uint8_t)(hi >> 48));
    }
}

// ---- randomised flatten (utils.jl:198-241) ----------------------------------------------------------
// The draws come from a ChaCha counter stream (the RFC 8439 block function with SGFHE_RND_ROUNDS
// rounds, 8 by default: "ChaCha8") keyed with the caller's 32 bytes, so the digit perturbations are
// cryptographically strong when the key is.  (Rounds 1-3 used Philox4x32-10 keyed by 64 bits: a
// statistical generator.)  Functional parity only: the reference draws from the caller's Julia
// rng, whose stream cannot be reproduced here.
// The draw for a digit pair is addressed by a counter that does not depend on how a batch is cut
// into chunks or scheduled: the 128 bits of coefficient x are words 4 (x mod 4) .. + 3 of the block
// whose state words 12 .. 15 are
//   (x div 4   with x = coefficient index (c << log2 m) + i of the accumulator pair (c = 0: a, 1: b),
//    y: 0 for the initial accumulators, k + 1 after k-loop iteration k
//       (pack_encrypted_bits' flatten of as_i: 2^31 | i),
//    z: index of the bootstrap within the call (of the ciphertext, for packing),
//    w: number of the call since sgfhe_set_random_flatten),
// so a thread of k_crt_lean_rnd (four adjacent coefficients) computes exactly one block.
// oracle/bigint_oracle.py restates the same stream, so the randomised mode is bit-comparable with
// the oracle.
#ifndef SGFHE_RND_ROUNDS
#define SGFHE_RND_ROUNDS 8
#endif
struct ChaChaKey {
    uint32_t k[8];
};
// ChaCha block function (RFC 8439 section 2.3 with ROUNDS rounds): state words 12 .. 15 = c12 .. c15
template <int ROUNDS>
__device__ __forceinline__ void chacha_block(const ChaChaKey &key, uint32_t c12, uint32_t c13,
                                             uint32_t c14, uint32_t c15, uint32_t (&out)[16]) {
    static_assert(ROUNDS % 2 == 0, "double rounds");
    const uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u,
                            key.k[0], key.k[1], key.k[2], key.k[3],
                            key.k[4], key.k[5], key.k[6], key.k[7], c12, c13, c14, c15};
    uint32_t x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = s[i];
#define SGFHE_QR(a, b, c, d)                                                             \
    x[a] += x[b]; x[d] = __builtin_rotateleft32(x[d] ^ x[a], 16);                        \
    x[c] += x[d]; x[b] = __builtin_rotateleft32(x[b] ^ x[c], 12);                        \
    x[a] += x[b]; x[d] = __builtin_rotateleft32(x[d] ^ x[a], 8);                         \
    x[c] += x[d]; x[b] = __builtin_rotateleft32(x[b] ^ x[c], 7);
#pragma unroll
    for (int r = 0; r < ROUNDS / 2; r++) {
        SGFHE_QR(0, 4, 8, 12) SGFHE_QR(1, 5, 9, 13) SGFHE_QR(2, 6, 10, 14) SGFHE_QR(3, 7, 11, 15)
        SGFHE_QR(0, 5, 10, 15) SGFHE_QR(1, 6, 11, 12) SGFHE_QR(2, 7, 8, 13) SGFHE_QR(3, 4, 9, 14)
    }
#undef SGFHE_QR
#pragma unroll
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}
// A call gathered from several callers (engine.hip Coalescer, round 5): every row of the batch draws from the
// stream of the ctx it came in on -- that ctx's key, call number and the row's index in ITS call -- so that each
// caller gets the bytes its call gives alone.  One record per row of the combined call, in device memory.
struct RndRow {
    ChaChaKey key;
    uint32_t call, boot;
};
struct RndArgs {
    ChaChaKey key;         // 32-byte key of the draw stream
    uint32_t call, chunk;  // per-call counter, index of the chunk's first bootstrap in the call
    const RndRow *rows;    // gathered call: the stream of every row of the call (kernels instantiated with ROWS); else null
};
// The stream of bootstrap `row` of the chunk: key, its index in its call (counter word z), its call number (w).
// ROWS = false leaves the key in the kernel arguments (scalar registers), as before the table existed.
template <bool ROWS>
__device__ __forceinline__ void rnd_stream(const RndArgs &ra, uint32_t row, ChaChaKey &key, uint32_t &cz, uint32_t &cw) {
    if constexpr (ROWS) {
        const RndRow r = ra.rows[ra.chunk + row];
        key = r.key; cz = r.boot; cw = r.call;
    } else {
        key = ra.key; cz = ra.chunk + row; cw = ra.call;
    }
}
// the 128 bits of one coefficient (ctr.x = its index): a quarter of its quad's block
__device__ __forceinline__ uint4 rnd128(const ChaChaKey &key, uint4 ctr) {
    uint32_t w[16];
    chacha_block<SGFHE_RND_ROUNDS>(key, ctr.x >> 2, ctr.y, ctr.z, ctr.w, w);
    const uint32_t q = ctr.x & 3u;
    return make_uint4(q == 0 ? w[0] : q == 1 ? w[4] : q == 2 ? w[8] : w[12],
                      q == 0 ? w[1] : q == 1 ? w[5] : q == 2 ? w[9] : w[13],
                      q == 0 ? w[2] : q == 1 ? w[6] : q == 2 ? w[10] : w[14],
                      q == 0 ? w[3] : q == 1 ? w[7] : q == 2 ? w[11] : w[15]);
}
// Digits of the randomised flatten of acc, given xn = (acc + (s + xmax)(1 + B)) mod Q:
//   r_i = v_i + xmax uniform in [0, 2 xmax];  x2 = (xn - r_0 - r_1 B) mod Q = a' of utils.jl:179
//   for the shifted value;  (lo, hi) = divmod(x2, B);  e_i = (lo, hi) + r_i  ( = u_i + s + xmax ).
__device__ __forceinline__ ulonglong2 random_digits(u128 xn, const CrtConst *CC, const ChaChaKey &key,
                                                    uint4 ctr) {
    const uint4 rv = rnd128(key, ctr);
    const uint64_t span = 2 * CC->xmax + 1;
    const uint64_t r0 = (uint64_t)(((u128)(((uint64_t)rv.y << 32) | rv.x) * span) >> 64);
    const uint64_t r1 = (uint64_t)(((u128)(((uint64_t)rv.w << 32) | rv.z) * span) >> 64);
    const u128 B = CC->B, Q = CC->Q;
    const u128 tsub = mod_wide((u128)r1 * (uint64_t)B + r0, Q, CC->invQ, nullptr);
    const u128 x2 = xn >= tsub ? xn - tsub : xn + Q - tsub;
    uint64_t hi;
    const u128 lo = mod_wide(x2, B, CC->invB, &hi);
    return make_ulonglong2((uint64_t)lo + r0, hi + r1);
}

// ---- digit -> residue ------------------------------------------------------------------------
// (e - s) * R^-1 mod p for a stored digit e < 2^48: |REDC(e)| <= 2^16 + p / 2, plus the centred
// constant -(s R^-1): |result| <= p + 2^16.
__device__ __forceinline__ int32_t digit_reduce(uint64_t e, const Mod &md, int32_t sR) {
    return sredc((int64_t)e, md) + sR;
}

// ---- k_extprod ----------------------------------------------------------------------------------
// grid = chunk * NPR workgroups of T = m / 16 threads; chunk is a multiple of 8.  Workgroups are
// dealt round-robin to the 8 XCDs (blocks g and g + 8 share an XCD, MI355X_MICROARCH.md), so
// the NPR prime-workgroups of one bootstrap are given the same g mod 8: they re-read the same
// digits and hit in that XCD's L2.  At m = 8192 a workgroup is 512 threads, at most 128 VGPRs and
// 64 KiB of LDS (32 KiB NTT exchange buffer + 32 KiB thread-private accumulator of the second
// product polynomial): two workgroups share a CU, so one computes butterflies while the other
// sits in an LDS exchange, a barrier or a load.
//   dig    [chunk][2][2][m]     digit planes (see above)
//   keyk   [NPR][4][2][m]       NTT-domain key slice of iteration k (slot order, scaled by kappa,
//                               centred residues as int32)
//   yres   [chunk][2][NPR][m]   output residues y_i = (M/p_i)^-1 * D mod p_i (+ hoff): non-negative
//                               representatives below 5.7 p_i (6.2 p_i for the last prime)
//   ua     [chunk][n]           j = u.a[k] of every bootstrap (fhe.jl:566,580)
// The four digit polynomials u = [a_lo, a_hi, b_lo, b_hi] (fhe.jl:524-526) go through the forward
// NTT one at a time (phase = key row); the two product polynomials through the inverse NTT one at
// a time.  The running NTT-domain sum z_0 lives in registers, z_1 in LDS.
#ifndef SGFHE_EXT_WAVES
#define SGFHE_EXT_WAVES 4
#endif
// Digit loads of a phase in groups (round 5, profiles/r05_exp_wide.txt).  The WIDE instantiations have three loads
// per digit -- 48 values in flight beside the 32 accumulator registers -- and hipcc spilled five accumulator pairs
// around the phase loop (44-52 bytes of scratch per lane, rounds 3-4): Params(2048) with the randomised flatten ran
// at 297 bootstraps/s with the spills and runs at 345 without them (+16 %, same call).  The digits of a phase are
// requested in SPLIT groups, each reduced before the next is requested (a scheduling barrier between the groups
// bounds the values in flight): 2 groups remove the scratch at m = 8192 / 16384, 4 at m = 4096 too; the 32-bit
// accumulator instead (-DSGFHE_WIDE_ACC32: 106-125 VGPRs, no scratch either) measured 339.
#ifndef SGFHE_WIDE_SPLIT
#define SGFHE_WIDE_SPLIT(LOGM) ((LOGM) >= 13 ? 2 : 4)
#endif
#ifndef SGFHE_DET_SPLIT     // the same for the two-plane kernels (A/B builds: -DSGFHE_DET_SPLIT=2)
#define SGFHE_DET_SPLIT 0
#endif
// LE: points per thread (2^LE); 16 wherever that leaves a full wavefront, 8 for m <= 512
// WIDE: the digit planes carry a third plane (bits 48..55; MODE_WIDE): a separate instantiation, so
// that the deterministic kernel is textually what it was.
template <int LOGM, int LE, bool WIDE = false>
__global__ void __launch_bounds__((NttGeom<LOGM, LE>::T), (NttGeom<LOGM, LE>::T >= 256 ? SGFHE_EXT_WAVES : 1))
k_extprod(const uint64_t *__restrict__ dig, const int32_t *__restrict__ keyk,
          uint32_t *__restrict__ yres, const uint32_t *__restrict__ ua, PrimeSet PS, uint32_t k,
          uint32_t n, uint32_t mode) {
    using G = NttGeom<LOGM, LE>;
    constexpr int M = G::M, T = G::T, E = G::E;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
#ifdef SGFHE_EXT_PRIO  // experiment: issue priority over a co-running k_crt_lean (second lane)
    __builtin_amdgcn_s_setprio(SGFHE_EXT_PRIO);
#endif
    int32_t *const z1 = reinterpret_cast<int32_t *>(lds) + M + threadIdx.x;  // z1[e * T]: private to the thread, conflict-free

    const int tid = threadIdx.x;
    const uint32_t g = blockIdx.x;
    const uint32_t slot = g >> 3;
    const uint32_t npr = PS[0].npr;
    const uint32_t b = (slot / npr) * 8 + (g & 7);
    const uint32_t pi = slot % npr;
    const PrimeK P = PS[pi];
    const Mod md = mod_of(P);
    // Buffer descriptors over this launch's digit planes, key slice and residues (all below 4 GiB:
    // checked when the chunk size is set).  Built from kernel arguments and gridDim only.
    const uint32_t cpad = gridDim.x / npr;  // bootstraps of the (padded) chunk
    const BufRsrc rdig = make_rsrc(dig, cpad * 32u * (uint32_t)M);
    const BufRsrc rkey = make_rsrc(keyk, npr * 32u * (uint32_t)M);
#ifdef SGFHE_ABL_NO_YRES  // timing-only build: zero records, the range check drops every residue store
    const BufRsrc ryres = make_rsrc(yres, 0u);
#else
    const BufRsrc ryres = make_rsrc(yres, cpad * npr * 8u * (uint32_t)M);
#endif

    const int32_t sRd = (mode & MODE_RANDOM) ? P.sRr : P.sR;  // digit offset of the flatten mode
    constexpr int SPLIT = E == 16 ? (WIDE ? SGFHE_WIDE_SPLIT(LOGM) : SGFHE_DET_SPLIT) : 0;
#ifdef SGFHE_ACC0_32
    constexpr bool ACC32 = true;
#elif defined(SGFHE_WIDE_ACC32)   // A/B build of round 5: the 32-bit accumulator in the WIDE instantiations only
    constexpr bool ACC32 = WIDE && E == 16;
#else
    constexpr bool ACC32 = false;
#endif
    // ACC32: column 0 summed like column 1, Montgomery-reduced per phase (< 2.9 * 2^29); else the 64-bit
    // NTT-domain sum of column 0 over the four phases (|.| < 1.98 * 2^60)
    typename std::conditional<ACC32, int32_t, int64_t>::type acc0[E];
#pragma unroll
    for (int e = 0; e < E; e++) { acc0[e] = 0; z1[e * T] = 0; }
    // A real loop (not unrolled): one copy of the forward NTT in the instruction stream, and the
    // loads of a later phase cannot be hoisted over the registers of an earlier one.
#ifdef SGFHE_PH_UNROLL
#pragma unroll SGFHE_PH_UNROLL
#else
#pragma unroll 1
#endif
    for (int ph = 0; ph < 4; ph++) {
        // The thread index is made opaque per iteration: otherwise the ~60 loop-invariant LDS /
        // twiddle addresses derived from it are hoisted out of the loop and spilled.
        const int tid = (int)threadIdx.x + (int)opaque_zero();
        // 1. digits -> residues (flatten_poly, utils.jl:253-264, lifted to signed integers)
        int32_t x[1][E];
        // Buffer loads: the lane supplies one 32-bit offset per plane (4 tid / 2 tid); record,
        // plane and coefficient stride go into the scalar offset (SALU) -- no vector instruction
        // is spent on addresses (the flat form cost 23 per phase).
        const uint32_t srec = (b * 2u + (uint32_t)(ph >> 1)) * 16u * (uint32_t)M;  // byte offset of the record
        const uint32_t slo = srec + (uint32_t)(ph & 1) * 4u * (uint32_t)M;
        const uint32_t shi = srec + 8u * (uint32_t)M + (uint32_t)(ph & 1) * 2u * (uint32_t)M;
        const uint32_t vlo = 4u * (uint32_t)tid, vhi = 2u * (uint32_t)tid;
#pragma unroll
        for (int e = 0; e < E; e++) {
#ifdef SGFHE_ABL_NO_DIG
            x[0][e] = digit_reduce((uint64_t)(tid + e) * 0x9E3779B97F4Aull, md, sRd);  // timing-only build
#else
            // (the part of the stride below 4096 goes into the instruction's immediate offset)
            const uint32_t OL = (uint32_t)(4 * T * e), OH = (uint32_t)(2 * T * e);  // constants after unrolling
            const uint32_t lo = buf_ld_u32(rdig, vlo + (OL & 4095u), slo + (OL & ~4095u));
            uint32_t hi = buf_ld_u16(rdig, vhi + (OH & 4095u), shi + (OH & ~4095u));
            if constexpr (WIDE)
                hi |= buf_ld_u8(rdig, (uint32_t)tid + ((uint32_t)(T * e) & 4095u),
                                srec + 12u * (uint32_t)M + (uint32_t)(ph & 1) * (uint32_t)M + ((uint32_t)(T * e) & ~4095u)) << 16;
            x[0][e] = digit_reduce(lo | ((uint64_t)hi << 32), md, sRd);
#endif
            if constexpr (SPLIT > 1)    // next group of digit loads only after this group's reductions
                if ((e + 1) % (E / SPLIT) == 0 && e + 1 < E) __builtin_amdgcn_sched_barrier(0);
        }
        // the exchange buffer is reused: every wave must have finished the previous phase's loads
        SGFHE_SYNC();

        // 2. forward NTT of u[ph].  The key slice rows of this phase (from L2: every bootstrap of
        //    the launch reads the same slice) are requested after the transform.  Requesting them
        //    before its last pass (-DSGFHE_KEY_EARLY) keeps 32 more registers live through that
        //    pass: at the 128-register budget hipcc then spills or shuffles the accumulators, and
        //    the kernel is slower (profiles/r02_exp_buffer_addressing.txt).
#ifdef SGFHE_KEY_EARLY
        constexpr bool KEY_EARLY = LOGM >= 13;
#else
        constexpr bool KEY_EARLY = false;
#endif
        const uint32_t skey = (pi * 8u + (uint32_t)ph * 2u) * 4u * (uint32_t)M;  // byte offset of key row ph, column 0
        const uint32_t vkey = 4u * (uint32_t)E * (uint32_t)tid;
        int4 ka4[E / 4], kb4[E / 4];
        auto load_key = [&]() {
#pragma unroll
            for (int h = 0; h < E / 4; h++) {
#ifdef SGFHE_ABL_NO_KEY
                ka4[h] = make_int4(tid, h, ph, 7), kb4[h] = make_int4(h, tid, 5, ph);  // timing-only
#else
                ka4[h] = buf_ld_i4(rkey, vkey + 16u * (uint32_t)h, skey);
                kb4[h] = buf_ld_i4(rkey, vkey + 16u * (uint32_t)h, skey + 4u * (uint32_t)M);
#endif
            }
        };
        if constexpr (KEY_EARLY) {
            ntt_forward<LOGM, 1, LE>(x, lds, P.twf, tid, md, load_key);
        } else {
            ntt_forward<LOGM, 1, LE>(x, lds, P.twf, tid, md);
            load_key();
        }

        // 3. pointwise: z_c += U * K[ph][c]   (fhe.jl:527-528 in the NTT domain), |U| < 3.95 * 2^29,
        //    |K| <= p / 2
        //    column 0: 64-bit multiply-accumulate (one v_mad_i64_i32 per product), reduced once after
        //              the loop.  (SGFHE_ACC0_32: Montgomery-reduced per phase into 16 registers
        //              instead of 32 -- 104 VGPRs and no spills, but 2 more multiplies per product:
        //              measured 230.8 against 225.1 us per launch, profiles/r02_acc32_* vs r02_v1_*.)
        //    column 1: Montgomery-reduced (|.| < 0.75 * 2^29) and added to the LDS accumulator
        const Mod &mdp = md;
#pragma unroll
        for (int h = 0; h < E / 4; h++) {
            const int4 a = ka4[h], bq = kb4[h];
            const int32_t ka[4] = {a.x, a.y, a.z, a.w};
            const int32_t kb[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int e = 4 * h + t;
                const int32_t u = x[0][e];
                if constexpr (ACC32) acc0[e] += smont(u, ka[t], mdp);
                else acc0[e] += (int64_t)u * ka[t];
                int32_t *zp = reinterpret_cast<int32_t *>(lds) + M + e * T + tid;
                *zp += smont(u, kb[t], mdp);
            }
        }
    }
    // 4. inverse NTT of both columns at once (same twiddles, one set of exchanges; the second
    //    exchange buffer is the LDS area that held z_1):
    //    P_c = (M/p)^-1 * sum_row u_row (*) C_row[c]  mod p
    int32_t z[2][E];
    // With 16 points per thread the first inverse pass takes column 0 as the Montgomery step
    // leaves it (|.| < 1.49 * 2^29) and reduces six more of its sums instead (ntt.h inv_red_mask).
#if defined(SGFHE_ACC0_32)
    constexpr bool WIDE0 = false;
#elif SGFHE_INV_RADIX4
    // (the radix-4 inverse reduces column 0 on entry like column 1)
    constexpr bool WIDE0 = !SGFHE_INV_R4(LOGM) && LE == 4 && NttGeom<LOGM, LE>::SLAST_INV >= 0 &&
                           !(NttGeom<LOGM, LE>::RHO == 0 && NttGeom<LOGM, LE>::STOP == 0);
#else
    constexpr bool WIDE0 = LE == 4 && NttGeom<LOGM, LE>::SLAST_INV >= 0 &&
                           !(NttGeom<LOGM, LE>::RHO == 0 && NttGeom<LOGM, LE>::STOP == 0);
#endif
#pragma unroll
    for (int e = 0; e < E; e++) {
        if constexpr (ACC32) {
            z[0][e] = sred((int32_t)acc0[e], md);
        } else {
            const int32_t r0 = sredc(acc0[e], md);   // |REDC| < 1.49 * 2^29
            z[0][e] = WIDE0 ? r0 : sred(r0, md);
        }
        z[1][e] = sred(z1[e * T], md);           // four phases: < 2.99 * 2^29
    }
    SGFHE_SYNC();  // every thread has taken its z_1 out of the buffer the exchanges now reuse
    ntt_inverse<LOGM, 2, LE, (WIDE0 && !ACC32)>(z, lds, P.twi, tid, md);  // |z| < 1.4 * 2^29

    // Output addressing as for the digit loads: scalar offsets per column and coefficient, one
    // 32-bit lane offset.
    const uint32_t sy0 = (b * 2u * npr + pi) * 4u * (uint32_t)M;
    const uint32_t sy1 = sy0 + npr * 4u * (uint32_t)M;
    const uint32_t vout = 4u * (uint32_t)tid;
    // 5. y = x^j P - P  (mul_by_xj_minus_one, fhe.jl:554-556, applied to the product)
    if (mode & MODE_PLAIN) {
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int e = 0; e < E; e++)
                buf_st_u32(ryres, vout + ((uint32_t)(4 * T * e) & 4095u),
                              (c ? sy1 : sy0) + ((uint32_t)(4 * T * e) & ~4095u),
                              condsub(sfull(z[c][e], md) + P.hoff, (uint32_t)P.p));
        return;
    }
    const uint32_t j = ua[(size_t)b * n + k];
    const uint32_t yoff = 3u * (uint32_t)P.p + P.hoff;
#ifdef SGFHE_EPI_PLAIN   // (A/B build, round 4: not adopted -- see below)
    // Both product polynomials go to LDS in the plain layout (word i of polynomial c at c m + i): lane
    // t writes word t + T e and reads word (t - j + T e) mod m, consecutive lanes consecutive words
    // either way, so neither access has a bank conflict whatever j is.  (Round 3 kept the transform's
    // swizzled layout here; its j-shifted reads then collide where the run of 64 source indices
    // crosses a swizzle boundary: 2.6 % of the kernel's LDS cycles, profiles/r03_v10_counters.json.)
    // The buffer is the one the transform's last loads came from: every wave has to be done with them.
    SGFHE_SYNC();
    {
        int32_t *const ldsi = reinterpret_cast<int32_t *>(lds);
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int e = 0; e < E; e++) ldsi[c * M + T * e + tid] = z[c][e];
    }
    SGFHE_SYNC();
    {
        const uint32_t s0 = ((uint32_t)tid - j) & (2 * M - 1);
        const uint32_t low = s0 & ((1u << G::STOP) - 1u);
        const uint32_t h0 = s0 >> G::STOP;
#pragma unroll
        for (int e = 0; e < E; e++) {
            const uint32_t he = h0 + e;
            const uint32_t addr = ((he & (E - 1)) << G::STOP) | low;
            // x^m = -1: the source is negated when bit LE of he is set.  -v = (v ^ -1) + 1, so
            // with smask = 0 / -1 the output is (v ^ smask) + (yoff - smask - z): one subtract and
            // one xor-add per residue, the per-e constants shared by both columns.
            const uint32_t smask = 0u - ((he >> LE) & 1u);
            const uint32_t yoe = yoff - smask;
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const uint32_t v = lds[c * M + addr];
                // |+-v - z| < 2.7 p: + 3 p makes a non-negative representative below 5.7 p (no
                // reduction at all: k_crt_acc takes any such residues); + (p - 1) / 2 for the last
                // prime, folded into the same wave-uniform constant
                const uint32_t y = (v ^ smask) + (yoe - (uint32_t)z[c][e]);
                buf_st_u32(ryres, vout + ((uint32_t)(4 * T * e) & 4095u),
                              (c ? sy1 : sy0) + ((uint32_t)(4 * T * e) & ~4095u), y);
            }
        }
    }
#else
    // The transform's swizzled layout is kept for this exchange.  Its j-shifted reads collide where
    // the run of 64 source indices crosses a swizzle boundary (2.6 % of the kernel's LDS cycles,
    // profiles/r03_v10_counters.json); the plain layout above has no conflict for any j but needs
    // one more workgroup barrier, and measured the same at m = 8192 (2108 against 2110 bootstraps/s,
    // same call) and 2.3-2.4 % slower at m = 4096 and 16384 (profiles/r04_exp_epilogue.txt): LDS
    // bandwidth is not what this kernel waits for.
    lds_store<LOGM, 2, LE, G::STOP>(z, lds, tid);
    SGFHE_SYNC();
    // Source index of output coefficient i = tid + T e is s_e = (i - j) mod 2m = s_0 + T e:
    // the swizzled low part is computed once per thread.
    {
        constexpr uint32_t LOWMASK = (1u << G::STOP) - 1u;
        const uint32_t s0 = ((uint32_t)tid - j) & (2 * M - 1);
        const uint32_t lowswz = swz<LE>(s0 & LOWMASK);
        const uint32_t h0 = s0 >> G::STOP;
#pragma unroll
        for (int e = 0; e < E; e++) {
            const uint32_t he = h0 + e;
            const uint32_t hipart = (he & (E - 1)) << G::STOP;
            // swz is XOR-linear; for m = 8192 the e bits lie above every bit it reads
            const uint32_t addr = hipart ^ lowswz ^ (G::STOP >= 9 ? 0u : swz_bits<LE>(hipart));
            const uint32_t smask = 0u - ((he >> LE) & 1u);
            const uint32_t yoe = yoff - smask;
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const uint32_t v = lds[c * M + addr];
                const uint32_t y = (v ^ smask) + (yoe - (uint32_t)z[c][e]);
                buf_st_u32(ryres, vout + ((uint32_t)(4 * T * e) & 4095u),
                              (c ? sy1 : sy0) + ((uint32_t)(4 * T * e) & ~4095u), y);
            }
        }
    }
#endif
}

// ---- k_crt_acc --------------------------------------------------------------------------------
// One thread per (bootstrap, c, coefficient).  With y'_i = (D + H') (M/p_i)^-1 mod p_i:
//   D + H' = sum_i y'_i (M/p_i) - alpha M,  alpha = floor(sum_i y'_i / p_i)
// and |D| <= 0.4 M (5 m B Q <= M is checked at ctx creation) puts the fractional part of the sum
// within 0.5 +- 0.4, so alpha is exact in float (the estimate is off by < 10^-6).  The identity
// holds for any non-negative representatives y'_i = y_i + t p_i (alpha grows by t): k_extprod
// hands over residues below 5.7 p_i (6.2 p_i for the last prime), so alpha < 6 npr + 1.  Then x'_new = (x'_old + D) mod Q and the new digits are
// (x'_new mod B, x'_new / B)  (flatten, utils.jl:155-189).
//
// Arithmetic: S = sum_i y'_i c_i + T[alpha] + x'_old < 2^35 Q.  Its quotient by Q is estimated in
// double precision (error < 1), the remainder is formed modulo 2^96 in three 32-bit limbs
// (v_mad_u64_u32 chains) and corrected by at most one +-Q; same scheme for the division by B.
struct U96 {
    uint32_t w0, w1, w2;
};
__device__ __forceinline__ void mad96(U96 &a, uint32_t y, const uint32_t (&c)[3]) {
    const uint64_t p0 = (uint64_t)y * c[0] + a.w0;
    const uint64_t p1 = (uint64_t)y * c[1] + a.w1 + (p0 >> 32);
    a.w0 = (uint32_t)p0;
    a.w1 = (uint32_t)p1;
    a.w2 = y * c[2] + a.w2 + (uint32_t)(p1 >> 32);
}
// S = sum_i y_i c_i + T[alpha] (+ hi B + lo of the previous digits) reduced modulo Q: the
// canonical residue of x'_new (or of D when there is no previous accumulator), as three limbs.
template <int NP>
__device__ __forceinline__ U96 crt_reduce(const uint32_t (&y)[NP], const CrtConst *__restrict__ CC,
                                          bool have_old, ulonglong2 d) {
    float f = 0.f;
    double Sd = 0.0;
#pragma unroll
    for (int q = 0; q < NP; q++) {
        f += (float)y[q] * CC->invp[q];
#ifndef SGFHE_ABL_CRT_NOSD  // (timing-only build without the double-precision sum: wrong results)
        Sd += (double)y[q] * CC->cd[q];
#endif
    }
    const int alpha = (int)f;
    const uint4 tv = *reinterpret_cast<const uint4 *>(CC->T32[alpha]);
    U96 a = {tv.x, tv.y, tv.z};
    Sd += CC->Td[alpha];
#pragma unroll
    for (int q = 0; q < NP; q++) mad96(a, y[q], CC->c32[q]);
    if (have_old) {  // x'_old = hi B + lo
        const uint64_t B = (uint64_t)CC->B;
        const uint32_t h0 = (uint32_t)d.y, h1 = (uint32_t)(d.y >> 32);
        const uint32_t b0 = (uint32_t)B, b1 = (uint32_t)(B >> 32);
        const uint64_t p0 = (uint64_t)h0 * b0 + a.w0 + (uint32_t)d.x;
        const uint64_t p1 = (uint64_t)h0 * b1 + (uint64_t)h1 * b0 + a.w1 + (uint32_t)(d.x >> 32) + (p0 >> 32);
        a.w0 = (uint32_t)p0;
        a.w1 = (uint32_t)p1;
        a.w2 += h1 * b1 + (uint32_t)(p1 >> 32);
#ifndef SGFHE_ABL_CRT_NOSD
        Sd += (double)d.y * CC->Bd + (double)d.x;
#endif
    }
    // subtract q1 Q, q1 = floor(S / Q) +- 1
    {
        const uint64_t q1 = (uint64_t)(Sd * CC->invQ);
        const uint32_t qa = (uint32_t)q1, qb = (uint32_t)(q1 >> 32);
        const uint64_t m0 = (uint64_t)qa * CC->Q32[0];
        const uint64_t m1 = (uint64_t)qa * CC->Q32[1] + (uint64_t)qb * CC->Q32[0] + (m0 >> 32);
        const uint32_t m2 = qa * CC->Q32[2] + qb * CC->Q32[1] + (uint32_t)(m1 >> 32);
        const uint64_t r0 = (uint64_t)a.w0 - (uint32_t)m0;
        const uint64_t r1 = (uint64_t)a.w1 - (uint32_t)m1 - ((r0 >> 32) & 1u);
        a.w2 = a.w2 - m2 - (uint32_t)((r1 >> 32) & 1u);
        a.w0 = (uint32_t)r0;
        a.w1 = (uint32_t)r1;
    }
    // the remainder is in (-Q, 2Q) modulo 2^96: one correction
    {
        const uint64_t Qlo = ((uint64_t)CC->Q32[1] << 32) | CC->Q32[0];
        const uint32_t Qhi = CC->Q32[2];
        uint64_t lo = ((uint64_t)a.w1 << 32) | a.w0;
        uint32_t hi = a.w2;
        if ((int32_t)hi < 0) {  // negative: add Q
            const uint64_t nlo = lo + Qlo;
            hi = hi + Qhi + (nlo < lo);
            lo = nlo;
        } else if (hi > Qhi || (hi == Qhi && lo >= Qlo)) {
            const uint64_t nlo = lo - Qlo;
            hi = hi - Qhi - (lo < Qlo);
            lo = nlo;
        }
        a.w0 = (uint32_t)lo;
        a.w1 = (uint32_t)(lo >> 32);
        a.w2 = hi;
    }
    return a;
}

template <int NP, bool ROWS = false>
__global__ void __launch_bounds__(256)
k_crt_acc(const uint32_t *__restrict__ yres, uint64_t *__restrict__ dig,
          const CrtConst *__restrict__ CC, uint32_t total, uint32_t logm, uint32_t mode,
          RndArgs ra, uint32_t iter) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const uint32_t M = 1u << logm;
    const uint32_t i = t & (M - 1);
    const uint32_t bc = t >> logm;
    const uint32_t yo = 4u * ((bc * NP << logm) + i);  // byte offset
    uint32_t y[NP];
#pragma unroll
#ifdef SGFHE_ABL_NO_YRES  // timing-only build: no residue loads (wrong results)
    for (int q = 0; q < NP; q++) y[q] = (t * 2654435761u + (uint32_t)q * 40503u) >> 3;
#else
    for (int q = 0; q < NP; q++) y[q] = ld_off<uint32_t>(yres, yo + ((uint32_t)(4 * q) << logm));
#endif
    const bool have_old = !(mode & MODE_NOACC);
    const bool wide = (mode & MODE_WIDE) != 0;
    const ulonglong2 d = have_old ? load_digits(dig, bc, i, M, wide) : make_ulonglong2(0, 0);
#ifdef SGFHE_ABL_CRT_MEMONLY  // timing-only build: every load and store, no arithmetic (wrong results)
    {
        uint32_t acc = 0;
#pragma unroll
        for (int q = 0; q < NP; q++) acc ^= y[q];
        store_digits(dig, bc, i, M, d.x ^ acc, d.y + acc);
        return;
    }
#endif
    const U96 a = crt_reduce(y, CC, have_old, d);
    const uint64_t xlo = ((uint64_t)a.w1 << 32) | a.w0;
    if (mode & MODE_CANON) {  // canonical residues, interleaved {lo, hi} words
        reinterpret_cast<ulonglong2 *>(dig)[t] = make_ulonglong2(xlo, (uint64_t)a.w2);
        return;
    }
    if (mode & MODE_RANDOM) {
        ChaChaKey key;
        uint32_t cz, cw;
        rnd_stream<ROWS>(ra, bc >> 1, key, cz, cw);
        const uint4 ctr = make_uint4(((bc & 1u) << logm) + i, iter, cz, cw);
        const ulonglong2 e = random_digits(((u128)a.w2 << 64) | xlo, CC, key, ctr);
        store_digits(dig, bc, i, M, e.x, e.y, wide);
        return;
    }
    // digits: hi = x' / B (double estimate +- 1), lo = x' - hi B (exact modulo 2^64)
    const uint64_t B = (uint64_t)CC->B;
    const double xd = (double)a.w2 * 18446744073709551616.0 + (double)xlo;
    uint64_t hq = (uint64_t)(xd * CC->invB);
    int64_t lo = (int64_t)(xlo - hq * B);
    if (lo < 0) { lo += (int64_t)B; hq--; }
    else if ((uint64_t)lo >= B) { lo -= (int64_t)B; hq++; }
    store_digits(dig, bc, i, M, (uint64_t)lo, hq);
}

// The k-loop's own case (deterministic flatten, previous accumulator present) with two adjacent
// coefficients per thread: 8-byte residue and digit-word loads / stores and 4-byte loads / stores
// of the 16-bit high words -- half the memory instructions for the same bytes.  Same arithmetic.
template <int NP>
__global__ void __launch_bounds__(256)
k_crt_acc2(const uint32_t *__restrict__ yres, uint64_t *__restrict__ dig,
           const CrtConst *__restrict__ CC, uint32_t pairs, uint32_t logm) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= pairs) return;
    const uint32_t M = 1u << logm;
    const uint32_t i = (2u * t) & (M - 1);  // even
    const uint32_t bc = (2u * t) >> logm;
    const uint32_t yo = 4u * ((bc * NP << logm) + i);
    uint2 yv[NP];
#pragma unroll
    for (int q = 0; q < NP; q++) yv[q] = ld_off<uint2>(yres, yo + ((uint32_t)(4 * q) << logm));
    const uint32_t rec = bc * 16u * M;
    const uint32_t ol = rec + 4u * i, oh = rec + 8u * M + 2u * i;
    const uint2 l0 = ld_off<uint2>(dig, ol), l1 = ld_off<uint2>(dig, ol + 4u * M);
    const uint32_t h0 = ld_off<uint32_t>(dig, oh), h1 = ld_off<uint32_t>(dig, oh + 2u * M);
#ifdef SGFHE_ABL_CRT_MEMONLY  // timing-only build: every load and store, no arithmetic (wrong results)
    {
        uint32_t ax = 0, ay = 0;
#pragma unroll
        for (int q = 0; q < NP; q++) { ax ^= yv[q].x; ay += yv[q].y; }
        st_off<uint2>(dig, ol, make_uint2(l0.x ^ ax, l0.y + ay));
        st_off<uint2>(dig, ol + 4u * M, make_uint2(l1.x + ay, l1.y ^ ax));
        st_off<uint32_t>(dig, oh, h0 ^ ax);
        st_off<uint32_t>(dig, oh + 2u * M, h1 + ay);
        return;
    }
#endif
    const uint64_t B = (uint64_t)CC->B;
    uint32_t olo[2][2], ohi[2][2];  // [digit][coefficient]
#pragma unroll
    for (int j = 0; j < 2; j++) {
        uint32_t y[NP];
#pragma unroll
        for (int q = 0; q < NP; q++) y[q] = j ? yv[q].y : yv[q].x;
        const uint32_t hw0 = j ? (h0 >> 16) : (h0 & 0xFFFFu), hw1 = j ? (h1 >> 16) : (h1 & 0xFFFFu);
        const ulonglong2 d = make_ulonglong2((j ? l0.y : l0.x) | ((uint64_t)hw0 << 32),
                                             (j ? l1.y : l1.x) | ((uint64_t)hw1 << 32));
        const U96 a = crt_reduce(y, CC, true, d);
        const uint64_t xlo = ((uint64_t)a.w1 << 32) | a.w0;
        const double xd = (double)a.w2 * 18446744073709551616.0 + (double)xlo;
        uint64_t hq = (uint64_t)(xd * CC->invB);
        int64_t lo = (int64_t)(xlo - hq * B);
        if (lo < 0) { lo += (int64_t)B; hq--; }
        else if ((uint64_t)lo >= B) { lo -= (int64_t)B; hq++; }
        olo[0][j] = (uint32_t)lo; ohi[0][j] = (uint32_t)((uint64_t)lo >> 32);
        olo[1][j] = (uint32_t)hq; ohi[1][j] = (uint32_t)(hq >> 32);
    }
    st_off<uint2>(dig, ol, make_uint2(olo[0][0], olo[0][1]));
    st_off<uint2>(dig, ol + 4u * M, make_uint2(olo[1][0], olo[1][1]));
    st_off<uint32_t>(dig, oh, (ohi[0][0] & 0xFFFFu) | (ohi[0][1] << 16));
    st_off<uint32_t>(dig, oh + 2u * M, (ohi[1][0] & 0xFFFFu) | (ohi[1][1] << 16));
}

// ---- k_crt_lean: the k-loop's CRT + accumulate + flatten in integer arithmetic ---------------------
// Same function as k_crt_acc2 (deterministic flatten, previous accumulator present) at less than
// half the vector instructions: the kernel is HBM-bound on its own, but on the second lane it runs
// beside the other chunk's k_extprod, which is bound by vector-ALU issue -- there every instruction
// of this kernel is paid in full (profiles/r03_exp_crt_ceiling.txt).  No floating point, no table:
//   alpha = (sum_i y_i w_i) >> 58,  w_i = floor(2^58 / p_i)           (5 multiply-adds)
//   S = sum_i y'_i c_i + alpha (-M mod Q) + hi_old B + lo_old  <  2^34.5 Q,  y'_i = y_i except
//     y'_last = y_last - (p_last - 1) / 2: that takes H' = (M / p_last) (p_last - 1) / 2 out again
//     as NL sums L_k over 29-bit limbs of the constants: every product is below 2^61, so a limb sum
//     is one chain of v_mad_u64_u32 with no carry handling (S = sum_k L_k 2^(29 k))
//   q = ((S >> t) mq) >> 72 with mq = floor(2^(t + 72) / Q) < 2^44, t = bits(Q) - 29, S >> t taken
//     from the top three limb sums: q is floor(S / Q) or one less
//   x = S - q Q modulo 2^96, in [0, 2 Q): one conditional subtraction
//   hq = ((x >> t2) mb) >> (bits(B) + 51 - t2), mb = floor(2^(bits(B) + 51) / B) <= 2^52:
//     floor(x / B) or one less;  lo = x - hq B modulo 2^64, in [0, 2 B): one conditional subtraction.
// tests/rns_model.py::CrtLean restates this with every intermediate checked against its register
// width and both estimates checked against the exact quotients.
// Four adjacent coefficients per thread: 16-byte residue and digit-word accesses, 8-byte accesses
// of the 16-bit high words.
struct CrtLean {
    uint32_t c[NPR_MAX][4];   // (M / p_i) mod Q in 29-bit limbs
    uint32_t w[NPR_MAX];      // floor(2^58 / p_i)
    uint32_t cMn[4];          // (-M) mod Q in 29-bit limbs
    uint32_t hoff;            // (p_last - 1) / 2, the offset k_extprod adds to the last prime's residue
    uint32_t Qw[3];           // Q in 32-bit words
    uint32_t B0, B1;          // B = B0 + B1 2^29
    uint32_t Bw0, Bw1;        // B in 32-bit words
    uint32_t mq0, mq1;        // floor(2^(t + 72) / Q)
    uint32_t mb0, mb1;        // floor(2^(bits(B) + 51) / B)
    uint32_t a;               // top limb sum << a, the next >> 29 - a, the next >> 58 - a
    uint32_t t2, sB;          // x >> t2;  final shift of the digit estimate
    uint32_t nl;              // limbs in use (2, 3 or 4); 0 = parameter set outside this kernel's bounds
    uint32_t cR[4];           // randomised flatten: (-2 xmax (1 + B)) mod Q in 29-bit limbs
    uint32_t xm2lo, xm2hi;    //   2 xmax
};

// (x m) >> 64 for x < 2^63.5 and m = m1 2^32 + m0 with m1 <= 2^20
__device__ __forceinline__ uint64_t mulhi64_lean(uint64_t x, uint32_t m0, uint32_t m1) {
    const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32);
    const uint64_t t1 = (uint64_t)x0 * m1 + __umulhi(x0, m0);
    const uint64_t t2 = (uint64_t)x1 * m0 + t1;
    return (uint64_t)x1 * m1 + (t2 >> 32);
}

template <int NP, int NL, bool RND = false>
__device__ __forceinline__ void crt_lean_one(const uint32_t (&y)[NP], uint64_t lo_o, uint64_t hi_o,
                                             const CrtLean *__restrict__ K, uint64_t &lo_n,
                                             uint64_t &hi_n) {
    // alpha
    uint64_t acc = 0;
#pragma unroll
    for (int q = 0; q < NP; q++) acc += (uint64_t)y[q] * K->w[q];
    const uint32_t alpha = (uint32_t)(acc >> 58);
    // limb sums
    const uint32_t ylast = y[NP - 1] - K->hoff;   // >= 0: the residue proper is non-negative
    uint64_t L[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) {
        uint64_t l = k == 0 ? lo_o : 0;
#pragma unroll
        for (int q = 0; q < NP; q++) l += (uint64_t)(q == NP - 1 ? ylast : y[q]) * K->c[q][k];
        L[k] = l + (uint64_t)alpha * K->cMn[k];
        if constexpr (RND) L[k] += K->cR[k];
    }
    {   // + hi_old B
        const uint32_t h0 = (uint32_t)hi_o, h1 = (uint32_t)(hi_o >> 32);
        L[0] += (uint64_t)h0 * K->B0;
        L[1] += (uint64_t)h0 * K->B1;
        L[1] += ((uint64_t)h1 * K->B0) << 3;
        if constexpr (NL >= 3) L[2] += ((uint64_t)h1 * K->B1) << 3;
    }
    // quotient by Q
    const uint32_t a = K->a;
    uint64_t X = (L[NL - 1] << a) + (L[NL - 2] >> (29 - a));
    if constexpr (NL >= 3) X += L[NL - 3] >> (58 - a);
    const uint64_t q1 = mulhi64_lean(X, K->mq0, K->mq1) >> 8;
    // S modulo 2^96
    uint64_t slo = L[0];
    uint32_t shi = 0;
    {
        const uint64_t v = L[1] << 29;
        slo += v;
        shi += (uint32_t)(L[1] >> 35) + (slo < v);
    }
    if constexpr (NL >= 3) {
        const uint64_t v = L[2] << 58;
        slo += v;
        shi += (uint32_t)(L[2] >> 6) + (slo < v);
    }
    if constexpr (NL >= 4) shi += (uint32_t)L[3] << 23;
    // x = S - q1 Q modulo 2^96
    uint64_t xlo;
    uint32_t xhi;
    {
        const uint32_t qa = (uint32_t)q1, qb = (uint32_t)(q1 >> 32);
        const uint64_t m0 = (uint64_t)qa * K->Qw[0];
        const uint64_t m1 = (uint64_t)qa * K->Qw[1] + (m0 >> 32) + (uint64_t)qb * K->Qw[0];  // modulo 2^64
        const uint32_t m2 = qa * K->Qw[2] + qb * K->Qw[1] + (uint32_t)(m1 >> 32);
        const uint64_t mlo = (m1 << 32) | (uint32_t)m0;
        xlo = slo - mlo;
        xhi = shi - m2 - (slo < mlo);
    }
    {   // in [0, 2 Q): subtract Q when that does not borrow
        const uint64_t Qlo = ((uint64_t)K->Qw[1] << 32) | K->Qw[0];
        const uint64_t dlo = xlo - Qlo;
        const uint32_t bor = xlo < Qlo;
        const uint32_t dhi = xhi - K->Qw[2] - bor;
        const bool below = xhi < K->Qw[2] || (xhi == K->Qw[2] && bor);   // x < Q
        xlo = below ? xlo : dlo;
        xhi = below ? xhi : dhi;
    }
    // digits
    const uint32_t t2 = K->t2;
    const uint64_t X2 = t2 ? ((xlo >> t2) | ((uint64_t)xhi << (64 - t2))) : xlo;
    uint64_t hq = mulhi64_lean(X2, K->mb0, K->mb1) >> K->sB;
    const uint64_t Bv = ((uint64_t)K->Bw1 << 32) | K->Bw0;
    uint64_t lo = xlo - hq * Bv;   // modulo 2^64: in [0, 2 B)
    if (lo >= Bv) { lo -= Bv; hq += 1; }
    lo_n = lo;
    hi_n = hq;
}

template <int NP, int NL>
__global__ void __launch_bounds__(256)
k_crt_lean(const uint32_t *__restrict__ yres, uint64_t *__restrict__ dig,
           const CrtLean *__restrict__ K, uint32_t quads, uint32_t logm) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= quads) return;
    const uint32_t M = 1u << logm;
    const uint32_t i = (4u * t) & (M - 1);  // multiple of 4
    const uint32_t bc = (4u * t) >> logm;
    const uint32_t yo = 4u * ((bc * NP << logm) + i);
    // The residues are read exactly once: non-temporal loads keep them from displacing what the other
    // lane's k_extprod re-reads from the caches (key slice, digit planes).  Same call, two lanes of
    // 256 (profiles/r03_exp_cache_policy.txt): 2076 against 2016 bootstraps/s, +3.0 %, and +2.9 % on
    // a second box; the old digits loaded the same way as well: +2.2 % (they are in the caches, the
    // external product of this chunk has just read them), so those stay plain loads.
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    uint4 yv[NP];
#pragma unroll
    for (int q = 0; q < NP; q++) {
#ifdef SGFHE_ABL_NO_YLOAD       // timing-only build: no residue loads (wrong results)
        yv[q] = make_uint4(t + q, yo, i + 7u * q, bc);
#elif defined(SGFHE_CRT_PLAIN_LOADS)  // (A/B builds)
        yv[q] = ld_off<uint4>(yres, yo + ((uint32_t)(4 * q) << logm));
#else
        const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(
            reinterpret_cast<const char *>(yres) + yo + ((uint32_t)(4 * q) << logm)));
        yv[q] = make_uint4(t.x, t.y, t.z, t.w);
#endif
    }
    const uint32_t rec = bc * 16u * M;
    const uint32_t ol = rec + 4u * i, oh = rec + 8u * M + 2u * i;
    const uint4 l0 = ld_off<uint4>(dig, ol), l1 = ld_off<uint4>(dig, ol + 4u * M);
    const uint2 h0 = ld_off<uint2>(dig, oh), h1 = ld_off<uint2>(dig, oh + 2u * M);
    const uint32_t l0w[4] = {l0.x, l0.y, l0.z, l0.w}, l1w[4] = {l1.x, l1.y, l1.z, l1.w};
    const uint32_t h0w[4] = {h0.x & 0xFFFFu, h0.x >> 16, h0.y & 0xFFFFu, h0.y >> 16};
    const uint32_t h1w[4] = {h1.x & 0xFFFFu, h1.x >> 16, h1.y & 0xFFFFu, h1.y >> 16};
    uint64_t nlo[4], nhi[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint32_t y[NP];
#pragma unroll
        for (int q = 0; q < NP; q++) y[q] = j == 0 ? yv[q].x : j == 1 ? yv[q].y : j == 2 ? yv[q].z : yv[q].w;
        crt_lean_one<NP, NL>(y, l0w[j] | ((uint64_t)h0w[j] << 32), l1w[j] | ((uint64_t)h1w[j] << 32), K,
                             nlo[j], nhi[j]);
    }
    st_off<uint4>(dig, ol, make_uint4((uint32_t)nlo[0], (uint32_t)nlo[1], (uint32_t)nlo[2], (uint32_t)nlo[3]));
    st_off<uint4>(dig, ol + 4u * M,
                  make_uint4((uint32_t)nhi[0], (uint32_t)nhi[1], (uint32_t)nhi[2], (uint32_t)nhi[3]));
    st_off<uint2>(dig, oh, make_uint2((uint32_t)(nlo[0] >> 32) | ((uint32_t)(nlo[1] >> 32) << 16),
                                      (uint32_t)(nlo[2] >> 32) | ((uint32_t)(nlo[3] >> 32) << 16)));
    st_off<uint2>(dig, oh + 2u * M, make_uint2((uint32_t)(nhi[0] >> 32) | ((uint32_t)(nhi[1] >> 32) << 16),
                                               (uint32_t)(nhi[2] >> 32) | ((uint32_t)(nhi[3] >> 32) << 16)));
}

// The same with ONE coefficient per thread, for the latency form of the k-loop (a call of a few gates):
// there the kernel is a dependent link of the launch chain, one wave per SIMD, and what counts is the
// length of a thread's instruction stream (117 instead of 468), not the width of its memory accesses.
template <int NP, int NL>
__global__ void __launch_bounds__(256)
k_crt_lean1(const uint32_t *__restrict__ yres, uint64_t *__restrict__ dig,
            const CrtLean *__restrict__ K, uint32_t total, uint32_t logm) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const uint32_t M = 1u << logm;
    const uint32_t i = t & (M - 1);
    const uint32_t bc = t >> logm;
    const uint32_t yo = 4u * ((bc * NP << logm) + i);
    uint32_t y[NP];
#pragma unroll
    for (int q = 0; q < NP; q++) y[q] = ld_off<uint32_t>(yres, yo + ((uint32_t)(4 * q) << logm));
    const ulonglong2 d = load_digits(dig, bc, i, M);
    uint64_t lo, hi;
    crt_lean_one<NP, NL>(y, d.x, d.y, K, lo, hi);
    store_digits(dig, bc, i, M, lo, hi);
}

// The randomised flatten (utils.jl:198-241; random_digits above) through the same limb sums.  With
// r_i the draws in [0, 2 xmax] and (e_lo, e_hi) the old stored digits, the value to flatten is
//   x2 = (x_old + D - r_0 - r_1 B) mod Q,   x_old == e_hi B + e_lo  (mod Q),
// so the draws enter as 2 xmax - r_i >= 0 added to the old digits, the constant
// cR = (-2 xmax (1 + B)) mod Q puts the shift back, and the one quotient step of crt_lean_one gives
// x2 directly -- no reduction of r_1 B + r_0, no second quotient.  New stored digits: those of x2
// plus r_i.  Same stream addressing as k_crt_acc (random_digits), bit-identical output (tests: the oracle; the
// register widths: tests/rns_model.py CrtLean.digits_random).
template <int NP, int NL, bool WIDE, bool ROWS = false>
__global__ void __launch_bounds__(256)
k_crt_lean_rnd(const uint32_t *__restrict__ yres, uint64_t *__restrict__ dig,
               const CrtLean *__restrict__ K, uint32_t quads, uint32_t logm, RndArgs ra, uint32_t iter) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= quads) return;
    const uint32_t M = 1u << logm;
    const uint32_t i = (4u * t) & (M - 1);  // multiple of 4
    const uint32_t bc = (4u * t) >> logm;
    const uint32_t yo = 4u * ((bc * NP << logm) + i);
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    uint4 yv[NP];
#pragma unroll
    for (int q = 0; q < NP; q++) {
        const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(
            reinterpret_cast<const char *>(yres) + yo + ((uint32_t)(4 * q) << logm)));
        yv[q] = make_uint4(t.x, t.y, t.z, t.w);
    }
    const uint32_t rec = bc * 16u * M;
    const uint32_t ol = rec + 4u * i, oh = rec + 8u * M + 2u * i, ot = rec + 12u * M + i;
    const uint4 l0 = ld_off<uint4>(dig, ol), l1 = ld_off<uint4>(dig, ol + 4u * M);
    const uint2 h0 = ld_off<uint2>(dig, oh), h1 = ld_off<uint2>(dig, oh + 2u * M);
    uint32_t t0 = 0, t1 = 0;   // third plane: bits 48..55 of the four stored digits
    if constexpr (WIDE) { t0 = ld_off<uint32_t>(dig, ot); t1 = ld_off<uint32_t>(dig, ot + M); }
    const uint32_t l0w[4] = {l0.x, l0.y, l0.z, l0.w}, l1w[4] = {l1.x, l1.y, l1.z, l1.w};
    const uint32_t h0w[4] = {h0.x & 0xFFFFu, h0.x >> 16, h0.y & 0xFFFFu, h0.y >> 16};
    const uint32_t h1w[4] = {h1.x & 0xFFFFu, h1.x >> 16, h1.y & 0xFFFFu, h1.y >> 16};
    const uint64_t xm2 = ((uint64_t)K->xm2hi << 32) | K->xm2lo, span = xm2 + 1;
    const uint32_t cx = ((bc & 1u) << logm) + i;
    ChaChaKey key;
    uint32_t cz, cw;
    rnd_stream<ROWS>(ra, bc >> 1, key, cz, cw);
    uint32_t rw[16];   // the draws of the thread's four coefficients: one block of the stream
    chacha_block<SGFHE_RND_ROUNDS>(key, cx >> 2, iter, cz, cw, rw);
    uint64_t nlo[4], nhi[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint32_t y[NP];
#pragma unroll
        for (int q = 0; q < NP; q++) y[q] = j == 0 ? yv[q].x : j == 1 ? yv[q].y : j == 2 ? yv[q].z : yv[q].w;
        const uint64_t r0 = __umul64hi(((uint64_t)rw[4 * j + 1] << 32) | rw[4 * j], span);
        const uint64_t r1 = __umul64hi(((uint64_t)rw[4 * j + 3] << 32) | rw[4 * j + 2], span);
        uint32_t hw0 = h0w[j], hw1 = h1w[j];
        if constexpr (WIDE) {
            hw0 |= ((t0 >> (8 * j)) & 0xFFu) << 16;
            hw1 |= ((t1 >> (8 * j)) & 0xFFu) << 16;
        }
        const uint64_t e_lo = l0w[j] | ((uint64_t)hw0 << 32), e_hi = l1w[j] | ((uint64_t)hw1 << 32);
        crt_lean_one<NP, NL, true>(y, e_lo + (xm2 - r0), e_hi + (xm2 - r1), K, nlo[j], nhi[j]);
        nlo[j] += r0;
        nhi[j] += r1;
    }
    st_off<uint4>(dig, ol, make_uint4((uint32_t)nlo[0], (uint32_t)nlo[1], (uint32_t)nlo[2], (uint32_t)nlo[3]));
    st_off<uint4>(dig, ol + 4u * M,
                  make_uint4((uint32_t)nhi[0], (uint32_t)nhi[1], (uint32_t)nhi[2], (uint32_t)nhi[3]));
    st_off<uint2>(dig, oh, make_uint2(((uint32_t)(nlo[0] >> 32) & 0xFFFFu) | ((uint32_t)(nlo[1] >> 32) << 16),
                                      ((uint32_t)(nlo[2] >> 32) & 0xFFFFu) | ((uint32_t)(nlo[3] >> 32) << 16)));
    st_off<uint2>(dig, oh + 2u * M, make_uint2(((uint32_t)(nhi[0] >> 32) & 0xFFFFu) | ((uint32_t)(nhi[1] >> 32) << 16),
                                               ((uint32_t)(nhi[2] >> 32) & 0xFFFFu) | ((uint32_t)(nhi[3] >> 32) << 16)));
    if constexpr (WIDE) {
        uint32_t w0 = 0, w1 = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            w0 |= ((uint32_t)(nlo[j] >> 48) & 0xFFu) << (8 * j);
            w1 |= ((uint32_t)(nhi[j] >> 48) & 0xFFu) << (8 * j);
        }
        st_off<uint32_t>(dig, ot, w0);
        st_off<uint32_t>(dig, ot + M, w1);
    }
}

// The randomised flatten with ONE coefficient per thread, for the latency form (a call of a few gates): every
// thread computes the ChaCha block of its quad of coefficients -- four times the block function across the
// launch, which an otherwise idle device does not notice -- and uses its own four words of it, so a thread's
// instruction stream is one block and one CRT instead of one block and four.  Same stream addressing, same
// digits as k_crt_lean_rnd.  QUARTER: residues from the partial values of the quarter kernels.  Not for the
// three-plane digit records (B >= 2^46).
template <int NP, int NL, bool QUARTER, bool ROWS = false>
__global__ void __launch_bounds__(256)
k_crt_lean_rnd1(const uint32_t *__restrict__ yres, uint64_t *__restrict__ dig,
                const CrtLean *__restrict__ K, uint32_t total, uint32_t logm, RndArgs ra, uint32_t iter,
                const PrimeK *__restrict__ PS) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const uint32_t M = 1u << logm;
    const uint32_t i = t & (M - 1);
    const uint32_t bc = t >> logm;
    uint32_t y[NP];
    if constexpr (QUARTER) {
        const uint32_t MS = M >> 2, jq = i & (MS - 1), qq = i >> (logm - 2);
        const int32_t *ypart = reinterpret_cast<const int32_t *>(yres);
#pragma unroll
        for (int q = 0; q < NP; q++) {
            const PrimeK P = PS[q];
            const Mod md = mod_of(P);
            const int32_t *yp = ypart + (((size_t)bc * NP + q) * 4) * MS + jq;
            int32_t r;
            if (qq & 1u) {
                const int32_t c1 = smont(yp[0] - yp[MS], P.v2, md), c3 = smont(yp[2 * MS] - yp[3 * MS], P.v3, md);
                r = (qq & 2u) ? smont(c1 - c3, P.v1, md) : c1 + c3;
            } else {
                const int32_t c0 = sred(yp[0] + yp[MS], md), c2 = sred(yp[2 * MS] + yp[3 * MS], md);
                r = (qq & 2u) ? smont(c0 - c2, P.v1, md) : c0 + c2;
            }
            y[q] = (uint32_t)(r + 3 * P.p) + P.hoff;
        }
    } else {
        const uint32_t yo = 4u * ((bc * NP << logm) + i);
#pragma unroll
        for (int q = 0; q < NP; q++) y[q] = ld_off<uint32_t>(yres, yo + ((uint32_t)(4 * q) << logm));
    }
    const ulonglong2 d = load_digits(dig, bc, i, M);
    const uint64_t xm2 = ((uint64_t)K->xm2hi << 32) | K->xm2lo, span = xm2 + 1;
    const uint32_t cx = ((bc & 1u) << logm) + i;
    ChaChaKey key;
    uint32_t cz, cw;
    rnd_stream<ROWS>(ra, bc >> 1, key, cz, cw);
    uint32_t rw[16];
    chacha_block<SGFHE_RND_ROUNDS>(key, cx >> 2, iter, cz, cw, rw);
    const uint32_t sel = cx & 3u;   // this coefficient's four words of the block
    const uint32_t w0 = sel == 0 ? rw[0] : sel == 1 ? rw[4] : sel == 2 ? rw[8] : rw[12];
    const uint32_t w1 = sel == 0 ? rw[1] : sel == 1 ? rw[5] : sel == 2 ? rw[9] : rw[13];
    const uint32_t w2 = sel == 0 ? rw[2] : sel == 1 ? rw[6] : sel == 2 ? rw[10] : rw[14];
    const uint32_t w3 = sel == 0 ? rw[3] : sel == 1 ? rw[7] : sel == 2 ? rw[11] : rw[15];
    const uint64_t r0 = __umul64hi(((uint64_t)w1 << 32) | w0, span);
    const uint64_t r1 = __umul64hi(((uint64_t)w3 << 32) | w2, span);
    uint64_t lo, hi;
    crt_lean_one<NP, NL, true>(y, d.x + (xm2 - r0), d.y + (xm2 - r1), K, lo, hi);
    store_digits(dig, bc, i, M, lo + r0, hi + r1);
}

// ---- one launch per iteration: the residues never leave the compute unit (round 4, PROTOTYPE) -------------------
// Compiled only with -DSGFHE_WITH_ITER_ALL (tools/exp_r4_iter_all.sh); selected at run time by SGFHE_ITER_ALL=1.
// Bit-exact (tests/test_gpu_parity.py::test_params1024_vs_oracle under that switch), and 17 % SLOWER than the two
// kernels it replaces (1737 against 2083 bootstraps/s, same call, profiles/r04_exp_iter_all.txt): one workgroup of
// 1024 threads per compute unit means every one of its ~135 workgroup barriers per iteration stalls the whole
// compute unit, where two co-resident k_extprod workgroups fill each other's stalls; 8 points per thread cost a
// fifth LDS pass per transform; and the register file (128 per thread) holds the transform's working set beside
// 56 accumulators only, so column 1 of three primes lives in LDS.  Kept as the starting point for a design that
// removes the hand-off without giving up two independent instruction streams per compute unit.
#ifdef SGFHE_WITH_ITER_ALL
// k_extprod hands 20 bytes per coefficient to the CRT kernel through memory, and that hand-off costs 11 % of the
// path's throughput -- as clock: the socket is at its power limit (profiles/r04_exp_clock_handoff.txt).  Here one
// workgroup of m / 8 threads owns a bootstrap with ALL its primes: every thread keeps the NTT-domain sums of both
// product columns for every prime in registers (NP x 2 x 8), runs the forward transforms prime by prime inside the
// phase loop, then per prime the inverse pair and the rotation in place, and ends with the CRT / accumulate /
// flatten of its own 16 coefficients (crt_lean_one) -- no yres, no second kernel.  8 points per thread (the
// register file holds 128 registers for each of 1024 threads); deterministic flatten, NP <= 5.
// The prime index is a template parameter (recursion over PI) so that the accumulator arrays are only ever
// indexed by constants: a `#pragma unroll` inside the rolled phase loop is not honoured, and a dynamically
// indexed array lives in scratch memory.  Column 1 of the first LP primes accumulates in LDS (private words, as
// k_extprod's z1), the rest in registers.
template <int LOGM, int NP, int LP>
struct IterAll {
    static constexpr int LE = 3;
    using G = NttGeom<LOGM, LE>;
    static constexpr int M = G::M, T = G::T, E = G::E;
    // private LDS word e of prime pi's column 1 (pi < LP): behind the 2 m words of the exchange buffer
    static __device__ __forceinline__ int32_t *priv(uint32_t *lds, int pi, int tid) {
        return reinterpret_cast<int32_t *>(lds) + (2 + pi) * M + tid;
    }
    template <int PI>
    static __device__ __forceinline__ void forward(int32_t (&a0)[NP][E], int32_t (&a1)[NP - LP][E],
                                                   const uint64_t *__restrict__ dig, uint32_t b,
                                                   const int32_t *__restrict__ keyk, PrimeSet PS, uint32_t *lds, int ph) {
        if constexpr (PI < NP) {
            const PrimeK P = PS[PI];
            const Mod md = mod_of(P);
            // addresses behind an opaque zero: the loads of this prime are not hoisted over the work of the one before
            const int tid = (int)threadIdx.x + (int)opaque_zero();
            const uint32_t *dl = digit_lo_plane(dig, (size_t)b * 2 + (ph >> 1), M) + (ph & 1) * M;
            const uint16_t *dh = digit_hi_plane(dig, (size_t)b * 2 + (ph >> 1), M) + (ph & 1) * M;
            int32_t x[1][E];
#pragma unroll
            for (int e = 0; e < E; e++)
                x[0][e] = digit_reduce(dl[tid + T * e] | ((uint64_t)dh[tid + T * e] << 32), md, P.sR);
            SGFHE_SYNC();   // the exchange buffer is reused
            ntt_forward<LOGM, 1, LE>(x, lds, P.twf, tid, md);
            const int32_t *kp = keyk + ((size_t)PI * 8 + ph * 2) * M + E * (tid + (int)opaque_zero());
#pragma unroll
            for (int h = 0; h < E / 4; h++) {
                const int4 a = reinterpret_cast<const int4 *>(kp)[h];
                const int4 bq = reinterpret_cast<const int4 *>(kp + M)[h];
                const int32_t ka[4] = {a.x, a.y, a.z, a.w}, kb[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int e = 4 * h + t;
                    a0[PI][e] += smont(x[0][e], ka[t], md);   // four phases: < 2.99 * 2^29
                    if constexpr (PI < LP) priv(lds, PI, tid)[e * T] += smont(x[0][e], kb[t], md);
                    else a1[PI - LP][e] += smont(x[0][e], kb[t], md);
                }
            }
            forward<PI + 1>(a0, a1, dig, b, keyk, PS, lds, ph);
        }
    }
    // inverse pair, rotation and residue form of prime PI, back into the accumulators' places
    template <int PI>
    static __device__ __forceinline__ void inverse(int32_t (&a0)[NP][E], int32_t (&a1)[NP - LP][E], PrimeSet PS,
                                                   uint32_t *lds, uint32_t j) {
        if constexpr (PI < NP) {
            const PrimeK P = PS[PI];
            const Mod md = mod_of(P);
            const int tid = (int)threadIdx.x + (int)opaque_zero();
            int32_t z[2][E];
#pragma unroll
            for (int e = 0; e < E; e++) {
                z[0][e] = sred(a0[PI][e], md);
                if constexpr (PI < LP) z[1][e] = sred(priv(lds, PI, tid)[e * T], md);
                else z[1][e] = sred(a1[PI - LP][e], md);
            }
            SGFHE_SYNC();
            ntt_inverse<LOGM, 2, LE>(z, lds, P.twi, tid, md);   // |.| < 1.4 * 2^29
            lds_store<LOGM, 2, LE, G::STOP>(z, lds, tid);
            SGFHE_SYNC();
            constexpr uint32_t LOWMASK = (1u << G::STOP) - 1u;
            const uint32_t s0 = ((uint32_t)tid - j) & (2 * M - 1);
            const uint32_t lowswz = swz<LE>(s0 & LOWMASK);
            const uint32_t h0 = s0 >> G::STOP;
            const uint32_t yoff = 3u * (uint32_t)P.p + P.hoff;
#pragma unroll
            for (int e = 0; e < E; e++) {
                const uint32_t he = h0 + e;
                const uint32_t hipart = (he & (E - 1)) << G::STOP;
                const uint32_t addr = hipart ^ lowswz ^ (G::STOP >= 9 ? 0u : swz_bits<LE>(hipart));
                const uint32_t smask = 0u - ((he >> LE) & 1u);
                const uint32_t yoe = yoff - smask;
                // the residues k_extprod stores
                a0[PI][e] = (int32_t)((lds[addr] ^ smask) + (yoe - (uint32_t)z[0][e]));
                const int32_t y1 = (int32_t)((lds[M + addr] ^ smask) + (yoe - (uint32_t)z[1][e]));
                if constexpr (PI < LP) priv(lds, PI, tid)[e * T] = y1;
                else a1[PI - LP][e] = y1;
            }
            inverse<PI + 1>(a0, a1, PS, lds, j);
        }
    }
};

template <int LOGM, int NP, int NL>
__global__ void __launch_bounds__((NttGeom<LOGM, 3>::T))
k_iter_all(uint64_t *__restrict__ dig, const int32_t *__restrict__ keyk, const uint32_t *__restrict__ ua,
           PrimeSet PS, const CrtLean *__restrict__ K, uint32_t k, uint32_t n) {
#ifndef SGFHE_IA_LP
#define SGFHE_IA_LP 3
#endif
    constexpr int LP = SGFHE_IA_LP;
    using IA = IterAll<LOGM, NP, LP>;
    constexpr int M = IA::M, T = IA::T, E = IA::E;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];   // (2 + LP) m words
    const uint32_t b = blockIdx.x;
    int32_t a0[NP][E], a1[NP - LP][E];
#pragma unroll
    for (int e = 0; e < E; e++) {
#pragma unroll
        for (int q = 0; q < NP; q++) a0[q][e] = 0;
#pragma unroll
        for (int q = 0; q < NP - LP; q++) a1[q][e] = 0;
#pragma unroll
        for (int q = 0; q < LP; q++) IA::priv(lds, q, (int)threadIdx.x)[e * T] = 0;
    }
#pragma unroll 1
    for (int ph = 0; ph < 4; ph++) IA::template forward<0>(a0, a1, dig, b, keyk, PS, lds, ph);
    IA::template inverse<0>(a0, a1, PS, lds, ua[(size_t)b * n + k]);
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int e = 0; e < E; e++) {
#ifdef SGFHE_IA_NOCRT
            if (c + e >= 0) continue;
#endif
            const int tid = (int)threadIdx.x + (int)opaque_zero();   // one coefficient's loads at a time
            uint32_t y[NP];
#pragma unroll
            for (int q = 0; q < NP; q++)
                y[q] = (uint32_t)(c == 0 ? a0[q][e] : q < LP ? IA::priv(lds, q, tid)[e * T] : a1[q < LP ? 0 : q - LP][e]);
            const uint32_t i = (uint32_t)tid + (uint32_t)(T * e);
            const ulonglong2 d = load_digits(dig, (size_t)b * 2 + c, i, M);
            uint64_t lo, hi;
            crt_lean_one<NP, NL>(y, d.x, d.y, K, lo, hi);
            store_digits(dig, (size_t)b * 2 + c, i, M, lo, hi);
        }
}
#endif  // SGFHE_WITH_ITER_ALL

// ---- small-batch ("latency") form of the external product --------------------------------------
// A call with a handful of gates leaves most of the 256 CUs idle while one workgroup per
// (bootstrap, prime) walks through 4 forward and 2 inverse transforms.  Here the same work is cut
// into 4 + 2 independent workgroups per (bootstrap, prime), three launches per iteration:
//   k_fwd_phase   (bootstrap, prime, key row ph)  digit plane -> forward NTT -> products with the
//                 two key polynomials of that row, |.| < 0.74 * 2^29        -> zpart
//   k_inv_column  (bootstrap, prime, column c)    sum of the four partial products -> inverse NTT
//                 -> (x^j - 1) rotation -> residues                         -> yres
//   k_crt_acc     as before.
//   zpart [chunk][npr][4][2][m]   slot order (the order of the key slices)
// Both run with 8 points per thread where that fits a workgroup (LE = 3: twice the threads on
// each transform); the slot order of a transform does not depend on the points per thread.
template <int LOGM, int LE, bool WIDE = false>
__global__ void __launch_bounds__((NttGeom<LOGM, LE>::T), (NttGeom<LOGM, LE>::T >= 256 ? 4 : 1))
k_fwd_phase(const uint64_t *__restrict__ dig, const int32_t *__restrict__ keyk,
            int32_t *__restrict__ zpart, PrimeSet PS, uint32_t mode) {
    using G = NttGeom<LOGM, LE>;
    constexpr int M = G::M, T = G::T, E = G::E;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x;
    const uint32_t npr = PS[0].npr;
    const uint32_t ph = blockIdx.x & 3u;
    const uint32_t pi = (blockIdx.x >> 2) % npr;
    const uint32_t b = (blockIdx.x >> 2) / npr;
    const PrimeK P = PS[pi];
    const Mod md = mod_of(P);
    const int32_t sRd = (mode & MODE_RANDOM) ? P.sRr : P.sR;

    int32_t x[1][E];
    const uint32_t *dl = digit_lo_plane(dig, (size_t)b * 2 + (ph >> 1), M) + (ph & 1) * M;
    const uint16_t *dh = digit_hi_plane(dig, (size_t)b * 2 + (ph >> 1), M) + (ph & 1) * M;
    const uint8_t *dt = reinterpret_cast<const uint8_t *>(dig + ((size_t)b * 2 + (ph >> 1)) * 2 * M) + 12 * M + (ph & 1) * M;
#pragma unroll
    for (int e = 0; e < E; e++) {
        uint64_t d = dl[tid + T * e] | ((uint64_t)dh[tid + T * e] << 32);
        if constexpr (WIDE) d |= (uint64_t)dt[tid + T * e] << 48;
        x[0][e] = digit_reduce(d, md, sRd);
    }
    ntt_forward<LOGM, 1, LE>(x, lds, P.twf, tid, md);

    const int32_t *kp = keyk + ((size_t)pi * 8 + ph * 2) * M + E * tid;
    int32_t *zp = zpart + ((((size_t)b * npr + pi) * 4 + ph) * 2) * M + E * tid;
#pragma unroll
    for (int h = 0; h < E / 4; h++) {
        const int4 a = reinterpret_cast<const int4 *>(kp)[h];
        const int4 bq = reinterpret_cast<const int4 *>(kp + M)[h];
        const int32_t ka[4] = {a.x, a.y, a.z, a.w};
        const int32_t kb[4] = {bq.x, bq.y, bq.z, bq.w};
        int32_t r0[4], r1[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int32_t u = x[0][4 * h + t];
            r0[t] = smont(u, ka[t], md);  // |.| < 0.74 * 2^29
            r1[t] = smont(u, kb[t], md);
        }
        reinterpret_cast<int4 *>(zp)[h] = make_int4(r0[0], r0[1], r0[2], r0[3]);
        reinterpret_cast<int4 *>(zp + M)[h] = make_int4(r1[0], r1[1], r1[2], r1[3]);
    }
}

template <int LOGM, int LE>
__global__ void __launch_bounds__((NttGeom<LOGM, LE>::T), (NttGeom<LOGM, LE>::T >= 256 ? 4 : 1))
k_inv_column(const int32_t *__restrict__ zpart, uint32_t *__restrict__ yres,
             const uint32_t *__restrict__ ua, PrimeSet PS, uint32_t k, uint32_t n) {
    using G = NttGeom<LOGM, LE>;
    constexpr int M = G::M, T = G::T, E = G::E;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x;
    const uint32_t npr = PS[0].npr;
    const uint32_t c = blockIdx.x & 1u;
    const uint32_t pi = (blockIdx.x >> 1) % npr;
    const uint32_t b = (blockIdx.x >> 1) / npr;
    const PrimeK P = PS[pi];
    const Mod md = mod_of(P);

    int32_t z[1][E];
    const int32_t *zp = zpart + ((((size_t)b * npr + pi) * 4) * 2 + c) * M + E * tid;
#pragma unroll
    for (int h = 0; h < E / 4; h++) {
        int32_t acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int ph = 0; ph < 4; ph++) {
            const int4 v = reinterpret_cast<const int4 *>(zp + (size_t)ph * 2 * M)[h];
            acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;  // < 2.99 * 2^29
        }
#pragma unroll
        for (int t = 0; t < 4; t++) z[0][4 * h + t] = sred(acc[t], md);
    }
    ntt_inverse<LOGM, 1, LE>(z, lds, P.twi, tid, md);

    // y = x^j P - P  (as in k_extprod)
    uint32_t *yb = yres + (((size_t)b * 2 + c) * npr + pi) * M;
    const uint32_t j = ua[(size_t)b * n + k];
    const uint32_t yoff = 3u * (uint32_t)P.p + P.hoff;  // as in k_extprod
    lds_store<LOGM, 1, LE, G::STOP>(z, lds, tid);  // own addresses: the thread's last loads
    SGFHE_SYNC();
    constexpr uint32_t LOWMASK = (1u << G::STOP) - 1u;
    const uint32_t s0 = ((uint32_t)tid - j) & (2 * M - 1);
    const uint32_t lowswz = swz<LE>(s0 & LOWMASK);
    const uint32_t h0 = s0 >> G::STOP;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint32_t he = h0 + e;
        const uint32_t hipart = (he & (E - 1)) << G::STOP;
        const uint32_t addr = hipart ^ lowswz ^ (G::STOP >= 9 ? 0u : swz_bits<LE>(hipart));
        const int32_t v = (int32_t)lds[addr];
        const int32_t vs = (he & E) ? -v : v;  // x^m = -1
        yb[tid + T * e] = (uint32_t)(vs - z[0][e]) + yoff;
    }
}

// ---- quarter form of the latency kernels (round 4) ------------------------------------------------------
// A call of a few gates is a dependent chain of 3 n launches, each a handful of workgroups on an
// otherwise idle device: what counts is the length of one workgroup's work.  Here each transform of
// m points is cut across four workgroups of m / 32 threads:
//   k_fwd_quarter  (bootstrap, prime, key row, quarter q): reads all four quarters of the digit plane,
//                  forms quarter q of the first two Cooley-Tukey stages (one radix-4 combination per
//                  point: the arithmetic of ntt.h fwd_step4), runs the remaining log2(m) - 2 stages as
//                  an (m / 4)-point transform on the twiddle sub-tree of that quarter (tables twq), and
//                  multiplies with its quarter of the two key polynomials             -> zpart (unchanged layout)
//   k_inv_quarter  (bootstrap, prime, column, quarter q): sums the four partial products of its slots,
//                  applies (x^j - 1) in the NTT domain -- slot s holds the value at psi^(2 brv(s) + 1), so
//                  the factor is psi^(j (2 brv(s) + 1)) - 1, from a table of psi powers -- and runs the first
//                  log2(m) - 2 Gentleman-Sande stages on its quarter                   -> ypart[...][q][m / 4]
//   k_crt_lean1q   one thread per coefficient: the last two inverse stages (a radix-4 combination of the
//                  four partial values per prime, only the output this coefficient needs), then the CRT /
//                  accumulate / flatten of k_crt_lean1.
// No rotation epilogue, no LDS exchange for it, a quarter of the butterflies per workgroup.  Deterministic
// flatten, m >= 4096, up to `split_max` gates (engine.hip); same residues, same digits, same bytes.
template <int LOGM, int LE>
__global__ void __launch_bounds__((NttGeom<LOGM - 2, LE>::T))
k_fwd_quarter(const uint64_t *__restrict__ dig, const int32_t *__restrict__ keyk,
              int32_t *__restrict__ zpart, PrimeSet PS, uint32_t mode) {
    constexpr int LS = LOGM - 2;
    using G = NttGeom<LS, LE>;
    constexpr int M = 1 << LOGM, MS = G::M, T = G::T, E = G::E;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x;
    const uint32_t npr = PS[0].npr;
    const uint32_t q = blockIdx.x & 3u, ph = (blockIdx.x >> 2) & 3u;
    const uint32_t pi = (blockIdx.x >> 4) % npr, b = (blockIdx.x >> 4) / npr;
    const PrimeK P = PS[pi];
    const Mod md = mod_of(P);
    const int32_t sRd = (mode & MODE_RANDOM) ? P.sRr : P.sR;   // digit offset of the flatten mode
    const uint32_t *dl = digit_lo_plane(dig, (size_t)b * 2 + (ph >> 1), M) + (ph & 1) * M;
    const uint16_t *dh = digit_hi_plane(dig, (size_t)b * 2 + (ph >> 1), M) + (ph & 1) * M;
    // quarter q of the first two stages: with X0..X3 the coefficients i, i + m/4, i + m/2, i + 3m/4,
    //   u = f1 X2;  q < 2: (X0 + u) +- (f2 X1 + fp2 X3);  q >= 2: (X0 - u) +- (f3 X1 + fp3 X3)
    const int32_t wB = (q & 2) ? P.f3 : P.f2, wP = (q & 2) ? P.fp3 : P.fp2;
    int32_t x[1][E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint32_t i = (uint32_t)tid + (uint32_t)(T * e);
        int32_t X[4];
#pragma unroll
        for (int t = 0; t < 4; t++)
            X[t] = digit_reduce(dl[i + t * MS] | ((uint64_t)dh[i + t * MS] << 32), md, sRd);   // |.| <= p + 2^16
        const int32_t u = smont(X[2], P.f1, md);                                              // < 0.57 * 2^29
        const int32_t a = (q & 2) ? X[0] - u : X[0] + u;
        const int32_t w = sredc((int64_t)X[1] * wB + (int64_t)X[3] * wP, md);                   // < 0.63 * 2^29
        x[0][e] = sred_floor((q & 1) ? a - w : a + w, md);                                    // [0, p]
    }
    ntt_forward<LS, 1, LE>(x, lds, P.twq + (size_t)q * M, tid, md);
    const int32_t *kp = keyk + ((size_t)pi * 8 + ph * 2) * M + q * MS + E * tid;
    int32_t *zp = zpart + ((((size_t)b * npr + pi) * 4 + ph) * 2) * M + q * MS + E * tid;
#pragma unroll
    for (int h = 0; h < E / 4; h++) {
        const int4 a = reinterpret_cast<const int4 *>(kp)[h];
        const int4 bq = reinterpret_cast<const int4 *>(kp + M)[h];
        const int32_t ka[4] = {a.x, a.y, a.z, a.w};
        const int32_t kb[4] = {bq.x, bq.y, bq.z, bq.w};
        int32_t r0[4], r1[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int32_t u = x[0][4 * h + t];
            r0[t] = smont(u, ka[t], md);  // |.| < 0.74 * 2^29
            r1[t] = smont(u, kb[t], md);
        }
        reinterpret_cast<int4 *>(zp)[h] = make_int4(r0[0], r0[1], r0[2], r0[3]);
        reinterpret_cast<int4 *>(zp + M)[h] = make_int4(r1[0], r1[1], r1[2], r1[3]);
    }
}

template <int LOGM, int LE>
__global__ void __launch_bounds__((NttGeom<LOGM - 2, LE>::T))
k_inv_quarter(const int32_t *__restrict__ zpart, int32_t *__restrict__ ypart,
              const uint32_t *__restrict__ ua, PrimeSet PS, uint32_t k, uint32_t n) {
    constexpr int LS = LOGM - 2;
    using G = NttGeom<LS, LE>;
    constexpr int M = 1 << LOGM, MS = G::M, T = G::T, E = G::E;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x;
    const uint32_t npr = PS[0].npr;
    const uint32_t q = blockIdx.x & 3u, c = (blockIdx.x >> 2) & 1u;
    const uint32_t pi = (blockIdx.x >> 3) % npr, b = (blockIdx.x >> 3) / npr;
    const PrimeK P = PS[pi];
    const Mod md = mod_of(P);
    const uint32_t j = ua[(size_t)b * n + k];
    // the factor of slot s = q m/4 + E tid + e: psi^(j (2 brv(s) + 1) mod 2m) - 1, requested first (a gather)
    int32_t dfac[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint32_t sl = q * (uint32_t)MS + (uint32_t)(E * tid + e);
        const uint32_t br = __brev(sl) >> (32 - LOGM);
        dfac[e] = P.pw[(j * (2u * br + 1u)) & (2u * M - 1u)];
    }
    int32_t z[1][E];
    const int32_t *zp = zpart + ((((size_t)b * npr + pi) * 4) * 2 + c) * M + q * MS + E * tid;
#pragma unroll
    for (int h = 0; h < E / 4; h++) {
        int32_t acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int ph = 0; ph < 4; ph++) {
            const int4 v = reinterpret_cast<const int4 *>(zp + (size_t)ph * 2 * M)[h];
            acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;  // < 2.99 * 2^29
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            // (psi^e - 1) R mod p from two centred residues: in (-p, p), centred again for the product's bound
            const int32_t d = scentre(dfac[4 * h + t] - P.r1, md);
            z[0][4 * h + t] = smont(acc[t], d, md);                      // < 0.19 + 0.5 = 0.69 * 2^29
        }
    }
    ntt_inverse<LS, 1, LE>(z, lds, P.twq + (size_t)q * M + MS, tid, md);   // natural order inside the quarter, |.| < 1.4 * 2^29
    int32_t *yb = ypart + ((((size_t)b * 2 + c) * npr + pi) * 4 + q) * MS;
#pragma unroll
    for (int e = 0; e < E; e++) yb[tid + T * e] = z[0][e];
}

// k_fwd_quarter and k_inv_quarter in one launch (round 4, second step): a quarter of the slots is closed
// under everything between the radix-4 head of the forward transforms and the radix-4 tail of the inverse
// ones, so one workgroup per (bootstrap, prime, quarter) can go from digit planes to partial products
// without a hand-over through memory -- one dependent launch less per iteration of the chain.  Two groups
// of m / 32 threads: group g transforms key rows 2 g and 2 g + 1 side by side (the two digits of one
// accumulator polynomial), forms its share of both product columns with 64-bit multiply-adds,
//   w_c = REDC(u_2g K[2g][c] + u_2g+1 K[2g+1][c]),   |.| < (2 * 3.95 * 2^29 * 2^28) / 2^32 + p / 2 < 0.99 * 2^29,
// keeps w_g, hands w_(1-g) to the other group through LDS, and runs the inverse quarter of column g on
// (w_g + w'_g) (psi^(j (2 brv(s) + 1)) - 1): |sum| < 1.98 * 2^29, after the Montgomery product < 0.63 * 2^29.
// Same residues modulo every prime as the two-launch form, hence the same digits.
//   lds: [group][2][m/4] transform exchange, then [2][m/4] hand-over  (6 m/4 words)
template <int LOGM, int LE>
__global__ void __launch_bounds__((2 * NttGeom<LOGM - 2, LE>::T))
k_ext_quarter(const uint64_t *__restrict__ dig, const int32_t *__restrict__ keyk, int32_t *__restrict__ ypart,
              const uint32_t *__restrict__ ua, PrimeSet PS, uint32_t mode, uint32_t k, uint32_t n) {
    constexpr int LS = LOGM - 2;
    using G = NttGeom<LS, LE>;
    constexpr int M = 1 << LOGM, MS = G::M, T = G::T, E = G::E;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t g = (uint32_t)threadIdx.x / (uint32_t)T;          // key rows 2 g, 2 g + 1; product column g
    const int tl = (int)((uint32_t)threadIdx.x & (uint32_t)(T - 1));
    const uint32_t npr = PS[0].npr;
    const uint32_t q = blockIdx.x & 3u;
    const uint32_t pi = (blockIdx.x >> 2) % npr, b = (blockIdx.x >> 2) / npr;
    const PrimeK P = PS[pi];
    const Mod md = mod_of(P);
    const int32_t sRd = (mode & MODE_RANDOM) ? P.sRr : P.sR;
    uint32_t *const glds = lds + (size_t)g * 2 * MS;
    int32_t *const hand = reinterpret_cast<int32_t *>(lds) + 4 * MS;

    // requested first: the rotation factors of column g's slots (a gather behind the load of j) and the key
    const uint32_t j = ua[(size_t)b * n + k];
    int32_t dfac[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint32_t sl = q * (uint32_t)MS + (uint32_t)(E * tl + e);
        const uint32_t br = __brev(sl) >> (32 - LOGM);
        dfac[e] = P.pw[(j * (2u * br + 1u)) & (2u * M - 1u)];
    }
    int4 kk[2][2][E / 4];                                           // [row of the pair][column][..]
    {
        const int32_t *kp = keyk + ((size_t)pi * 8 + g * 4) * M + q * MS + E * tl;
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int h = 0; h < E / 4; h++) kk[r][c][h] = reinterpret_cast<const int4 *>(kp + (size_t)(r * 2 + c) * M)[h];
    }
    // head: quarter q of the first two stages of both rows (k_fwd_quarter)
    const int32_t wB = (q & 2) ? P.f3 : P.f2, wP = (q & 2) ? P.fp3 : P.fp2;
    int32_t x[2][E];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const uint32_t *dl = digit_lo_plane(dig, (size_t)b * 2 + g, M) + r * M;
        const uint16_t *dh = digit_hi_plane(dig, (size_t)b * 2 + g, M) + r * M;
#pragma unroll
        for (int e = 0; e < E; e++) {
            const uint32_t i = (uint32_t)tl + (uint32_t)(T * e);
            int32_t X[4];
#pragma unroll
            for (int t = 0; t < 4; t++)
                X[t] = digit_reduce(dl[i + t * MS] | ((uint64_t)dh[i + t * MS] << 32), md, sRd);
            const int32_t u = smont(X[2], P.f1, md);
            const int32_t a = (q & 2) ? X[0] - u : X[0] + u;
            const int32_t w = sredc((int64_t)X[1] * wB + (int64_t)X[3] * wP, md);
            x[r][e] = sred_floor((q & 1) ? a - w : a + w, md);
        }
    }
    ntt_forward<LS, 2, LE>(x, glds, P.twq + (size_t)q * M, tl, md);
    // products of the pair, both columns; the other group's column goes to the hand-over area
    int32_t own[E];
#pragma unroll
    for (int h = 0; h < E / 4; h++) {
        int32_t w[2][4];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int4 a4 = kk[0][c][h], b4 = kk[1][c][h];
            const int32_t ka[4] = {a4.x, a4.y, a4.z, a4.w}, kb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int t = 0; t < 4; t++)
                w[c][t] = sredc((int64_t)x[0][4 * h + t] * ka[t] + (int64_t)x[1][4 * h + t] * kb[t], md);
        }
        const int4 w0 = make_int4(w[0][0], w[0][1], w[0][2], w[0][3]), w1 = make_int4(w[1][0], w[1][1], w[1][2], w[1][3]);
        const int4 keep = g ? w1 : w0, give = g ? w0 : w1;
        own[4 * h] = keep.x; own[4 * h + 1] = keep.y; own[4 * h + 2] = keep.z; own[4 * h + 3] = keep.w;
        reinterpret_cast<int4 *>(hand + (size_t)(1u - g) * MS + E * tl)[h] = give;
    }
    SGFHE_SYNC();   // hand-over written; every wave is through its forward exchanges
    int32_t z[1][E];
#pragma unroll
    for (int h = 0; h < E / 4; h++) {
        const int4 o = reinterpret_cast<const int4 *>(hand + (size_t)g * MS + E * tl)[h];
        const int32_t ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int32_t d = scentre(dfac[4 * h + t] - P.r1, md);
            z[0][4 * h + t] = smont(own[4 * h + t] + ov[t], d, md);
        }
    }
    ntt_inverse<LS, 1, LE>(z, glds, P.twq + (size_t)q * M + MS, tl, md);
    int32_t *yb = ypart + ((((size_t)b * 2 + g) * npr + pi) * 4 + q) * MS;
#pragma unroll
    for (int e = 0; e < E; e++) yb[tl + T * e] = z[0][e];
}

// One thread per coefficient i of (bootstrap, c): the last two Gentleman-Sande stages on the four partial
// values Y_0..Y_3 at i mod m/4 of every prime --
//   C0 = Y0 + Y1, C1 = v2 (Y0 - Y1), C2 = Y2 + Y3, C3 = v3 (Y2 - Y3);
//   coefficient in quarter 0: C0 + C2;  1: C1 + C3;  2: v1 (C0 - C2);  3: v1 (C1 - C3)
// (only the one this coefficient needs; the quarter is the same for a whole workgroup) -- then the residue in
// the form k_extprod hands over (+ 3 p, + (p - 1) / 2 on the last prime) and crt_lean_one.
template <int NP, int NL>
__global__ void __launch_bounds__(256)
k_crt_lean1q(const int32_t *__restrict__ ypart, uint64_t *__restrict__ dig, PrimeSet PS,
             const CrtLean *__restrict__ K, uint32_t total, uint32_t logm) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const uint32_t M = 1u << logm, MS = M >> 2;
    const uint32_t i = t & (M - 1);
    const uint32_t bc = t >> logm;
    const uint32_t jq = i & (MS - 1), qq = i >> (logm - 2);
    uint32_t y[NP];
#pragma unroll
    for (int q = 0; q < NP; q++) {
        const PrimeK P = PS[q];
        const Mod md = mod_of(P);
        const int32_t *yp = ypart + (((size_t)bc * NP + q) * 4) * MS + jq;
        int32_t r;
        if (qq & 1u) {          // C1, C3
            const int32_t c1 = smont(yp[0] - yp[MS], P.v2, md), c3 = smont(yp[2 * MS] - yp[3 * MS], P.v3, md);
            r = (qq & 2u) ? smont(c1 - c3, P.v1, md) : c1 + c3;             // < 0.75 / < 1.5 * 2^29
        } else {                // C0, C2
            const int32_t c0 = sred(yp[0] + yp[MS], md), c2 = sred(yp[2 * MS] + yp[3 * MS], md);   // sums < 2.8: reduced to 0.51
            r = (qq & 2u) ? smont(c0 - c2, P.v1, md) : c0 + c2;             // < 0.75 / < 1.03 * 2^29
        }
        y[q] = (uint32_t)(r + 3 * P.p) + P.hoff;                            // non-negative, below 4.6 p
    }
    const ulonglong2 d = load_digits(dig, bc, i, M);
    uint64_t lo, hi;
    crt_lean_one<NP, NL>(y, d.x, d.y, K, lo, hi);
    store_digits(dig, bc, i, M, lo, hi);
}

// ---- k_init -------------------------------------------------------------------------------------
// u = lwe1 + lwe2 (fhe.jl:566); a = 0 (fhe.jl:570); b = x^(-u.b) t DQ_tilde (fhe.jl:572-573) with
// t = initial_poly (fhe.jl:535-548): +1 on [0, Dr), 0 at Dr, -1 on (Dr, m).  Every coefficient of
// b is 0, +DQ_tilde or -DQ_tilde, so the three possible digit pairs are precomputed.
// One thread per (bootstrap of the padded chunk, coefficient).
template <bool ROWS = false>
__global__ void __launch_bounds__(256)
k_init(const uint64_t *__restrict__ a1, const uint64_t *__restrict__ b1,
       const uint64_t *__restrict__ a2, const uint64_t *__restrict__ b2,
       uint64_t *__restrict__ dig, uint32_t *__restrict__ ua, const CrtConst *__restrict__ CC,
       uint32_t nvalid, uint32_t chunk, uint32_t n, uint32_t logm, uint32_t mode, RndArgs ra) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    const uint32_t M = 1u << logm;
    if (t >= chunk * M) return;
    const uint32_t i = t & (M - 1);
    const uint32_t b = t >> logm;
    const uint32_t rmask = 2 * M - 1;  // r = 2 m
    const bool valid = b < nvalid;
    if (i < n) ua[(size_t)b * n + i] =
        valid ? (uint32_t)((a1[(size_t)b * n + i] + a2[(size_t)b * n + i]) & rmask) : 0u;
    int tv = 0;
    if (valid) {
        const uint32_t ub = (uint32_t)((b1[b] + b2[b]) & rmask);
        const uint32_t Dr = M / 2;  // r / 4
        const uint32_t src = (i + ub) & rmask;
        const uint32_t s = src & (M - 1);
        tv = s < Dr ? 1 : (s == Dr ? 0 : -1);
        if (src & M) tv = -tv;
    }
    if (mode & MODE_RANDOM) {  // flatten(rng, .) of a = 0 and of b in {0, +-DQ_tilde}
        const u128 Q = CC->Q;
        const u128 offr = CC->offneg_rnd ? Q - CC->offneg_rnd : 0;
        u128 xb = offr + (tv > 0 ? CC->DQ : (tv < 0 ? Q - CC->DQ : 0));
        if (xb >= Q) xb -= Q;
        ChaChaKey key;
        uint32_t cz, cw;
        rnd_stream<ROWS>(ra, b, key, cz, cw);
        const ulonglong2 ea = random_digits(offr, CC, key, make_uint4(i, 0u, cz, cw));
        const ulonglong2 eb = random_digits(xb, CC, key, make_uint4(M + i, 0u, cz, cw));
        store_digits(dig, (size_t)b * 2 + 0, i, M, ea.x, ea.y, (mode & MODE_WIDE) != 0);
        store_digits(dig, (size_t)b * 2 + 1, i, M, eb.x, eb.y, (mode & MODE_WIDE) != 0);
        return;
    }
    store_digits(dig, (size_t)b * 2 + 0, i, M, CC->dig0.x, CC->dig0.y);
    ulonglong2 d = CC->dig0;
    if (valid) d = tv > 0 ? CC->digP : (tv < 0 ? CC->digN : CC->dig0);
    store_digits(dig, (size_t)b * 2 + 1, i, M, d.x, d.y);
}

// ---- k_final ------------------------------------------------------------------------------------
// LWE extraction (fhe.jl:585-592, extract :237-244 in its i >= n branch) and ModRed
// (fhe.jl:616-618,644-648; rescale utils.jl:78-92).  One thread per (bootstrap, t in [0, n]).
__device__ __forceinline__ u128 acc_from_digits(ulonglong2 d, const CrtConst *CC,
                                                uint32_t mode = 0) {
    if (mode & MODE_RANDOM)  // digits up to 4 B: reduce properly
        return mod_wide((u128)d.y * (uint64_t)CC->B + d.x + CC->offneg_rnd, CC->Q, CC->invQ, nullptr);
    u128 x = (u128)d.y * (uint64_t)CC->B + d.x + CC->offneg;
    if (x >= CC->Q) x -= CC->Q;
    return x;
}
__device__ __forceinline__ uint64_t modred(u128 x, const CrtConst *CC) {
    const u128 num = x << CC->logr;  // x * r
    uint64_t q;
    const u128 rem = mod_wide(num, CC->Q, CC->invQ, &q);
    if (rem >= CC->roundthr) {
        q += 1;
        if (q == (1ull << CC->logr)) q = 0;
    }
    return q;
}
__global__ void __launch_bounds__(256)
k_final(const uint64_t *__restrict__ dig, uint64_t *__restrict__ out,
        const CrtConst *__restrict__ CC, uint32_t nvalid, uint32_t n, uint32_t logm, uint32_t raw,
        uint32_t mode) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nvalid * (n + 1)) return;
    const uint32_t M = 1u << logm;
    const uint32_t b = t / (n + 1), e = t % (n + 1);
    const size_t ba = (size_t)b * 2 + 0, bb = (size_t)b * 2 + 1;
    const u128 Q = CC->Q;
    const bool wide = (mode & MODE_WIDE) != 0;
    u128 va, vo;
    if (e < n) {
        va = acc_from_digits(load_digits(dig, ba, 3 * M / 4 - e, M, wide), CC, mode);
        const u128 w = acc_from_digits(load_digits(dig, ba, M / 4 - e, M, wide), CC, mode);
        vo = w ? Q - w : 0;
    } else {
        va = CC->DQ + acc_from_digits(load_digits(dig, bb, 3 * M / 4, M, wide), CC, mode);
        if (va >= Q) va -= Q;
        const u128 w = acc_from_digits(load_digits(dig, bb, M / 4, M, wide), CC, mode);
        vo = CC->DQ >= w ? CC->DQ - w : CC->DQ + Q - w;
    }
    const u128 vx = vo >= va ? vo - va : vo + Q - va;
    const size_t stride = n + 1;
    if (raw) {
        ulonglong2 *o = reinterpret_cast<ulonglong2 *>(out) + (size_t)b * 3 * stride + e;
        o[0] = make_ulonglong2((uint64_t)va, (uint64_t)(va >> 64));
        o[stride] = make_ulonglong2((uint64_t)vo, (uint64_t)(vo >> 64));
        o[2 * stride] = make_ulonglong2((uint64_t)vx, (uint64_t)(vx >> 64));
    } else {
        uint64_t *o = out + (size_t)b * 3 * stride + e;
        o[0] = modred(va, CC);
        o[stride] = modred(vo, CC);
        o[2 * stride] = modred(vx, CC);
    }
}

// digits -> canonical accumulators (debug hook)
__global__ void __launch_bounds__(256)
k_dump_acc(const uint64_t *__restrict__ dig, ulonglong2 *__restrict__ out,
           const CrtConst *__restrict__ CC, uint32_t total, uint32_t logm, uint32_t mode) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const uint32_t M = 1u << logm;
    const u128 x = acc_from_digits(load_digits(dig, t >> logm, t & (M - 1), M, (mode & MODE_WIDE) != 0), CC, mode);
    out[t] = make_ulonglong2((uint64_t)x, (uint64_t)(x >> 64));
}

// stored digit planes -> [bootstrap][c][digit][m] uint64 (debug hook): e_i = u_i + s, or
// u_i + s + xmax in the randomised mode, u the reference's flatten result as a signed integer
__global__ void __launch_bounds__(256)
k_dump_digits(const uint64_t *__restrict__ dig, uint64_t *__restrict__ out, uint32_t total,
              uint32_t logm, uint32_t mode) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const uint32_t M = 1u << logm, bc = t >> logm, i = t & (M - 1);
    const ulonglong2 d = load_digits(dig, bc, i, M, (mode & MODE_WIDE) != 0);
    out[((size_t)bc * 2 + 0) * M + i] = d.x;
    out[((size_t)bc * 2 + 1) * M + i] = d.y;
}

// canonical residues -> digits of x' = (v + off) mod Q  (flatten, utils.jl:155-189)
__global__ void __launch_bounds__(256)
k_flatten_canon(const ulonglong2 *__restrict__ in, uint64_t *__restrict__ dig,
                const CrtConst *__restrict__ CC, uint32_t total, uint32_t logm) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const uint32_t M = 1u << logm;
    const ulonglong2 v = in[t];
    u128 x = (((u128)v.y << 64) | v.x) + (CC->Q - CC->offneg);
    if (x >= CC->Q) x -= CC->Q;
    if (x >= CC->Q) x -= CC->Q;
    uint64_t hi;
    const u128 lo = mod_wide(x, CC->B, CC->invB, &hi);
    store_digits(dig, t >> logm, t & (M - 1), M, (uint64_t)lo, hi);
}

// ==================================================================================================
// Packing LWEs into an RLWE ciphertext: pack_encrypted_bits / shortened_external_product
// (src/fhe.jl:632-641,660-696), second caller of the hot path.
//   1. n bootstraps (trivial encryption of 1, bit_i), AND branch, un-reduced      (k-loop above)
//   2. k_pack_flatten: as_i = polynomial of the i-th LWE coefficients (fhe.jl:675-677), flattened
//   3. k_shortprod:   for a group of slices i: sum_i flatten(as_i) * C_i[2:4, :]  in the NTT domain
//                     (exact integers stay below 0.4 M for `G` slices per group), inverse NTT
//   4. k_pack_finish: CRT of every group, sum mod Q, w = ModRed(-W), v = ModRed(b - V)
// ==================================================================================================

// raw: [count * n][3][n + 1] 16-byte residues (RAW_MODQ bootstrap output); only gate 0 (AND) is
// read.  pdig: [count][n slices][2 digits][len] uint64 with len = n coefficients for the
// deterministic flatten (the zero padding of as_i, fhe.jl:675-677, has the constant digits of 0)
// and len = m for the randomised one (flatten_poly draws for every coefficient of the resized
// polynomial, utils.jl:253-264).
__global__ void __launch_bounds__(256)
k_pack_flatten(const ulonglong2 *__restrict__ raw, uint64_t *__restrict__ pdig,
               const CrtConst *__restrict__ CC, uint32_t count, uint32_t n, uint32_t logm,
               uint32_t mode, RndArgs ra) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t len = (mode & MODE_RANDOM) ? (1u << logm) : n;
    if (t >= (size_t)count * n * len) return;
    const uint32_t j = (uint32_t)(t % len);         // bit index = coefficient index of as_i
    const uint32_t i = (uint32_t)((t / len) % n);   // LWE coefficient index = slice
    const uint32_t ci = (uint32_t)(t / ((size_t)n * len));
    const u128 Q = CC->Q;
    u128 x = 0;
    if (j < n) {
        const ulonglong2 v = raw[(((size_t)ci * n + j) * 3 + 0) * (n + 1) + i];
        x = ((u128)v.y << 64) | v.x;
    }
    uint64_t *d = pdig + (((size_t)ci * n + i) * 2) * len + j;
    if (mode & MODE_RANDOM) {
        x += CC->offneg_rnd ? Q - CC->offneg_rnd : 0;   // + (s + xmax)(1 + B)
        if (x >= Q) x -= Q;
        const ulonglong2 e = random_digits(x, CC, ra.key, make_uint4(j, 0x80000000u | i, ci, ra.call));
        d[0] = e.x;
        d[len] = e.y;
        return;
    }
    x += Q - CC->offneg;   // + off
    if (x >= Q) x -= Q;
    uint64_t hi;
    const u128 lo = mod_wide(x, CC->B, CC->invB, &hi);
    d[0] = (uint64_t)lo;
    d[len] = hi;
}

// grid = count * groups * NPR workgroups of T threads.  Workgroup (ci, g, pi) runs the 2 G phases
// (slice i in the group, digit) through the forward NTT and accumulates both product columns
// lazily (mod 2p; z0 in registers, z1 in LDS), then two inverse NTTs.
//   yg [count][groups][2][NPR][m] residues (+ hoff), in [0, p]
template <int LOGM>
__global__ void __launch_bounds__((NttGeom<LOGM, LOGE>::T), (NttGeom<LOGM, LOGE>::T >= 256 ? 4 : 1))
k_shortprod(const uint64_t *__restrict__ pdig, const int32_t *__restrict__ keyhat,
            uint32_t *__restrict__ yg, PrimeSet PS, const CrtConst *__restrict__ CC, uint32_t n,
            uint32_t G, uint32_t groups, uint32_t mode) {
    using GE = NttGeom<LOGM, LOGE>;
    constexpr int M = GE::M, T = GE::T, E = GE::E;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    int32_t *const ldsi = reinterpret_cast<int32_t *>(lds);
    const uint32_t npr = PS[0].npr;
    const uint32_t pi = blockIdx.x % npr;
    const uint32_t g = (blockIdx.x / npr) % groups;
    const uint32_t ci = blockIdx.x / (npr * groups);
    const PrimeK P = PS[pi];
    const Mod md = mod_of(P);
    const int32_t sRd = (mode & MODE_RANDOM) ? P.sRr : P.sR;
    // residues of the digits of a zero coefficient (x' = off): the zero padding of as_i
    // (deterministic flatten of 0; the randomised mode draws for the padding too: k_pack_flatten
    // writes all m coefficients there)
    const int32_t zlo = digit_reduce(CC->dig0.x, md, P.sR), zhi = digit_reduce(CC->dig0.y, md, P.sR);
    const uint32_t len = (mode & MODE_RANDOM) ? (uint32_t)M : n;  // stored coefficients per digit polynomial

    int32_t z0[1][E];
#pragma unroll
    for (int e = 0; e < E; e++) { z0[0][e] = 0; ldsi[M + e * T + threadIdx.x] = 0; }
#pragma unroll 1
    for (uint32_t ph = 0; ph < 2 * G; ph++) {
        const int tid = (int)threadIdx.x + (int)opaque_zero();
        const uint32_t i = g * G + (ph >> 1), digit = ph & 1;
        int32_t x[1][E];
        const uint64_t *d = pdig + (((size_t)ci * n + i) * 2 + digit) * len;
#pragma unroll
        for (int e = 0; e < E; e++) {
            const uint32_t idx = tid + T * e;
            x[0][e] = idx < len ? digit_reduce(d[idx], md, sRd) : (digit ? zhi : zlo);
        }
        SGFHE_SYNC();
        ntt_forward<LOGM, 1, LOGE>(x, lds, P.twf, tid, md);
        // key rows l + 1 .. 2 l of slice i (fhe.jl:638-639): polynomials (2 + digit) * 2 + c
        const int32_t *kp = keyhat + (((size_t)i * npr + pi) * 8 + (2 + digit) * 2) * M + E * tid;
#pragma unroll
        for (int h = 0; h < E / 4; h++) {
            const int4 a = reinterpret_cast<const int4 *>(kp)[h];
            const int4 bq = reinterpret_cast<const int4 *>(kp + M)[h];
            const int32_t ka[4] = {a.x, a.y, a.z, a.w};
            const int32_t kb[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int e = 4 * h + t;
                const int32_t u = x[0][e];
                z0[0][e] = sred(z0[0][e] + smont(u, ka[t], md), md);  // stays below 0.52 * 2^29
                int32_t *zp = ldsi + M + e * T + tid;
                *zp = sred(*zp + smont(u, kb[t], md), md);
            }
        }
    }
    const int tid = threadIdx.x;
    uint32_t *yb = yg + (((size_t)ci * groups + g) * 2 * npr + pi) * M;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        if (c > 0) {
#pragma unroll
            for (int e = 0; e < E; e++) z0[0][e] = ldsi[M + e * T + tid];
            SGFHE_SYNC();
        }
        ntt_inverse<LOGM, 1, LOGE>(z0, lds, P.twi, tid, md);
#pragma unroll
        for (int e = 0; e < E; e++)
            yb[(size_t)c * npr * M + tid + T * e] = condsub(sfull(z0[0][e], md) + P.hoff, (uint32_t)P.p);
    }
}

// One thread per (ciphertext, coefficient k < m): W = sum_g CRT(y_g column 0), V likewise
// (fhe.jl:686-687); w = ModRed(-W), v = ModRed(b_k - V) (fhe.jl:689-693), b_k = the un-reduced
// LWE constant of bit k for k < n and 0 beyond (resize, fhe.jl:678).
template <int NP>
__global__ void __launch_bounds__(256)
k_pack_finish(const uint32_t *__restrict__ yg, const ulonglong2 *__restrict__ raw,
              uint64_t *__restrict__ out_w, uint64_t *__restrict__ out_v,
              const CrtConst *__restrict__ CC, uint32_t count, uint32_t n, uint32_t logm,
              uint32_t groups) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t M = 1u << logm;
    if (t >= (size_t)count * M) return;
    const uint32_t kk = (uint32_t)(t & (M - 1));
    const uint32_t ci = (uint32_t)(t >> logm);
    const u128 Q = CC->Q;
    u128 acc[2] = {0, 0};
    for (uint32_t g = 0; g < groups; g++) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const uint32_t *yp = yg + ((((size_t)ci * groups + g) * 2 + c) * NP) * M + kk;
            uint32_t y[NP];
#pragma unroll
            for (int q = 0; q < NP; q++) y[q] = yp[(size_t)q * M];
            const U96 a = crt_reduce(y, CC, false, make_ulonglong2(0, 0));
            acc[c] += ((u128)a.w2 << 64) | ((uint64_t)a.w1 << 32) | a.w0;
            if (acc[c] >= Q) acc[c] -= Q;
        }
    }
    const u128 w1 = acc[0] ? Q - acc[0] : 0;
    u128 bk = 0;
    if (kk < n) {
        const ulonglong2 v = raw[(((size_t)ci * n + kk) * 3 + 0) * (n + 1) + n];
        bk = ((u128)v.y << 64) | v.x;
    }
    const u128 v1 = bk >= acc[1] ? bk - acc[1] : bk + Q - acc[1];
    out_w[t] = modred(w1, CC);
    out_v[t] = modred(v1, CC);
}

// ==================================================================================================
// Bootstrap-key generation on the device: BootstrapKey(rng, sk) (src/fhe.jl:181-201), SURVEY.md 8f
// row N2.  Randomness: ChaCha20 (RFC 8439 block function) keyed with the caller's 32-byte seed,
// one stream per (domain, key row), nonce = (domain, k * 4 + row, 0), block counter from 0:
//   domain 1  the uniform polynomial a_row: coefficient i = words 4 (i mod 4) .. + 3 of block i / 4,
//             lo = w0 | w1 << 32, hi = w2 | w3 << 32, value (hi 2^64 + lo) mod Q   (fhe.jl:193)
//   domain 2  the noise e_row: coefficient i = words 2 (i mod 8), + 1 of block i / 8,
//             d = w0 | w1 << 32, value d mod (2 noise + 1) - noise                   (fhe.jl:194)
// The streams are counter-addressed, so every thread computes its own block; the oracle's
// generator (oracle/sgfhe_oracle.c, oracle/bigint_oracle.py) reads the same streams and produces
// the same key from the same seed.
// Per batch of rows:  k_keygen_draw -> k_polymul_s (a (*) s, exact, per prime) -> k_crt_acc (CANON)
// -> k_keygen_finish (b = a (*) s + e, + s_k G on the constant terms) -> k_key_transform.
// ==================================================================================================

__device__ __forceinline__ void chacha20_block(const ChaChaKey &key, uint32_t counter, uint32_t n0,
                                               uint32_t n1, uint32_t n2, uint32_t (&out)[16]) {
    chacha_block<20>(key, counter, n0, n1, n2, out);
}
// (hi 2^64 + lo) mod Q for 2^16 <= Q < 2^94: long division in base 2^32 (quotients < 2^32 are
// exact through the double-precision estimate of mod_wide)
__device__ __forceinline__ u128 mod128(uint64_t hi, uint64_t lo, const CrtConst *CC) {
    const u128 Q = CC->Q;
    u128 r = mod_wide((u128)(hi >> 32), Q, CC->invQ, nullptr);
    r = mod_wide((r << 32) | (uint32_t)hi, Q, CC->invQ, nullptr);
    r = mod_wide((r << 32) | (uint32_t)(lo >> 32), Q, CC->invQ, nullptr);
    return mod_wide((r << 32) | (uint32_t)lo, Q, CC->invQ, nullptr);
}
// rows [row0, row0 + R) (row = k * 4 + gadget row): canonical a into acan[r][m], noise into e[r][m].
// One thread per 64-byte block of the a stream (four coefficients); every second thread also takes
// one block of the e stream (eight coefficients).
__global__ void __launch_bounds__(256)
k_keygen_draw(ulonglong2 *__restrict__ acan, int32_t *__restrict__ e,
              const CrtConst *__restrict__ CC, ChaChaKey key, uint32_t noise, uint32_t row0,
              uint32_t R, uint32_t logm) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    const uint32_t M = 1u << logm, bpr = M >> 2;  // a-stream blocks per row
    if (t >= R * bpr) return;
    const uint32_t blk = t & (bpr - 1), r = t / bpr;
    uint32_t w[16];
    chacha20_block(key, blk, 1u, row0 + r, 0u, w);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint64_t lo = ((uint64_t)w[4 * q + 1] << 32) | w[4 * q];
        const uint64_t hi = ((uint64_t)w[4 * q + 3] << 32) | w[4 * q + 2];
        const u128 a = mod128(hi, lo, CC);
        acan[(size_t)r * M + 4 * blk + q] = make_ulonglong2((uint64_t)a, (uint64_t)(a >> 64));
    }
    if (blk & 1u) return;
    chacha20_block(key, blk >> 1, 2u, row0 + r, 0u, w);
    const uint64_t span = 2 * (uint64_t)noise + 1;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const uint64_t d = (((uint64_t)w[2 * q + 1] << 32) | w[2 * q]) % span;
        e[(size_t)r * M + 4 * blk + q] = (int32_t)d - (int32_t)noise;
    }
}

// Residue mod p of a canonical value C < 2^96 given as three 32-bit limbs, lifted to the centred
// representative of C mod Q first when `lift` (C > Q / 2 -> C - Q): |result| < 3.5 * 2^29.
__device__ __forceinline__ int32_t limbs_mod_p(ulonglong2 v, bool lift, const PrimeK &P, const Mod &md) {
    const uint32_t c0 = (uint32_t)v.x, c1 = (uint32_t)(v.x >> 32), c2 = (uint32_t)v.y;
    int32_t r = smontu(c0, P.r1, md) + smontu(c1, P.r2, md) + smontu(c2, P.r3, md);  // each |.| < p
    if (lift) r -= P.qmodp;
    return r;
}

// shat[pi][slot] = NTT(s)[slot] * m^-1 * (M/p)^-1 * R mod p, centred  (s = the secret key, zero padded)
template <int LOGM>
__global__ void __launch_bounds__((NttGeom<LOGM, LOGE>::T))
k_shat(const uint64_t *__restrict__ sk, int32_t *__restrict__ shat, PrimeSet PS, uint32_t n) {
    using G = NttGeom<LOGM, LOGE>;
    constexpr int M = G::M, T = G::T, E = G::E;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x;
    const uint32_t pi = blockIdx.x;
    const PrimeK P = PS[pi];
    const Mod md = mod_of(P);
    int32_t x[1][E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint32_t idx = tid + T * e;
        x[0][e] = idx < n ? (int32_t)(sk[idx] & 1) : 0;
    }
    ntt_forward<LOGM, 1, LOGE>(x, lds, P.twf, tid, md);
    // kappaR = R^2 m^-1 (M/p)^-1 * R: smont(x, kappaR) = x R^2 m^-1 e; one more Montgomery step
    // with 1 brings it down to x R m^-1 e
#pragma unroll
    for (int e = 0; e < E; e++) {
        int32_t u = smont(x[0][e], P.kappaR, md);
        u = smont(u, 1, md);
        shat[(size_t)pi * M + E * tid + e] = scentre(u, md);
    }
}

// y[r][pi][m] = (M/p)^-1 * (a_r (*) s) mod p (+ hoff), a lifted to (-Q/2, Q/2]
template <int LOGM>
__global__ void __launch_bounds__((NttGeom<LOGM, LOGE>::T))
k_polymul_s(const ulonglong2 *__restrict__ acan, const int32_t *__restrict__ shat,
            uint32_t *__restrict__ y, PrimeSet PS, const CrtConst *__restrict__ CC) {
    using G = NttGeom<LOGM, LOGE>;
    constexpr int M = G::M, T = G::T, E = G::E;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x;
    const uint32_t npr = PS[0].npr;
    const uint32_t r = blockIdx.x / npr, pi = blockIdx.x % npr;
    const PrimeK P = PS[pi];
    const Mod md = mod_of(P);
    const u128 halfQ = CC->halfQ;
    int32_t x[1][E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const ulonglong2 v = acan[(size_t)r * M + tid + T * e];
        const u128 C = ((u128)v.y << 64) | v.x;
        x[0][e] = sred(limbs_mod_p(v, C > halfQ, P, md), md);
    }
    ntt_forward<LOGM, 1, LOGE>(x, lds, P.twf, tid, md);
#pragma unroll
    for (int e = 0; e < E; e++)  // |x| < 3.95 * 2^29, |shat| <= p / 2: |product R^-1| < 0.75 * 2^29
        x[0][e] = smont(x[0][e], shat[(size_t)pi * M + E * tid + e], md);
    __syncthreads();
    ntt_inverse<LOGM, 1, LOGE>(x, lds, P.twi, tid, md);
#pragma unroll
    for (int e = 0; e < E; e++)
        y[((size_t)r * npr + pi) * M + tid + T * e] = condsub(sfull(x[0][e], md) + P.hoff, (uint32_t)P.p);
}

// canon[(r * 2 + col)][m]: col 0 = a_r + s_k G[row][0], col 1 = a_r (*) s + e_r + s_k G[row][1]
// (constant-term adds, fhe.jl:195-196); prod holds the canonical a_r (*) s.
__global__ void __launch_bounds__(256)
k_keygen_finish(const ulonglong2 *__restrict__ acan, const ulonglong2 *__restrict__ prod,
                const int32_t *__restrict__ e, const uint64_t *__restrict__ sk,
                ulonglong2 *__restrict__ canon, const CrtConst *__restrict__ CC, uint32_t row0,
                uint32_t R, uint32_t logm) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    const uint32_t M = 1u << logm;
    if (t >= R * M) return;
    const uint32_t i = t & (M - 1), r = t >> logm;
    const uint32_t grow = (row0 + r) & 3, kk = (row0 + r) >> 2;
    const u128 Q = CC->Q;
    u128 a = ((u128)acan[t].y << 64) | acan[t].x;
    u128 b = ((u128)prod[t].y << 64) | prod[t].x;
    const int32_t ev = e[t];
    b += ev >= 0 ? (u128)(uint32_t)ev : Q - (u128)(uint32_t)(-ev);
    if (b >= Q) b -= Q;
    if (i == 0 && (sk[kk] & 1)) {  // G = [1 0; B 0; 0 1; 0 B] (fhe.jl:119-122)
        const u128 g = (grow & 1) ? CC->B % Q : (u128)1;
        if (grow < 2) { a += g; if (a >= Q) a -= Q; }
        else { b += g; if (b >= Q) b -= Q; }
    }
    canon[((size_t)r * 2 + 0) * M + i] = make_ulonglong2((uint64_t)a, (uint64_t)(a >> 64));
    canon[((size_t)r * 2 + 1) * M + i] = make_ulonglong2((uint64_t)b, (uint64_t)(b >> 64));
}

// ---- k_key_transform ------------------------------------------------------------------------------
// BootstrapKey.key (fhe.jl:176-201) canonical residues -> device form: centred lift to
// (-Q/2, Q/2], residue mod p_i, scaled by kappa_i, forward NTT, slot order.
//   canon  [npolys][m] 16-byte residues (polys in [k][row][col] order)
//   keyhat [k][NPR][row*2+col][m]
template <int LOGM>
__global__ void __launch_bounds__((NttGeom<LOGM, LOGE>::T))
k_key_transform(const ulonglong2 *__restrict__ canon, int32_t *__restrict__ keyhat, PrimeSet PS,
                const CrtConst *__restrict__ CC, uint32_t poly0, uint32_t *__restrict__ bad) {
    using G = NttGeom<LOGM, LOGE>;
    constexpr int M = G::M, T = G::T, E = G::E;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x;
    const uint32_t npr = PS[0].npr;
    const uint32_t pl = blockIdx.x / npr;  // polynomial within this staging batch
    const uint32_t pi = blockIdx.x % npr;
    const PrimeK P = PS[pi];
    const Mod md = mod_of(P);
    const u128 halfQ = CC->halfQ;
    int32_t x[1][E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const ulonglong2 v = canon[(size_t)pl * M + tid + T * e];
        const u128 C = ((u128)v.y << 64) | v.x;
        if (C >= CC->Q) *bad = 1u;  // not a canonical residue: reported by the upload call
        x[0][e] = smont(limbs_mod_p(v, C > halfQ, P, md), P.kappaR, md);  // |.| < 0.72 * 2^29
    }
    ntt_forward<LOGM, 1, LOGE>(x, lds, P.twf, tid, md);
    const uint32_t pg = poly0 + pl;  // global polynomial index = k * 8 + row * 2 + col
    int32_t *dst = keyhat + (((size_t)(pg >> 3) * npr + pi) * 8 + (pg & 7)) * M + E * tid;
#pragma unroll
    for (int h = 0; h < E / 4; h++) {
        int32_t o[4];
#pragma unroll
        for (int t = 0; t < 4; t++) o[t] = scentre(sred(sred_floor(x[0][4 * h + t], md), md), md);  // [-(p-1)/2, (p-1)/2]
        reinterpret_cast<int4 *>(dst)[h] = make_int4(o[0], o[1], o[2], o[3]);
    }
}

// ---- k_key_derive -------------------------------------------------------------------------------------
// The key of a ctx's smaller basis (primes 0 .. nps-1, deterministic flatten) from the key of its larger
// one (primes 0 .. npb-1, randomised flatten): the device form is scaled by (M_rns / p_i)^-1 mod p_i,
// and M_big = M_small * (the extra primes), so keyhat_small = keyhat_big * (product of the extra primes)
// mod p_i -- one constant per prime, no transform.  fac.f[i] = that constant in Montgomery form, centred.
// The result is the centred canonical residue, i.e. byte for byte what k_key_transform writes for the
// smaller basis.  One thread per element of the smaller key: keyhat[k][prime][row * 2 + col][slot].
struct KeyFactors {
    int32_t f[NPR_MAX];
};
__global__ void __launch_bounds__(256)
k_key_derive(const int32_t *__restrict__ big, int32_t *__restrict__ small, PrimeSet PS, KeyFactors fac,
             uint32_t npb, uint32_t nps, uint32_t logm, size_t total) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const size_t per_prime = (size_t)8 << logm;                 // 8 polynomials of m slots
    const size_t e = t % per_prime;
    const uint32_t i = (uint32_t)((t / per_prime) % nps);
    const size_t k = t / (per_prime * nps);
    const PrimeK P = PS[i];
    const Mod md = mod_of(P);
    const int32_t v = big[(k * npb + i) * per_prime + e];
    small[t] = scentre(sred(smont(v, fac.f[i], md), md), md);
}

// ---- RNS2Number boundary conversions (src/rns.jl, BASELINE.json config 4) --------------------------
// The reference's two-modulus residue type holds a coefficient of Z_Q, Q = m1 m2, as
// (v1, v2) = (x mod m1, x mod m2) (rns.jl:16-18) and converts back with the Fermat idempotents
// c1 = m2^(m1-1), c2 = m1^(m2-1) mod Q: x = (v1 c1 + v2 c2) mod Q (rns.jl:32-40).  With
// c1 = m2 (m2^-1 mod m1) and c2 = m1 (m1^-1 mod m2) that is
//   x = ((v1 i21) mod m1) m2 + ((v2 i12) mod m2) m1,  minus Q if >= Q.
// Both kernels work in place on 16-byte elements ({v1, v2} <-> {x lo, x hi}).
struct Rns2Const {
    uint64_t m1, m2;   // limb moduli, distinct primes below 2^47
    uint64_t i21, i12; // m2^-1 mod m1, m1^-1 mod m2
    double inv1, inv2; // 1 / m1, 1 / m2
};
__global__ void __launch_bounds__(256)
k_rns2_to_canon(ulonglong2 *__restrict__ buf, size_t count, Rns2Const rc,
                const CrtConst *__restrict__ CC, uint32_t *__restrict__ bad) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= count) return;
    const ulonglong2 v = buf[t];
    if (v.x >= rc.m1 || v.y >= rc.m2) { *bad = 1u; return; }  // not an RNS2Number of these moduli
    const u128 h1 = mod_wide((u128)v.x * rc.i21, (u128)rc.m1, rc.inv1, nullptr);
    const u128 h2 = mod_wide((u128)v.y * rc.i12, (u128)rc.m2, rc.inv2, nullptr);
    u128 x = h1 * rc.m2 + h2 * rc.m1;
    if (x >= CC->Q) x -= CC->Q;
    buf[t] = make_ulonglong2((uint64_t)x, (uint64_t)(x >> 64));
}
__global__ void __launch_bounds__(256)
k_canon_to_rns2(ulonglong2 *__restrict__ buf, size_t count, Rns2Const rc) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= count) return;
    const ulonglong2 v = buf[t];
    const u128 x = ((u128)v.y << 64) | v.x;  // < Q = m1 m2: both quotients are below 2^47
    buf[t] = make_ulonglong2((uint64_t)mod_wide(x, (u128)rc.m1, rc.inv1, nullptr),
                             (uint64_t)mod_wide(x, (u128)rc.m2, rc.inv2, nullptr));
}

// ---- k_debug_ntt ----------------------------------------------------------------------------------
template <int LOGM>
__global__ void __launch_bounds__((NttGeom<LOGM, LOGE>::T))
k_debug_ntt(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, PrimeSet PS, uint32_t pi,
            uint32_t inverse) {
    using G = NttGeom<LOGM, LOGE>;
    constexpr int T = G::T, E = G::E;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x;
    const PrimeK P = PS[pi];
    const Mod md = mod_of(P);
    int32_t x[1][E];
    if (!inverse) {
#pragma unroll
        for (int e = 0; e < E; e++) x[0][e] = (int32_t)in[tid + T * e];  // [0, p)
        ntt_forward<LOGM, 1, LOGE>(x, lds, P.twf, tid, md);
#pragma unroll
        for (int e = 0; e < E; e++) out[E * tid + e] = sfull(sred_floor(x[0][e], md), md);
    } else {
#pragma unroll
        for (int e = 0; e < E; e++) x[0][e] = sred((int32_t)in[E * tid + e], md);
        ntt_inverse<LOGM, 1, LOGE>(x, lds, P.twi, tid, md);
#pragma unroll
        for (int e = 0; e < E; e++) out[tid + T * e] = scanon(smont(x[0][e], P.minvR, md), md);
    }
}

}  // namespace sgfhe
