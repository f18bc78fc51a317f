// Word-size modular arithmetic for the RNS primes (p < 2^30) on gfx950.
//
// The native integer multiplier of a CDNA4 lane is 32 x 32 (v_mul_lo_u32 / v_mul_hi_u32 /
// v_mad_u64_u32), so the wide ring Z_Q (Q up to 94 bits; DarkIntegers MgModUInt{UInt128, Q} in
// the reference, src/fhe.jl:83-85,104) is replaced on the device by exact integer arithmetic in
// a residue number system of 30-bit NTT primes.  Everything here is exact; laziness ranges are
// stated per function.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sgfhe {

// Conditional subtraction: x >= m ? x - m : x.  Written through the borrow so that hipcc emits
// v_sub_co_u32 + v_cndmask_b32 (measured 1.80 add-equivalents for the pair on gfx950) instead
// of v_sub_u32 + v_min_u32 (2.83): tools/ubench_int.hip.
__device__ __forceinline__ uint32_t condsub(uint32_t x, uint32_t m) {
    uint32_t d;
    const bool borrow = __builtin_usub_overflow(x, m, &d);
    return borrow ? x : d;
}
// x in [0, 2p) -> [0, p).
__device__ __forceinline__ uint32_t csub(uint32_t x, uint32_t p) { return condsub(x, p); }

// Montgomery reduction, R = 2^32: T < p * 2^32 -> T * R^-1 mod p in [0, 2p).
// ninv = -p^-1 mod 2^32.
__device__ __forceinline__ uint32_t redc64(uint64_t T, uint32_t p, uint32_t ninv) {
    uint32_t tlo = (uint32_t)T, thi = (uint32_t)(T >> 32);
    uint32_t mq = tlo * ninv;
    uint32_t h = __umulhi(mq, p);
    // tlo + lo(mq * p) == 0 mod 2^32, carry out iff tlo != 0
    return thi + h + (tlo != 0u);
}

// The same reduction written for v_mad_u64_u32: hi32(T + (T_lo * ninv) * p).  Valid for any
// T < 2^64 - 2^32 p; returns T * R^-1 mod p in [0, T / 2^32 + p).
__device__ __forceinline__ uint32_t redc_mad(uint64_t T, uint32_t p, uint32_t ninv) {
    const uint32_t mq = (uint32_t)T * ninv;
    return (uint32_t)(((uint64_t)mq * p + T) >> 32);
}

// a * b * R^-1 mod p in [0, p); a * b < p * 2^32 required (e.g. a < 2^32, b < p).
__device__ __forceinline__ uint32_t mont_mul(uint32_t a, uint32_t b, uint32_t p, uint32_t ninv) {
    return csub(redc64((uint64_t)a * b, p, ninv), p);
}

// Modulus record kept in registers by the NTT code.
struct Mod {
    uint32_t p, ninv, p2;  // prime, -p^-1 mod 2^32, 2p
};

// Lazy Montgomery multiplication by a constant held in Montgomery form (wM = w * 2^32 mod p):
// returns w * y mod p in [0, 2p) for any y < 2^32.  Two v_mad_u64_u32 and one v_mul_lo_u32
// (measured on gfx950: 1.73 + 1.63 + 1.73 add-equivalents, against 5.8 for the Shoup form
// mul_hi + 2 mul_lo + sub; tools/ubench_int.hip).
__device__ __forceinline__ uint32_t mont_lazy(uint32_t y, uint32_t wM, const Mod &md) {
    const uint64_t T = (uint64_t)wM * y;             // < p * 2^32
    const uint32_t mq = (uint32_t)T * md.ninv;
    const uint64_t U = (uint64_t)mq * md.p + T;      // low word cancels; < 2p * 2^32
    return (uint32_t)(U >> 32);
}

// Forward (Cooley-Tukey) Harvey butterfly: X, Y in [0, 4p) -> X + wY, X - wY in [0, 4p).
__device__ __forceinline__ void bfly_fwd(uint32_t &X, uint32_t &Y, uint32_t wM, const Mod &md) {
    const uint32_t x = condsub(X, md.p2);  // [0, 2p)
    const uint32_t t = mont_lazy(Y, wM, md);
    X = x + t;
    Y = x + md.p2 - t;
}

// Inverse (Gentleman-Sande) Harvey butterfly: X, Y in [0, 2p) -> X + Y, w (X - Y) in [0, 2p).
__device__ __forceinline__ void bfly_inv(uint32_t &X, uint32_t &Y, uint32_t wM, const Mod &md) {
    const uint32_t s = X + Y;
    const uint32_t t = X + md.p2 - Y;
    X = condsub(s, md.p2);
    Y = mont_lazy(t, wM, md);
}

}  // namespace sgfhe
