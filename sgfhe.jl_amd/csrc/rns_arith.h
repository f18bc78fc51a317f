// Word-size modular arithmetic for the RNS primes (p < 2^29) on gfx950.
//
// The native integer multiplier of a CDNA4 lane is 32 x 32 (v_mul_lo_u32 / v_mul_hi / v_mad_64_32),
// so the wide ring Z_Q (Q up to 94 bits; DarkIntegers MgModUInt{UInt128, Q} in the reference,
// src/fhe.jl:83-85,104) is replaced on the device by exact integer arithmetic in a residue number
// system of 29-bit NTT primes.  Everything here is exact.
//
// Representation: SIGNED lazy residues.  A value is an int32 anywhere in (-2^31, 2^31); with
// p < 2^29 that is a window of (-4p, 4p), so several butterfly stages can add and subtract without
// any conditional correction and without the +2p offsets an unsigned lazy form needs.  Products go
// through a signed Montgomery reduction whose result is already centred (|t| < 0.75 p), and a
// value is pulled back to about (-p/2, p/2) by `sred` only where the range analysis below calls
// for it (once per radix-16 pass in the forward transform -- there in the cheaper floor form
// `sred_floor`, to [0, p] -- and every second stage in the inverse).
// Measured on MI355X (tools/ubench_bfly.hip, profiles/r02_ubench_bfly.txt): 0.75 x the time of the
// unsigned 30-bit Harvey butterfly of round 1 in the forward pass, 0.78 x in the inverse.
//
// Range bookkeeping is in units of 2^29 (> p); every bound is re-derived, with the exact pass
// structure, by tests/rns_model.py (RangeModel) for every prime and ring size.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sgfhe {

// Modulus record kept in registers by the NTT code.
struct Mod {
    int32_t p;      // prime, < 2^29
    int32_t negp;   // -p
    uint32_t pinv;  // p^-1 mod 2^32
};

// Signed Montgomery reduction, R = 2^32:  T R^-1 mod p  for any |T| < 2^62.
//   m = T_lo p^-1 mod 2^32 (as int32), so T - m p == 0 mod 2^32 and the high word is the exact
//   quotient: |result| <= |T| / 2^32 + p / 2.
// Two instructions: v_mul_lo_u32, v_mad_i64_i32.
__device__ __forceinline__ int32_t sredc(int64_t T, const Mod &md) {
    const int32_t m = (int32_t)((uint32_t)T * md.pinv);
    return (int32_t)(((int64_t)m * md.negp + T) >> 32);
}

// a w R^-1 mod p for any int32 a and |w| <= p / 2 (constants are held centred, in Montgomery form
// w = v R mod p, so the result is a v mod p):  |result| <= |a| / 16 + p / 2  <  0.75 p.
// v_mad_i64_i32, v_mul_lo_u32, v_mad_i64_i32.
__device__ __forceinline__ int32_t smont(int32_t a, int32_t w, const Mod &md) {
    return sredc((int64_t)a * w, md);
}
// The same for an unsigned 32-bit limb a (key upload: limbs of a canonical residue).
__device__ __forceinline__ int32_t smontu(uint32_t a, int32_t w, const Mod &md) {
    return sredc((int64_t)(uint64_t)a * w, md);
}

// Range reduction: any x with x + 2^28 < 2^31 (i.e. x < 3.5 * 2^29) and x > -2^31 + ...:
//   q = round(x / 2^29),  r = x - q p = (x - q 2^29) + q (2^29 - p),
//   |r| <= 2^28 + 4 (2^29 - p)   ( < 0.511 * 2^29 for the primes in use ).
__device__ __forceinline__ int32_t sred(int32_t x, const Mod &md) {
    const int32_t q = (x + (1 << 28)) >> 29;
    return x - q * md.p;
}

// Floor form for ANY int32 x:  q = floor(x / 2^29) in [-4, 3],  r = x - q p = (x mod 2^29) + q delta
// with delta = 2^29 - p:  r in [-4 delta, 2^29 + 3 delta).  One instruction less than `sred`
// (shift, multiply, subtract) and no precondition; the price is a bound of 1.0 instead of 0.51.
// Used where the next operations only add to the value (the forward transform).
__device__ __forceinline__ int32_t sred_floor(int32_t x, const Mod &md) { return x - (x >> 29) * md.p; }

// x in (-p, p) -> the canonical representative in [0, p).
__device__ __forceinline__ uint32_t scanon(int32_t x, const Mod &md) {
    return (uint32_t)(x + ((x >> 31) & md.p));
}
// any |x| < 3.5 * 2^29 -> [0, p)
__device__ __forceinline__ uint32_t sfull(int32_t x, const Mod &md) { return scanon(sred(x, md), md); }

// x in (-p, p) -> the centred canonical representative in [-(p-1)/2, (p-1)/2].
__device__ __forceinline__ int32_t scentre(int32_t x, const Mod &md) {
    const int32_t h = (md.p - 1) >> 1;
    if (x > h) x -= md.p;
    if (x < -h) x += md.p;
    return x;
}

// Conditional subtraction on unsigned words: x >= m ? x - m : x (v_sub_co_u32 + v_cndmask_b32).
__device__ __forceinline__ uint32_t condsub(uint32_t x, uint32_t m) {
    uint32_t d;
    const bool borrow = __builtin_usub_overflow(x, m, &d);
    return borrow ? x : d;
}

// Forward (Cooley-Tukey) butterfly: X' = X + w Y, Y' = X - w Y.  Any Y; |X'|, |Y'| <= |X| + |t|,
// |t| <= |Y| / 16 + p / 2.  Five instructions (three multiplies, an add, a sub).
__device__ __forceinline__ void bfly_fwd(int32_t &X, int32_t &Y, int32_t wM, const Mod &md) {
    const int32_t t = smont(Y, wM, md);
    const int32_t x = X;
    X = x + t;
    Y = x - t;
}

// Inverse (Gentleman-Sande) butterfly: X' = X + Y (range-reduced when RED), Y' = w (X - Y).
// Needs |X| + |Y| < 2^31 (and < 3.5 * 2^29 when RED).
// (red is a compile-time constant at every call site)
__device__ __forceinline__ void bfly_inv(int32_t &X, int32_t &Y, int32_t wM, const Mod &md, bool red) {
    const int32_t s = X + Y;
    const int32_t d = X - Y;
    X = red ? sred(s, md) : s;
    Y = smont(d, wM, md);
}

}  // namespace sgfhe
