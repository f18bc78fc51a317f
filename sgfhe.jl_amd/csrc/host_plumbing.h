// Host side of the steps either side of the hot path: the ciphertext
// plumbing of nucypher/SGFHE.jl that turns a message into the LWE pairs bootstrap() consumes and its
// outputs back into bits -- extract / split_ciphertext, private encryption, the space-optimal
// form and its normalisation, decryption of LWEs and RLWEs, bit (un)packing.  Plain C++, no
// device: these are O(n^2) word operations on polynomials of length n <= 2048 over Z_r, r = 2^(t+1).
// Reference citations are relative to /root/reference.
//
// Every function is pure: where the reference draws from an rng (u and w of _encrypt_private,
// src/fhe.jl:315,319) the caller passes the draws, so any host language keeps its own generator
// and two implementations can be compared bit for bit.
#pragma once

#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <vector>

namespace sgfhe_host {

// ---- SHAKE-256 (FIPS 202) ---------------------------------------------------------------------------
// prng_expand (src/utils.jl:63-68) is marked "should be done with SHAKE-128 or 256" in the reference
// (it seeds a MersenneTwister with hash(seq) there); this is that primitive.
inline void keccak_f1600(uint64_t (&s)[25]) {
    static const uint64_t RC[24] = {
        0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull,
        0x000000000000808bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
        0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
        0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull,
        0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
        0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    static const int ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14,
                                27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    static const int PIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4,
                                15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    for (int round = 0; round < 24; round++) {
        uint64_t bc[5];
        for (int i = 0; i < 5; i++) bc[i] = s[i] ^ s[i + 5] ^ s[i + 10] ^ s[i + 15] ^ s[i + 20];
        for (int i = 0; i < 5; i++) {
            const uint64_t t = bc[(i + 4) % 5] ^ ((bc[(i + 1) % 5] << 1) | (bc[(i + 1) % 5] >> 63));
            for (int j = 0; j < 25; j += 5) s[j + i] ^= t;
        }
        uint64_t t = s[1];
        for (int i = 0; i < 24; i++) {
            const int j = PIL[i];
            const uint64_t b = s[j];
            s[j] = (t << ROT[i]) | (t >> (64 - ROT[i]));
            t = b;
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = s[j + i];
            for (int i = 0; i < 5; i++) s[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        s[0] ^= RC[round];
    }
}
inline void shake256(const uint8_t *in, size_t inlen, uint8_t *out, size_t outlen) {
    const size_t rate = 136;
    uint64_t s[25];
    memset(s, 0, sizeof s);
    uint8_t block[136];
    while (inlen >= rate) {
        for (size_t i = 0; i < rate / 8; i++) {
            uint64_t w;
            memcpy(&w, in + 8 * i, 8);   // little-endian host (x86-64 / the only targets of this build)
            s[i] ^= w;
        }
        keccak_f1600(s);
        in += rate;
        inlen -= rate;
    }
    memset(block, 0, rate);
    memcpy(block, in, inlen);
    block[inlen] ^= 0x1F;
    block[rate - 1] ^= 0x80;
    for (size_t i = 0; i < rate / 8; i++) {
        uint64_t w;
        memcpy(&w, block + 8 * i, 8);
        s[i] ^= w;
    }
    keccak_f1600(s);
    while (outlen) {
        const size_t take = outlen < rate ? outlen : rate;
        memcpy(out, s, take);
        out += take;
        outlen -= take;
        if (outlen) keccak_f1600(s);
    }
}

// ---- bits -----------------------------------------------------------------------------------------
// packbits (src/utils.jl:36-42): bits[t][n] (row i = bit i of every element) -> n integers
inline void packbits(const uint8_t *bits, size_t t, size_t n, uint64_t *out) {
    for (size_t j = 0; j < n; j++) out[j] = 0;
    for (size_t i = 0; i < t; i++)
        for (size_t j = 0; j < n; j++) out[j] += (uint64_t)(bits[i * n + j] & 1) << i;
}
// unpackbits (src/utils.jl:48-54)
inline void unpackbits(const uint64_t *arr, size_t n, size_t itemsize, uint8_t *bits) {
    for (size_t i = 0; i < itemsize; i++)
        for (size_t j = 0; j < n; j++) bits[i * n + j] = (uint8_t)((arr[j] >> i) & 1);
}
// prng_expand (src/utils.jl:63-68) with SHAKE-256: the n seed bits are packed most significant bit
// first into bytes, the stream is read most significant bit first as a factor x n bit matrix.
inline void prng_expand(const uint8_t *seq, size_t n, size_t factor, uint64_t *out) {
    std::vector<uint8_t> packed((n + 7) / 8, 0);
    for (size_t i = 0; i < n; i++)
        if (seq[i] & 1) packed[i / 8] |= (uint8_t)(0x80u >> (i % 8));
    std::vector<uint8_t> stream((factor * n + 7) / 8);
    shake256(packed.data(), packed.size(), stream.data(), stream.size());
    for (size_t j = 0; j < n; j++) out[j] = 0;
    for (size_t i = 0; i < factor; i++)
        for (size_t j = 0; j < n; j++) {
            const size_t bit = i * n + j;
            out[j] += (uint64_t)((stream[bit / 8] >> (7 - bit % 8)) & 1) << i;
        }
}

// ---- polynomials over Z_r, r a power of two ---------------------------------------------------------
// a * s mod (x^n + 1, r) (`Polynomial * Polynomial` of src/fhe.jl:322,479): wraps modulo 2^64, r | 2^64
inline void negacyclic_mul(const uint64_t *a, const uint64_t *s, size_t n, uint64_t rmask, uint64_t *out) {
    std::vector<uint64_t> full(2 * n, 0);
    for (size_t i = 0; i < n; i++) {
        const uint64_t si = s[i];
        if (!si) continue;
        for (size_t j = 0; j < n; j++) full[i + j] += a[j] * si;
    }
    for (size_t i = 0; i < n; i++) out[i] = (full[i] - full[i + n]) & rmask;
}

// extract(a, i, n) (src/fhe.jl:237-244), i 1-based, a of length N over Z_r
inline void extract(const uint64_t *a, size_t N, size_t i, size_t n, uint64_t rmask, uint64_t *out) {
    if (i < n) {
        for (size_t k = 0; k < i; k++) out[k] = a[i - 1 - k] & rmask;                    // a[i:-1:1]
        for (size_t k = 0; k < n - i; k++) out[i + k] = (0 - a[N - 1 - k]) & rmask;      // -a[end:-1:end-(n-i-1)]
    } else {
        for (size_t k = 0; k < n; k++) out[k] = a[i - 1 - k] & rmask;                    // a[i:-1:i-n+1]
    }
}

// ---- polynomials over Z_q, q the public-key modulus (q < 2^31) -------------------------------------
// k * u mod (x^n + 1, q) for k in [0, q) and a short signed u (|u_i| <= 1: the private key bits,
// the ternary u of _encrypt_public): `Polynomial * Polynomial` of src/fhe.jl:164,399-400.
inline void negacyclic_mul_short(const uint64_t *k, const int8_t *u, size_t n, uint64_t q, int64_t *out) {
    std::vector<int64_t> full(2 * n, 0);
    for (size_t i = 0; i < n; i++) {
        const int64_t ui = u[i];
        if (!ui) continue;
        for (size_t j = 0; j < n; j++) full[i + j] += (int64_t)k[j] * ui;   // |.| < n q < 2^42
    }
    for (size_t i = 0; i < n; i++) {
        int64_t v = (full[i] - full[i + n]) % (int64_t)q;
        out[i] = v < 0 ? v + (int64_t)q : v;
    }
}
// rescale(new_max, x, old_max, round_result) (src/utils.jl:78-92): floor or round of
// x new_max / old_max, the rounded value new_max wrapping to 0; x new_max < 2^64 here
inline uint64_t rescale(uint64_t new_max, uint64_t x, uint64_t old_max, bool round_result) {
    const uint64_t prod = x * new_max;
    uint64_t quo = prod / old_max;
    if (round_result) {
        const uint64_t rem = prod % old_max;
        if (rem >= old_max / 2 + (old_max & 1)) quo++;
        if (quo == new_max) quo = 0;
    }
    return quo;
}

}  // namespace sgfhe_host
