/* The C restatement (oracle/sgfhe_oracle.c, test infrastructure) under AddressSanitizer + UBSan:
 * the checker that every parity test leans on must itself be memory-clean.  Params(64): key
 * generation, four gate bootstraps in the reference's shape and in the GPU path's algebra (equal
 * outputs), the truth table, pack_encrypted_bits of one ciphertext; round 4: the randomised flatten in
 * both loops (equal outputs, rows picked out of a larger call), its single-residue and draw-stream
 * entry points, and the NTT-domain loop over a small RNS2Number ring against the reference-shaped
 * limb-wise loop.  Built and run by tests/test_host_sanitizer.py. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sgfhe_oracle.h"

int main(void) {
    uint64_t w[10];
    if (sgo_params_make(64, w)) return 2;
    const size_t n = w[0], m = w[2];
    sgo_ctx *c = sgo_ctx_create(w);
    if (!c) return 3;
    uint64_t *sk = calloc(n, 8);
    sgo_private_key(c, 1, sk);
    uint8_t seed[32];
    memset(seed, 0, sizeof seed);
    seed[0] = 2;
    uint64_t *bkey = malloc(n * 8 * m * 16), *khat = malloc(n * 8 * m * 16);
    sgo_bootstrap_key(c, sk, seed, n, bkey, 1);
    const uint8_t bits[8] = {0, 0, 0, 1, 1, 0, 1, 1};
    uint64_t *a = malloc(8 * n * 8), b[8];
    sgo_lwe_encrypt_bits(c, sk, bits, 8, 3, a, b);
    uint64_t *a1 = malloc(4 * n * 8), *a2 = malloc(4 * n * 8), b1[4], b2[4];
    for (int i = 0; i < 4; i++) {
        memcpy(a1 + i * n, a + (2 * i) * n, n * 8);
        memcpy(a2 + i * n, a + (2 * i + 1) * n, n * 8);
        b1[i] = b[2 * i];
        b2[i] = b[2 * i + 1];
    }
    const size_t ow = 4 * 3 * (n + 1);
    uint64_t *out = calloc(ow, 8), *out2 = calloc(ow, 8), *acc = malloc(4 * 2 * m * 16);
    if (sgo_bootstrap_batch(c, bkey, a1, b1, a2, b2, 4, out, 0, n, acc, 1)) return 4;
    if (sgo_key_transform(c, bkey, khat, 1)) return 5;
    if (sgo_bootstrap_batch_opt(c, khat, a1, b1, a2, b2, 4, out2, 0, n, NULL, 1)) return 6;
    if (memcmp(out, out2, ow * 8)) { fprintf(stderr, "the two forms differ\n"); return 7; }
    for (int i = 0; i < 4; i++) {
        const int y1 = bits[2 * i], y2 = bits[2 * i + 1], want[3] = {y1 & y2, y1 | y2, y1 ^ y2};
        for (int g = 0; g < 3; g++) {
            const uint64_t *lwe = out + ((size_t)i * 3 + g) * (n + 1);
            if (sgo_lwe_decrypt_bit(c, sk, lwe, lwe[n]) != want[g]) { fprintf(stderr, "gate %d of pair %d\n", g, i); return 8; }
        }
    }
    /* pack_encrypted_bits of n LWEs (the 8 above, repeated) */
    uint64_t *pa = malloc(n * n * 8), *pb = malloc(n * 8), *pw = malloc(m * 8), *pv = malloc(m * 8);
    for (size_t i = 0; i < n; i++) {
        memcpy(pa + i * n, a + (i % 8) * n, n * 8);
        pb[i] = b[i % 8];
    }
    if (sgo_pack_encrypted_bits(c, bkey, pa, pb, pw, pv, 1)) return 9;
    uint64_t digest = 1469598103934665603ull;
    /* randomised flatten (utils.jl:198-241) on the ChaCha8 stream: both loops, then two rows picked
     * out of the same call by their stream indices */
    {
        uint8_t fkey[32];
        for (int i = 0; i < 32; i++) fkey[i] = (uint8_t)(11 + i);
        uint64_t *r1 = calloc(ow, 8), *r2 = calloc(ow, 8), *r3 = calloc(2 * 3 * (n + 1), 8);
        if (sgo_bootstrap_batch_rnd(c, 0, bkey, a1, b1, a2, b2, 4, r1, 0, n, NULL, 1, fkey, 2, 10, NULL)) return 10;
        if (sgo_bootstrap_batch_rnd(c, 1, khat, a1, b1, a2, b2, 4, r2, 0, n, acc, 1, fkey, 2, 10, NULL)) return 11;
        if (memcmp(r1, r2, ow * 8)) { fprintf(stderr, "randomised: the two forms differ\n"); return 12; }
        if (!memcmp(r1, out, ow * 8)) { fprintf(stderr, "randomised = deterministic\n"); return 13; }
        const uint32_t boots[2] = {13, 11};                 /* rows 3 and 1 of the call above */
        uint64_t *pa1 = malloc(2 * n * 8), *pa2 = malloc(2 * n * 8), pb1[2], pb2[2];
        for (int i = 0; i < 2; i++) {
            const int row = (int)boots[i] - 10;
            memcpy(pa1 + i * n, a1 + row * n, n * 8);
            memcpy(pa2 + i * n, a2 + row * n, n * 8);
            pb1[i] = b1[row];
            pb2[i] = b2[row];
        }
        if (sgo_bootstrap_batch_rnd(c, 1, khat, pa1, pb1, pa2, pb2, 2, r3, 0, n, NULL, 1, fkey, 2, 0, boots)) return 14;
        for (int i = 0; i < 2; i++)
            if (memcmp(r3 + (size_t)i * 3 * (n + 1), r1 + (size_t)(boots[i] - 10) * 3 * (n + 1), 3 * (n + 1) * 8)) {
                fprintf(stderr, "picked row %d differs\n", i);
                return 15;
            }
        int64_t *draws = malloc(m * 2 * 8);
        sgo_flatten_draws(c, fkey, 1, 5, 12, 2, draws);
        uint64_t av[2] = {12345678901234ull, 0}, fl[4];
        sgo_flatten_random(c, av, draws[0], draws[1], fl);
        for (size_t i = 0; i < ow; i++) digest = (digest ^ r1[i]) * 1099511628211ull;
        digest = (digest ^ fl[0] ^ fl[2] ^ (uint64_t)draws[2 * m - 1]) * 1099511628211ull;
        free(r1); free(r2); free(r3); free(pa1); free(pa2); free(draws);
    }
    /* the RNS2Number ring (src/rns.jl) on a small ring: NTT-domain limb-wise loop = reference-shaped one */
    {
        const uint64_t m1 = 33556993ull, m2 = 33560833ull;  /* primes = 1 mod 256 just above 2^25 */
        const size_t n2 = 16, mm = 128;
        uint64_t w2[10] = {n2, 16 * n2, mm, 2, 0, 0, m1, 0, 0, 0};
        const unsigned __int128 Q2 = (unsigned __int128)m1 * m2;
        w2[4] = (uint64_t)Q2; w2[5] = (uint64_t)(Q2 >> 64);
        w2[8] = (uint64_t)(Q2 / 8); w2[9] = (uint64_t)((Q2 / 8) >> 64);
        sgo_ctx *c2 = sgo_ctx_create(w2);
        if (!c2 || sgo_ctx_set_rns2(c2, m1, m2)) return 16;
        uint64_t *sk2 = calloc(n2, 8), *k2 = malloc(n2 * 8 * mm * 16), *kh2 = malloc(2 * n2 * 8 * mm * 16);
        sgo_private_key(c2, 5, sk2);
        sgo_bootstrap_key(c2, sk2, seed, 2, k2, 1);
        if (sgo_key_transform(c2, k2, kh2, 1)) return 17;
        uint64_t *la = malloc(4 * n2 * 8), lb[4];
        sgo_lwe_encrypt_bits(c2, sk2, bits, 4, 9, la, lb);
        uint64_t *o1 = calloc(2 * 3 * (n2 + 1), 8), *o2 = calloc(2 * 3 * (n2 + 1), 8);
        uint64_t lb1[2] = {lb[0], lb[2]}, lb2[2] = {lb[1], lb[3]};
        uint64_t *la1 = malloc(2 * n2 * 8), *la2 = malloc(2 * n2 * 8);
        memcpy(la1, la, n2 * 8); memcpy(la1 + n2, la + 2 * n2, n2 * 8);
        memcpy(la2, la + n2, n2 * 8); memcpy(la2 + n2, la + 3 * n2, n2 * 8);
        if (sgo_bootstrap_batch(c2, k2, la1, lb1, la2, lb2, 2, o1, 0, n2, NULL, 1)) return 18;
        if (sgo_bootstrap_batch_opt(c2, kh2, la1, lb1, la2, lb2, 2, o2, 0, n2, NULL, 1)) return 19;
        if (memcmp(o1, o2, 2 * 3 * (n2 + 1) * 8)) { fprintf(stderr, "rns2: the two forms differ\n"); return 20; }
        for (size_t i = 0; i < 2 * 3 * (n2 + 1); i++) digest = (digest ^ o1[i]) * 1099511628211ull;
        free(sk2); free(k2); free(kh2); free(la); free(o1); free(o2); free(la1); free(la2);
        sgo_ctx_destroy(c2);
    }
    for (size_t i = 0; i < ow; i++) digest = (digest ^ out[i]) * 1099511628211ull;
    for (size_t i = 0; i < m; i++) digest = (digest ^ pw[i] ^ (pv[i] << 20)) * 1099511628211ull;
    printf("%016llx\n", (unsigned long long)digest);
    free(pa); free(pb); free(pw); free(pv); free(out); free(out2); free(acc); free(a1); free(a2); free(a);
    free(bkey); free(khat); free(sk);
    sgo_ctx_destroy(c);
    return 0;
}
