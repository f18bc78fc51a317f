/* The C restatement (oracle/sgfhe_oracle.c, test infrastructure) under AddressSanitizer + UBSan:
 * the checker that every parity test leans on must itself be memory-clean.  Params(64): key
 * generation, four gate bootstraps in the reference's shape and in the GPU path's algebra (equal
 * outputs), the truth table, pack_encrypted_bits of one ciphertext.  Built and run by
 * tests/test_host_sanitizer.py. */
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
    for (size_t i = 0; i < ow; i++) digest = (digest ^ out[i]) * 1099511628211ull;
    for (size_t i = 0; i < m; i++) digest = (digest ^ pw[i] ^ (pv[i] << 20)) * 1099511628211ull;
    printf("%016llx\n", (unsigned long long)digest);
    free(pa); free(pb); free(pw); free(pv); free(out); free(out2); free(acc); free(a1); free(a2); free(a);
    free(bkey); free(khat); free(sk);
    sgo_ctx_destroy(c);
    return 0;
}
