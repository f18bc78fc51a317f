/*
 * gate_demo.c -- the reference's README example (README.md:10-29 of nucypher/SGFHE.jl) in plain C
 * over the C ABI of libsgfhe_hip.so alone (include/sgfhe_hip.h): private key, bootstrap key on the
 * GPU, encryption of a block of n bits, split into LWEs, one batch of gate bootstraps, decryption.
 * No Python, no Julia: what a host in any language has to call.
 *
 *   gcc -std=c99 -O2 -I include examples/gate_demo.c -L sgfhe.jl_amd/csrc -lsgfhe_hip \
 *       -Wl,-rpath,$PWD/sgfhe.jl_amd/csrc -o examples/gate_demo && examples/gate_demo
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sgfhe_hip.h"

#define CHECK(call)                                                                               \
    do {                                                                                          \
        int32_t rc_ = (call);                                                                     \
        if (rc_ != SGFHE_OK) {                                                                    \
            fprintf(stderr, "%s -> %d: %s\n", #call, (int)rc_, ctx ? sgfhe_last_error_string(ctx) : ""); \
            return 1;                                                                             \
        }                                                                                         \
    } while (0)

/* xorshift64*: the demo's own generator (the draws of encrypt are arguments of the C ABI) */
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd(void) {
    rng_state ^= rng_state >> 12;
    rng_state ^= rng_state << 25;
    rng_state ^= rng_state >> 27;
    return rng_state * 0x2545F4914F6CDD1Dull;
}

int main(void) {
    /* Params(64) (src/fhe.jl:43-97) */
    enum { N = 64, M = 512 };
    sgfhe_params p;
    memset(&p, 0, sizeof p);
    p.n = N;
    p.r = 16 * N;
    p.m = M;
    p.ell = 2;
    p.Q[0] = 5494391545392009217ull;
    p.B[0] = 2348810240ull;
    p.DQ_tilde[0] = 686798943174001152ull;
    const int64_t w_range = (int64_t)(p.r / 4 / 8);          /* Dr / 8 (src/fhe.jl:318) */

    sgfhe_ctx *ctx = NULL;
    if (sgfhe_abi_version() != SGFHE_ABI_VERSION) {
        fprintf(stderr, "library implements ABI %u, header is %u\n", sgfhe_abi_version(), SGFHE_ABI_VERSION);
        return 1;
    }
    CHECK(sgfhe_ctx_create(&p, 0, &ctx));

    /* PrivateKey (src/fhe.jl:130-138) and BootstrapKey on the device (src/fhe.jl:181-201) */
    uint64_t sk[N];
    uint8_t seed[32];
    for (int i = 0; i < N; i++) sk[i] = rnd() & 1;
    for (int i = 0; i < 32; i++) seed[i] = (uint8_t)rnd();
    CHECK(sgfhe_bkey_generate(ctx, sk, N, seed, N));

    /* encrypt a block of n bits (src/fhe.jl:310-328, 369-372) */
    uint8_t message[N], u[N];
    int64_t w[N];
    uint64_t a[N], b[N];
    for (int i = 0; i < N; i++) {
        message[i] = (uint8_t)(rnd() & 1);
        u[i] = (uint8_t)(rnd() & 1);
        w[i] = (int64_t)(rnd() % (uint64_t)(2 * w_range + 1)) - w_range;
    }
    CHECK(sgfhe_host_encrypt_private(&p, sk, u, w, message, a, b));

    /* split_ciphertext (src/fhe.jl:287-290): n LWEs */
    static uint64_t lwe_a[N * N], lwe_b[N];
    CHECK(sgfhe_host_split_ciphertext(&p, a, b, N, lwe_a, lwe_b));
    uint8_t check[N];
    CHECK(sgfhe_host_decrypt_lwe(&p, sk, lwe_a, lwe_b, N, check));
    if (memcmp(check, message, N) != 0) { fprintf(stderr, "split / decrypt mismatch\n"); return 1; }

    /* bootstrap(bkey, nothing, bit_2i, bit_2i+1) for i < n / 2 in one batch (src/fhe.jl:608-621) */
    enum { BATCH = N / 2 };
    static uint64_t a1[BATCH * N], a2[BATCH * N], b1[BATCH], b2[BATCH], out[BATCH * 3 * (N + 1)];
    for (int t = 0; t < BATCH; t++) {
        memcpy(a1 + t * N, lwe_a + (2 * t) * N, N * sizeof(uint64_t));
        memcpy(a2 + t * N, lwe_a + (2 * t + 1) * N, N * sizeof(uint64_t));
        b1[t] = lwe_b[2 * t];
        b2[t] = lwe_b[2 * t + 1];
    }
    CHECK(sgfhe_bootstrap_batch(ctx, a1, b1, a2, b2, BATCH, out, 0));

    /* decrypt the three gates of every pair (src/fhe.jl:504-507) */
    int bad = 0;
    for (int g = 0; g < 3; g++) {
        static uint64_t ga[BATCH * N], gb[BATCH];
        uint8_t bits[BATCH];
        for (int t = 0; t < BATCH; t++) {
            const uint64_t *o = out + ((size_t)t * 3 + g) * (N + 1);
            memcpy(ga + t * N, o, N * sizeof(uint64_t));
            gb[t] = o[N];
        }
        CHECK(sgfhe_host_decrypt_lwe(&p, sk, ga, gb, BATCH, bits));
        for (int t = 0; t < BATCH; t++) {
            const int y1 = message[2 * t], y2 = message[2 * t + 1];
            const int want = g == 0 ? (y1 & y2) : g == 1 ? (y1 | y2) : (y1 ^ y2);
            bad += bits[t] != want;
        }
    }
    CHECK(sgfhe_ctx_destroy(ctx));
    ctx = NULL;
    if (bad) { fprintf(stderr, "%d gates decrypt wrongly\n", bad); return 1; }
    printf("gate_demo OK: %d AND / OR / XOR gate bootstraps at Params(64) decrypt correctly (%s, build %s)\n",
           BATCH, sgfhe_version(), sgfhe_build_id());
    return 0;
}
