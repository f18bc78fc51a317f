/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * C restatement ("port") of the gate bootstrap of nucypher/SGFHE.jl (both flatten modes), in the
 * reference's own shape: 128-bit Montgomery residues (DarkIntegers MgModUInt{UInt128,Q},
 * /root/reference/src/fhe.jl:83-85,104), one full NTT polynomial multiply per
 * `Polynomial * Polynomial` call site (8 per external product, fhe.jl:527-528).
 *
 * PARITY STATUS: "parity unpinned" at the bit level against the Julia build (the reference has
 * no golden vectors and Julia / DarkIntegers.jl ~0.1.0 are absent here); pinned mathematically
 * (exact arithmetic on canonical representatives) and by the reference's property tests,
 * restated in tests/.  Cross-checked bit for bit against oracle/bigint_oracle.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * All 128-bit values cross the ABI as little-endian {lo, hi} uint64 pairs.
 */
#ifndef SGFHE_ORACLE_H
#define SGFHE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sgo_ctx sgo_ctx;

/* words = {n, r, m, ell, Q_lo, Q_hi, B_lo, B_hi, DQtilde_lo, DQtilde_hi} (fhe.jl:27-41). */
sgo_ctx *sgo_ctx_create(const uint64_t *words);
void sgo_ctx_destroy(sgo_ctx *ctx);
/* 1 if Q is prime with 2m | Q-1 (NTT multiply), 0 if the schoolbook multiply is used. */
int sgo_ctx_uses_ntt(const sgo_ctx *ctx);
/* RNS2Number ring of src/rns.jl (BASELINE.json config 4): Q = m1 m2 with two NTT-friendly primes
 * (rule of src/fhe2.jl:57-58).  Polynomial products are then computed as the reference's RNS type
 * does: limb-wise (rns.jl:51-60, one NTT multiply per limb) with the conversions of rns.jl:16-18
 * (split) and rns.jl:32-40 (CRT).  Returns 0 on success. */
int sgo_ctx_set_rns2(sgo_ctx *ctx, uint64_t m1, uint64_t m2);
int sgo_ctx_uses_rns2(const sgo_ctx *ctx);

/* utils.jl:7-28 find_modulus; qmax_* = 0,0 means "no upper bound". Returns 0 on success. */
int sgo_find_modulus(uint64_t n, const uint64_t *qmin, const uint64_t *qmax, uint64_t *out);
/* fhe.jl:43-97 Params(n): fills words[10] as for sgo_ctx_create. Returns 0 on success. */
int sgo_params_make(uint64_t n, uint64_t *words);

/* utils.jl:78-92 rescale(new_max, x, old_max, round_result) on 128-bit operands. */
void sgo_rescale(const uint64_t *new_max, const uint64_t *x, const uint64_t *old_max,
                 int round_result, uint64_t *out);
/* utils.jl:155-189 deterministic flatten of one residue: out[ell] residues mod Q. */
void sgo_flatten(const sgo_ctx *ctx, const uint64_t *a, uint64_t *out);
/* Exact negacyclic product mod (x^m + 1, Q): DarkIntegers Polynomial * Polynomial. */
void sgo_poly_mul(const sgo_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out);
void sgo_poly_mul_schoolbook(const sgo_ctx *ctx, const uint64_t *a, const uint64_t *b,
                             uint64_t *out);
/* fhe.jl:519-530 external_product(nothing, a, b, A[4][2], B, ell). */
void sgo_external_product(const sgo_ctx *ctx, const uint64_t *a, const uint64_t *b,
                          const uint64_t *A, uint64_t *a_res, uint64_t *b_res);

/* Private key and LWE test plumbing with the build's SplitMix64 (same draws as bigint_oracle.py). */
void sgo_private_key(const sgo_ctx *ctx, uint64_t seed, uint64_t *sk /* [n] bits */);
/* fhe.jl:181-201; bkey is [n][4][2][m] residues (2 words each); noise bound as fhe.jl:194.
 * Randomness: ChaCha20 streams of the 32-byte seed (layout in sgfhe_oracle.c), the same as
 * sgfhe_bkey_generate of the HIP engine. */
void sgo_bootstrap_key(const sgo_ctx *ctx, const uint64_t *sk, const uint8_t *seed, uint64_t noise,
                       uint64_t *bkey, int threads);
/* `count` LWEs of bits[count] from one generator seeded with `seed` (fhe.jl:310-328,287-290). */
void sgo_lwe_encrypt_bits(const sgo_ctx *ctx, const uint64_t *sk, const uint8_t *bits,
                          size_t count, uint64_t seed, uint64_t *a /* [count][n] */,
                          uint64_t *b /* [count] */);
/* fhe.jl:504-507 */
int sgo_lwe_decrypt_bit(const sgo_ctx *ctx, const uint64_t *sk, const uint64_t *a, uint64_t b);

/*
 * fhe.jl:559-621.  out is [batch][3][n+1] (a[0..n) then b; order AND, OR, XOR):
 *   raw == 0: uint64 words in [0, r) after ModRed (fhe.jl:616-618)
 *   raw == 1: 128-bit residues mod Q ({lo, hi} pairs) as returned by _bootstrap_internal.
 * n_iters < n truncates the k-loop (timing samples / intermediate checks); acc_out, if not
 * NULL, receives the accumulator pair [batch][2][m] after the loop.
 * threads: OpenMP threads over the batch.  Returns 0 on success.
 */
int sgo_bootstrap_batch(const sgo_ctx *ctx, const uint64_t *bkey, const uint64_t *a1,
                        const uint64_t *b1, const uint64_t *a2, const uint64_t *b2, size_t batch,
                        uint64_t *out, int raw, uint64_t n_iters, uint64_t *acc_out, int threads);

/*
 * `cpu_opt` of BASELINE.md section 3: the same bootstrap in the algebra of the GPU path -- key in
 * the NTT domain (sgo_key_transform), acc <- acc + (x^j - 1) sum_row u_row (*) C_k[row], i.e.
 * 4 forward + 2 inverse NTTs per iteration instead of 24 -- still 128-bit Montgomery arithmetic
 * mod Q, bit-identical outputs.  Prime NTT-friendly Q, or the RNS2Number ring (sgo_ctx_set_rns2):
 * there khat holds the key limb-wise, [2][n][4][2][m] residues mod m_limb (twice the size of bkey),
 * and an iteration is 2 x (4 + 2) limb NTTs with the CRT of rns.jl:32-40 per column.  Same arguments
 * as sgo_bootstrap_batch with khat in place of bkey.
 */
int sgo_key_transform(const sgo_ctx *ctx, const uint64_t *bkey, uint64_t *khat, int threads);
int sgo_bootstrap_batch_opt(const sgo_ctx *ctx, const uint64_t *khat, const uint64_t *a1,
                            const uint64_t *b1, const uint64_t *a2, const uint64_t *b2, size_t batch,
                            uint64_t *out, int raw, uint64_t n_iters, uint64_t *acc_out, int threads);

/*
 * The randomised flatten, flatten(rng::AbstractRNG, ...) of utils.jl:198-241, and bootstrap(bkey,
 * rng, ...) through it.  The reference draws from the caller's Julia rng (not reproducible outside
 * Julia); here the draws are the HIP engine's ChaCha8 counter stream (layout in sgfhe_oracle.c),
 * so the engine's randomised mode can be pinned word for word at full batch.
 *   sgo_flatten_random: one residue with the two draws x0, x1 in [-xmax, xmax] given: out[2] residues
 *   sgo_flatten_draws:  the draws [m][2] of one polynomial (accumulator cc, flatten tag y = k for the
 *                       flatten feeding k-loop iteration k, bootstrap `boot` of call `call`)
 *   sgo_bootstrap_batch_rnd: as sgo_bootstrap_batch (opt = 0) / sgo_bootstrap_batch_opt (opt = 1);
 *                       bootstrap t of the batch draws as bootstrap boot0 + t of call `call`, or as
 *                       bootstrap boots[t] when `boots` is not NULL (rows picked out of a larger call).
 */
void sgo_flatten_random(const sgo_ctx *ctx, const uint64_t *a, int64_t x0, int64_t x1, uint64_t *out);
void sgo_flatten_draws(const sgo_ctx *ctx, const uint8_t *key32, unsigned cc, uint32_t y, uint32_t boot,
                       uint32_t call, int64_t *draws);
int sgo_bootstrap_batch_rnd(const sgo_ctx *ctx, int opt, const uint64_t *key, const uint64_t *a1,
                            const uint64_t *b1, const uint64_t *a2, const uint64_t *b2, size_t batch,
                            uint64_t *out, int raw, uint64_t n_iters, uint64_t *acc_out, int threads,
                            const uint8_t *key32, uint32_t call, uint32_t boot0, const uint32_t *boots);

/* fhe.jl:660-696 pack_encrypted_bits(bkey, nothing, enc_bits): a [n][n], b [n] over Z_r ->
 * RLWE (w, v), [m] words in [0, r) each. */
int sgo_pack_encrypted_bits(const sgo_ctx *ctx, const uint64_t *bkey, const uint64_t *a,
                            const uint64_t *b, uint64_t *w, uint64_t *v, int threads);
/* The same with the bootstraps in the NTT-domain algebra when khat (sgo_key_transform) is given -- bkey is
 * needed either way for the half-width products -- and, with key32 != NULL, pack_encrypted_bits(bkey, rng,
 * enc_bits) on the engine's ChaCha8 stream: ciphertext `ct` of call `call`; bootstrap j of the group draws as
 * bootstrap ct n + j of the call (fhe.jl:673), the flatten of as_i with tag 2^31 | i as "bootstrap" ct
 * (fhe.jl:683-684; the layout oracle/bigint_oracle.py pack_encrypted_bits restates). */
int sgo_pack_encrypted_bits_ex(const sgo_ctx *ctx, const uint64_t *bkey, const uint64_t *khat, const uint64_t *a,
                               const uint64_t *b, uint64_t *w, uint64_t *v, int threads, const uint8_t *key32,
                               uint32_t ct, uint32_t call);

#ifdef __cplusplus
}
#endif
#endif
