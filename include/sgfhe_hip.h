/*
 * sgfhe_hip.h -- C ABI of libsgfhe_hip.so, the MI355X (gfx950) gate-bootstrap engine.
 *
 * Drop-in boundary for the hot path of nucypher/SGFHE.jl (paths relative to /root/reference):
 * the reference has no FFI of its own (pure Julia); each entry point
 * below names the Julia function or data structure whose work it takes over, and
 * INTEGRATION.md shows the `ccall` stub a maintainer adds on the Julia side.
 *
 * Conventions
 *   - every function returns an int32 status: 0 = OK, negative = sgfhe_status error
 *   - no exception crosses the boundary; sgfhe_last_error_string() explains the last failure
 *   - the caller owns every buffer it passes; the library owns device memory behind the handle
 *   - a ctx is bound to one device; any number of ctxs may share a device.  Every entry point
 *     that takes a ctx locks it for the duration of the call, so a ctx may be shared by host
 *     threads (calls on one ctx are serialised; calls on different ctxs run concurrently).
 *     That includes the asynchronous entry point sgfhe_bootstrap_batch_device: the work of
 *     successive calls on one ctx is ordered on the device in the order the calls were made,
 *     whatever stream each call names (each call's first kernel waits, on the device, for the
 *     previous call's last one: the ctx owns the work buffers every call uses), so two threads
 *     or two streams sharing a ctx get the bytes of the same calls made one after the other
 *     (the reference call is pure, src/fhe.jl:608-621).  Callers that want their calls to OVERLAP on
 *     the device -- Julia tasks each running bootstrap(bkey, ...) on one key -- take one clone of the
 *     ctx each (sgfhe_ctx_clone, ABI revision 7): clones share the device key and constants and own
 *     their work buffers and streams
 *   - residues mod Q cross the boundary as canonical representatives in [0, Q), little-endian
 *     `limbs` x uint64 each, limbs = 2 (16 bytes, the reference's UInt128 storage width) unless
 *     stated otherwise; LWE words over Z_r are one uint64 each, exactly the memory of
 *     `Vector{ModUInt{UInt64, r}}` (src/fhe.jl:206-209)
 */
#ifndef SGFHE_HIP_H
#define SGFHE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sgfhe_ctx sgfhe_ctx;

typedef enum {
    SGFHE_OK = 0,
    SGFHE_ERR_INVALID_ARG = -1,   /* NULL pointer, size mismatch, malformed parameter set */
    SGFHE_ERR_UNSUPPORTED = -2,   /* parameter set outside what the engine implements */
    SGFHE_ERR_NO_DEVICE = -3,     /* no usable gfx950 device / HIP runtime failure at start-up */
    SGFHE_ERR_HIP = -4,           /* a HIP runtime call failed (see last_error_string) */
    SGFHE_ERR_NO_KEY = -5,        /* bootstrap requested before a bootstrap key was uploaded */
    SGFHE_ERR_OOM = -6            /* host or device allocation failed */
} sgfhe_status;

/*
 * Scheme parameters: the fields of `Params` (src/fhe.jl:27-41) that the hot path reads.
 * Params(n) (src/fhe.jl:43-97) fixes r = 16 n, m = r / 2, ell = 2 (src/fhe.jl:576),
 * B = 35 r^2 n, DQ_tilde = Q / 8; synthetic sets (BASELINE.json configs 3 / 4) choose their own
 * Q and B with the same structure.  Q need not be prime: the engine computes the exact integer
 * negacyclic product in a residue number system of word-size NTT primes and reduces mod Q.
 */
typedef struct {
    uint64_t n;           /* LWE dimension (polynomial length of the small ring) */
    uint64_t r;           /* LWE modulus, power of two, r = 2 m */
    uint64_t m;           /* bootstrap polynomial length, power of two, 2^6 .. 2^14 */
    uint64_t ell;         /* gadget decomposition length; must be 2 */
    uint64_t Q[2];        /* bootstrap modulus, Q < 2^94 */
    uint64_t B[2];        /* gadget base, B^2 >= Q, B < 2^47 */
    uint64_t DQ_tilde[2]; /* Q / 8 for Params(n) */
} sgfhe_params;

/* flags of sgfhe_bootstrap_batch* */
#define SGFHE_FLAG_RAW_MODQ 1u /* return _bootstrap_internal's LWEs over Z_Q (fhe.jl:559-595) as
                                  16-byte residues instead of ModRed words (fhe.jl:616-618) */
#define SGFHE_FLAG_RAW_RNS2 2u /* with RAW_MODQ: every residue leaves as the RNS2Number limb pair
                                  (x mod m1, x mod m2) (src/rns.jl:16-18) of the moduli given to
                                  sgfhe_bkey_upload_rns2, instead of {lo, hi} of x */

/* Library / build information: "sgfhe_hip <version> gfx950". */
const char *sgfhe_version(void);
/* Revision of this header the library was built against.  A binding compares it with the
 * SGFHE_ABI_VERSION it was written for and refuses a stale library (julia/SGFHEHip.jl __init__,
 * sgfhe.jl_amd/_lib.py).  Bumped whenever a signature, a struct layout, a flag value or the
 * meaning of an argument changes. */
#define SGFHE_ABI_VERSION 7u
uint32_t sgfhe_abi_version(void);
/* Identity of the kernel sources the library was compiled from: the first 16 hex digits of the
 * SHA-256 over csrc/{*.h, *.hip} (in file-name order), followed by "+<flags>" when the build used
 * extra -D flags (timing-only ablation builds).  bench.py quotes profile counters only when they
 * were collected on a library with the same id. */
const char *sgfhe_build_id(void);

/* Create an engine for one parameter set on HIP device `device`.
 * Replaces: the type-level set-up Julia does when `Params(n)` fixes MgModUInt{LargeType, Q}
 * (src/fhe.jl:71-85,102-104) -- here: RNS primes, twiddle tables, CRT and flatten constants. */
int32_t sgfhe_ctx_create(const sgfhe_params *p, int device, sgfhe_ctx **out);
/* The same with creation flags.
 * Both flatten modes of the reference (`rng = nothing` and `rng::AbstractRNG` of bootstrap /
 * pack_encrypted_bits, src/utils.jl:155-189 and :198-241) are available on every ctx.  The digits
 * of the randomised one are four times larger, and at Params(1024) that takes a sixth 29-bit RNS
 * prime: such a ctx keeps a basis per mode (ABI revision 6) -- the key in both forms (1.34 + 1.61 GB
 * at Params(1024); the five-prime form is derived from the six-prime one on the device), constants
 * for both -- so the deterministic mode runs on five primes whatever the ctx may be asked later, and
 * sgfhe_set_random_flatten switches.  The other Params(n) need one basis for both.
 *   SGFHE_CTX_DETERMINISTIC_ONLY  the smaller basis only (no second key form): on such a ctx
 *       sgfhe_set_random_flatten(enable = 1) fails with SGFHE_ERR_UNSUPPORTED where the randomised
 *       mode would need the extra prime.  Its key blob is the smaller basis's.
 *   SGFHE_CTX_RANDOM_FLATTEN      accepted and without effect (up to ABI revision 5 it asked for the
 *       larger basis, which then also served the deterministic mode, about 20 % slower). */
#define SGFHE_CTX_RANDOM_FLATTEN 1u
#define SGFHE_CTX_DETERMINISTIC_ONLY 2u
int32_t sgfhe_ctx_create_ex(const sgfhe_params *p, int device, uint32_t flags, sgfhe_ctx **out);
/*
 * A second ctx on the same device, parameter set and KEY, for an independent caller (ABI revision 7).
 * The reference call is pure (src/fhe.jl:608-621): any number of Julia tasks may run
 * bootstrap(bkey, ...) on one BootstrapKey side by side.  Calls on ONE ctx are serialised and ordered
 * on the device, because the ctx owns the work buffers every call uses; a clone is what gives a second
 * caller its own: it SHARES with `ctx` everything a bootstrap only reads -- the device key in every
 * form (1.34 + 1.61 GB at Params(1024): not copied), twiddle tables, per-prime and CRT constants -- and
 * OWNS its lanes' work buffers (allocated by its first call, sized by its largest), streams, events,
 * host staging, flatten mode (deterministic to begin with; its own ChaCha key and call counter), error
 * string and lock.  The scheduling knobs (sgfhe_set_chunk / _lanes / _small_batch_max) are inherited as
 * they stand.  Calls on different clones are independent -- each gives the bytes the same call gives on
 * `ctx` -- and small host-pointer calls made at the same time are gathered into one launch chain
 * (sgfhe_set_coalesce below), which is how eight callers get 5.8 (Params(1024)) to 6.7 (Params(512)) times one
 * caller's rate.
 * `ctx` must hold a key (SGFHE_ERR_NO_KEY otherwise).  While a key is shared -- by `ctx` and at least
 * one clone, or by clones alone -- it is read-only: sgfhe_bkey_upload / _upload_rns2 / _generate /
 * _import_device_form on any of the sharers fail with SGFHE_ERR_INVALID_ARG (export is allowed).  The
 * shared memory is freed with the last ctx that holds it, so `ctx` and its clones may be destroyed in
 * any order; every one of them is destroyed with sgfhe_ctx_destroy.
 * As with sgfhe_ctx_create, *out is set even when the call fails after allocating it (read the error,
 * then destroy it).
 */
int32_t sgfhe_ctx_clone(sgfhe_ctx *ctx, sgfhe_ctx **out);
/*
 * Gathering of small calls across the ctxs that share a key (ABI revision 7).  Separate launch chains overlap only so
 * far on this device (four hardware queues: 1.8 x one caller's rate at two callers, 3 x at four and beyond), while ONE
 * chain of g gates costs little more than a chain of one (Params(1024): 15 ms for 1 gate, 21 ms for 8, 27 ms for 16).
 * So sgfhe_bootstrap_batch calls of at most `req_max` gates (default 32) made at the same time on ctxs that share a
 * key -- a ctx and its clones, each driven by its own host thread -- are run as ONE call: the caller that finds no combined call in flight takes
 * every request waiting (up to `gates_max` gates, default 256), runs them as one batch on its own ctx and hands each
 * caller its rows; callers arriving meanwhile form the next round, whose leader waits up to `window_us` (default 300)
 * for as many callers as the last two rounds had.  A caller on its own never waits, and a ctx without clones is not
 * affected at all.  A row of the result does not depend on the rows beside it, so every caller gets the bytes its
 * call gives alone -- with the randomised flatten too: every row of a gathered call draws from the stream of the ctx
 * it came in on (that ctx's key, the number of its call, the row's index in its call), and deterministic and
 * randomised requests form separate rounds.  The asynchronous entry point (sgfhe_bootstrap_batch_device) and
 * sgfhe_pack_encrypted_bits are never gathered.
 * The setting belongs to the shared key: it applies to every ctx that shares it.  enable = 0 switches gathering off
 * (SGFHE_COALESCE=0 in the environment does the same at ctx creation); the other arguments are then ignored.
 * sgfhe_coalesce_stats: stats[4] = {combined calls run, requests served, gates, most requests in one call}.
 */
int32_t sgfhe_set_coalesce(sgfhe_ctx *ctx, int enable, uint32_t req_max, uint32_t gates_max, uint32_t window_us);
int32_t sgfhe_coalesce_stats(sgfhe_ctx *ctx, uint64_t *stats, int reset);
int32_t sgfhe_ctx_destroy(sgfhe_ctx *ctx);
const char *sgfhe_last_error_string(const sgfhe_ctx *ctx);

/* Batch-scheduling knobs.  chunk: bootstraps that move through the k-loop in lock-step
 * (rounded up to a multiple of 8; a chunk's buffers must stay below 4 GiB).  0 = automatic: a size
 * near the Infinity Cache budget (256 per lane at Params(1024)), and with two lanes a batch above
 * 48 gates is cut into an even number of equal chunks, so that both lanes carry the same load.
 * lanes: 2 (default) runs pairs of chunks on two HIP streams, so that the memory-bound CRT kernel
 * of one chunk runs beside the arithmetic-bound external product of the other (+2 ... 5 % at
 * Params(1024), profiles/r03_exp_lanes_sweep.txt); 1 runs the chunks of a batch one after the
 * other, with twice the default chunk.  Every setting gives bit-identical results, in both
 * flatten modes. */
int32_t sgfhe_set_chunk(sgfhe_ctx *ctx, uint32_t chunk);
/* Chunks of at most this many bootstraps (default 24, 16 at m = 16384; 0 = never, at most 256) run the k-loop in
 * its small-batch form: 6 workgroups per (bootstrap, RNS prime) and three launches per iteration
 * instead of 1 and two, which shortens the serial chain a single bootstrap() call waits for.
 * Same results bit for bit. */
int32_t sgfhe_set_small_batch_max(sgfhe_ctx *ctx, uint32_t max_bootstraps);
int32_t sgfhe_set_lanes(sgfhe_ctx *ctx, uint32_t lanes);

/*
 * Flatten mode of the external product.  enable = 0 (default): deterministic flatten, the
 * `rng = nothing` branch (src/utils.jl:155-189), bit-exact with the reference.  enable = 1:
 * randomised flatten, the `rng::AbstractRNG` branch (src/utils.jl:198-241): every digit gets a
 * uniform v in [-3B/2, 3B/2] drawn from a ChaCha counter stream (the RFC 8439 block function with
 * 8 rounds, "ChaCha8") keyed with `key32`; digits lie in (-2B, 2B].  The draw of a coefficient is
 * addressed by (coefficient, iteration, index of the bootstrap in the call, number of the call
 * since this function): results do not depend on chunk size, lanes or the small-batch threshold,
 * and oracle/bigint_oracle.py reproduces them bit for bit.  They decrypt like the reference's
 * but are not bit-comparable with it (the stream of the caller's Julia rng cannot be reproduced:
 * a host draws the 32 key bytes from that rng instead, julia/SGFHEHip.jl).  Applies to later
 * bootstrap / pack calls, on every ctx not created with SGFHE_CTX_DETERMINISTIC_ONLY (see
 * sgfhe_ctx_create_ex; a ctx with a basis per mode switches to the other one, queued work is not
 * affected).  Every Params(n) the reference can build is covered, n = 64 ... 2048
 * (B up to 2^47: above 2^46 the stored digits take a third plane of the digit record).
 * sgfhe_set_random_flatten_key takes the full 32-byte key (the reference draws every v_i from the
 * caller's rng, src/utils.jl:229: with a key from a cryptographic source the perturbations are
 * cryptographically strong); sgfhe_set_random_flatten is the short form for tests and benchmarks,
 * key = `seed` as 32 little-endian bytes (64 bits of entropy).  (ABI revisions up to 4 drew from
 * Philox4x32-10 keyed by 64 bits.)
 */
int32_t sgfhe_set_random_flatten(sgfhe_ctx *ctx, int enable, uint64_t seed);
int32_t sgfhe_set_random_flatten_key(sgfhe_ctx *ctx, int enable, const uint8_t *key32);

/*
 * Upload a bootstrap key.  `canonical` (host memory) holds value.(p.coeffs) of
 * `BootstrapKey.key` (src/fhe.jl:176-201) in index order [k in 0..n)[row in 0..4)[col in 0..2)
 * [coef in 0..m), each residue 2 x uint64 little-endian; n_words = n * 8 * m * 2.
 * One-time: converts to the device form (per-prime forward NTT, Montgomery-scaled).
 */
int32_t sgfhe_bkey_upload(sgfhe_ctx *ctx, const uint64_t *canonical, size_t n_words);

/*
 * Same for a key held as RNS2Number{UInt64, M1, M2} pairs (src/rns.jl:8-24; type_Q of
 * Scheme2, src/fhe2.jl:76): each coefficient is (v1, v2) = (x mod m1, x mod m2).  The boundary
 * conversion is the CRT of src/rns.jl:32-40, done on the device; requires m1 * m2 == Q, both
 * prime and below 2^47.  A limb outside [0, m_i) is SGFHE_ERR_INVALID_ARG.
 */
int32_t sgfhe_bkey_upload_rns2(sgfhe_ctx *ctx, const uint64_t *pairs, size_t n_words, uint64_t m1,
                               uint64_t m2);
/* The two conversions of src/rns.jl on `count` coefficients (host pointers, run on the device):
 * to_pairs = 0: (v1, v2) -> canonical x = (v1 c1 + v2 c2) mod (m1 m2) (rns.jl:32-40);
 * to_pairs = 1: canonical x -> (x mod m1, x mod m2) (rns.jl:16-18).  m1 m2 must equal Q. */
int32_t sgfhe_rns2_convert(sgfhe_ctx *ctx, int to_pairs, const uint64_t *in, size_t count,
                           uint64_t m1, uint64_t m2, uint64_t *out);

/*
 * BootstrapKey(rng, sk) (src/fhe.jl:181-201) generated on the device, directly in device form:
 * for every k and gadget row a uniform a_row in [0, Q)^m, noise e_row in [-noise, noise]^m
 * (the reference uses noise = n, src/fhe.jl:194), b_row = a_row * s + e_row, plus s_k G on the
 * constant terms.  sk: n words, bit 0 of each is the key bit (PrivateKey.key, src/fhe.jl:130-138).
 * Randomness: ChaCha20 (RFC 8439) keyed with the 32-byte `seed`, separate counter-addressed
 * streams for the uniform polynomials and for the noise of every key row (layout in
 * csrc/kernels.h).  The key's entropy is the seed's: pass 32 bytes from a cryptographic generator.
 */
int32_t sgfhe_bkey_generate(sgfhe_ctx *ctx, const uint64_t *sk, size_t n_sk, const uint8_t *seed,
                            uint32_t noise);

/* Device-form key blob (for the one-time RCCL broadcast rank 0 -> peers). */
int32_t sgfhe_bkey_device_form_bytes(const sgfhe_ctx *ctx, size_t *bytes);
int32_t sgfhe_bkey_export_device_form(sgfhe_ctx *ctx, void *dst_device);
int32_t sgfhe_bkey_import_device_form(sgfhe_ctx *ctx, const void *src_device);

/*
 * bootstrap(bkey, nothing, enc_bit1, enc_bit2) (src/fhe.jl:608-621) over a batch.
 *   a1, a2 : [batch][n] uint64 in [0, r)   (EncryptedBit.lwe.a, src/fhe.jl:206-209,272-274)
 *   b1, b2 : [batch]    uint64 in [0, r)   (EncryptedBit.lwe.b)
 *   out    : [batch][3][n + 1] uint64: a[0..n) then b; gate order AND, OR, XOR
 *            (with SGFHE_FLAG_RAW_MODQ: [batch][3][n + 1][2], residues mod Q)
 * Deterministic flatten (rng = nothing, src/utils.jl:155-189) unless sgfhe_set_random_flatten[_key]
 * selected the other.  Host pointers; synchronous.  The arrays travel chunk by chunk through
 * page-locked staging buffers the ctx keeps (up to 1 GiB each; sgfhe_release_host_staging frees
 * them): a chunk's inputs go up while the chunks before it compute and its results come down while
 * the chunks after it compute, so a large batch runs at the rate of device-resident buffers
 * (larger arrays, and SGFHE_HOST_PIN=0 in the environment, are copied directly before and after).
 * A caller in a loop should keep its `out` buffer: releasing a multi-megabyte
 * array between calls (munmap) can stall the next call's kernels by tens of milliseconds.
 * SGFHE_DEBUG_IO=1 in the environment prints the phases of every call to stderr.
 */
int32_t sgfhe_bootstrap_batch(sgfhe_ctx *ctx, const uint64_t *a1, const uint64_t *b1,
                              const uint64_t *a2, const uint64_t *b2, size_t batch, uint64_t *out,
                              uint32_t flags);

/* Same with every buffer resident in device memory; asynchronous.  `stream` (a hipStream_t) is the
 * stream the call's work is queued on, NULL = the ctx's own: `out` is complete when that stream
 * reaches the end of the call's work (ordinary stream order for whatever the caller queues next on
 * it), and the inputs must stay valid until then.  Calls on one ctx do not overlap on the device:
 * a call's work starts after the work of every earlier call on this ctx, on any stream, has
 * finished (see the conventions above).  sgfhe_sync waits on the host for all work queued on the
 * ctx. */
int32_t sgfhe_bootstrap_batch_device(sgfhe_ctx *ctx, const uint64_t *a1, const uint64_t *b1,
                                     const uint64_t *a2, const uint64_t *b2, size_t batch,
                                     uint64_t *out, uint32_t flags, void *stream);
int32_t sgfhe_sync(sgfhe_ctx *ctx);
/* Frees the staging buffers sgfhe_bootstrap_batch keeps on the ctx (device and page-locked host
 * memory, sized by the largest batch seen); the next call allocates them again. */
int32_t sgfhe_release_host_staging(sgfhe_ctx *ctx);

/*
 * external_product(nothing, a, b, A, Val(B), Val(2)) (src/fhe.jl:519-530), the operation
 * test/internals.test.jl:144-166 checks.  a, b: [m][2]; A: [4][2][m][2] (row-major A[row][col]);
 * a_res, b_res: [m][2].  Host pointers; synchronous.  Parity / debug hook.
 */
int32_t sgfhe_external_product(sgfhe_ctx *ctx, const uint64_t *a, const uint64_t *b,
                               const uint64_t *A, uint64_t *a_res, uint64_t *b_res);

/*
 * One k-loop iteration on caller-chosen operands, exactly as the hot path runs it
 * (src/fhe.jl:580-581): (a, b) <- external_product(nothing, a, b, (x^j - 1) C .+ G, Val(B), Val(2))
 * = (a, b) + (x^j - 1) sum_row flatten(a, b)_row (*) C[row], through k_flatten_canon -> k_extprod
 * (with the rotation) -> k_crt_acc.  a, b: [m][2]; C: [4][2][m][2] canonical residues (one key
 * slice); j in [0, 2 m).  Host pointers; synchronous.  Parity / debug hook: lets a test drive the
 * exact-integer CRT to the bound it is sized for (digits +-B/2, key residues +-Q/2, j = m).
 */
int32_t sgfhe_debug_cmux(sgfhe_ctx *ctx, const uint64_t *a, const uint64_t *b, const uint64_t *C,
                         uint64_t j, uint64_t *a_res, uint64_t *b_res);

/*
 * pack_encrypted_bits(bkey, nothing, enc_bits) (src/fhe.jl:660-696, with
 * shortened_external_product :632-641): `count` groups of n LWEs each -> `count` RLWE
 * ciphertexts over Z_r.
 *   a : [count][n][n] uint64, b : [count][n] uint64   (the n EncryptedBits of every group)
 *   out_w, out_v : [count][m] uint64 in [0, r)         (Ciphertext.rlwe.a / .b coefficients)
 * Runs count * n gate bootstraps (trivial encryption of 1 paired with every bit, AND branch,
 * un-reduced), then the n half-width external products against the key on the device.
 * Flatten mode: the one sgfhe_set_random_flatten selected for this ctx, for the gate bootstraps
 * and for the flatten of every as_i alike (the reference passes the same `rng` to both,
 * src/fhe.jl:673,683-684): deterministic by default.  Host pointers; synchronous.
 */
int32_t sgfhe_pack_encrypted_bits(sgfhe_ctx *ctx, const uint64_t *a, const uint64_t *b,
                                  size_t count, uint64_t *out_w, uint64_t *out_v);

/* Parity / debug hook: run the first n_iters iterations of the k-loop (src/fhe.jl:579-582) and
 * return the accumulator pair (a, b) as canonical residues, acc: [batch][2][m][2]. */
int32_t sgfhe_debug_accumulators(sgfhe_ctx *ctx, const uint64_t *a1, const uint64_t *b1,
                                 const uint64_t *a2, const uint64_t *b2, size_t batch,
                                 uint64_t n_iters, uint64_t *acc);

/* Parity / debug hook: the stored digit planes after n_iters iterations, i.e. the flatten result
 * (src/utils.jl:155-241) the next external product will consume: digits [batch][2][2][m] uint64,
 * index order (accumulator c: 0 = a, 1 = b)(digit i)(coefficient).  Stored value e_i = u_i + s
 * with s = B/2 - 1 (even B) or (B - 1)/2 and u_i the reference's i-th flatten output taken as a
 * signed integer, u_i in (-B/2, B/2]; in the randomised mode e_i = u_i + s + xmax,
 * xmax = 3 (B / 2), u_i in (-2B, 2B] (test/internals.test.jl:50-66).  sum_i u_i B^i == acc mod Q. */
int32_t sgfhe_debug_digits(sgfhe_ctx *ctx, const uint64_t *a1, const uint64_t *b1,
                           const uint64_t *a2, const uint64_t *b2, size_t batch, uint64_t n_iters,
                           uint64_t *digits);

/* Parity / debug hook: flatten_poly(nothing, ., Val(B), Val(2)) (src/utils.jl:155-189,253-264) of
 * two polynomials on its own (k_flatten_canon): values [2][m][2] canonical residues ->
 * digits [2][2][m] uint64 in the stored form of sgfhe_debug_digits (e_i = u_i + s).  Host pointers. */
int32_t sgfhe_debug_flatten(sgfhe_ctx *ctx, const uint64_t *values, uint64_t *digits);

/* Parity / debug hook: negacyclic NTT of one polynomial modulo RNS prime `prime_index`.
 * in/out: [m] uint32 residues; forward maps natural order to the engine's slot order,
 * inverse maps back (and divides by m).  Host pointers. */
int32_t sgfhe_debug_ntt(sgfhe_ctx *ctx, uint32_t prime_index, int inverse, const uint32_t *in,
                        uint32_t *out);
/* Number of RNS primes and their values (primes[] must hold 8 entries). */
int32_t sgfhe_debug_primes(const sgfhe_ctx *ctx, uint32_t *count, uint32_t *primes);

/*
 * The ciphertext plumbing either side of the path, on the host: no
 * device, no ctx; `p` supplies n, r = 2 m (t = log2 r - 1, Dr = r / 4).  Polynomials over Z_r
 * are arrays of uint64 in [0, r) (the memory of Polynomial{ModUInt{UInt64, r}}), bit arrays are
 * one uint8 per bit, bit matrices [rows][n] row-major.  Every function is pure: the random draws
 * of the reference (src/fhe.jl:315,319) are arguments, so the caller keeps its own generator.
 */
/* deterministic_expand(params, u) (src/fhe.jl:304-307): u[n] seed bits -> a[n] over Z_r.
 * prng_expand (src/utils.jl:63-68) is built on SHAKE-256 here, the primitive the reference names
 * as intended (utils.jl:64); the reference itself seeds a MersenneTwister with hash(seq). */
int32_t sgfhe_host_deterministic_expand(const sgfhe_params *p, const uint8_t *u, uint64_t *a);
/* _encrypt_private(key, rng, message) (src/fhe.jl:310-328) given its draws u[n] (bits) and
 * w[n] in [-Dr/8, Dr/8]: sk[n] key bits (bit 0 of each word), message[n] bits -> RLWE (a, b). */
int32_t sgfhe_host_encrypt_private(const sgfhe_params *p, const uint64_t *sk, const uint8_t *u,
                                   const int64_t *w, const uint8_t *message, uint64_t *a, uint64_t *b);
/* The v of encrypt_optimal(key::PrivateKey, ...) (src/fhe.jl:339-345): b[n] -> v[5][n] bits. */
int32_t sgfhe_host_pack_private(const sgfhe_params *p, const uint64_t *b, uint8_t *v);
/* normalize_ciphertext(::PrivateEncryptedCiphertext) (src/fhe.jl:354-359): (u[n], v[5][n]) -> (a, b). */
int32_t sgfhe_host_normalize_private(const sgfhe_params *p, const uint8_t *u, const uint8_t *v,
                                     uint64_t *a, uint64_t *b);
/* split_ciphertext(ct) (src/fhe.jl:287-290, extract :237-244) of an RLWE with polynomials of
 * length N = n (PackedCiphertext) or N = m (Ciphertext): lwe_a[n][n], lwe_b[n] -- the inputs of
 * sgfhe_bootstrap_batch. */
int32_t sgfhe_host_split_ciphertext(const sgfhe_params *p, const uint64_t *a, const uint64_t *b,
                                    size_t N, uint64_t *lwe_a, uint64_t *lwe_b);
/* decrypt(key, ::EncryptedBit) (src/fhe.jl:504-507) for `count` LWEs: lwe_a[count][n], lwe_b[count]
 * (e.g. one gate of sgfhe_bootstrap_batch's output) -> bits[count]. */
int32_t sgfhe_host_decrypt_lwe(const sgfhe_params *p, const uint64_t *sk, const uint64_t *lwe_a,
                               const uint64_t *lwe_b, size_t count, uint8_t *bits);
/* decrypt(key, ::Union{Ciphertext, PackedCiphertext}) (src/fhe.jl:471-494), N = n or m -> bits[n]. */
int32_t sgfhe_host_decrypt_rlwe(const sgfhe_params *p, const uint64_t *sk, const uint64_t *a,
                                const uint64_t *b, size_t N, uint8_t *bits);

/*
 * The public-key side: PublicKey, _encrypt_public and the
 * space-optimal public ciphertext, on the host.  q is the public-key modulus of Params(n)
 * (find_modulus(2 n, r n), src/fhe.jl:57; q < 2^31), Dq = q / 4 (src/fhe.jl:88).  Polynomials over
 * Z_q are uint64 in [0, q).  Pure functions: the caller passes the reference's draws.
 */
/* PublicKey(rng, sk) (src/fhe.jl:146-168) given its draws k0[n] in [0, q) and the centred noise
 * e[n] in [-e_max, e_max] (e_max the largest integer below Dq / (41 n), :159-160): k1 = k0 s + e.
 * The reference writes `polynomial - e_max` (:161); the noise is centred on every coefficient
 * here (DarkIntegers' Polynomial - scalar rule is not checkable in this build; decryption holds
 * either way). */
int32_t sgfhe_host_public_key(const sgfhe_params *p, uint64_t q, const uint64_t *sk, const uint64_t *k0,
                              const int64_t *e, uint64_t *k1);
/* _encrypt_public(key, rng, message) (src/fhe.jl:386-409) given its draws u[n] in {-1, 0, 1},
 * w1[n] in [-Dq / (41 n), Dq / (41 n)] and w2[n] in [-Dq / 82, Dq / 82]: -> RLWE (a, b) over Z_r,
 * b a multiple of 2^(t - 5). */
int32_t sgfhe_host_encrypt_public(const sgfhe_params *p, uint64_t q, const uint64_t *k0, const uint64_t *k1,
                                  const int8_t *u, const int64_t *w1, const int64_t *w2,
                                  const uint8_t *message, uint64_t *a, uint64_t *b);
/* The bit matrices of encrypt_optimal(key::PublicKey, ...) (src/fhe.jl:420-436):
 * (a[n], b[n]) -> a_bits[t + 1][n], b_bits[6][n]. */
int32_t sgfhe_host_pack_public(const sgfhe_params *p, const uint64_t *a, const uint64_t *b,
                               uint8_t *a_bits, uint8_t *b_bits);
/* normalize_ciphertext(::PublicEncryptedCiphertext) (src/fhe.jl:444-449). */
int32_t sgfhe_host_normalize_public(const sgfhe_params *p, const uint8_t *a_bits, const uint8_t *b_bits,
                                    uint64_t *a, uint64_t *b);

/*
 * Measurement hook for bench.py: HIP-event timings taken on the ctx stream around sampled
 * launches of the two per-iteration kernels since the last reset (stats must hold 8 doubles).
 *   stats[0] = average external-product kernel time (ms)   stats[1] = its sampled launches
 *   stats[2] = average CRT/accumulate kernel time (ms)     stats[3] = its sampled launches
 *   stats[4] = bootstraps per external-product launch (chunk actually used)
 *   stats[5] = average device time of a whole sgfhe_bootstrap_batch_device call (ms), from its
 *              first to its last kernel on both lanes     stats[6] = calls   stats[7] = batch
 * With two lanes the kernels of the lanes overlap: stats[0] and [2] are then durations under
 * co-execution and do not add up to an iteration; stats[5] is the wall time.
 */
int32_t sgfhe_timing_enable(sgfhe_ctx *ctx, int enable);
int32_t sgfhe_timing_read(sgfhe_ctx *ctx, double *stats, int reset);
/* Names of the two k-loop kernels this ctx launches in its present flatten mode, as they appear in
 * a rocprofv3 kernel trace ("k_extprod<13, 4, false>", "k_crt_lean<5, 3>"; parameter sets outside
 * k_crt_lean's bounds and SGFHE_CRT_LEAN=0 give k_crt_acc2 / k_crt_acc), NUL-terminated into the
 * caller's buffers (64 bytes suffice). */
int32_t sgfhe_kernel_names(const sgfhe_ctx *ctx, char *extprod, size_t extprod_cap, char *crt,
                           size_t crt_cap);

#ifdef __cplusplus
}
#endif
#endif
