/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  See sgfhe_oracle.h for scope and parity status.
 *
 * Reference-shaped C restatement of SGFHE.jl's deterministic bootstrap.  Each function cites
 * the reference file:line it follows (paths relative to /root/reference).
 */
#include "sgfhe_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

struct sgo_ctx {
    uint64_t n, r, m, ell, Dr;
    u128 Q, B, DQ_tilde;
    /* Montgomery, R = 2^128 (DarkIntegers MgModUInt{UInt128, Q}) */
    u128 ninv, R1, R2;
    int use_ntt;
    unsigned logm;
    u128 *psi_br;   /* psi^bitrev(i) * R mod Q */
    u128 *ipsi_br;  /* psi^-bitrev(i) * R mod Q */
    u128 minv_R2;   /* m^-1 * R^2 mod Q */
    /* flatten constants, utils.jl:162-169 */
    u128 fl_s, fl_offset;
    /* RNS2Number ring (src/rns.jl), Q = m1 m2: one word-size NTT context per limb modulus and the
     * CRT idempotents c1 = m2^(m1-1), c2 = m1^(m2-1) mod Q of rns.jl:36-37 */
    struct sgo_ctx *limb[2];
    u128 rns_m[2], rns_c[2];
};

static inline u128 ld128(const uint64_t *p) { return ((u128)p[1] << 64) | p[0]; }
static inline void st128(uint64_t *p, u128 x) { p[0] = (uint64_t)x; p[1] = (uint64_t)(x >> 64); }

/* ---------------------------------------------------------------- 128-bit modular arithmetic */

static inline void mul128(u128 a, u128 b, u128 *hi, u128 *lo) {
    uint64_t a0 = (uint64_t)a, a1 = (uint64_t)(a >> 64);
    uint64_t b0 = (uint64_t)b, b1 = (uint64_t)(b >> 64);
    u128 p00 = (u128)a0 * b0, p01 = (u128)a0 * b1, p10 = (u128)a1 * b0, p11 = (u128)a1 * b1;
    u128 mid = (p00 >> 64) + (uint64_t)p01 + (uint64_t)p10;
    *lo = (u128)(uint64_t)p00 | (mid << 64);
    *hi = p11 + (p01 >> 64) + (p10 >> 64) + (mid >> 64);
}

typedef struct { u128 Q, ninv, R1, R2; } mont_t;

static void mont_init(mont_t *mt, u128 Q) {
    /* Q odd, Q < 2^126 */
    u128 inv = Q; /* correct to 3 bits */
    for (int i = 0; i < 7; i++) inv *= 2 - Q * inv;
    mt->Q = Q;
    mt->ninv = (u128)0 - inv;
    u128 r1 = ((u128)0 - 1) % Q;
    r1 = (r1 + 1) % Q;
    mt->R1 = r1;
    u128 r2 = r1;
    for (int i = 0; i < 128; i++) { r2 <<= 1; if (r2 >= Q) r2 -= Q; }
    mt->R2 = r2;
}

static inline u128 mont_mul(const mont_t *mt, u128 a, u128 b) {
    u128 th, tl, uh, ul;
    mul128(a, b, &th, &tl);
    u128 mq = tl * mt->ninv;
    mul128(mq, mt->Q, &uh, &ul);
    u128 r = th + uh + (tl != 0);
    if (r >= mt->Q) r -= mt->Q;
    return r;
}

static inline u128 addmod(u128 a, u128 b, u128 Q) { u128 s = a + b; if (s >= Q) s -= Q; return s; }
static inline u128 submod(u128 a, u128 b, u128 Q) { return a >= b ? a - b : a + Q - b; }

static u128 mulmod_plain(const mont_t *mt, u128 a, u128 b) {
    return mont_mul(mt, mont_mul(mt, a, mt->R2), b);
}

static u128 powmod(const mont_t *mt, u128 base, u128 e) {
    u128 acc = mt->R1; /* 1 in Montgomery form */
    u128 b = mont_mul(mt, base % mt->Q, mt->R2);
    while (e) {
        if (e & 1) acc = mont_mul(mt, acc, b);
        b = mont_mul(mt, b, b);
        e >>= 1;
    }
    return mont_mul(mt, acc, 1);
}

/* Strong-probable-prime test to 40 fixed bases: stands in for Primes.isprime (utils.jl:19). */
static int is_prime_u128(u128 x) {
    static const unsigned bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53,
        59, 61, 67, 71, 73, 79, 83, 89, 97, 101, 103, 107, 109, 113, 127, 131, 137, 139, 149,
        151, 157, 163, 167, 173};
    if (x < 2) return 0;
    for (unsigned i = 0; i < 12; i++) {
        if (x % bases[i] == 0) return x == bases[i];
    }
    mont_t mt;
    mont_init(&mt, x);
    u128 d = x - 1;
    int s = 0;
    while ((d & 1) == 0) { d >>= 1; s++; }
    for (unsigned i = 0; i < sizeof(bases) / sizeof(bases[0]); i++) {
        u128 a = bases[i];
        if (a % x == 0) continue;
        u128 y = powmod(&mt, a, d);
        if (y == 1 || y == x - 1) continue;
        int comp = 1;
        for (int k = 0; k < s - 1; k++) {
            y = mulmod_plain(&mt, y, y);
            if (y == x - 1) { comp = 0; break; }
        }
        if (comp) return 0;
    }
    return 1;
}

/* utils.jl:7-28 */
static int find_modulus_u128(u128 n, u128 qmin, u128 qmax, int has_max, u128 *out) {
    u128 j = (qmin - 1 + n - 1) / n; /* cld(qmin - 1, n) */
    for (;;) {
        u128 q = j * n + 1;
        if (has_max && q > qmax) return -1;
        if (is_prime_u128(q)) { *out = q; return 0; }
        j++;
    }
}

int sgo_find_modulus(uint64_t n, const uint64_t *qmin, const uint64_t *qmax, uint64_t *out) {
    u128 q;
    u128 mx = ld128(qmax);
    int rc = find_modulus_u128(n, ld128(qmin), mx, mx != 0, &q);
    if (rc == 0) st128(out, q);
    return rc;
}

/* fhe.jl:43-97 */
int sgo_params_make(uint64_t n, uint64_t *words) {
    if (n < 64 || (n & (n - 1))) return -1;                 /* fhe.jl:45-46 */
    u128 r = (u128)n * 16;                                   /* fhe.jl:53 */
    u128 m = r / 2;                                          /* fhe.jl:62 */
    u128 r4n2 = r * r * r * r * n * n;
    u128 Q;
    if (find_modulus_u128(2 * m, r4n2 * 1220, r4n2 * 1225, 1, &Q)) return -2;   /* fhe.jl:64-69 */
    u128 B = r * r * n * 35;                                 /* fhe.jl:87 */
    words[0] = n; words[1] = (uint64_t)r; words[2] = (uint64_t)m; words[3] = 2;
    st128(words + 4, Q); st128(words + 6, B); st128(words + 8, Q / 8);          /* fhe.jl:90 */
    return 0;
}

/* ---------------------------------------------------------------- ctx */

static unsigned bitrev(unsigned x, unsigned bits) {
    unsigned r = 0;
    for (unsigned i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

sgo_ctx *sgo_ctx_create(const uint64_t *w) {
    sgo_ctx *c = (sgo_ctx *)calloc(1, sizeof(*c));
    c->n = w[0]; c->r = w[1]; c->m = w[2]; c->ell = w[3];
    c->Q = ld128(w + 4); c->B = ld128(w + 6); c->DQ_tilde = ld128(w + 8);
    c->Dr = c->r / 4;                                        /* fhe.jl:88 */
    if (!(c->Q & 1) || c->ell != 2 || (c->m & (c->m - 1))) { free(c); return NULL; }
    mont_t mt;
    mont_init(&mt, c->Q);
    c->ninv = mt.ninv; c->R1 = mt.R1; c->R2 = mt.R2;
    /* utils.jl:162-169 */
    c->fl_s = (c->B & 1) ? (c->B - 1) / 2 : c->B / 2 - 1;
    c->fl_offset = mulmod_plain(&mt, (1 + c->B) % c->Q, c->fl_s % c->Q);
    while (((u128)1 << c->logm) < c->m) c->logm++;
    u128 two_m = 2 * (u128)c->m;
    if ((c->Q - 1) % two_m == 0 && is_prime_u128(c->Q)) {
        /* primitive 2m-th root of unity: g^m == -1 */
        u128 psi = 0;
        for (u128 x = 2; x < 1000; x++) {
            u128 g = powmod(&mt, x, (c->Q - 1) / two_m);
            if (powmod(&mt, g, c->m) == c->Q - 1) { psi = g; break; }
        }
        if (psi) {
            c->use_ntt = 1;
            u128 ipsi = powmod(&mt, psi, c->Q - 2);
            c->psi_br = (u128 *)malloc(sizeof(u128) * c->m);
            c->ipsi_br = (u128 *)malloc(sizeof(u128) * c->m);
            u128 p = mt.R1, ip = mt.R1;
            u128 psiM = mont_mul(&mt, psi, mt.R2), ipsiM = mont_mul(&mt, ipsi, mt.R2);
            for (uint64_t i = 0; i < c->m; i++) {
                unsigned br = bitrev((unsigned)i, c->logm);
                c->psi_br[br] = p;
                c->ipsi_br[br] = ip;
                p = mont_mul(&mt, p, psiM);
                ip = mont_mul(&mt, ip, ipsiM);
            }
            u128 minv = powmod(&mt, c->m, c->Q - 2);
            c->minv_R2 = mont_mul(&mt, mont_mul(&mt, minv, mt.R2), mt.R2);
        }
    }
    return c;
}

void sgo_ctx_destroy(sgo_ctx *c) {
    if (!c) return;
    sgo_ctx_destroy(c->limb[0]);
    sgo_ctx_destroy(c->limb[1]);
    free(c->psi_br); free(c->ipsi_br); free(c);
}

/* src/rns.jl:8-24: hold the coefficients of Z_Q, Q = m1 m2, as RNS2Number limb pairs.  Both limb
 * moduli must be NTT-friendly primes (2m | m_i - 1: the rule of src/fhe2.jl:57-58). */
int sgo_ctx_set_rns2(sgo_ctx *c, uint64_t m1, uint64_t m2) {
    if ((u128)m1 * m2 != c->Q || m1 == m2) return -1;
    uint64_t w[10] = {c->n, c->r, c->m, 2, 0, 0, 2, 0, 0, 0};
    const uint64_t ms[2] = {m1, m2};
    for (int i = 0; i < 2; i++) {
        w[4] = ms[i];
        sgo_ctx_destroy(c->limb[i]);
        c->limb[i] = sgo_ctx_create(w);
        if (!c->limb[i] || !c->limb[i]->use_ntt) return -2;
        c->rns_m[i] = ms[i];
    }
    mont_t mt;
    mont_init(&mt, c->Q);
    c->rns_c[0] = powmod(&mt, m2, (u128)m1 - 1);            /* rns.jl:36 */
    c->rns_c[1] = powmod(&mt, m1, (u128)m2 - 1);            /* rns.jl:37 */
    return 0;
}
int sgo_ctx_uses_rns2(const sgo_ctx *c) { return c->limb[0] != NULL; }

int sgo_ctx_uses_ntt(const sgo_ctx *c) { return c->use_ntt; }

static inline mont_t ctx_mont(const sgo_ctx *c) {
    mont_t mt = {c->Q, c->ninv, c->R1, c->R2};
    return mt;
}

/* ---------------------------------------------------------------- polynomial arithmetic */

/* forward negacyclic NTT (Cooley-Tukey, merged psi twist), natural -> bit-reversed */
static void ntt_fwd(const sgo_ctx *c, u128 *a) {
    mont_t mt = ctx_mont(c);
    u128 Q = c->Q;
    size_t t = c->m;
    for (size_t mm = 1; mm < c->m; mm <<= 1) {
        t >>= 1;
        for (size_t i = 0; i < mm; i++) {
            u128 W = c->psi_br[mm + i];
            size_t j1 = 2 * i * t;
            for (size_t j = j1; j < j1 + t; j++) {
                u128 U = a[j], V = mont_mul(&mt, a[j + t], W);
                a[j] = addmod(U, V, Q);
                a[j + t] = submod(U, V, Q);
            }
        }
    }
}

/* inverse (Gentleman-Sande), bit-reversed -> natural, unscaled */
static void ntt_inv(const sgo_ctx *c, u128 *a) {
    mont_t mt = ctx_mont(c);
    u128 Q = c->Q;
    size_t t = 1;
    for (size_t mm = c->m; mm > 1; mm >>= 1) {
        size_t h = mm >> 1, j1 = 0;
        for (size_t i = 0; i < h; i++) {
            u128 W = c->ipsi_br[h + i];
            for (size_t j = j1; j < j1 + t; j++) {
                u128 U = a[j], V = a[j + t];
                a[j] = addmod(U, V, Q);
                a[j + t] = mont_mul(&mt, submod(U, V, Q), W);
            }
            j1 += 2 * t;
        }
        t <<= 1;
    }
}

static void poly_mul_schoolbook(const sgo_ctx *c, const u128 *a, const u128 *b, u128 *out) {
    mont_t mt = ctx_mont(c);
    size_t m = c->m;
    u128 Q = c->Q;
    u128 *acc = (u128 *)calloc(m, sizeof(u128));
    for (size_t i = 0; i < m; i++) {
        if (a[i] == 0) continue;
        u128 aM = mont_mul(&mt, a[i], mt.R2);
        for (size_t j = 0; j < m; j++) {
            u128 p = mont_mul(&mt, aM, b[j]);
            size_t k = i + j;
            if (k < m) acc[k] = addmod(acc[k], p, Q);
            else acc[k - m] = submod(acc[k - m], p, Q);
        }
    }
    memcpy(out, acc, m * sizeof(u128));
    free(acc);
}

/* DarkIntegers `Polynomial * Polynomial` (call sites fhe.jl:195,527-528): exact product mod
 * (x^m + 1, Q); [DI-recall] NTT path (2 forward + pointwise + 1 inverse) when Q is prime with
 * 2m | Q - 1, else a non-NTT exact algorithm. */
static void poly_mul(const sgo_ctx *c, const u128 *a, const u128 *b, u128 *out);

/* Polynomial * Polynomial over RNS2Number{UInt64, m1, m2} coefficients: conversion from the
 * integer is (x mod m1, x mod m2) (rns.jl:16-18), `*`, `+`, `-` act limb-wise (rns.jl:51-60), so
 * the product is one NTT multiply per limb; conversion back is the CRT of rns.jl:32-40. */
static void poly_mul_rns2(const sgo_ctx *c, const u128 *a, const u128 *b, u128 *out) {
    size_t m = c->m;
    mont_t mt;
    mont_init(&mt, c->Q);
    u128 *la = (u128 *)malloc(4 * m * sizeof(u128));
    u128 *lb = la + m, *r1 = la + 2 * m, *r2 = la + 3 * m;
    for (int i = 0; i < 2; i++) {
        for (size_t j = 0; j < m; j++) { la[j] = a[j] % c->rns_m[i]; lb[j] = b[j] % c->rns_m[i]; }
        poly_mul(c->limb[i], la, lb, i ? r2 : r1);
    }
    for (size_t j = 0; j < m; j++)                            /* rns.jl:38 */
        out[j] = addmod(mulmod_plain(&mt, r1[j], c->rns_c[0]), mulmod_plain(&mt, r2[j], c->rns_c[1]), c->Q);
    free(la);
}

static void poly_mul(const sgo_ctx *c, const u128 *a, const u128 *b, u128 *out) {
    if (!c->use_ntt && c->limb[0]) { poly_mul_rns2(c, a, b, out); return; }
    if (!c->use_ntt) { poly_mul_schoolbook(c, a, b, out); return; }
    mont_t mt = ctx_mont(c);
    size_t m = c->m;
    u128 *fa = (u128 *)malloc(2 * m * sizeof(u128));
    u128 *fb = fa + m;
    memcpy(fa, a, m * sizeof(u128));
    memcpy(fb, b, m * sizeof(u128));
    ntt_fwd(c, fa);
    ntt_fwd(c, fb);
    for (size_t i = 0; i < m; i++) fa[i] = mont_mul(&mt, fa[i], fb[i]); /* x * R^-1 */
    ntt_inv(c, fa);
    for (size_t i = 0; i < m; i++) out[i] = mont_mul(&mt, fa[i], c->minv_R2);
    free(fa);
}

/* DarkIntegers mul_by_monomial(p, j), any integer j taken mod 2m (theory.md:23-32). */
static void mul_by_monomial(const sgo_ctx *c, const u128 *a, uint64_t j, u128 *out) {
    size_t m = c->m;
    j %= 2 * m;
    for (size_t i = 0; i < m; i++) {
        size_t k = i + j;
        int neg = (k / m) & 1;
        out[k % m] = neg ? (a[i] ? c->Q - a[i] : 0) : a[i];
    }
}

/* ---------------------------------------------------------------- rescale / flatten */

/* 256 / 128 -> quotient (assumed to fit 128 bits), remainder.  DarkIntegers divremhilo. */
static void divrem256(u128 hi, u128 lo, u128 d, u128 *q, u128 *r) {
    u128 rem = 0, quo = 0;
    for (int i = 255; i >= 0; i--) {
        int top = (int)(rem >> 127);
        rem = (rem << 1) | (i >= 128 ? (hi >> (i - 128)) & 1 : (lo >> i) & 1);
        quo <<= 1;
        if (top || rem >= d) { rem -= d; quo |= 1; }
    }
    *q = quo; *r = rem;
}

/* utils.jl:78-92 */
static u128 rescale(u128 new_max, u128 x, u128 old_max, int round_result) {
    u128 hi, lo, q, r;
    mul128(x, new_max, &hi, &lo);                            /* utils.jl:81 */
    divrem256(hi, lo, old_max, &q, &r);                      /* utils.jl:82 */
    if (round_result) {
        if (r >= old_max / 2 + (old_max & 1)) {              /* utils.jl:84 */
            q += 1;
            if (q == new_max) q = 0;                         /* utils.jl:86-88 */
        }
    }
    return q;
}

void sgo_rescale(const uint64_t *new_max, const uint64_t *x, const uint64_t *old_max,
                 int round_result, uint64_t *out) {
    st128(out, rescale(ld128(new_max), ld128(x), ld128(old_max), round_result));
}

/* utils.jl:155-189 with ell = 2 */
static inline void flatten2(const sgo_ctx *c, u128 a, u128 *d0, u128 *d1) {
    u128 Q = c->Q;
    a = addmod(a, c->fl_offset, Q);                          /* utils.jl:179 */
    u128 quot = a / c->B;                                    /* utils.jl:172: r = quotient */
    u128 rem = a - quot * c->B;
    *d1 = submod(quot % Q, c->fl_s % Q, Q);                  /* utils.jl:183-185 */
    *d0 = submod(rem % Q, c->fl_s % Q, Q);
}

void sgo_flatten(const sgo_ctx *c, const uint64_t *a, uint64_t *out) {
    u128 d0, d1;
    flatten2(c, ld128(a), &d0, &d1);
    st128(out, d0); st128(out + 2, d1);
}

/* ---- randomised flatten: flatten(rng::AbstractRNG, a, Val(B), Val(l)), utils.jl:198-241, ell = 2 ----
 * x_i = rand(rng, -xmax:xmax) (:229-231); rand_a = a - sum x_i B^(i-1) (:233-234);
 * y = flatten(nothing, rand_a) (:236); result x_i + y_i (:237-239), all in Z_Q.
 * The reference draws from the caller's Julia rng, whose stream cannot be reproduced outside Julia
 * (SURVEY.md F6).  The draws here are the HIP engine's: a ChaCha8 counter stream keyed with 32
 * bytes (sgfhe.jl_amd/csrc/kernels.h rnd128 / random_digits; oracle/bigint_oracle.py ChaChaFlatten
 * restates the same stream).  The 128 bits of coefficient x = (c << log2 m) + j (accumulator c:
 * 0 = a, 1 = b) in the flatten tagged y are words 4 (x mod 4) .. + 3 of the block whose state
 * words 12 .. 15 are (x div 4, y, index of the bootstrap within the call, number of the call);
 * r_0 = (lo64 span) >> 64, r_1 = (hi64 span) >> 64 with span = 2 xmax + 1; x_i = r_i - xmax. */
static void chacha_block(const uint32_t key[8], uint32_t w12, uint32_t w13, uint32_t w14, uint32_t w15,
                         int rounds, uint32_t out[16]);
#define SGO_RND_ROUNDS 8

typedef struct {
    uint32_t key[8];
    uint32_t call, boot;
} rnd_t;

/* utils.jl:210-214 */
static inline u128 flatten_xmax(const sgo_ctx *c) {
    return (c->B & 1) ? (c->B - 1) / 2 * 3 : c->B / 2 * 3;
}

/* convert(T, x) of a signed draw (utils.jl:230): x mod Q */
static inline u128 signed_to_mod(int64_t x, u128 Q) {
    return x >= 0 ? (u128)(uint64_t)x % Q : (Q - (u128)(uint64_t)(-x) % Q) % Q;
}

static inline void flatten2_rnd(const sgo_ctx *c, u128 a, int64_t x0, int64_t x1, u128 *d0, u128 *d1) {
    mont_t mt = ctx_mont(c);
    u128 Q = c->Q;
    u128 X0 = signed_to_mod(x0, Q), X1 = signed_to_mod(x1, Q);
    u128 rand_a = submod(a, X0, Q);                                      /* utils.jl:233-234, i = 1 */
    rand_a = submod(rand_a, mulmod_plain(&mt, X1, c->B % Q), Q);         /* i = 2: x[2] * B */
    u128 y0, y1;
    flatten2(c, rand_a, &y0, &y1);                                       /* utils.jl:236 */
    *d0 = addmod(X0, y0, Q);                                             /* utils.jl:237-239 */
    *d1 = addmod(X1, y1, Q);
}

/* the two draws of the coefficient with stream index x, out of its block */
static inline void draws_of(const sgo_ctx *c, const uint32_t blk[16], uint32_t x, int64_t *x0, int64_t *x1) {
    const uint32_t *w = blk + 4 * (x & 3);
    const u128 xmax = flatten_xmax(c), span = 2 * xmax + 1;
    const uint64_t lo = ((uint64_t)w[1] << 32) | w[0], hi = ((uint64_t)w[3] << 32) | w[2];
    *x0 = (int64_t)(uint64_t)(((u128)lo * span) >> 64) - (int64_t)(uint64_t)xmax;
    *x1 = (int64_t)(uint64_t)(((u128)hi * span) >> 64) - (int64_t)(uint64_t)xmax;
}

/* flatten_poly(rng, a, ...) (utils.jl:253-264): coefficient j draws its ell values in order */
static void flatten_poly2(const sgo_ctx *c, const rnd_t *g, unsigned cc, uint32_t y, const u128 *a,
                          u128 *d0, u128 *d1) {
    size_t m = c->m;
    if (!g) {
        for (size_t j = 0; j < m; j++) flatten2(c, a[j], &d0[j], &d1[j]);
        return;
    }
    uint32_t blk[16];
    for (size_t j = 0; j < m; j++) {
        const uint32_t x = ((uint32_t)cc << c->logm) + (uint32_t)j;
        if ((x & 3) == 0 || j == 0) chacha_block(g->key, x >> 2, y, g->boot, g->call, SGO_RND_ROUNDS, blk);
        int64_t x0, x1;
        draws_of(c, blk, x, &x0, &x1);
        flatten2_rnd(c, a[j], x0, x1, &d0[j], &d1[j]);
    }
}

void sgo_flatten_random(const sgo_ctx *c, const uint64_t *a, int64_t x0, int64_t x1, uint64_t *out) {
    u128 d0, d1;
    flatten2_rnd(c, ld128(a), x0, x1, &d0, &d1);
    st128(out, d0); st128(out + 2, d1);
}

/* draws[m][2] of one polynomial: accumulator cc (0 = a, 1 = b), flatten tag y, bootstrap `boot` of
 * call `call` of the stream keyed with key32 */
void sgo_flatten_draws(const sgo_ctx *c, const uint8_t *key32, unsigned cc, uint32_t y, uint32_t boot,
                       uint32_t call, int64_t *draws) {
    rnd_t g;
    for (int i = 0; i < 8; i++)
        g.key[i] = (uint32_t)key32[4 * i] | ((uint32_t)key32[4 * i + 1] << 8) |
                   ((uint32_t)key32[4 * i + 2] << 16) | ((uint32_t)key32[4 * i + 3] << 24);
    g.call = call; g.boot = boot;
    uint32_t blk[16];
    for (size_t j = 0; j < c->m; j++) {
        const uint32_t x = ((uint32_t)cc << c->logm) + (uint32_t)j;
        if ((x & 3) == 0 || j == 0) chacha_block(g.key, x >> 2, y, boot, call, SGO_RND_ROUNDS, blk);
        draws_of(c, blk, x, &draws[2 * j], &draws[2 * j + 1]);
    }
}

void sgo_poly_mul(const sgo_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out) {
    poly_mul(c, (const u128 *)a, (const u128 *)b, (u128 *)out);
}

void sgo_poly_mul_schoolbook(const sgo_ctx *c, const uint64_t *a, const uint64_t *b,
                             uint64_t *out) {
    poly_mul_schoolbook(c, (const u128 *)a, (const u128 *)b, (u128 *)out);
}

/* fhe.jl:519-530.  A is [4][2][m]; scratch holds 6 m residues.  g = NULL: rng = nothing; else the
 * flatten of a draws first (fhe.jl:524), then the flatten of b (fhe.jl:525), tagged y. */
static void external_product(const sgo_ctx *c, const u128 *a, const u128 *b, const u128 *A,
                             u128 *a_res, u128 *b_res, u128 *scratch, const rnd_t *g, uint32_t y) {
    size_t m = c->m;
    u128 Q = c->Q;
    u128 *u = scratch;            /* 4 polys: a_lo, a_hi, b_lo, b_hi  (fhe.jl:524-526) */
    u128 *prod = scratch + 4 * m;
    u128 *ra = scratch + 5 * m;   /* a_res / b_res may alias a / b */
    flatten_poly2(c, g, 0, y, a, u, u + m);                  /* utils.jl:253-264 */
    flatten_poly2(c, g, 1, y, b, u + 2 * m, u + 3 * m);
    memset(ra, 0, m * sizeof(u128));
    for (int i = 0; i < 4; i++) {                            /* fhe.jl:527 */
        poly_mul(c, u + i * m, A + (size_t)(i * 2 + 0) * m, prod);
        for (size_t k = 0; k < m; k++) ra[k] = addmod(ra[k], prod[k], Q);
    }
    u128 *rb = b_res;
    memset(rb, 0, m * sizeof(u128));
    for (int i = 0; i < 4; i++) {                            /* fhe.jl:528 */
        poly_mul(c, u + i * m, A + (size_t)(i * 2 + 1) * m, prod);
        for (size_t k = 0; k < m; k++) rb[k] = addmod(rb[k], prod[k], Q);
    }
    memcpy(a_res, ra, m * sizeof(u128));
}

void sgo_external_product(const sgo_ctx *c, const uint64_t *a, const uint64_t *b,
                          const uint64_t *A, uint64_t *a_res, uint64_t *b_res) {
    size_t m = c->m;
    u128 *scratch = (u128 *)malloc(6 * m * sizeof(u128));
    u128 *ta = (u128 *)malloc(2 * m * sizeof(u128));
    memcpy(ta, a, m * sizeof(u128));
    memcpy(ta + m, b, m * sizeof(u128));
    external_product(c, ta, ta + m, (const u128 *)A, (u128 *)a_res, (u128 *)b_res, scratch, NULL, 0);
    free(ta);
    free(scratch);
}

/* ---------------------------------------------------------------- PRNG, keys, LWE */

typedef struct { uint64_t s; } splitmix_t;

static inline uint64_t sm_next(splitmix_t *g) {
    uint64_t z = (g->s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static inline u128 sm_below_wide(splitmix_t *g, u128 bound) {
    uint64_t hi = sm_next(g), lo = sm_next(g);
    return (((u128)hi << 64) | lo) % bound;
}

/* fhe.jl:130-138 */
void sgo_private_key(const sgo_ctx *c, uint64_t seed, uint64_t *sk) {
    splitmix_t g = {seed};
    for (uint64_t i = 0; i < c->n; i++) sk[i] = sm_next(&g) & 1;
}

/* ChaCha block function (RFC 8439, section 2.3) with `rounds` rounds: key 8 words, state words
 * 12 .. 15 = w12 .. w15 (block counter and nonce for ChaCha20). */
static void chacha_block(const uint32_t key[8], uint32_t w12, uint32_t w13, uint32_t w14, uint32_t w15,
                         int rounds, uint32_t out[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u,
                      key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                      w12, w13, w14, w15};
    uint32_t x[16];
    memcpy(x, s, sizeof x);
#define ROTL(v, n) (((v) << (n)) | ((v) >> (32 - (n))))
#define QR(a, b, c, d)                                                                        \
    x[a] += x[b]; x[d] ^= x[a]; x[d] = ROTL(x[d], 16); x[c] += x[d]; x[b] ^= x[c]; x[b] = ROTL(x[b], 12); \
    x[a] += x[b]; x[d] ^= x[a]; x[d] = ROTL(x[d], 8);  x[c] += x[d]; x[b] ^= x[c]; x[b] = ROTL(x[b], 7);
    for (int i = 0; i < rounds / 2; i++) {
        QR(0, 4, 8, 12) QR(1, 5, 9, 13) QR(2, 6, 10, 14) QR(3, 7, 11, 15)
        QR(0, 5, 10, 15) QR(1, 6, 11, 12) QR(2, 7, 8, 13) QR(3, 4, 9, 14)
    }
#undef QR
#undef ROTL
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}
static void chacha20_block(const uint32_t key[8], uint32_t counter, const uint32_t nonce[3],
                           uint32_t out[16]) {
    chacha_block(key, counter, nonce[0], nonce[1], nonce[2], 20, out);
}

/* fhe.jl:181-201.  Randomness: ChaCha20 keyed with the 32-byte seed, one stream per
 * (domain, key row) = nonce (domain, k * 4 + row, 0); domain 1: the uniform polynomials a_row
 * (coefficient i = words 4 (i mod 4) .. + 3 of block i / 4: lo = w0 | w1 << 32,
 * hi = w2 | w3 << 32, value (hi 2^64 + lo) mod Q), domain 2: the noise e_row (coefficient i =
 * words 2 (i mod 8), + 1 of block i / 8: d = w0 | w1 << 32, value d mod (2 noise + 1) - noise).
 * The same streams as sgfhe_bkey_generate (csrc/kernels.h), so both give the same key. */
void sgo_bootstrap_key(const sgo_ctx *c, const uint64_t *sk, const uint8_t *seed, uint64_t noise,
                       uint64_t *bkey_words, int threads) {
    size_t m = c->m, n = c->n;
    u128 Q = c->Q;
    u128 *bkey = (u128 *)bkey_words;
    u128 *ext_key = (u128 *)calloc(m, sizeof(u128));         /* fhe.jl:185 resize */
    for (size_t i = 0; i < n; i++) ext_key[i] = sk[i];
    uint32_t key[8];
    for (int i = 0; i < 8; i++)
        key[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) |
                 ((uint32_t)seed[4 * i + 2] << 16) | ((uint32_t)seed[4 * i + 3] << 24);
    const u128 G[4][2] = {{1, 0}, {c->B % Q, 0}, {0, 1}, {0, c->B % Q}};   /* fhe.jl:119-122 */
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic)
    for (long kr = 0; kr < (long)(n * 4); kr++) {
        size_t k = kr / 4;
        int row = kr % 4;
        u128 *aj = bkey + ((k * 4 + row) * 2 + 0) * m;
        u128 *bj = bkey + ((k * 4 + row) * 2 + 1) * m;
        u128 *ej = (u128 *)malloc(m * sizeof(u128));
        uint32_t blk[16];
        const uint32_t nonce_a[3] = {1u, (uint32_t)kr, 0u}, nonce_e[3] = {2u, (uint32_t)kr, 0u};
        for (size_t i = 0; i < m; i++) {                                             /* :193 */
            if ((i & 3) == 0) chacha20_block(key, (uint32_t)(i >> 2), nonce_a, blk);
            const uint32_t *w = blk + 4 * (i & 3);
            u128 v = ((u128)(((uint64_t)w[3] << 32) | w[2]) << 64) | (((uint64_t)w[1] << 32) | w[0]);
            aj[i] = v % Q;
        }
        for (size_t i = 0; i < m; i++) {                                             /* :194 */
            if ((i & 7) == 0) chacha20_block(key, (uint32_t)(i >> 3), nonce_e, blk);
            const uint32_t *w = blk + 2 * (i & 7);
            uint64_t d = ((((uint64_t)w[1] << 32) | w[0])) % (2 * noise + 1);
            ej[i] = d >= noise ? (u128)(d - noise) : Q - (u128)(noise - d);
        }
        poly_mul(c, aj, ext_key, bj);                                               /* :195 */
        for (size_t i = 0; i < m; i++) bj[i] = addmod(bj[i], ej[i], Q);
        if (ext_key[k]) {                                    /* :196 constant-term add of s_k G */
            aj[0] = addmod(aj[0], G[row][0], Q);
            bj[0] = addmod(bj[0], G[row][1], Q);
        }
        free(ej);
    }
    free(ext_key);
}

/* fhe.jl:310-328 + split_ciphertext :287-290: a uniform, b = <a,s> + w + bit Dr, |w| <= Dr/8 */
void sgo_lwe_encrypt_bits(const sgo_ctx *c, const uint64_t *sk, const uint8_t *bits, size_t count,
                          uint64_t seed, uint64_t *a, uint64_t *b) {
    splitmix_t g = {seed};
    uint64_t r = c->r, w_range = c->Dr / 8;
    for (size_t t = 0; t < count; t++) {
        uint64_t acc = 0;
        for (uint64_t i = 0; i < c->n; i++) {
            uint64_t x = sm_next(&g) % r;
            a[t * c->n + i] = x;
            acc = (acc + x * sk[i]) % r;
        }
        uint64_t w = sm_next(&g) % (2 * w_range + 1);
        acc = (acc + w + r - w_range + (bits[t] ? c->Dr : 0)) % r;
        b[t] = acc;
    }
}

/* fhe.jl:504-507 */
int sgo_lwe_decrypt_bit(const sgo_ctx *c, const uint64_t *sk, const uint64_t *a, uint64_t b) {
    uint64_t r = c->r, acc = 0;
    for (uint64_t i = 0; i < c->n; i++) acc = (acc + a[i] * sk[i]) % r;
    uint64_t b1 = (b + r - acc) % r;
    return (int)(((b1 + c->Dr / 2) % r) / c->Dr);
}

/* ---------------------------------------------------------------- bootstrap */

/* fhe.jl:237-244, 0-based i0 = i - 1 with i >= n (both call sites, fhe.jl:586,589) */
static inline u128 extract_at(const u128 *a, size_t i0, size_t k) { return a[i0 - k]; }

/* One k-loop iteration in the algebra of the GPU path (`cpu_opt`): with
 * A = (x^j - 1) C_k + G and sum_i u_i G_i = (a, b) (the restore property of flatten,
 * test/internals.test.jl:138-140,161-165) the external product of fhe.jl:580-581 is
 *   acc <- acc + (x^j - 1) sum_row u_row (*) C_k[row],
 * so the key is held in the NTT domain (khat) and an iteration is 4 forward + 2 inverse NTTs
 * instead of 24.  Exact arithmetic mod Q: bit-identical to the reference-shaped path. */
static void iteration_opt(const sgo_ctx *c, const u128 *khat_k, uint64_t j, u128 *a, u128 *b,
                          u128 *work /* 6 m */, const rnd_t *g, uint32_t y) {
    size_t m = c->m;
    u128 Q = c->Q;
    mont_t mt = ctx_mont(c);
    u128 *u[4] = {work, work + m, work + 2 * m, work + 3 * m};
    u128 *P = work + 4 * m, *rot = work + 5 * m;
    flatten_poly2(c, g, 0, y, a, u[0], u[1]);                           /* fhe.jl:524-526 */
    flatten_poly2(c, g, 1, y, b, u[2], u[3]);
    for (int row = 0; row < 4; row++) ntt_fwd(c, u[row]);
    for (int col = 0; col < 2; col++) {
        for (size_t i = 0; i < m; i++) {
            u128 acc = 0;
            for (int row = 0; row < 4; row++)
                acc = addmod(acc, mont_mul(&mt, u[row][i], khat_k[((size_t)row * 2 + col) * m + i]), Q);
            P[i] = acc;
        }
        ntt_inv(c, P);
        for (size_t i = 0; i < m; i++) P[i] = mont_mul(&mt, P[i], c->minv_R2);
        mul_by_monomial(c, P, j, rot);                                  /* fhe.jl:554-556 on the product */
        u128 *dst = col ? b : a;
        for (size_t i = 0; i < m; i++) dst[i] = addmod(dst[i], submod(rot[i], P[i], Q), Q);
    }
}

/* The same iteration over the RNS2Number ring (src/rns.jl), Q = m1 m2: the key is held limb-wise in
 * the NTT domain (khat2[limb][k][row][col][slot], residues mod m_limb), the digit polynomials are
 * reduced into each limb (rns.jl:16-18), multiplied and summed there (rns.jl:51-60 act limb-wise), and
 * the two limb results of a column are put together by the CRT of rns.jl:32-40 before the rotation --
 * 2 x (4 + 2) NTTs per iteration instead of the 2 x 24 of the reference-shaped loop; exact, so the
 * same residues mod Q. */
static void iteration_opt_rns2(const sgo_ctx *c, const u128 *khat2, size_t polys_per_limb, uint64_t k,
                               uint64_t j, u128 *a, u128 *b, u128 *work /* 15 m */, const rnd_t *g,
                               uint32_t y) {
    size_t m = c->m;
    u128 Q = c->Q;
    mont_t mtQ;
    mont_init(&mtQ, Q);
    u128 *u[4] = {work, work + m, work + 2 * m, work + 3 * m};
    u128 *lu = work + 4 * m;            /* [limb][row][m] */
    u128 *PL = work + 12 * m;           /* [limb][m]: the column's product in each limb */
    u128 *P = work + 14 * m;
    flatten_poly2(c, g, 0, y, a, u[0], u[1]);                           /* fhe.jl:524-526 */
    flatten_poly2(c, g, 1, y, b, u[2], u[3]);
    for (int li = 0; li < 2; li++) {
        const sgo_ctx *L = c->limb[li];
        for (int row = 0; row < 4; row++) {
            u128 *t = lu + ((size_t)li * 4 + row) * m;
            for (size_t i = 0; i < m; i++) t[i] = u[row][i] % c->rns_m[li];          /* rns.jl:16-18 */
            ntt_fwd(L, t);
        }
    }
    for (int col = 0; col < 2; col++) {
        for (int li = 0; li < 2; li++) {
            const sgo_ctx *L = c->limb[li];
            mont_t mt = ctx_mont(L);
            const u128 *kh = khat2 + ((size_t)li * polys_per_limb + (size_t)k * 8) * m;
            u128 *out = PL + (size_t)li * m;
            for (size_t i = 0; i < m; i++) {
                u128 acc = 0;
                for (int row = 0; row < 4; row++)
                    acc = addmod(acc, mont_mul(&mt, lu[((size_t)li * 4 + row) * m + i],
                                               kh[((size_t)row * 2 + col) * m + i]), L->Q);
                out[i] = acc;
            }
            ntt_inv(L, out);
            for (size_t i = 0; i < m; i++) out[i] = mont_mul(&mt, out[i], L->minv_R2);
        }
        for (size_t i = 0; i < m; i++)                                                /* rns.jl:32-40 */
            P[i] = addmod(mulmod_plain(&mtQ, PL[i], c->rns_c[0]), mulmod_plain(&mtQ, PL[m + i], c->rns_c[1]), Q);
        u128 *rot = PL;                                                               /* (free again) */
        mul_by_monomial(c, P, j, rot);                                  /* fhe.jl:554-556 on the product */
        u128 *dst = col ? b : a;
        for (size_t i = 0; i < m; i++) dst[i] = addmod(dst[i], submod(rot[i], P[i], Q), Q);
    }
}

static int bootstrap_one(const sgo_ctx *c, const u128 *bkey, const u128 *khat, const uint64_t *a1,
                         uint64_t b1, const uint64_t *a2, uint64_t b2, uint64_t n_iters,
                         u128 *out_raw, u128 *acc_out, const rnd_t *g) {
    size_t m = c->m, n = c->n;
    u128 Q = c->Q;
    mont_t mt = ctx_mont(c);
    u128 *buf = (u128 *)malloc((size_t)(2 + 1 + 15 + 1) * m * sizeof(u128));   /* A (8 m) + scratch (6 m) or the 15 m of iteration_opt_rns2 */
    if (!buf) return -1;
    u128 *a = buf, *b = buf + m, *t = buf + 2 * m, *A = buf + 3 * m;
    u128 *scratch = buf + 11 * m, *rot = buf + 18 * m;

    uint64_t ub = (b1 + b2) % c->r;                                     /* fhe.jl:566 */
    /* fhe.jl:535-548 initial_poly: sum_{j=-(Dr-1)}^{Dr-1} x^j */
    memset(t, 0, m * sizeof(u128));
    for (long j = -(long)(c->Dr - 1); j <= (long)(c->Dr - 1); j++) {
        if (j >= 0) t[j % (long)m] = addmod(t[j % (long)m], 1, Q);
        else t[(size_t)(j + (long)m)] = submod(t[(size_t)(j + (long)m)], 1, Q);
    }
    memset(a, 0, m * sizeof(u128));                                     /* fhe.jl:570 */
    mul_by_monomial(c, t, 2 * m - ub % (2 * m), b);                     /* fhe.jl:572-573 */
    u128 dqM = mont_mul(&mt, c->DQ_tilde % Q, mt.R2);
    for (size_t i = 0; i < m; i++) b[i] = mont_mul(&mt, b[i], dqM);
    const u128 G[4][2] = {{1, 0}, {c->B % Q, 0}, {0, 1}, {0, c->B % Q}};

    for (uint64_t k = 0; k < n_iters && k < n; k++) {                   /* fhe.jl:579-582 */
        uint64_t j = (a1[k] + a2[k]) % c->r;                            /* fhe.jl:566 */
        if (khat && c->limb[0]) {
            iteration_opt_rns2(c, khat, n * 8, k, j, a, b, A, g, (uint32_t)k);
            continue;
        }
        if (khat) {
            iteration_opt(c, khat + (size_t)k * 8 * m, j, a, b, A, g, (uint32_t)k);
            continue;
        }
        for (int row = 0; row < 4; row++) {
            for (int col = 0; col < 2; col++) {                         /* fhe.jl:580, :554-556 */
                const u128 *C = bkey + (((size_t)k * 4 + row) * 2 + col) * m;
                u128 *Ap = A + (size_t)(row * 2 + col) * m;
                mul_by_monomial(c, C, j, rot);
                for (size_t i = 0; i < m; i++) Ap[i] = submod(rot[i], C[i], Q);
                Ap[0] = addmod(Ap[0], G[row][col], Q);                  /* `.+ G`: constant term */
            }
        }
        external_product(c, a, b, A, a, b, scratch, g, (uint32_t)k);    /* fhe.jl:581 */
    }
    if (acc_out) {
        memcpy(acc_out, a, m * sizeof(u128));
        memcpy(acc_out + m, b, m * sizeof(u128));
    }
    if (out_raw) {
        u128 *and_ = out_raw, *or_ = out_raw + (n + 1), *xor_ = out_raw + 2 * (n + 1);
        size_t i_and = 3 * m / 4, i_or = m / 4;                         /* fhe.jl:585-590 (0-based) */
        for (size_t k = 0; k < n; k++) {
            and_[k] = extract_at(a, i_and, k);
            u128 v = extract_at(a, i_or, k);
            or_[k] = v ? Q - v : 0;
        }
        and_[n] = addmod(c->DQ_tilde % Q, b[i_and], Q);
        or_[n] = submod(c->DQ_tilde % Q, b[i_or], Q);
        for (size_t k = 0; k <= n; k++) xor_[k] = submod(or_[k], and_[k], Q);   /* fhe.jl:592 */
    }
    free(buf);
    return 0;
}

static int bootstrap_batch(const sgo_ctx *c, int opt, const uint64_t *bkey, const uint64_t *a1,
                           const uint64_t *b1, const uint64_t *a2, const uint64_t *b2, size_t batch,
                           uint64_t *out, int raw, uint64_t n_iters, uint64_t *acc_out, int threads,
                           const rnd_t *g0, const uint32_t *boots) {
    size_t n = c->n, m = c->m;
    int rc = 0;
    if (opt && !c->use_ntt && !c->limb[0]) return -2;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic)
    for (long t = 0; t < (long)batch; t++) {
        u128 *rawbuf = (u128 *)malloc(3 * (n + 1) * sizeof(u128));
        rnd_t g;
        if (g0) { g = *g0; g.boot = boots ? boots[t] : g0->boot + (uint32_t)t; }   /* its index in the call */
        int r1 = bootstrap_one(c, opt ? NULL : (const u128 *)bkey, opt ? (const u128 *)bkey : NULL,
                               a1 + (size_t)t * n, b1[t], a2 + (size_t)t * n, b2[t], n_iters,
                               out ? rawbuf : NULL,
                               acc_out ? (u128 *)acc_out + (size_t)t * 2 * m : NULL, g0 ? &g : NULL);
        if (r1) {
#pragma omp atomic write
            rc = r1;
        } else if (out) {
            if (raw) {
                memcpy((u128 *)out + (size_t)t * 3 * (n + 1), rawbuf, 3 * (n + 1) * sizeof(u128));
            } else {
                for (size_t i = 0; i < 3 * (n + 1); i++)               /* fhe.jl:616-618,644-648 */
                    out[(size_t)t * 3 * (n + 1) + i] =
                        (uint64_t)rescale(c->r, rawbuf[i], c->Q, 1);
            }
        }
        free(rawbuf);
    }
    return rc;
}

int sgo_bootstrap_batch(const sgo_ctx *c, const uint64_t *bkey, const uint64_t *a1,
                        const uint64_t *b1, const uint64_t *a2, const uint64_t *b2, size_t batch,
                        uint64_t *out, int raw, uint64_t n_iters, uint64_t *acc_out, int threads) {
    return bootstrap_batch(c, 0, bkey, a1, b1, a2, b2, batch, out, raw, n_iters, acc_out, threads, NULL, NULL);
}

/* NTT-domain form of the bootstrap key for sgo_bootstrap_batch_opt: every polynomial through the
 * forward transform (same [k][row][col][slot] order).  Needs Q prime with 2m | Q - 1. */
int sgo_key_transform(const sgo_ctx *c, const uint64_t *bkey, uint64_t *khat, int threads) {
    size_t m = c->m, polys = c->n * 8;
    if (!c->use_ntt && c->limb[0]) {   /* RNS2Number ring: khat2[limb][poly][slot], residues mod m_limb */
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
        for (long p = 0; p < (long)(2 * polys); p++) {
            const int li = (int)(p / (long)polys);
            const u128 *src = (const u128 *)bkey + (size_t)(p % (long)polys) * m;
            u128 *dst = (u128 *)khat + (size_t)p * m;
            for (size_t i = 0; i < m; i++) dst[i] = src[i] % c->rns_m[li];          /* rns.jl:16-18 */
            ntt_fwd(c->limb[li], dst);
        }
        return 0;
    }
    if (!c->use_ntt) return -2;
    memcpy(khat, bkey, polys * m * sizeof(u128));
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
    for (long p = 0; p < (long)polys; p++) ntt_fwd(c, (u128 *)khat + (size_t)p * m);
    return 0;
}

int sgo_bootstrap_batch_opt(const sgo_ctx *c, const uint64_t *khat, const uint64_t *a1,
                            const uint64_t *b1, const uint64_t *a2, const uint64_t *b2, size_t batch,
                            uint64_t *out, int raw, uint64_t n_iters, uint64_t *acc_out, int threads) {
    return bootstrap_batch(c, 1, khat, a1, b1, a2, b2, batch, out, raw, n_iters, acc_out, threads, NULL, NULL);
}

/* bootstrap(bkey, rng, ...) (fhe.jl:608-621 with rng::AbstractRNG): the randomised flatten of
 * utils.jl:198-241 on the ChaCha8 stream keyed with key32; bootstrap t of the batch is bootstrap
 * boot0 + t of call `call`, or bootstrap boots[t] when `boots` is given (rows picked out of a larger
 * call).  opt != 0: `key` is the NTT-domain key (sgo_key_transform) and the k-loop runs in the GPU
 * path's algebra; same outputs. */
int sgo_bootstrap_batch_rnd(const sgo_ctx *c, int opt, const uint64_t *key, const uint64_t *a1,
                            const uint64_t *b1, const uint64_t *a2, const uint64_t *b2, size_t batch,
                            uint64_t *out, int raw, uint64_t n_iters, uint64_t *acc_out, int threads,
                            const uint8_t *key32, uint32_t call, uint32_t boot0, const uint32_t *boots) {
    rnd_t g;
    for (int i = 0; i < 8; i++)
        g.key[i] = (uint32_t)key32[4 * i] | ((uint32_t)key32[4 * i + 1] << 8) |
                   ((uint32_t)key32[4 * i + 2] << 16) | ((uint32_t)key32[4 * i + 3] << 24);
    g.call = call;
    g.boot = boot0;
    return bootstrap_batch(c, opt, key, a1, b1, a2, b2, batch, out, raw, n_iters, acc_out, threads, &g, boots);
}

/* ---------------------------------------------------------------- packing */

/* fhe.jl:632-641: flatten(rng, a) * A[l+1:2l, :]; A is the 4x2 slice [4][2][m].  g != NULL: the randomised
 * flatten, the draws of the polynomial addressed as accumulator 0 of the flatten tagged y. */
static void shortened_external_product(const sgo_ctx *c, const u128 *a, const u128 *A, u128 *a_res,
                                       u128 *b_res, u128 *scratch /* 3 m */, const rnd_t *g, uint32_t y) {
    size_t m = c->m;
    u128 Q = c->Q;
    u128 *u = scratch, *prod = scratch + 2 * m;
    flatten_poly2(c, g, 0, y, a, u, u + m);                                        /* fhe.jl:637 */
    memset(a_res, 0, m * sizeof(u128));
    memset(b_res, 0, m * sizeof(u128));
    for (int i = 0; i < 2; i++) {
        poly_mul(c, u + i * m, A + (size_t)((2 + i) * 2 + 0) * m, prod);          /* fhe.jl:638 */
        for (size_t k = 0; k < m; k++) a_res[k] = addmod(a_res[k], prod[k], Q);
        poly_mul(c, u + i * m, A + (size_t)((2 + i) * 2 + 1) * m, prod);          /* fhe.jl:639 */
        for (size_t k = 0; k < m; k++) b_res[k] = addmod(b_res[k], prod[k], Q);
    }
}

/* fhe.jl:660-696.  a: [n][n] LWE vectors over Z_r, b: [n]; w, v: [m] words in [0, r).
 * khat != NULL: the n bootstraps run in the NTT-domain algebra (sgo_key_transform; same bytes, a quarter
 * of the time); bkey is needed either way, for the half-width products.  key32 != NULL: rng != nothing on the
 * engine's ChaCha8 stream -- this ciphertext is number `ct` of call `call`: its bootstrap j draws as bootstrap
 * ct n + j of the call (fhe.jl:673), the flatten of as_i as accumulator 0 of the flatten tagged 2^31 | i of
 * "bootstrap" ct (fhe.jl:683-684).  Returns 0 on success. */
int sgo_pack_encrypted_bits_ex(const sgo_ctx *c, const uint64_t *bkey, const uint64_t *khat, const uint64_t *a,
                               const uint64_t *b, uint64_t *w, uint64_t *v, int threads, const uint8_t *key32,
                               uint32_t ct, uint32_t call) {
    size_t n = c->n, m = c->m;
    u128 Q = c->Q;
    int rc = 0;
    rnd_t g0;
    if (key32) {
        for (int i = 0; i < 8; i++)
            g0.key[i] = (uint32_t)key32[4 * i] | ((uint32_t)key32[4 * i + 1] << 8) |
                        ((uint32_t)key32[4 * i + 2] << 16) | ((uint32_t)key32[4 * i + 3] << 24);
        g0.call = call;
        g0.boot = ct;
    }
    if (khat && !c->use_ntt) return -2;
    u128 *raw = (u128 *)malloc(n * (n + 1) * sizeof(u128)); /* AND LWE of every bit: a[0..n), b */
    uint64_t *zeros = (uint64_t *)calloc(n, sizeof(uint64_t));
    u128 *wt = (u128 *)calloc(m, sizeof(u128)), *vt = (u128 *)calloc(m, sizeof(u128));
    if (!raw || !zeros || !wt || !vt) return -1;
    /* fhe.jl:669-673: bootstrap(trivial 1, bit_i), AND branch, un-reduced */
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic)
    for (long t = 0; t < (long)n; t++) {
        u128 *out3 = (u128 *)malloc(3 * (n + 1) * sizeof(u128));
        rnd_t g = g0;
        g.boot = ct * (uint32_t)n + (uint32_t)t;
        int r1 = bootstrap_one(c, khat ? NULL : (const u128 *)bkey, (const u128 *)khat, zeros, c->Dr,
                               a + (size_t)t * n, b[t], n, out3, NULL, key32 ? &g : NULL);
        if (r1) {
#pragma omp atomic write
            rc = r1;
        } else {
            memcpy(raw + (size_t)t * (n + 1), out3, (n + 1) * sizeof(u128));
        }
        free(out3);
    }
    if (rc == 0) {
        /* fhe.jl:675-687 */
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
        {
            u128 *as = (u128 *)calloc(m, sizeof(u128));
            u128 *ar = (u128 *)malloc(2 * m * sizeof(u128));
            u128 *scratch = (u128 *)malloc(3 * m * sizeof(u128));
            u128 *wl = (u128 *)calloc(m, sizeof(u128)), *vl = (u128 *)calloc(m, sizeof(u128));
#pragma omp for schedule(dynamic)
            for (long i = 0; i < (long)n; i++) {
                for (size_t j = 0; j < n; j++) as[j] = raw[j * (n + 1) + (size_t)i];   /* :676 */
                shortened_external_product(c, as, (const u128 *)bkey + (size_t)i * 8 * m, ar, ar + m,
                                           scratch, key32 ? &g0 : NULL, (1u << 31) | (uint32_t)i);
                for (size_t k = 0; k < m; k++) {
                    wl[k] = addmod(wl[k], ar[k], Q);
                    vl[k] = addmod(vl[k], ar[m + k], Q);
                }
            }
#pragma omp critical
            for (size_t k = 0; k < m; k++) {
                wt[k] = addmod(wt[k], wl[k], Q);
                vt[k] = addmod(vt[k], vl[k], Q);
            }
            free(as); free(ar); free(scratch); free(wl); free(vl);
        }
        for (size_t k = 0; k < m; k++) {
            u128 w1 = wt[k] ? Q - wt[k] : 0;                                     /* fhe.jl:689 */
            u128 bk = k < n ? raw[k * (n + 1) + n] : 0;                          /* fhe.jl:678 */
            u128 v1 = submod(bk, vt[k], Q);                                      /* fhe.jl:690 */
            w[k] = (uint64_t)rescale(c->r, w1, Q, 1);                            /* fhe.jl:692-693 */
            v[k] = (uint64_t)rescale(c->r, v1, Q, 1);
        }
    }
    free(raw); free(zeros); free(wt); free(vt);
    return rc;
}

int sgo_pack_encrypted_bits(const sgo_ctx *c, const uint64_t *bkey, const uint64_t *a,
                            const uint64_t *b, uint64_t *w, uint64_t *v, int threads) {
    return sgo_pack_encrypted_bits_ex(c, bkey, NULL, a, b, w, v, threads, NULL, 0, 0);
}
