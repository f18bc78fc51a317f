"""
TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

Big-integer CPU restatement of the deterministic (`rng = nothing`) gate bootstrap of
nucypher/SGFHE.jl, written as literally as possible against the reference sources so that it is
"obviously correct".  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import anything under `oracle/`.

PARITY STATUS: "parity unpinned" at the bit level against the Julia build.  The reference holds
no golden vectors / known-answer tests, Julia and DarkIntegers.jl
(~0.1.0, un-vendored, /root/reference/Project.toml:7,20) are absent from this image, so the
oracle is pinned (a) mathematically: with `rng = nothing` every step is exact arithmetic on
canonical representatives of Z_Q / Z_r, so any correct implementation yields the same bytes, and
(b) by the reference's own property tests restated in tests/test_oracle_properties.py
(test/internals.test.jl:26-166, test/api.test.jl:45-83).

Every function cites the reference file:line it follows (paths relative to /root/reference).
Polynomial products use Kronecker substitution on Python integers: an independent algorithm
from the NTTs used by the C oracle and by the HIP engine.
"""

from dataclasses import dataclass

MASK64 = (1 << 64) - 1


# ----------------------------------------------------------------------------------------------
# Deterministic PRNGs shared (bit for bit) with oracle/sgfhe_oracle.c -- the build's own generators,
# not Julia's MersenneTwister: SplitMix64 for the test plumbing (private keys, LWE
# encryptions), ChaCha20 for the bootstrap key (the stream the HIP engine's key generation uses).
# ----------------------------------------------------------------------------------------------

class SplitMix64:
    def __init__(self, seed):
        self.s = seed & MASK64

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)

    def below_wide(self, bound):
        """Value in [0, bound) from 128 random bits: ((hi << 64) | lo) mod bound; hi drawn first."""
        hi = self.next()
        lo = self.next()
        return ((hi << 64) | lo) % bound

    def below(self, bound):
        """Value in [0, bound) from one 64-bit draw (bound < 2^63)."""
        return self.next() % bound


def chacha_blocks(key32, w12, w13, w14, w15, rounds=20):
    """ChaCha block function (RFC 8439 section 2.3, `rounds` rounds) on whole ranges of blocks:
    key32 = 32 bytes; w12 .. w15 = state words 12 .. 15, each a scalar or an array (broadcast).
    Returns an [nblocks][16] array of words."""
    import numpy as np
    key = [int.from_bytes(key32[4 * i:4 * i + 4], "little") for i in range(8)]
    ws = [np.atleast_1d(np.asarray(v, dtype=np.uint64).astype(np.uint32)) for v in (w12, w13, w14, w15)]
    nblocks = max(len(v) for v in ws)
    init = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + key
    s = [np.full(nblocks, v, dtype=np.uint32) for v in init] + [np.broadcast_to(v, (nblocks,)).copy() for v in ws]
    x = [v.copy() for v in s]

    def rotl(v, n):
        return (v << np.uint32(n)) | (v >> np.uint32(32 - n))

    def qr(a, b, c, d):
        x[a] = x[a] + x[b]; x[d] = rotl(x[d] ^ x[a], 16)
        x[c] = x[c] + x[d]; x[b] = rotl(x[b] ^ x[c], 12)
        x[a] = x[a] + x[b]; x[d] = rotl(x[d] ^ x[a], 8)
        x[c] = x[c] + x[d]; x[b] = rotl(x[b] ^ x[c], 7)

    assert rounds % 2 == 0
    with np.errstate(over="ignore"):
        for _ in range(rounds // 2):
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
        return np.stack([x[i] + s[i] for i in range(16)], axis=1)


def chacha20_blocks(key32, nonce, nblocks):
    """ChaCha20 (RFC 8439) for block counters 0 .. nblocks-1, nonce = three 32-bit words: the
    generator of the bootstrap key, shared with oracle/sgfhe_oracle.c and the HIP engine's
    k_keygen_draw."""
    import numpy as np
    return chacha_blocks(key32, np.arange(nblocks, dtype=np.uint64), nonce[0], nonce[1], nonce[2], 20)


def seed_bytes(seed):
    """Key-generation seed: 32 bytes, or an int taken as 32 little-endian bytes."""
    if isinstance(seed, (bytes, bytearray)):
        assert len(seed) == 32
        return bytes(seed)
    return int(seed).to_bytes(32, "little")


# ----------------------------------------------------------------------------------------------
# Primality / find_modulus
# ----------------------------------------------------------------------------------------------

_SMALL_PRIMES = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)


def is_prime(x):
    """Deterministic Miller-Rabin for x < 3.3e24, probabilistic-strong beyond (40 fixed bases).
    Stands in for Primes.isprime (src/utils.jl:19)."""
    if x < 2:
        return False
    for p in _SMALL_PRIMES:
        if x % p == 0:
            return x == p
    d = x - 1
    s = 0
    while d % 2 == 0:
        d //= 2
        s += 1
    bases = _SMALL_PRIMES + (41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97, 101, 103, 107,
                             109, 113, 127, 131, 137, 139, 149, 151, 157, 163, 167, 173)
    for a in bases:
        if a % x == 0:
            continue
        y = pow(a, d, x)
        if y == 1 or y == x - 1:
            continue
        for _ in range(s - 1):
            y = y * y % x
            if y == x - 1:
                break
        else:
            return False
    return True


def find_modulus(n, qmin, qmax=None):
    """src/utils.jl:7-28: smallest prime q >= qmin with q - 1 a multiple of n (q <= qmax)."""
    j = -((-(qmin - 1)) // n)          # cld(qmin - 1, n)
    while True:
        q = j * n + 1
        if qmax is not None and q > qmax:
            break
        if is_prime(q):
            return q
        j += 1
    raise ValueError("Could not find a modulus between %d and %s" % (qmin, qmax))


# ----------------------------------------------------------------------------------------------
# Params
# ----------------------------------------------------------------------------------------------

@dataclass(frozen=True)
class Params:
    """src/fhe.jl:27-99.  `ell` is the decomposition length fixed at fhe.jl:576."""
    n: int
    r: int
    q: int
    Q: int
    t: int
    m: int
    B: int
    Dr: int
    Dq: int
    DQ_tilde: int
    ell: int = 2

    @staticmethod
    def make(n):
        """Params(n): src/fhe.jl:43-97."""
        assert n >= 64 and (n & (n - 1)) == 0                       # fhe.jl:45-46
        r = 16 * n                                                  # fhe.jl:53
        q = find_modulus(2 * n, r * n)                              # fhe.jl:57
        t = r.bit_length() - 1 - 1                                  # fhe.jl:61  log2(r) - 1
        m = r // 2                                                  # fhe.jl:62
        Qmin = r ** 4 * n ** 2 * 1220                               # fhe.jl:64
        Qmax = r ** 4 * n ** 2 * 1225                               # fhe.jl:65
        Q = find_modulus(2 * m, Qmin, Qmax)                         # fhe.jl:69
        B = r ** 2 * n * 35                                         # fhe.jl:87
        return Params(n=n, r=r, q=q, Q=Q, t=t, m=m, B=B, Dr=r // 4, Dq=q // 4,
                      DQ_tilde=Q // 8)                              # fhe.jl:88-90

    @staticmethod
    def custom(n, Q, B, r=None, m=None, DQ_tilde=None):
        """Synthetic parameter sets (BASELINE.json configs 3 and 4): same
        structure (r = 16 n, m = r / 2, ell = 2) with a caller-chosen modulus and gadget base."""
        r = 16 * n if r is None else r
        m = r // 2 if m is None else m
        assert B * B >= Q                                           # src/utils.jl:145
        return Params(n=n, r=r, q=0, Q=Q, t=r.bit_length() - 2, m=m, B=B, Dr=r // 4, Dq=0,
                      DQ_tilde=Q // 8 if DQ_tilde is None else DQ_tilde)


def gadget_matrix(p):
    """src/fhe.jl:119-122."""
    return [[1, 0], [p.B, 0], [0, 1], [0, p.B]]


# ----------------------------------------------------------------------------------------------
# Polynomials in Z_Q[x]/(x^N + 1): plain Python lists of canonical residues
# ----------------------------------------------------------------------------------------------

def poly_mul(a, b, Q):
    """DarkIntegers `Polynomial * Polynomial` with negacyclic_modulus: exact product mod
    (x^N + 1, Q) (call sites src/fhe.jl:195,527-528).  Kronecker substitution."""
    N = len(a)
    assert len(b) == N
    slot = (2 * (Q - 1).bit_length() + N.bit_length() + 8 + 7) // 8      # bytes per slot
    pa = int.from_bytes(b"".join(c.to_bytes(slot, "little") for c in a), "little")
    pb = int.from_bytes(b"".join(c.to_bytes(slot, "little") for c in b), "little")
    prod = (pa * pb).to_bytes(2 * N * slot, "little")
    out = [0] * N
    for i in range(N):
        lo = int.from_bytes(prod[i * slot:(i + 1) * slot], "little")
        hi = int.from_bytes(prod[(i + N) * slot:(i + N + 1) * slot], "little")
        out[i] = (lo - hi) % Q
    return out


def poly_mul_schoolbook(a, b, Q):
    """O(N^2) reference for poly_mul (used only by tests)."""
    N = len(a)
    out = [0] * N
    for i, x in enumerate(a):
        if x == 0:
            continue
        for j, y in enumerate(b):
            k = i + j
            if k < N:
                out[k] = (out[k] + x * y) % Q
            else:
                out[k - N] = (out[k - N] - x * y) % Q
    return out


def poly_add(a, b, Q):
    return [(x + y) % Q for x, y in zip(a, b)]


def poly_sub(a, b, Q):
    return [(x - y) % Q for x, y in zip(a, b)]


def mul_by_monomial(a, j, Q):
    """DarkIntegers mul_by_monomial(p, j): p * x^j mod (x^N + 1), any integer j
    (docs/src/theory.md:23-32; call sites src/fhe.jl:555,573)."""
    N = len(a)
    j %= 2 * N
    out = [0] * N
    for i, c in enumerate(a):
        k = i + j
        s = (k // N) & 1
        out[k % N] = (Q - c) % Q if s else c
    return out


def resize(a, new_len):
    """DarkIntegers resize: zero-pad (src/fhe.jl:185)."""
    return list(a) + [0] * (new_len - len(a))


# ----------------------------------------------------------------------------------------------
# rescale / reduce_modulus / flatten
# ----------------------------------------------------------------------------------------------

def rescale(new_max, x, old_max, round_result):
    """src/utils.jl:78-92."""
    q, r = divmod(x * new_max, old_max)               # mulhilo + divremhilo, utils.jl:81-82
    if round_result:
        if r >= old_max // 2 + (1 if old_max % 2 else 0):       # utils.jl:84
            q += 1
            if q == new_max:                                     # utils.jl:86-88
                q = 0
    return q


def reduce_modulus(new_modulus, x, old_modulus, floor_result=False, new_max=None):
    """src/utils.jl:107-117 (scalar)."""
    return rescale(new_modulus if new_max is None else new_max, x, old_modulus, not floor_result)


def flatten(a, B, ell, Q):
    """Deterministic flatten, src/utils.jl:155-189.  Returns `ell` residues mod Q."""
    s = (B - 1) // 2 if B % 2 else B // 2 - 1                     # utils.jl:162-166
    pwrs = [B ** i for i in range(ell)]                           # utils.jl:168
    offset = sum(pwrs) * s                                        # utils.jl:169
    decomp = [0] * ell
    a = (a + offset) % Q                                          # utils.jl:179
    for i in range(ell - 1, 0, -1):                               # utils.jl:170-175
        quot, a = divmod(a, pwrs[i])                              # `r, a = divrem(...)`: r = quotient
        decomp[i] = quot
    decomp[0] = a                                                 # utils.jl:181
    return [(d - s) % Q for d in decomp]                          # utils.jl:183-185


def flatten_xmax(B):
    """Bound of the random shift of the randomised flatten, src/utils.jl:210-214."""
    return (B - 1) // 2 * 3 if B % 2 else B // 2 * 3


def flatten_random(draw, a, B, ell, Q):
    """Randomised flatten, src/utils.jl:198-241: x_i = rand(rng, -xmax:xmax) for i = 1..ell
    (:229-231), rand_a = a - sum x_i B^(i-1) (:233-234), y = flatten(nothing, rand_a) (:236),
    result x_i + y_i (:237-239), all in Z_Q.  `draw(i)` returns the i-th draw (0-based) of this
    coefficient as an integer in [-xmax, xmax]."""
    x = [draw(i) for i in range(ell)]
    rand_a = a
    for i in range(ell):
        rand_a = (rand_a - x[i] * B ** i) % Q
    y = flatten(rand_a, B, ell, Q)
    return [(x[i] + y[i]) % Q for i in range(ell)]


def flatten_poly(a, B, ell, Q, draws=None):
    """src/utils.jl:253-264.  draws = None: deterministic (rng = nothing); else a callable
    (coefficient index j, digit index i) -> integer in [-xmax, xmax] standing for the reference's
    rng: flatten_poly visits the coefficients in order and every coefficient draws ell values."""
    results = [[0] * len(a) for _ in range(ell)]
    for j, c in enumerate(a):
        if draws is None:
            d = flatten(c, B, ell, Q)
        else:
            d = flatten_random(lambda i, j=j: draws(j, i), c, B, ell, Q)
        for i in range(ell):
            results[i][j] = d[i]
    return results


# ----------------------------------------------------------------------------------------------
# The randomness of the HIP engine's randomised flatten, restated: a ChaCha counter stream (the
# RFC 8439 block function with RND_ROUNDS = 8 rounds, "ChaCha8") keyed with 32 bytes.  The 128
# bits of coefficient x are words 4 (x mod 4) .. + 3 of the block whose state words 12 .. 15 are
#   (x div 4   with x = (c << log2 m) + j for coefficient j of accumulator c (0 = a, 1 = b),
#    y: k for the flatten feeding k-loop iteration k (0-based); 2^31 | i for the flatten of
#       as_i in pack_encrypted_bits,
#    z: index of the bootstrap within the call (of the ciphertext, for packing),
#    w: number of the call since the key was set).
# They give r_0 = (lo64 * span) >> 64, r_1 = (hi64 * span) >> 64 with span = 2 xmax + 1, and the
# draws x_i = r_i - xmax (sgfhe.jl_amd/csrc/kernels.h rnd128 / random_digits).
# The reference draws from the caller's Julia rng, whose stream cannot be reproduced here
# (SURVEY.md F6): this pins the engine's random mode to the reference's *algorithm* on the
# engine's stream.
# ----------------------------------------------------------------------------------------------

RND_ROUNDS = 8


def rnd128(key32, ctr, rounds=RND_ROUNDS):
    """The four 32-bit words of the draw addressed by ctr = (x, y, z, w)."""
    x, y, z, w = (int(v) & 0xFFFFFFFF for v in ctr)
    blk = chacha_blocks(key32, x >> 2, y, z, w, rounds)[0]
    return [int(v) for v in blk[4 * (x & 3):4 * (x & 3) + 4]]


class ChaChaFlatten:
    """Draw source of one bootstrap: rng(c, y)(j, i) is the draw for digit i of coefficient j of
    accumulator c in the flatten tagged y.  seed: the 32-byte key, or an int taken as 32
    little-endian bytes (sgfhe_set_random_flatten)."""

    def __init__(self, p, seed, boot=0, call=0):
        self.p, self.key, self.boot, self.call = p, seed_bytes(seed), boot, call
        self.xmax = flatten_xmax(p.B)
        self.logm = p.m.bit_length() - 1

    def draws(self, c, y):
        import numpy as np
        span = 2 * self.xmax + 1
        m = self.p.m
        # every block of the polynomial at once (coefficients 4 q .. 4 q + 3 share block q)
        x0 = (c << self.logm) >> 2
        blocks = chacha_blocks(self.key, np.arange(x0, x0 + (m + 3) // 4, dtype=np.uint64), y, self.boot,
                               self.call, RND_ROUNDS)
        words = blocks.reshape(-1).tolist()

        def f(j, i):
            lo, hi = words[4 * j + 2 * i], words[4 * j + 2 * i + 1]
            return ((((hi << 32) | lo) * span) >> 64) - self.xmax
        return f


# ----------------------------------------------------------------------------------------------
# Keys, LWE plumbing
# ----------------------------------------------------------------------------------------------

def private_key(p, seed):
    """src/fhe.jl:130-138: n random bits (the build's PRNG, one draw per bit, low bit)."""
    g = SplitMix64(seed)
    return [g.next() & 1 for _ in range(p.n)]


def bootstrap_key(p, sk, seed, noise=None):
    """src/fhe.jl:181-201.  Returns key[k][row][col] = list of m residues mod Q.
    Randomness (the build's own, not Julia's): ChaCha20 keyed with the 32-byte seed, one stream
    per (domain, key row), nonce = (domain, k * 4 + row, 0); domain 1 = a_row (coefficient i from
    words 4 (i mod 4) .. + 3 of block i / 4 as a 128-bit value mod Q), domain 2 = e_row
    (coefficient i from words 2 (i mod 8), + 1 of block i / 8 as a 64-bit value mod 2 noise + 1)."""
    key32 = seed_bytes(seed)
    noise = p.n if noise is None else noise
    if not 0 <= noise < (1 << 30) or 2 * noise >= p.Q:            # the bound sgfhe_bkey_generate enforces
        raise ValueError("noise must be below 2^30 and below Q / 2")
    ext_key = resize(sk, p.m)                                     # fhe.jl:185
    G = gadget_matrix(p)                                          # fhe.jl:190
    key = []
    for k in range(p.n):
        C = []
        for row in range(4):
            wa = chacha20_blocks(key32, (1, k * 4 + row, 0), (p.m + 3) // 4).reshape(-1).tolist()
            we = chacha20_blocks(key32, (2, k * 4 + row, 0), (p.m + 7) // 8).reshape(-1).tolist()
            aj = [(wa[4 * i] | (wa[4 * i + 1] << 32) | (wa[4 * i + 2] << 64) | (wa[4 * i + 3] << 96)) % p.Q
                  for i in range(p.m)]                            # fhe.jl:193
            ej = [((we[2 * i] | (we[2 * i + 1] << 32)) % (2 * noise + 1) - noise) % p.Q
                  for i in range(p.m)]                            # fhe.jl:194
            bj = poly_add(poly_mul(aj, ext_key, p.Q), ej, p.Q)    # fhe.jl:195
            # fhe.jl:196: `.+ ext_key.coeffs[i] * G` adds to the constant coefficient
            aj[0] = (aj[0] + ext_key[k] * G[row][0]) % p.Q
            bj[0] = (bj[0] + ext_key[k] * G[row][1]) % p.Q
            C.append([aj, bj])
        key.append(C)
    return key


def lwe_encrypt_bit(p, sk, bit, g):
    """One LWE of `bit` over Z_r with the noise of src/fhe.jl:318-322 after split_ciphertext
    (fhe.jl:287-290): a uniform in [0, r)^n, b = <a, s> + w + bit * Dr, |w| <= Dr / 8."""
    a = [g.below(p.r) for _ in range(p.n)]
    w_range = p.Dr // 8
    w = g.below(2 * w_range + 1) - w_range
    b = (sum(x * s for x, s in zip(a, sk)) + w + bit * p.Dr) % p.r
    return a, b


def lwe_decrypt_bit(p, sk, lwe):
    """src/fhe.jl:504-507."""
    a, b = lwe
    b1 = (b - sum(x * s for x, s in zip(a, sk))) % p.r
    return ((b1 + p.Dr // 2) % p.r) // p.Dr


def extract(a, i, n, Q):
    """src/fhe.jl:237-244 with the reference's 1-based i."""
    N = len(a)
    assert i <= N
    if i < n:
        head = [a[k - 1] for k in range(i, 0, -1)]
        tail = [(Q - a[k - 1]) % Q for k in range(N, N - (n - i - 1) - 1, -1)]
        return head + tail
    return [a[k - 1] for k in range(i, i - n, -1)]


# ----------------------------------------------------------------------------------------------
# Bootstrap
# ----------------------------------------------------------------------------------------------

def external_product(a, b, A, B, ell, Q, draws_a=None, draws_b=None):
    """src/fhe.jl:519-530.  draws_a / draws_b = None: rng = nothing; else the draw sources of the
    flatten of a (first, fhe.jl:524) and of b (second, fhe.jl:525)."""
    u = flatten_poly(a, B, ell, Q, draws_a) + flatten_poly(b, B, ell, Q, draws_b)   # fhe.jl:524-526
    N = len(a)
    a_res = [0] * N
    b_res = [0] * N
    for i in range(2 * ell):
        a_res = poly_add(a_res, poly_mul(u[i], A[i][0], Q), Q)           # fhe.jl:527
        b_res = poly_add(b_res, poly_mul(u[i], A[i][1], Q), Q)           # fhe.jl:528
    return a_res, b_res


def initial_poly(p):
    """src/fhe.jl:535-548."""
    coeffs = [0] * p.m
    for i in range(-(p.Dr - 1), p.Dr):
        sign = 1 if ((i // p.m) % 2 == 0) else -1
        coeffs[i % p.m] = (coeffs[i % p.m] + sign) % p.Q
    return coeffs


def mul_by_xj_minus_one(poly, j, Q):
    """src/fhe.jl:554-556."""
    return poly_sub(mul_by_monomial(poly, j, Q), poly, Q)


def bootstrap_internal(p, bkey, lwe1, lwe2, trace=None, rng=None):
    """src/fhe.jl:559-595.  Returns three LWEs over Z_Q: (AND, OR, XOR).  rng = None is the
    reference's `rng = nothing`; a ChaChaFlatten selects the randomised flatten."""
    Q = p.Q
    ua = [(x + y) % p.r for x, y in zip(lwe1[0], lwe2[0])]               # fhe.jl:566
    ub = (lwe1[1] + lwe2[1]) % p.r
    t = initial_poly(p)                                                  # fhe.jl:568
    a = [0] * p.m                                                        # fhe.jl:570
    b = [(c * p.DQ_tilde) % Q for c in mul_by_monomial(t, -ub, Q)]       # fhe.jl:572-573
    G = gadget_matrix(p)
    for k in range(p.n):                                                 # fhe.jl:579-582
        A = []
        for row in range(4):
            Arow = []
            for col in range(2):
                x = mul_by_xj_minus_one(bkey[k][row][col], ua[k], Q)
                x[0] = (x[0] + G[row][col]) % Q                          # `.+ G`: constant term
                Arow.append(x)
            A.append(Arow)
        if rng is None:
            a, b = external_product(a, b, A, p.B, p.ell, Q)
        else:
            a, b = external_product(a, b, A, p.B, p.ell, Q, rng.draws(0, k), rng.draws(1, k))
        if trace is not None:
            trace(k, a, b)
    m, n = p.m, p.n
    and_a = extract(a, 3 * m // 4 + 1, n, Q)                             # fhe.jl:585-587
    and_b = (p.DQ_tilde + b[3 * m // 4]) % Q
    or_a = [(Q - x) % Q for x in extract(a, m // 4 + 1, n, Q)]           # fhe.jl:588-590
    or_b = (p.DQ_tilde - b[m // 4]) % Q
    xor_a = [(x - y) % Q for x, y in zip(or_a, and_a)]                   # fhe.jl:592
    xor_b = (or_b - and_b) % Q
    return (and_a, and_b), (or_a, or_b), (xor_a, xor_b)


def bootstrap(p, bkey, lwe1, lwe2, rng=None):
    """src/fhe.jl:608-621.  Returns three LWEs over Z_r: (AND, OR, XOR)."""
    out = []
    for a, b in bootstrap_internal(p, bkey, lwe1, lwe2, rng=rng):
        out.append(([reduce_modulus(p.r, x, p.Q) for x in a],            # fhe.jl:616-618,644-648
                    reduce_modulus(p.r, b, p.Q)))
    return tuple(out)


# ----------------------------------------------------------------------------------------------
# RNS2Number (src/rns.jl) -- config 4 boundary conversions
# ----------------------------------------------------------------------------------------------

def rns2_from_int(x, m1, m2):
    """src/rns.jl:16-18."""
    return x % m1, x % m2


def rns2_to_int(v1, v2, m1, m2):
    """src/rns.jl:32-40."""
    m = m1 * m2
    c1 = pow(m2, m1 - 1, m)
    c2 = pow(m1, m2 - 1, m)
    return (v1 * c1 + v2 * c2) % m


# ----------------------------------------------------------------------------------------------
# Packing LWEs into an RLWE ciphertext (second caller of the hot path)
# ----------------------------------------------------------------------------------------------

def shortened_external_product(a, A, B, ell, Q, draws=None):
    """src/fhe.jl:632-641: flatten(a) * A[l+1:2l, :]."""
    u = flatten_poly(a, B, ell, Q, draws)                                 # fhe.jl:637
    N = len(a)
    a_res = [0] * N
    b_res = [0] * N
    for i in range(ell):
        a_res = poly_add(a_res, poly_mul(u[i], A[ell + i][0], Q), Q)      # fhe.jl:638
        b_res = poly_add(b_res, poly_mul(u[i], A[ell + i][1], Q), Q)      # fhe.jl:639
    return a_res, b_res


def reduce_modulus_poly(new_modulus, poly, old_modulus):
    """src/utils.jl:120-127."""
    return [reduce_modulus(new_modulus, x, old_modulus) for x in poly]


def pack_encrypted_bits(p, bkey, enc_bits, seed=None, ct=0, call=0):
    """src/fhe.jl:660-696.  enc_bits: n LWEs (a, b) over Z_r.  seed = None: rng = nothing; else
    the engine's ChaCha stream (ciphertext `ct` of call `call`): bootstrap j of the group is
    bootstrap ct * n + j of the call, and the flatten of as_i draws with y = 2^31 | i, z = ct.
    Returns the RLWE (w, v) over Z_r, two lists of m coefficients."""
    Q = p.Q
    assert len(enc_bits) == p.n                                           # fhe.jl:667
    enc_trivial = ([0] * p.n, p.Dr)                                       # fhe.jl:669-671
    rngs = [None if seed is None else ChaChaFlatten(p, seed, ct * p.n + j, call) for j in range(p.n)]
    new_lwes = [bootstrap_internal(p, bkey, enc_trivial, eb, rng=rngs[j])[0]
                for j, eb in enumerate(enc_bits)]                         # fhe.jl:673
    as_ = [resize([new_lwes[j][0][i] for j in range(p.n)], p.m) for i in range(p.n)]  # :675-677
    b = resize([lw[1] for lw in new_lwes], p.m)                           # fhe.jl:678
    w_tilde = [0] * p.m
    v_tilde = [0] * p.m
    pack_rng = None if seed is None else ChaChaFlatten(p, seed, ct, call)
    for i in range(p.n):                                                  # fhe.jl:683-687
        draws = None if seed is None else pack_rng.draws(0, (1 << 31) | i)
        w, v = shortened_external_product(as_[i], bkey[i], p.B, p.ell, Q, draws)
        w_tilde = poly_add(w_tilde, w, Q)
        v_tilde = poly_add(v_tilde, v, Q)
    w1 = [(Q - x) % Q for x in w_tilde]                                   # fhe.jl:689
    v1 = poly_sub(b, v_tilde, Q)                                          # fhe.jl:690
    return reduce_modulus_poly(p.r, w1, Q), reduce_modulus_poly(p.r, v1, Q)   # fhe.jl:692-693


def poly_mul_mod_pow2(a, b, r):
    """Exact negacyclic product over Z_r (r a power of two), schoolbook on the sparse key."""
    N = len(a)
    out = [0] * N
    for i, y in enumerate(b):
        if y == 0:
            continue
        for j, x in enumerate(a):
            k = i + j
            if k < N:
                out[k] = (out[k] + x * y) % r
            else:
                out[k - N] = (out[k - N] - x * y) % r
    return out


def decrypt_ciphertext(p, sk, w, v):
    """decrypt(key, ::Ciphertext) (src/fhe.jl:471-494): first n coefficients of b - key * a."""
    key_poly = resize(sk, p.m)                                            # fhe.jl:475
    prod = poly_mul_mod_pow2(w, key_poly, p.r)
    b1 = [(x - y) % p.r for x, y in zip(v, prod)][:p.n]                   # fhe.jl:479-482
    return [((x + p.Dr // 2) % p.r) // p.Dr for x in b1]                  # fhe.jl:491-493


def split_ciphertext(p, w, v):
    """split_ciphertext(::Ciphertext) (src/fhe.jl:287-290): n LWEs from a length-m RLWE."""
    return [([x % p.r for x in extract(w, i, p.n, p.r)], v[i - 1]) for i in range(1, p.n + 1)]
