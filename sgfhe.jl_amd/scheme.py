"""Host-side mirror of the SGFHE.jl API around the hot path (paths relative to /root/reference):
`PrivateKey`, `BootstrapKey`, `encrypt` / `split_ciphertext` / `decrypt`, `bootstrap`.

Same names, argument order and error behaviour as src/fhe.jl; the work of `bootstrap()`
(src/fhe.jl:608-621) is done by the HIP engine behind the C ABI.  Random numbers come from a
numpy Generator (the reference's MersenneTwister streams are not reproducible outside Julia,
SURVEY.md F6): `rng=None` is the deterministic, bit-exact bootstrap; a Generator selects the
randomised flatten (functional parity only).
"""

import numpy as np

from .engine import Engine
from .params import Params

_M64 = 0xFFFFFFFFFFFFFFFF


# ---- small host helpers ---------------------------------------------------------------------

def _negacyclic_mul_small(a, s, r):
    """a * s mod (x^n + 1, r) for length-n integer vectors with r a power of two <= 2^32
    (the `Polynomial * Polynomial` of src/fhe.jl:322,479 over Z_r)."""
    n = len(a)
    a = np.asarray(a, dtype=np.uint64)
    s = np.asarray(s, dtype=np.uint64)
    full = np.zeros(2 * n, dtype=np.uint64)
    for i in np.nonzero(s)[0]:
        full[i:i + n] += a * s[i]            # wraps mod 2^64; r | 2^64
    return (full[:n] - full[n:]) & np.uint64(r - 1)


def _negacyclic_mul_signed(a, s, q):
    """a * s mod (x^n + 1, q) for a in [0, q) (q < 2^31) and a short signed s (|s_i| small):
    the `Polynomial * Polynomial` over Z_q of src/fhe.jl:164,399-400."""
    n = len(a)
    full = np.convolve(np.asarray(a, dtype=np.int64), np.asarray(s, dtype=np.int64))
    full = np.concatenate([full, np.zeros(2 * n - len(full), dtype=np.int64)])
    return ((full[:n] - full[n:]) % q).astype(np.uint64)


def _rescale(new_max, x, old_max, round_result):
    """rescale (src/utils.jl:78-92) on a vector: floor or round of x new_max / old_max, with the
    rounded value new_max wrapping to 0.  All operands below 2^31."""
    prod = np.asarray(x, dtype=np.uint64) * np.uint64(new_max)
    quo, rem = prod // np.uint64(old_max), prod % np.uint64(old_max)
    if round_result:
        quo = quo + (rem >= np.uint64(old_max // 2 + (old_max & 1))).astype(np.uint64)
        quo[quo == new_max] = 0
    return quo


def _negacyclic_mul_big(a, b, Q):
    """Exact product mod (x^N + 1, Q) of Python-int coefficient lists (Kronecker substitution)."""
    N = len(a)
    slot = (2 * (Q - 1).bit_length() + N.bit_length() + 15) // 8
    pa = int.from_bytes(b"".join(int(c).to_bytes(slot, "little") for c in a), "little")
    pb = int.from_bytes(b"".join(int(c).to_bytes(slot, "little") for c in b), "little")
    prod = (pa * pb).to_bytes(2 * N * slot, "little")
    out = [0] * N
    for i in range(N):
        lo = int.from_bytes(prod[i * slot:(i + 1) * slot], "little")
        hi = int.from_bytes(prod[(i + N) * slot:(i + N + 1) * slot], "little")
        out[i] = (lo - hi) % Q
    return out


def _uniform_below(rng, bound, count):
    """`count` Python ints uniform in [0, bound) by rejection on bound.bit_length() bits."""
    bits = bound.bit_length()
    mask = (1 << bits) - 1
    out = []
    while len(out) < count:
        raw = rng.integers(0, 1 << 63, size=(2 * (count - len(out)), 2), dtype=np.uint64)
        for hi, lo in raw.tolist():
            v = ((hi << 63) | lo) & mask
            if v < bound:
                out.append(v)
    return out[:count]


# ---- keys ---------------------------------------------------------------------------------------

class PrivateKey:
    """PrivateKey(params, rng) (src/fhe.jl:130-138): n random bits."""

    def __init__(self, params, rng):
        self.params = params
        self.key = rng.integers(0, 2, size=params.n, dtype=np.uint64)


class PublicKey:
    """PublicKey(rng, sk) (src/fhe.jl:146-168): (k0, k1 = k0 s + e) over Z_q[x]/(x^n + 1).
    The noise is centred on every coefficient, e_i in [-e_max, e_max] (the reference writes
    `polynomial - e_max`, a DarkIntegers Polynomial - scalar whose coefficient rule cannot be
    checked here; decryption holds either way)."""

    def __init__(self, rng, sk):
        p = sk.params
        if not p.q:
            raise AssertionError("synthetic parameter sets have no public-key modulus q")
        self.params = p
        self.k0 = rng.integers(0, p.q, size=p.n, dtype=np.uint64)               # fhe.jl:156
        quo, rem = divmod(p.Dq, 41 * p.n)                                        # fhe.jl:159-160
        e_max = quo - (rem == 0)
        e = rng.integers(0, 2 * e_max + 1, size=p.n).astype(np.int64) - e_max    # fhe.jl:161
        self.k1 = ((_negacyclic_mul_signed(self.k0, sk.key.astype(np.int64), p.q).astype(np.int64)
                    + e) % p.q).astype(np.uint64)                                # fhe.jl:163-164


class BootstrapKey:
    """BootstrapKey(rng, sk) (src/fhe.jl:176-201), resident on the GPU in the engine's form.

    `BootstrapKey(rng, sk)` generates the key on the device (seeded from `rng`);
    `BootstrapKey.from_canonical(params, residues)` uploads an existing key given as
    value.(coeffs) in [k][row][col][coef] order (what the Julia shim passes);
    `BootstrapKey(rng, sk, on_host=True)` generates it with host big-integer arithmetic."""

    def __init__(self, rng, sk, device=0, engine=None, on_host=False, random_flatten=False):
        """Both `rng = None` and `rng != None` calls work with every key (at Params(1024) the engine keeps
        the key in the five-prime form of the deterministic flatten and the six-prime form of the
        randomised one); random_flatten is accepted for compatibility and has no effect."""
        params = sk.params
        self.params = params
        self.engine = engine or Engine(params, device, random_flatten=random_flatten)
        if on_host:
            self.engine.upload_key(self._generate(rng, sk))
        else:
            # 256 bits from the caller's generator (pass a cryptographic one for real keys, e.g.
            # np.random.Generator over a CSPRNG bit generator, or use Engine.generate_key with
            # os.urandom(32)); the device expands them with ChaCha20
            self.engine.generate_key(sk.key, rng.bytes(32))

    @classmethod
    def from_canonical(cls, params, residues, device=0, engine=None):
        self = object.__new__(cls)
        self.params = params
        self.engine = engine or Engine(params, device)
        self.engine.upload_key(residues)
        return self

    @classmethod
    def from_rns2(cls, params, pairs, m1, m2, device=0, engine=None):
        """Key held as RNS2Number (v1, v2) limb pairs (src/rns.jl:8-24), Q = m1 * m2."""
        self = object.__new__(cls)
        self.params = params
        self.engine = engine or Engine(params, device)
        self.engine.upload_key_rns2(pairs, m1, m2)
        return self

    @staticmethod
    def _generate(rng, sk):
        p = sk.params
        Q, m, n = p.Q, p.m, p.n
        ext_key = [int(x) for x in sk.key] + [0] * (m - n)                # fhe.jl:185
        G = ((1, 0), (p.B, 0), (0, 1), (0, p.B))                          # fhe.jl:119-122
        out = np.zeros((n, 4, 2, m, 2), dtype=np.uint64)
        for k in range(n):
            for row in range(4):
                aj = _uniform_below(rng, Q, m)                            # fhe.jl:193
                ej = rng.integers(-n, n + 1, size=m)                      # fhe.jl:194
                bj = _negacyclic_mul_big(aj, ext_key, Q)                  # fhe.jl:195
                bj = [(x + int(e)) % Q for x, e in zip(bj, ej)]
                aj[0] = (aj[0] + ext_key[k] * G[row][0]) % Q              # fhe.jl:196
                bj[0] = (bj[0] + ext_key[k] * G[row][1]) % Q
                for col, poly in enumerate((aj, bj)):
                    out[k, row, col, :, 0] = [v & _M64 for v in poly]
                    out[k, row, col, :, 1] = [v >> 64 for v in poly]
        return out


# ---- ciphertexts --------------------------------------------------------------------------------

class LWE:
    """LWE{T} (src/fhe.jl:206-223): a batch-capable (a, b) pair over Z_r as uint64 arrays."""

    def __init__(self, a, b):
        self.a = np.ascontiguousarray(a, dtype=np.uint64)
        self.b = np.uint64(b)

    def __eq__(self, other):
        return np.array_equal(self.a, other.a) and self.b == other.b


class EncryptedBit:
    """EncryptedBit (src/fhe.jl:272-278)."""

    def __init__(self, lwe):
        self.lwe = lwe

    def __eq__(self, other):
        return self.lwe == other.lwe


class RLWE:
    def __init__(self, a, b):
        self.a, self.b = a, b


class PackedCiphertext:
    """PackedCiphertext (src/fhe.jl:252-255)."""

    def __init__(self, params, rlwe):
        self.params, self.rlwe = params, rlwe


class Ciphertext:
    """Ciphertext (src/fhe.jl:263-266): an RLWE over Z_r[x]/(x^m + 1) produced by
    pack_encrypted_bits; the message sits in the first n coefficients."""

    def __init__(self, params, rlwe):
        self.params, self.rlwe = params, rlwe


def packbits(bits):
    """packbits (src/utils.jl:36-42): a (t, n) bit array -> n integers, row i = bit i."""
    bits = np.asarray(bits, dtype=np.uint64)
    return (bits << np.arange(bits.shape[0], dtype=np.uint64)[:, None]).sum(axis=0, dtype=np.uint64)


def unpackbits(arr, itemsize):
    """unpackbits (src/utils.jl:48-54): n integers -> (itemsize, n) bit array."""
    arr = np.asarray(arr, dtype=np.uint64)
    return ((arr[None, :] >> np.arange(itemsize, dtype=np.uint64)[:, None]) & np.uint64(1)).astype(bool)


def prng_expand(seq, factor):
    """prng_expand (src/utils.jl:63-68): n seed bits -> n pseudo-random `factor`-bit integers,
    deterministically.  The reference seeds a MersenneTwister with hash(seq) and marks SHAKE as
    the intended primitive (utils.jl:64); this mirror uses SHAKE-256 of the packed seed bits."""
    import hashlib
    seq = np.asarray(seq, dtype=bool)
    n = len(seq)
    stream = hashlib.shake_256(np.packbits(seq).tobytes()).digest((factor * n + 7) // 8)
    bits = np.unpackbits(np.frombuffer(stream, dtype=np.uint8))[:factor * n].reshape(factor, n)
    return packbits(bits)


def deterministic_expand(params, u):
    """deterministic_expand (src/fhe.jl:304-307)."""
    return prng_expand(u, params.t + 1) & np.uint64(params.r - 1)


class PrivateEncryptedCiphertext:
    """PrivateEncryptedCiphertext (src/fhe.jl:297-301): 6 n bits for n message bits."""

    def __init__(self, params, u, v):
        self.params, self.u, self.v = params, u, v


def _encrypt_private(key, rng, message):
    """_encrypt_private (src/fhe.jl:310-328)."""
    p = key.params
    message = np.asarray(message, dtype=np.uint64)
    if len(message) != p.n:
        raise AssertionError("message must have length n (src/fhe.jl:313)")
    u = rng.integers(0, 2, size=p.n).astype(bool)                         # fhe.jl:315
    a = deterministic_expand(p, u)                                        # fhe.jl:316
    w_range = p.Dr // 8                                                   # fhe.jl:318
    w = rng.integers(-w_range, w_range + 1, size=p.n).astype(np.int64).astype(np.uint64)
    b = (_negacyclic_mul_small(a, key.key, p.r) + w + message * np.uint64(p.Dr)) & np.uint64(p.r - 1)
    sh = np.uint64(p.t - 4)
    b = (b >> sh) << sh                                                   # fhe.jl:325
    return u, RLWE(a, b)


class PublicEncryptedCiphertext:
    """PublicEncryptedCiphertext (src/fhe.jl:380-384): (t + 1) + 6 bits per message bit."""

    def __init__(self, params, a_bits, b_bits):
        self.params, self.a_bits, self.b_bits = params, a_bits, b_bits


def _encrypt_public(key, rng, message):
    """_encrypt_public (src/fhe.jl:386-409)."""
    p = key.params
    message = np.asarray(message, dtype=np.int64)
    if len(message) != p.n:
        raise AssertionError("message must have length n")
    u = rng.integers(-1, 2, size=p.n).astype(np.int64)                    # fhe.jl:390
    w1_max = p.Dq // (41 * p.n)                                           # fhe.jl:392
    w1 = rng.integers(-w1_max, w1_max + 1, size=p.n).astype(np.int64)
    w2_max = p.Dq // 82                                                   # fhe.jl:395
    w2 = rng.integers(-w2_max, w2_max + 1, size=p.n).astype(np.int64)
    a1 = (_negacyclic_mul_signed(key.k0, u, p.q).astype(np.int64) + w1) % p.q
    a2 = (_negacyclic_mul_signed(key.k1, u, p.q).astype(np.int64) + w2 + message * p.Dq) % p.q
    a = _rescale(p.r, a1, p.q, True)                                      # fhe.jl:402
    shift = p.t - 5
    if p.r % (1 << shift):
        raise AssertionError("r must be a multiple of 2^(t - 5) (src/fhe.jl:404)")
    b = _rescale(p.r >> shift, a2, p.q, False) << np.uint64(shift)        # fhe.jl:405-406
    return RLWE(a, b & np.uint64(p.r - 1))


def encrypt(key, rng, message):
    """encrypt(key::PrivateKey, rng, message) (src/fhe.jl:369-372) and
    encrypt(key::PublicKey, rng, message) (src/fhe.jl:457-459)."""
    if isinstance(key, PublicKey):
        return PackedCiphertext(key.params, _encrypt_public(key, rng, message))
    u, rlwe = _encrypt_private(key, rng, message)
    return PackedCiphertext(key.params, rlwe)


def encrypt_optimal(key, rng, message):
    """encrypt_optimal(key::PrivateKey, rng, message) (src/fhe.jl:339-345): 6 bits per message
    bit; encrypt_optimal(key::PublicKey, ...) (src/fhe.jl:420-436): t + 7 bits per message bit."""
    p = key.params
    if isinstance(key, PublicKey):
        rlwe = _encrypt_public(key, rng, message)
        a_bits = unpackbits(rlwe.a, p.t + 1)                              # fhe.jl:429
        b_bits = unpackbits(rlwe.b >> np.uint64(p.t - 5), 6)              # fhe.jl:431-432
        return PublicEncryptedCiphertext(p, a_bits, b_bits)
    u, rlwe = _encrypt_private(key, rng, message)
    b_packed = rlwe.b >> np.uint64(p.t - 4)                               # fhe.jl:342
    return PrivateEncryptedCiphertext(p, u, unpackbits(b_packed, 5))


def normalize_ciphertext(ct):
    """normalize_ciphertext(::PrivateEncryptedCiphertext) (src/fhe.jl:354-359) and
    normalize_ciphertext(::PublicEncryptedCiphertext) (src/fhe.jl:445-450)."""
    p = ct.params
    if isinstance(ct, PublicEncryptedCiphertext):
        a = packbits(ct.a_bits) & np.uint64(p.r - 1)
        b = (packbits(ct.b_bits) << np.uint64(p.t - 5)) & np.uint64(p.r - 1)
        return PackedCiphertext(p, RLWE(a, b))
    a = deterministic_expand(p, ct.u)
    b = (packbits(ct.v) << np.uint64(p.t - 4)) & np.uint64(p.r - 1)
    return PackedCiphertext(p, RLWE(a, b))


def extract(a, i, n):
    """extract(a, i, n) (src/fhe.jl:237-244) with the reference's 1-based i; a over Z_r."""
    a = np.asarray(a, dtype=np.uint64)
    N = len(a)
    if i > N:
        raise AssertionError("i <= N required (src/fhe.jl:238)")
    if i < n:
        head = a[np.arange(i - 1, -1, -1)]
        tail = np.uint64(0) - a[np.arange(N - 1, N - 1 - (n - i), -1)]
        return np.concatenate([head, tail])
    return a[np.arange(i - 1, i - 1 - n, -1)]


def split_ciphertext(ct):
    """split_ciphertext (src/fhe.jl:287-290) for PackedCiphertext and Ciphertext."""
    p = ct.params
    mask = np.uint64(p.r - 1)
    return [EncryptedBit(LWE(extract(ct.rlwe.a, i, p.n) & mask, ct.rlwe.b[i - 1]))
            for i in range(1, p.n + 1)]


def pack_encrypted_bits(bkey, rng, enc_bits):
    """pack_encrypted_bits(bkey, rng, enc_bits) (src/fhe.jl:660-696): n EncryptedBits -> one
    RLWE Ciphertext, on the HIP engine.  rng = None: deterministic flatten (bit-exact); a numpy
    Generator: randomised flatten of the n bootstraps and of the shortened external products."""
    p = bkey.params
    if len(enc_bits) != p.n:
        raise AssertionError("exactly n encrypted bits are required (src/fhe.jl:667)")
    a = np.stack([e.lwe.a for e in enc_bits])[None, :, :]
    b = np.array([e.lwe.b for e in enc_bits], dtype=np.uint64)[None, :]
    with bkey.engine.lock:                       # mode and call stay together (threads sharing a key)
        _set_flatten_mode(bkey, rng)
        w, v = bkey.engine.pack_encrypted_bits(a, b)
    return Ciphertext(p, RLWE(w[0], v[0]))


def decrypt(key, ct):
    """decrypt(key, ::EncryptedBit) (src/fhe.jl:504-507) and
    decrypt(key, ::Union{Ciphertext, PackedCiphertext}) (src/fhe.jl:471-494)."""
    p = key.params
    mask = np.uint64(p.r - 1)
    if isinstance(ct, EncryptedBit):
        b1 = (int(ct.lwe.b) - int(np.sum(ct.lwe.a * key.key, dtype=np.uint64))) % p.r
        return bool(((b1 + p.Dr // 2) % p.r) // p.Dr)
    key_poly = key.key
    if isinstance(ct, Ciphertext):                                        # fhe.jl:474-478
        key_poly = np.concatenate([key.key, np.zeros(p.m - p.n, dtype=np.uint64)])
    b1 = (ct.rlwe.b - _negacyclic_mul_small(ct.rlwe.a, key_poly, p.r)) & mask
    if isinstance(ct, Ciphertext):
        b1 = b1[:p.n]                                                     # fhe.jl:481-482
    return (((b1 + np.uint64(p.Dr // 2)) & mask) // np.uint64(p.Dr)).astype(bool)


# ---- bootstrap -----------------------------------------------------------------------------------

def _set_flatten_mode(bkey, rng):
    """rng = None: deterministic flatten (bit-exact with the reference's `rng = nothing`); a numpy
    Generator: randomised flatten on the device, its ChaCha8 draw stream keyed with 32 bytes of it."""
    if rng is None:
        bkey.engine.set_random_flatten(False)
    else:
        bkey.engine.set_random_flatten(True, rng.bytes(32))


def bootstrap(bkey, rng, enc_bit1, enc_bit2):
    """bootstrap(bkey, rng, enc_bit1, enc_bit2) (src/fhe.jl:608-621): returns EncryptedBits of
    AND, OR, XOR.  A batch of 1 through the HIP engine."""
    with bkey.engine.lock:                       # mode and call stay together (threads sharing a key)
        _set_flatten_mode(bkey, rng)
        out = bkey.engine.bootstrap_batch(enc_bit1.lwe.a[None, :], [enc_bit1.lwe.b],
                                          enc_bit2.lwe.a[None, :], [enc_bit2.lwe.b])
    n = bkey.params.n
    return tuple(EncryptedBit(LWE(out[0, g, :n], out[0, g, n])) for g in range(3))


def bootstrap_batch(bkey, rng, enc_bits1, enc_bits2):
    """Batched form: two equally long lists of EncryptedBit -> list of (AND, OR, XOR) triples."""
    if len(enc_bits1) != len(enc_bits2):
        raise ValueError("ragged batch")
    n = bkey.params.n
    if not enc_bits1:
        return []
    a1 = np.stack([e.lwe.a for e in enc_bits1])
    a2 = np.stack([e.lwe.a for e in enc_bits2])
    b1 = np.array([e.lwe.b for e in enc_bits1], dtype=np.uint64)
    b2 = np.array([e.lwe.b for e in enc_bits2], dtype=np.uint64)
    with bkey.engine.lock:
        _set_flatten_mode(bkey, rng)
        out = bkey.engine.bootstrap_batch(a1, b1, a2, b2)
    return [tuple(EncryptedBit(LWE(out[t, g, :n], out[t, g, n])) for g in range(3))
            for t in range(out.shape[0])]
