"""SURVEY.md section 8f, row N3 as host C++ behind the C ABI (`sgfhe_host_*`, csrc/host_plumbing.h)
against the numpy mirror of the same reference functions (sgfhe.jl_amd/scheme.py), bit for bit, and
against the reference's own properties (test/api.test.jl:8-42: encrypt -> decrypt, split -> per-bit
decrypt, optimal -> normalize -> decrypt).  No GPU."""

import hashlib

import numpy as np
import pytest


class _Fixed:
    """A stand-in generator that replays given draws: `_encrypt_private` of scheme.py asks for
    u = integers(0, 2, n) and then w = integers(-w_range, w_range + 1, n)."""

    def __init__(self, u, w):
        self.q = [np.asarray(u), np.asarray(w)]

    def integers(self, lo, hi=None, size=None, dtype=None):
        return self.q.pop(0)


@pytest.mark.parametrize("n", [64, 512])
def test_private_encryption_chain_equals_numpy_mirror(S, n):
    p = S.Params(n)
    rng = np.random.default_rng(n)
    sk = S.PrivateKey(p, rng)
    msg = rng.integers(0, 2, size=n).astype(bool)
    u = rng.integers(0, 2, size=n)
    w = rng.integers(-(p.Dr // 8), p.Dr // 8 + 1, size=n)
    # deterministic_expand: SHAKE-256 of the packed seed bits
    assert np.array_equal(S.host.deterministic_expand(p, u), S.scheme.deterministic_expand(p, u.astype(bool)))
    # _encrypt_private with the same draws
    u_ref, rlwe = S.scheme._encrypt_private(sk, _Fixed(u, w), msg)
    a, b = S.host.encrypt_private(p, sk.key, u, w, msg)
    assert np.array_equal(a, rlwe.a) and np.array_equal(b, rlwe.b)
    # decrypt(::PackedCiphertext), split_ciphertext + decrypt(::EncryptedBit)
    assert np.array_equal(S.host.decrypt_rlwe(p, sk.key, a, b), msg)
    la, lb = S.host.split_ciphertext(p, a, b)
    bits = S.split_ciphertext(S.PackedCiphertext(p, rlwe))
    assert np.array_equal(la, np.stack([e.lwe.a for e in bits]))
    assert np.array_equal(lb, np.array([e.lwe.b for e in bits], dtype=np.uint64))
    assert np.array_equal(S.host.decrypt_lwe(p, sk.key, la, lb), msg)
    assert [S.decrypt(sk, e) for e in bits] == list(msg)
    # encrypt_optimal -> normalize_ciphertext (6 bits per message bit)
    v = S.host.pack_private(p, b)
    assert np.array_equal(v.astype(bool), S.unpackbits(rlwe.b >> np.uint64(p.t - 4), 5))
    na, nb = S.host.normalize_private(p, u, v)
    ref = S.normalize_ciphertext(S.PrivateEncryptedCiphertext(p, u.astype(bool), v.astype(bool)))
    assert np.array_equal(na, ref.rlwe.a) and np.array_equal(nb, ref.rlwe.b)
    assert np.array_equal(na, a) and np.array_equal(nb, b)       # b keeps only its 5 high bits
    assert np.array_equal(S.host.decrypt_rlwe(p, sk.key, na, nb), msg)


def test_split_of_a_packed_ciphertext_of_length_m(S):
    """split_ciphertext / decrypt of a `Ciphertext` (length m, the output of pack_encrypted_bits):
    extract takes its i < n branch with the negated tail (src/fhe.jl:239-241)."""
    p = S.Params(64)
    rng = np.random.default_rng(5)
    sk = S.PrivateKey(p, rng)
    a = rng.integers(0, p.r, size=p.m, dtype=np.uint64)
    key_poly = np.concatenate([sk.key, np.zeros(p.m - p.n, dtype=np.uint64)])
    msg = rng.integers(0, 2, size=p.n).astype(np.uint64)
    b = S.scheme._negacyclic_mul_small(a, key_poly, p.r)
    b[:p.n] = (b[:p.n] + msg * np.uint64(p.Dr) + rng.integers(0, p.Dr // 8, size=p.n, dtype=np.uint64)) & np.uint64(p.r - 1)
    ct = S.Ciphertext(p, S.RLWE(a, b))
    assert np.array_equal(S.host.decrypt_rlwe(p, sk.key, a, b), S.decrypt(sk, ct))
    assert np.array_equal(S.decrypt(sk, ct), msg.astype(bool))
    la, lb = S.host.split_ciphertext(p, a, b)
    bits = S.split_ciphertext(ct)
    assert np.array_equal(la, np.stack([e.lwe.a for e in bits]))
    assert np.array_equal(S.host.decrypt_lwe(p, sk.key, la, lb), msg.astype(bool))
    for i in (1, 2, p.n - 1, p.n, p.m):                   # extract on both branches (1-based i)
        if i <= p.n:
            assert np.array_equal(la[i - 1], S.extract(a, i, p.n) & np.uint64(p.r - 1))


def test_shake256_known_answers(S):
    """The SHAKE-256 under prng_expand against hashlib on lengths around the 136-byte rate."""
    import ctypes
    p = S.Params(2048)                                   # n = 2048 seed bits = 256 bytes: two blocks
    for seed in range(3):
        u = np.random.default_rng(seed).integers(0, 2, size=p.n).astype(np.uint8)
        got = S.host.deterministic_expand(p, u)
        stream = hashlib.shake_256(np.packbits(u.astype(bool)).tobytes()).digest(((p.t + 1) * p.n + 7) // 8)
        bits = np.unpackbits(np.frombuffer(stream, dtype=np.uint8))[:(p.t + 1) * p.n].reshape(p.t + 1, p.n)
        assert np.array_equal(got, S.packbits(bits) & np.uint64(p.r - 1))


def test_bad_arguments_are_refused(S):
    p = S.Params(64)
    z = np.zeros(p.n, dtype=np.uint64)
    with pytest.raises(ValueError):                       # noise outside [-Dr/8, Dr/8]
        S.host.encrypt_private(p, z, z, np.full(p.n, p.Dr // 8 + 1), z)
    with pytest.raises(ValueError):                       # neither n nor m coefficients
        S.host.split_ciphertext(p, np.zeros(100, dtype=np.uint64), np.zeros(100, dtype=np.uint64))


class _Replay:
    """Replays given draws in the order scheme.py asks for them."""

    def __init__(self, *draws):
        self.q = [np.asarray(d) for d in draws]

    def integers(self, lo, hi=None, size=None, dtype=None):
        return self.q.pop(0)


@pytest.mark.parametrize("n", [64, 512, 1024])
def test_public_key_chain_equals_numpy_mirror(S, n):
    """Row N4's host side: PublicKey, _encrypt_public, encrypt_optimal(::PublicKey) and its
    normalisation behind the C ABI against the numpy mirror on the same draws, bit for bit, and
    the reference's properties (test/api.test.jl:26-42: public encrypt -> decrypt; optimal ->
    normalize -> decrypt), with the draws at their extremes as well."""
    p = S.Params(n)
    rng = np.random.default_rng(100 + n)
    sk = S.PrivateKey(p, rng)
    quo, rem = divmod(p.Dq, 41 * n)
    e_max = quo - (rem == 0)
    w1_max, w2_max = p.Dq // (41 * n), p.Dq // 82
    for extreme in (False, True):
        k0 = rng.integers(0, p.q, size=n, dtype=np.uint64)
        if extreme:
            e = rng.choice([-e_max, e_max], size=n)
            u = rng.choice([-1, 1], size=n)
            w1 = rng.choice([-w1_max, w1_max], size=n)
            w2 = rng.choice([-w2_max, w2_max], size=n)
            k0[:4] = [0, p.q - 1, 1, p.q // 2]
        else:
            e = rng.integers(-e_max, e_max + 1, size=n)
            u = rng.integers(-1, 2, size=n)
            w1 = rng.integers(-w1_max, w1_max + 1, size=n)
            w2 = rng.integers(-w2_max, w2_max + 1, size=n)
        msg = rng.integers(0, 2, size=n).astype(bool)
        pk = S.PublicKey(_Replay(k0, e + e_max), sk)        # the mirror draws 0 .. 2 e_max and centres
        k1 = S.host.public_key(p, sk.key, k0, e)
        assert np.array_equal(k1, pk.k1) and np.array_equal(k0, pk.k0)
        rlwe = S.scheme._encrypt_public(pk, _Replay(u, w1, w2), msg)
        a, b = S.host.encrypt_public(p, k0, k1, u, w1, w2, msg)
        assert np.array_equal(a, rlwe.a) and np.array_equal(b, rlwe.b)
        assert np.array_equal(S.host.decrypt_rlwe(p, sk.key, a, b), msg)
        assert np.array_equal(S.decrypt(sk, S.PackedCiphertext(p, rlwe)), msg)
        ab, bb = S.host.pack_public(p, a, b)
        ct = S.encrypt_optimal(pk, _Replay(u, w1, w2), msg)
        assert np.array_equal(ab.astype(bool), ct.a_bits) and np.array_equal(bb.astype(bool), ct.b_bits)
        assert ab.shape == (p.t + 1, n) and bb.shape == (6, n)
        na, nb = S.host.normalize_public(p, ab, bb)
        ref = S.normalize_ciphertext(ct)
        assert np.array_equal(na, ref.rlwe.a) and np.array_equal(nb, ref.rlwe.b)
        assert np.array_equal(na, a) and np.array_equal(nb, b)   # b carries only its 6 high bits
        la, lb = S.host.split_ciphertext(p, na, nb)
        assert np.array_equal(S.host.decrypt_lwe(p, sk.key, la, lb), msg)


def test_public_key_side_refuses_malformed_input(S):
    import ctypes
    p = S.Params(64)
    rng = np.random.default_rng(3)
    sk = S.PrivateKey(p, rng)
    k0 = rng.integers(0, p.q, size=p.n, dtype=np.uint64)
    e = np.zeros(p.n, dtype=np.int64)
    k1 = S.host.public_key(p, sk.key, k0, e)
    bad = e.copy()
    bad[5] = p.Dq // (41 * p.n) + 1
    with pytest.raises(Exception):
        S.host.public_key(p, sk.key, k0, bad)                 # noise outside [-e_max, e_max]
    k0b = k0.copy()
    k0b[0] = p.q
    with pytest.raises(Exception):
        S.host.public_key(p, sk.key, k0b, e)                  # residue outside [0, q)
    u = np.zeros(p.n, dtype=np.int8)
    z = np.zeros(p.n, dtype=np.int64)
    ub = u.copy()
    ub[1] = 2
    with pytest.raises(Exception):
        S.host.encrypt_public(p, k0, k1, ub, z, z, np.zeros(p.n, dtype=np.uint8))
    q_saved = p.q
    try:
        p.q = q_saved + 1                                     # not congruent to 1 modulo 2 n
        with pytest.raises(Exception):
            S.host.public_key(p, sk.key, k0, e)
    finally:
        p.q = q_saved
