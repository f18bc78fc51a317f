"""Host-side mirror of the SGFHE.jl API (sgfhe.jl_amd/): parameters, ciphertext plumbing, and
the C-ABI library's exported symbols.  No GPU compute."""

import ctypes
import os
import re

import numpy as np
import pytest

import bigint_oracle as BO

TABLE_P = {  # SURVEY.md section 8, Table P (derived from src/fhe.jl:43-97)
    64: dict(r=1024, m=512, t=9, q=65537, Q=5494391545392009217, B=2348810240, Dr=256,
             DQ_tilde=686798943174001152),
    512: dict(r=8192, m=4096, t=12, q=4205569, Q=1440321777275241790332929, B=1202590842880,
              Dr=2048, DQ_tilde=180040222159405223791616),
    1024: dict(r=16384, m=8192, t=13, q=16801793, Q=92180593745615474572738561, B=9620726743040,
               Dr=4096, DQ_tilde=11522574218201934321592320),
}


@pytest.mark.parametrize("n", [64, 512, 1024])
def test_params_table(S, oc, n):
    p = S.Params(n)
    for k, v in TABLE_P[n].items():
        assert getattr(p, k) == v, k
    assert p.ell == 2 and p.Dq == p.q // 4
    o = BO.Params.make(n)
    assert (o.Q, o.q, o.B, o.DQ_tilde) == (p.Q, p.q, p.B, p.DQ_tilde)
    d = oc.params_make(n)
    assert (d["Q"], d["B"], d["DQ_tilde"], d["m"], d["r"]) == (p.Q, p.B, p.DQ_tilde, p.m, p.r)
    assert S.isprime(p.Q) and (p.Q - 1) % (2 * p.m) == 0 and p.B * p.B >= p.Q


def test_params_rejects_bad_n(S):
    for n in (32, 96, 0):
        with pytest.raises(AssertionError):
            S.Params(n)
    with pytest.raises(AssertionError):
        S.Params.custom(8, 1 << 40, 1 << 10)      # B^2 < Q


def test_find_modulus(S):
    assert S.find_modulus(128, 1024 * 64) == 65537
    with pytest.raises(ValueError):
        S.find_modulus(1 << 20, 10, 20)


def test_encrypt_split_decrypt_roundtrip(S):
    """test/api.test.jl:33-42 (split_ciphertext + per-bit decrypt) and :8-17 style round trip."""
    p = S.Params(64)
    rng = np.random.default_rng(0)
    sk = S.PrivateKey(p, rng)
    msg = rng.integers(0, 2, size=p.n).astype(bool)
    ct = S.encrypt(sk, rng, msg)
    assert np.array_equal(S.decrypt(sk, ct), msg)
    bits = S.split_ciphertext(ct)
    assert len(bits) == p.n
    assert [S.decrypt(sk, b) for b in bits] == list(msg)
    with pytest.raises(AssertionError):
        S.encrypt(sk, rng, msg[:-1])


def test_encrypt_optimal_normalize_decrypt(S):
    """test/api.test.jl:8-17: encrypt_optimal -> normalize_ciphertext -> decrypt, Params(512)."""
    p = S.Params(512)
    rng = np.random.default_rng(1)
    sk = S.PrivateKey(p, rng)
    msg = rng.integers(0, 2, size=p.n).astype(bool)
    ct = S.encrypt_optimal(sk, rng, msg)
    assert ct.u.shape == (p.n,) and ct.v.shape == (5, p.n)          # 6 n bits in total
    assert np.array_equal(S.decrypt(sk, S.normalize_ciphertext(ct)), msg)


@pytest.mark.parametrize("n", [64, 512])
def test_public_key_encryption(S, n):
    """test/api.test.jl:20-30: encrypt_optimal with a PublicKey -> normalize -> decrypt with the
    private key; also encrypt(::PublicKey) -> PackedCiphertext -> split -> per-bit decrypt."""
    rng = np.random.default_rng(5 + n)
    params = S.Params(n)
    key = S.PrivateKey(params, rng)
    pkey = S.PublicKey(rng, key)
    for trial in range(4):
        message = rng.integers(0, 2, size=params.n).astype(bool)
        ct = S.encrypt_optimal(pkey, rng, message)
        assert ct.a_bits.shape == (params.t + 1, params.n) and ct.b_bits.shape == (6, params.n)
        assert np.array_equal(S.decrypt(key, S.normalize_ciphertext(ct)), message)
        packed = S.encrypt(pkey, rng, message)
        assert np.array_equal(S.decrypt(key, packed), message)
    bits = S.split_ciphertext(packed)
    assert [S.decrypt(key, eb) for eb in bits[:16]] == list(message[:16])


def test_packbits_roundtrip_and_expand_determinism(S):
    rng = np.random.default_rng(2)
    vals = rng.integers(0, 1 << 13, size=100, dtype=np.uint64)
    assert np.array_equal(S.packbits(S.unpackbits(vals, 13)), vals)
    u = rng.integers(0, 2, size=64).astype(bool)
    a1, a2 = S.prng_expand(u, 10), S.prng_expand(u.copy(), 10)
    assert np.array_equal(a1, a2) and int(a1.max()) < 1 << 10
    u2 = u.copy()
    u2[3] ^= True
    assert not np.array_equal(S.prng_expand(u2, 10), a1)


def test_extract_matches_reference_semantics(S):
    """src/fhe.jl:237-244 against the oracle's literal restatement."""
    Q = 1 << 10
    a = list(range(1, 17))
    for n in (4, 16):
        for i in range(1, 17):
            got = [int(v) % Q for v in S.extract(np.array(a, dtype=np.uint64), i, n)]
            assert got == BO.extract(a, i, n, Q)


def test_params_keyword_arguments_of_the_reference(S):
    """Params(n; rlwe_type, mod_repr) (src/fhe.jl:43-47,71-85): the keywords choose how the reference
    stores residues, not their values -- same numbers, the reference's assertions (a type too narrow
    for Q, an unknown representation), the reference's defaults (UInt64 / UInt128 by the size of Q,
    MgModUInt).  test/performance.test.jl:32,59,86 build Params(64) with MLUInt and both
    representations."""
    base = S.Params(64)
    assert (base.rlwe_type, base.mod_repr) == ("UInt64", "MgModUInt")
    assert S.Params(512).rlwe_type == "UInt128"
    for kw in (dict(rlwe_type="UInt128"), dict(rlwe_type="MLUInt{2, UInt64}"), dict(rlwe_type=128),
               dict(mod_repr="ModUInt"), dict(mod_repr="MgModUInt", rlwe_type="MLUInt{4, UInt32}")):
        p = S.Params(64, **kw)
        assert p == base and hash(p) == hash(base)             # values, not representation
        assert p.rlwe_type == kw.get("rlwe_type", "UInt64") and p.mod_repr == kw.get("mod_repr", "MgModUInt")
    with pytest.raises(AssertionError):
        S.Params(512, rlwe_type="UInt64")                      # sizeof(rlwe_type) * 8 > log2(Q), src/fhe.jl:80
    with pytest.raises(AssertionError):
        S.Params(64, rlwe_type="MLUInt{1, UInt32}")
    with pytest.raises(AssertionError):
        S.Params(64, mod_repr="Montgomery")                    # src/fhe.jl:47
    with pytest.raises(AssertionError):
        S.Params(64, rlwe_type="Float64")


def test_pack_selects_flatten_mode_and_checks_length(S):
    """pack_encrypted_bits(bkey, rng, enc_bits): a wrong number of bits is the reference's assertion
    (src/fhe.jl:667), raised before the engine is touched; otherwise rng picks the flatten mode on
    the engine (as for bootstrap) and the call follows it under the engine's lock, so that threads
    sharing a key cannot interleave between the two."""
    import threading
    calls = []
    p = S.Params(64)

    class FakeEngine:
        lock = threading.RLock()

        def set_random_flatten(self, enable, seed=0):
            assert self.lock._is_owned()
            calls.append(("mode", bool(enable), seed))

        def pack_encrypted_bits(self, a, b):
            assert self.lock._is_owned()
            calls.append(("pack", a.shape, b.shape))
            return np.zeros((1, p.m), dtype=np.uint64), np.zeros((1, p.m), dtype=np.uint64)

        def bootstrap_batch(self, a1, b1, a2, b2):
            assert self.lock._is_owned()
            calls.append(("bootstrap", np.asarray(a1).shape))
            return np.zeros((np.asarray(a1).shape[0], 3, p.n + 1), dtype=np.uint64)

    class Dummy:
        params = p
        engine = FakeEngine()
    with pytest.raises(AssertionError):
        S.pack_encrypted_bits(Dummy(), np.random.default_rng(0), [])
    with pytest.raises(AssertionError):
        S.pack_encrypted_bits(Dummy(), None, [])
    assert calls == []
    bit = S.EncryptedBit(S.LWE(np.zeros(p.n, dtype=np.uint64), np.uint64(0)))
    S.pack_encrypted_bits(Dummy(), np.random.default_rng(0), [bit] * p.n)
    S.pack_encrypted_bits(Dummy(), None, [bit] * p.n)
    S.bootstrap(Dummy(), None, bit, bit)
    S.bootstrap_batch(Dummy(), np.random.default_rng(1), [bit, bit], [bit, bit])
    kinds = [c[0] for c in calls]
    assert kinds == ["mode", "pack", "mode", "pack", "mode", "bootstrap", "mode", "bootstrap"]
    assert calls[0][1] is True and len(calls[0][2]) == 32 and calls[2][1:] == (False, 0)
    assert calls[4][1] is False and calls[6][1] is True


def test_library_exports_every_declared_symbol(S):
    """The C-ABI library loads and exports exactly what include/sgfhe_hip.h declares."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "sgfhe_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(sgfhe_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(S.EXPORTED_SYMBOLS)
    L = ctypes.CDLL(S.build())
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in S.lib().sgfhe_version()


def test_ctx_create_argument_errors(S):
    """Status codes without touching a device: NULL arguments."""
    L = S.lib()
    assert L.sgfhe_ctx_create(None, 0, None) == -1
    assert L.sgfhe_bootstrap_batch(None, None, None, None, None, 0, None, 0) == -1
    assert L.sgfhe_ctx_destroy(None) == 0


def test_no_gpu_fails_loudly(S):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(S.SgfheError) as ei:
        S.Engine(S.Params(64))
    assert ei.value.code == -3
