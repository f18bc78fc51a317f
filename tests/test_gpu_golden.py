"""HIP engine against the committed golden vectors (tests/golden/, big-integer oracle) and the
edge cases of the boundary.  Run on the GPU box with `pytest -m gpu`."""

import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def h_ints(vals, nbytes=16):
    h = hashlib.sha256()
    for v in vals:
        h.update(int(v).to_bytes(nbytes, "little"))
    return h.hexdigest()


def _u128_ints(arr):
    flat = np.ascontiguousarray(arr).reshape(-1, 2)
    return [int(lo) | (int(hi) << 64) for lo, hi in flat]


def test_engine_matches_random_flatten_golden(S, oc):
    """golden/p64rnd.json: two bootstraps in ONE call with the randomised flatten on the ChaCha8
    draw stream of a 32-byte key -- accumulators after 1, 2 and 64 iterations, raw and ModRed
    outputs equal the committed big-integer results."""
    d = json.load(open(os.path.join(G, "p64rnd.json")))
    params = S.Params(64)
    o = oc.Oracle.from_params(params)
    sk = o.private_key(d["sk_seed"])
    bkey = o.bootstrap_key(sk, d["key_seed"])
    assert hashlib.sha256(np.ascontiguousarray(bkey).tobytes()).hexdigest() == d["key_sha256"]
    fkey = bytes.fromhex(d["flatten_key_hex"])
    a1 = np.array([c["lwe1"]["a"] for c in d["cases"]], dtype=np.uint64)
    a2 = np.array([c["lwe2"]["a"] for c in d["cases"]], dtype=np.uint64)
    b1, b2 = [c["lwe1"]["b"] for c in d["cases"]], [c["lwe2"]["b"] for c in d["cases"]]
    eng = S.Engine(params)
    try:
        eng.upload_key(bkey)
        for k in ("1", "2", "64"):
            eng.set_random_flatten(True, fkey)                      # call number back to 0
            acc = eng.debug_accumulators(a1, b1, a2, b2, int(k))
            for j, case in enumerate(d["cases"]):
                assert [h_ints(_u128_ints(acc[j, 0])), h_ints(_u128_ints(acc[j, 1]))] == case["acc_sha256_after"][k]
        eng.set_random_flatten(True, fkey)
        raw = eng.bootstrap_batch(a1, b1, a2, b2, raw=True)
        eng.set_random_flatten(True, fkey)
        out = eng.bootstrap_batch(a1, b1, a2, b2)
        for j, case in enumerate(d["cases"]):
            for g in range(3):
                assert h_ints(_u128_ints(raw[j, g])) == case["raw_sha256"][g]
                assert [int(v) for v in out[j, g]] == case["out"][g]
        # the 64-bit short form is the same key with zero upper bytes, not this key
        eng.set_random_flatten(True, int.from_bytes(fkey[:8], "little"))
        assert not np.array_equal(eng.bootstrap_batch(a1, b1, a2, b2), out)
        eng.set_random_flatten(True, int.from_bytes(fkey[:8], "little"))
        short = eng.bootstrap_batch(a1, b1, a2, b2)
        assert eng._L.sgfhe_set_random_flatten(eng._h, 1, int.from_bytes(fkey[:8], "little")) == 0   # the C short form
        assert np.array_equal(eng.bootstrap_batch(a1, b1, a2, b2), short)
        with pytest.raises(ValueError):
            eng.set_random_flatten(True, b"short")
    finally:
        eng.close()


@pytest.mark.parametrize("name", ["p64", "p512", "p1024"])
def test_engine_matches_golden(S, oc, name):
    path = os.path.join(G, name + ".json")
    if not os.path.exists(path):
        pytest.skip("golden/%s.json not generated" % name)
    d = json.load(open(path))
    n = d["params"]["n"]
    params = S.Params(n)
    assert str(params.Q) == d["params"]["Q"]
    o = oc.Oracle.from_params(params)                 # key regenerated from its seed
    sk = o.private_key(d["sk_seed"])
    bkey = o.bootstrap_key(sk, d["key_seed"])
    assert hashlib.sha256(np.ascontiguousarray(bkey).tobytes()).hexdigest() == d["key_sha256"]
    eng = S.Engine(params)
    eng.upload_key(bkey)
    for case in d["cases"]:
        a1, b1 = np.array([case["lwe1"]["a"]], dtype=np.uint64), [case["lwe1"]["b"]]
        a2, b2 = np.array([case["lwe2"]["a"]], dtype=np.uint64), [case["lwe2"]["b"]]
        for k, (ha, hb) in case["acc_sha256_after"].items():
            acc = eng.debug_accumulators(a1, b1, a2, b2, int(k))
            assert h_ints(_u128_ints(acc[0, 0])) == ha, "acc_a after %s iterations" % k
            assert h_ints(_u128_ints(acc[0, 1])) == hb, "acc_b after %s iterations" % k
        raw = eng.bootstrap_batch(a1, b1, a2, b2, raw=True)
        out = eng.bootstrap_batch(a1, b1, a2, b2)
        for g in range(3):
            assert h_ints(_u128_ints(raw[0, g])) == case["raw_sha256"][g]
            if "out" in case:
                assert [int(v) for v in out[0, g]] == case["out"][g]
            else:
                assert h_ints([int(v) for v in out[0, g]], 8) == case["out_sha256"][g]
    eng.close()


def test_external_product_golden(S):
    d = json.load(open(os.path.join(G, "extprod.json")))
    Q, B, m = int(d["Q"]), int(d["B"]), d["m"]
    eng = S.Engine(S.Params.custom(d["n"], Q, B))

    def u(vals):
        a = np.zeros((len(vals), 2), dtype=np.uint64)
        a[:, 0] = [v & 0xFFFFFFFFFFFFFFFF for v in vals]
        a[:, 1] = [v >> 64 for v in vals]
        return a
    A = np.stack([np.stack([u(d["A"][r][c]) for c in range(2)]) for r in range(4)])
    ra, rb = eng.external_product(u(d["a"]), u(d["b"]), A)
    assert _u128_ints(ra) == d["a_res"] and _u128_ints(rb) == d["b_res"]
    eng.close()


@pytest.mark.parametrize("B", [4, 5, 6, 7])
def test_flatten_kernel_exhaustive_small_moduli(S, B):
    """k_flatten_canon on its own against the exhaustive tables of test/internals.test.jl:69-112
    (deterministic branch, l = 2: the engine's decomposition length): every a in Z_q, q = B^2 - 1,
    restores (sum(b .* B.^(0:l-1)) == a) and lands in the limits of decomposition_limits
    (:48-66); B in {4, 5} also equal the committed big-integer tables (tests/golden/tables.json)."""
    import bigint_oracle as BO
    q = B * B - 1
    params = S.Params.custom(8, q, B)
    eng = S.Engine(params)
    m = params.m
    vals = np.zeros((2, m, 2), dtype=np.uint64)
    flat = np.arange(2 * m, dtype=np.uint64) % q          # every residue, several times over
    vals[:, :, 0] = flat.reshape(2, m)
    dig = eng.debug_flatten(vals)
    s = B // 2 - 1 if B % 2 == 0 else (B - 1) // 2
    lim_lo, lim_hi = q - s, B - s - 1
    table = None
    for t in json.load(open(os.path.join(G, "tables.json")))["flatten"]:
        if t["B"] == B and t["ell"] == 2:
            table = t["values"]
    for c in range(2):
        for j in range(m):
            a = int(vals[c, j, 0])
            d = [(int(dig[c, i, j]) - s) % q for i in range(2)]       # stored e_i = u_i + s
            assert d == BO.flatten(a, B, 2, q)
            assert (d[0] + d[1] * B) % q == a
            assert all(x <= lim_hi or x >= lim_lo for x in d)
            if table is not None:
                assert d == table[a]
    with pytest.raises(S.SgfheError):
        bad = vals.copy()
        bad[0, 0, 0] = q
        eng.debug_flatten(bad)
    eng.close()


# ---- edge cases of the boundary ---------------------------------------------------------------------

@pytest.fixture(scope="module", params=["small-batch form", "throughput form"])
def p64(S, oc, request):
    """Params(64) engine; every test using it runs once with the default small-batch threshold
    (chunks of <= 24 bootstraps take k_fwd_phase / k_inv_column) and once with the threshold at 0
    (every chunk takes k_extprod<9>)."""
    params = S.Params(64)
    o = oc.Oracle.from_params(params)
    sk = o.private_key(1)
    bkey = o.bootstrap_key(sk, 2)
    eng = S.Engine(params)
    eng.default_small = 24 if request.param == "small-batch form" else 0
    eng.set_small_batch_max(eng.default_small)
    eng.upload_key(bkey)
    yield params, o, sk, bkey, eng
    eng.close()


def test_empty_single_and_ragged_batches(S, p64):
    params, o, sk, bkey, eng = p64
    n = params.n
    empty = eng.bootstrap_batch(np.zeros((0, n), dtype=np.uint64), [], np.zeros((0, n), dtype=np.uint64), [])
    assert empty.shape == (0, 3, n + 1)
    bits = np.random.default_rng(1).integers(0, 2, size=2 * 21).astype(np.uint8)
    a, b = o.lwe_encrypt_bits(sk, bits, 77)
    a1, b1, a2, b2 = a[0::2], b[0::2], a[1::2], b[1::2]
    ref = o.bootstrap_batch(bkey, a1, b1, a2, b2)
    for batch in (1, 7, 8, 9, 21):                       # chunk padding is a multiple of 8
        out = eng.bootstrap_batch(a1[:batch], b1[:batch], a2[:batch], b2[:batch])
        assert np.array_equal(out, ref[:batch])
    with pytest.raises(ValueError):
        eng.bootstrap_batch(a1[:3], b1[:2], a2[:3], b2[:3])


def test_chunk_size_independence_and_determinism(S, p64):
    params, o, sk, bkey, eng = p64
    bits = np.random.default_rng(2).integers(0, 2, size=2 * 40).astype(np.uint8)
    a, b = o.lwe_encrypt_bits(sk, bits, 78)
    a1, b1, a2, b2 = a[0::2], b[0::2], a[1::2], b[1::2]
    base = eng.bootstrap_batch(a1, b1, a2, b2)
    assert np.array_equal(base, eng.bootstrap_batch(a1, b1, a2, b2))      # manual.md:155-172
    for chunk in (8, 16, 24):
        eng.set_chunk(chunk)
        assert np.array_equal(eng.bootstrap_batch(a1, b1, a2, b2), base)
    eng.set_lanes(2)                                   # two streams, chunks alternate
    assert np.array_equal(eng.bootstrap_batch(a1, b1, a2, b2), base)
    eng.set_lanes(1)
    eng.set_chunk(0)
    for small in (0, 8, 40, 32):                       # large form only / mixed by chunk / small only
        eng.set_small_batch_max(small)
        for chunk in (0, 8, 24):
            eng.set_chunk(chunk)
            for lanes in (1, 2):                       # 2: pairs of chunks on two streams
                eng.set_lanes(lanes)
                assert np.array_equal(eng.bootstrap_batch(a1, b1, a2, b2), base)
    eng.set_lanes(1)
    eng.set_chunk(0)
    eng.set_small_batch_max(eng.default_small)         # leave the shared engine as the fixture made it
    assert np.array_equal(base, o.bootstrap_batch(bkey, a1, b1, a2, b2))


def test_extreme_lwe_values(S, p64):
    """Trivial LWEs (all-zero a: every x^j - 1 factor vanishes), maximal words r - 1, and the
    trivial encryption of 1 used by pack_encrypted_bits (src/fhe.jl:669-673)."""
    params, o, sk, bkey, eng = p64
    n, r = params.n, params.r
    a1 = np.zeros((4, n), dtype=np.uint64)
    a2 = np.zeros((4, n), dtype=np.uint64)
    b1 = np.array([0, params.Dr, r - 1, params.Dr], dtype=np.uint64)
    b2 = np.array([0, 0, r - 1, params.Dr], dtype=np.uint64)
    a1[2] = r - 1
    a2[2] = r - 1
    a2[3] = np.random.default_rng(3).integers(0, r, size=n, dtype=np.uint64)
    assert np.array_equal(eng.bootstrap_batch(a1, b1, a2, b2), o.bootstrap_batch(bkey, a1, b1, a2, b2))
    assert np.array_equal(eng.bootstrap_batch(a1, b1, a2, b2, raw=True),
                          o.bootstrap_batch(bkey, a1, b1, a2, b2, raw=True))


def test_no_key_and_bad_parameters_fail_loudly(S):
    eng = S.Engine(S.Params(64))
    z = np.zeros((1, 64), dtype=np.uint64)
    with pytest.raises(S.SgfheError) as ei:
        eng.bootstrap_batch(z, [0], z, [0])
    assert ei.value.code == -5
    with pytest.raises(ValueError):
        eng.upload_key(np.zeros(10, dtype=np.uint64))
    p = S.Params(64)
    bad = np.zeros((p.n, 4, 2, p.m, 2), dtype=np.uint64)
    bad[3, 1, 0, 7, 0] = p.Q & 0xFFFFFFFFFFFFFFFF             # == Q: not canonical
    bad[3, 1, 0, 7, 1] = p.Q >> 64
    with pytest.raises(S.SgfheError) as ei:
        eng.upload_key(bad)
    assert ei.value.code == -1
    eng.close()
    with pytest.raises(S.SgfheError):
        S.Engine(S.Params.custom(8, 1 << 100, 1 << 61))       # Q >= 2^94


def test_device_form_export_import(S, p64):
    import torch
    params, o, sk, bkey, eng = p64
    nbytes = eng.key_device_form_bytes()
    assert len(eng.primes()) == 4                        # Params(64): 4 primes cover 20 m B Q
    assert nbytes == 64 + params.n * 4 * 8 * params.m * 4  # 64-byte header + payload
    blob = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    eng.export_key_device_form(blob.data_ptr())
    eng2 = S.Engine(params)
    eng2.import_key_device_form(blob.data_ptr())
    bits = np.array([1, 0, 1, 1], dtype=np.uint8)
    a, b = o.lwe_encrypt_bits(sk, bits, 5)
    assert np.array_equal(eng2.bootstrap_batch(a[0::2], b[0::2], a[1::2], b[1::2]),
                          eng.bootstrap_batch(a[0::2], b[0::2], a[1::2], b[1::2]))
    eng2.close()
    # the header pins the parameter set: a blob of Params(64) is refused by every other ctx, and
    # so is a damaged header
    hdr = blob[:64].cpu().numpy().copy()
    assert bytes(hdr[:8]) == b"SGFHEKEY"
    import bigint_oracle as BO
    other = S.Engine(S.Params.custom(64, BO.find_modulus(1024, 1 << 61), 1 << 31))
    with pytest.raises(S.SgfheError) as ei:
        other.import_key_device_form(blob.data_ptr())
    assert ei.value.code == -1 and "parameter set" in str(ei.value)
    other.close()
    eng3 = S.Engine(params)
    for off, what in ((0, "magic"), (8, "version")):
        bad = blob.clone()
        bad[off] ^= 0x5A
        with pytest.raises(S.SgfheError) as ei:
            eng3.import_key_device_form(bad.data_ptr())
        assert ei.value.code == -1 and what in str(ei.value)
    with pytest.raises(S.SgfheError):                     # and a ctx that never imported has no key
        z = np.zeros((1, params.n), dtype=np.uint64)
        eng3.bootstrap_batch(z, [0], z, [0])
    eng3.close()


def test_batch_position_independence_params1024(S):
    """Full-size ring, 600 bootstraps (more than one 512-chunk): identical inputs at different
    batch positions / chunks give identical outputs, different inputs give different ones."""
    import bench
    params = S.Params(1024)
    eng = S.Engine(params)
    eng.upload_key(bench.random_key(params, 3))
    rng = np.random.default_rng(4)
    batch = 600
    a1 = rng.integers(0, params.r, size=(batch, params.n), dtype=np.uint64)
    a2 = rng.integers(0, params.r, size=(batch, params.n), dtype=np.uint64)
    b1 = rng.integers(0, params.r, size=batch, dtype=np.uint64)
    b2 = rng.integers(0, params.r, size=batch, dtype=np.uint64)
    for dup in (255, 256, 511, 599):
        a1[dup], a2[dup], b1[dup], b2[dup] = a1[0], a2[0], b1[0], b2[0]
    out = eng.bootstrap_batch(a1, b1, a2, b2)
    for dup in (255, 256, 511, 599):
        assert np.array_equal(out[dup], out[0])
    assert not np.array_equal(out[1], out[0])
    single = eng.bootstrap_batch(a1[:1], b1[:1], a2[:1], b2[:1])
    assert np.array_equal(single[0], out[0])
    eng.close()


# ---- packing (pack_encrypted_bits, src/fhe.jl:660-696) ---------------------

def test_pack_encrypted_bits_params64(S, oc, p64):
    """test/api.test.jl:86-108 (deterministic branch): split -> pack_encrypted_bits -> decrypt both
    ways; the RLWE equals the oracle's bit for bit, and the committed golden vector."""
    params, o, sk, bkey, eng = p64
    n = params.n
    bits = np.random.default_rng(21).integers(0, 2, size=2 * n).astype(np.uint8)
    a, b = o.lwe_encrypt_bits(sk, bits, 22)
    w, v = eng.pack_encrypted_bits(a.reshape(2, n, n), b.reshape(2, n))       # two ciphertexts
    for ci in range(2):
        rw, rv = o.pack_encrypted_bits(bkey, a[ci * n:(ci + 1) * n], b[ci * n:(ci + 1) * n])
        assert np.array_equal(w[ci], rw) and np.array_equal(v[ci], rv)
    # host-side API round trip (Ciphertext -> decrypt directly, and via split_ciphertext)
    key = S.PrivateKey.__new__(S.PrivateKey)
    key.params, key.key = params, np.asarray(sk, dtype=np.uint64)
    ct = S.Ciphertext(params, S.RLWE(w[0], v[0]))
    assert np.array_equal(S.decrypt(key, ct), bits[:n].astype(bool))
    assert [S.decrypt(key, eb) for eb in S.split_ciphertext(ct)] == list(bits[:n].astype(bool))
    path = os.path.join(G, "pack64.json")
    if os.path.exists(path):
        d = json.load(open(path))
        gw, gv = eng.pack_encrypted_bits(np.array(d["a"], dtype=np.uint64)[None],
                                         np.array(d["b"], dtype=np.uint64)[None])
        assert [int(x) for x in gw[0]] == d["w"] and [int(x) for x in gv[0]] == d["v"]


@pytest.mark.parametrize("n", [8, 32])
def test_pack_small_synthetic(S, oc, n):
    import bigint_oracle as BO
    m = 8 * n
    Q = BO.find_modulus(2 * m, 1 << 50)
    params = S.Params.custom(n, Q, 1 << 26)
    o = oc.Oracle.from_params(params)
    sk = o.private_key(7)
    bkey = o.bootstrap_key(sk, 8, noise=2)
    eng = S.Engine(params)
    eng.upload_key(bkey)
    bits = np.random.default_rng(n).integers(0, 2, size=n).astype(np.uint8)
    a, b = o.lwe_encrypt_bits(sk, bits, 9)
    w, v = eng.pack_encrypted_bits(a[None], b[None])
    rw, rv = o.pack_encrypted_bits(bkey, a, b)
    assert np.array_equal(w[0], rw) and np.array_equal(v[0], rv)
    eng.close()


def test_pack_params512_decrypts(S, oc):
    """Full-size ring through a size-independent property: pack(bits) decrypts to bits
    (the oracle's pack at n = 512 would take minutes)."""
    params = S.Params(512)
    o = oc.Oracle.from_params(params)
    sk = o.private_key(31)
    bkey = o.bootstrap_key(sk, 32)
    eng = S.Engine(params)
    eng.upload_key(bkey)
    bits = np.random.default_rng(33).integers(0, 2, size=params.n).astype(np.uint8)
    a, b = o.lwe_encrypt_bits(sk, bits, 34)
    w, v = eng.pack_encrypted_bits(a[None], b[None])
    key = S.PrivateKey.__new__(S.PrivateKey)
    key.params, key.key = params, np.asarray(sk, dtype=np.uint64)
    assert np.array_equal(S.decrypt(key, S.Ciphertext(params, S.RLWE(w[0], v[0]))), bits.astype(bool))
    eng.close()


# ---- key generation on the device (BootstrapKey, src/fhe.jl:181-201) -----------

@pytest.mark.parametrize("n", [64, 512])
def test_device_keygen_equals_oracle_keygen(S, oc, n):
    """Same 32-byte seed (ChaCha20 streams) -> the device-generated key is byte-identical (in device form) to the
    oracle's key uploaded through sgfhe_bkey_upload, and bootstraps with it decrypt correctly."""
    import torch
    params = S.Params(n)
    o = oc.Oracle.from_params(params)
    sk = o.private_key(51)
    bkey = o.bootstrap_key(sk, 52)
    e1 = S.Engine(params)
    e1.upload_key(bkey)
    e2 = S.Engine(params)
    e2.generate_key(sk, 52)
    nbytes = e1.key_device_form_bytes()
    b1 = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    b2 = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    e1.export_key_device_form(b1.data_ptr())
    e2.export_key_device_form(b2.data_ptr())
    assert torch.equal(b1, b2)
    bits = np.array([0, 1, 1, 1], dtype=np.uint8)
    a, b = o.lwe_encrypt_bits(sk, bits, 53)
    out = e2.bootstrap_batch(a[0::2], b[0::2], a[1::2], b[1::2])
    dec = [list(o.lwe_decrypt_bits(sk, out[:, g, :n], out[:, g, n])) for g in range(3)]
    assert dec == [[0, 1], [1, 1], [1, 0]]
    e1.close()
    e2.close()


def test_host_api_end_to_end(S):
    """README example of the reference (README.md:10-29) through the Python mirror: keys,
    encrypt, split, bootstrap, decrypt -- no oracle involved."""
    rng = np.random.default_rng(3)
    params = S.Params(64)
    key = S.PrivateKey(params, rng)
    bkey = S.BootstrapKey(rng, key)                       # generated on the device
    msg = rng.integers(0, 2, size=params.n).astype(bool)
    bits = S.split_ciphertext(S.encrypt(key, rng, msg))
    for i in range(0, 8, 2):
        r_and, r_or, r_xor = S.bootstrap(bkey, None, bits[i], bits[i + 1])
        assert S.decrypt(key, r_and) == (msg[i] & msg[i + 1])
        assert S.decrypt(key, r_or) == (msg[i] | msg[i + 1])
        assert S.decrypt(key, r_xor) == (msg[i] ^ msg[i + 1])
    ct = S.pack_encrypted_bits(bkey, None, bits)
    assert np.array_equal(S.decrypt(key, ct), msg)
    ct = S.pack_encrypted_bits(bkey, rng, bits)           # test/api.test.jl:86-108, use_rng = true
    assert np.array_equal(S.decrypt(key, ct), msg)
    pkey = S.PublicKey(rng, key)                          # public-key ciphertexts feed the gates too
    pbits = S.split_ciphertext(S.encrypt(pkey, rng, msg))
    for i in range(0, 8, 2):
        r_and, r_or, r_xor = S.bootstrap(bkey, None, pbits[i], pbits[i + 1])
        assert S.decrypt(key, r_and) == (msg[i] & msg[i + 1])
        assert S.decrypt(key, r_xor) == (msg[i] ^ msg[i + 1])
    bkey_h = S.BootstrapKey(rng, key, on_host=True)       # host big-integer generation
    r = S.bootstrap(bkey_h, None, bits[0], bits[1])
    assert S.decrypt(key, r[0]) == (msg[0] & msg[1])


# ---- randomised flatten (row N4: rng != nothing, src/utils.jl:198-241) ----------------------------------

def test_random_flatten_decrypts_and_differs(S, p64):
    """flatten(rng, ...) changes ciphertexts, not plaintexts (test/api.test.jl:61-92 runs the same
    truth table for rng = nothing and a MersenneTwister).  Functional parity only: Julia's stream
    is not reproducible, so the check is decryption plus noise size."""
    params, o, sk, bkey, eng = p64
    bits = np.array([0, 0, 0, 1, 1, 0, 1, 1] * 3, dtype=np.uint8)
    a, b = o.lwe_encrypt_bits(sk, bits, 79)
    a1, b1, a2, b2 = a[0::2], b[0::2], a[1::2], b[1::2]
    det = eng.bootstrap_batch(a1, b1, a2, b2)
    want = np.stack([bits[0::2] & bits[1::2], bits[0::2] | bits[1::2], bits[0::2] ^ bits[1::2]], 1)
    outs = []
    try:
        for seed in (1, 2, 1):
            eng.set_random_flatten(True, seed)
            out = eng.bootstrap_batch(a1, b1, a2, b2)
            outs.append(out)
            for g in range(3):
                dec = o.lwe_decrypt_bits(sk, out[:, g, :-1], out[:, g, -1])
                assert np.array_equal(dec, want[:, g])
        assert np.array_equal(outs[0], outs[2])            # same seed, same stream
        assert not np.array_equal(outs[0], outs[1])
        assert not np.array_equal(outs[0], det)
        # the accumulators stay canonical residues and are different RLWE pairs
        eng.set_random_flatten(True, 5)
        acc_r = _u128_ints(eng.debug_accumulators(a1, b1, a2, b2, params.n))
        eng.set_random_flatten(False)
        acc_d = _u128_ints(eng.debug_accumulators(a1, b1, a2, b2, params.n))
        assert max(acc_r) < params.Q and acc_r != acc_d
    finally:
        eng.set_random_flatten(False)
    assert np.array_equal(eng.bootstrap_batch(a1, b1, a2, b2), det)   # back to bit-exact


def test_random_flatten_chunks_draw_distinct_streams(S, p64):
    """Identical inputs in different batch positions / chunks must not reuse random draws."""
    params, o, sk, bkey, eng = p64
    a, b = o.lwe_encrypt_bits(sk, np.array([1, 1], dtype=np.uint8), 80)
    a1 = np.repeat(a[0:1], 20, 0); b1 = np.repeat(b[0:1], 20)
    a2 = np.repeat(a[1:2], 20, 0); b2 = np.repeat(b[1:2], 20)
    try:
        eng.set_random_flatten(True, 9)
        eng.set_chunk(8)
        out = eng.bootstrap_batch(a1, b1, a2, b2)
        again = eng.bootstrap_batch(a1, b1, a2, b2)        # call counter advances the stream
    finally:
        eng.set_chunk(0)
        eng.set_random_flatten(False)
    rows = {out[i].tobytes() for i in range(20)}
    assert len(rows) == 20
    assert not np.array_equal(out, again)
    for g, w in enumerate((1, 1, 0)):
        assert np.all(o.lwe_decrypt_bits(sk, out[:, g, :-1], out[:, g, -1]) == w)


def test_pack_encrypted_bits_randomised(S, oc, p64):
    """test/api.test.jl:86-108 with use_rng = true: split -> pack_encrypted_bits(bkey, rng, .) ->
    decrypt directly and via split_ciphertext; the RLWE differs from the deterministic one and
    from another seed's, and repeats for the same seed and call number."""
    params, o, sk, bkey, eng = p64
    n = params.n
    bits = np.random.default_rng(6).integers(0, 2, size=n).astype(np.uint8)
    a, b = o.lwe_encrypt_bits(sk, bits, 81)
    w0, v0 = eng.pack_encrypted_bits(a[None], b[None])
    key = S.PrivateKey.__new__(S.PrivateKey)
    key.params, key.key = params, np.asarray(sk, dtype=np.uint64)
    outs = []
    try:
        for seed in (3, 4, 3):
            eng.set_random_flatten(True, seed)
            w1, v1 = eng.pack_encrypted_bits(a[None], b[None])
            ct = S.Ciphertext(params, S.RLWE(w1[0], v1[0]))
            assert np.array_equal(S.decrypt(key, ct), bits.astype(bool))
            assert [S.decrypt(key, eb) for eb in S.split_ciphertext(ct)] == list(bits.astype(bool))
            outs.append((w1, v1))
    finally:
        eng.set_random_flatten(False)
    assert np.array_equal(outs[0][0], outs[2][0]) and np.array_equal(outs[0][1], outs[2][1])
    assert not np.array_equal(outs[0][0], outs[1][0])
    assert not np.array_equal(outs[0][0], w0)
    w2, v2 = eng.pack_encrypted_bits(a[None], b[None])      # and back to the bit-exact path
    assert np.array_equal(w0, w2) and np.array_equal(v0, v2)


def test_random_flatten_params1024_and_host_api(S):
    rng = np.random.default_rng(8)
    params = S.Params(1024)
    key = S.PrivateKey(params, rng)
    bkey = S.BootstrapKey(rng, key)                       # a basis per flatten mode: five and six primes
    assert len(bkey.engine.primes()) == 5
    msg = rng.integers(0, 2, size=params.n).astype(bool)
    bits = S.split_ciphertext(S.encrypt(key, rng, msg))
    res = S.bootstrap_batch(bkey, rng, bits[0:16:2], bits[1:16:2])
    for i, (r_and, r_or, r_xor) in enumerate(res):
        x, y = msg[2 * i], msg[2 * i + 1]
        assert S.decrypt(key, r_and) == (x & y)
        assert S.decrypt(key, r_or) == (x | y)
        assert S.decrypt(key, r_xor) == (x ^ y)
    r = S.bootstrap(bkey, None, bits[0], bits[1])           # and back to deterministic
    assert S.decrypt(key, r[2]) == (msg[0] ^ msg[1])


def test_chained_gates_soak(S):
    """examples/depth.jl:36-78: feed (AND, XOR) of one level into the next for many levels; every
    level must decrypt and the LWE error must stay inside the decryption margin (no growth with
    depth: each bootstrap refreshes the noise).  16 independent chains in one batch, alternating
    the deterministic and the randomised flatten."""
    rng = np.random.default_rng(21)
    params = S.Params(64)
    key = S.PrivateKey(params, rng)
    bkey = S.BootstrapKey(rng, key)
    msg = rng.integers(0, 2, size=params.n).astype(bool)
    bits = S.split_ciphertext(S.encrypt(key, rng, msg))
    e1, e2 = list(bits[0:32:2]), list(bits[1:32:2])
    y1, y2 = msg[0:32:2].copy(), msg[1:32:2].copy()

    def lwe_error(eb, ref):
        e = (int(eb.lwe.b) - int(np.sum(eb.lwe.a * key.key, dtype=np.uint64)) - int(ref) * params.Dr) % params.r
        return e - params.r if e > params.r // 2 else e

    worst = 0
    for level in range(24):
        res = S.bootstrap_batch(bkey, rng if level % 2 else None, e1, e2)
        for i, (r_and, r_or, r_xor) in enumerate(res):
            assert S.decrypt(key, r_and) == (y1[i] & y2[i])
            assert S.decrypt(key, r_or) == (y1[i] | y2[i])
            assert S.decrypt(key, r_xor) == (y1[i] ^ y2[i])
        e1 = [r[0] for r in res]
        e2 = [r[2] for r in res]
        y1, y2 = y1 & y2, y1 ^ y2
        worst = max(worst, max(abs(lwe_error(e, y)) for e, y in zip(e1 + e2, list(y1) + list(y2))))
    assert worst < params.Dr // 2


# ---- fixtures of the Julia reference itself (sgfhe.jl_amd/julia/make_fixtures.jl), when present ------

def _engine_julia_case(S, oc, d, key):
    import julia_fixture as JF
    p = d["params"]
    params = S.Params.custom(p["n"], int(p["Q"]), int(p["B"]), DQ_tilde=int(p["DQ_tilde"]))
    assert (params.r, params.m) == (p["r"], p["m"])
    eng = S.Engine(params)
    try:
        eng.upload_key(key)                                     # the reference's own key, as it is
        a1, b1, a2, b2 = JF.inputs(d)
        out = eng.bootstrap_batch(a1, b1, a2, b2)               # the drop-in call, rng = nothing
        raw = eng.bootstrap_batch(a1, b1, a2, b2, raw=True)
        o = oc.Oracle.from_params(params)
        sk = np.array(d["sk"], dtype=np.uint64)
        JF.check(d, out, [[_u128_ints(raw[i, g]) for g in range(3)] for i in range(raw.shape[0])],
                 lambda i, g: int(o.lwe_decrypt_bits(sk, out[i, g, :p["n"]], out[i, g, p["n"]])[0]))
    finally:
        eng.close()


@pytest.mark.parametrize("n", [64, 512, 1024])
def test_engine_matches_julia_reference_fixture(S, oc, n):
    """The HIP engine against bootstrap(bkey, nothing, ...) / _bootstrap_internal of the Julia
    reference itself, on the reference's own key (tests/golden/julia_p<n>.json, written by
    julia/make_fixtures.jl under Julia).  Skips until a maintainer provides the files: Julia is not
    available in this pipeline."""
    import julia_fixture as JF
    fx = JF.load(G, n)
    if fx is None:
        pytest.skip("golden/julia_p%d.json not present (run julia/make_fixtures.jl under Julia)" % n)
    _engine_julia_case(S, oc, *fx)


def test_engine_julia_fixture_path_on_a_self_made_file(S, oc, tmp_path):
    """The same consuming code on a file of that layout written from the oracle's results."""
    import julia_fixture as JF
    o = oc.Oracle.make(64)
    sk = o.private_key(9)
    bkey = o.bootstrap_key(sk, 10)
    bits = np.array([0, 1, 1, 1], dtype=np.uint8)
    a, b = o.lwe_encrypt_bits(sk, bits, 11)
    out = o.bootstrap_batch(bkey, a[0::2], b[0::2], a[1::2], b[1::2])
    raw = o.bootstrap_batch(bkey, a[0::2], b[0::2], a[1::2], b[1::2], raw=True)
    JF.write_like_julia(str(tmp_path), o, 64, sk, bkey, bits, a, b, out,
                        [[oc.u128_to_ints(raw[i, g]) for g in range(3)] for i in range(2)])
    _engine_julia_case(S, oc, *JF.load(str(tmp_path), 64))
