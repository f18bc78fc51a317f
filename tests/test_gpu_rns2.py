"""BASELINE.json config 4: the bootstrap over the composite modulus Q = B * Bp of two NTT-friendly
primes, the ring the reference's RNS2Number type (src/rns.jl) and Scheme2 parameters
(src/fhe2.jl:57-60,98-101) describe.  The reference itself has no bootstrap over that type
(SURVEY.md F5): the oracle is the C restatement with limb-wise NTT products (rns.jl:51-60) and the
conversions of rns.jl:16-18 / :32-40, cross-checked with the big-integer oracle.
Run on the GPU box with `pytest -m gpu`."""

import hashlib
import json
import os

import numpy as np
import pytest

import bigint_oracle as BO

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _ints(arr):
    flat = np.ascontiguousarray(arr).reshape(-1, 2)
    return [int(lo) | (int(hi) << 64) for lo, hi in flat]


def _u128(vals):
    out = np.zeros((len(vals), 2), dtype=np.uint64)
    out[:, 0] = [v & 0xFFFFFFFFFFFFFFFF for v in vals]
    out[:, 1] = [v >> 64 for v in vals]
    return out


def h_ints(vals, nbytes=16):
    h = hashlib.sha256()
    for v in vals:
        h.update(int(v).to_bytes(nbytes, "little"))
    return h.hexdigest()


def test_rns2_conversions_match_rns_jl(S):
    """k_rns2_to_canon / k_canon_to_rns2 against the literal restatement of src/rns.jl:16-18 and
    :32-40 (BO.rns2_from_int / rns2_to_int), at config 4's 43-bit primes and at small ones."""
    import bench
    for B, Bp, n in ((None, None, 1024), (BO.find_modulus(256, (1 << 24) + 300), None, 16)):
        if B is None:
            B, Bp = bench.rns2_moduli(S)
        else:
            Bp, B = B, BO.find_modulus(256, B + 1)
        Q = B * Bp
        eng = S.Engine(S.Params.custom(n, Q, B))
        rng = np.random.default_rng(B % 1000)
        vals = [0, 1, Q - 1, B, Bp, Q // 2, Q // 2 + 1] + [
            (int(rng.integers(0, 1 << 62)) << 40 | int(rng.integers(0, 1 << 40))) % Q for _ in range(4000)]
        pairs = eng.rns2_convert(_u128(vals), B, Bp, to_pairs=True)
        want = [BO.rns2_from_int(v, B, Bp) for v in vals]
        assert [(int(x), int(y)) for x, y in pairs] == want
        back = eng.rns2_convert(pairs, B, Bp, to_pairs=False)
        assert _ints(back) == vals
        assert _ints(back) == [BO.rns2_to_int(x, y, B, Bp) for x, y in want]
        bad = pairs.copy()
        bad[5, 0] = B                                          # v1 == m1: not a residue
        with pytest.raises(S.SgfheError) as ei:
            eng.rns2_convert(bad, B, Bp, to_pairs=False)
        assert ei.value.code == -1
        with pytest.raises(S.SgfheError):
            eng.rns2_convert(pairs, B, Bp + 2, to_pairs=False)  # m1 * m2 != Q
        eng.close()


@pytest.mark.parametrize("form", ["small", "large"])
def test_rns2_small_ring_three_ways(S, oc, form):
    """n = 16 over Q = B * Bp: the engine with the key uploaded as canonical residues, as RNS2Number
    limb pairs, the C oracle with schoolbook products and with limb-wise NTT products (rns.jl:51-60),
    and the big-integer oracle all agree; raw outputs also leave as (v1, v2) pairs."""
    n, m = 16, 128
    Bp = BO.find_modulus(2 * m, 1 << 24)
    B = BO.find_modulus(2 * m, Bp + 1)                         # rule of src/fhe2.jl:57-58
    Q = B * Bp
    params = S.Params.custom(n, Q, B)
    o_sb = oc.Oracle.from_params(params)
    o_rns = oc.Oracle.from_params(params, rns2=(B, Bp))
    assert not o_sb.uses_ntt and not o_sb.uses_rns2 and o_rns.uses_rns2
    sk = o_sb.private_key(9)
    bkey = o_sb.bootstrap_key(sk, 10, noise=2)
    assert np.array_equal(bkey, o_rns.bootstrap_key(sk, 10, noise=2))
    vals = oc.u128_to_ints(bkey)
    pairs = np.array([BO.rns2_from_int(v, B, Bp) for v in vals], dtype=np.uint64).reshape(bkey.shape)
    e1, e2 = S.Engine(params), S.Engine(params)
    for e in (e1, e2):
        if form == "large":
            e.set_small_batch_max(0)
    e1.upload_key(bkey)
    e2.upload_key_rns2(pairs, B, Bp)
    bits = np.array([0, 1, 1, 1, 0, 0, 1, 0, 0, 1], dtype=np.uint8)
    a, b = o_sb.lwe_encrypt_bits(sk, bits, 11)
    a1, b1, a2, b2 = a[0::2], b[0::2], a[1::2], b[1::2]
    ref = o_sb.bootstrap_batch(bkey, a1, b1, a2, b2)
    assert np.array_equal(ref, o_rns.bootstrap_batch(bkey, a1, b1, a2, b2))
    assert np.array_equal(e1.bootstrap_batch(a1, b1, a2, b2), ref)
    assert np.array_equal(e2.bootstrap_batch(a1, b1, a2, b2), ref)
    raw = e2.bootstrap_batch(a1, b1, a2, b2, raw=True)
    assert np.array_equal(raw, o_rns.bootstrap_batch(bkey, a1, b1, a2, b2, raw=True))
    rp = e2.bootstrap_batch(a1, b1, a2, b2, rns2=True)
    assert [(int(x), int(y)) for x, y in rp.reshape(-1, 2)] == [BO.rns2_from_int(v, B, Bp) for v in _ints(raw)]
    with pytest.raises(S.SgfheError):
        e1.bootstrap_batch(a1, b1, a2, b2, rns2=True)          # no RNS2 moduli on that ctx
    bp = BO.Params.custom(n, Q, B)
    bk = [[[vals[((k * 4 + r) * 2 + c) * m:((k * 4 + r) * 2 + c + 1) * m] for c in range(2)]
           for r in range(4)] for k in range(n)]
    big = BO.bootstrap_internal(bp, bk, ([int(x) for x in a1[0]], int(b1[0])), ([int(x) for x in a2[0]], int(b2[0])))
    for g in range(3):
        assert _ints(raw[0, g]) == big[g][0] + [big[g][1]]
    dec = o_sb.lwe_decrypt_bits(sk, ref[:, 2, :n], ref[:, 2, n])
    assert np.array_equal(dec, bits[0::2] ^ bits[1::2])
    e1.close()
    e2.close()


@pytest.fixture(scope="module")
def cfg4(S, oc, exp):
    """Config 4 at its real size: n = 1024, m = 8192, the two primes of bench.rns2_moduli; key from
    the oracle's generator (1 GiB of canonical residues), handed to the engine as RNS2Number limb
    pairs made by the device conversion that test_rns2_conversions_match_rns_jl pins."""
    import bench
    params = bench.make_params(S, "rns2")
    B, Bp = bench.rns2_moduli(S)
    assert params.Q == B * Bp and params.B == B and 86 < np.log2(float(params.Q)) < 87
    o = oc.Oracle.from_params(params, rns2=(B, Bp))
    sk = o.private_key(41)
    eng = exp.engine(S, params)
    if exp.live:
        assert len(eng.primes()) == 5
    # (the key itself is made on the GPU box too: the RNS2Number upload at full size needs its limb pairs)
    from conftest import oracle_threads
    bkey = exp.lazy(lambda: o.bootstrap_key(sk, 42, threads=oracle_threads()))
    if exp.live:
        pairs = eng.rns2_convert(bkey(), B, Bp, to_pairs=True)
        for idx in ((0, 0, 0, 0), (1023, 3, 1, 8191), (512, 2, 0, 77)):      # spot checks of the hand-over
            v = int(bkey()[idx][0]) | (int(bkey()[idx][1]) << 64)
            assert (int(pairs[idx][0]), int(pairs[idx][1])) == BO.rns2_from_int(v, B, Bp)
        eng.upload_key_rns2(pairs, B, Bp)
        del pairs
    # four input pairs through the oracle (about a minute per bootstrap and core, reference-shaped RNS2Number
    # loop): the raw LWEs over Z_Q, and their ModRed words by the literal rescale (src/utils.jl:78-92) --
    # recorded expectations (tests/expect.py); computed here only if one is missing or differs
    bits = np.array([0, 0, 0, 1, 1, 0, 1, 1], dtype=np.uint8)
    a, b = o.lwe_encrypt_bits(sk, bits, 43)
    lwe = (a[0::2], b[0::2], a[1::2], b[1::2])
    ref_raw = exp.lazy(lambda: o.bootstrap_batch(bkey(), *lwe, raw=True, threads=4))
    ref = exp.lazy(lambda: np.array([BO.reduce_modulus(params.r, v, params.Q) for v in _ints(ref_raw())],
                                    dtype=np.uint64).reshape(4, 3, params.n + 1))
    yield params, o, sk, bkey, eng, (B, Bp), bits, lwe, ref_raw, ref
    eng.close()


def test_config4_key_forms_agree(S, exp, cfg4):
    """The device-form key from the RNS2 upload is byte-identical to the key generated on the
    device from the same seed (composite Q through k_keygen_*), at full size."""
    if not exp.live:
        pytest.skip("engine against engine: nothing to record")
    import torch
    params, o, sk, bkey, eng, (B, Bp) = cfg4[:6]
    e2 = S.Engine(params)
    e2.generate_key(sk, 42)
    nbytes = eng.key_device_form_bytes()
    b1 = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    b2 = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    eng.export_key_device_form(b1.data_ptr())
    e2.export_key_device_form(b2.data_ptr())
    assert torch.equal(b1, b2)
    e2.close()


def test_config4_full_bootstraps_vs_oracle_and_fixture(S, oc, exp, cfg4):
    """Complete bootstraps (all 1024 iterations, k_final, ModRed) over the composite modulus against
    the oracle; accumulators at k in {1, 2, 512, 1024} and outputs against the committed fixture
    tests/golden/cfg4.json (big-integer cross-checked, tests/golden/make_golden.py)."""
    params, o, sk, bkey, eng, (B, Bp), bits, (a1, b1, a2, b2), ref_raw, ref = cfg4
    n = params.n
    out = exp.check("rns2.cfg4.out", eng.bootstrap_batch(a1, b1, a2, b2), ref)
    exp.check("rns2.cfg4.out0.c_modred", out[:1] if exp.live else None,          # the C restatement's ModRed too
              lambda: o.bootstrap_batch(bkey(), a1[:1], b1[:1], a2[:1], b2[:1]))
    raw = exp.check("rns2.cfg4.raw", eng.bootstrap_batch(a1, b1, a2, b2, raw=True), ref_raw)
    if exp.live:
        pairs = eng.bootstrap_batch(a1[:1], b1[:1], a2[:1], b2[:1], rns2=True)
        assert [(int(x), int(y)) for x, y in pairs.reshape(-1, 2)] == [
            BO.rns2_from_int(v, B, Bp) for v in _ints(raw[:1])]
    y1, y2 = bits[0::2], bits[1::2]
    for g, fn in enumerate((np.bitwise_and, np.bitwise_or, np.bitwise_xor)):
        assert np.array_equal(o.lwe_decrypt_bits(sk, out[:, g, :n], out[:, g, n]), fn(y1, y2))
    path = os.path.join(G, "cfg4.json")
    if not os.path.exists(path):
        pytest.skip("golden/cfg4.json not generated")
    if not exp.live:
        return
    d = json.load(open(path))
    assert d["params"]["Q"] == str(params.Q) and d["sk_seed"] == 41 and d["key_seed"] == 42
    case = d["cases"][0]
    ga1, gb1 = np.array([case["lwe1"]["a"]], dtype=np.uint64), [case["lwe1"]["b"]]
    ga2, gb2 = np.array([case["lwe2"]["a"]], dtype=np.uint64), [case["lwe2"]["b"]]
    for k, (ha, hb) in case["acc_sha256_after"].items():
        acc = eng.debug_accumulators(ga1, gb1, ga2, gb2, int(k))
        assert h_ints(_ints(acc[0, 0])) == ha and h_ints(_ints(acc[0, 1])) == hb, "after %s" % k
    graw = eng.bootstrap_batch(ga1, gb1, ga2, gb2, raw=True)
    gout = eng.bootstrap_batch(ga1, gb1, ga2, gb2)
    for g in range(3):
        assert h_ints(_ints(graw[0, g])) == case["raw_sha256"][g]
        assert h_ints([int(v) for v in gout[0, g]], 8) == case["out_sha256"][g]


def test_config4_full_batch_4096(S, oc, exp, cfg4):
    """BASELINE.json config 4 at its batch: 4096 bootstraps (8 chunks of 512), four oracle-verified
    input pairs tiled 1024 times in a shuffled order, so every output word is pinned to an oracle
    word; the truth table decrypts."""
    params, o, sk, bkey, eng, (B, Bp), bits, (a1, b1, a2, b2), ref_raw, ref = cfg4
    n = params.n
    idx = np.random.default_rng(44).permutation(np.repeat(np.arange(4), 1024))
    out = exp.check("rns2.cfg4.full4096", eng.bootstrap_batch(a1[idx], b1[idx], a2[idx], b2[idx]), lambda: ref()[idx])
    assert out.shape == (4096, 3, n + 1)
    y1, y2 = bits[0::2], bits[1::2]
    for g, fn in enumerate((np.bitwise_and, np.bitwise_or, np.bitwise_xor)):
        dec = o.lwe_decrypt_bits(sk, out[::97, g, :n], out[::97, g, n])
        assert np.array_equal(dec, fn(y1, y2)[idx[::97]])
