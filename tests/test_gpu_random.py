"""Randomised flatten (`rng::AbstractRNG` branch, src/utils.jl:198-241) on the HIP engine against
the oracle's literal restatement on the same ChaCha8 stream: accumulators, raw and ModRed outputs
and the digits themselves bit for bit, the reference's digit limits (test/internals.test.jl:48-112,
use_rng = true) on the device digits, and randomised pack_encrypted_bits (test/api.test.jl:86-108).
Run on the GPU box with `pytest -m gpu`."""

import numpy as np
import pytest

import bigint_oracle as BO

pytestmark = pytest.mark.gpu

SEED = 0x0123456789ABCDEF


def _key_lists(oc, bkey, n, m):
    vals = oc.u128_to_ints(bkey)
    return [[[vals[((k * 4 + r) * 2 + c) * m:((k * 4 + r) * 2 + c + 1) * m] for c in range(2)]
             for r in range(4)] for k in range(n)]


def _setup(S, oc, params, key_seed, noise=None, random_flatten=False):
    o = oc.Oracle.from_params(params)
    sk = o.private_key(key_seed)
    bkey = o.bootstrap_key(sk, key_seed + 1, noise=noise)
    eng = S.Engine(params, random_flatten=random_flatten)
    eng.upload_key(bkey)
    bp = BO.Params.custom(params.n, params.Q, params.B, DQ_tilde=params.DQ_tilde)
    return o, sk, bkey, eng, bp, _key_lists(oc, bkey, params.n, params.m)


def _oracle_run(bp, bk, a1, b1, a2, b2, boot, call, checkpoints):
    """Oracle bootstrap of one input pair as bootstrap `boot` of call `call`."""
    rng = BO.ChaChaFlatten(bp, SEED, boot, call)
    acc = {}

    def trace(k, a, b):
        if (k + 1) in checkpoints:
            acc[k + 1] = (list(a), list(b))
    raw = BO.bootstrap_internal(bp, bk, ([int(x) for x in a1], int(b1)), ([int(x) for x in a2], int(b2)),
                                trace=trace, rng=rng)
    return raw, acc, rng


def _ints(arr):
    flat = np.ascontiguousarray(arr).reshape(-1, 2)
    return [int(lo) | (int(hi) << 64) for lo, hi in flat]


@pytest.mark.parametrize("ring", ["params64", "synthetic", "wide base"])
def test_random_mode_equals_oracle_bit_for_bit(S, oc, ring):
    if ring == "params64":
        params, noise = S.Params(64), None
    elif ring == "synthetic":                   # m = 128: another pass structure, odd-free base
        n = 16
        params, noise = S.Params.custom(n, BO.find_modulus(16 * n, 1 << 52), 1 << 27), 2
    else:
        # B = 3 * 2^45 + 1 >= 2^46 (odd): the stored digits u + s + xmax reach 4 B = 1.5 * 2^48 and
        # take the third plane of the digit record (MODE_WIDE), as at Params(2048) (B = 35 * 2^41);
        # Q just under B^2, 92 bits: six RNS primes, four 29-bit limbs in k_crt_lean
        n = 8
        params, noise = S.Params.custom(n, BO.find_modulus(16 * n, 1 << 92), 3 * (1 << 45) + 1), 2
    o, sk, bkey, eng, bp, bk = _setup(S, oc, params, 300, noise)
    n, m, B, Q = params.n, params.m, params.B, params.Q
    if ring == "wide base":
        assert len(eng.primes()) == 6 and B >= 1 << 46
    bits = np.array([1, 1, 0, 1, 1, 0], dtype=np.uint8)
    a, b = o.lwe_encrypt_bits(sk, bits, 301)
    a1, b1, a2, b2 = a[0::2], b[0::2], a[1::2], b[1::2]
    batch = 3
    eng.set_random_flatten(True, SEED)
    checkpoints = (1, 2, n)
    call = 0
    s_shift = (B // 2 - 1 if B % 2 == 0 else (B - 1) // 2) + BO.flatten_xmax(B)
    for it in checkpoints:                      # every call advances the call counter
        acc = eng.debug_accumulators(a1, b1, a2, b2, it)
        for t in range(batch):
            _, ref, _ = _oracle_run(bp, bk, a1[t], b1[t], a2[t], b2[t], t, call, {it})
            assert _ints(acc[t, 0]) == ref[it][0], "acc_a, bootstrap %d after %d" % (t, it)
            assert _ints(acc[t, 1]) == ref[it][1], "acc_b, bootstrap %d after %d" % (t, it)
        call += 1
    # the digits the next iteration consumes: the oracle's flatten output, inside (-2B, 2B]
    dig = eng.debug_digits(a1, b1, a2, b2, 2)
    for t in range(batch):
        _, ref, rng = _oracle_run(bp, bk, a1[t], b1[t], a2[t], b2[t], t, call, {2})
        for c in range(2):
            want = BO.flatten_poly(ref[2][c], B, 2, Q, rng.draws(c, 2))
            for i in range(2):
                u = [int(v) - s_shift for v in dig[t, c, i]]
                assert all(-2 * B < x <= 2 * B for x in u)        # internals.test.jl:48-52
                if ring == "wide base":                           # the third plane is really in use
                    assert max(int(v) for v in dig[t, c, i]) >= 1 << 48
                assert [x % Q for x in u] == want[i]
            restored = [(int(dig[t, c, 0, j]) - s_shift + (int(dig[t, c, 1, j]) - s_shift) * B) % Q
                        for j in range(m)]
            assert restored == ref[2][c]                          # sum(u .* B.^(0:l-1)) == a
    call += 1
    raw = eng.bootstrap_batch(a1, b1, a2, b2, raw=True)
    for t in range(batch):
        ref_raw, _, _ = _oracle_run(bp, bk, a1[t], b1[t], a2[t], b2[t], t, call, set())
        for g in range(3):
            assert _ints(raw[t, g]) == ref_raw[g][0] + [ref_raw[g][1]]
    call += 1
    out = eng.bootstrap_batch(a1, b1, a2, b2)
    for t in range(batch):
        ref_raw, _, _ = _oracle_run(bp, bk, a1[t], b1[t], a2[t], b2[t], t, call, set())
        for g in range(3):
            want = [BO.reduce_modulus(params.r, x, Q) for x in ref_raw[g][0] + [ref_raw[g][1]]]
            assert [int(v) for v in out[t, g]] == want
    y1, y2 = bits[0::2], bits[1::2]
    for g, fn in enumerate((np.bitwise_and, np.bitwise_or, np.bitwise_xor)):
        assert np.array_equal(o.lwe_decrypt_bits(sk, out[:, g, :n], out[:, g, n]), fn(y1, y2))
    eng.close()


def test_random_mode_independent_of_scheduling(S, oc):
    """The draw of a coefficient is addressed by the bootstrap's index in the call, so chunk size,
    lanes and the small-batch threshold do not change a randomised result (include/sgfhe_hip.h)."""
    params = S.Params(64)
    o, sk, bkey, eng, bp, bk = _setup(S, oc, params, 310)
    bits = np.random.default_rng(5).integers(0, 2, size=2 * 40).astype(np.uint8)
    a, b = o.lwe_encrypt_bits(sk, bits, 311)
    a1, b1, a2, b2 = a[0::2], b[0::2], a[1::2], b[1::2]

    def run(chunk, lanes, small):
        eng.set_chunk(chunk)
        eng.set_lanes(lanes)
        eng.set_small_batch_max(small)
        eng.set_random_flatten(True, SEED)          # resets the call counter
        return eng.bootstrap_batch(a1, b1, a2, b2)
    base = run(0, 1, 24)
    for chunk, lanes, small in ((8, 1, 24), (16, 2, 24), (24, 1, 0), (8, 2, 0), (0, 1, 0), (16, 1, 8)):
        assert np.array_equal(run(chunk, lanes, small), base), (chunk, lanes, small)
    # bootstrap 17 of the batch alone is bootstrap 0 of its call: a different stream
    eng.set_random_flatten(True, SEED)
    single = eng.bootstrap_batch(a1[17:18], b1[17:18], a2[17:18], b2[17:18])
    assert not np.array_equal(single[0], base[17])
    ref_raw, _, _ = _oracle_run(bp, bk, a1[17], b1[17], a2[17], b2[17], 17, 0, set())
    want = [BO.reduce_modulus(params.r, x, params.Q) for x in ref_raw[0][0] + [ref_raw[0][1]]]
    assert [int(v) for v in base[17, 0]] == want
    eng.close()


def test_randomised_pack_equals_oracle(S, oc):
    """pack_encrypted_bits(bkey, rng, enc_bits) (src/fhe.jl:660-696, test/api.test.jl:86-108 with
    use_rng = true) bit for bit against the oracle on the engine's stream: the n bootstraps and the
    flatten of every as_i (all m coefficients of the resized polynomial) are randomised."""
    n = 8
    params = S.Params.custom(n, BO.find_modulus(16 * n, 1 << 50), 1 << 26)
    o, sk, bkey, eng, bp, bk = _setup(S, oc, params, 320, noise=2)
    bits = np.array([1, 0, 0, 1, 1, 1, 0, 1, 0, 0, 1, 0, 1, 1, 1, 0], dtype=np.uint8)
    a, b = o.lwe_encrypt_bits(sk, bits, 321)
    eng.set_random_flatten(True, SEED)
    first = eng.bootstrap_batch(a[:2], b[:2], a[2:4], b[2:4])          # call 0
    w, v = eng.pack_encrypted_bits(a.reshape(2, n, n), b.reshape(2, n))   # call 1, two ciphertexts
    skl = [int(x) for x in sk]
    for ct in range(2):
        lwes = [([int(x) for x in a[ct * n + i]], int(b[ct * n + i])) for i in range(n)]
        pw, pv = BO.pack_encrypted_bits(bp, bk, lwes, seed=SEED, ct=ct, call=1)
        assert [int(x) for x in w[ct]] == pw and [int(x) for x in v[ct]] == pv
        assert BO.decrypt_ciphertext(bp, skl, pw, pv) == [int(x) for x in bits[ct * n:(ct + 1) * n]]
    eng.set_random_flatten(False)
    wd, vd = eng.pack_encrypted_bits(a.reshape(2, n, n), b.reshape(2, n))
    assert not np.array_equal(wd, w)
    lw, lv = o.pack_encrypted_bits(bkey, a[:n], b[:n])                  # deterministic: C oracle
    assert np.array_equal(wd[0], lw) and np.array_equal(vd[0], lv)
    assert first.shape == (2, 3, n + 1)
    eng.close()


def test_both_modes_on_every_ctx_at_params1024(S):
    """Params(1024): the deterministic flatten runs on five 29-bit primes, the randomised one needs
    a sixth.  A default ctx keeps a basis per mode and switches (ABI revision 6; up to revision 5 the
    sixth prime had to be asked for at ctx creation and then slowed the deterministic mode down as
    well); a ctx created with SGFHE_CTX_DETERMINISTIC_ONLY has the smaller basis only and refuses."""
    params = S.Params(1024)
    eng = S.Engine(params)
    assert len(eng.primes()) == 5 and eng.kernel_names()[1] == "k_crt_lean<5, 3>"
    eng.set_random_flatten(True, 1)
    assert len(eng.primes()) == 6 and eng.kernel_names()[1] == "k_crt_lean_rnd<6, 3, false>"
    eng.set_random_flatten(False)
    assert len(eng.primes()) == 5
    six = 64 + params.n * 6 * 8 * params.m * 4
    assert eng.key_device_form_bytes() == six            # the blob is the larger basis's key
    eng.close()
    old = S.Engine(params, random_flatten=True)          # the flag of revisions 3-5: no effect
    assert len(old.primes()) == 5
    old.close()
    det = S.Engine(params, deterministic_only=True)
    assert len(det.primes()) == 5 and det.key_device_form_bytes() == 64 + params.n * 5 * 8 * params.m * 4
    with pytest.raises(S.SgfheError) as ei:
        det.set_random_flatten(True, 1)
    assert ei.value.code == -2 and "SGFHE_CTX_DETERMINISTIC_ONLY" in str(ei.value)
    det.close()
    for n in (64, 512, 2048):                             # one basis serves both modes
        e = S.Engine(S.Params(n), deterministic_only=True)
        k = len(e.primes())
        e.set_random_flatten(True, 1)
        assert len(e.primes()) == k
        e.close()


def test_random_mode_refused_when_its_reductions_would_overflow(S):
    """A parameter set with B^2 far above Q (4 B^2 / Q >= 2^50): the randomised flatten divides
    values up to 4 B^2 by Q with a double-precision quotient estimate, so the mode is refused
    (SGFHE_ERR_UNSUPPORTED) instead of returning silently wrong digits; the deterministic flatten of
    the same ring works."""
    params = S.Params.custom(8, (1 << 17) + 1, 1 << 45)
    eng = S.Engine(params)
    with pytest.raises(S.SgfheError) as ei:
        eng.set_random_flatten(True, 1)
    assert ei.value.code == -2
    key = np.zeros((params.n, 4, 2, params.m, 2), dtype=np.uint64)
    key[..., 0] = np.random.default_rng(1).integers(0, params.Q, size=key.shape[:-1], dtype=np.uint64)
    eng.upload_key(key)
    rng = np.random.default_rng(2)
    a1 = rng.integers(0, params.r, size=(2, params.n), dtype=np.uint64)
    b1 = rng.integers(0, params.r, size=2, dtype=np.uint64)
    acc = eng.debug_accumulators(a1, b1, a1, b1, 2)
    import bigint_oracle as BO
    bp = BO.Params.custom(params.n, params.Q, params.B, DQ_tilde=params.DQ_tilde)
    vals = [int(lo) | (int(hi) << 64) for lo, hi in key.reshape(-1, 2)]
    m = params.m
    bk = [[[vals[((k * 4 + r) * 2 + c) * m:((k * 4 + r) * 2 + c + 1) * m] for c in range(2)]
           for r in range(4)] for k in range(params.n)]
    got = {}
    BO.bootstrap_internal(bp, bk, ([int(x) for x in a1[0]], int(b1[0])), ([int(x) for x in a1[0]], int(b1[0])),
                          trace=lambda k, a, b: got.__setitem__(k + 1, (list(a), list(b))))
    assert _ints(acc[0, 0]) == got[2][0] and _ints(acc[0, 1]) == got[2][1]
    eng.close()


def test_key_generation_rejects_noise_out_of_range(S):
    """sgfhe_bkey_generate forms the noise in int32 and lifts it as Q - |e|: noise >= 2^30 or
    >= Q / 2 is SGFHE_ERR_INVALID_ARG, not a silently invalid key (the oracle's generators raise
    on the same bound)."""
    import oracle_c
    params = S.Params(64)
    eng = S.Engine(params)
    sk = np.zeros(params.n, dtype=np.uint64)
    for noise in (1 << 30, (1 << 32) - 1):
        with pytest.raises(S.SgfheError) as ei:
            eng.generate_key(sk, 5, noise=noise)
        assert ei.value.code == -1
    eng.generate_key(sk, 5, noise=(1 << 30) - 1)
    eng.close()
    small = S.Params.custom(8, (1 << 17) + 1, 1 << 9)
    eng = S.Engine(small)
    with pytest.raises(S.SgfheError) as ei:
        eng.generate_key(np.zeros(8, dtype=np.uint64), 5, noise=(1 << 16) + 1)   # 2 noise >= Q
    assert ei.value.code == -1
    eng.close()
    with pytest.raises(ValueError):
        oracle_c.Oracle.from_params(params).bootstrap_key(sk, 5, noise=1 << 30)


def test_random_flatten_key_arguments(S):
    """sgfhe_set_random_flatten_key: a missing key is SGFHE_ERR_INVALID_ARG when the mode is
    switched on and accepted when it is switched off; a ctx handle is required."""
    eng = S.Engine(S.Params(64))
    L = S.lib()
    assert L.sgfhe_set_random_flatten_key(eng._h, 1, None) == -1
    assert L.sgfhe_set_random_flatten_key(eng._h, 0, None) == 0
    assert L.sgfhe_set_random_flatten_key(None, 1, bytes(32)) == -1
    assert L.sgfhe_set_random_flatten_key(eng._h, 1, bytes(32)) == 0
    eng.close()
