"""GPU parity tests: the HIP engine (through the C ABI) against the oracle, bit for bit.
Run on the GPU box with `pytest -m gpu`."""

import os

import numpy as np
import pytest

import bigint_oracle as BO
import rns_model as RM

pytestmark = pytest.mark.gpu


def _synthetic(n, qbits=50, bbits=None, seed=0):
    m = 8 * n
    Q = BO.find_modulus(2 * m, 1 << qbits)
    B = 1 << ((Q.bit_length() + 1) // 2 if bbits is None else bbits)
    return Q, B


def _inputs(oc_obj, sk, batch, seed):
    rng = np.random.default_rng(seed)
    bits = rng.integers(0, 2, size=2 * batch).astype(np.uint8)
    a, b = oc_obj.lwe_encrypt_bits(sk, bits, seed)
    return bits, a[0::2], b[0::2], a[1::2], b[1::2]


@pytest.mark.parametrize("logm", [6, 7, 8, 9, 10, 11, 12, 13, 14])
def test_ntt_matches_model(S, logm):
    """sgfhe_debug_ntt: forward = evaluations in slot order, inverse(forward(x)) = x."""
    m = 1 << logm
    n = m // 8
    Q, B = _synthetic(n)
    eng = S.Engine(S.Params.custom(n, Q, B))
    primes = eng.primes()
    rng = np.random.default_rng(logm)
    C = RM.Consts(n, m, Q, B, Q // 8)
    assert primes == C.primes                               # same count and the same primes
    N = RM.NttModel(logm)
    for pi in (0, len(primes) - 1):
        p = primes[pi]
        poly = rng.integers(0, p, size=m, dtype=np.uint64)
        fwd = eng.debug_ntt(pi, poly.astype(np.uint32))
        model = N.forward(N.to_regs(poly), C.pk[pi]["twf"], C.pk[pi]).reshape(-1) % p
        assert np.array_equal(fwd.astype(np.uint64), model)
        if logm <= 8:
            ref = RM.ntt_reference([int(v) for v in poly], C.pk[pi]["psi"], p)
            assert [int(v) for v in fwd] == ref
        back = eng.debug_ntt(pi, fwd, inverse=True)
        assert np.array_equal(back.astype(np.uint64), poly)
    eng.close()


@pytest.mark.parametrize("use_gadget", [True, False])
def test_external_product(S, oc, use_gadget):
    """test/internals.test.jl:144-166: q = 2^60 - 1 (composite), B = 2^30, length 64; with the
    pure gadget matrix the external product is the identity; with a random matrix it must equal
    the oracle."""
    m, n = 64, 8
    B = 1 << 30
    Q = B * B - 1
    params = S.Params.custom(n, Q, B)
    eng = S.Engine(params)
    o = oc.Oracle.from_params(params)
    rng = np.random.default_rng(11)
    def rnd(shape):
        v = np.zeros(shape + (2,), dtype=np.uint64)
        v[..., 0] = rng.integers(0, Q, size=shape, dtype=np.uint64)
        return v
    a, b = rnd((m,)), rnd((m,))
    A = np.zeros((4, 2, m, 2), dtype=np.uint64) if use_gadget else rnd((4, 2, m))
    for row, col, g in ((0, 0, 1), (1, 0, B), (2, 1, 1), (3, 1, B)):
        A[row, col, 0, 0] = (int(A[row, col, 0, 0]) + g) % Q
    ra, rb = eng.external_product(a, b, A)
    if use_gadget:
        assert np.array_equal(ra, a) and np.array_equal(rb, b)
    ea, eb = o.external_product(a, b, A)
    assert np.array_equal(ra, ea) and np.array_equal(rb, eb)
    eng.close()


FORMS = ["small-batch form", "throughput form"]


def _engine(S, exp, params, form):
    """Engine whose small chunks take the small-batch kernels (k_fwd_phase / k_inv_column, default
    threshold 24) or, threshold 0, the throughput kernel k_extprod<LOGM> like a large batch."""
    eng = exp.engine(S, params)
    if form == "throughput form":
        eng.set_small_batch_max(0)
    return eng


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("n", [8, 16, 32, 128, 256])
def test_small_synthetic_bootstrap(S, oc, exp, n, form):
    """Synthetic rings (m = 64 ... 2048: every pass structure below the full-size rings) vs the
    oracle: accumulators after every iteration count, raw LWEs mod Q, and ModRed words."""
    Q, B = _synthetic(n)
    params = S.Params.custom(n, Q, B)
    o = oc.Oracle.from_params(params)
    sk = o.private_key(100 + n)
    bkey = o.bootstrap_key(sk, 200 + n, noise=2)
    eng = _engine(S, exp, params, form)
    eng.upload_key(bkey)
    batch = 5
    bits, a1, b1, a2, b2 = _inputs(o, sk, batch, 300 + n)
    tag = "parity.synth%d" % n
    for it in (1, 2, n):
        exp.check(tag + ".acc%d" % it, eng.debug_accumulators(a1, b1, a2, b2, it),
                  lambda: o.bootstrap_batch(bkey, a1, b1, a2, b2, n_iters=it, want_acc=True)[1],
                  "accumulators differ after %d iterations" % it)
    exp.check(tag + ".raw", eng.bootstrap_batch(a1, b1, a2, b2, raw=True),
              lambda: o.bootstrap_batch(bkey, a1, b1, a2, b2, raw=True))
    exp.check(tag + ".out", eng.bootstrap_batch(a1, b1, a2, b2), lambda: o.bootstrap_batch(bkey, a1, b1, a2, b2))
    eng.close()


def test_params64_bootstrap_truth_table(S, oc):
    """test/api.test.jl:45-83 (deterministic branch) at Params(64), through the GPU: outputs equal
    the oracle bit for bit and decrypt to AND / OR / XOR."""
    params = S.Params(64)
    o = oc.Oracle.from_params(params)
    sk = o.private_key(1)
    bkey = o.bootstrap_key(sk, 2)
    eng = S.Engine(params)
    eng.upload_key(bkey)
    batch = 32
    bits, a1, b1, a2, b2 = _inputs(o, sk, batch, 3)
    out = eng.bootstrap_batch(a1, b1, a2, b2)
    ref = o.bootstrap_batch(bkey, a1, b1, a2, b2)
    assert np.array_equal(out, ref)
    y1, y2 = bits[0::2], bits[1::2]
    for g, fn in enumerate((np.bitwise_and, np.bitwise_or, np.bitwise_xor)):
        dec = o.lwe_decrypt_bits(sk, out[:, g, :params.n], out[:, g, params.n])
        assert np.array_equal(dec, fn(y1, y2))
    # the host-pointer entry point with the caller's own result array (the form a loop uses), through
    # the pinned staging buffers and through direct copies; a wrong array is refused
    mine = np.full(out.shape, 0xDEADBEEF, dtype=np.uint64)
    assert eng.bootstrap_batch(a1, b1, a2, b2, out=mine) is mine and np.array_equal(mine, ref)
    raw = eng.bootstrap_batch(a1, b1, a2, b2, raw=True)
    mine_raw = np.zeros(raw.shape, dtype=np.uint64)
    eng.bootstrap_batch(a1, b1, a2, b2, raw=True, out=mine_raw)
    assert np.array_equal(mine_raw, raw)
    for bad in (np.zeros(out.shape, dtype=np.int64), np.zeros(out.shape[:2], dtype=np.uint64),
                np.zeros((batch, 3, 2 * (params.n + 1)), dtype=np.uint64)[:, :, ::2]):
        with pytest.raises(ValueError):
            eng.bootstrap_batch(a1, b1, a2, b2, out=bad)
    eng.close()
    os.environ["SGFHE_HOST_PIN"] = "0"
    try:
        direct = S.Engine(params)
        direct.upload_key(bkey)
        assert np.array_equal(direct.bootstrap_batch(a1, b1, a2, b2), ref)
        direct.close()
    finally:
        del os.environ["SGFHE_HOST_PIN"]


# ---- BASELINE.json configurations at full ring size --------------------------------------------

def _big_case(S, oc, exp, tag, params, batch, key_seed, in_seed, valid_key, iters_checked, form=FORMS[0]):
    """The oracle here is the REFERENCE-SHAPED loop (24 NTT products per iteration, canonical key): its
    answers are recorded expectations (tests/expect.py); both kernel forms are held to the same digests."""
    o = oc.Oracle.from_params(params)
    sk = o.private_key(key_seed)
    eng = _engine(S, exp, params, form)
    if valid_key:
        bkey = exp.lazy(lambda: o.bootstrap_key(sk, key_seed + 1))
        if params.n >= 1024:
            eng.generate_key(sk, key_seed + 1)          # byte-identical to the oracle's (test_gpu_golden.py)
        elif exp.live:
            eng.upload_key(bkey())
    else:
        import bench
        rk = bench.random_key(params, key_seed)
        bkey = exp.lazy(lambda: rk)
        eng.upload_key(rk)
    bits, a1, b1, a2, b2 = _inputs(o, sk, batch, in_seed)
    for it in iters_checked:
        exp.check(tag + ".acc%d" % it, eng.debug_accumulators(a1[:1], b1[:1], a2[:1], b2[:1], it),
                  lambda: o.bootstrap_batch(bkey(), a1[:1], b1[:1], a2[:1], b2[:1], n_iters=it, want_acc=True)[1],
                  "accumulators differ after %d iterations" % it)
    out = exp.check(tag + ".out", eng.bootstrap_batch(a1, b1, a2, b2), lambda: o.bootstrap_batch(bkey(), a1, b1, a2, b2))
    if valid_key:
        y1, y2 = bits[0::2], bits[1::2]
        for g, fn in enumerate((np.bitwise_and, np.bitwise_or, np.bitwise_xor)):
            dec = o.lwe_decrypt_bits(sk, out[:, g, :params.n], out[:, g, params.n])
            assert np.array_equal(dec, fn(y1, y2))
    eng.close()
    return out


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("n", [128, 256])
def test_params128_256_vs_oracle(S, oc, exp, n, form):
    """The reference's parameter sets between Params(64) and Params(512) (Q 68.25 / 74.25 bits,
    m = 1024 / 2048: the pass structures with a 2- and a 3-stage partial pass): complete
    bootstraps, bit-exact and decrypting."""
    _big_case(S, oc, exp, "parity.p%d" % n, S.Params(n), batch=6, key_seed=40 + n, in_seed=50 + n, valid_key=True,
              iters_checked=(1, n), form=form)


@pytest.mark.parametrize("form", FORMS)
def test_params512_vs_oracle(S, oc, exp, form):
    """BASELINE.json config 2 ring (Params(512), Q 80.25 bits): bit-exact vs the oracle, decrypts."""
    _big_case(S, oc, exp, "parity.p512", S.Params(512), batch=8, key_seed=11, in_seed=12, valid_key=True,
              iters_checked=(1, 2), form=form)


def test_params512_full_batch_1024(S, oc, exp):
    """BASELINE.json config 2 at its full batch: 1024 bootstraps in one call (one 1024-chunk).
    The oracle covers 8 distinct input pairs; the batch tiles them 128 times in a shuffled order,
    so every one of the 1024 x 3 x 513 output words is pinned to an oracle word."""
    params = S.Params(512)
    o = oc.Oracle.from_params(params)
    sk = o.private_key(11)
    eng = exp.engine(S, params)
    eng.generate_key(sk, 12)
    bits, a1, b1, a2, b2 = _inputs(o, sk, 8, 13)
    idx = np.random.default_rng(14).permutation(np.repeat(np.arange(8), 128))
    out = exp.check("parity.p512.full1024", eng.bootstrap_batch(a1[idx], b1[idx], a2[idx], b2[idx]),
                    lambda: o.bootstrap_batch(o.bootstrap_key(sk, 12), a1, b1, a2, b2)[idx])
    assert out.shape == (1024, 3, params.n + 1)
    eng.close()


@pytest.mark.parametrize("form", FORMS)
def test_params1024_vs_oracle(S, oc, exp, form):
    """BASELINE.json config 4' (the reference's own Params(1024), Q 86.25 bits)."""
    _big_case(S, oc, exp, "parity.p1024", S.Params(1024), batch=4, key_seed=21, in_seed=22, valid_key=True,
              iters_checked=(1, 2), form=form)


def test_params1024_full_batch_4096(S, oc, exp):
    """The bench workload at full size (Params(1024), batch 4096 = 8 chunks of 512): 4 input pairs
    verified by the oracle, tiled 1024 times in a shuffled order; every output word is pinned to
    an oracle word, across all chunks and batch positions.  The key comes from the device
    generator (byte-identical to the oracle's for the same seed, tests/test_gpu_golden.py)."""
    params = S.Params(1024)
    o = oc.Oracle.from_params(params)
    sk = o.private_key(21)
    eng = exp.engine(S, params)
    eng.generate_key(sk, 22)
    bits, a1, b1, a2, b2 = _inputs(o, sk, 4, 23)
    ref4 = exp.lazy(lambda: o.bootstrap_batch(o.bootstrap_key(sk, 22), a1, b1, a2, b2))   # reference-shaped loop
    exp.check("parity.p1024.in23.out4", eng.bootstrap_batch(a1, b1, a2, b2), ref4)
    idx = np.random.default_rng(24).permutation(np.repeat(np.arange(4), 1024))
    out = exp.check("parity.p1024.full4096", eng.bootstrap_batch(a1[idx], b1[idx], a2[idx], b2[idx]), lambda: ref4()[idx])
    y1, y2 = bits[0::2], bits[1::2]
    dec = o.lwe_decrypt_bits(sk, out[:4096:512, 2, :params.n], out[:4096:512, 2, params.n])
    assert np.array_equal(dec, (y1 ^ y2)[idx[:4096:512]])
    eng.close()


@pytest.mark.parametrize("form", FORMS)
def test_params2048_largest_reference_ring(S, oc, exp, form):
    """Params(2048): the largest parameter set the reference can build (Q 92.25 bits < 2^128,
    src/fhe.jl:74-77): m = 16384, six RNS primes, B just above 2^46.  The first two k-loop
    iterations against the oracle (only key slices 0 and 1 are filled), then complete gate
    bootstraps with a device-generated key, checked by decryption, in both flatten modes
    (src/utils.jl:155-189 and :198-241)."""
    import bench
    params = S.Params(2048)
    o = oc.Oracle.from_params(params)
    if not exp.live:
        pytest.skip("two live oracle iterations and decryption checks: nothing to record "
                    "(complete Params(2048) bootstraps against recorded bytes: tests/test_gpu_round5.py)")
    eng = _engine(S, exp, params, form)
    assert len(eng.primes()) == 6                          # 5 m B Q needs six 29-bit primes
    key = np.zeros((params.n, 4, 2, params.m, 2), dtype=np.uint64)
    key[:2] = bench.random_key(params, 41)[:2]
    eng.upload_key(key)
    rng = np.random.default_rng(42)
    a1 = rng.integers(0, params.r, size=(2, params.n), dtype=np.uint64)
    a2 = rng.integers(0, params.r, size=(2, params.n), dtype=np.uint64)
    b1 = rng.integers(0, params.r, size=2, dtype=np.uint64)
    b2 = rng.integers(0, params.r, size=2, dtype=np.uint64)
    for it in (1, 2):
        _, acc_ref = o.bootstrap_batch(key, a1, b1, a2, b2, n_iters=it, want_acc=True, threads=2)
        assert np.array_equal(eng.debug_accumulators(a1, b1, a2, b2, it), acc_ref)
    del key
    sk = o.private_key(43)
    eng.generate_key(sk, 44)
    bits = np.array([0, 0, 0, 1, 1, 0, 1, 1] * 2, dtype=np.uint8)
    a, b = o.lwe_encrypt_bits(sk, bits, 45)
    out = eng.bootstrap_batch(a[0::2], b[0::2], a[1::2], b[1::2])
    y1, y2 = bits[0::2], bits[1::2]
    for g, fn in enumerate((np.bitwise_and, np.bitwise_or, np.bitwise_xor)):
        dec = o.lwe_decrypt_bits(sk, out[:, g, :params.n], out[:, g, params.n])
        assert np.array_equal(dec, fn(y1, y2))
    # the randomised flatten on the same ring (B = 35 * 2^41 > 2^46: its stored digits reach
    # 4 B > 2^48 and use the third plane of the digit record): decrypt-level truth table, and the
    # deterministic result comes back unchanged afterwards
    eng.set_random_flatten(True, 1)
    rnd = eng.bootstrap_batch(a[0::2], b[0::2], a[1::2], b[1::2])
    assert not np.array_equal(rnd, out)
    for g, fn in enumerate((np.bitwise_and, np.bitwise_or, np.bitwise_xor)):
        dec = o.lwe_decrypt_bits(sk, rnd[:, g, :params.n], rnd[:, g, params.n])
        assert np.array_equal(dec, fn(y1, y2))
    eng.set_random_flatten(False)
    assert np.array_equal(eng.bootstrap_batch(a[0::2], b[0::2], a[1::2], b[1::2]), out)
    eng.close()


@pytest.mark.parametrize("form", FORMS)
def test_synthetic_single_limb_1024(S, oc, exp, form):
    """BASELINE.json config 3: n = 1024, single-limb 64-bit prime Q', B' = 2^32 (synthetic: the
    parity target is the oracle at the same parameters, not decryption)."""
    import bench
    params = bench.make_params(S, "synth64")
    _big_case(S, oc, exp, "parity.synth64", params, batch=2, key_seed=31, in_seed=32, valid_key=False,
              iters_checked=(1, 3), form=form)


def test_synthetic_single_limb_full_batch_4096(S, oc, exp):
    """BASELINE.json config 3 at its full batch: n = 1024 over a single-limb 64-bit prime, batch
    4096 (8 chunks of 512 on the four-prime grid at m = 8192).  Four random LWE input pairs are
    verified by the oracle on the synthetic key of the bench (`bench.random_key`), tiled 1024
    times in a shuffled order: every one of the 4096 x 3 x 1025 output words is pinned to an
    oracle word, across all chunks and batch positions.  (Synthetic ring: the parity target is the
    oracle at the same parameters, not decryption)"""
    import bench
    params = bench.make_params(S, "synth64")
    o = oc.Oracle.from_params(params)
    key = bench.random_key(params, 31)
    eng = exp.engine(S, params)
    if exp.live:
        assert len(eng.primes()) == 4
    eng.upload_key(key)
    rng = np.random.default_rng(33)
    a1 = rng.integers(0, params.r, size=(4, params.n), dtype=np.uint64)
    a2 = rng.integers(0, params.r, size=(4, params.n), dtype=np.uint64)
    b1 = rng.integers(0, params.r, size=4, dtype=np.uint64)
    b2 = rng.integers(0, params.r, size=4, dtype=np.uint64)
    idx = np.random.default_rng(34).permutation(np.repeat(np.arange(4), 1024))
    out = exp.check("parity.synth64.full4096", eng.bootstrap_batch(a1[idx], b1[idx], a2[idx], b2[idx]),
                    lambda: o.bootstrap_batch(key, a1, b1, a2, b2, threads=4)[idx])
    assert out.shape == (4096, 3, params.n + 1)
    eng.close()


def test_config5_per_gpu_shards_of_65536(S, oc, exp):
    """BASELINE.json config 5 on the one GPU a box has: Params(1024), a logical batch of 65536
    bootstraps cut into the 8 contiguous shards `shard_range(65536, g, 8)` that 8 ranks would
    take (8192 each = 16 chunks of 512).  Every shard runs through Engine.bootstrap_batch as a
    rank would run it and every one of its 8192 x 3 x 1025 words is pinned to an oracle word
    (4 oracle-verified input pairs tiled in a shuffled order over the 65536 rows); two
    neighbouring shards equal one call on their 16384 rows (the batch independence of
    src/fhe.jl:579-582 that the sharding relies on)."""
    params = S.Params(1024)
    o = oc.Oracle.from_params(params)
    sk = o.private_key(21)
    eng = exp.engine(S, params)
    eng.generate_key(sk, 22)
    bits, a1, b1, a2, b2 = _inputs(o, sk, 4, 23)
    # the four oracle rows every one of the 65536 is a copy of (the key and inputs of
    # test_params1024_full_batch_4096: the same recorded expectation)
    ref = exp.check("parity.p1024.in23.out4", eng.bootstrap_batch(a1, b1, a2, b2),
                    lambda: o.bootstrap_batch(o.bootstrap_key(sk, 22), a1, b1, a2, b2, threads=4))
    if not exp.live:
        return
    total, world = 65536, 8
    idx = np.random.default_rng(25).permutation(np.repeat(np.arange(4), total // 4))
    shards = []
    for g in range(world):
        lo, hi = S.distributed.shard_range(total, g, world)
        assert hi - lo == 8192
        rows = idx[lo:hi]
        out = eng.bootstrap_batch(a1[rows], b1[rows], a2[rows], b2[rows])
        assert out.shape == (8192, 3, params.n + 1)
        assert np.array_equal(out, ref[rows]), "shard %d differs from the oracle" % g
        shards.append(out)
    # one call over the rows of shards 3 and 4 (16384 rows across the middle of the batch; every shard is
    # already pinned word for word above, so the whole 65536 in one call would only repeat that)
    lo, hi = S.distributed.shard_range(total, 3, world)[0], S.distributed.shard_range(total, 4, world)[1]
    two = eng.bootstrap_batch(a1[idx[lo:hi]], b1[idx[lo:hi]], a2[idx[lo:hi]], b2[idx[lo:hi]])
    assert two.tobytes() == np.concatenate(shards[3:5], axis=0).tobytes()
    whole = np.concatenate(shards, axis=0)
    y1, y2 = bits[0::2], bits[1::2]
    dec = o.lwe_decrypt_bits(sk, whole[::4096, 0, :params.n], whole[::4096, 0, params.n])
    assert np.array_equal(dec, (y1 & y2)[idx[::4096]])
    eng.close()


def test_params64_soak_4096_random_bootstraps(S, oc, exp):
    """4096 independent gate bootstraps at Params(64) with a device-generated key, every output
    word against the oracle (random LWE inputs and valid encryptions mixed), all three gates
    decrypting where the inputs are encryptions: a wide net for rare range / carry cases of the
    signed lazy arithmetic (the worst case proper is tests/test_gpu_worstcase.py)."""
    params = S.Params(64)
    o = oc.Oracle.from_params(params)
    sk = o.private_key(71)
    eng = exp.engine(S, params)
    eng.generate_key(sk, 72)
    rng = np.random.default_rng(73)
    bits = rng.integers(0, 2, size=2 * 2048).astype(np.uint8)
    a, b = o.lwe_encrypt_bits(sk, bits, 74)
    a1 = np.concatenate([a[0::2], rng.integers(0, params.r, size=(2048, params.n), dtype=np.uint64)])
    a2 = np.concatenate([a[1::2], rng.integers(0, params.r, size=(2048, params.n), dtype=np.uint64)])
    b1 = np.concatenate([b[0::2], rng.integers(0, params.r, size=2048, dtype=np.uint64)])
    b2 = np.concatenate([b[1::2], rng.integers(0, params.r, size=2048, dtype=np.uint64)])
    out = exp.check("parity.p64.soak4096", eng.bootstrap_batch(a1, b1, a2, b2),
                    lambda: o.bootstrap_batch(o.bootstrap_key(sk, 72), a1, b1, a2, b2, threads=16))
    y1, y2 = bits[0::2], bits[1::2]
    for g, fn in enumerate((np.bitwise_and, np.bitwise_or, np.bitwise_xor)):
        dec = o.lwe_decrypt_bits(sk, out[:2048, g, :params.n], out[:2048, g, params.n])
        assert np.array_equal(dec, fn(y1, y2))
    eng.close()
