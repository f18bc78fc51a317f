"""Round-4 parity evidence on the GPU:

 * the Params(1024) ctx with a basis per flatten mode (five primes deterministic, six randomised:
   the reference's documented call bootstrap(bkey, rng, ...), README.md:24, docs/src/manual.md:144,
   src/utils.jl:198-241, works on every ctx) under the oracle in both modes, at its ring and at
   full batch; the five-prime key form derived on the device from the six-prime one;
 * a soak on DISTINCT inputs at every full-size configuration (test/api.test.jl:45-83 widened);
 * calls on one ctx from two streams / two threads equal the same calls made one after the other
   (the reference call is pure, src/fhe.jl:608-621);
 * the randomised flatten on parameter sets with a two-limb modulus and a base far above sqrt(Q).

Run on the GPU box with `pytest -m gpu`.  Everything goes through the C ABI.

Round 5: the oracle halves of the fixed-seed comparisons are recorded digests (tests/expect.py,
tests/golden/gpu_expect.json, made by the oracle alone in the build container); the live oracle runs on the
GPU box only where a digest is missing or differs."""

import hashlib
import json
import os
import threading

import numpy as np
import pytest

import bigint_oracle as BO

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FKEY = bytes(range(11, 43))


from conftest import oracle_threads as _threads  # noqa: E402


def h_ints(vals, nbytes=16):
    h = hashlib.sha256()
    for v in vals:
        h.update(int(v).to_bytes(nbytes, "little"))
    return h.hexdigest()


def _u128_ints(arr):
    flat = np.ascontiguousarray(arr).reshape(-1, 2)
    return [int(lo) | (int(hi) << 64) for lo, hi in flat]


def _mixed_inputs(o, sk, p, count, seed):
    """`count` LWE input pairs: encryptions of all four bit pairs (twice) up front, then uniformly
    random words (every rotation amount, not only valid encryptions)."""
    rng = np.random.default_rng(seed)
    a1 = rng.integers(0, p.r, size=(count, p.n), dtype=np.uint64)
    a2 = rng.integers(0, p.r, size=(count, p.n), dtype=np.uint64)
    b1 = rng.integers(0, p.r, size=count, dtype=np.uint64)
    b2 = rng.integers(0, p.r, size=count, dtype=np.uint64)
    bits = np.array([0, 0, 0, 1, 1, 0, 1, 1] * 2, dtype=np.uint8)
    k = min(len(bits) // 2, count)
    ea, eb = o.lwe_encrypt_bits(sk, bits[:2 * k], seed + 1)
    a1[:k], b1[:k], a2[:k], b2[:k] = ea[0::2], eb[0::2], ea[1::2], eb[1::2]
    return bits[:2 * k], a1, b1, a2, b2


@pytest.fixture(scope="module")
def p1024six(S, oc, exp):
    """Params(1024) on a default ctx (a basis per flatten mode) with the key of
    tests/golden/p1024*.json; the oracle's key in the NTT domain, made on first use (on the GPU box: only
    when a recorded expectation is missing or differs)."""
    params = S.Params(1024)
    o = oc.Oracle.from_params(params)
    sk = o.private_key(21)
    khat = exp.lazy(lambda: o.key_transform(o.bootstrap_key(sk, 22, threads=_threads()), threads=_threads()))
    eng = exp.engine(S, params)
    if exp.live:
        assert len(eng.primes()) == 5
    eng.generate_key(sk, 22)                    # byte-identical to the oracle's key (test_gpu_golden.py)
    yield params, o, sk, khat, eng
    eng.close()


# ---- 1. the dual-basis Params(1024) ctx, deterministic mode (derived five-prime key) ----------------

FORMS4 = ["small-batch form", "throughput form"]


@pytest.mark.parametrize("form", FORMS4)
def test_dual_basis_ctx_deterministic_mode_vs_oracle(S, oc, exp, p1024six, form):
    """The deterministic mode of a ctx that also holds the randomised mode's six-prime basis: five
    primes, the key form derived on the device (k_key_derive) -- accumulators after k = 1, 2 and n
    iterations, raw LWEs mod Q and ModRed words of 4 bootstraps against the C restatement; the
    same again after a round trip through the randomised mode."""
    params, o, sk, khat, eng = p1024six
    eng.set_random_flatten(False)
    eng.set_small_batch_max(0 if form == "throughput form" else 24)
    eng.set_random_flatten(True, FKEY)          # there and back: the basis switch leaves nothing behind
    if exp.live:
        assert len(eng.primes()) == 6
    eng.set_random_flatten(False)
    if exp.live:
        assert eng.kernel_names() == ("k_extprod<13, 4, false>", "k_crt_lean<5, 3>") and len(eng.primes()) == 5
    bits, a1, b1, a2, b2 = _mixed_inputs(o, sk, params, 4, 61)
    T = _threads()
    # (both kernel forms must give the oracle's bytes: one tag serves both parametrisations)
    for it in (1, 2, params.n):
        exp.check("r4.dual.det.acc%d" % it, eng.debug_accumulators(a1, b1, a2, b2, it),
                  lambda: o.bootstrap_batch(khat(), a1, b1, a2, b2, n_iters=it, want_acc=True, opt=True, threads=T)[1],
                  "accumulators after %d" % it)
    exp.check("r4.dual.det.raw", eng.bootstrap_batch(a1, b1, a2, b2, raw=True),
              lambda: o.bootstrap_batch(khat(), a1, b1, a2, b2, raw=True, opt=True, threads=T))
    out = exp.check("r4.dual.det.out", eng.bootstrap_batch(a1, b1, a2, b2),
                    lambda: o.bootstrap_batch(khat(), a1, b1, a2, b2, opt=True, threads=T))
    y1, y2 = bits[0::2], bits[1::2]
    for g, fn in enumerate((np.bitwise_and, np.bitwise_or, np.bitwise_xor)):
        assert np.array_equal(o.lwe_decrypt_bits(sk, out[:, g, :params.n], out[:, g, params.n]), fn(y1, y2))
    eng.set_small_batch_max(24)


def test_dual_basis_ctx_deterministic_full_batch_4096(S, oc, exp, p1024six):
    """The dual-basis ctx in its deterministic mode at the full batch: 8 distinct oracle-verified
    input pairs tiled over 4096 rows in a shuffled order, every output word pinned; equal to the
    bytes of a ctx created with SGFHE_CTX_DETERMINISTIC_ONLY, whose five-prime key form comes
    straight from the transform instead of the derivation."""
    params, o, sk, khat, eng = p1024six
    eng.set_random_flatten(False)
    bits, a1, b1, a2, b2 = _mixed_inputs(o, sk, params, 8, 62)
    idx = np.random.default_rng(63).permutation(np.repeat(np.arange(8), 512))
    out = exp.check("r4.dual.det.full4096", eng.bootstrap_batch(a1[idx], b1[idx], a2[idx], b2[idx]),
                    lambda: o.bootstrap_batch(khat(), a1, b1, a2, b2, opt=True, threads=_threads())[idx])
    if exp.live:
        five = S.Engine(params, deterministic_only=True)
        assert len(five.primes()) == 5
        five.generate_key(sk, 22)
        assert five.bootstrap_batch(a1[idx[:64]], b1[idx[:64]], a2[idx[:64]], b2[idx[:64]]).tobytes() == \
            out[:64].tobytes()
        five.close()


# ---- 1. the dual-basis Params(1024) ctx, randomised mode (six primes) ---------------------------------

def test_dual_basis_ctx_random_mode_accumulators_vs_oracle(S, oc, exp, p1024six):
    """bootstrap(bkey, rng, ...) at the reference's own Params(1024): accumulators after k = 1, 2 and
    n iterations, raw LWEs and ModRed words of 4 bootstraps bit for bit against the C restatement of
    utils.jl:198-241 on the same ChaCha8 stream (small-batch and throughput kernels)."""
    params, o, sk, khat, eng = p1024six
    bits, a1, b1, a2, b2 = _mixed_inputs(o, sk, params, 4, 64)
    T = _threads()
    for form in (24, 0):
        eng.set_small_batch_max(form)
        eng.set_random_flatten(True, FKEY)                      # call number back to 0
        if exp.live:
            assert eng.kernel_names()[1] == "k_crt_lean_rnd<6, 3, false>" and len(eng.primes()) == 6
        call = 0
        for it in (1, 2, params.n):     # (both kernel forms against the same recorded oracle bytes)
            exp.check("r4.dual.rnd.acc%d" % it, eng.debug_accumulators(a1, b1, a2, b2, it),
                      lambda: o.bootstrap_batch(khat(), a1, b1, a2, b2, n_iters=it, want_acc=True, opt=True,
                                                threads=T, rnd=(FKEY, call))[1], repr((form, it)))
            call += 1
        exp.check("r4.dual.rnd.raw", eng.bootstrap_batch(a1, b1, a2, b2, raw=True),
                  lambda: o.bootstrap_batch(khat(), a1, b1, a2, b2, raw=True, opt=True, threads=T, rnd=(FKEY, call)))
        call += 1
        out = exp.check("r4.dual.rnd.out", eng.bootstrap_batch(a1, b1, a2, b2),
                        lambda: o.bootstrap_batch(khat(), a1, b1, a2, b2, opt=True, threads=T, rnd=(FKEY, call)))
        y1, y2 = bits[0::2], bits[1::2]
        for g, fn in enumerate((np.bitwise_and, np.bitwise_or, np.bitwise_xor)):
            assert np.array_equal(o.lwe_decrypt_bits(sk, out[:, g, :params.n], out[:, g, params.n]), fn(y1, y2))
    eng.set_small_batch_max(24)
    eng.set_random_flatten(False)


def test_dual_basis_ctx_random_mode_full_batch_4096(S, oc, exp, p1024six):
    """The workload of `bench.py --flatten random` at its ring and batch.  In the randomised mode
    every row of a call has its own draws (counter word = its index in the call), so tiling inputs
    does not tile outputs: 8 distinct input pairs are tiled over the 4096 rows, and 96 rows -- the
    first and last row of every 256-row chunk of both lanes plus 64 random positions -- are pinned
    word for word to the C restatement run at exactly those stream indices; every one of the 4096
    rows built from an encryption pair must decrypt to its gate values."""
    params, o, sk, khat, eng = p1024six
    n = params.n
    bits, a1, b1, a2, b2 = _mixed_inputs(o, sk, params, 8, 65)
    idx = np.random.default_rng(66).permutation(np.repeat(np.arange(8), 512))
    eng.set_random_flatten(True, FKEY)
    eng.bootstrap_batch(a1[:2], b1[:2], a2[:2], b2[:2])                     # call 0: so that the big call is call 1
    out = eng.bootstrap_batch(a1[idx], b1[idx], a2[idx], b2[idx])           # call 1
    rows = sorted(set([c * 256 for c in range(16)] + [c * 256 + 255 for c in range(16)] +
                      [int(v) for v in np.random.default_rng(67).choice(4096, size=64, replace=False)]))
    rows = np.array(rows)
    src = idx[rows]
    pinned = exp.check("r4.dual.rnd.full4096.rows", out[rows] if exp.live else None,
                       lambda: o.bootstrap_batch(khat(), a1[src], b1[src], a2[src], b2[src], opt=True,
                                                 threads=_threads(), rnd=(FKEY, 1, rows.astype(np.uint32))))
    if not exp.live:
        return          # (recording: the decryption of all 4096 rows needs the engine's array)
    assert pinned.shape[0] == len(rows)
    # the same inputs at two different rows give different words (their draws differ) ...
    same = np.flatnonzero(idx == idx[0])
    assert not np.array_equal(out[same[0]], out[same[1]])
    # ... and all of them decrypt
    y1, y2 = bits[0::2], bits[1::2]
    for g, fn in enumerate((np.bitwise_and, np.bitwise_or, np.bitwise_xor)):
        dec = o.lwe_decrypt_bits(sk, out[:, g, :n], out[:, g, n])
        assert np.array_equal(dec, fn(y1, y2)[idx])
    eng.set_random_flatten(False)


def test_dual_basis_ctx_matches_big_integer_golden(S, oc, exp, p1024six):
    """tests/golden/p1024rnd.json, made by the literal big-integer restatement: accumulator hashes
    after 1, 2, 512 and 1024 iterations, raw and ModRed output hashes of the randomised bootstrap at
    Params(1024), as bootstrap 0 of call 0 and as bootstrap 5 of call 2 of the stream."""
    path = os.path.join(G, "p1024rnd.json")
    if not os.path.exists(path):
        pytest.skip("golden/p1024rnd.json not generated")
    if not exp.live:
        pytest.skip("engine against a committed fixture: nothing to record")
    d = json.load(open(path))
    params, o, sk, khat, eng = p1024six
    assert str(params.Q) == d["params"]["Q"] and d["key_seed"] == 22 and d["sk_seed"] == 21
    fkey = bytes.fromhex(d["flatten_key_hex"])
    n = params.n
    for case in d["cases"]:
        boot, call = case["boot"], case["call"]
        rows = boot + 1
        a1 = np.tile(np.array([case["lwe1"]["a"]], dtype=np.uint64), (rows, 1))
        a2 = np.tile(np.array([case["lwe2"]["a"]], dtype=np.uint64), (rows, 1))
        b1 = np.full(rows, case["lwe1"]["b"], dtype=np.uint64)
        b2 = np.full(rows, case["lwe2"]["b"], dtype=np.uint64)

        def at_call():                      # a fresh stream advanced to call number `call`
            eng.set_random_flatten(True, fkey)
            for _ in range(call):
                eng.debug_accumulators(a1[:1], b1[:1], a2[:1], b2[:1], 1)
        for k, (ha, hb) in case["acc_sha256_after"].items():
            at_call()
            acc = eng.debug_accumulators(a1, b1, a2, b2, int(k))
            assert [h_ints(_u128_ints(acc[boot, 0])), h_ints(_u128_ints(acc[boot, 1]))] == [ha, hb], k
        at_call()
        raw = eng.bootstrap_batch(a1, b1, a2, b2, raw=True)
        at_call()
        out = eng.bootstrap_batch(a1, b1, a2, b2)
        for g in range(3):
            assert h_ints(_u128_ints(raw[boot, g])) == case["raw_sha256"][g]
            assert h_ints([int(v) for v in out[boot, g]], 8) == case["out_sha256"][g]
            assert [int(v) for v in out[boot, g, :8]] + [int(out[boot, g, n])] == case["out_head"][g]
    eng.set_random_flatten(False)


# ---- 2. soak on distinct inputs, every full-size configuration ----------------------------------------

@pytest.mark.parametrize("name,count", [("params1024", 64), ("params512", 256), ("synth64", 32), ("rns2", 16)])
def test_soak_distinct_inputs_vs_oracle(S, oc, exp, name, count):
    """test/api.test.jl:45-83 widened: `count` DISTINCT bootstraps (uniformly random LWE words, so
    every rotation amount occurs, plus encryptions of all four bit pairs) tiled to a full batch so
    that the default schedule (two lanes of full chunks) runs; every output word of every copy
    against the C restatement; the encryption pairs decrypt to the truth table.  config 4 (rns2:
    composite Q = B Bp) goes through the restatement's RNS2Number mode (src/rns.jl).  Then the same
    batch in the randomised flatten mode (bootstrap(bkey, rng, ...)), 32 rows of it pinned (16 on the RNS ring,
    whose restatement is the slowest)."""
    import bench
    p = bench.make_params(S, name)
    T = _threads()
    if name == "rns2":
        B, Bp = bench.rns2_moduli(S)
        o = oc.Oracle.from_params(p, rns2=(B, Bp))
    else:
        o = oc.Oracle.from_params(p)
    sk = o.private_key(31)
    valid = name in ("params1024", "params512", "rns2")      # synth64 has no noise budget
    eng = exp.engine(S, p)
    if valid:
        eng.generate_key(sk, 32)
        bkey = exp.lazy(lambda: o.bootstrap_key(sk, 32, threads=T))
    else:
        rk = bench.random_key(p, 32)
        eng.upload_key(rk)
        bkey = exp.lazy(lambda: rk)
    bits, a1, b1, a2, b2 = _mixed_inputs(o, sk, p, count, 33)
    assert o.uses_ntt or o.uses_rns2
    # NTT-domain loop (for rns2: limb-wise, CRT per column; bit-identical to the reference-shaped loops:
    # test_oracle_properties.py)
    khat = exp.lazy(lambda: o.key_transform(bkey(), threads=T))
    full = 4096 if name != "params512" else 1024             # BASELINE.json's batch of each configuration
    idx = np.random.default_rng(35).permutation(np.resize(np.arange(count), full))
    out = exp.check("r4.soak.%s.det" % name, eng.bootstrap_batch(a1[idx], b1[idx], a2[idx], b2[idx]),
                    lambda: o.bootstrap_batch(khat(), a1, b1, a2, b2, threads=T, opt=True)[idx])
    assert out.shape == (full, 3, p.n + 1)
    if valid:
        k = len(bits) // 2
        y1, y2 = bits[0::2], bits[1::2]
        first = np.array([int(np.flatnonzero(idx == t)[0]) for t in range(k)])     # a copy of each encryption pair
        for g, fn in enumerate((np.bitwise_and, np.bitwise_or, np.bitwise_xor)):
            assert np.array_equal(o.lwe_decrypt_bits(sk, out[first, g, :p.n], out[first, g, p.n]), fn(y1, y2))
    # the randomised flatten on the same ring, key and batch (Params(1024) has its own tests above): every
    # row draws at its own index in the call, so 32 rows spread over the chunks of both lanes are pinned
    # word for word to the C restatement run at those stream indices
    if name != "params1024":
        eng.set_random_flatten(True, FKEY)
        out_r = eng.bootstrap_batch(a1[idx], b1[idx], a2[idx], b2[idx])           # call 0
        rows = np.unique(np.concatenate([[0, full - 1], np.random.default_rng(36).choice(full, size=14 if name == "rns2" else 30,
                                                                                      replace=False)]))
        src = idx[rows]
        pinned = exp.check("r4.soak.%s.rnd.rows" % name, out_r[rows] if exp.live else None,
                           lambda: o.bootstrap_batch(khat(), a1[src], b1[src], a2[src], b2[src], threads=T, opt=True,
                                                     rnd=(FKEY, 0, rows.astype(np.uint32))))
        assert not np.array_equal(pinned, out[rows])
        eng.set_random_flatten(False)
        if exp.live:
            assert np.array_equal(eng.bootstrap_batch(a1[idx[:8]], b1[idx[:8]], a2[idx[:8]], b2[idx[:8]]), out[:8])
    eng.close()


# ---- 3. one ctx, two streams / two threads -----------------------------------------------------------

def _device_case(S, oc, params, batch, seed):
    import torch
    o = oc.Oracle.from_params(params)
    sk = o.private_key(seed)
    eng = S.Engine(params)
    eng.generate_key(sk, seed + 1)
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    sets = []
    for _ in range(2):
        a1 = torch.randint(0, params.r, (batch, params.n), dtype=torch.int64, device="cuda", generator=g)
        a2 = torch.randint(0, params.r, (batch, params.n), dtype=torch.int64, device="cuda", generator=g)
        b1 = torch.randint(0, params.r, (batch,), dtype=torch.int64, device="cuda", generator=g)
        b2 = torch.randint(0, params.r, (batch,), dtype=torch.int64, device="cuda", generator=g)
        sets.append((a1, b1, a2, b2))
    torch.cuda.synchronize()
    return eng, sets


@pytest.mark.parametrize("batch", [8, 600])
def test_two_streams_on_one_ctx_equal_sequential_calls(S, oc, batch):
    """sgfhe_bootstrap_batch_device with caller streams: two calls on two torch streams without a
    host synchronisation in between -- they share the ctx's work buffers, so the engine orders them
    on the device -- give the bytes of the same two calls made one after the other on the ctx
    stream; a third call on the ctx stream right behind them too.  batch 8 takes the small-batch
    kernels, 600 the throughput kernels on two lanes (ragged last chunks)."""
    import torch
    params = S.Params(512)
    eng, sets = _device_case(S, oc, params, batch, 91)
    shape = (batch, 3, params.n + 1)
    seq = [torch.zeros(shape, dtype=torch.int64, device="cuda") for _ in range(2)]
    for (a1, b1, a2, b2), out in zip(sets, seq):
        eng.bootstrap_batch_device(a1.data_ptr(), b1.data_ptr(), a2.data_ptr(), b2.data_ptr(), batch, out.data_ptr())
        eng.sync()
    assert not torch.equal(seq[0], seq[1])
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for rep in range(3):
        par = [torch.full(shape, -1, dtype=torch.int64, device="cuda") for _ in range(3)]
        torch.cuda.synchronize()
        for (a1, b1, a2, b2), out, st in zip(sets, par, streams):
            eng.bootstrap_batch_device(a1.data_ptr(), b1.data_ptr(), a2.data_ptr(), b2.data_ptr(), batch,
                                       out.data_ptr(), stream=st.cuda_stream)
        a1, b1, a2, b2 = sets[0]
        eng.bootstrap_batch_device(a1.data_ptr(), b1.data_ptr(), a2.data_ptr(), b2.data_ptr(), batch,
                                   par[2].data_ptr())                      # NULL: the ctx stream
        # each caller stream sees its own result complete in stream order
        with torch.cuda.stream(streams[0]):
            c0 = par[0].clone()
        with torch.cuda.stream(streams[1]):
            c1 = par[1].clone()
        eng.sync()                                                         # waits for all three calls
        assert torch.equal(par[2], seq[0])
        streams[0].synchronize()
        streams[1].synchronize()
        assert torch.equal(c0, seq[0]) and torch.equal(c1, seq[1])
        assert torch.equal(par[0], seq[0]) and torch.equal(par[1], seq[1])
    # a host-pointer call right behind an asynchronous one on a caller stream
    a1, b1, a2, b2 = sets[1]
    out = torch.full(shape, -1, dtype=torch.int64, device="cuda")
    eng.bootstrap_batch_device(a1.data_ptr(), b1.data_ptr(), a2.data_ptr(), b2.data_ptr(), batch, out.data_ptr(),
                               stream=streams[0].cuda_stream)
    h = [t.cpu().numpy().view(np.uint64) for t in sets[0]]
    hout = eng.bootstrap_batch(h[0], h[1], h[2], h[3])
    streams[0].synchronize()
    assert np.array_equal(hout.view(np.int64), seq[0].cpu().numpy()) and torch.equal(out, seq[1])
    eng.close()


def test_two_threads_share_a_ctx_through_the_device_entry_point(S, oc):
    """Two host threads, each with its own stream, issue asynchronous calls on one shared ctx in a
    loop (the thread-sharing promise of include/sgfhe_hip.h for sgfhe_bootstrap_batch_device)."""
    import torch
    params = S.Params(512)
    batch = 40
    eng, sets = _device_case(S, oc, params, batch, 95)
    shape = (batch, 3, params.n + 1)
    seq = []
    for a1, b1, a2, b2 in sets:
        out = torch.zeros(shape, dtype=torch.int64, device="cuda")
        eng.bootstrap_batch_device(a1.data_ptr(), b1.data_ptr(), a2.data_ptr(), b2.data_ptr(), batch, out.data_ptr())
        eng.sync()
        seq.append(out)
    errors = []

    def worker(i):
        try:
            torch.cuda.set_device(0)
            st = torch.cuda.Stream()
            a1, b1, a2, b2 = sets[i]
            for _ in range(6):
                out = torch.full(shape, -1, dtype=torch.int64, device="cuda")
                torch.cuda.current_stream().synchronize()
                eng.bootstrap_batch_device(a1.data_ptr(), b1.data_ptr(), a2.data_ptr(), b2.data_ptr(), batch,
                                           out.data_ptr(), stream=st.cuda_stream)
                st.synchronize()
                if not torch.equal(out, seq[i]):
                    errors.append("thread %d: wrong bytes" % i)
        except Exception as e:                                              # noqa: BLE001
            errors.append(repr(e))
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert errors == []
    eng.close()


# ---- two-limb moduli with a base far above sqrt(Q) in the randomised mode ----------------

@pytest.mark.parametrize("bbits,expect", [(31, "k_crt_acc<4>"), (29, "k_crt_lean_rnd<4, 2, false>")])
def test_random_flatten_two_limb_modulus_wide_base(S, oc, bbits, expect):
    """Q ~ 2^57 (two 29-bit limbs in k_crt_lean) with B = 2^31 > sqrt(Q): the old hi digit plus the
    draws reaches 7 B >= 2^32, where the two-limb kernel would drop the h1 B1 2^61 term of hi B --
    the engine has to take the general kernel there (and the lean one at B = 2^29, where the term is
    zero).  Bit for bit against the C restatement, both flatten modes."""
    n, m = 8, 64
    Q = BO.find_modulus(2 * m, (1 << 57) - (1 << 40))
    B = 1 << bbits
    params = S.Params.custom(n, Q, B)
    o = oc.Oracle.from_params(params)
    sk = o.private_key(5)
    bkey = o.bootstrap_key(sk, 6, noise=2)
    eng = S.Engine(params)
    eng.upload_key(bkey)
    eng.set_small_batch_max(0)
    rng = np.random.default_rng(7)
    batch = 24
    a1 = rng.integers(0, params.r, size=(batch, n), dtype=np.uint64)
    a2 = rng.integers(0, params.r, size=(batch, n), dtype=np.uint64)
    b1 = rng.integers(0, params.r, size=batch, dtype=np.uint64)
    b2 = rng.integers(0, params.r, size=batch, dtype=np.uint64)
    assert np.array_equal(eng.bootstrap_batch(a1, b1, a2, b2, raw=True),
                          o.bootstrap_batch(bkey, a1, b1, a2, b2, raw=True))
    eng.set_random_flatten(True, FKEY)
    assert eng.kernel_names()[1] == expect
    for it, call in ((1, 0), (2, 1), (n, 2)):
        _, ref = o.bootstrap_batch(bkey, a1, b1, a2, b2, n_iters=it, want_acc=True, rnd=(FKEY, call))
        assert np.array_equal(eng.debug_accumulators(a1, b1, a2, b2, it), ref), it
    assert np.array_equal(eng.bootstrap_batch(a1, b1, a2, b2, raw=True),
                          o.bootstrap_batch(bkey, a1, b1, a2, b2, raw=True, rnd=(FKEY, 3)))
    assert np.array_equal(eng.bootstrap_batch(a1, b1, a2, b2),
                          o.bootstrap_batch(bkey, a1, b1, a2, b2, rnd=(FKEY, 4)))
    eng.close()


def test_host_staging_can_be_released(S, oc):
    """sgfhe_release_host_staging: the page-locked and device staging buffers of the host-pointer
    entry point are freed and come back on the next call; results unchanged; batches that take the
    pipelined path with one lane, two lanes and a ragged tail."""
    params = S.Params(64)
    o = oc.Oracle.from_params(params)
    sk = o.private_key(1)
    bkey = o.bootstrap_key(sk, 2)
    eng = S.Engine(params)
    eng.upload_key(bkey)
    rng = np.random.default_rng(3)
    for batch in (1, 100, 1030):
        a1 = rng.integers(0, params.r, size=(batch, params.n), dtype=np.uint64)
        a2 = rng.integers(0, params.r, size=(batch, params.n), dtype=np.uint64)
        b1 = rng.integers(0, params.r, size=batch, dtype=np.uint64)
        b2 = rng.integers(0, params.r, size=batch, dtype=np.uint64)
        ref = o.bootstrap_batch(bkey, a1, b1, a2, b2, threads=_threads())
        assert np.array_equal(eng.bootstrap_batch(a1, b1, a2, b2), ref)
        eng.release_host_staging()
        assert np.array_equal(eng.bootstrap_batch(a1, b1, a2, b2), ref)
        assert np.array_equal(eng.bootstrap_batch(a1, b1, a2, b2, raw=True),
                              o.bootstrap_batch(bkey, a1, b1, a2, b2, raw=True, threads=_threads()))
    eng.close()


# ---- the latency form at its true sizes (round 4: no padding to 8 gates, one coefficient per CRT thread) ----

@pytest.mark.parametrize("ring", ["params64", "synthetic m = 256", "params512", "params1024"])
def test_latency_form_calls_of_one_to_nine_gates(S, oc, exp, ring):
    """A call of g gates launches g gates' workgroups (up to round 3 the small-batch grids were padded
    to a multiple of 8, so the literal drop-in call -- one gate -- did eight gates' work); up to 8
    gates the CRT kernel takes one coefficient per thread (k_crt_lean1); and for m >= 4096 calls of up
    to 7 gates cut every transform across four workgroups (k_fwd_quarter / k_inv_quarter /
    k_crt_lean1q: the first two forward stages and the last two inverse stages outside the
    quarter-size transforms, (x^j - 1) applied in the NTT domain).  Sizes from 1 to 9 against the C
    restatement, raw and ModRed, accumulators after 1 and 2 iterations; the one-workgroup transforms
    (SGFHE_SMALL_SPLIT=0) and the padded launches of rounds 1-3 (SGFHE_SMALL_PADDED=1) give the same bytes.
    From 7 gates per chain (m = 4096, 8192) the two transform kernels of the quarter form are one launch
    (k_ext_quarter: sizes 7, 8 and the 7 + 6 of a 13-gate call here; SGFHE_SMALL_FUSED=1 takes it from one gate)."""
    if ring == "params64":
        params, noise = S.Params(64), None
    elif ring in ("params512", "params1024"):
        params, noise = S.Params(int(ring[6:])), None
    else:
        n = 32
        params, noise = S.Params.custom(n, BO.find_modulus(16 * n, 1 << 50), 1 << 26), 2
    o = oc.Oracle.from_params(params)
    sk = o.private_key(7)
    T = _threads()
    # Params(1024) keeps the UPLOAD of a reference-shaped key (what julia/SGFHEHip.jl's HipBootstrapKey(bkey)
    # does) in the suite; the other full-size ring is keyed on the device from the same seed (byte-identical)
    big = ring in ("params512", "params1024")
    bkey = exp.lazy(lambda: o.bootstrap_key(sk, 8, noise=noise, threads=T))
    khat = exp.lazy(lambda: o.key_transform(bkey(), threads=T))
    eng = exp.engine(S, params)
    if ring == "params512":
        eng.generate_key(sk, 8)
    elif exp.live:
        eng.upload_key(bkey())
    tag = "r4.latency.%s." % ring.replace(" ", "")
    sizes = {"params512": (1, 3, 4, 5, 7, 8, 13), "params1024": (1, 7, 8)}.get(ring, (1, 2, 3, 4, 5, 8, 9))
    bits, a1, b1, a2, b2 = _mixed_inputs(o, sk, params, max(sizes), 9)   # (round 4 made 9 rows: its "13" ran 9 gates)
    # (an oracle row depends on its own inputs only: one run over all rows serves every size)
    all_out = exp.lazy(lambda: o.bootstrap_batch(khat(), a1, b1, a2, b2, opt=True, threads=T))
    for g in sizes:
        sl = slice(0, g)
        exp.check(tag + "out%d" % g, eng.bootstrap_batch(a1[sl], b1[sl], a2[sl], b2[sl]), lambda: all_out()[sl], repr(g))
        if not big or g == 7:
            exp.check(tag + "raw%d" % g, eng.bootstrap_batch(a1[sl], b1[sl], a2[sl], b2[sl], raw=True),
                      lambda: o.bootstrap_batch(khat(), a1[sl], b1[sl], a2[sl], b2[sl], raw=True, opt=True, threads=T),
                      repr(g))
            for it in (1, 2):
                exp.check(tag + "acc%d.%d" % (g, it), eng.debug_accumulators(a1[sl], b1[sl], a2[sl], b2[sl], it),
                          lambda: o.bootstrap_batch(khat(), a1[sl], b1[sl], a2[sl], b2[sl], n_iters=it, want_acc=True,
                                                    opt=True, threads=T)[1], repr((g, it)))
    # the randomised flatten through the same forms (the quarter form serves both modes)
    eng.set_random_flatten(True, FKEY)
    for call, g in enumerate((1, 3, 7, 9) if ring != "params1024" else (1, 7)):
        sl = slice(0, g)
        exp.check(tag + "rnd%d" % g, eng.bootstrap_batch(a1[sl], b1[sl], a2[sl], b2[sl]),
                  lambda: o.bootstrap_batch(khat(), a1[sl], b1[sl], a2[sl], b2[sl], opt=True, threads=T,
                                            rnd=(FKEY, call)), repr(("random", g)))
    eng.close()
    for knob in ("SGFHE_SMALL_PADDED", "SGFHE_SMALL_SPLIT", "SGFHE_SMALL_FUSED"):
        os.environ[knob] = "0" if knob == "SGFHE_SMALL_SPLIT" else "1"
        try:
            old = exp.engine(S, params)
        finally:
            del os.environ[knob]
        if noise is None:
            old.generate_key(sk, 8)
        elif exp.live:
            old.upload_key(bkey())
        exp.check(tag + "out3", old.bootstrap_batch(a1[:3], b1[:3], a2[:3], b2[:3]), lambda: all_out()[:3], knob)
        old.close()


def test_latency_form_boundaries_at_params512(S, oc, exp):
    """The call sizes at which the engine changes its form of the k-loop, Params(512) (m = 4096: every latency
    form exists), both flatten modes, every output word against the C restatement: 6 gates (the last size of
    the two-launch quarter form), 12 (the largest single chain of k_ext_quarter launches on five primes), 14
    (7 + 7 on the two lanes), 24 (12 + 12, the largest call in the latency form), 25 (13 + 12 ... the first
    call in the throughput form, an odd size) -- and with the randomised flatten 10, 11 (the fused chain's limit
    on the basis that mode runs on), 13, 24, 25, each call on its own stream position (call counter)."""
    params = S.Params(512)
    o = oc.Oracle.from_params(params)
    sk = o.private_key(31)
    T = _threads()
    khat = exp.lazy(lambda: o.key_transform(o.bootstrap_key(sk, 32, threads=T), threads=T))
    eng = exp.engine(S, params)
    eng.generate_key(sk, 32)
    bits, a1, b1, a2, b2 = _mixed_inputs(o, sk, params, 25, 33)
    all_out = exp.lazy(lambda: o.bootstrap_batch(khat(), a1, b1, a2, b2, opt=True, threads=T))
    for g in (6, 12, 14, 24, 25):
        sl = slice(25 - g, 25)                       # the tail: other rows than the sizes before
        out = exp.check("r4.bound512.out%d" % g, eng.bootstrap_batch(a1[sl], b1[sl], a2[sl], b2[sl]),
                        lambda: all_out()[sl], repr(g))
    y1, y2 = bits[0::2], bits[1::2]
    dec = o.lwe_decrypt_bits(sk, out[:len(y1), 2, :params.n], out[:len(y1), 2, params.n])     # g = 25: all rows
    assert np.array_equal(dec, y1 ^ y2)
    eng.set_random_flatten(True, FKEY)
    for call, g in enumerate((10, 11, 13, 24, 25)):
        sl = slice(0, g)
        exp.check("r4.bound512.rnd%d" % g, eng.bootstrap_batch(a1[sl], b1[sl], a2[sl], b2[sl]),
                  lambda: o.bootstrap_batch(khat(), a1[sl], b1[sl], a2[sl], b2[sl], opt=True, threads=T,
                                            rnd=(FKEY, call)), repr(("random", g)))
    eng.close()
