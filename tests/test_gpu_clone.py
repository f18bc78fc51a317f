"""sgfhe_ctx_clone (include/sgfhe_hip.h, ABI revision 7): independent callers on ONE key.

The reference call is pure (src/fhe.jl:608-621): any number of Julia tasks may run
bootstrap(bkey, ...) on one BootstrapKey side by side.  A ctx serialises its calls (it owns the work
buffers they use); a clone shares the device key and constants and owns its buffers and streams, so
calls on different clones overlap on the device -- and must still give, each, the bytes the same call
gives alone."""

import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FKEY = bytes(range(50, 82))


def _inputs(params, batch, seed):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, params.r, size=(batch, params.n), dtype=np.uint64),
            rng.integers(0, params.r, size=batch, dtype=np.uint64),
            rng.integers(0, params.r, size=(batch, params.n), dtype=np.uint64),
            rng.integers(0, params.r, size=batch, dtype=np.uint64))


def _run_threads(jobs):
    gate = threading.Barrier(len(jobs))
    out, err = [None] * len(jobs), []

    def body(i):
        try:
            gate.wait()
            out[i] = jobs[i]()
        except Exception as exc:                 # surfaced in the main thread
            err.append(exc)
    ts = [threading.Thread(target=body, args=(i,)) for i in range(len(jobs))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not err, err
    return out


@pytest.mark.parametrize("n", [512, 1024])
def test_eight_threads_single_gate_calls_on_eight_clones(S, gpu_keys, n):
    """8 host threads, each with its own clone, each making single-gate calls (the
    literal drop-in signature) at the same time: every call returns the bytes of the same call made
    alone on the ctx the clones were taken from."""
    params, _o, _sk, eng = gpu_keys.engine(n)
    eng.set_random_flatten(False)
    calls = 6 if n == 512 else 4
    work = [[_inputs(params, 1, 7000 + 16 * t + i) for i in range(calls)] for t in range(8)]
    ref = [[eng.bootstrap_batch(*w) for w in ws] for ws in work]          # sequential, on the parent
    clones = [eng.clone() for _ in range(8)]
    try:
        for gather in (True, False):
            # gathered: calls that arrive together run as ONE launch chain on the leader's ctx (sgfhe_set_coalesce,
            # on by default for ctxs that share a key); not gathered: every clone runs its own chain on its own stream
            eng.set_coalesce(gather)
            eng.coalesce_stats(reset=True)
            got = _run_threads([lambda e=e, ws=ws: [e.bootstrap_batch(*w) for w in ws] for e, ws in zip(clones, work)])
            for t in range(8):
                for a, b in zip(got[t], ref[t]):
                    assert a.tobytes() == b.tobytes(), (gather, t)
            st = clones[3].coalesce_stats()                  # the statistics belong to the shared key
            if gather:
                # every request went through a combined call, and with eight threads released together on calls
                # of several milliseconds some call carried more than one request
                assert st["requests"] == 8 * calls and st["gates"] == 8 * calls and st["calls"] < st["requests"]
                assert 2 <= st["max_requests"] <= 8
            else:
                assert st["requests"] == 0
        assert ref[0][0].tobytes() != ref[1][0].tobytes()
    finally:
        eng.set_coalesce(True)
        for e in clones:
            e.close()


@pytest.mark.parametrize("n", [64, 512, 2048])
def test_gathered_randomised_calls_keep_every_callers_own_draw_stream(S, oc, gpu_keys, n):
    """bootstrap(hkey, rng, ...) from eight tasks at once (the reference's documented call, README.md:24): every
    clone has its own flatten key and call counter, and a gathered call draws for every row from the stream of the
    ctx the row came in on (kernels.h RndRow) -- so each caller gets the bytes of the same calls made alone, call
    after call, whatever was gathered with them; one caller's first call is also held against the C restatement of
    src/utils.jl:198-241 on that caller's stream.  Calls of 1, 3, 12 and 2 gates per caller: gathered, the
    small ones run the latency form's ROWS kernels (k_crt_lean_rnd1), the 12-gate ones the throughput form's
    (k_extprod + k_crt_lean_rnd<.., ROWS>: 25 gates or more per round); Params(64) (m = 512: a wave of the CRT
    kernel spans several rows), Params(512), and Params(2048) (the WIDE instantiations: third digit plane)."""
    params, o, sk, eng = gpu_keys.engine(n)
    eng.set_random_flatten(False)
    sizes = (1, 3, 12, 2) if n != 2048 else (1, 4)
    work = [[_inputs(params, g, 7300 + 16 * t + i) for i, g in enumerate(sizes)] for t in range(8)]
    keys = [bytes((13 * t + i) & 0xFF for i in range(32)) for t in range(8)]
    alone = eng.clone()
    eng.set_coalesce(False)                       # the reference streams: nothing gathered
    ref = []
    for t in range(8):
        alone.set_random_flatten(True, keys[t])   # call counter 0
        ref.append([alone.bootstrap_batch(*w) for w in work[t]])          # calls 0, 1, 2, 3 of stream t
    alone.close()
    if n != 2048:       # (Params(2048)'s randomised mode is held against the restatement in test_gpu_round5.py)
        khat = gpu_keys.khat(n)
        assert np.array_equal(ref[5][1], o.bootstrap_batch(khat, *work[5][1], opt=True, rnd=(keys[5], 1)))
    clones = [eng.clone() for _ in range(8)]
    try:
        eng.set_coalesce(True)
        eng.coalesce_stats(reset=True)

        def job(t):
            clones[t].set_random_flatten(True, keys[t])
            return [clones[t].bootstrap_batch(*w) for w in work[t]]
        got = _run_threads([lambda t=t: job(t) for t in range(8)])
        for t in range(8):
            for a, b in zip(got[t], ref[t]):
                assert a.tobytes() == b.tobytes(), t
        st = eng.coalesce_stats()
        assert st["requests"] == 8 * len(sizes) and st["max_requests"] >= 2
        assert ref[0][0].tobytes() != ref[1][0].tobytes()
    finally:
        for e in clones:
            e.close()


def test_clones_keep_their_own_mode_buffers_and_knobs(S, oc, gpu_keys):
    """Clones beside the parent, all busy at once with different call sizes (latency form, two chains,
    throughput form), one of them in the randomised flatten mode on its own ChaCha key: each stream of
    results equals the oracle-checked bytes of the parent alone (deterministic) and the C restatement's
    (randomised, src/utils.jl:198-241)."""
    params, o, sk, eng = gpu_keys.engine(512)
    khat = gpu_keys.khat(512)
    eng.set_random_flatten(False)
    sizes = (1, 9, 40, 3, 130)
    work = [_inputs(params, b, 7100 + i) for i, b in enumerate(sizes)]
    ref = [eng.bootstrap_batch(*w) for w in work]
    assert np.array_equal(ref[1], o.bootstrap_batch(khat, *work[1], opt=True))      # the parent against the oracle
    a, b, c = eng.clone(), eng.clone(), eng.clone()
    try:
        c.set_random_flatten(True, FKEY)                     # this clone only
        b.set_lanes(1)
        assert eng.kernel_names()[1].startswith("k_crt_lean<") and c.kernel_names()[1].startswith("k_crt_lean_rnd<")
        rnd_ref = [o.bootstrap_batch(khat, *work[i], rnd=(FKEY, i), opt=True) for i in (0, 1)]
        got = _run_threads([
            lambda: [eng.bootstrap_batch(*w) for w in work],
            lambda: [a.bootstrap_batch(*w) for w in reversed(work)],
            lambda: [b.bootstrap_batch(*w) for w in work],
            lambda: [c.bootstrap_batch(*work[i]) for i in (0, 1)],
        ])
        for g, r in zip(got[0], ref):
            assert g.tobytes() == r.tobytes()
        for g, r in zip(got[1], reversed(ref)):
            assert g.tobytes() == r.tobytes()
        for g, r in zip(got[2], ref):
            assert g.tobytes() == r.tobytes()
        for g, r in zip(got[3], rnd_ref):
            assert np.array_equal(g, r)
        # the parent's mode was not touched by the clone's
        assert eng.bootstrap_batch(*work[0]).tobytes() == ref[0].tobytes()
    finally:
        for e in (a, b, c):
            e.close()


def test_shared_key_is_read_only_and_outlives_the_parent(S):
    """While clones exist every key writer on a sharer is refused (the others may be reading the key on
    the device); the blob export is not.  The shared memory goes with the LAST sharer: a clone works
    after the ctx it came from is destroyed, and a ctx whose clones are gone takes a new key."""
    params = S.Params(64)
    rng = np.random.default_rng(5)
    sk = rng.integers(0, 2, size=params.n, dtype=np.uint64)
    eng = S.Engine(params)
    with pytest.raises(S.SgfheError) as ei:                 # nothing to share yet
        eng.clone()
    assert ei.value.code == -5
    eng.generate_key(sk, 9)
    w = _inputs(params, 5, 1)
    ref = eng.bootstrap_batch(*w)
    cl = eng.clone()
    cl2 = cl.clone()                                        # a clone of a clone shares the same block
    for e in (eng, cl, cl2):
        with pytest.raises(S.SgfheError) as ei:
            e.generate_key(sk, 10)
        assert ei.value.code == -1 and "shared" in str(ei.value)
        assert e.bootstrap_batch(*w).tobytes() == ref.tobytes()    # the refused call changed nothing
    import torch
    blob = torch.empty(eng.key_device_form_bytes(), dtype=torch.uint8, device="cuda:0")
    cl.export_key_device_form(blob.data_ptr())              # reading the shared key is allowed
    with pytest.raises(S.SgfheError):
        cl2.import_key_device_form(blob.data_ptr())
    eng.close()                                             # the parent first
    assert cl.bootstrap_batch(*w).tobytes() == ref.tobytes()
    cl2.close()
    cl.generate_key(sk, 10)                                 # sole owner now: writable again
    other = cl.bootstrap_batch(*w)
    assert other.tobytes() != ref.tobytes()
    cl.import_key_device_form(blob.data_ptr())
    assert cl.bootstrap_batch(*w).tobytes() == ref.tobytes()
    cl.close()
