"""Concurrency contract of the C ABI (include/sgfhe_hip.h): a ctx is bound to a device, any number
of ctxs may share a device, and every entry point locks its ctx, so host threads may share one.
The reference call is pure (src/fhe.jl:608-621); a server embedding the engine will call it from
several threads."""

import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _inputs(params, batch, seed):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, params.r, size=(batch, params.n), dtype=np.uint64),
            rng.integers(0, params.r, size=batch, dtype=np.uint64),
            rng.integers(0, params.r, size=(batch, params.n), dtype=np.uint64),
            rng.integers(0, params.r, size=batch, dtype=np.uint64))


def _run_threads(jobs):
    """jobs: list of callables; each runs in its own thread, all released together."""
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


def test_two_ctxs_two_threads_on_one_device(S):
    """Two engines on device 0 (different keys), each driven by its own host thread at the same
    time, large and small batches interleaved: byte-equal to the same calls made one after the
    other."""
    params = S.Params(512)
    engs, work = [], []
    for t in range(2):
        eng = S.Engine(params, device=0)
        eng.generate_key(np.random.default_rng(100 + t).integers(0, 2, size=params.n, dtype=np.uint64), 200 + t)
        engs.append(eng)
        work.append([_inputs(params, b, 300 + 10 * t + i) for i, b in enumerate((40, 3, 70))])
    ref = [[eng.bootstrap_batch(*w) for w in ws] for eng, ws in zip(engs, work)]
    got = _run_threads([lambda e=eng, ws=ws: [e.bootstrap_batch(*w) for w in ws]
                        for eng, ws in zip(engs, work)])
    for t in range(2):
        for a, b in zip(got[t], ref[t]):
            assert a.tobytes() == b.tobytes()
    assert ref[0][0].tobytes() != ref[1][0].tobytes()          # the two keys really differ
    for eng in engs:
        eng.close()


def test_one_ctx_shared_by_threads(S):
    """Four threads call one ctx at once (the per-ctx lock serialises them): every call returns
    what it returns alone, in both entry points (host buffers, and the debug hook that shares the
    lane buffers)."""
    params = S.Params(64)
    eng = S.Engine(params, device=0)
    eng.generate_key(np.random.default_rng(1).integers(0, 2, size=params.n, dtype=np.uint64), 2)
    work = [_inputs(params, b, 400 + i) for i, b in enumerate((5, 300, 17, 64))]
    ref = [eng.bootstrap_batch(*w) for w in work]
    ref_acc = eng.debug_accumulators(*work[0], 3)
    jobs = [lambda w=w: [eng.bootstrap_batch(*w) for _ in range(3)] for w in work]
    jobs.append(lambda: [eng.debug_accumulators(*work[0], 3) for _ in range(3)])
    got = _run_threads(jobs)
    for i, r in enumerate(ref):
        for g in got[i]:
            assert g.tobytes() == r.tobytes()
    for g in got[-1]:
        assert g.tobytes() == ref_acc.tobytes()
    eng.close()
