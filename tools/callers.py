#!/usr/bin/env python3
"""Independent callers on one key (sgfhe_ctx_clone, include/sgfhe_hip.h): aggregate gates per second with
1, 2, 4, 8, 16 host threads, each making calls of `--gates` gates through the drop-in host-pointer entry
point on a clone of its own, against the same threads sharing ONE ctx (whose calls the ctx serialises).

    python tools/callers.py --n 1024 --gates 1 --callers 1,2,4,8,16 --seconds 3

One line per caller count: calls/s, gates/s, ratio to one caller, mean call latency.  The results of every
thread are compared with the same calls made alone (bit-exact or the run aborts).  VERDICT r4 item 3;
the reference call is pure (src/fhe.jl:608-621), so a Julia host may run it from many tasks."""

import argparse
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgfhe_jl_amd as S  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--gates", type=int, default=1)
    ap.add_argument("--callers", default="1,2,4,8,16")
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--random", action="store_true", help="bootstrap(bkey, rng, ...): the randomised flatten")
    ap.add_argument("--shared", action="store_true", help="also time the same threads on ONE ctx")
    args = ap.parse_args()
    params = S.Params(args.n)
    rng = np.random.default_rng(1)
    eng = S.Engine(params)
    eng.generate_key(rng.integers(0, 2, size=params.n, dtype=np.uint64), 2)
    counts = [int(x) for x in args.callers.split(",")]
    nmax = max(counts)
    work = [(rng.integers(0, params.r, size=(args.gates, params.n), dtype=np.uint64),
             rng.integers(0, params.r, size=args.gates, dtype=np.uint64),
             rng.integers(0, params.r, size=(args.gates, params.n), dtype=np.uint64),
             rng.integers(0, params.r, size=args.gates, dtype=np.uint64)) for _ in range(nmax)]
    if args.random:
        eng.set_random_flatten(True, 5)
    ref = []
    for w in work:                       # the same calls alone (call 0 of the stream in the randomised mode)
        if args.random:
            eng.set_random_flatten(True, 5)
        ref.append(eng.bootstrap_batch(*w))
    clones = [eng.clone() for _ in range(nmax)]
    print("Params(%d), calls of %d gate(s), %s flatten, %.1f s per point, build %s" %
          (args.n, args.gates, "randomised" if args.random else "deterministic", args.seconds, eng.build_id()))

    def run(engines, k):
        """k threads, thread t on engines[t]; returns (calls, wall seconds)."""
        stop = time.perf_counter() + args.seconds
        calls = [0] * k
        bad = []
        gate = threading.Barrier(k + 1)

        def body(t):
            e, w = engines[t], work[t]
            out = np.zeros_like(ref[t])
            if args.random:
                e.set_random_flatten(True, 5)
            e.bootstrap_batch(*w, out=out)            # warm: buffers, first-touch
            gate.wait()
            while time.perf_counter() < stop:
                if args.random:
                    e.set_random_flatten(True, 5)     # call counter back to 0: same draws as the reference call
                e.bootstrap_batch(*w, out=out)
                calls[t] += 1
                if calls[t] % 16 == 1 and out.tobytes() != ref[t].tobytes():
                    bad.append(t)
                    return
        ts = [threading.Thread(target=body, args=(t,)) for t in range(k)]
        for t in ts:
            t.start()
        gate.wait()
        t0 = time.perf_counter()
        stop = t0 + args.seconds
        for t in ts:
            t.join()
        dt = time.perf_counter() - t0
        if bad:
            raise SystemExit("thread(s) %r returned bytes that differ from the call made alone" % bad)
        return sum(calls), dt

    base = None
    for k in counts:
        c, dt = run(clones, k)
        rate = c * args.gates / dt
        base = base or rate
        line = "callers %2d (one clone each): %7.1f calls/s  %8.1f gates/s  x%.2f  %.2f ms per call" % (
            k, c / dt, rate, rate / base, 1e3 * dt * k / max(c, 1))
        if args.shared and k > 1:
            c1, dt1 = run([eng] * k, k)
            line += "   | sharing one ctx: %8.1f gates/s" % (c1 * args.gates / dt1)
        print(line, flush=True)
    for e in clones:
        e.close()
    eng.close()


if __name__ == "__main__":
    main()
