"""The reference's own benchmark cases (/root/reference/test/performance.test.jl:28-143, BenchmarkTools
minimum times of single calls at Params(64), plus flatten at Params(1024)) on this build, with the
CPU restatement (oracle/, one thread) timed beside each on the same host:
  flatten()            :28-78    one value -> 2 digits; here per value over a polynomial pair
  external_product()   :81-111   a, b against G = gadget (identity), Params(64)
  bootstrap()          :114-139  one call, deterministic and randomised flatten, Params(64)
and the same single call at Params(512) / Params(1024).  Minimum of `reps` calls, like @benchmark.
usage (GPU box): python tools/reference_benchmarks.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np

import sgfhe_jl_amd as S
import oracle_c


def best(fn, reps):
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return min(t)


def words(vals):
    out = np.zeros((len(vals), 2), dtype=np.uint64)
    out[:, 0] = [v & 0xFFFFFFFFFFFFFFFF for v in vals]
    out[:, 1] = [v >> 64 for v in vals]
    return out


def main():
    import random
    rng = np.random.default_rng(7)
    pyrng = random.Random(7)
    for n in (64, 1024):                                   # flatten(): :28-78
        p = S.Params(n)
        eng = S.Engine(p)
        vals = words([pyrng.randrange(p.Q) for _ in range(2 * p.m)])
        v = vals.reshape(2, p.m, 2)
        eng.debug_flatten(v)
        t = best(lambda: eng.debug_flatten(v), 20)
        print("flatten, Params(%d), deterministic: %.2f us per call of 2 x %d values (%.1f ns per value, "
              "host buffers in and out)" % (n, t * 1e6, p.m, t * 1e9 / (2 * p.m)), flush=True)
        eng.close()

    p = S.Params(64)                                        # external_product(): :81-111
    o = oracle_c.Oracle.from_params(p)
    eng = S.Engine(p)
    a = words([int(x) for x in rng.integers(0, 2 ** 62, size=p.m, dtype=np.uint64)])
    b = words([int(x) for x in rng.integers(0, 2 ** 62, size=p.m, dtype=np.uint64)])
    a[:, 0] %= np.uint64(p.Q)
    b[:, 0] %= np.uint64(p.Q)
    G = np.zeros((4, 2, p.m, 2), dtype=np.uint64)
    for row, (g0, g1) in enumerate(((1, 0), (p.B, 0), (0, 1), (0, p.B))):
        G[row, 0, 0, 0], G[row, 1, 0, 0] = g0, g1
    ra, rb = eng.external_product(a, b, G)
    assert np.array_equal(ra, a) and np.array_equal(rb, b)   # :103-105
    t = best(lambda: eng.external_product(a, b, G), 20)
    tc = best(lambda: o.external_product(a, b, G), 5)
    print("external_product, Params(64) (ring of m = %d): %.1f us per call on the GPU (key slice transformed "
          "in the call), %.1f us CPU restatement" % (p.m, t * 1e6, tc * 1e6), flush=True)
    eng.close()

    for n in (64, 512, 1024):                               # bootstrap(): :114-139
        p = S.Params(n)
        o = oracle_c.Oracle.from_params(p)
        sk = o.private_key(1)
        eng = S.Engine(p, random_flatten=True)
        eng.generate_key(sk, 2)
        bits = np.array([1, 0], dtype=np.uint8)
        la, lb = o.lwe_encrypt_bits(sk, bits, 3)
        args = (la[0:1], lb[0:1], la[1:2], lb[1:2])
        for mode in ("deterministic", "random"):
            eng.set_random_flatten(mode == "random", 99)
            out = eng.bootstrap_batch(*args)
            for g, want in enumerate((0, 1, 1)):
                assert int(o.lwe_decrypt_bits(sk, out[:, g, :p.n], out[:, g, p.n])[0]) == want
            t = best(lambda: eng.bootstrap_batch(*args), 10 if n < 1024 else 5)
            line = "bootstrap, Params(%d), %s: %.2f ms per call (host buffers in and out)" % (n, mode, t * 1e3)
            if mode == "deterministic" and n <= 512:
                bk = o.bootstrap_key(sk, 2)
                tc = best(lambda: o.bootstrap_batch(bk, *args), 1)
                line += ", %.1f ms CPU restatement (one thread)" % (tc * 1e3)
            print(line, flush=True)
        eng.close()


if __name__ == "__main__":
    main()
