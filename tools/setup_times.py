#!/usr/bin/env python3
"""What a host pays before its first bootstrap: ctx creation, key generation on the device, upload of a key
made on the host (sgfhe_bkey_upload: 2 l x 2 polynomials x n slices of canonical residues -> device form), a clone,
the first call.  Run on the GPU box: python tools/setup_times.py [n ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgfhe_jl_amd as S  # noqa: E402


def t(f):
    t0 = time.perf_counter()
    r = f()
    return r, (time.perf_counter() - t0) * 1e3


def main():
    for n in [int(x) for x in sys.argv[1:]] or [512, 1024, 2048]:
        params = S.Params(n)
        rng = np.random.default_rng(3)
        sk = rng.integers(0, 2, size=n, dtype=np.uint64)
        eng, ms_ctx = t(lambda: S.Engine(params))
        _, ms_gen = t(lambda: (eng.generate_key(sk, 4), eng.sync()))
        a = rng.integers(0, params.r, size=(1, n), dtype=np.uint64)
        b = rng.integers(0, params.r, size=1, dtype=np.uint64)
        _, ms_first = t(lambda: eng.bootstrap_batch(a, b, a, b))
        _, ms_second = t(lambda: eng.bootstrap_batch(a, b, a, b))
        cl, ms_clone = t(eng.clone)
        _, ms_clone_first = t(lambda: cl.bootstrap_batch(a, b, a, b))
        cl.close()
        # a host-made key: uniformly random canonical residues stand in (the cost does not depend on the values)
        Q = params.Q
        words = 2 * 2 * 2 * n * params.m          # (2 l rows) x 2 columns x n slices x m coefficients
        key = np.empty((words, 2), dtype=np.uint64)
        key[:, 0] = rng.integers(0, 1 << 63, size=words, dtype=np.uint64)
        key[:, 1] = rng.integers(0, int(Q >> 64), size=words, dtype=np.uint64)
        eng2, ms_ctx2 = t(lambda: S.Engine(params))
        _, ms_up = t(lambda: (eng2.upload_key(key), eng2.sync()))
        eng2.close()
        eng.close()
        print("Params(%d): ctx %.0f ms (a second one %.0f), device keygen %.0f ms, upload of a %.2f GB host key %.0f ms, "
              "first call %.1f ms, second %.1f ms, clone %.1f ms, its first call %.1f ms"
              % (n, ms_ctx, ms_ctx2, ms_gen, key.nbytes / 1e9, ms_up, ms_first, ms_second, ms_clone, ms_clone_first), flush=True)


if __name__ == "__main__":
    main()
