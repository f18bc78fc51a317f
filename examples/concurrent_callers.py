"""Independent callers on one bootstrap key (round 5): what N Julia tasks running
`bootstrap(hkey, nothing, bit1, bit2)` (src/fhe.jl:608-621 is a pure function) amount to on the GPU engine.

Every caller takes a CLONE of the engine (sgfhe_ctx_clone: the device key is shared, not copied) and calls the
drop-in entry point with one gate; the library gathers the calls that arrive together into one launch chain, so
eight callers get about six times the rate of one, each the bytes its call gives alone.

    python examples/concurrent_callers.py [n = 512] [callers = 8]
"""
import sys
import threading
import time

import numpy as np

import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgfhe_jl_amd as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
callers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(0)
params = S.Params(n)
key = S.PrivateKey(params, rng)
bkey = S.BootstrapKey(rng, key)                       # generated on the GPU
bits = [bool(b) for b in rng.integers(0, 2, size=2 * callers)]
enc = S.split_ciphertext(S.encrypt(key, rng, np.array(bits + [False] * (params.n - len(bits)))))

results = [None] * callers


def task(t, engine):
    a1, b1 = enc[2 * t].lwe.a[None], np.array([enc[2 * t].lwe.b])
    a2, b2 = enc[2 * t + 1].lwe.a[None], np.array([enc[2 * t + 1].lwe.b])
    for _ in range(20):                               # twenty gates, one call each
        results[t] = engine.bootstrap_batch(a1, b1, a2, b2)


clones = [bkey.engine.clone() for _ in range(callers)]
t0 = time.perf_counter()
threads = [threading.Thread(target=task, args=(t, clones[t])) for t in range(callers)]
for th in threads:
    th.start()
for th in threads:
    th.join()
dt = time.perf_counter() - t0
for t in range(callers):
    out = results[t][0]
    got = [S.decrypt(key, S.EncryptedBit(S.LWE(out[g, :params.n], out[g, params.n]))) for g in range(3)]
    y1, y2 = bits[2 * t], bits[2 * t + 1]
    assert got == [y1 and y2, y1 or y2, y1 != y2], (t, got)
print("%d callers x 20 single-gate calls at Params(%d): %.1f gates/s, every triple decrypts to AND / OR / XOR; %s"
      % (callers, n, callers * 20 / dt, clones[0].coalesce_stats()))
for c in clones:
    c.close()
