"""Timing of pack_encrypted_bits at Params(1024) with a synthetic key."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sgfhe_jl_amd as S, bench
p = S.Params(int(sys.argv[1]) if len(sys.argv) > 1 else 1024)
count = int(sys.argv[2]) if len(sys.argv) > 2 else 4
eng = S.Engine(p)
eng.upload_key(bench.random_key(p, 1))
rng = np.random.default_rng(0)
a = rng.integers(0, p.r, size=(count, p.n, p.n), dtype=np.uint64)
b = rng.integers(0, p.r, size=(count, p.n), dtype=np.uint64)
eng.pack_encrypted_bits(a[:1], b[:1])
t0 = time.perf_counter(); eng.pack_encrypted_bits(a, b); dt = time.perf_counter() - t0
t1 = time.perf_counter(); eng.bootstrap_batch(a.reshape(-1, p.n), b.reshape(-1), a.reshape(-1, p.n), b.reshape(-1), raw=True); db = time.perf_counter() - t1
print("Params(%d): %d ciphertexts (%d bootstraps) packed in %.3f s = %.2f ciphertexts/s; the %d raw bootstraps alone %.3f s (host buffers)" % (p.n, count, count * p.n, dt, count / dt, count * p.n, db))
