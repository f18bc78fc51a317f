import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import sgfhe_jl_amd as S
p = S.Params(1024)
eng = S.Engine(p)
eng.generate_key(np.random.default_rng(11).integers(0, 2, size=p.n, dtype=np.uint64), 1)
rng = np.random.default_rng(3)
for B in [int(x) for x in sys.argv[1:]]:
    a1 = rng.integers(0, p.r, size=(B, p.n), dtype=np.uint64); a2 = rng.integers(0, p.r, size=(B, p.n), dtype=np.uint64)
    b1 = rng.integers(0, p.r, size=B, dtype=np.uint64); b2 = rng.integers(0, p.r, size=B, dtype=np.uint64)
    for rep in range(3):
        t0 = time.perf_counter(); o = eng.bootstrap_batch(a1, b1, a2, b2); dt = time.perf_counter() - t0
        print("B %d rep %d %.2f ms" % (B, rep, dt * 1e3), flush=True)
