#!/usr/bin/env python3
"""What serialises concurrent single-gate callers?  (profiles/r05_concurrent.txt)
  1. how long the HOST takes to queue one call's launches against how long the device takes to run them
     (sgfhe_bootstrap_batch_device returns when everything is queued; sgfhe_sync when it has run);
  2. the same with K threads queueing on K clones at once: per-thread queueing time and wall time."""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import sgfhe_jl_amd as S  # noqa: E402

n = int(os.environ.get("SGFHE_LATENCY_N", "1024"))
gates = int(sys.argv[1]) if len(sys.argv) > 1 else 1
params = S.Params(n)
rng = np.random.default_rng(1)
eng = S.Engine(params)
eng.generate_key(rng.integers(0, 2, size=params.n, dtype=np.uint64), 2)
print("Params(%d), %d gate(s) per call, GPU_MAX_HW_QUEUES=%s" % (n, gates, os.environ.get("GPU_MAX_HW_QUEUES", "(default 4)")))


def dev_inputs():
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    a1 = torch.randint(0, params.r, (gates, params.n), dtype=torch.int64, device="cuda", generator=g)
    a2 = torch.randint(0, params.r, (gates, params.n), dtype=torch.int64, device="cuda", generator=g)
    b1 = torch.randint(0, params.r, (gates,), dtype=torch.int64, device="cuda", generator=g)
    b2 = torch.randint(0, params.r, (gates,), dtype=torch.int64, device="cuda", generator=g)
    out = torch.zeros((gates, 3, params.n + 1), dtype=torch.int64, device="cuda")
    return a1, b1, a2, b2, out


def one(e, bufs, reps=5):
    a1, b1, a2, b2, out = bufs
    q, w = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        e.bootstrap_batch_device(a1.data_ptr(), b1.data_ptr(), a2.data_ptr(), b2.data_ptr(), gates, out.data_ptr())
        t1 = time.perf_counter()
        e.sync()
        t2 = time.perf_counter()
        q.append((t1 - t0) * 1e3)
        w.append((t2 - t0) * 1e3)
    return min(q), min(w)


bufs = dev_inputs()
one(eng, bufs, 2)
q, w = one(eng, bufs)
print("one caller: launches queued after %.2f ms, device done after %.2f ms" % (q, w))
for k in (2, 4, 8):
    clones = [eng.clone() for _ in range(k)]
    allb = [dev_inputs() for _ in range(k)]
    for e, b in zip(clones, allb):
        one(e, b, 1)
    res = [None] * k
    gate = threading.Barrier(k)

    def body(t):
        gate.wait()
        res[t] = one(clones[t], allb[t], 4)
    ts = [threading.Thread(target=body, args=(t,)) for t in range(k)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    wall = (time.perf_counter() - t0) * 1e3
    print("%d callers on %d clones: per call queued after %s ms, done after %s ms; 4 calls each in %.1f ms -> %.1f gates/s"
          % (k, k, "/".join("%.1f" % r[0] for r in res), "/".join("%.1f" % r[1] for r in res), wall, k * 4 * gates / wall * 1e3))
    for e in clones:
        e.close()
eng.close()
