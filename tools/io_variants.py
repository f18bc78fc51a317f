"""Which part of the pipelined host-pointer call costs device time: SGFHE_IO_EXP variants (engine.hip:
1 = results collected only at the end, 2 = all results after the last kernel, 4 = all inputs before the
first kernel) alternating with device-resident calls in one process, so that clock drift falls on all alike.
usage (GPU box): python tools/io_variants.py [config] [batch] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
import sgfhe_jl_amd as S

name = sys.argv[1] if len(sys.argv) > 1 else "params1024"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
p = bench.make_params(S, name)
eng = S.Engine(p)
eng.generate_key(np.random.default_rng(11).integers(0, 2, size=p.n, dtype=np.uint64), 1)
g = torch.Generator(device="cuda")
g.manual_seed(1)
a1 = torch.randint(0, p.r, (B, p.n), dtype=torch.int64, device="cuda", generator=g)
a2 = torch.randint(0, p.r, (B, p.n), dtype=torch.int64, device="cuda", generator=g)
b1 = torch.randint(0, p.r, (B,), dtype=torch.int64, device="cuda", generator=g)
b2 = torch.randint(0, p.r, (B,), dtype=torch.int64, device="cuda", generator=g)
out = torch.zeros((B, 3, p.n + 1), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
h = [t.cpu().numpy().view(np.uint64) for t in (a1, b1, a2, b2)]
hout = eng.bootstrap_batch(h[0], h[1], h[2], h[3])
variants = [v for v in os.environ.get("VARIANTS", "dev,0,1,2,4,7").split(",")]
times = {v: [] for v in variants}
for rep in range(reps + 1):
    for v in variants:
        t0 = time.perf_counter()
        if v == "dev":
            eng.bootstrap_batch_device(a1.data_ptr(), b1.data_ptr(), a2.data_ptr(), b2.data_ptr(), B, out.data_ptr())
            eng.sync()
        else:
            os.environ["SGFHE_IO_EXP"] = v
            hout = eng.bootstrap_batch(h[0], h[1], h[2], h[3], out=hout)
        if rep:
            times[v].append((time.perf_counter() - t0) * 1e3)
base = np.mean(times[variants[0]])
for v in variants:
    print("%s batch %d, %-4s: %.2f ms per call (mean of %d, min %.2f)  / %s = %.4f"
          % (name, B, v, np.mean(times[v]), reps, np.min(times[v]), variants[0], np.mean(times[v]) / base), flush=True)
eng.close()
