"""Phases of the pipelined host-pointer call (SGFHE_DEBUG_IO=1) beside the device-resident step:
usage (GPU box): SGFHE_DEBUG_IO=1 python tools/io_phases.py [config] [batch]"""
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
p = bench.make_params(S, name)
eng = S.Engine(p)
sk = np.random.default_rng(11).integers(0, 2, size=p.n, dtype=np.uint64)
eng.generate_key(sk, 1)
g = torch.Generator(device="cuda")
g.manual_seed(1)
a1 = torch.randint(0, p.r, (B, p.n), dtype=torch.int64, device="cuda", generator=g)
a2 = torch.randint(0, p.r, (B, p.n), dtype=torch.int64, device="cuda", generator=g)
b1 = torch.randint(0, p.r, (B,), dtype=torch.int64, device="cuda", generator=g)
b2 = torch.randint(0, p.r, (B,), dtype=torch.int64, device="cuda", generator=g)
out = torch.zeros((B, 3, p.n + 1), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
h = [t.cpu().numpy().view(np.uint64) for t in (a1, b1, a2, b2)]
hout = eng.bootstrap_batch(h[0], h[1], h[2], h[3])           # allocates the staging buffers
# alternating, so that the drift of the clock with temperature (about 1 % over the first minutes of a
# run on this pool) falls on both legs alike
dev, host = [], []
for rep in range(int(os.environ.get("REPS", "5"))):
    t0 = time.perf_counter()
    eng.bootstrap_batch_device(a1.data_ptr(), b1.data_ptr(), a2.data_ptr(), b2.data_ptr(), B, out.data_ptr())
    eng.sync()
    dev.append((time.perf_counter() - t0) * 1e3)
    print("device-resident call: %.2f ms" % dev[-1], flush=True)
    t0 = time.perf_counter()
    hout = eng.bootstrap_batch(h[0], h[1], h[2], h[3], out=hout)
    host.append((time.perf_counter() - t0) * 1e3)
    print("host-pointer call:    %.2f ms" % host[-1], flush=True)
print("%s batch %d: device-resident %.2f ms, host pointers %.2f ms per call (means of %d alternating calls): "
      "host / device = %.4f" % (name, B, np.mean(dev), np.mean(host), len(dev), np.mean(host) / np.mean(dev)))
eng.close()
