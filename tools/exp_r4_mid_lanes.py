"""Calls of 9-48 gates: one chunk on one stream against two halves on the two lanes (each half in the latency
form when it is at most 24 gates).  usage (GPU box): python tools/exp_r4_mid_lanes.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sgfhe_jl_amd as S

p = S.Params(1024)
eng = S.Engine(p, deterministic_only=True)
eng.generate_key(np.random.default_rng(11).integers(0, 2, size=p.n, dtype=np.uint64), 1)
g = torch.Generator(device="cuda"); g.manual_seed(1)
for B in (10, 12, 16, 20, 24, 32, 40, 48):
    a1 = torch.randint(0, p.r, (B, p.n), dtype=torch.int64, device="cuda", generator=g)
    a2 = torch.randint(0, p.r, (B, p.n), dtype=torch.int64, device="cuda", generator=g)
    b1 = torch.randint(0, p.r, (B,), dtype=torch.int64, device="cuda", generator=g)
    b2 = torch.randint(0, p.r, (B,), dtype=torch.int64, device="cuda", generator=g)
    res = {}
    outs = {}
    for name, chunk in (("one chunk", 0), ("two halves", ((B + 1) // 2 + 7) // 8 * 8)):
        eng.set_chunk(chunk)
        out = torch.zeros((B, 3, p.n + 1), dtype=torch.int64, device="cuda")
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.bootstrap_batch_device(a1.data_ptr(), b1.data_ptr(), a2.data_ptr(), b2.data_ptr(), B, out.data_ptr())
            eng.sync(); best = min(best, time.perf_counter() - t0)
        res[name] = best * 1e3
        outs[name] = out
    assert torch.equal(outs["one chunk"], outs["two halves"])
    print("batch %3d: one chunk %.2f ms, two halves on two lanes %.2f ms (chunks of %d)" %
          (B, res["one chunk"], res["two halves"], ((B + 1) // 2 + 7) // 8 * 8), flush=True)
eng.close()
