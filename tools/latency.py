"""Latency of small batches at Params(1024) (SGFHE_LATENCY_N=512: another Params(n)): one call of
bootstrap_batch_device per size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sgfhe_jl_amd as S

p = S.Params(int(os.environ.get("SGFHE_LATENCY_N", "1024")))
eng = S.Engine(p)
if os.environ.get("SGFHE_SMALL_MAX"):
    eng.set_small_batch_max(int(os.environ["SGFHE_SMALL_MAX"]))
eng.generate_key(np.random.default_rng(11).integers(0, 2, size=p.n, dtype=np.uint64), 1)
if os.environ.get("SGFHE_LATENCY_RANDOM"):      # the randomised flatten (bootstrap(bkey, rng, ...))
    eng.set_random_flatten(True, 1)
g = torch.Generator(device="cuda"); g.manual_seed(1)
for B in [int(x) for x in sys.argv[1:]] or [1, 8, 64, 256, 512]:
    a1 = torch.randint(0, p.r, (B, p.n), dtype=torch.int64, device="cuda", generator=g)
    a2 = torch.randint(0, p.r, (B, p.n), dtype=torch.int64, device="cuda", generator=g)
    b1 = torch.randint(0, p.r, (B,), dtype=torch.int64, device="cuda", generator=g)
    b2 = torch.randint(0, p.r, (B,), dtype=torch.int64, device="cuda", generator=g)
    out = torch.zeros((B, 3, p.n + 1), dtype=torch.int64, device="cuda")
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.bootstrap_batch_device(a1.data_ptr(), b1.data_ptr(), a2.data_ptr(), b2.data_ptr(), B, out.data_ptr())
        eng.sync(); dt = time.perf_counter() - t0
    # the drop-in signature (src/fhe.jl:608-610): host pointers in and out (sgfhe_bootstrap_batch)
    ha1, ha2 = a1.cpu().numpy().view(np.uint64), a2.cpu().numpy().view(np.uint64)
    hb1, hb2 = b1.cpu().numpy().view(np.uint64), b2.cpu().numpy().view(np.uint64)
    # a caller in a loop keeps its result array (SGFHE_LATENCY_FRESH_OUT=1: a fresh one per call, whose
    # release -- an munmap of pages the GPU driver has seen -- can stall the next call's kernels by
    # 25-40 ms: profiles/r03_exp_host_pinned.txt)
    hout = None
    for rep in range(3):
        t0 = time.perf_counter()
        hout = eng.bootstrap_batch(ha1, hb1, ha2, hb2, out=None if os.environ.get("SGFHE_LATENCY_FRESH_OUT") else hout)
        hdt = time.perf_counter() - t0
    assert os.environ.get("SGFHE_LATENCY_RANDOM") or np.array_equal(hout.view(np.int64), out.cpu().numpy())   # (every randomised call draws anew)
    print("batch %4d: %8.2f ms per call (device buffers), %8.2f ms (host buffers), %8.1f bootstraps/s"
          % (B, dt * 1e3, hdt * 1e3, B / dt), flush=True)
