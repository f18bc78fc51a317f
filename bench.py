#!/usr/bin/env python3
"""bench.py -- bootstraps/sec of the MI355X gate-bootstrap engine (BASELINE.json metric).

One "step" = one pass of the hot path (`bootstrap()` of nucypher/SGFHE.jl,
/root/reference/src/fhe.jl:608-621) over one batch of independent gate bootstraps resident in
HBM.  Default workload: the reference's own Params(1024) (Q = 92180593745615474572738561,
86.25-bit prime; SURVEY.md config 4'), batch 4096 per GPU, synthetic uniformly random bootstrap
LWE inputs and a bootstrap key generated on the device from a fixed seed (a valid key of a
random secret; sgfhe_bkey_generate).

Multi-GPU (launched by torch.distributed.run, one rank per GPU): the batch shards across ranks,
rank 0 builds the device-form key and broadcasts it once over RCCL; there is no collective in
the timed region ("scaling": "weak").

Prints ONE JSON line on rank 0.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0     # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
W_BYTES = {"params1024": 16, "params512": 16, "params2048": 16, "params64": 8, "synth64": 8}


def make_params(S, name):
    if name == "params1024":
        return S.Params(1024)
    if name == "params512":
        return S.Params(512)
    if name == "params2048":  # the largest parameter set the reference can build
        return S.Params(2048)
    if name == "params64":
        return S.Params(64)
    if name == "synth64":   # BASELINE.json config 3: n = 1024 with a single-limb 64-bit prime
        m = 8192
        Q = S.find_modulus(2 * m, (1 << 63) - (1 << 40))
        return S.Params.custom(1024, Q, 1 << 32)
    raise SystemExit("unknown --config " + name)


def random_key(p, seed):
    """Synthetic bootstrap key: canonical residues in [0, Q), [n][4][2][m][2] uint64."""
    rng = np.random.default_rng(seed)
    shape = (p.n, 4, 2, p.m)
    key = np.empty(shape + (2,), dtype=np.uint64)
    qhi = p.Q >> 64
    if qhi:
        key[..., 0] = rng.integers(0, 1 << 64, size=shape, dtype=np.uint64)
        key[..., 1] = rng.integers(0, qhi, size=shape, dtype=np.uint64)   # hi < Q_hi => value < Q
    else:
        key[..., 0] = rng.integers(0, p.Q, size=shape, dtype=np.uint64)
        key[..., 1] = 0
    return key


def algorithmic_bytes_per_bootstrap(p, W, batch):
    """n m W (4 + 8 / batch) + 40 (n + 1)."""
    return p.n * p.m * W * (4 + 8.0 / batch) + 40 * (p.n + 1)


def _counters(config, chunk):
    """Per-launch PMC averages of the newest committed profile (tools/profile_round.sh collects
    them in separate profiler runs, not inside this process).  None unless they match this
    workload (Params(1024), same chunk)."""
    if config != "params1024":
        return None
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_counters.json")), reverse=True):
        with open(path) as f:
            d = json.load(f)
        if d.get("chunk", 256) == chunk:
            return dict(d["kernels"], source=os.path.basename(path))
    return None


def measured_traffic(config, chunk):
    """HBM bytes per k_extprod launch (2 FETCH_SIZE + WRITE_SIZE, gfx950 correction) or None."""
    c = _counters(config, chunk)
    return c["k_extprod"].get("traffic_bytes_per_launch") if c else None


# Issue rates measured on the MI355X by tools/ubench_int.hip (profiles/r01_ubench_valu.txt), in
# 10^12 lane-operations per second, and k_extprod's static instruction mix (fractions of its VALU
# instructions: v_mad_u64_u32, v_mul_lo_u32, everything else).
VALU_RATE = {"mad64": 32.39, "mul": 34.38, "simple": 56.08}
VALU_MIX = {"mad64": 0.216, "mul": 0.111, "simple": 0.673}


def valu_roofline(config, chunk, ext_s):
    """The bound that actually limits k_extprod: integer VALU issue.  achieved = VALU
    instructions per launch (PMC SQ_INSTS_VALU) x 64 lanes / launch time; peak = the
    micro-benchmarked issue rate of the same instruction mix."""
    c = _counters(config, chunk)
    if not c or "SQ_INSTS_VALU" not in c.get("k_extprod", {}) or ext_s <= 0:
        return None
    insts = c["k_extprod"]["SQ_INSTS_VALU"]
    peak = 1.0 / sum(VALU_MIX[k] / VALU_RATE[k] for k in VALU_MIX)
    ach = insts * 64 / ext_s / 1e12
    return {"bound": "valu-int32", "achieved": ach, "peak": peak, "unit": "Tlane-op/s",
            "frac": ach / peak, "valu_insts_per_launch": insts, "source": c["source"]}


def cpu_baseline(p, sk, key_seed, seconds_target=15.0):
    """Oracle 'port' (oracle/sgfhe_oracle.c: reference-shaped, 128-bit Montgomery, 24 NTTs per
    iteration) timed on the host cores of this box: one independent bootstrap per thread (OpenMP
    over the batch, the same sharding the GPUs use), over a k-loop truncated to about
    `seconds_target` seconds and scaled to the full loop.  Test infrastructure used as a
    reported baseline only.  The key is the oracle's own generation from the same seed (the same
    key as on the device)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_c
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(16, cores))            # a one-GPU box's CPU share
    o = oracle_c.Oracle.from_params(p)
    key = o.bootstrap_key(sk, key_seed)
    rng = np.random.default_rng(7)
    a = rng.integers(0, p.r, size=(2, cores, p.n), dtype=np.uint64)
    b = rng.integers(0, p.r, size=(2, cores), dtype=np.uint64)
    t0 = time.perf_counter()
    o.bootstrap_batch(key, a[0], b[0], a[1], b[1], n_iters=2, threads=cores)
    per_iter = max((time.perf_counter() - t0) / 2, 1e-6)
    iters = int(min(p.n, max(4, seconds_target / per_iter)))
    t0 = time.perf_counter()
    o.bootstrap_batch(key, a[0], b[0], a[1], b[1], n_iters=iters, threads=cores)
    dt = time.perf_counter() - t0
    full = dt * p.n / iters
    return {"value": cores / full, "unit": "bootstraps/sec", "cores": cores, "kind": "port",
            "per_core": 1.0 / full,
            "sample": "%d bootstraps in parallel (one per thread), first %d of %d k-loop iterations "
                      "(%.1f s), scaled x%.2f; reference-shaped C restatement"
                      % (cores, iters, p.n, dt, p.n / iters)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="params1024")
    ap.add_argument("--batch", type=int, default=4096, help="bootstraps per GPU per step")
    ap.add_argument("--chunk", type=int, default=0, help="lock-step chunk (0 = engine default)")
    ap.add_argument("--lanes", type=int, default=1, help="1 = chunks in sequence, 2 = two streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import sgfhe_jl_amd as S

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:      # launched by torch.distributed.run: RCCL over xGMI
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    p = make_params(S, args.config)
    W = W_BYTES[args.config]
    eng = S.Engine(p, device=local_rank)
    if args.chunk:
        eng.set_chunk(args.chunk)
    eng.set_lanes(args.lanes)

    # ---- bootstrap key: rank 0 transforms, peers receive the device form over RCCL -------------
    kbytes = eng.key_device_form_bytes()
    sk = np.random.default_rng(11).integers(0, 2, size=p.n, dtype=np.uint64)
    KEY_SEED = 1
    if rank == 0:
        eng.generate_key(sk, KEY_SEED)
    if dist:
        blob = torch.empty(kbytes, dtype=torch.uint8, device="cuda")
        if rank == 0:
            eng.export_key_device_form(blob.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dist.broadcast(blob, src=0)
        torch.cuda.synchronize()
        bcast_s = time.perf_counter() - t0
        if rank != 0:
            eng.import_key_device_form(blob.data_ptr())
        del blob
    else:
        bcast_s = 0.0

    # ---- synthetic LWE inputs, resident in HBM ---------------------------------------------------
    g = torch.Generator(device="cuda")
    g.manual_seed(1234 + rank)
    B = args.batch
    a1 = torch.randint(0, p.r, (B, p.n), dtype=torch.int64, device="cuda", generator=g)
    a2 = torch.randint(0, p.r, (B, p.n), dtype=torch.int64, device="cuda", generator=g)
    b1 = torch.randint(0, p.r, (B,), dtype=torch.int64, device="cuda", generator=g)
    b2 = torch.randint(0, p.r, (B,), dtype=torch.int64, device="cuda", generator=g)
    out = torch.zeros((B, 3, p.n + 1), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()

    def step():
        eng.bootstrap_batch_device(a1.data_ptr(), b1.data_ptr(), a2.data_ptr(), b2.data_ptr(), B,
                                   out.data_ptr())

    for _ in range(args.warmup):
        step()
    eng.sync()
    eng.timing_enable(True)
    eng.timing_read(reset=True)

    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.sync()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    tm = eng.timing_read(reset=True)
    eng.timing_enable(False)

    if rank == 0:
        total = world * args.steps * B
        value = total / dt
        per_boot = algorithmic_bytes_per_bootstrap(p, W, B)
        chunk = tm["chunk"] or B
        # one k_extprod launch = `chunk` bootstraps x one k-loop iteration = chunk / n bootstraps
        launch_bytes = per_boot * chunk / p.n
        ext_s = tm["extprod_ms"] * 1e-3
        crt_s = tm["crt_ms"] * 1e-3
        achieved = launch_bytes / ext_s / 1e9 if ext_s > 0 else 0.0
        pair = launch_bytes / (ext_s + crt_s) / 1e9 if ext_s + crt_s > 0 else 0.0
        res = {
            "metric": "bootstraps/sec", "value": value, "unit": "bootstraps/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "%s gate bootstraps (AND/OR/XOR), batch %d per GPU, "
                                   "deterministic flatten" % (args.config, B),
                       "n": p.n, "m": p.m, "log2_Q": round(float(np.log2(float(p.Q))), 2),
                       "batch_per_gpu": B, "chunk": chunk, "lanes": args.lanes,
                       "rns_primes": len(eng.primes()),
                       "key": "generated on the device from a seed (valid key)",
                       "key_broadcast_s": round(bcast_s, 4)},
            "roofline": {"bound": "hbm", "kernel": "k_extprod", "achieved": achieved,
                         "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS,
                         "traffic": measured_traffic(args.config, chunk),
                         "launch_ms": tm["extprod_ms"], "launch_samples": tm["extprod_samples"],
                         "algorithmic_bytes_per_launch": launch_bytes,
                         "pair_kernel": "k_crt_acc", "pair_launch_ms": tm["crt_ms"],
                         "pair_achieved": pair, "pair_frac": pair / PEAK_HBM_GBS,
                         "whole_job_frac": per_boot * value / world / (PEAK_HBM_GBS * 1e9),
                         "valu": valu_roofline(args.config, chunk, ext_s)},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(p, sk, KEY_SEED)
        print(json.dumps(res))
    if dist:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
