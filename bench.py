#!/usr/bin/env python3
"""bench.py -- bootstraps/sec of the MI355X gate-bootstrap engine (BASELINE.json metric).

One "step" = one pass of the hot path (`bootstrap()` of nucypher/SGFHE.jl,
/root/reference/src/fhe.jl:608-621) over one batch of independent gate bootstraps resident in
HBM.  Default workload: the reference's own Params(1024) (Q = 92180593745615474572738561,
86.25-bit prime; SURVEY.md config 4'), batch 4096 per GPU (8192 per GPU on 8 GPUs = config 5's
65536), synthetic uniformly random bootstrap LWE inputs and a bootstrap key generated on the
device from a fixed seed (a valid key of a random secret; sgfhe_bkey_generate).

Other workloads (parity-test cases, `--config`): params512 (config 2, use --batch 1024), synth64
(config 3: n = 1024 over a 64-bit prime), rns2 (config 4: n = 1024 over the composite
Q = B * Bp of two 43-bit NTT primes, the RNS2Number ring of src/rns.jl / src/fhe2.jl:57-60),
params64, params2048.

Multi-GPU: `python bench.py --gpus N` starts N ranks itself (torch.distributed.run, one per GPU,
rendezvous on 127.0.0.1) when it was not already started by a launcher; the batch shards across
ranks, rank 0 builds the device-form key and broadcasts it once over RCCL; there is no
collective in the timed region ("scaling": "weak").

Prints ONE JSON line on rank 0.
"""

import argparse
import csv
import glob
import hashlib
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0     # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
NOMINAL_SCLK_HZ = 2.4e9   # "Max clock 2400 MHz", same guide (the clock under this workload is 2.1-2.3 GHz)
W_BYTES = {"params1024": 16, "params512": 16, "params2048": 16, "params64": 8, "synth64": 8,
           "rns2": 16}


def rns2_moduli(S, n=1024):
    """BASELINE.json config 4: two NTT-friendly primes Bp < B, both 1 mod r, from the rule of
    src/fhe2.jl:57-58 (find_modulus(r, bound), then the next one) with bound = ceil(sqrt(1220
    r^4 n^2)) so that Q = B Bp keeps the noise margin of Params(n)."""
    import math
    r = 16 * n
    bound = math.isqrt(1220 * r ** 4 * n ** 2) + 1
    Bp = S.find_modulus(r, bound)
    B = S.find_modulus(r, Bp + 1)
    return B, Bp


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
    if name == "rns2":      # BASELINE.json config 4: composite Q = B * Bp, gadget base B
        B, Bp = rns2_moduli(S)
        return S.Params.custom(1024, B * Bp, B)
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


def _counters(config, chunk, build_id):
    """Per-launch PMC averages and rocprofv3 kernel-trace durations of the newest committed
    profile (tools/profile_round.sh collects them in separate profiler runs, not inside this
    process).  They are quoted only when that profile was taken on a library with the same
    sgfhe_build_id() as the one loaded here (hash of csrc/ plus any ablation flags).
    Returns (counters, reason)."""
    if config != "params1024":
        return None, "counters are collected for the params1024 workload only"
    stale = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_counters.json")), reverse=True):
        with open(path) as f:
            d = json.load(f)
        if d.get("chunk", 256) != chunk:
            continue
        if d.get("build_id", d.get("source_hash")) != build_id:
            stale = stale or os.path.basename(path)
            continue
        return dict(d["kernels"], source=os.path.basename(path), valu_mix=d.get("valu_mix"),
                    rocprof_avg_us=d.get("rocprof_avg_us"), rocprof_iter_us=d.get("rocprof_iter_us"),
                    rocprof_alone_avg_us=d.get("rocprof_alone_avg_us")), None
    if stale:
        return None, "the loaded library (build id %s) is not the one %s was collected on" % (build_id, stale)
    return None, "no committed counters for chunk %d" % chunk


def _elf_interpreter():
    """The interpreter binary to put after `rocprofv3 ... --`: the resolved sys.executable, and only if it
    is an ELF file.  The profiler's preloaded library initialises the GPU before the program starts, so a
    shim in between (pyenv / conda wrapper script, `#!/usr/bin/env`) would be an exec from a process that
    has initialised the GPU -- which this pool forbids.  Returns (path, None) or (None, reason)."""
    exe = os.path.realpath(sys.executable or "")
    try:
        with open(exe, "rb") as f:
            magic = f.read(4)
    except OSError as e:
        return None, "cannot read the interpreter %r: %s" % (exe, e)
    if magic != b"\x7fELF":
        return None, "the interpreter %r is not an ELF binary (a wrapper script must not sit behind rocprofv3)" % exe
    return exe, None


def live_counters(args, chunk, kernels, rnd):
    """Never raises: whatever goes wrong in the child passes or in reading their output, the headline line
    is still printed, with the committed counters of the same build and the reason."""
    try:
        return _live_counters(args, chunk, kernels, rnd)
    except Exception as e:                              # noqa: BLE001 -- a measurement aid must not cost the line
        return None, "live counter passes failed: %r" % (e,)


def _live_counters(args, chunk, kernels, rnd):
    """HBM traffic and VALU instruction counts of the two k-loop kernels measured NOW, on this box:
    rocprofv3 --pmc passes run as child processes of this bench (one counter group per pass,
    --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes), each over one chunk of
    the same workload on one lane -- the launches are the ones of the timed region, the counters are
    per launch.  FETCH_SIZE / WRITE_SIZE come in KB, and on gfx950 FETCH_SIZE counts half the bytes of
    wide coalesced reads: traffic = (2 FETCH_SIZE + WRITE_SIZE) * 1024 B.  Returns (dict, None) or
    (None, reason); the caller falls back to the committed profile of the same build."""
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 is not on PATH"
    exe, why = _elf_interpreter()
    if exe is None:
        return None, why
    one = [exe, os.path.abspath(__file__), "--config", args.config, "--lanes", "1", "--chunk", str(chunk),
           "--batch", str(chunk), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-host-io",
           "--no-live-counters", "--flatten", "random" if rnd else "deterministic"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["TMPDIR"] = "/tmp"
    out = {k: {} for k in kernels}
    t0 = time.perf_counter()
    for group in (["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_INSTS_VALU", "SQ_WAVES"]):
        d = tempfile.mkdtemp(prefix="sgfhe_pmc_", dir="/tmp")
        try:
            r = subprocess.run(["rocprofv3", "--pmc", *group, "--kernel-trace", "--output-format", "csv",
                                "-d", d, "-o", "run", "--", *one], env=env, cwd="/tmp", timeout=300,
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
            if r.returncode != 0:
                return None, "rocprofv3 --pmc %s failed: %s" % (" ".join(group), r.stderr[-300:])
            vals = {}   # kernel -> grid -> counter -> [values]
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f, newline="") as fh:
                    for row in csv.DictReader(fh):
                        for k in kernels:
                            if "::" + k + "<" in row["Kernel_Name"]:
                                vals.setdefault(k, {}).setdefault(row["Grid_Size"], {}).setdefault(
                                    row["Counter_Name"], []).append(float(row["Counter_Value"]))
            for k in kernels:
                if k not in vals:
                    return None, "no %s launches in the --pmc %s pass" % (k, " ".join(group))
                # the k-loop's launches: the most frequent grid of that kernel
                g = max(vals[k].values(), key=lambda cs: max(len(v) for v in cs.values()))
                for cname, v in g.items():
                    out[k][cname] = sum(v) / len(v)
                    out[k]["launches"] = len(v)
        except subprocess.TimeoutExpired:
            return None, "rocprofv3 --pmc %s timed out" % " ".join(group)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    for k in kernels:
        if "FETCH_SIZE" not in out[k] or "WRITE_SIZE" not in out[k]:
            return None, "the --pmc passes returned no FETCH_SIZE / WRITE_SIZE for %s" % k
        out[k]["traffic_bytes_per_launch"] = (2 * out[k]["FETCH_SIZE"] + out[k]["WRITE_SIZE"]) * 1024
    out["source"] = ("measured in this run: rocprofv3 --pmc child passes (FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU "
                     "SQ_WAVES, --kernel-trace only) over one %d-bootstrap chunk on one lane, %.0f s"
                     % (chunk, time.perf_counter() - t0))
    return out, None


# Issue rates measured on the MI355X by tools/ubench_int.hip (profiles/r01_ubench_valu.txt,
# r03_ubench_valu.txt), in 10^12 lane-operations per second: 64-bit multiply-adds, 32-bit multiplies,
# the full-rate simple instructions (add, sub, two-operand logic, right shifts, moves) and the
# other simple ones (left shifts, min / max, three-operand and carry forms, v_alignbit: 36-38).
VALU_RATE = {"mad64": 32.39, "mul": 34.38, "fast": 56.08, "slow": 36.0,
             "simple": 56.08}   # "simple": profiles before r03_v10 did not split fast / slow


RAW_LANES_PER_CU = 4 * 32   # four SIMDs issuing 32 lanes per clock (a wave64 instruction in two passes)


def valu_roofline(c, ext_s, cus=None, sclk_hz=None):
    """The bound that actually limits k_extprod: integer VALU issue.  achieved = VALU
    instructions per launch (PMC SQ_INSTS_VALU) x 64 lanes / launch time; peak = the
    micro-benchmarked issue rate of the kernel's own static instruction mix (fractions of
    64-bit multiply-adds, 32-bit multiplies and everything else, tools/valu_mix.py)."""
    if not c or "SQ_INSTS_VALU" not in c.get("k_extprod", {}) or not c.get("valu_mix") or ext_s <= 0:
        return None
    mix = c["valu_mix"]
    insts = c["k_extprod"]["SQ_INSTS_VALU"]
    peak = 1.0 / sum(mix[k] / VALU_RATE[k] for k in VALU_RATE if k in mix)
    ach = insts * 64 / ext_s / 1e12
    res = {"bound": "valu-int32", "achieved": ach, "peak": peak, "unit": "Tlane-op/s",
           "frac": ach / peak,
           "peak_is": "issue rate of this kernel's own static instruction mix, priced with tools/ubench_int.hip "
                      "(64-bit multiply-adds and 32-bit multiplies issue at about half the add rate): NOT the "
                      "device's raw lane rate -- that is raw_peak / raw_frac",
           "valu_insts_per_launch": insts, "mix": mix, "source": c["source"]}
    if cus and sclk_hz:   # the unpriced figure beside it: every lane, every clock, nominal clock
        raw = cus * RAW_LANES_PER_CU * sclk_hz / 1e12
        res.update({"raw_peak": raw, "raw_frac": ach / raw,
                    "raw_peak_is": "%d CUs x 4 SIMDs x 32 lanes x %.2f GHz nominal" % (cus, sclk_hz / 1e9)})
    return res


def _cpu_budget():
    """(cores this process may run on, cgroup CPU quota in cores or None)."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
    return avail, quota


def cpu_baseline(p, sk, key_seed, cap, seconds_target=8.0, seconds_full=30.0):
    """The CPU path timed on the host cores of this box, in the same run.
    `port` = oracle/sgfhe_oracle.c, reference-shaped (128-bit Montgomery, 24 NTTs per iteration --
    what the Julia reference executes), one independent bootstrap per thread (OpenMP over the
    batch, the sharding the GPUs use), over the WHOLE k-loop when that is estimated to take at most
    `seconds_full` seconds (Params(1024) on this pool's boxes: about 16 s -- "1024 of 1024 iterations", no
    scaling), else over a k-loop truncated to about `seconds_target` seconds and scaled to the full loop; `opt` = the same arithmetic in the GPU path's algebra (key in the
    NTT domain, 4 + 2 NTTs per iteration; bit-identical).  Timed on every core this process may
    use -- the affinity mask, or the cgroup CPU quota where that is smaller (the GPU boxes of the
    pool show 256 cores and a quota of 16: 32 threads measured no faster than 16) -- as the
    headline `value`, and on one GPU's share of the affinity mask (an eighth: `share`) when that
    is a different number.  Test infrastructure used as a reported baseline only.  The key is the oracle's own
    generation from the same seed (the same key as on the device)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_c
    avail, quota = _cpu_budget()
    full = max(1, min(cap, avail) if cap else avail)
    if quota:   # a cgroup CPU quota below the affinity mask: more threads than that add nothing
        full = max(1, min(full, int(quota + 0.5)))
    share = min(full, max(1, avail // 8))
    o = oracle_c.Oracle.from_params(p)
    key = o.bootstrap_key(sk, key_seed)
    khat = o.key_transform(key, threads=full) if o.uses_ntt else None
    rng = np.random.default_rng(7)

    def timed(k, opt, cores):
        a = rng.integers(0, p.r, size=(2, cores, p.n), dtype=np.uint64)
        b = rng.integers(0, p.r, size=(2, cores), dtype=np.uint64)
        o.bootstrap_batch(k, a[0], b[0], a[1], b[1], n_iters=1, threads=cores, opt=opt)   # thread start-up
        t0 = time.perf_counter()
        o.bootstrap_batch(k, a[0], b[0], a[1], b[1], n_iters=4, threads=cores, opt=opt)
        per_iter = max((time.perf_counter() - t0) / 4, 1e-6)
        iters = p.n if per_iter * p.n <= seconds_full else int(min(p.n, max(4, seconds_target / per_iter)))
        t0 = time.perf_counter()
        o.bootstrap_batch(k, a[0], b[0], a[1], b[1], n_iters=iters, threads=cores, opt=opt)
        dt = time.perf_counter() - t0
        return dt * p.n / iters, iters, dt

    def leg(cores):
        t, iters, dt = timed(key, False, cores)
        r = {"value": cores / t, "unit": "bootstraps/sec", "cores": cores, "kind": "port",
             "per_core": 1.0 / t,
             "sample": "%d bootstraps in parallel (one per thread), %s%d of %d k-loop iterations "
                       "(%.1f s)%s; reference-shaped C restatement"
                       % (cores, "" if iters == p.n else "first ", iters, p.n, dt,
                          "" if iters == p.n else ", scaled x%.2f" % (p.n / iters)),
             "iterations_timed": iters, "iterations_total": p.n}
        if khat is not None:
            ot, oiters, odt = timed(khat, True, cores)
            r["opt"] = {"value": cores / ot, "per_core": 1.0 / ot,
                        "sample": "same threads, GPU-path algebra (NTT-domain key, 6 NTTs per iteration "
                                  "instead of 24), %s%d of %d iterations (%.1f s)"
                                  % ("" if oiters == p.n else "first ", oiters, p.n, odt),
                        "iterations_timed": oiters}
        return r
    res = leg(full)
    res.update({"cores_available": avail, "cores_cap": cap or None,
                "cgroup_cpu_quota": quota})
    if share != full:
        res["share"] = dict(leg(share), note="one GPU's share of the host: cores_available / 8")
    return res


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """`bench.py --gpus N` without a launcher: start N ranks as children (one per GPU) and relay
    rank 0's JSON line.  This process never touches the GPU (a process that has initialised HIP
    must not start or become another GPU program)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    for l in proc.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1])
    sys.exit(proc.returncode if proc.returncode else (0 if lines else 1))


def dry_run(args):
    """The N-rank control flow of main() with gloo on the CPU and no engine: rendezvous, barrier,
    MAX over ranks, one line from rank 0.  Exercised by tests/test_distributed_cpu.py."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > 1 or "RANK" in os.environ:
        dist.init_process_group("gloo")
        dist.barrier()
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tmax = float(t.item())
        ones = torch.ones(1, dtype=torch.int64)          # the proof-of-ranks field of the real line
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        group_ranks = int(ones.item())
        backend = dist.get_backend()
        dist.barrier()
        dist.destroy_process_group()
        if group_ranks != args.gpus:
            raise SystemExit("the process group holds %d ranks, --gpus says %d: no line" % (group_ranks, args.gpus))
    else:
        tmax, group_ranks, backend = 1.0, None, None
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "max_over_ranks": tmax,
                          "steps": args.steps, "warmup": args.warmup,
                          "config": {"group_ranks": group_ranks, "collective_backend": backend, "rccl_ranks": None,
                                     "key_broadcast_gbs": None}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="params1024")
    ap.add_argument("--batch", type=int, default=0,
                    help="bootstraps per GPU per step (default 4096; 8192 on 8 GPUs = config 5)")
    ap.add_argument("--chunk", type=int, default=0, help="lock-step chunk (0 = engine default)")
    ap.add_argument("--lanes", type=int, default=0,
                    help="0 = engine default (2: pairs of chunks on two streams), 1 = chunks in sequence")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-isolated", action="store_true",
                    help="skip the extra untimed step that measures each kernel alone (lanes = 1)")
    ap.add_argument("--no-host-io", action="store_true",
                    help="skip the extra step timed through host pointers (sgfhe_bootstrap_batch)")
    ap.add_argument("--no-live-counters", action="store_true",
                    help="do not run the rocprofv3 --pmc child passes that measure roofline.traffic and the "
                         "VALU instruction count on this box (the committed profile of the same build is quoted "
                         "instead); also skipped with --no-host-io or --no-cpu-baseline (quick A/B runs)")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="cap on the cpu_baseline threads (0 = every core this process may use; "
                         "the leg on an eighth of them is reported beside it)")
    ap.add_argument("--flatten", choices=["deterministic", "random"], default="deterministic",
                    help="random: the rng::AbstractRNG branch (src/utils.jl:198-241; at Params(1024) "
                         "the ctx's six-prime basis); not the headline")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher rehearsal without a GPU: the ranks rendezvous over gloo, take the "
                         "MAX of a dummy timing and rank 0 prints a stub line (tests/)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args)          # does not return
    if args.dry_run:
        return dry_run(args)

    import torch
    import sgfhe_jl_amd as S

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # Rehearsal on a box with fewer GPUs than ranks (tests/test_gpu_multi.py): the ranks share the
    # devices that exist and talk over gloo (RCCL refuses two ranks on one device).  Never a
    # measurement: the line says so in config.rehearsal.
    rehearsal = os.environ.get("SGFHE_BENCH_SHARE_GPU") == "1"
    if rehearsal:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:      # launched by torch.distributed.run: RCCL over xGMI
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    p = make_params(S, args.config)
    W = W_BYTES[args.config]
    B = args.batch or (8192 if world == 8 else 4096)
    rnd = args.flatten == "random"
    # the headline (deterministic flatten) on a ctx with the one basis it needs: the key blob that the
    # ranks exchange is then the five-prime form (1.34 GB at Params(1024)); `--flatten random` takes a
    # default ctx (a basis per mode where the randomised one needs a prime more)
    eng = S.Engine(p, device=local_rank, deterministic_only=not rnd)
    if rnd:
        eng.set_random_flatten(True, 0x5EED + rank)
    if args.lanes:
        eng.set_lanes(args.lanes)
    if args.chunk:
        eng.set_chunk(args.chunk)

    # ---- bootstrap key: rank 0 generates, peers receive the device form over RCCL ---------------
    sk = np.random.default_rng(11).integers(0, 2, size=p.n, dtype=np.uint64)
    KEY_SEED = 1
    t0 = time.perf_counter()
    if rank == 0:
        eng.generate_key(sk, KEY_SEED)
    keygen_s = time.perf_counter() - t0
    bcast_s = S.distributed.broadcast_key(eng, src=0)[1] if dist else 0.0

    # ---- synthetic LWE inputs, resident in HBM ---------------------------------------------------
    g = torch.Generator(device="cuda")
    g.manual_seed(1234 + rank)
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
    lanes = 2 if (args.lanes or 2) == 2 and B > (tm["chunk"] or B) else 1
    iso = None
    if lanes == 2 and not args.no_isolated:
        # The kernels of the two lanes overlap, so their sampled durations are durations under
        # co-execution.  One extra, untimed step with the same chunks in sequence gives the
        # durations of each kernel alone on the device.
        eng.set_lanes(1)
        eng.set_chunk(tm["chunk"])
        step()
        iso = eng.timing_read(reset=True)
        eng.set_lanes(2)
    eng.timing_enable(False)

    # ---- the metric as SURVEY.md 8(d) words it: host buffers in, host buffers out (PCIe inside) ----
    # Same number of steps as the headline, after a warm-up of its own (the first call through host
    # pointers also allocates the engine's staging buffers and touches the pages of the result array),
    # bracketed by the same barriers, MAX over ranks.  sgfhe_bootstrap_batch moves the arrays chunk by
    # chunk beside the kernels.  The clock of this pool's boxes drifts by up to 1 % over the first
    # minutes of a run (power limit and temperature), so the ratio to the device-resident rate comes
    # from steps that ALTERNATE between the two entry points -- one synchronous device-resident step,
    # one host-pointer step, `steps` times -- not from this leg against the headline timed earlier.
    host_io = None
    if not args.no_host_io:
        ha1, ha2 = a1.cpu().numpy().view(np.uint64), a2.cpu().numpy().view(np.uint64)
        hb1, hb2 = b1.cpu().numpy().view(np.uint64), b2.cpu().numpy().view(np.uint64)
        t1 = time.perf_counter()
        hout = eng.bootstrap_batch(ha1, hb1, ha2, hb2)       # sgfhe_bootstrap_batch: H2D, k-loop, D2H
        hdt_first = time.perf_counter() - t1
        for _ in range(max(0, args.warmup - 1)):
            eng.bootstrap_batch(ha1, hb1, ha2, hb2, out=hout)
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        hdt = ddt = 0.0
        for _ in range(args.steps):
            t1 = time.perf_counter()
            step()
            eng.sync()
            t2 = time.perf_counter()
            eng.bootstrap_batch(ha1, hb1, ha2, hb2, out=hout)   # synchronous: returns with `hout` complete
            t3 = time.perf_counter()
            ddt += t2 - t1
            hdt += t3 - t2
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        same = None if rnd else bool(np.array_equal(hout.view(np.int64), out.cpu().numpy()))
        if dist:
            tmax = torch.tensor([hdt, ddt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            hdt, ddt = float(tmax[0].item()), float(tmax[1].item())
        host_io = {"value": world * args.steps * B / hdt, "unit": "bootstraps/sec",
                   "ms_per_step": hdt / args.steps * 1e3, "steps": args.steps,
                   "warmup": max(1, args.warmup), "first_call_ms": hdt_first * 1e3,
                   "device_resident_ms_per_step_alternating": ddt / args.steps * 1e3,
                   "vs_device_resident": ddt / hdt,
                   "equals_device_resident_output": same,
                   "note": "inputs and outputs in host memory (sgfhe_bootstrap_batch, the drop-in "
                           "signature): includes H2D of 2 (n + 1) and D2H of 3 (n + 1) words per "
                           "bootstrap, pipelined chunk by chunk beside the kernels; vs_device_resident "
                           "= time of the device-resident steps / time of the host-pointer steps of one "
                           "alternating sequence (one synchronous call each per step)"}

    # ---- N ranks: proof on the line that they talked ---------------------------
    # all-reduce(SUM) of 1 over the process group the ranks were given: RCCL over xGMI when launched by
    # torch.distributed.run on N devices (gloo in a rehearsal).  The line is refused unless it equals --gpus.
    group_ranks = None
    if dist:
        ones = torch.ones(1, dtype=torch.int64, device="cuda" if not rehearsal else "cpu")
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        group_ranks = int(ones.item())
        if group_ranks != args.gpus:
            raise SystemExit("the process group holds %d ranks, --gpus says %d: no line" % (group_ranks, args.gpus))
    props = torch.cuda.get_device_properties(local_rank)

    if rank == 0:
        total = world * args.steps * B
        value = total / dt
        per_boot = algorithmic_bytes_per_bootstrap(p, W, B)
        chunk = tm["chunk"] or B
        # The unit of the roofline: one k-loop iteration of one chunk = one k_extprod launch + one
        # k_crt_lean launch = `chunk` bootstraps x one iteration = chunk / n bootstraps' worth of
        # algorithmic bytes.  Its duration is the device wall time of the call (HIP events from the
        # first to the last kernel of a step, both lanes) divided by the chunk-iterations in it:
        # with two lanes the two kernels of different chunks overlap and per-kernel durations do
        # not add up.
        launch_bytes = per_boot * chunk / p.n
        iter_s = tm["call_ms"] * 1e-3 * chunk / (tm["call_batch"] * p.n) if tm["calls"] else 0.0
        achieved = launch_bytes / iter_s / 1e9 if iter_s > 0 else 0.0
        ext_s = tm["extprod_ms"] * 1e-3
        build_id = eng.build_id()
        ctr, why = _counters(args.config, chunk, build_id)
        live = live_why = None
        how = ("overlapped with the other lane's kernels" if lanes == 2 else "alone on the device")
        # the kernels the engine actually launches for this parameter set and flatten mode
        # (k_crt_lean, or k_crt_lean_rnd / k_crt_acc2 / k_crt_acc outside its bounds)
        ext_full, crt_full = eng.kernel_names()
        crt_name = crt_full.split("<")[0]
        kern = {"k_extprod": {"name": ext_full, "launch_ms": tm["extprod_ms"],
                              "launch_samples": tm["extprod_samples"], "launch_ms_is": how},
                crt_name: {"name": crt_full, "launch_ms": tm["crt_ms"], "launch_samples": tm["crt_samples"],
                           "launch_ms_is": how}}
        if crt_name != "k_crt_lean":
            ctr, why = None, "the committed counters are k_crt_lean's; this run launches " + crt_full
        alone = iso or tm
        for k, key in (("k_extprod", "extprod_ms"), (crt_name, "crt_ms")):
            kern[k]["launch_ms_alone"] = alone[key]
            kern[k]["kernel_achieved"] = launch_bytes / (alone[key] * 1e-3) / 1e9 if alone[key] > 0 else 0.0
            kern[k]["kernel_frac"] = kern[k]["kernel_achieved"] / PEAK_HBM_GBS
        # (part of the full line only: the quick modes of the A/B scripts, --no-host-io / --no-cpu-baseline,
        # skip it, and so does a run that is itself under a profiler, e.g. tools/profile_round.sh)
        if (world == 1 and not rehearsal and not args.no_live_counters and not args.no_host_io
                and not args.no_cpu_baseline):
            # (the engine of this process is idle meanwhile; its memory stays allocated)
            live, live_why = live_counters(args, chunk, ["k_extprod", crt_name], rnd)
        traffic = None
        rp_ms = None
        if live:   # this box, this run
            for k in ("k_extprod", crt_name):
                kern[k]["traffic"] = live[k]["traffic_bytes_per_launch"]
                kern[k]["valu_insts_per_launch"] = live[k].get("SQ_INSTS_VALU")
                kern[k]["waves_per_launch"] = live[k].get("SQ_WAVES")
            traffic = live["k_extprod"]["traffic_bytes_per_launch"] + live[crt_name]["traffic_bytes_per_launch"]
            if ctr:   # durations of the committed rocprofv3 trace, and the static instruction mix
                rp = ctr.get("rocprof_avg_us") or {}
                rpa = ctr.get("rocprof_alone_avg_us") or {}
                for k in ("k_extprod", "k_crt_lean"):
                    if k in rp and k in kern:
                        kern[k]["launch_ms_rocprof"] = rp[k] * 1e-3
                    if k in rpa and k in kern:
                        kern[k]["launch_ms_alone_rocprof"] = rpa[k] * 1e-3
                rp_ms = ctr.get("rocprof_iter_us") and ctr["rocprof_iter_us"] * 1e-3
                ctr = dict(ctr, k_extprod=dict(ctr.get("k_extprod", {}), SQ_INSTS_VALU=live["k_extprod"]["SQ_INSTS_VALU"]),
                           source=live["source"] + "; instruction mix from " + ctr["source"])
        elif ctr:
            t_ext = ctr.get("k_extprod", {}).get("traffic_bytes_per_launch")
            t_crt = ctr.get("k_crt_lean", {}).get("traffic_bytes_per_launch")
            kern["k_extprod"]["traffic"] = t_ext
            kern["k_crt_lean"]["traffic"] = t_crt
            if t_ext is not None and t_crt is not None:
                traffic = t_ext + t_crt
            rp = ctr.get("rocprof_avg_us") or {}
            rpa = ctr.get("rocprof_alone_avg_us") or {}
            for k in ("k_extprod", "k_crt_lean"):
                if k in rp:
                    kern[k]["launch_ms_rocprof"] = rp[k] * 1e-3
                if k in rpa:
                    kern[k]["launch_ms_alone_rocprof"] = rpa[k] * 1e-3
            rp_ms = ctr.get("rocprof_iter_us") and ctr["rocprof_iter_us"] * 1e-3
        res = {
            "metric": "bootstraps/sec", "value": value, "unit": "bootstraps/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i32", "data": "synthetic",
            "config": {"workload": "%s gate bootstraps (AND/OR/XOR), batch %d per GPU, "
                                   "%s flatten" % (args.config, B, args.flatten),
                       "n": p.n, "m": p.m, "log2_Q": round(float(np.log2(float(p.Q))), 2),
                       "batch_per_gpu": B, "chunk": chunk, "lanes": lanes,
                       "rns_primes": len(eng.primes()),
                       "key": "generated on the device from a seed (valid key)",
                       "keygen_s": round(keygen_s, 3), "key_bytes": eng.key_device_form_bytes(),
                       "key_broadcast_s": round(bcast_s, 4),
                       # one broadcast of the device-form key blob rank 0 -> peers (null without peers)
                       "key_broadcast_gbs": (round(eng.key_device_form_bytes() / bcast_s / 1e9, 2)
                                             if dist and world > 1 and bcast_s > 0 else None),
                       "collective_backend": (dist.get_backend() if dist else None),
                       "rccl_ranks": group_ranks if (dist and not rehearsal) else None,
                       "group_ranks": group_ranks,
                       "build_id": build_id,
                       **({"rehearsal": "ranks share %d GPU(s) over gloo: not a measurement"
                                        % torch.cuda.device_count()} if rehearsal else {})},
            "roofline": {"bound": "hbm",
                         "kernel": "k_extprod + %s (one k-loop iteration of a %d-bootstrap chunk%s)"
                                   % (crt_name, chunk, "; the two kernels of the two lanes' chunks overlap" if lanes == 2 else ""),
                         "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": achieved / PEAK_HBM_GBS,
                         "traffic": traffic,
                         "traffic_ratio": traffic / launch_bytes if traffic else None,
                         "traffic_note": (live["source"] if live else
                                          (("HBM bytes of both launches from the committed PMC passes (%s), "
                                            "not measured in this run: %s" % (ctr["source"], live_why)) if ctr
                                           else "%s; %s" % (why, live_why))),
                         "algorithmic_bytes_per_launch": launch_bytes,
                         "launch_ms": iter_s * 1e3,
                         "launch_ms_source": "HIP events on the ctx stream around whole steps (first to last "
                                             "kernel, both lanes) / chunk-iterations per step; per-kernel "
                                             "samples (every 64th iteration) under `kernels`",
                         "launch_ms_rocprof": rp_ms,
                         "kernels": kern,
                         "whole_job_frac": per_boot * value / world / (PEAK_HBM_GBS * 1e9),
                         "valu": valu_roofline(ctr, (alone["extprod_ms"]) * 1e-3, props.multi_processor_count,
                                               NOMINAL_SCLK_HZ)},
        }
        if host_io:
            res["host_io"] = host_io
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(p, sk, KEY_SEED, args.cpu_threads)
        print(json.dumps(res))
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
