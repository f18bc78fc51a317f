"""N > 1 path on CPU: world_size-2 gloo processes shard a batch with the package's own sharding
code; the shards, gathered, must be byte-identical to the single-process result (SURVEY.md 8e
consistency test).  The per-shard compute is the oracle here (no GPU); on GPUs the same functions
wrap Engine.bootstrap_batch."""

import os
import socket
import sys

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, tmpdir):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import oracle_c
    import sgfhe_jl_amd as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    params = S.Params(64)
    o = oracle_c.Oracle.from_params(params)
    sk = o.private_key(1)
    # rank 0 owns the key; peers receive it by broadcast (CPU stand-in for the device-form blob)
    if rank == 0:
        bkey = o.bootstrap_key(sk, 2, threads=2)
        blob = torch.from_numpy(bkey.view(np.uint8).reshape(-1).copy())
    else:
        blob = torch.empty(params.n * 8 * params.m * 16, dtype=torch.uint8)
    dist.broadcast(blob, src=0)
    bkey = blob.numpy().view(np.uint64).reshape(params.n, 4, 2, params.m, 2)
    batch = 7                                   # ragged on purpose: 3 + 4
    bits = np.random.default_rng(5).integers(0, 2, size=2 * batch).astype(np.uint8)
    a, b = o.lwe_encrypt_bits(sk, bits, 6)
    fn = lambda a1, b1, a2, b2: o.bootstrap_batch(bkey, a1, b1, a2, b2, threads=2)
    lo, hi, out = S.distributed.bootstrap_sharded(fn, a[0::2], b[0::2], a[1::2], b[1::2], rank, world)
    assert (lo, hi) == S.distributed.shard_range(batch, rank, world)
    full = S.distributed.gather_outputs(out, batch, world)
    np.save(os.path.join(tmpdir, "full_%d.npy" % rank), full)
    dist.destroy_process_group()


def test_shard_range_partitions():
    import sgfhe_jl_amd as S
    for batch in (0, 1, 7, 8, 4096, 65536):
        for world in (1, 2, 3, 8):
            edges = [S.distributed.shard_range(batch, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == batch
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_matches_single_process(tmp_path, oc):
    import torch.multiprocessing as mp
    import sgfhe_jl_amd as S
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    params = S.Params(64)
    o = oc.Oracle.from_params(params)
    sk = o.private_key(1)
    bkey = o.bootstrap_key(sk, 2)
    bits = np.random.default_rng(5).integers(0, 2, size=14).astype(np.uint8)
    a, b = o.lwe_encrypt_bits(sk, bits, 6)
    ref = o.bootstrap_batch(bkey, a[0::2], b[0::2], a[1::2], b[1::2])
    for r in range(world):
        full = np.load(os.path.join(str(tmp_path), "full_%d.npy" % r))
        assert full.tobytes() == ref.tobytes()


@pytest.mark.parametrize("gpus", [1, 2, 8])
def test_bench_self_launch_dry_run(gpus):
    """`python bench.py --gpus N` starts its own ranks (torch.distributed.run on 127.0.0.1) and
    relays exactly one JSON line from rank 0; N = 1 runs in-process.  --dry-run swaps the engine
    for a gloo rendezvous so the launcher path runs without a GPU."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus),
                        "--steps", "3", "--warmup", "1", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == gpus and d["steps"] == 3 and d["warmup"] == 1
    assert d["max_over_ranks"] == float(gpus)           # MAX over ranks of (1 + rank)
    # the proof that N ranks talked: all-reduce(SUM) of 1 over the process group, asserted
    # equal to --gpus before the line is printed; the real line carries it as config.rccl_ranks over RCCL
    c = d["config"]
    if gpus > 1:
        assert c["group_ranks"] == gpus and c["collective_backend"] == "gloo"
    else:
        assert c["group_ranks"] is None
    assert c["rccl_ranks"] is None and c["key_broadcast_gbs"] is None      # (gloo rehearsal: no RCCL figure)


def test_bench_refuses_mismatched_world():
    """Started by a launcher with another world size, bench.py exits non-zero instead of
    reporting a line for the wrong N."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
