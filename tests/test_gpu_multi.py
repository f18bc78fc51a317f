"""Two-device path on real hardware (SURVEY.md 8e consistency test): rank 0 generates the key,
exports the device-form blob, RCCL broadcasts it, the peer imports it (header verified), each
rank bootstraps its contiguous shard, and the gathered bytes equal the one-GPU result.
Skipped on a one-GPU box; the N > 1 control flow is also covered on the CPU by
tests/test_distributed_cpu.py."""

import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _inputs(params, batch):
    rng = np.random.default_rng(77)
    return (rng.integers(0, params.r, size=(batch, params.n), dtype=np.uint64),
            rng.integers(0, params.r, size=batch, dtype=np.uint64),
            rng.integers(0, params.r, size=(batch, params.n), dtype=np.uint64),
            rng.integers(0, params.r, size=batch, dtype=np.uint64))


def _worker(rank, world, port, tmpdir, one_gpu=False):
    """One rank of the sharded job.  one_gpu: both ranks drive device 0 and the key blob travels
    over gloo (RCCL refuses two ranks on one device); everything else is the production path."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import sgfhe_jl_amd as S
    dev = 0 if one_gpu else rank
    torch.cuda.set_device(dev)
    if one_gpu:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", dev))
    params = S.Params(64)
    eng = S.Engine(params, device=dev)
    sk = np.random.default_rng(5).integers(0, 2, size=params.n, dtype=np.uint64)
    if rank == 0:
        eng.generate_key(sk, 9)
    nbytes, seconds = S.distributed.broadcast_key(eng, src=0)
    assert nbytes == eng.key_device_form_bytes() and seconds > 0
    batch = 37                                               # ragged shards
    a1, b1, a2, b2 = _inputs(params, batch)
    lo, hi, out = S.distributed.bootstrap_sharded(eng.bootstrap_batch, a1, b1, a2, b2, rank, world)
    assert (lo, hi) == S.distributed.shard_range(batch, rank, world)
    full = S.distributed.gather_outputs(out, batch, world)
    np.save(os.path.join(tmpdir, "full_%d.npy" % rank), full)
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


def test_two_ranks_on_one_gpu_equal_one_process(tmp_path, S):
    """The multi-rank flow on the hardware this box has: two processes (one ctx each on device 0),
    rank 0 generates the key and exports the device-form blob, the blob is broadcast (gloo here),
    rank 1 imports it after the header check, each rank bootstraps its contiguous shard, and the
    gathered outputs are byte-identical to one process doing the whole batch."""
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), True), nprocs=world, join=True)
    params = S.Params(64)
    eng = S.Engine(params, device=0)
    eng.generate_key(np.random.default_rng(5).integers(0, 2, size=params.n, dtype=np.uint64), 9)
    ref = eng.bootstrap_batch(*_inputs(params, 37))
    eng.close()
    for r in range(world):
        full = np.load(os.path.join(str(tmp_path), "full_%d.npy" % r))
        assert full.tobytes() == ref.tobytes()


def test_two_gpu_shards_equal_one_gpu(tmp_path, S):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    params = S.Params(64)
    eng = S.Engine(params, device=0)
    eng.generate_key(np.random.default_rng(5).integers(0, 2, size=params.n, dtype=np.uint64), 9)
    ref = eng.bootstrap_batch(*_inputs(params, 37))
    eng.close()
    for r in range(world):
        full = np.load(os.path.join(str(tmp_path), "full_%d.npy" % r))
        assert full.tobytes() == ref.tobytes()


def test_bench_two_gpus_self_launch():
    """python bench.py --gpus 2 prints one line with n_gpus = 2 and a measured key broadcast."""
    import json
    import subprocess
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config",
                        "params64", "--batch", "512", "--steps", "1", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["key_broadcast_s"] > 0 and d["value"] > 0
