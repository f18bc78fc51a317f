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


_GPU_CHILD = r"""
import sys
import numpy as np
import torch
sys.path.insert(0, sys.argv[1])
import sgfhe_jl_amd as S
mode, blob_path, rank, world, batch, out_path = sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
params = S.Params(64)
eng = S.Engine(params, device=0)
nbytes = eng.key_device_form_bytes()
if mode == "export":          # rank 0 builds the key and exports the device-form blob
    eng.generate_key(np.random.default_rng(5).integers(0, 2, size=params.n, dtype=np.uint64), 9)
    blob = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    eng.export_key_device_form(blob.data_ptr())
    torch.cuda.synchronize()
    blob.cpu().numpy().tofile(blob_path)
else:                         # a rank imports the broadcast blob (header verified) and runs its shard
    blob = torch.from_numpy(np.fromfile(blob_path, dtype=np.uint8)).cuda()
    assert blob.numel() == nbytes
    eng.import_key_device_form(blob.data_ptr())
    rng = np.random.default_rng(77)
    a1 = rng.integers(0, params.r, size=(batch, params.n), dtype=np.uint64)
    b1 = rng.integers(0, params.r, size=batch, dtype=np.uint64)
    a2 = rng.integers(0, params.r, size=(batch, params.n), dtype=np.uint64)
    b2 = rng.integers(0, params.r, size=batch, dtype=np.uint64)
    lo, hi, out = S.distributed.bootstrap_sharded(eng.bootstrap_batch, a1, b1, a2, b2, rank, world)
    assert (lo, hi) == S.distributed.shard_range(batch, rank, world)
    np.save(out_path, out)
eng.close()
"""


def _rank_of_eight(rank, world, port, tmpdir, batch, wave):
    """One of 8 gloo ranks.  The pool admits at most 6 processes on a GPU at once, so a rank never
    opens the GPU itself: its device work (export on rank 0, import + shard on every rank) runs in
    a short-lived child, and the ranks take the GPU in waves of `wave`."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import sgfhe_jl_amd as S
    # A rank must not open the GPU (8 ranks + their GPU children would exceed the guard):
    # dist.barrier() initialises the HIP runtime even on a gloo group, so the barriers here are
    # all-reduces of a CPU tensor, and torch.cuda reports no device inside a rank.
    torch.cuda.is_available = lambda: False
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def cpu_barrier():
        dist.all_reduce(torch.zeros(1))

    def child(mode, blob_path, out_path):
        subprocess.run([sys.executable, "-c", _GPU_CHILD, root, mode, blob_path, str(rank), str(world),
                        str(batch), out_path], check=True, timeout=600)
    mine = os.path.join(tmpdir, "blob_%d.bin" % rank)
    if rank == 0:
        child("export", mine, "")
        blob = torch.from_numpy(np.fromfile(mine, dtype=np.uint8))
        size = torch.tensor([blob.numel()], dtype=torch.int64)
    else:
        size = torch.zeros(1, dtype=torch.int64)
    dist.broadcast(size, src=0)
    if rank != 0:
        blob = torch.empty(int(size.item()), dtype=torch.uint8)
    dist.broadcast(blob, src=0)                              # the one-time key broadcast (gloo here)
    if rank != 0:
        blob.numpy().tofile(mine)
    out_path = os.path.join(tmpdir, "shard_%d.npy" % rank)
    for w in range((world + wave - 1) // wave):
        if rank // wave == w:
            child("shard", mine, out_path)
        cpu_barrier()
    full = S.distributed.gather_outputs(np.load(out_path), batch, world)
    np.save(os.path.join(tmpdir, "full_%d.npy" % rank), full)
    cpu_barrier()
    dist.destroy_process_group()


def test_eight_ranks_on_one_gpu_equal_one_process(tmp_path, S):
    """World size 8 (BASELINE.json config 5's rank count) rehearsed on one GPU: rank 0 generates
    the key and exports the device-form blob, the blob is broadcast to 7 peers, every rank imports
    it (header check) and bootstraps its contiguous shard of a ragged batch (8 does not divide
    203), the shards are all-gathered, and every rank's gathered bytes equal one process doing
    the whole batch.  The 8 ranks share device 0 in waves of 4 short-lived GPU children (the
    pool's process guard), and the broadcast runs over gloo; the RCCL / one-device-per-rank form
    of the same flow is test_two_gpu_shards_equal_one_gpu."""
    import torch.multiprocessing as mp
    world, batch = 8, 203
    mp.spawn(_rank_of_eight, args=(world, _free_port(), str(tmp_path), batch, 4), nprocs=world, join=True)
    params = S.Params(64)
    eng = S.Engine(params, device=0)
    eng.generate_key(np.random.default_rng(5).integers(0, 2, size=params.n, dtype=np.uint64), 9)
    ref = eng.bootstrap_batch(*_inputs(params, batch))
    eng.close()
    sizes = [S.distributed.shard_range(batch, r, world) for r in range(world)]
    assert len({hi - lo for lo, hi in sizes}) == 2           # ragged: 25 and 26
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
    # the ranks proved they talked over RCCL, and the broadcast has a rate
    assert d["config"]["rccl_ranks"] == 2 and d["config"]["collective_backend"] == "nccl"
    assert d["config"]["key_broadcast_gbs"] > 0


@pytest.mark.parametrize("ranks", [2, 4])
def test_bench_ranks_rehearsal_on_one_gpu(ranks):
    """The N > 1 control flow of bench.py on the hardware a one-GPU box has: `--gpus N` starts N
    ranks (torch.distributed.run on 127.0.0.1), rank 0 generates the key, the device-form blob is
    broadcast and imported by the peers, every rank times its shard between barriers, the MAX over
    ranks is taken and rank 0 prints one line with the whole-job value.  The ranks share device 0
    and talk over gloo (SGFHE_BENCH_SHARE_GPU=1: RCCL refuses two ranks on one device), and the
    line carries config.rehearsal; the RCCL form is test_bench_two_gpus_self_launch."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["SGFHE_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ranks), "--config",
                        "params64", "--batch", "512", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["config"]["key_broadcast_s"] > 0 and "rehearsal" in d["config"]
    # all-reduce(SUM) of 1 over the group == --gpus, asserted by bench.py before it prints; a rehearsal over
    # gloo claims no RCCL figure
    assert d["config"]["group_ranks"] == ranks and d["config"]["collective_backend"] == "gloo"
    assert d["config"]["rccl_ranks"] is None and d["config"]["key_broadcast_gbs"] > 0
    assert d["value"] > 0 and abs(d["value"] - ranks * 512 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["host_io"]["equals_device_resident_output"] is True and "cpu_baseline" not in d


def test_rccl_world_of_one_broadcast_and_import(tmp_path, S):
    """The production form of _worker (backend "nccl" = RCCL, device_id given, device-form blob
    broadcast on the device, import, sharded bootstrap, gather) with the one rank a one-GPU box
    allows: RCCL is initialised and carries the broadcast / all-gather / barrier calls of
    sgfhe_jl_amd.distributed on real hardware; what a second device adds is the transport only.
    The gathered bytes equal the plain one-process call."""
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    params = S.Params(64)
    eng = S.Engine(params, device=0)
    eng.generate_key(np.random.default_rng(5).integers(0, 2, size=params.n, dtype=np.uint64), 9)
    ref = eng.bootstrap_batch(*_inputs(params, 37))
    eng.close()
    assert np.load(os.path.join(str(tmp_path), "full_0.npy")).tobytes() == ref.tobytes()


def test_bench_under_torchrun_with_rccl_one_rank():
    """bench.py started exactly as the driver starts its N > 1 runs (python -m torch.distributed.run
    --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...), at
    the N this box has: the rank initialises RCCL with its device, broadcasts the Params(512) key
    blob (168 MB) through it, takes both barriers and the MAX all-reduce on the device, and prints
    the one line."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(root, "bench.py"), "--gpus", "1", "--config", "params512", "--batch",
                        "64", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and "rehearsal" not in d["config"]
    assert d["config"]["key_broadcast_s"] > 0          # the broadcast went through RCCL
    assert d["config"]["rccl_ranks"] == 1 and d["config"]["collective_backend"] == "nccl"   # all-reduce(SUM) of 1 on RCCL
    assert d["config"]["key_broadcast_gbs"] is None    # (no peer: nothing moved)
    assert d["host_io"]["equals_device_resident_output"] is True
