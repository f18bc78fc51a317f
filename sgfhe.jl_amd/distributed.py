"""Multi-GPU plumbing: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The gate bootstraps of a batch are independent (the k-loop inside one bootstrap is sequential
and never split, /root/reference/src/fhe.jl:579-582), so the batch shards contiguously across
ranks with NO collective on the data path.  The only collective is the one-time broadcast of
the device-form bootstrap key from the rank that built it.
"""

import numpy as np


def shard_range(batch, rank, world):
    """Contiguous shard [lo, hi) of rank `rank`: bootstraps [rank * batch / world, ...)."""
    return (rank * batch) // world, ((rank + 1) * batch) // world


def broadcast_key(engine, src=0, group=None):
    """One-time RCCL broadcast of the device-form key blob (64-byte header + payload,
    sgfhe_bkey_export/import_device_form) from rank `src` to every other rank of the group; the
    import verifies the header against the receiving ctx.  Returns (bytes, seconds)."""
    import time
    import torch
    import torch.distributed as dist
    nbytes = engine.key_device_form_bytes()
    blob = torch.empty(nbytes, dtype=torch.uint8, device=torch.device("cuda", engine.device))
    if dist.get_rank(group) == src:
        engine.export_key_device_form(blob.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dist.broadcast(blob, src=src, group=group)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist.get_rank(group) != src:
        engine.import_key_device_form(blob.data_ptr())
    return nbytes, dt


def bootstrap_sharded(bootstrap_fn, a1, b1, a2, b2, rank, world):
    """Run `bootstrap_fn(a1, b1, a2, b2) -> [shard][3][n+1]` on this rank's contiguous shard of a
    host batch.  Returns (lo, hi, out_shard).  `bootstrap_fn` is `Engine.bootstrap_batch` in
    production."""
    a1 = np.asarray(a1)
    lo, hi = shard_range(a1.shape[0], rank, world)
    out = bootstrap_fn(a1[lo:hi], np.asarray(b1)[lo:hi], np.asarray(a2)[lo:hi], np.asarray(b2)[lo:hi])
    return lo, hi, out


def gather_outputs(out_shard, batch, world, group=None):
    """Optional: all-gather the per-rank output shards (host arrays) into the full batch on every
    rank.  Not part of the timed path."""
    import torch
    import torch.distributed as dist
    parts = [None] * world
    dist.all_gather_object(parts, np.ascontiguousarray(out_shard), group=group)
    full = np.concatenate(parts, axis=0)
    assert full.shape[0] == batch
    return full
