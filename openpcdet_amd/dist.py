"""Multi-GPU plumbing of the hot path: one process per GPU, frames sharded by rank, no data-path collective.

What the reference does: `DistributedSampler` hands rank r the frames `indices[r::world]`
(pcdet/datasets/__init__.py:39-59,77-82), each process owns one GPU (pcdet/utils/common_utils.py:194-195), inference exchanges
nothing, and the training step has exactly one exchange -- the DDP gradient all-reduce (tools/train.py:162).  Here:

  init_from_env()          torch.distributed over RCCL ("nccl" on ROCm) when launched by torchrun, gloo on CPU (tests)
  frame_shard()            the sampler's partition rule, padded like the sampler pads so every rank runs the same step count
  timed_steps()            barrier + device sync on both sides, MAX over ranks: the bracket bench.py's contract asks for
  allreduce_gradients()    bucketed, averaged gradient all-reduce: few large collectives sized for per-link-bound xGMI rings
"""
import os
import time

import torch
import torch.distributed as dist


def init_from_env(device=None):
    """-> (rank, world).  Initialises the default process group when WORLD_SIZE > 1 (MASTER_ADDR/PORT, RANK from the env)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if device is not None and torch.device(device).type == "cuda":
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group("gloo")
    return rank, world


def frame_shard(num_frames, rank, world, pad=True):
    """indices of the frames rank `rank` processes: indices[rank::world]; with pad (the sampler's default) the index list is
    first extended by wrapping around so that it divides evenly."""
    idx = list(range(num_frames))
    if pad and num_frames > 0 and num_frames % world:
        idx += idx[:world - num_frames % world]
    return idx[rank::world]


def _sync(device):
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)


def timed_steps(step, steps, device=None, first=0):
    """run step(i) for i in [first, first + steps) between two (barrier + device sync) fences; returns the wall time of the
    slowest rank in seconds (the same value on every rank)."""
    on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    _sync(device)
    if on:
        dist.barrier()
    _sync(device)
    t0 = time.perf_counter()
    for i in range(first, first + steps):
        step(i)
    _sync(device)
    if on:
        dist.barrier()
    _sync(device)
    elapsed = time.perf_counter() - t0
    if on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def allreduce_gradients(params, bucket_bytes=64 << 20):
    """average .grad of `params` over all ranks in place.  Gradients are packed into flat buckets of ~bucket_bytes per dtype
    (the VoxelResBackBone8x gradient is ~34 MB fp32: one bucket) so the ring runs a few large transfers instead of one
    small one per parameter; parameters without a gradient are skipped (every rank must skip the same ones)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    world = dist.get_world_size()
    by_dtype = {}
    for p in params:
        if p.grad is not None:
            by_dtype.setdefault((p.grad.dtype, p.grad.device), []).append(p.grad)
    launched = 0
    for (_, _), grads in by_dtype.items():
        bucket, size = [], 0
        def flush():
            nonlocal bucket, size, launched
            if not bucket:
                return
            flat = torch.cat([g.reshape(-1) for g in bucket])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.div_(world)
            off = 0
            for g in bucket:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
            launched += 1
            bucket, size = [], 0
        for g in grads:
            bucket.append(g)
            size += g.numel() * g.element_size()
            if size >= bucket_bytes:
                flush()
        flush()
    return launched
