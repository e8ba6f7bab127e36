"""CPU, world_size 2 over gloo: the N > 1 plumbing bench.py and a training step use (openpcdet_amd/dist.py) -- frame sharding by
rank, the barrier-bracketed timing with MAX over ranks, and the bucketed gradient all-reduce.  No kernel runs here: the hot path
itself exchanges nothing between ranks."""
import os
import socket
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from openpcdet_amd import dist as pdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w = pdist.init_from_env(device=None)
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    res = {}
    # frames sharded like DistributedSampler (pcdet/datasets/__init__.py:56): indices[rank::world], padded by wrap-around
    res["shard"] = pdist.frame_shard(7, rank, world)
    res["shard_nopad"] = pdist.frame_shard(7, rank, world, pad=False)
    # timing bracket: rank 1 is slower; both ranks must report rank 1's time
    done = []

    def step(i):
        time.sleep(0.02 * (rank + 1))
        done.append(i)

    res["elapsed"] = pdist.timed_steps(step, 5, device=None, first=3)
    res["steps"] = done
    # bucketed gradient all-reduce == mean over ranks, parameters without grad skipped, several buckets
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(s)) for s in ((16, 27, 16), (16,), (300, 7), (5,), (64, 64))]
    for j, p in enumerate(params):
        if j != 3:
            p.grad = torch.full_like(p, float(rank + 1)) * (j + 1)
    res["buckets"] = pdist.allreduce_gradients(params, bucket_bytes=4096)
    res["grads"] = [None if p.grad is None else p.grad.clone() for p in params]
    half = [torch.nn.Parameter(torch.zeros(8, dtype=torch.float64))]
    half[0].grad = torch.arange(8, dtype=torch.float64) * (rank + 1)
    pdist.allreduce_gradients(half)
    res["f64"] = half[0].grad.clone()
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert a["shard"] == [0, 2, 4, 6] and b["shard"] == [1, 3, 5, 0]          # 7 frames padded to 8 by wrapping
    assert a["shard_nopad"] == [0, 2, 4, 6] and b["shard_nopad"] == [1, 3, 5]
    assert sorted(a["shard_nopad"] + b["shard_nopad"]) == list(range(7))
    assert a["steps"] == b["steps"] == [3, 4, 5, 6, 7]
    assert a["elapsed"] == b["elapsed"] and a["elapsed"] >= 5 * 0.04          # the slow rank's time, on both
    assert a["buckets"] == b["buckets"] >= 2
    for j in range(5):
        if j == 3:
            assert a["grads"][j] is None and b["grads"][j] is None
        else:
            want = torch.full_like(a["grads"][j], 1.5 * (j + 1))
            assert torch.equal(a["grads"][j], want) and torch.equal(b["grads"][j], want)
    assert torch.equal(a["f64"], torch.arange(8, dtype=torch.float64) * 1.5) and torch.equal(a["f64"], b["f64"])


def test_single_process_is_a_no_op():
    assert pdist.frame_shard(5, 0, 1) == [0, 1, 2, 3, 4]
    p = torch.nn.Parameter(torch.zeros(3))
    p.grad = torch.ones(3)
    assert pdist.allreduce_gradients([p]) == 0 and torch.equal(p.grad, torch.ones(3))
    seen = []
    assert pdist.timed_steps(seen.append, 3) >= 0 and seen == [0, 1, 2]
