#!/usr/bin/env python3
"""Training step of BASELINE.json configs[3] on ONE GPU: voxelize -> MeanVFE -> VoxelResBackBone8x (train mode, bf16 autocast) ->
HeightCompression -> loss -> backward (dgrad + wgrad kernels) -> gradient all-reduce (DistributedDataParallel over RCCL, world
size = the ranks launched) -> AdamW step, on Waymo-range multi-sweep frames (tools/train_utils/train_utils.py:55-60,
tools/train.py:130-131,162).  Prints ONE JSON line: ms per step, frames/s, and per-kernel-class device time with the
useful-flop fraction of the dense bf16 MFMA peak for forward, dgrad and wgrad (flops = 2 P Cin Cout each, P = valid pairs).

    python tools/bench_train.py --batch 2 --points 250000 --steps 10 [--sync-bn] [--no-ddp]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_train.py ...

The 8-GPU half of configs[3] is the driver's to run; correctness of the N > 1 path is covered by tests/test_dist_gloo.py."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openpcdet_amd import backbones, synth, voxelize  # noqa: E402
from openpcdet_amd import dist as pdist  # noqa: E402

PEAK = 2500.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--points", type=int, default=250000, help="points per frame (4 sweeps x 250k = the 1M-point multi-frame input at 4)")
    ap.add_argument("--sweeps", type=int, default=1, help="frames merged into one input cloud")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sync-bn", action="store_true")
    ap.add_argument("--no-ddp", action="store_true")
    ap.add_argument("--no-fused-opt", action="store_true", help="torch's default (foreach) AdamW instead of the fused one")
    args = ap.parse_args()
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if "RANK" not in os.environ:      # single process: still a real process group, so DDP / SyncBN run their collectives (world 1)
        os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"))
    rank, world = pdist.init_from_env(dev)
    if not torch.distributed.is_initialized():
        torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    cfg = synth.CONFIGS["waymo_centerpoint"]
    grid = [int(round((cfg["range"][3 + j] - cfg["range"][j]) / cfg["voxel"][j])) for j in range(3)]
    torch.manual_seed(5)
    net = backbones.VoxelResBackBone8x(input_channels=cfg["C"], grid_size=np.array(grid)).to(dev).train()
    if args.sync_bn:
        net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)
    hc = backbones.HeightCompression()
    model = net if args.no_ddp else torch.nn.parallel.DistributedDataParallel(net, device_ids=[local_rank])
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, fused=not args.no_fused_opt)      # one multi-tensor kernel per step
    gen = voxelize.PointToVoxel(cfg["voxel"], cfg["range"], cfg["C"], cfg["max_voxels"], cfg["T"])
    clouds = []
    for b in range(args.batch):
        parts = [synth.config_cloud("waymo_centerpoint", 100 * rank + 10 * b + s, n_points=args.points) for s in range(args.sweeps)]
        clouds.append(torch.from_numpy(np.concatenate(parts)).to(dev))

    def make_batch():
        feats, coords = [], []
        for b, pts in enumerate(clouds):
            o = gen.run(pts, want_voxels=False, want_mean=True)
            m = int(o["num_voxels"].item())
            feats.append(o["mean"][:m])
            coords.append(torch.cat([torch.full((m, 1), b, dtype=torch.int32, device=dev), o["coords"][:m]], 1))
        return {"voxel_features": torch.cat(feats), "voxel_coords": torch.cat(coords).float(), "batch_size": args.batch}

    def step():
        bd = make_batch()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            bd = model(bd)
            bev = hc(bd)["spatial_features"]
            loss = (bev.float() ** 2).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 10.0)      # GRAD_NORM_CLIP, train_utils.py:59-60
        opt.step()
        return loss, bd

    for _ in range(args.warmup):
        loss, bd = step()
    torch.cuda.synchronize()
    assert all(p.grad is not None for p in net.parameters()), "a parameter got no gradient"
    elapsed = pdist.timed_steps(lambda i: step(), args.steps, device=dev, first=args.warmup)
    loss, bd = step()
    torch.cuda.synchronize()
    # useful flops of one step: 2 P Cin Cout per conv forward (P = valid rulebook pairs); dgrad and wgrad cost the same each
    from openpcdet_amd.spconv.pytorch.conv import SparseConvolution
    books = bd["encoded_spconv_tensor"].indice_dict
    pairs = {k: int((v.nbr >= 0).sum().item()) for k, v in books.items() if hasattr(v, "nbr")}
    fwd = sum(2 * pairs[m.indice_key] * m.in_channels * m.out_channels
              for m in net.modules() if isinstance(m, SparseConvolution) and m.indice_key in pairs)
    n_vox = int(bd["voxel_coords"].shape[0])
    out = {"metric": "training step VoxelResBackBone8x (Waymo range, bf16 autocast), frames/s", "value": round(world * args.batch * args.steps / elapsed, 2),
           "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "dtype": "bf16", "data": "synthetic",
           "config": {"workload": "CenterPoint/PV-RCNN++ Waymo-range training step (BASELINE.json configs[3]): %d frames x %d points x %d sweeps, "
                                  "0.1 m voxels" % (args.batch, args.points, args.sweeps), "voxels": n_vox, "ddp": not args.no_ddp, "sync_bn": args.sync_bn,
                      "optimizer": "AdamW" + ("" if args.no_fused_opt else " (fused)"), "grad_clip": 10.0},
           "loss": float(loss.item()), "flops_fwd": fwd, "flops_step_fwd_dgrad_wgrad": 3 * fwd,
           "mfma_frac_whole_step": round(3 * fwd / (elapsed / args.steps) / 1e12 / PEAK, 5)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
