#!/usr/bin/env python3
"""VectorPool ops at PV-RCNN++ sizes on one GPU: B frames x N voxel centres, M keypoints per frame,
3x3x3 local grid.  Prints one JSON line per op: device time (HIP events, median of --iters), support-point tests per second,
the bytes the brute-force search streams through the caches (M x n x 12) and the compulsory HBM bytes, plus the CPU oracle's
time on a bounded sample of the same input (tests/ and this tool are the only users of oracle/)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpcdet_amd import vector_pool as vp  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--support", type=int, default=60000, help="voxel centres per frame (x_conv3 of a Waymo frame)")
    ap.add_argument("--new", type=int, default=4096, help="keypoints per frame")
    ap.add_argument("--channels", type=int, default=32)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--cpu-points", type=int, default=256, help="new points of the CPU oracle sample (0 = skip)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    r = np.random.default_rng(3)
    B, N, M, C = args.batch, args.support, args.new, args.channels
    xyz = (r.uniform(-75, 75, (B * N, 3)) * [1, 1, 0.03]).astype(np.float32)
    feat = r.normal(size=(B * N, C)).astype(np.float32)
    new = xyz[r.integers(0, N, B * M) + np.repeat(np.arange(B) * N, M)] + r.normal(0, 0.2, (B * M, 3)).astype(np.float32)
    new = new.astype(np.float32)
    xc, nc = np.full(B, N, np.int32), np.full(B, M, np.int32)
    t_xyz, t_feat, t_new, t_xc, t_nc = (torch.from_numpy(a).to(dev) for a in (xyz, feat, new, xc, nc))
    R, grids, G = 2.4, (3, 3, 3), 27
    centers = vp.VectorPoolAggregationModule.get_dense_voxels_by_center(t_new, R, grids).contiguous()
    ops = {
        "three_nn (list + 3-NN, radius 2R, cube)": lambda: vp.three_nn_for_vector_pool_by_two_step(t_xyz, t_xc, t_new, centers, t_nc, R, -1, 0, 1000, G, 2.0),
        "vector_pool avg": lambda: vp.vector_pool_with_voxel_query_op(t_xyz, t_xc, t_feat, t_new, t_nc, 3, 3, 3, R, C, 1, 100, -1, 0, 0),
        "vector_pool first choice": lambda: vp.vector_pool_with_voxel_query_op(t_xyz, t_xc, t_feat, t_new, t_nc, 3, 3, 3, R, C, 1, 100, -1, 0, 1),
    }
    tests = B * M * N
    for name, fn in ops.items():
        ms = timed(fn, args.iters)
        print(json.dumps({"op": name, "ms": round(ms, 3), "frames": B, "support_per_frame": N, "new_per_frame": M, "channels": C,
                          "point_tests_per_s": round(tests / ms * 1e3, 0), "streamed_GBps": round(tests * 12 / ms / 1e6, 1),
                          "compulsory_MB": round((B * N * (12 + 4 * C) + B * M * G * (4 * C + 16)) / 1e6, 1),
                          "note": "includes the host read of the count (one .item()) each op makes"}), flush=True)
    if args.cpu_points > 0:
        from oracle import loader
        m = min(args.cpu_points, M)
        sub, snc = new[:m], np.array([m] + [0] * (B - 1), np.int32)
        t0 = time.perf_counter()
        loader.vp_three_nn(xyz, xc, sub, centers[:m].cpu().numpy(), snc, 2 * R, -1, 0)
        t1 = time.perf_counter()
        loader.vector_pool(xyz, feat, xc, sub, snc, C, grids, R, 1, -1, 0, 0, 10 ** 7)
        t2 = time.perf_counter()
        print(json.dumps({"cpu_oracle": "1 thread, %d new points of frame 0" % m, "three_nn_ms": round((t1 - t0) * 1e3, 2),
                          "vector_pool_avg_ms": round((t2 - t1) * 1e3, 2),
                          "scaled_to_all_new_points_ms": [round((t1 - t0) * 1e3 * B * M / m, 1), round((t2 - t1) * 1e3 * B * M / m, 1)]}), flush=True)


if __name__ == "__main__":
    main()
