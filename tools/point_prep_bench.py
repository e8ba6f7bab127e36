"""Timing of the point-level preparation (csrc/point_prep.hip) beside the torch ops it replaced and the numpy steps of the
reference (oracle/point_prep.py, timed as the CPU baseline: test infrastructure used as a checker/baseline only).

    python tools/point_prep_bench.py [--points 300000] [--c 5] [--iters 50]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpcdet_amd import data_augmentor as A, data_processor  # noqa: E402
from oracle import point_prep as O  # noqa: E402


def gpu_ms(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=300000)
    ap.add_argument("--c", type=int, default=5)
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    n, c = args.points, args.c
    r = np.random.default_rng(0)
    p = r.uniform(-1, 1, (n, c)).astype(np.float32)
    p[:, :3] *= np.array([60, 60, 4], np.float32)
    rng = (-54.0, -54.0, 54.0, 54.0)
    ops = [(A.FLIP_X, 0, 0, 0), A.rotate_op(0.3), (A.SCALE, 1.02, 0, 0), (A.TRANSLATE, 0.1, 0.2, 0.0)]
    d = torch.from_numpy(p).cuda()
    perm = torch.randperm(n, device="cuda")

    def torch_mask_shuffle():
        m = (d[:, 0] >= rng[0]) & (d[:, 0] <= rng[2]) & (d[:, 1] >= rng[1]) & (d[:, 1] <= rng[3])
        k = d[m]
        return k[torch.randperm(k.shape[0], device="cuda")]

    def hip_mask_shuffle():
        return data_processor.prepare_points(d, torch.randperm(n, device="cuda"), rng)

    def hip_mask_shuffle_seeded():
        return data_processor.prepare_points(d, None, rng, shuffle_seed=int(np.random.randint(1, 2 ** 63 - 1, dtype=np.int64)))

    out = {"points": n, "c": c,
           "transform_4ops_ms": gpu_ms(lambda: A.transform_points(d, ops), args.iters),
           "prepare_mask_shuffle_kernels_ms": gpu_ms(lambda: data_processor.prepare_points(d, perm, None), args.iters),
           "prepare_seeded_kernels_ms": gpu_ms(lambda: data_processor.prepare_points(d, None, None, shuffle_seed=5), args.iters),
           "mask_shuffle_hip_seeded_with_count_read_ms": gpu_ms(hip_mask_shuffle_seeded, args.iters),
           "mask_shuffle_hip_with_randperm_and_count_read_ms": gpu_ms(hip_mask_shuffle, args.iters),
           "mask_shuffle_torch_ops_ms": gpu_ms(torch_mask_shuffle, args.iters)}
    # bytes one prepare moves: permutation entry + flag write and read + the row in and out
    bytes_prepare = n * (8 + 2 + 2 * 4 * c)
    out["prepare_GBps"] = bytes_prepare / out["prepare_mask_shuffle_kernels_ms"] / 1e6
    out["transform_GBps"] = n * 24 / out["transform_4ops_ms"] / 1e6
    t = time.perf_counter()
    reps = 5
    for _ in range(reps):
        q = O.transform_points(p, ops)
        O.prepare(q, np.random.permutation(n), rng)
    out["cpu_numpy_transform_mask_shuffle_ms"] = (time.perf_counter() - t) / reps * 1e3
    print(json.dumps(out))


if __name__ == "__main__":
    main()
