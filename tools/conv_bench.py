#!/usr/bin/env python3
"""Per-launch device time of every sparse-conv layer of the nuScenes VoxelResBackBone8x plan, serial (one stream, HIP events
around each launch), with rows / valid pairs / useful flops / fraction of the dense bf16 MFMA peak and compulsory bytes.

    python tools/conv_bench.py --batch 16 --passes 10 [--json out.json]

Kernel variants are selected by the environment (PVX_CONV_SLAB, PVX_CONV_V2, ...), so one GPU call can A/B several."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from openpcdet_amd import engine, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--passes", type=int, default=10)
    ap.add_argument("--cap-factor", type=float, default=2.0)
    ap.add_argument("--json", default=None)
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg = synth.CONFIGS[bench.WORKLOAD]
    B = args.batch
    net = bench.build_model(cfg, dev)
    eng = engine.HotPathEngine(net, B, cfg["range"], cfg["voxel"], cfg["T"], cfg["max_voxels"], cfg["C"], cfg["n_points"],
                               dtype=torch.bfloat16, device=dev, level_capacity_factor=args.cap_factor)
    frames = [torch.from_numpy(synth.config_cloud(bench.WORKLOAD, 3 + i)).to(dev) for i in range(B)]
    for b in range(B):
        eng.points[b].copy_(frames[b])
    ev = {}

    def hook(name, fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        ev.setdefault(name, []).append((a, b))

    for p in range(args.passes + 2):
        if p == 2:
            ev.clear()
        eng.voxelize()
        eng.backbone(hook=hook)
    torch.cuda.synchronize()
    assert not eng.overflowed()
    flops, per = eng.conv_flops()
    rows = []
    tot_ms = 0.0
    e = 2
    for (name, n, pairs, f), rec in zip(per, eng.conv_records):
        ts = sorted(a.elapsed_time(b) for a, b in ev[name])
        med = ts[len(ts) // 2]
        tot_ms += med
        n_in = min(int(rec["level_out"].n_dev.item()), rec["level_out"].cap) if rec["subm"] else None
        byts = n * rec["cout"] * e + n * rec["kvol"] * 4 + (n_in if n_in is not None else 0) * rec["cin"] * e
        rows.append(dict(name=name, rows=n, pairs=pairs, pairs_per_row=round(pairs / max(n, 1), 2), us=round(med * 1e3, 1),
                         min_us=round(ts[0] * 1e3, 1), tflops=round(f / (med * 1e-3) / 1e12, 1),
                         frac=round(f / (med * 1e-3) / 1e12 / bench.PEAK_BF16_TFLOPS, 4),
                         gbps_alg=round(byts / (med * 1e-3) / 1e9, 0)))
    geo = {k: round(sorted(a.elapsed_time(b) for a, b in v)[len(v) // 2] * 1e3, 1) for k, v in ev.items() if not k.startswith("conv") or k.startswith("conv_out")}
    out = dict(tag=args.tag, batch=B, conv_ms=round(tot_ms, 4), flops=flops, frac=round(flops / (tot_ms * 1e-3) / 1e12 / bench.PEAK_BF16_TFLOPS, 4),
               rows_per_level=eng.level_rows(), layers=rows, other_us=geo)
    print("%s B=%d conv_ms=%.3f frac=%.4f" % (args.tag, B, tot_ms, out["frac"]))
    for r in rows:
        print("  %-22s rows=%8d ppr=%5.2f  %8.1f us (min %8.1f)  %7.1f TF  frac=%.4f  alg %5.0f GB/s" % (
            r["name"], r["rows"], r["pairs_per_row"], r["us"], r["min_us"], r["tflops"], r["frac"], r["gbps_alg"]))
    print("  other:", geo)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
