"""As span_stats.py, but per (dz, dy) SUBGROUP of three offsets (dx = -1, 0, 1): the rows a 256-site tile needs from one y-line
offset.  Offsets are already walked in this order (k = (dz * 3 + dy) * 3 + dx), so per-subgroup windows keep the MFMA order."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from openpcdet_amd import engine, synth

dev = torch.device("cuda", 0)
cfg = synth.CONFIGS[bench.WORKLOAD]
B = 16
net = bench.build_model(cfg, dev)
eng = engine.HotPathEngine(net, B, cfg["range"], cfg["voxel"], cfg["T"], cfg["max_voxels"], cfg["C"], cfg["n_points"],
                           dtype=torch.bfloat16, device=dev, level_capacity_factor=2.0)
for b in range(B):
    eng.points[b].copy_(torch.from_numpy(synth.config_cloud(bench.WORKLOAD, 3 + b)).to(dev))
eng.voxelize(); eng.backbone(); torch.cuda.synchronize()
seen = set()
q = torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], device=dev)
for r in eng.conv_records:
    nbr = r["nbr"]
    if r["kvol"] != 27 or nbr.data_ptr() in seen:
        continue
    seen.add(nbr.data_ptr())
    n = min(int(r["level_out"].n_dev.item()), r["level_out"].cap)
    TM = 256
    nt = n // TM
    for name, groups, per in (("dz group (9 offsets)", 3, 9), ("(dz, dy) subgroup (3 offsets)", 9, 3)):
        v = nbr[:, :nt * TM].view(groups, per, nt, TM).permute(0, 2, 1, 3).reshape(groups, nt, per * TM).long()
        valid = v >= 0
        lo = torch.where(valid, v, torch.full_like(v, 1 << 40)).min(-1).values
        hi = torch.where(valid, v, torch.full_like(v, -1)).max(-1).values
        span = torch.where(hi >= 0, hi - lo + 1, torch.zeros_like(hi)).float()
        rows_total = span.clamp(max=4096).sum(0).mean().item()
        print("%-18s %-30s span pct50/90/99/99.9/max: %s  frac<=320 %.4f  <=384 %.4f  <=448 %.4f  <=512 %.4f   rows loaded per tile (sum of spans) %.0f"
              % (r["name"], name, [int(x) for x in torch.quantile(span.flatten(), q).tolist()], (span <= 320).float().mean().item(),
                 (span <= 384).float().mean().item(), (span <= 448).float().mean().item(), (span <= 512).float().mean().item(), rows_total))
