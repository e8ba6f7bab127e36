"""Row-span statistics of the rulebooks of the nuScenes plan: per 256-site (and 128-site) tile and dz group, span = max - min + 1 of
the neighbour rows; what fraction of tiles / pairs fits a window of R rows, and what a 2-cluster window (largest gap removed) needs."""
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
for r in eng.conv_records:
    nbr = r["nbr"]
    if r["kvol"] != 27 or nbr.data_ptr() in seen:
        continue
    seen.add(nbr.data_ptr())
    n = min(int(r["level_out"].n_dev.item()), r["level_out"].cap)
    for TM in (256, 128):
        nt = n // TM
        v = nbr[:, :nt * TM].view(3, 9, nt, TM).permute(0, 2, 1, 3).reshape(3, nt, 9 * TM).long()
        valid = v >= 0
        big = torch.where(valid, v, torch.full_like(v, 1 << 40))
        small = torch.where(valid, v, torch.full_like(v, -1))
        lo, hi = big.min(-1).values, small.max(-1).values
        span = torch.where(hi >= 0, hi - lo + 1, torch.zeros_like(hi)).float()
        # 2 clusters: sort, largest gap
        s = torch.sort(big, -1).values
        cnt = valid.sum(-1)
        d = (s[..., 1:] - s[..., :-1])
        idx = torch.arange(d.shape[-1], device=dev)[None, None, :]
        d = torch.where(idx < (cnt[..., None] - 1), d, torch.zeros_like(d))
        gap = d.max(-1).values.float()
        span2 = torch.clamp(span - gap + 1, min=0)
        q = torch.tensor([0.5, 0.9, 0.99, 1.0], device=dev)
        print("%s rows=%d TM=%d  span pct50/90/99/max:" % (r["name"], n, TM), [int(x) for x in torch.quantile(span.flatten(), q).tolist()],
              " two-cluster:", [int(x) for x in torch.quantile(span2.flatten(), q).tolist()],
              " frac(span<=440)=%.3f frac2(<=440)=%.3f frac(<=896)=%.3f frac2(<=896)=%.3f" % (
                  (span <= 440).float().mean().item(), (span2 <= 440).float().mean().item(),
                  (span <= 896).float().mean().item(), (span2 <= 896).float().mean().item()))
