"""Load balance of the slab kernels' static tile walk (XCD x gets the x-th eighth of the tiles, its 32 workgroups stride through it):
per-tile work proxy = number of (wave, offset) pairs that issue MFMAs (some lane of the wave's 32 sites has that neighbour) + a constant,
summed per workgroup; reports max / mean over workgroups and over XCDs, against a dynamic walk's bound (mean + one tile)."""
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
    nt = (n + 255) // 256
    pad = nt * 256 - n
    v = nbr[:, :n]
    if pad:
        v = torch.cat([v, torch.full((27, pad), -1, dtype=v.dtype, device=dev)], 1)
    on = (v.view(27, nt, 8, 32) >= 0).any(-1).sum((0, 2)).float()        # per tile: (wave, offset) pairs with MFMAs, 0..216
    for fixed in (40.0, 100.0):                                           # fixed per-tile cost in the same unit (loads, barriers, epilogue)
        cost = (on + fixed).cpu().numpy()
        wg_sum = np.zeros(256)
        for x in range(8):
            t0, t1 = nt * x // 8, nt * (x + 1) // 8
            for w in range(32):
                wg_sum[x * 32 + w] = cost[t0 + w:t1:32].sum()
        xcd = wg_sum.reshape(8, 32)
        print("%s tiles=%d on/tile mean %.0f sd %.0f | fixed %.0f: WG max/mean %.3f, XCD-mean max/mean %.3f, within-XCD max/mean %.3f, dynamic bound %.3f"
              % (r["name"], nt, on.mean().item(), on.std().item(), fixed, wg_sum.max() / wg_sum.mean(), xcd.mean(1).max() / wg_sum.mean(),
                 (xcd.max(1) / xcd.mean(1)).max(), 1 + cost.max() / wg_sum.mean()))
