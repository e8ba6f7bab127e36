"""How much MFMA work of the 64- and 128-channel submanifold layers multiplies rows that do not exist, and how much of it whole
(32-site block, offset) pairs could skip -- in row order, and with the 256 sites of a tile re-ordered by their neighbour mask."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["PVX_ENGINE_FAR_ONLY"] = "0"
import bench
from openpcdet_amd import engine, synth

dev = torch.device("cuda", 0)
cfg = synth.CONFIGS[bench.WORKLOAD]
B = 16
net = bench.build_model(cfg, dev)
eng = engine.HotPathEngine(net, B, cfg["range"], cfg["voxel"], cfg["T"], cfg["max_voxels"], cfg["C"], cfg["n_points"], dtype=torch.bfloat16, device=dev,
                           level_capacity_factor=2.0)
for b in range(B):
    eng.points[b].copy_(torch.from_numpy(synth.config_cloud(bench.WORKLOAD, 3 + b)).to(dev))
eng.voxelize(); eng.backbone(); torch.cuda.synchronize()
seen = set()
for r in eng.conv_records:
    nbr = r["nbr"]
    if r["kvol"] != 27 or nbr.data_ptr() in seen or not r["subm"] or r["cin"] < 32:
        continue
    seen.add(nbr.data_ptr())
    n = min(int(r["level_out"].n_dev.item()), r["level_out"].cap)
    nt = n // 256
    valid = (nbr[:, :nt * 256] >= 0).view(27, nt, 256).permute(1, 2, 0).contiguous()          # [tile, site, k]
    useful = valid.float().mean().item()
    for blk in (32, 64):
        nb = 256 // blk
        in_order = valid.view(nt, nb, blk, 27).any(2).float().mean().item()
        # sort the sites of a tile by their mask read as an integer (several bit orders: dz-major as stored, and by popcount first)
        w = (2 ** torch.arange(27, device=dev, dtype=torch.int64))
        key = (valid.long() * w).sum(-1)
        best = in_order
        for name, k2 in (("mask", key), ("popcount", valid.sum(-1).long() * (1 << 27) + key),
                         ("dz groups", (valid[:, :, :9].any(-1).long() * 4 + valid[:, :, 18:].any(-1).long() * 2) * (1 << 27) + key)):
            order = torch.argsort(k2, dim=1)
            v2 = torch.gather(valid, 1, order[:, :, None].expand(-1, -1, 27))
            frac = v2.view(nt, nb, blk, 27).any(2).float().mean().item()
            best = min(best, frac)
            print("%s rows %d: block %d, sites sorted by %-9s -> (block, offset) pairs that need MFMAs: %.3f" % (r["name"], n, blk, name, frac))
        print("%s: useful pairs %.3f of 27 N; blocks of %d in row order need %.3f, best re-order %.3f" % (r["name"], useful, blk, in_order, best))
