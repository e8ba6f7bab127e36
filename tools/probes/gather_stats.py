"""Gathered windows in the plan's tile books: per book, the share of live (tile, dz) phases that are contiguous and fit, gathered (and
whether the list holds every row), or still carry far codes; list lengths."""
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
eng = engine.HotPathEngine(net, B, cfg["range"], cfg["voxel"], cfg["T"], cfg["max_voxels"], cfg["C"], cfg["n_points"], dtype=torch.bfloat16, device=dev,
                           level_capacity_factor=2.0)
for b in range(B):
    eng.points[b].copy_(torch.from_numpy(synth.config_cloud(bench.WORKLOAD, 3 + b)).to(dev))
eng.voxelize(); eng.backbone(); torch.cuda.synchronize()
seen = set()
for r in eng.conv_records:
    if r["book"] is None or r["book"].data_ptr() in seen:
        continue
    seen.add(r["book"].data_ptr())
    lvl = r["level_out"]
    n = min(int(lvl.n_dev.item()), lvl.cap)
    tiles = (n + 255) // 256
    tile_bytes = eng.L.pvx_tilebook_bytes(256)
    blocks = r["book"][:eng.L.pvx_tilebook_bytes(lvl.cap)].view(torch.int32).view(-1, tile_bytes // 4)[:tiles].cpu().numpy().view(np.uint32)
    flags = blocks[:, 6]
    cnt = blocks[:, [1, 3, 5]]
    codes = blocks[:, 8:].copy().view(np.uint16).reshape(tiles, 27, 256)
    live = np.stack([(flags >> g) & 1 for g in range(3)], 1).astype(bool)
    allin = np.stack([(flags >> (4 + g)) & 1 for g in range(3)], 1).astype(bool)
    gath = np.stack([(flags >> (8 + g)) & 1 for g in range(3)], 1).astype(bool)
    farp = np.stack([(codes[:, 9 * g:9 * g + 9] == 0xFFFE).any((1, 2)) for g in range(3)], 1)
    print("%s rcap %d: %d live phases | contiguous all-in %.3f | gathered %.3f (complete %.3f) | phases with far codes %.4f | mean rows staged: contiguous %.0f, gathered %.0f"
          % (r["name"], r["rcap"], live.sum(), (live & allin & ~gath).sum() / live.sum(), (live & gath).sum() / live.sum(),
             (live & gath & allin).sum() / max((live & gath).sum(), 1), (live & farp).sum() / live.sum(),
             cnt[live & ~gath].mean(), cnt[live & gath].mean() if (live & gath).any() else 0))
