"""How many rulebook entries of the 128-channel level fall outside a centred window of R rows per (256-site tile, dz group): per phase
the fraction with any far entry, far entries and distinct far rows per phase, and per 64-site wave quarter.  PVX_ENGINE_FAR_ONLY=0."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["PVX_ENGINE_FAR_ONLY"] = "0"
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
    if r["kvol"] != 27 or nbr.data_ptr() in seen or not r["subm"] or r["cin"] < 64:
        continue
    seen.add(nbr.data_ptr())
    n = min(int(r["level_out"].n_dev.item()), r["level_out"].cap)
    TM = 256
    nt = n // TM
    v = nbr[:, :nt * TM].view(3, 9, nt, TM).permute(0, 2, 1, 3).contiguous().long()      # [g, tile, kk, site]
    valid = v >= 0
    big = torch.where(valid, v, torch.full_like(v, 1 << 40)).view(3, nt, -1)
    small = torch.where(valid, v, torch.full_like(v, -1)).view(3, nt, -1)
    lo, hi = big.min(-1).values, small.max(-1).values
    span = torch.where(hi >= 0, hi - lo + 1, torch.zeros_like(hi))
    print(r["name"], "rows", n, "tiles", nt, "pairs/site %.2f" % (valid.sum().item() / n))
    for R in (320, 359, 423, 487, 640, 1024):
        wlo = torch.where(span <= R, lo, lo + (span - R) // 2)[:, :, None, None]
        far = valid & ((v < wlo) | (v >= wlo + R))
        per_phase = far.view(3, nt, -1).sum(-1).float()
        any_phase = (per_phase > 0).float().mean().item()
        # per wave quarter (64 sites) and offset: any far lane
        fw = far.view(3, nt, 9, 4, 64).any(-1).float()       # [g, tile, kk, wave]
        wave_phase_any = fw.any(2).float().mean().item() if False else (fw.sum(2) > 0).float().mean().item()
        kk_per_wave_phase = fw.sum(2).mean().item()
        q = torch.quantile(per_phase.flatten(), torch.tensor([0.5, 0.9, 0.99, 1.0], device=dev)).tolist()
        print("  R=%4d: phases with far %.3f  wave-phases with far %.3f  far (offset, wave) per wave-phase %.2f  far entries per phase mean %.1f p50/90/99/max %s  far share of pairs %.4f" % (
            R, any_phase, wave_phase_any, kk_per_wave_phase, per_phase.mean().item(), [int(x) for x in q], far.sum().item() / max(valid.sum().item(), 1)))
