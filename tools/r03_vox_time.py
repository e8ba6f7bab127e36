"""per-kernel device time of the batch voxelizer alone: rocprofv3 --kernel-trace --stats -- python3 tools/r03_vox_time.py   (or plain: event time)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from openpcdet_amd import engine, synth
dev = torch.device("cuda", 0)
cfg = synth.CONFIGS[bench.WORKLOAD]
B = int(os.environ.get("B", "16"))
net = bench.build_model(cfg, dev)
eng = engine.HotPathEngine(net, B, cfg["range"], cfg["voxel"], cfg["T"], cfg["max_voxels"], cfg["C"], cfg["n_points"], dtype=torch.bfloat16, device=dev,
                           level_capacity_factor=2.0)
for b in range(B):
    eng.points[b].copy_(torch.from_numpy(synth.config_cloud(bench.WORKLOAD, 3 + b)).to(dev))
for _ in range(3):
    eng.voxelize()
torch.cuda.synchronize()
a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    eng.voxelize()
b_.record()
torch.cuda.synchronize()
print("voxelize B=%d: %.1f us per call, %d voxels" % (B, a.elapsed_time(b_) * 100, int(eng.levels[0].n_dev.item())))
