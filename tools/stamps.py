"""diagnostic: per-phase cycle sums of the sparse-conv tile loop (PVX_ABLATE=32), one conv layer at a time"""
import ctypes, os, sys
os.environ["PVX_ABLATE"] = os.environ.get("PVX_ABLATE", "32")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpcdet_amd import _lib, synth
from oracle import loader
L = _lib.lib()
L.pvx_debug_conv_stamps.restype = ctypes.c_int
L.pvx_debug_conv_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda:0")
from openpcdet_amd.spconv import pytorch as spconv
cfg = synth.CONFIGS["nuscenes_centerpoint"]
# level-2-like site set: voxelize 4 frames at 2x voxel size to get ~realistic density
import bench
net = bench.build_model(cfg, dev)
from openpcdet_amd import engine
eng = engine.HotPathEngine(net, 4, cfg["range"], cfg["voxel"], cfg["T"], cfg["max_voxels"], cfg["C"], cfg["n_points"],
                           dtype=torch.bfloat16, device=dev, level_capacity_factor=2.0)
frames = [torch.from_numpy(synth.config_cloud("nuscenes_centerpoint", 3 + i)).to(dev) for i in range(4)]
eng.load_points(frames)
eng.voxelize()
out = np.zeros(1024 * 8, np.uint64)
def hook(name, fn):
    fn()
    torch.cuda.synchronize()
    if name.startswith("conv") and not name.startswith("conv_out"):
        L.pvx_debug_conv_stamps(out.ctypes.data, out.size)
        a = out.reshape(1024, 8)[:, :5].astype(np.float64)
        live = a[:, 4] > 0
        m = a[live]
        print("%-22s WGs=%4d tiles/WG=%.1f  per-tile cycles: wait_prev=%.0f stage=%.0f main=%.0f epi=%.0f  | per-WG total=%.0f" % (
            name, live.sum(), m[:, 4].mean(), (m[:, 0] / m[:, 4]).mean(), (m[:, 1] / m[:, 4]).mean(), (m[:, 2] / m[:, 4]).mean(),
            (m[:, 3] / m[:, 4]).mean(), m[:, :4].sum(1).mean()))
eng.backbone(hook=hook)
