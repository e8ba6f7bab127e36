"""How evenly a conv launch's workgroups finish (diagnostic build with -DPVX_FIN_STAMPS, loaded through PVX_LIB): per layer of the
streaming (128-channel) and rolling (32-channel) kernels, first / last tick of every workgroup -> run-time distribution and the share of
the launch the average workgroup idles at the tail.

    bash tools/build_fin.sh && PVX_LIB=.../libpcdet_fin.so python tools/probes/fin_spread.py [batch]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from openpcdet_amd import _lib, engine, synth

dev = torch.device("cuda", 0)
cfg = synth.CONFIGS[bench.WORKLOAD]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
net = bench.build_model(cfg, dev)
eng = engine.HotPathEngine(net, B, cfg["range"], cfg["voxel"], cfg["T"], cfg["max_voxels"], cfg["C"], cfg["n_points"], dtype=torch.bfloat16, device=dev,
                           level_capacity_factor=2.0)
for b in range(B):
    eng.points[b].copy_(torch.from_numpy(synth.config_cloud(bench.WORKLOAD, 3 + b)).to(dev))
raw = ctypes.CDLL(_lib.LIB_PATH)
for p in range(2):
    eng.voxelize()

    def hook(name, fn):
        fn()
        if p == 1 and name.startswith("conv") and ("32x32" in name or "128x128_k27" in name or "64x128" in name):
            torch.cuda.synchronize()
            buf = (ctypes.c_ulonglong * (2 * 64 * 1024 * 2))()
            ctr = (ctypes.c_uint * 2)()
            assert raw.pvx_debug_fin(buf, ctr) == 0
            fam = 1 if "32x32" in name else 0
            a = np.frombuffer(buf, dtype=np.uint64).reshape(2, 64, 1024, 2).astype(np.float64)[fam]
            a = a[np.argmax(a[:, :, 0].max(axis=1))]          # the ring slot of the latest launch
            a = a[a[:, 1] > 0]
            t0 = a[:, 0].min()
            run = (a[:, 1] - t0) / 100.0          # us
            print("%-22s %4d workgroups: finish min %.0f  p10 %.0f  mean %.0f  p90 %.0f  max %.0f us   idle at the tail %.1f %%   start spread %.1f us"
                  % (name, len(a), run.min(), np.percentile(run, 10), run.mean(), np.percentile(run, 90), run.max(), 100 * (1 - run.mean() / run.max()),
                     (a[:, 0].max() - t0) / 100.0))
            xc = run[:len(run) // 8 * 8].reshape(-1, 8)
            print("      per-XCD mean finish: " + " ".join("%.0f" % v for v in xc.mean(axis=0)))
    eng.backbone(hook=hook)
torch.cuda.synchronize()
