#!/usr/bin/env python3
"""Where a phase of the super-tile conv kernel (csrc/sparse_conv_st.hip) spends its cycles: runs the nuScenes plan once on a
diagnostic build of the library (-DPVX_ST_STAMPS, loaded through PVX_LIB) and prints per-segment cycle sums per phase.

    PVX_LIB=.../libpcdet_stamps.so python tools/st_stamps.py"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from openpcdet_amd import _lib, engine, synth  # noqa: E402

SEG = ["barrier1", "wait+lds_write", "barrier2", "find+request", "codes+prologue", "mfma_loop", "epilogue", "phases"]
# wave-specialised kernel (PVX_CONV_ST=1): waves 0-3 fill segments 0-3, waves 4-7 segments 4-7
SEG_WS = ["M: wait at B1", "M: B1..B2 (+epilogue)", "M: codes -> slots", "M: fast-path loop", "L: wait at B1", "L: vm wait + LDS write",
          "L: B2", "L: find + request issue"]
SEG_M = ["M: wait at B1", "M: B1..B2 (+epilogue)", "M: codes -> slots (fast)", "M: fast-path loop", "M: # fast phases", "M: # slow phases",
         "M: tails (loop end .. B1)", "M: slow-path loop"]


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg = synth.CONFIGS[bench.WORKLOAD]
    B = 16
    net = bench.build_model(cfg, dev)
    eng = engine.HotPathEngine(net, B, cfg["range"], cfg["voxel"], cfg["T"], cfg["max_voxels"], cfg["C"], cfg["n_points"],
                               dtype=torch.bfloat16, device=dev, level_capacity_factor=2.0)
    for b in range(B):
        eng.points[b].copy_(torch.from_numpy(synth.config_cloud(bench.WORKLOAD, 3 + b)).to(dev))
    only = os.environ.get("ST_ONLY", "64x64")
    for p in range(2):
        eng.voxelize()

        def hook(name, fn):
            fn()
            if only in name and "conv" in name:
                torch.cuda.synchronize()
                raw = ctypes.CDLL(_lib.LIB_PATH)
                buf = (ctypes.c_ulonglong * (256 * 8 * 8))()
                rc = raw.pvx_debug_st_stamps(buf, 256 * 8 * 8)
                assert rc == 0
                a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 8).astype(np.float64)
                if p == 1 and os.environ.get("PVX_CONV_ST", "1") != "2":
                    print(name)
                    for q in range(8):
                        w = a[:, :4, q]
                        print("   %-26s %9.0f per wave   (min wg %9.0f  max wg %9.0f)" % (SEG_M[q], w.mean(), w.mean(axis=1).min(), w.mean(axis=1).max()))
                    for q in range(4, 8):
                        w = a[:, 4:, q]
                        print("   %-26s %9.0f cyc per wave   (min wg %9.0f  max wg %9.0f)" % (SEG_WS[q], w.mean(), w.mean(axis=1).min(), w.mean(axis=1).max()))
                    print("   total M %.0f   L %.0f" % (a[:, :4, :4].sum(axis=2).mean(), a[:, 4:, 4:].sum(axis=2).mean()))
                    # per-workgroup run time (wave 0, first stamp to last): the launch ends with the slowest one
                    tot = a[:, 0, :4].sum(axis=1) + a[:, 0, 6] + a[:, 0, 7]
                    tot = tot[tot > 0]
                    print("   per-workgroup total (wave 0): min %.0f  p10 %.0f  mean %.0f  p90 %.0f  max %.0f   -> idle at the tail %.1f %% of the launch"
                          % (tot.min(), np.percentile(tot, 10), tot.mean(), np.percentile(tot, 90), tot.max(), 100 * (1 - tot.mean() / tot.max())))
                    xc = tot[:256].reshape(-1, 8)      # workgroup b runs on XCD b % 8
                    print("   per-XCD mean: " + " ".join("%.0f" % v for v in xc.mean(axis=0)))
                    if "128x128" in name:      # three roles: weight waves 4-5 (4 barrier waits, 5 write + request, 6 chunk barrier waits, 7 tails), window waves 6-7
                        for nm, sl in (("weight waves", slice(4, 6)), ("window waves", slice(6, 8))):
                            print("   %s: " % nm + "  ".join("seg%d %.0f" % (q, a[:, sl, q].mean()) for q in range(4, 8)))
                elif p == 1:
                    ph = a[:, :, 7].mean()
                    print("%s: phases per wave %.1f" % (name, ph))
                    tot = a[:, :, :7].sum(axis=2).mean()
                    for q in range(7):
                        per = a[:, :, q].mean()
                        print("   %-16s %9.0f cyc per wave  (%5.1f %%)  %7.0f per phase   wave0 %9.0f wave7 %9.0f" % (
                            SEG[q], per, 100 * per / tot, per / ph, a[:, 0, q].mean(), a[:, 7, q].mean()))
                    print("   total %.0f cyc per wave = %.1f us at 100 MHz ticks?" % (tot, tot / 100.0))
        eng.backbone(hook=hook)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
