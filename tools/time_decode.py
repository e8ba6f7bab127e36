"""times the CenterHead front end (6 heads x B frames: decode + batched NMS) on the GPU, HIP events; prints one line"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from openpcdet_amd import center_head_post
B, H, W, K = 16, 180, 180, 500
classes = [1, 2, 2, 1, 2, 2]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
pred = []
for c in classes:
    pred.append(dict(hm=torch.randn((B, c, H, W), device=dev, generator=g) * 1.5 - 3.0, center=torch.rand((B, 2, H, W), device=dev, generator=g),
                     center_z=torch.rand((B, 1, H, W), device=dev, generator=g) - 1.0, dim=torch.randn((B, 3, H, W), device=dev, generator=g) * 0.3 + 0.5,
                     rot=torch.randn((B, 2, H, W), device=dev, generator=g), vel=torch.randn((B, 2, H, W), device=dev, generator=g)))
mapping, base = [], 0
for c in classes:
    mapping.append(torch.arange(base, base + c, dtype=torch.int64, device=dev)); base += c
cfg = dict(MAX_OBJ_PER_SAMPLE=K, SCORE_THRESH=0.1, POST_CENTER_LIMIT_RANGE=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
           NMS_CONFIG=dict(NMS_TYPE="nms_gpu", NMS_THRESH=0.2, NMS_PRE_MAXSIZE=1000, NMS_POST_MAXSIZE=83))
args = (B, pred, mapping, [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0], [0.075, 0.075, 0.2], 8, cfg)
for _ in range(3):
    out = center_head_post.generate_predicted_boxes(*args)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    out = center_head_post.generate_predicted_boxes(*args)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print("center head post-processing, %d heads x %d frames, K=%d: %.3f ms per batch (wall, incl. the final host read), %d boxes kept in frame 0"
      % (len(classes), B, K, ms, out[0]["pred_boxes"].shape[0]))
