"""Drop-in for the decode half of pcdet/models/model_utils/centernet_utils.py on MI355X.

`decode_bbox_from_heatmap` keeps the reference's signature and return value (centernet_utils.py:186-254) and runs as ONE launch
of `pvx_center_decode` per head instead of two torch.topk calls, five permute + gather launches and a Python loop over the batch
(`_topk` :163-178, `_transpose_and_gather_feat` :156-160).  `circle_nms=True` and the `iou` branch are `assert False` /
unused in the reference's configs and raise NotImplementedError here."""
import ctypes

import torch

from . import _lib


def decode_bbox_from_heatmap(heatmap, rot_cos, rot_sin, center, center_z, dim, point_cloud_range=None, voxel_size=None,
                             feature_map_stride=None, vel=None, iou=None, K=100, circle_nms=False, score_thresh=None,
                             post_center_limit_range=None):
    if circle_nms:
        raise NotImplementedError("circle_nms is 'not checked yet' in the reference (centernet_utils.py:191-194)")
    if iou is not None:
        raise NotImplementedError("the iou-rectified branch (centernet_utils.py:215-216) is not built")
    assert post_center_limit_range is not None
    L = _lib.lib()
    dev = heatmap.device
    B, C, H, W = heatmap.shape
    f32 = lambda t: t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()
    heatmap, center, center_z, dim = f32(heatmap), f32(center), f32(center_z), f32(dim)
    vel = f32(vel) if vel is not None else None
    # rot_cos / rot_sin usually arrive as channel views rot[:, 0:1], rot[:, 1:2] of one [B, 2, H, W] tensor (center_head.py:337-338):
    # pass them as they are when each frame's plane is contiguous, with the batch stride of the parent
    def plane(t):
        t = t if t.dtype == torch.float32 else t.float()
        ok = t.dim() == 4 and t.shape[1] == 1 and t.stride(3) == 1 and t.stride(2) == W and t.stride(0) >= H * W
        return t if ok else t.contiguous()
    rot_cos, rot_sin = plane(rot_cos), plane(rot_sin)
    if rot_cos.stride(0) != rot_sin.stride(0):
        rot_cos, rot_sin = rot_cos.contiguous(), rot_sin.contiguous()
    nbox = 9 if vel is not None else 7
    boxes = torch.empty((B, K, nbox), dtype=torch.float32, device=dev)
    scores = torch.empty((B, K), dtype=torch.float32, device=dev)
    labels = torch.empty((B, K), dtype=torch.int32, device=dev)
    counts = torch.zeros((B,), dtype=torch.int32, device=dev)
    vs = (ctypes.c_float * 2)(float(voxel_size[0]), float(voxel_size[1]))
    rm = (ctypes.c_float * 2)(float(point_cloud_range[0]), float(point_cloud_range[1]))
    lim = (ctypes.c_float * 6)(*[float(v) for v in post_center_limit_range])
    thr = ctypes.byref(ctypes.c_float(float(score_thresh))) if score_thresh is not None else None
    with torch.cuda.device(dev):
        _lib.check(L.pvx_center_decode(_lib.ptr(heatmap), _lib.ptr(center), _lib.ptr(center_z), _lib.ptr(dim),
                                       ctypes.c_void_p(rot_cos.data_ptr()), ctypes.c_void_p(rot_sin.data_ptr()), rot_cos.stride(0), _lib.ptr(vel), B, C, H, W, int(K),
                                       float(feature_map_stride), vs, rm, lim, thr, _lib.ptr(boxes), _lib.ptr(scores),
                                       _lib.ptr(labels), _lib.ptr(counts), _lib.stream()), "pvx_center_decode")
    n = counts.tolist()          # the reference returns variable-length tensors per frame: one host read for the batch
    return [dict(pred_boxes=boxes[b, :n[b]], pred_scores=scores[b, :n[b]], pred_labels=labels[b, :n[b]].long())
            for b in range(B)]


def decode_bbox_from_voxels_nuscenes(batch_size, indices, obj, rot_cos, rot_sin, center, center_z, dim, vel=None, iou=None,
                                     point_cloud_range=None, voxel_size=None, voxels_3d=None, feature_map_stride=None, K=100,
                                     score_thresh=None, post_center_limit_range=None, add_features=None):
    """Same signature and return value as centernet_utils.py:303-367 (VoxelNeXtHead): per frame the K best (row, class)
    scores of `obj` [N, C], decoded from the sparse rows, ONE launch instead of a Python loop of topk / gather per frame.
    Rows of one frame must be contiguous in `indices` (they are: every strided conv of this package emits key order)."""
    if iou is not None or add_features is not None:
        raise NotImplementedError("the iou / add_features branches (centernet_utils.py:317-318,334-336) are not built")
    assert post_center_limit_range is not None
    L = _lib.lib()
    dev = obj.device
    B = int(batch_size)
    N, C = obj.shape
    batch_idx = indices[:, 0].contiguous()
    # frame boundaries without a host read: rows of frame b are [row_begin[b], row_begin[b+1])
    row_begin = torch.searchsorted(batch_idx, torch.arange(B + 1, device=dev, dtype=batch_idx.dtype)).to(torch.int32)
    f32 = lambda t: t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()
    spatial = indices[:, 1:].to(torch.int32).contiguous()
    stride_rows, stride = None, 1.0
    if isinstance(feature_map_stride, torch.Tensor):
        stride_rows = f32(feature_map_stride.view(-1))                      # :320-321, one stride per voxel
    else:
        stride = float(feature_map_stride)
    nbox = 9 if vel is not None else 7
    boxes = torch.empty((B, K, nbox), dtype=torch.float32, device=dev)
    scores = torch.empty((B, K), dtype=torch.float32, device=dev)
    labels = torch.empty((B, K), dtype=torch.int32, device=dev)
    counts = torch.zeros((B,), dtype=torch.int32, device=dev)
    vs = (ctypes.c_float * 2)(float(voxel_size[0]), float(voxel_size[1]))
    rm = (ctypes.c_float * 2)(float(point_cloud_range[0]), float(point_cloud_range[1]))
    lim = (ctypes.c_float * 6)(*[float(v) for v in post_center_limit_range])
    thr = ctypes.byref(ctypes.c_float(float(score_thresh))) if score_thresh is not None else None
    obj, center, center_z, dim, rot_cos, rot_sin = (f32(t) for t in (obj, center, center_z, dim, rot_cos, rot_sin))
    vel = f32(vel) if vel is not None else None
    with torch.cuda.device(dev):
        _lib.check(L.pvx_voxel_decode(_lib.ptr(obj), _lib.ptr(row_begin), _lib.ptr(spatial), spatial.shape[1], _lib.ptr(center),
                                      _lib.ptr(center_z), _lib.ptr(dim), _lib.ptr(rot_cos), _lib.ptr(rot_sin), _lib.ptr(vel),
                                      _lib.ptr(stride_rows), B, C, int(K), stride, vs, rm, lim, thr, _lib.ptr(boxes),
                                      _lib.ptr(scores), _lib.ptr(labels), _lib.ptr(counts), _lib.stream()), "pvx_voxel_decode")
    n = counts.tolist()
    return [dict(pred_boxes=boxes[b, :n[b]], pred_scores=scores[b, :n[b]], pred_labels=labels[b, :n[b]].long(),
                 pred_ious=None, add_features=None) for b in range(B)]
