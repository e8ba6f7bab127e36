"""CenterHead post-processing on MI355X: decode + NMS of every (head, frame) pair without the reference's Python loops.

`generate_predicted_boxes` returns what CenterHead.generate_predicted_boxes returns (center_head.py:322-391): one dict per frame
with pred_boxes [n, 7|9], pred_scores [n], pred_labels [n] (1-based class ids).  The reference runs, per head, two torch.topk +
five gathers (centernet_utils.py:186-254) and then, per head AND per frame, a topk + an NMS launch + a host read
(model_nms_utils.py:6-25: 6 heads x B frames round trips).  Here: one `pvx_center_decode` launch per head, ONE `pvx_nms_batched`
launch for all heads x frames, one host read at the end for the variable-length outputs.  NMS_TYPE nms_gpu / class-agnostic only
(what the CenterPoint and VoxelNeXt nuScenes configs use); other types go through openpcdet_amd.model_nms_utils per head."""
import ctypes

import torch

from . import _lib, centernet_utils


def _get(cfg, name, default=None):
    if isinstance(cfg, dict):
        return cfg.get(name, default)
    return getattr(cfg, name, default)


def generate_predicted_boxes(batch_size, pred_dicts, class_id_mapping_each_head, point_cloud_range, voxel_size,
                             feature_map_stride, post_process_cfg, has_vel=True):
    """pred_dicts: one dict per head with the raw head outputs 'hm', 'center', 'center_z', 'dim', 'rot'[, 'vel'] ([B, c, H, W]);
    class_id_mapping_each_head: list of int64 tensors (center_head.py:60-67); post_process_cfg: POST_PROCESSING of the
    dense-head config (MAX_OBJ_PER_SAMPLE, SCORE_THRESH, POST_CENTER_LIMIT_RANGE, NMS_CONFIG)."""
    nms_cfg = _get(post_process_cfg, "NMS_CONFIG")
    if _get(nms_cfg, "NMS_TYPE") != "nms_gpu":
        raise NotImplementedError("batched post-processing is built for NMS_TYPE nms_gpu (class-agnostic); got %r" % _get(nms_cfg, "NMS_TYPE"))
    K = int(_get(post_process_cfg, "MAX_OBJ_PER_SAMPLE"))
    thr = _get(post_process_cfg, "SCORE_THRESH")
    lim = list(_get(post_process_cfg, "POST_CENTER_LIMIT_RANGE"))
    pre_max, post_max = int(_get(nms_cfg, "NMS_PRE_MAXSIZE")), int(_get(nms_cfg, "NMS_POST_MAXSIZE"))
    heads = len(pred_dicts)
    dev = pred_dicts[0]["hm"].device
    L = _lib.lib()
    B = int(batch_size)
    S = heads * B
    nbox = 9 if has_vel else 7
    boxes = torch.empty((heads, B, K, nbox), dtype=torch.float32, device=dev)
    scores = torch.empty((heads, B, K), dtype=torch.float32, device=dev)
    labels = torch.empty((heads, B, K), dtype=torch.int32, device=dev)
    counts = torch.zeros((heads, B), dtype=torch.int32, device=dev)
    vs = (ctypes.c_float * 2)(float(voxel_size[0]), float(voxel_size[1]))
    rm = (ctypes.c_float * 2)(float(point_cloud_range[0]), float(point_cloud_range[1]))
    limc = (ctypes.c_float * 6)(*[float(v) for v in lim])
    thrc = ctypes.byref(ctypes.c_float(float(thr))) if thr is not None else None
    f32 = lambda t: t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()
    with torch.cuda.device(dev):
        st = _lib.stream()
        for h, pd in enumerate(pred_dicts):
            hm = f32(pd["hm"]).sigmoid()                                    # center_head.py:333
            dim = f32(pd["dim"]).exp()                                      # :336
            rot = f32(pd["rot"])                                            # channel 0 = cos, 1 = sin (:337-338)
            center, center_z = f32(pd["center"]), f32(pd["center_z"])
            vel = f32(pd["vel"]) if has_vel else None
            _, C, H, W = hm.shape
            _lib.check(L.pvx_center_decode(_lib.ptr(hm), _lib.ptr(center), _lib.ptr(center_z), _lib.ptr(dim),
                                           ctypes.c_void_p(rot.data_ptr()), ctypes.c_void_p(rot.data_ptr() + 4 * H * W), 2 * H * W,
                                           _lib.ptr(vel), B, C, H, W, K, float(feature_map_stride), vs, rm, limc, thrc,
                                           _lib.ptr(boxes[h]), _lib.ptr(scores[h]), _lib.ptr(labels[h]), _lib.ptr(counts[h]), st),
                       "pvx_center_decode")
        # the decode leaves every segment in descending score order, which is what class_agnostic_nms's topk would produce
        # (model_nms_utils.py:12-14); NMS_PRE_MAXSIZE only truncates
        n_in = torch.clamp(counts.view(S), max=pre_max)
        boxes7 = boxes.view(S, K, nbox)[..., :7].contiguous()
        keep = torch.empty((S, K), dtype=torch.int64, device=dev)
        num = torch.empty((S,), dtype=torch.int32, device=dev)
        ws = _lib.workspace(L.pvx_nms_workspace_bytes(K, S), dev)
        _lib.check(L.pvx_nms_batched(_lib.ptr(boxes7), _lib.ptr(n_in), S, K, ctypes.c_float(float(_get(nms_cfg, "NMS_THRESH"))), 0,
                                     _lib.ptr(keep), _lib.ptr(num), _lib.ptr(ws), ws.numel(), st), "pvx_nms_batched")
    kept = torch.clamp(num, max=post_max).view(heads, B).tolist()            # the one host read
    keep = keep.view(heads, B, K)
    ret = []
    for b in range(B):
        bb, ss, ll = [], [], []
        for h in range(heads):
            sel = keep[h, b, :kept[h][b]]
            bb.append(boxes[h, b][sel])
            ss.append(scores[h, b][sel])
            ll.append(class_id_mapping_each_head[h].to(dev)[labels[h, b][sel].long()])    # center_head.py:355
        ret.append(dict(pred_boxes=torch.cat(bb, 0), pred_scores=torch.cat(ss, 0), pred_labels=torch.cat(ll, 0) + 1))   # :387-389
    return ret
