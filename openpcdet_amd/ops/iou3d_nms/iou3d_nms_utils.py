"""Python face of the rotated IoU / NMS ops -- same names, arguments and results as the reference's
``pcdet/ops/iou3d_nms/iou3d_nms_utils.py`` (lines cited per function), running on the gfx950 kernels.

Extra (no reference counterpart): ``nms_batched`` runs many (frame, head) problems in one launch and keeps
every result on the device.
"""
import numpy as np
import torch

from ... import _lib
from . import iou3d_nms_cuda


def _as_tensor(x):
    """check_numpy_to_torch of the reference (pcdet/utils/common_utils.py) for the one caller that needs it."""
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x).float(), True
    if isinstance(x, (int, float)):
        return torch.tensor([x]).float(), True
    return x, False


def boxes_bev_iou_cpu(boxes_a, boxes_b):
    """iou3d_nms_utils.py:12-28.  (N,7),(M,7) CPU tensors or numpy -> (N,M) BEV IoU, host code only."""
    boxes_a, is_numpy = _as_tensor(boxes_a)
    boxes_b, is_numpy = _as_tensor(boxes_b)
    assert not (boxes_a.is_cuda or boxes_b.is_cuda), 'Only support CPU tensors'
    assert boxes_a.shape[1] == 7 and boxes_b.shape[1] == 7
    ans_iou = boxes_a.new_zeros(torch.Size((boxes_a.shape[0], boxes_b.shape[0])))
    iou3d_nms_cuda.boxes_iou_bev_cpu(boxes_a.contiguous(), boxes_b.contiguous(), ans_iou)
    return ans_iou.numpy() if is_numpy else ans_iou


def boxes_iou_bev(boxes_a, boxes_b):
    """iou3d_nms_utils.py:31-45.  (N,7),(M,7) GPU -> (N,M) BEV IoU."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    ans_iou = torch.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    iou3d_nms_cuda.boxes_iou_bev_gpu(boxes_a.contiguous(), boxes_b.contiguous(), ans_iou)
    return ans_iou


def _iou3d_from_bev(overlaps_bev, boxes_a, boxes_b, pairwise):
    """Height overlap x BEV overlap -> 3D IoU (iou3d_nms_utils.py:59-79, 95-115, 167-187)."""
    shape_b = (1, -1) if pairwise else (-1, 1)
    a_hi = (boxes_a[:, 2] + boxes_a[:, 5] / 2).view(-1, 1)
    a_lo = (boxes_a[:, 2] - boxes_a[:, 5] / 2).view(-1, 1)
    b_hi = (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(*shape_b)
    b_lo = (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(*shape_b)
    overlaps_h = torch.clamp(torch.min(a_hi, b_hi) - torch.max(a_lo, b_lo), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(*shape_b)
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-6)


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """iou3d_nms_utils.py:48-81.  (N,7),(M,7) -> (N,M) 3D IoU."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    overlaps_bev = torch.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    iou3d_nms_cuda.boxes_overlap_bev_gpu(boxes_a.contiguous(), boxes_b.contiguous(), overlaps_bev)
    return _iou3d_from_bev(overlaps_bev, boxes_a, boxes_b, pairwise=True)


def boxes_aligned_iou3d_gpu(boxes_a, boxes_b):
    """iou3d_nms_utils.py:83-117.  (N,7),(N,7) -> (N,1) 3D IoU of corresponding boxes."""
    assert boxes_a.shape[0] == boxes_b.shape[0]
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    overlaps_bev = torch.zeros((boxes_a.shape[0], 1), dtype=torch.float32, device=boxes_a.device)
    iou3d_nms_cuda.boxes_aligned_overlap_bev_gpu(boxes_a.contiguous(), boxes_b.contiguous(), overlaps_bev)
    return _iou3d_from_bev(overlaps_bev, boxes_a, boxes_b, pairwise=False)


def paired_boxes_iou3d_gpu(boxes_a, boxes_b):
    """iou3d_nms_utils.py:155-189.  (N,7),(N,7) -> (N,) 3D IoU."""
    assert boxes_a.shape[0] == boxes_b.shape[0]
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    overlaps_bev = torch.zeros((boxes_a.shape[0], 1), dtype=torch.float32, device=boxes_a.device)
    iou3d_nms_cuda.paired_boxes_overlap_bev_gpu(boxes_a.contiguous(), boxes_b.contiguous(), overlaps_bev)
    return _iou3d_from_bev(overlaps_bev, boxes_a, boxes_b, pairwise=False).view(-1)


def _nms_device(boxes_sorted, thresh, normal):
    """boxes already in score order -> (keep [N] int64 on device, num_keep [1] int32 on device); no host sync."""
    n = boxes_sorted.shape[0]
    dev = boxes_sorted.device
    keep = torch.empty(n, dtype=torch.int64, device=dev)
    num = torch.zeros(1, dtype=torch.int32, device=dev)
    if n > 0:
        L = _lib.lib()
        with torch.cuda.device(dev):
            ws = _lib.workspace(L.pvx_nms_workspace_bytes(n, 1), dev)
            _lib.check(L.pvx_nms(_lib.ptr(boxes_sorted), n, float(thresh), int(normal), _lib.ptr(keep), _lib.ptr(num),
                                 _lib.ptr(ws), ws.numel(), _lib.stream()), "pvx_nms")
    return keep, num


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """iou3d_nms_utils.py:120-135.  Returns (indices into `boxes` of the kept boxes in score order, None).

    One host read (the kept count) instead of the reference's mask D2H + keep H2D round trip."""
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    boxes = boxes[order].contiguous().float()
    keep, num = _nms_device(boxes, thresh, 0)
    return order[keep[:int(num.item())]].contiguous(), None


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """iou3d_nms_utils.py:138-152 (axis-aligned IoU)."""
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    boxes = boxes[order].contiguous().float()
    keep, num = _nms_device(boxes, thresh, 1)
    return order[keep[:int(num.item())]].contiguous(), None


NMS_SELECT_MAX = 16384      # pvx_nms_select_batched: one problem's sort keys live in one CU's LDS


def nms_select(boxes, scores, counts, thresh, normal=False, score_thresh=None, pre_max=None, post_max=None, out=None, ws=None):
    """class_agnostic_nms (model_nms_utils.py:6-25) for S problems at once, all of it on the device in three launches
    (pvx_nms_select_batched): score threshold, top-pre_max by score (ties: lower index first), NMS, keep[:post_max] mapped
    back to the caller's indices.

    boxes (S, cap, >=7) float32, scores (S, cap) float32, counts (S,) int32 or None; cap <= NMS_SELECT_MAX.
    -> (sel (S, post_max) int64, num (S,) int32): sel[s, :num[s]] index segment s's boxes, best score first; entries
    beyond num[s] repeat the best-scored index.  No host sync.  `out` = (sel, num) and `ws` reuse caller buffers."""
    S, cap = scores.shape
    dev = boxes.device
    assert boxes.dim() == 3 and boxes.shape[:2] == (S, cap) and boxes.shape[2] >= 7
    assert boxes.dtype == torch.float32 and scores.dtype == torch.float32 and cap <= NMS_SELECT_MAX
    boxes, scores = boxes.contiguous(), scores.contiguous()
    post = cap if post_max is None else int(post_max)
    if out is None:
        sel = torch.zeros((S, post), dtype=torch.int64, device=dev)
        num = torch.zeros(S, dtype=torch.int32, device=dev)
    else:
        sel, num = out
        assert sel.shape == (S, post) and sel.dtype == torch.int64 and sel.is_contiguous() and num.dtype == torch.int32
    if S > 0 and cap > 0 and post > 0:
        if counts is not None:
            counts = counts.to(torch.int32).contiguous()
        L = _lib.lib()
        with torch.cuda.device(dev):
            if ws is None:
                ws = _lib.workspace(L.pvx_nms_select_workspace_bytes(cap, S), dev)
            _lib.check(L.pvx_nms_select_batched(_lib.ptr(boxes), boxes.shape[2], _lib.ptr(scores), _lib.ptr(counts), S, cap,
                                                float('-inf') if score_thresh is None else float(score_thresh),
                                                0 if pre_max is None else int(pre_max), float(thresh), int(normal), post,
                                                _lib.ptr(sel), _lib.ptr(num), _lib.ptr(ws), ws.numel(), _lib.stream()),
                       "pvx_nms_select_batched")
    return sel, num


def nms_batched(boxes, scores, counts, thresh, normal=False):
    """Many independent NMS problems in one launch (CenterHead runs 6 heads x B frames,
    center_head.py:332-384).

    boxes (S, cap, 7), scores (S, cap), counts (S,) int32 = valid leading entries per segment (entries at or
    beyond counts[s] are ignored; give them any score).  Returns (sel (S, cap) int64, num (S,) int32) on the
    device: sel[s, :num[s]] index the kept boxes of segment s in descending score order.  No host sync.
    """
    S, cap = scores.shape
    dev = boxes.device
    if cap <= NMS_SELECT_MAX:
        return nms_select(boxes.float(), scores.float(), counts, thresh, normal)
    # larger problems: sort with torch, NMS over the sorted boxes
    idx = torch.arange(cap, device=dev).unsqueeze(0)
    masked = torch.where(idx < counts.view(-1, 1).long(), scores, torch.full_like(scores, float('-inf')))
    order = masked.sort(1, descending=True)[1]
    sorted_boxes = torch.gather(boxes[..., :7].float(), 1, order.unsqueeze(-1).expand(S, cap, 7)).contiguous()
    keep = torch.zeros((S, cap), dtype=torch.int64, device=dev)
    num = torch.zeros(S, dtype=torch.int32, device=dev)
    if S > 0 and cap > 0:
        L = _lib.lib()
        counts32 = counts.to(torch.int32).contiguous()
        with torch.cuda.device(dev):
            ws = _lib.workspace(L.pvx_nms_workspace_bytes(cap, S), dev)
            _lib.check(L.pvx_nms_batched(_lib.ptr(sorted_boxes), _lib.ptr(counts32), S, cap, float(thresh), int(normal),
                                         _lib.ptr(keep), _lib.ptr(num), _lib.ptr(ws), ws.numel(), _lib.stream()),
                       "pvx_nms_batched")
    valid = idx < num.view(-1, 1).long()
    sel = torch.gather(order, 1, torch.where(valid, keep, torch.zeros_like(keep)))
    return sel, num
