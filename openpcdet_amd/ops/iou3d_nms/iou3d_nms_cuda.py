"""Stand-in for the reference's pybind extension module ``pcdet.ops.iou3d_nms.iou3d_nms_cuda``
(pcdet/ops/iou3d_nms/src/iou3d_nms_api.cpp:11-20): the same eight functions with the same tensor-level
signatures and return values, forwarding to the C ABI.  The name keeps ``_cuda`` only because callers import it
under that name (iou3d_nms_utils.py:9, hungarian_assigner.py:3); there is no CUDA anywhere in it.

Importing this module does not touch the GPU (DataLoader workers import it for the CPU function,
database_sampler.py:489-493).
"""
import torch

from ... import _lib


def _chk_dev(*ts):
    for t in ts:
        if not t.is_cuda:
            raise ValueError("tensor must be on the GPU")
        if not t.is_contiguous():
            raise ValueError("tensor must be contiguous")
        if t.dtype != torch.float32:
            raise ValueError("tensor must be float32")


def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    """iou3d_nms.cpp:73-92.  ans_overlap (N, M) is written in place; returns 1."""
    _chk_dev(boxes_a, boxes_b, ans_overlap)
    L = _lib.lib()
    with torch.cuda.device(boxes_a.device):
        _lib.check(L.pvx_boxes_overlap_bev(_lib.ptr(boxes_a), boxes_a.shape[0], _lib.ptr(boxes_b), boxes_b.shape[0],
                                           _lib.ptr(ans_overlap), _lib.stream()), "pvx_boxes_overlap_bev")
    return 1


def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
    """iou3d_nms.cpp:117-135."""
    _chk_dev(boxes_a, boxes_b, ans_iou)
    L = _lib.lib()
    with torch.cuda.device(boxes_a.device):
        _lib.check(L.pvx_boxes_iou_bev(_lib.ptr(boxes_a), boxes_a.shape[0], _lib.ptr(boxes_b), boxes_b.shape[0],
                                       _lib.ptr(ans_iou), _lib.stream()), "pvx_boxes_iou_bev")
    return 1


def paired_boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    """iou3d_nms.cpp:94-115."""
    _chk_dev(boxes_a, boxes_b, ans_overlap)
    assert boxes_a.shape[0] == boxes_b.shape[0]
    L = _lib.lib()
    with torch.cuda.device(boxes_a.device):
        _lib.check(L.pvx_boxes_overlap_bev_paired(_lib.ptr(boxes_a), _lib.ptr(boxes_b), boxes_a.shape[0],
                                                  _lib.ptr(ans_overlap), _lib.stream()), "pvx_boxes_overlap_bev_paired")
    return 1


def boxes_aligned_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    """iou3d_nms.cpp:50-71 (same computation as the paired form)."""
    return paired_boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap)


def _nms(boxes, keep, thresh, normal):
    _chk_dev(boxes)
    if not keep.is_contiguous() or keep.dtype != torch.int64:
        raise ValueError("keep must be a contiguous int64 tensor")
    n = boxes.shape[0]
    if n == 0:
        return 0
    L = _lib.lib()
    dev = boxes.device
    with torch.cuda.device(dev):
        keep_dev = keep if keep.is_cuda else torch.empty(n, dtype=torch.int64, device=dev)
        num = torch.empty(1, dtype=torch.int32, device=dev)
        ws = _lib.workspace(L.pvx_nms_workspace_bytes(n, 1), dev)
        _lib.check(L.pvx_nms(_lib.ptr(boxes), n, float(thresh), normal, _lib.ptr(keep_dev), _lib.ptr(num),
                             _lib.ptr(ws), ws.numel(), _lib.stream()), "pvx_nms")
        num_out = int(num.item())
        if not keep.is_cuda:   # the reference's signature hands in a CPU LongTensor (iou3d_nms_utils.py:133)
            keep[:num_out] = keep_dev[:num_out].cpu()
    return num_out


def nms_gpu(boxes, keep, nms_overlap_thresh):
    """iou3d_nms.cpp:137-183: boxes (N,7) sorted by score on the GPU, keep (N) int64 (CPU or GPU); returns count."""
    return _nms(boxes, keep, nms_overlap_thresh, 0)


def nms_normal_gpu(boxes, keep, nms_overlap_thresh):
    """iou3d_nms.cpp:186-233."""
    return _nms(boxes, keep, nms_overlap_thresh, 1)


def _chk_cpu(*ts):
    for t in ts:
        if t.is_cuda or not t.is_contiguous() or t.dtype != torch.float32:
            raise ValueError("expected contiguous float32 CPU tensors")


def boxes_iou_bev_cpu(boxes_a, boxes_b, ans_iou):
    """iou3d_cpu.cpp:232-252, host code on host tensors."""
    _chk_cpu(boxes_a, boxes_b, ans_iou)
    L = _lib.lib()
    _lib.check(L.pvx_boxes_iou_bev_cpu(_lib.ptr(boxes_a), boxes_a.shape[0], _lib.ptr(boxes_b), boxes_b.shape[0],
                                       _lib.ptr(ans_iou)), "pvx_boxes_iou_bev_cpu")
    return 1


def boxes_aligned_iou_bev_cpu(boxes_a, boxes_b, ans_iou):
    """iou3d_cpu.cpp:254-273."""
    _chk_cpu(boxes_a, boxes_b, ans_iou)
    assert boxes_a.shape[0] == boxes_b.shape[0]
    L = _lib.lib()
    _lib.check(L.pvx_boxes_aligned_iou_bev_cpu(_lib.ptr(boxes_a), _lib.ptr(boxes_b), boxes_a.shape[0],
                                               _lib.ptr(ans_iou)), "pvx_boxes_aligned_iou_bev_cpu")
    return 1
