"""Score filtering + top-k + NMS + index mapping around the NMS op -- same three entry points, arguments and return
values as pcdet/models/model_utils/model_nms_utils.py:6-110 (callers: CenterHead.generate_predicted_boxes
center_head.py:362-374, VoxelNeXtHead voxelnext_head.py:461-465, Detector3DTemplate.post_processing
detector3d_template.py:256-281).  The NMS itself is openpcdet_amd.ops.iou3d_nms (pvx_nms on the device).

`nms_config` is what pcdet passes: a mapping with attribute access (EasyDict) holding NMS_TYPE, NMS_THRESH,
NMS_PRE_MAXSIZE, NMS_POST_MAXSIZE; a plain dict works too.
"""
import torch

from .ops.iou3d_nms import iou3d_nms_utils


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def _cfg(nms_config):
    return nms_config if hasattr(nms_config, 'NMS_THRESH') else _Cfg(nms_config)


def _topk_nms(scores, boxes, cfg):
    """indices (into scores/boxes) kept by NMS over the NMS_PRE_MAXSIZE best boxes, best first, at most NMS_POST_MAXSIZE"""
    if scores.shape[0] == 0:
        return []
    top_scores, top = torch.topk(scores, k=min(cfg.NMS_PRE_MAXSIZE, scores.shape[0]))
    nms = getattr(iou3d_nms_utils, cfg.NMS_TYPE)
    keep, _ = nms(boxes[top][:, 0:7], top_scores, cfg.NMS_THRESH, **cfg)
    return top[keep[:cfg.NMS_POST_MAXSIZE]]


_FUSED = {"nms_gpu": 0, "nms_normal_gpu": 1}


def class_agnostic_nms(box_scores, box_preds, nms_config, score_thresh=None):
    """-> (selected indices into the ORIGINAL box list, their scores)   (model_nms_utils.py:6-25)"""
    cfg = _cfg(nms_config)
    all_scores = box_scores
    n = box_scores.shape[0]
    if (box_scores.is_cuda and 0 < n <= iou3d_nms_utils.NMS_SELECT_MAX and cfg.NMS_TYPE in _FUSED
            and box_scores.dtype == torch.float32 and box_preds.dtype == torch.float32):
        # the whole function in three launches on the device; one host read (the kept count) shapes the result
        sel, num = iou3d_nms_utils.nms_select(box_preds.unsqueeze(0), box_scores.unsqueeze(0), None, cfg.NMS_THRESH,
                                              normal=_FUSED[cfg.NMS_TYPE], score_thresh=score_thresh,
                                              pre_max=cfg.NMS_PRE_MAXSIZE, post_max=min(cfg.NMS_POST_MAXSIZE, n))
        selected = sel[0, :int(num.item())]
        return selected, all_scores[selected]
    if score_thresh is not None:
        passed = box_scores >= score_thresh
        box_scores, box_preds = box_scores[passed], box_preds[passed]
    selected = _topk_nms(box_scores, box_preds, cfg)
    if score_thresh is not None:
        selected = passed.nonzero().view(-1)[selected]
    return selected, all_scores[selected]


def multi_classes_nms(cls_scores, box_preds, nms_config, score_thresh=None):
    """cls_scores (N, num_class), box_preds (N, 7+C) -> (scores, labels, boxes), one NMS per class column
    (model_nms_utils.py:28-69)"""
    cfg = _cfg(nms_config)
    out_scores, out_labels, out_boxes = [], [], []
    for k in range(cls_scores.shape[1]):
        if score_thresh is not None:
            passed = cls_scores[:, k] >= score_thresh
            scores_k, boxes_k = cls_scores[passed, k], box_preds[passed]
        else:
            scores_k, boxes_k = cls_scores[:, k], box_preds
        selected = _topk_nms(scores_k, boxes_k, cfg)
        out_scores.append(scores_k[selected])
        out_labels.append(scores_k.new_ones(len(selected)).long() * k)
        out_boxes.append(boxes_k[selected])
    return torch.cat(out_scores, dim=0), torch.cat(out_labels, dim=0), torch.cat(out_boxes, dim=0)


def class_specific_nms(box_scores, box_preds, box_labels, nms_config, score_thresh=None):
    """per-class thresholds / sizes given as lists in nms_config; rotated NMS always (model_nms_utils.py:72-110)"""
    cfg = _cfg(nms_config)
    selected = []
    for k in range(len(cfg.NMS_THRESH)):
        of_class = box_labels == k
        if score_thresh is not None and isinstance(score_thresh, float):
            of_class = of_class & (box_scores > score_thresh)
        elif score_thresh is not None and isinstance(score_thresh, list):
            of_class = of_class & (box_scores > score_thresh[k])
        rows = torch.nonzero(of_class)[:, 0]
        if rows.shape[0] > 0:
            keep, _ = iou3d_nms_utils.nms_gpu(box_preds[of_class], box_scores[of_class], thresh=cfg.NMS_THRESH[k],
                                              pre_maxsize=cfg.NMS_PRE_MAXSIZE[k], post_max_size=cfg.NMS_POST_MAXSIZE[k])
            selected.append(rows[keep])
    if len(selected) != 0:
        selected = torch.cat(selected)
    return selected, box_scores[selected]
