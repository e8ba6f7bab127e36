"""The sparse part of VoxelNeXtHead on this package's spconv-compatible modules: the per-branch conv stack and the class-wise
rectified NMS of the IoU-branch variant.  Same constructor arguments, module-tree names (a reference checkpoint's
`heads_list.<i>.<branch>.<j>...` keys load key for key) and return values as

    SeparateHead                      pcdet/models/dense_heads/voxelnext_head.py:13-47
    rotate_class_specific_nms_iou     pcdet/models/dense_heads/voxelnext_head.py:308-331

The rest of VoxelNeXtHead (target assignment, losses, the dense-tensor decode) is outside the hot path.
"""
import torch
from torch import nn
from torch.nn.init import kaiming_normal_

from . import model_nms_utils
from .spconv import pytorch as spconv


class SeparateHead(nn.Module):
    """One branch per entry of `sep_head_dict` ({name: {'out_channels': c, 'num_conv': n}}): (n - 1) x [SubMConv2d(C -> C, k) + BatchNorm1d
    + ReLU] as SparseSequentials, then SubMConv2d(C -> c, 1, bias), all inside a PLAIN nn.Sequential -- the SparseConvTensor passes
    through it module by module -- and `.features` of the result is the branch's output (voxelnext_head.py:22-30,45).

    With kernel_size 1 (the nuScenes / Argoverse2 configs) every conv is a row-wise linear map and no rulebook is built; with
    kernel_size 3 the first branch builds the 3 x 3 submanifold rulebook under its indice_key and, as in spconv, each branch keeps its
    own (the keys differ per branch)."""

    def __init__(self, input_channels, sep_head_dict, kernel_size, init_bias=-2.19, use_bias=False):
        super().__init__()
        self.sep_head_dict = sep_head_dict
        for name, spec in sep_head_dict.items():
            layers = [spconv.SparseSequential(
                spconv.SubMConv2d(input_channels, input_channels, kernel_size, padding=int(kernel_size // 2), bias=use_bias,
                                  indice_key=name),
                nn.BatchNorm1d(input_channels),
                nn.ReLU()) for _ in range(spec['num_conv'] - 1)]
            layers.append(spconv.SubMConv2d(input_channels, spec['out_channels'], 1, bias=True, indice_key=name + 'out'))
            fc = nn.Sequential(*layers)
            if 'hm' in name:
                fc[-1].bias.data.fill_(init_bias)                  # voxelnext_head.py:31-32
            else:
                for m in fc.modules():                             # voxelnext_head.py:34-38
                    if isinstance(m, spconv.SubMConv2d):
                        kaiming_normal_(m.weight.data)
                        if m.bias is not None:
                            nn.init.constant_(m.bias, 0)
            setattr(self, name, fc)

    def forward(self, x):
        return {name: getattr(self, name)(x).features for name in self.sep_head_dict}


def rotate_class_specific_nms_iou(boxes, scores, iou_preds, labels, rectifier, nms_configs, num_class=None):
    """voxelnext_head.py:308-331 (a method there; `num_class` = self.num_class, default len(rectifier)).

    boxes (N, 7+), scores (N,), iou_preds (N, 1) or (N,), labels (N,) in [0, num_class); per class c the score becomes
    score^(1 - rectifier[c]) * iou^rectifier[c], then class_agnostic_nms under nms_configs[c] (its own threshold and pre / post sizes).
    -> (boxes, rectified scores, labels) of the kept detections, class by class, best score first within a class."""
    assert isinstance(rectifier, list)
    if num_class is None:
        num_class = len(rectifier)
    out_boxes, out_scores, out_labels = [], [], []
    iou = iou_preds.reshape(iou_preds.shape[0], -1)[:, 0] if iou_preds.dim() > 1 else iou_preds
    for c in range(num_class):
        of_class = labels == c
        boxes_c, labels_c = boxes[of_class], labels[of_class]
        scores_c = torch.pow(scores[of_class], 1 - rectifier[c]) * torch.pow(iou[of_class], rectifier[c])
        selected, _ = model_nms_utils.class_agnostic_nms(box_scores=scores_c, box_preds=boxes_c, nms_config=nms_configs[c],
                                                         score_thresh=None)
        if isinstance(selected, list):          # an empty class
            selected = torch.zeros(0, dtype=torch.long, device=boxes.device)
        out_boxes.append(boxes_c[selected])
        out_scores.append(scores_c[selected])
        out_labels.append(labels_c[selected])
    return torch.cat(out_boxes, dim=0), torch.cat(out_scores, dim=0), torch.cat(out_labels, dim=0)
