"""CPU oracle (test infrastructure only) for the CenterPoint head decode, numpy float32.

Restates pcdet/models/model_utils/centernet_utils.py:163-254 (`_topk` :163-178, `_transpose_and_gather_feat` :156-160,
`decode_bbox_from_heatmap` :186-254) as called from pcdet/models/dense_heads/center_head.py:343-352.  The reference selects
with two torch.topk calls (K per class, then K over the C*K survivors); the set is the K largest heatmap values of the frame
either way, in descending order.  torch.topk leaves the order of EQUAL values unspecified; this oracle (and the HIP kernel)
fix it: lower flat index (class-major, then y, then x) first.  Pinned in tests/test_oracle_center_decode.py against the
reference's own tensor sequence executed with CPU torch ops on tie-free inputs (the reference file itself needs numba,
which is absent here: centernet_utils.py:6)."""
import numpy as np

F = np.float32


def decode_bbox_from_heatmap(heatmap, rot_cos, rot_sin, center, center_z, dim, point_cloud_range, voxel_size,
                             feature_map_stride, vel=None, K=100, score_thresh=None, post_center_limit_range=None):
    """-> list (one per frame) of dict(pred_boxes [n, 7|9] f32, pred_scores [n] f32, pred_labels [n] int64)"""
    heatmap = np.asarray(heatmap, F)
    B, C, H, W = heatmap.shape
    hw = H * W
    lim = np.asarray(post_center_limit_range, F)
    out = []
    for b in range(B):
        flat = heatmap[b].reshape(-1)
        order = np.argsort(-flat, kind="stable")[:min(K, flat.size)]          # :166-172, ties: lower flat index first
        scores = flat[order]
        cls = order // hw                                                      # :173 (index of the class plane)
        pix = order % hw                                                       # :168
        ys = (pix // W).astype(F)                                              # :169
        xs = (pix % W).astype(F)                                               # :170
        g = lambda t, ch: np.asarray(t, F)[b, ch].reshape(-1)[pix]             # :156-160 gather at the selected pixels
        xs = xs + g(center, 0)                                                 # :204
        ys = ys + g(center, 1)                                                 # :205
        xs = (xs * F(feature_map_stride)) * F(voxel_size[0]) + F(point_cloud_range[0])   # :207, left to right in float32
        ys = (ys * F(feature_map_stride)) * F(voxel_size[1]) + F(point_cloud_range[1])   # :208
        angle = np.arctan2(g(rot_sin, 0), g(rot_cos, 0)).astype(F)             # :203
        parts = [xs, ys, g(center_z, 0), g(dim, 0), g(dim, 1), g(dim, 2), angle]
        if vel is not None:
            parts += [g(vel, 0), g(vel, 1)]                                    # :211-213
        boxes = np.stack(parts, 1).astype(F)
        mask = (boxes[:, :3] >= lim[:3]).all(1) & (boxes[:, :3] <= lim[3:]).all(1)   # :223-224
        if score_thresh is not None:
            mask &= scores > F(score_thresh)                                   # :226-227
        out.append(dict(pred_boxes=boxes[mask], pred_scores=scores[mask], pred_labels=cls[mask].astype(np.int64)))
    return out


def decode_bbox_from_voxels_nuscenes(batch_size, indices, obj, rot_cos, rot_sin, center, center_z, dim, vel=None,
                                     point_cloud_range=None, voxel_size=None, feature_map_stride=None, K=100, score_thresh=None,
                                     post_center_limit_range=None):
    """centernet_utils.py:303-367 with `_topk_1d(nuscenes=True)` (:256-289) and `gather_feat_idx` (:291-301): per frame the K
    best of obj[rows of the frame].T (class-major flat index c * rows + r), ties: lower flat index first."""
    indices = np.asarray(indices)
    obj = np.asarray(obj, F)
    lim = np.asarray(post_center_limit_range, F)
    per_row_stride = not np.isscalar(feature_map_stride)
    out = []
    for b in range(batch_size):
        rows = np.nonzero(indices[:, 0] == b)[0]                               # :263 batch_inds
        nb = len(rows)
        flat = obj[rows].T.reshape(-1)                                         # :269 score = obj[batch_inds].permute(1, 0)
        order = np.argsort(-flat, kind="stable")[:min(K, flat.size)]           # :270-271
        scores = flat[order]
        cls = order // max(nb, 1)                                              # class plane of the flat index (:274 for rows >= K)
        row = rows[order % max(nb, 1)] if nb else rows[:0]                     # :275
        g = lambda t, ch: np.asarray(t, F)[row, ch]                            # :291-301
        st = np.asarray(feature_map_stride, F).reshape(-1)[row] if per_row_stride else F(feature_map_stride)
        xs = ((indices[row, -1].astype(F) + g(center, 0)) * st) * F(voxel_size[0]) + F(point_cloud_range[0])   # :324
        ys = ((indices[row, -2].astype(F) + g(center, 1)) * st) * F(voxel_size[1]) + F(point_cloud_range[1])   # :325
        angle = np.arctan2(g(rot_sin, 0), g(rot_cos, 0)).astype(F)             # :323
        parts = [xs, ys, g(center_z, 0), g(dim, 0), g(dim, 1), g(dim, 2), angle]
        if vel is not None:
            parts += [g(vel, 0), g(vel, 1)]
        boxes = np.stack(parts, 1).astype(F) if len(row) else np.zeros((0, 9 if vel is not None else 7), F)
        mask = (boxes[:, :3] >= lim[:3]).all(1) & (boxes[:, :3] <= lim[3:]).all(1)      # :345-346
        if score_thresh is not None:
            mask &= scores > F(score_thresh)                                   # :348-349
        out.append(dict(pred_boxes=boxes[mask], pred_scores=scores[mask], pred_labels=cls[mask].astype(np.int64)))
    return out
