"""The decode oracle against the reference's own tensor sequence (centernet_utils.py:163-254) executed with CPU torch ops.
The reference file itself is not importable here (it imports numba, centernet_utils.py:6), so its steps are restated below
line by line with the torch calls it makes; inputs are tie-free (torch.topk's order among equal scores is unspecified)."""
import numpy as np
import torch

from oracle import center_decode


def _gather_feat(feat, ind):                                      # centernet_utils.py:145-153 (mask=None)
    dim = feat.size(2)
    ind = ind.unsqueeze(2).expand(ind.size(0), ind.size(1), dim)
    return feat.gather(1, ind)


def _transpose_and_gather_feat(feat, ind):                        # :156-160
    feat = feat.permute(0, 2, 3, 1).contiguous()
    feat = feat.view(feat.size(0), -1, feat.size(3))
    return _gather_feat(feat, ind)


def _topk(scores, K):                                             # :163-178
    batch, num_class, height, width = scores.size()
    topk_scores, topk_inds = torch.topk(scores.flatten(2, 3), K)
    topk_inds = topk_inds % (height * width)
    topk_ys = (topk_inds // width).float()
    topk_xs = (topk_inds % width).int().float()
    topk_score, topk_ind = torch.topk(topk_scores.view(batch, -1), K)
    topk_classes = (topk_ind // K).int()
    topk_inds = _gather_feat(topk_inds.view(batch, -1, 1), topk_ind).view(batch, K)
    topk_ys = _gather_feat(topk_ys.view(batch, -1, 1), topk_ind).view(batch, K)
    topk_xs = _gather_feat(topk_xs.view(batch, -1, 1), topk_ind).view(batch, K)
    return topk_score, topk_inds, topk_classes, topk_ys, topk_xs


def reference_sequence(heatmap, rot_cos, rot_sin, center, center_z, dim, pcr, vs, stride, vel, K, score_thresh, lim):   # :186-254
    batch_size = heatmap.size(0)
    scores, inds, class_ids, ys, xs = _topk(heatmap, K=K)
    center = _transpose_and_gather_feat(center, inds).view(batch_size, K, 2)
    rot_sin = _transpose_and_gather_feat(rot_sin, inds).view(batch_size, K, 1)
    rot_cos = _transpose_and_gather_feat(rot_cos, inds).view(batch_size, K, 1)
    center_z = _transpose_and_gather_feat(center_z, inds).view(batch_size, K, 1)
    dim = _transpose_and_gather_feat(dim, inds).view(batch_size, K, 3)
    angle = torch.atan2(rot_sin, rot_cos)
    xs = xs.view(batch_size, K, 1) + center[:, :, 0:1]
    ys = ys.view(batch_size, K, 1) + center[:, :, 1:2]
    xs = xs * stride * vs[0] + pcr[0]
    ys = ys * stride * vs[1] + pcr[1]
    parts = [xs, ys, center_z, dim, angle]
    if vel is not None:
        parts.append(_transpose_and_gather_feat(vel, inds).view(batch_size, K, 2))
    final_box_preds = torch.cat(parts, dim=-1)
    final_scores = scores.view(batch_size, K)
    final_class_ids = class_ids.view(batch_size, K)
    lim = torch.tensor(lim).float()
    mask = (final_box_preds[..., :3] >= lim[:3]).all(2)
    mask &= (final_box_preds[..., :3] <= lim[3:]).all(2)
    if score_thresh is not None:
        mask &= (final_scores > score_thresh)
    return [dict(pred_boxes=final_box_preds[k, mask[k]], pred_scores=final_scores[k, mask[k]], pred_labels=final_class_ids[k, mask[k]])
            for k in range(batch_size)]


def make_inputs(seed, B, C, H, W, vel=True):
    r = np.random.default_rng(seed)
    hm = 1.0 / (1.0 + np.exp(-r.normal(-2.0, 1.5, (B, C, H, W))))          # sigmoid-ed scores, tie-free
    d = dict(heatmap=hm.astype(np.float32), center=r.uniform(0, 1, (B, 2, H, W)).astype(np.float32),
             center_z=r.uniform(-3, 2, (B, 1, H, W)).astype(np.float32),
             dim=np.exp(r.normal(0.5, 0.4, (B, 3, H, W))).astype(np.float32),
             rot=r.normal(size=(B, 2, H, W)).astype(np.float32),
             vel=r.normal(size=(B, 2, H, W)).astype(np.float32) if vel else None)
    return d


GEOM = dict(pcr=[-54.0, -54.0, -5.0, 54.0, 54.0, 3.0], vs=[0.075, 0.075, 0.2], stride=8,
            lim=[-40.0, -61.2, -2.0, 61.2, 40.0, 1.5])


def test_oracle_follows_the_reference_sequence():
    for seed, (B, C, H, W, K, thr, vel) in enumerate([(2, 2, 40, 36, 100, 0.1, True), (1, 1, 30, 30, 50, None, False),
                                                       (3, 3, 24, 20, 200, 0.3, True)]):
        d = make_inputs(seed, B, C, H, W, vel)
        t = {k: (torch.from_numpy(v) if v is not None else None) for k, v in d.items()}
        want = reference_sequence(t["heatmap"], t["rot"][:, 0:1], t["rot"][:, 1:2], t["center"], t["center_z"], t["dim"],
                                  GEOM["pcr"], GEOM["vs"], GEOM["stride"], t["vel"], K, thr, GEOM["lim"])
        got = center_decode.decode_bbox_from_heatmap(d["heatmap"], d["rot"][:, 0:1], d["rot"][:, 1:2], d["center"], d["center_z"],
                                                     d["dim"], GEOM["pcr"], GEOM["vs"], GEOM["stride"], vel=d["vel"], K=K,
                                                     score_thresh=thr, post_center_limit_range=GEOM["lim"])
        assert len(got) == B
        for g, w in zip(got, want):
            assert np.array_equal(g["pred_scores"], w["pred_scores"].numpy())
            assert np.array_equal(g["pred_labels"], w["pred_labels"].numpy().astype(np.int64))
            wb = w["pred_boxes"].numpy()
            assert g["pred_boxes"].shape == wb.shape
            assert np.array_equal(g["pred_boxes"][:, :6], wb[:, :6])          # positions / sizes: same float32 operation order
            assert np.allclose(g["pred_boxes"][:, 6:], wb[:, 6:], rtol=0, atol=1e-6)   # atan2: libm vs torch, <= 1 ulp


def test_oracle_tie_order_and_short_frames():
    hm = np.zeros((1, 2, 4, 5), np.float32)
    hm[0, 1, 2, 3] = 0.9
    hm[0, 0, 1, 1] = 0.5; hm[0, 1, 0, 0] = 0.5; hm[0, 0, 3, 4] = 0.5          # three equal scores: flat-index order
    z = np.zeros((1, 1, 4, 5), np.float32)
    out = center_decode.decode_bbox_from_heatmap(hm, z + 1, z, np.zeros((1, 2, 4, 5), np.float32), z, np.ones((1, 3, 4, 5), np.float32),
                                                 [0, 0, -5, 0, 0, 3], [1.0, 1.0, 1.0], 1, vel=None, K=100, score_thresh=0.1,
                                                 post_center_limit_range=[-100, -100, -100, 100, 100, 100])[0]
    assert out["pred_scores"].tolist() == [np.float32(0.9), 0.5, 0.5, 0.5]
    assert out["pred_labels"].tolist() == [1, 0, 0, 1]
    assert out["pred_boxes"][:, :2].tolist() == [[3.0, 2.0], [1.0, 1.0], [4.0, 3.0], [0.0, 0.0]]


# ---- sparse (VoxelNeXt) form: centernet_utils.py:256-367 re-executed with CPU torch ops --------------------------------------
def _topk_1d_nuscenes(batch_size, batch_idx, obj, K):                     # :256-289 with scores=None, nuscenes=True
    score_l, inds_l, cls_l = [], [], []
    for bs_idx in range(batch_size):
        batch_inds = batch_idx == bs_idx
        score = obj[batch_inds].permute(1, 0)
        topk_scores, topk_inds = torch.topk(score, min(K, score.shape[-1]))
        topk_score, topk_ind = torch.topk(topk_scores.view(-1), min(K, topk_scores.view(-1).shape[-1]))
        topk_classes = (topk_ind // K).int()
        topk_inds = topk_inds.view(-1).gather(0, topk_ind)
        score_l.append(topk_score); inds_l.append(topk_inds); cls_l.append(topk_classes)
    return torch.stack(score_l), torch.stack(inds_l), torch.stack(cls_l)


def _gather_feat_idx(feats, inds, batch_size, batch_idx):                 # :291-301
    dim = feats.size(-1)
    _inds = inds.unsqueeze(-1).expand(inds.size(0), inds.size(1), dim)
    return torch.stack([feats[batch_idx == b].gather(0, _inds[b]) for b in range(batch_size)])


def sparse_reference_sequence(batch_size, indices, obj, rot_cos, rot_sin, center, center_z, dim, vel, pcr, vs, stride, K, thr, lim):
    batch_idx = indices[:, 0]
    spatial_indices = indices[:, 1:]
    scores, inds, class_ids = _topk_1d_nuscenes(batch_size, batch_idx, obj, K)
    g = lambda t: _gather_feat_idx(t, inds, batch_size, batch_idx)
    center, rot_sin, rot_cos, center_z, dim, spatial_indices = g(center), g(rot_sin), g(rot_cos), g(center_z), g(dim), g(spatial_indices)
    if not isinstance(stride, int):
        stride = g(stride.unsqueeze(-1))
    angle = torch.atan2(rot_sin, rot_cos)
    xs = (spatial_indices[:, :, -1:] + center[:, :, 0:1]) * stride * vs[0] + pcr[0]
    ys = (spatial_indices[:, :, -2:-1] + center[:, :, 1:2]) * stride * vs[1] + pcr[1]
    parts = [xs, ys, center_z, dim, angle]
    if vel is not None:
        parts.append(g(vel))
    final_box_preds = torch.cat(parts, dim=-1)
    final_scores = scores.view(batch_size, K)
    final_class_ids = class_ids.view(batch_size, K)
    lim = torch.tensor(lim).float()
    mask = (final_box_preds[..., :3] >= lim[:3]).all(2) & (final_box_preds[..., :3] <= lim[3:]).all(2)
    if thr is not None:
        mask &= final_scores > thr
    return [dict(pred_boxes=final_box_preds[k, mask[k]], pred_scores=final_scores[k, mask[k]], pred_labels=final_class_ids[k, mask[k]])
            for k in range(batch_size)]


def make_sparse_inputs(seed, rows_per_frame, C, H=180, W=180, vel=True, per_row_stride=False):
    r = np.random.default_rng(seed)
    idx = []
    for b, n in enumerate(rows_per_frame):
        cells = np.sort(r.choice(H * W, n, replace=False))
        idx.append(np.stack([np.full(n, b), cells // W, cells % W], 1))
    indices = np.concatenate(idx, 0).astype(np.int32)
    N = len(indices)
    d = dict(indices=indices, obj=(1.0 / (1.0 + np.exp(-r.normal(-2.0, 1.5, (N, C))))).astype(np.float32),
             center=r.uniform(0, 1, (N, 2)).astype(np.float32), center_z=r.uniform(-3, 2, (N, 1)).astype(np.float32),
             dim=np.exp(r.normal(0.5, 0.4, (N, 3))).astype(np.float32), rot_cos=r.normal(size=(N, 1)).astype(np.float32),
             rot_sin=r.normal(size=(N, 1)).astype(np.float32), vel=r.normal(size=(N, 2)).astype(np.float32) if vel else None,
             stride=(r.choice([8.0, 16.0, 32.0], N).astype(np.float32) if per_row_stride else 8))
    return d


def test_sparse_oracle_follows_the_reference_sequence():
    for seed, (rows, C, K, thr, vel, prs) in enumerate([((900, 700), 2, 100, 0.1, True, False), ((400,), 1, 50, None, False, False),
                                                         ((600, 650, 800), 3, 200, 0.2, True, True)]):
        d = make_sparse_inputs(seed, rows, C, vel=vel, per_row_stride=prs)
        t = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in d.items()}
        want = sparse_reference_sequence(len(rows), t["indices"], t["obj"], t["rot_cos"], t["rot_sin"], t["center"], t["center_z"],
                                         t["dim"], t["vel"], GEOM["pcr"], GEOM["vs"], t["stride"], K, thr, GEOM["lim"])
        got = center_decode.decode_bbox_from_voxels_nuscenes(len(rows), d["indices"], d["obj"], d["rot_cos"], d["rot_sin"], d["center"],
                                                             d["center_z"], d["dim"], vel=d["vel"], point_cloud_range=GEOM["pcr"],
                                                             voxel_size=GEOM["vs"], feature_map_stride=d["stride"], K=K,
                                                             score_thresh=thr, post_center_limit_range=GEOM["lim"])
        for g, w in zip(got, want):
            assert np.array_equal(g["pred_scores"], w["pred_scores"].numpy())
            assert np.array_equal(g["pred_labels"], w["pred_labels"].numpy().astype(np.int64))
            wb = w["pred_boxes"].numpy()
            assert g["pred_boxes"].shape == wb.shape
            assert np.array_equal(g["pred_boxes"][:, :6], wb[:, :6])
            assert np.allclose(g["pred_boxes"][:, 6], wb[:, 6], rtol=0, atol=1e-6)
            assert np.array_equal(g["pred_boxes"][:, 7:], wb[:, 7:])
