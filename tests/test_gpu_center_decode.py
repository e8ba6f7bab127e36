"""pvx_center_decode (through the centernet_utils drop-in) against the numpy oracle: selection, order, labels and box positions
bit-exact; heading within 1e-6 (atan2f of the device vs libm)."""
import numpy as np
import pytest
import torch

from openpcdet_amd import centernet_utils
from oracle import center_decode
from test_oracle_center_decode import GEOM, make_inputs

pytestmark = pytest.mark.gpu


def run_both(d, K, thr, dev, views=True, lim=None, geom=GEOM):
    lim = geom["lim"] if lim is None else lim
    t = {k: (torch.from_numpy(v).to(dev) if v is not None else None) for k, v in d.items()}
    rc, rs = (t["rot"][:, 0:1], t["rot"][:, 1:2]) if views else (t["rot"][:, 0:1].contiguous(), t["rot"][:, 1:2].contiguous())
    got = centernet_utils.decode_bbox_from_heatmap(t["heatmap"], rc, rs, t["center"], t["center_z"], t["dim"],
                                                   point_cloud_range=geom["pcr"], voxel_size=geom["vs"],
                                                   feature_map_stride=geom["stride"], vel=t["vel"], K=K, score_thresh=thr,
                                                   post_center_limit_range=lim)
    want = center_decode.decode_bbox_from_heatmap(d["heatmap"], d["rot"][:, 0:1], d["rot"][:, 1:2], d["center"], d["center_z"],
                                                  d["dim"], geom["pcr"], geom["vs"], geom["stride"], vel=d["vel"], K=K,
                                                  score_thresh=thr, post_center_limit_range=lim)
    return got, want


def check(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g["pred_labels"].dtype == torch.int64
        assert np.array_equal(g["pred_scores"].cpu().numpy(), w["pred_scores"])
        assert np.array_equal(g["pred_labels"].cpu().numpy(), w["pred_labels"])
        gb = g["pred_boxes"].cpu().numpy()
        assert gb.shape == w["pred_boxes"].shape
        assert np.array_equal(gb[:, :6], w["pred_boxes"][:, :6])
        assert np.allclose(gb[:, 6], w["pred_boxes"][:, 6], rtol=0, atol=1e-6)
        assert np.array_equal(gb[:, 7:], w["pred_boxes"][:, 7:])


@pytest.mark.parametrize("B,C,H,W,K,thr,vel,views", [(2, 2, 40, 36, 100, 0.1, True, True), (1, 1, 30, 30, 50, None, False, False),
                                                     (4, 2, 180, 180, 500, 0.1, True, True), (3, 3, 24, 20, 1024, 0.3, True, True)])
def test_decode_matches_oracle(B, C, H, W, K, thr, vel, views, cuda_device):
    d = make_inputs(B * 100 + C, B, C, H, W, vel)
    check(*run_both(d, K, thr, cuda_device, views=views))


def test_decode_ties_short_frames_and_empty_results(cuda_device):
    # heavy ties: scores quantised to 32 levels -> the threshold value is shared by many cells, lowest flat indices win
    d = make_inputs(7, 2, 2, 50, 44, True)
    d["heatmap"] = (np.floor(d["heatmap"] * 32) / 32).astype(np.float32)
    check(*run_both(d, 300, None, cuda_device))
    check(*run_both(d, 300, 0.25, cuda_device))
    # K larger than the number of cells
    d = make_inputs(8, 2, 1, 6, 5, False)
    check(*run_both(d, 100, None, cuda_device))
    # nothing survives the masks
    d = make_inputs(9, 2, 2, 20, 20, True)
    got, want = run_both(d, 64, 2.0, cuda_device)
    check(got, want)
    assert all(g["pred_boxes"].shape[0] == 0 for g in got)
    got, want = run_both(d, 64, None, cuda_device, lim=[1000, 1000, 1000, 1001, 1001, 1001])
    check(got, want)
    # argument checks
    with pytest.raises(RuntimeError):
        run_both(d, 2000, None, cuda_device)
    t = torch.zeros((1, 1, 4, 4), device=cuda_device)
    with pytest.raises(NotImplementedError):
        centernet_utils.decode_bbox_from_heatmap(t, t, t, t, t, t, circle_nms=True, post_center_limit_range=[0] * 6)


def test_center_head_post_processing_all_heads_one_nms(cuda_device):
    """generate_predicted_boxes (6 heads x B frames: one decode launch per head, one batched NMS) against the oracle decode +
    the oracle NMS composed as center_head.py:332-389 composes them"""
    from openpcdet_amd import center_head_post
    from oracle import loader
    r = np.random.default_rng(3)
    B, H, W, K = 2, 64, 60, 200
    classes = [1, 2, 2, 1, 2, 2]
    mapping, base = [], 0
    for c in classes:
        mapping.append(torch.arange(base, base + c, dtype=torch.int64)); base += c
    cfg = dict(MAX_OBJ_PER_SAMPLE=K, SCORE_THRESH=0.1, POST_CENTER_LIMIT_RANGE=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
               NMS_CONFIG=dict(NMS_TYPE="nms_gpu", NMS_THRESH=0.2, NMS_PRE_MAXSIZE=150, NMS_POST_MAXSIZE=83))
    pcr, vs, stride = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0], [0.1, 0.1, 0.2], 8
    pred = []
    for c in classes:
        # clustered peaks so that the NMS has something to suppress
        hm = r.normal(-4.0, 1.0, (B, c, H, W))
        for _ in range(25):
            b, k, y, x = r.integers(B), r.integers(c), r.integers(2, H - 2), r.integers(2, W - 2)
            hm[b, k, y - 1:y + 2, x - 1:x + 2] += r.uniform(3.0, 6.0)
        pred.append(dict(hm=hm.astype(np.float32), center=r.uniform(0, 1, (B, 2, H, W)).astype(np.float32),
                         center_z=r.uniform(-2, 1, (B, 1, H, W)).astype(np.float32),
                         dim=r.normal(1.0, 0.3, (B, 3, H, W)).astype(np.float32),
                         rot=r.normal(size=(B, 2, H, W)).astype(np.float32), vel=r.normal(size=(B, 2, H, W)).astype(np.float32)))
    dev_pred = [{k: torch.from_numpy(v).to(cuda_device) for k, v in p.items()} for p in pred]
    got = center_head_post.generate_predicted_boxes(B, dev_pred, mapping, pcr, vs, stride, cfg, has_vel=True)
    assert len(got) == B
    want = [dict(b=[], s=[], l=[]) for _ in range(B)]
    tight = False
    for h, p in enumerate(dev_pred):
        hm = p["hm"].sigmoid().cpu().numpy(); dim = p["dim"].exp().cpu().numpy()          # the activations the device path used
        dec = center_decode.decode_bbox_from_heatmap(hm, pred[h]["rot"][:, 0:1], pred[h]["rot"][:, 1:2], pred[h]["center"],
                                                     pred[h]["center_z"], dim, pcr, vs, stride, vel=pred[h]["vel"], K=K,
                                                     score_thresh=cfg["SCORE_THRESH"],
                                                     post_center_limit_range=cfg["POST_CENTER_LIMIT_RANGE"])
        for b in range(B):
            bx, sc, lb = dec[b]["pred_boxes"][:150], dec[b]["pred_scores"][:150], dec[b]["pred_labels"][:150]
            keep, margin = loader.nms(bx[:, :7], 0.2) if len(bx) else (np.zeros(0, np.int64), 1.0)
            tight |= margin < 1e-4
            keep = keep[:83]
            want[b]["b"].append(bx[keep]); want[b]["s"].append(sc[keep]); want[b]["l"].append(mapping[h].numpy()[lb[keep]] + 1)
    assert not tight, "an IoU sits within 1e-4 of the threshold: pick another seed"
    total = 0
    for b in range(B):
        wb, wsc, wl = np.concatenate(want[b]["b"]), np.concatenate(want[b]["s"]), np.concatenate(want[b]["l"])
        total += len(wsc)
        assert np.array_equal(got[b]["pred_scores"].cpu().numpy(), wsc)
        assert np.array_equal(got[b]["pred_labels"].cpu().numpy(), wl)
        gb = got[b]["pred_boxes"].cpu().numpy()
        assert np.array_equal(gb[:, :6], wb[:, :6]) and np.array_equal(gb[:, 7:], wb[:, 7:])
        assert np.allclose(gb[:, 6], wb[:, 6], rtol=0, atol=1e-6)
    assert total > 50            # the case is not vacuous


@pytest.mark.parametrize("rows,C,K,thr,vel,prs", [((900, 700), 2, 100, 0.1, True, False), ((400,), 1, 50, None, False, False),
                                                  ((600, 650, 800), 3, 200, 0.2, True, True), ((30000, 25000), 2, 500, 0.1, True, False),
                                                  ((40, 0, 500), 2, 100, None, True, False)])
def test_sparse_decode_matches_oracle(rows, C, K, thr, vel, prs, cuda_device):
    """VoxelNeXt form (pvx_voxel_decode through decode_bbox_from_voxels_nuscenes), incl. a frame with fewer rows than K and an
    empty frame (the reference's torch.stack cannot express those; the natural result is checked)"""
    from test_oracle_center_decode import make_sparse_inputs
    d = make_sparse_inputs(len(rows) * 10 + C, rows, C, vel=vel, per_row_stride=prs)
    t = {k: (torch.from_numpy(v).to(cuda_device) if isinstance(v, np.ndarray) else v) for k, v in d.items()}
    got = centernet_utils.decode_bbox_from_voxels_nuscenes(len(rows), t["indices"], t["obj"], t["rot_cos"], t["rot_sin"], t["center"],
                                                           t["center_z"], t["dim"], vel=t["vel"], point_cloud_range=GEOM["pcr"],
                                                           voxel_size=GEOM["vs"], feature_map_stride=t["stride"], K=K,
                                                           score_thresh=thr, post_center_limit_range=GEOM["lim"])
    want = center_decode.decode_bbox_from_voxels_nuscenes(len(rows), d["indices"], d["obj"], d["rot_cos"], d["rot_sin"], d["center"],
                                                          d["center_z"], d["dim"], vel=d["vel"], point_cloud_range=GEOM["pcr"],
                                                          voxel_size=GEOM["vs"], feature_map_stride=d["stride"], K=K,
                                                          score_thresh=thr, post_center_limit_range=GEOM["lim"])
    check(got, want)
    assert all(g["pred_ious"] is None and g["add_features"] is None for g in got)
