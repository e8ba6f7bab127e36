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
