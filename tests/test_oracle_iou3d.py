"""CPU: pins the C restatement (oracle/iou3d_oracle.c) and the library's HOST entry points against
(a) golden vectors produced by the reference's own iou3d_cpu.cpp and (b), when oracle/_ref is present, the
reference build itself on fresh random inputs.  Bar: bit-exact IoU values and keep lists."""
import os

import numpy as np
import pytest
import torch

from oracle import loader
from openpcdet_amd.ops.iou3d_nms import iou3d_nms_utils


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "iou3d_golden.npz"))


def _bits(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


def test_oracle_matrix_matches_reference_golden(G):
    for name in ("dense200", "tile64", "tile65", "tile129", "single"):
        boxes = G[name + "_boxes"]
        got = loader.boxes_bev_matrix(boxes, boxes, 0)
        assert np.array_equal(_bits(got), _bits(G[name + "_iou"])), name
    got = loader.boxes_bev_matrix(G["rect_a"], G["rect_b"], 0)
    assert np.array_equal(_bits(got), _bits(G["rect_iou"]))


def test_oracle_edge_cases(G):
    eb = G["edge_boxes"]
    got = loader.boxes_bev_matrix(eb, eb, 0)
    assert np.array_equal(_bits(got), _bits(G["edge_iou"]))
    got = loader.boxes_bev_paired(eb, eb[::-1].copy(), 0)
    assert np.array_equal(_bits(got), _bits(G["edge_aligned_iou"]))
    # identical boxes -> IoU 1 within rounding, far apart -> exactly 0
    assert abs(G["edge_iou"][0, 1] - 1.0) < 1e-5
    assert G["edge_iou"][0, 6] == 0.0


def test_oracle_nms_matches_reference_golden(G):
    for name in G["case_names"]:
        name = str(name)
        keep, margin = loader.nms(G[name + "_boxes"], float(G[name + "_thresh"]))
        assert np.array_equal(keep, G[name + "_keep"]), name
        if np.isfinite(G[name + "_margin"]):
            assert margin <= float(G[name + "_margin"]) + 1e-12   # oracle sees a superset of the pairs


def test_oracle_nms_empty_and_normal():
    keep, _ = loader.nms(np.zeros((0, 7), np.float32), 0.5)
    assert keep.size == 0
    b = np.array([[0, 0, 0, 2, 2, 1, 0.3], [0.5, 0, 0, 2, 2, 1, 1.0], [5, 5, 0, 2, 2, 1, 0]], np.float32)
    keep, _ = loader.nms(b, 0.3, normal=True)   # headings ignored: 1.5*2/(8-3)=0.6 > 0.3
    assert keep.tolist() == [0, 2]


@pytest.mark.skipif(not loader.have_ref(), reason="oracle/_ref not built (reference sources absent)")
def test_oracle_vs_reference_build_random():
    rng = np.random.default_rng(7)
    for spread in (2.0, 8.0, 60.0):
        a = np.c_[rng.uniform(-spread, spread, (150, 2)), rng.uniform(-1, 1, 150), rng.uniform(0.2, 6, (150, 3)),
                  rng.uniform(-7, 7, 150)].astype(np.float32)
        b = np.c_[rng.uniform(-spread, spread, (90, 2)), rng.uniform(-1, 1, 90), rng.uniform(0.2, 6, (90, 3)),
                  rng.uniform(-7, 7, 90)].astype(np.float32)
        assert np.array_equal(_bits(loader.boxes_bev_matrix(a, b, 0)), _bits(loader.ref_boxes_iou_bev(a, b)))
        assert np.array_equal(_bits(loader.boxes_bev_paired(a[:90], b, 0)), _bits(loader.ref_boxes_aligned_iou_bev(a[:90], b)))


def test_library_host_entry_points_match_reference_golden(G):
    """pvx_boxes_iou_bev_cpu through the reference-shaped Python wrapper (numpy in -> numpy out)."""
    for name in ("dense200", "tile129"):
        boxes = G[name + "_boxes"]
        got = iou3d_nms_utils.boxes_bev_iou_cpu(boxes, boxes)
        assert isinstance(got, np.ndarray)
        assert np.array_equal(_bits(got), _bits(G[name + "_iou"])), name
    got = iou3d_nms_utils.boxes_bev_iou_cpu(torch.from_numpy(G["rect_a"]), torch.from_numpy(G["rect_b"]))
    assert np.array_equal(_bits(got.numpy()), _bits(G["rect_iou"]))
    eb = G["edge_boxes"]
    assert np.array_equal(_bits(iou3d_nms_utils.boxes_bev_iou_cpu(eb, eb)), _bits(G["edge_iou"]))
