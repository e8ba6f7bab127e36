"""Generates tests/golden/iou3d_golden.npz from the REFERENCE's own CPU rotated IoU
(/root/reference/pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp compiled to oracle/_ref/libiou3d_ref.so by
oracle/Makefile) -- run in the build container only:

    make -C oracle ref && python tests/golden/make_iou3d_golden.py

Stored per case: boxes (already in descending-score order), thresh, the reference IoU matrix (small cases),
and the keep list obtained by applying the reference's greedy rule (iou3d_nms.cpp:163-179: box j is dropped
iff a kept box i < j has iou(i, j) > thresh) to reference IoUs, plus the decision margin min|iou - thresh|.
Fixtures are data only (inputs + expected outputs); no reference source text is stored.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import loader  # noqa: E402


def synth_boxes(rng, n, spread, clustered=0):
    b = np.zeros((n, 7), np.float32)
    if clustered:
        centres = rng.uniform(-spread, spread, (clustered, 2))
        pick = rng.integers(0, clustered, n)
        b[:, 0:2] = centres[pick] + rng.normal(0, 0.6, (n, 2))
        base_heading = rng.uniform(-np.pi, np.pi, clustered)[pick]
        b[:, 6] = base_heading + rng.normal(0, 0.15, n)
    else:
        b[:, 0:2] = rng.uniform(-spread, spread, (n, 2))
        b[:, 6] = rng.uniform(-np.pi, np.pi, n)
    b[:, 2] = rng.uniform(-1, 1, n)
    b[:, 3] = np.abs(rng.normal(3.9, 0.4, n)) + 0.1
    b[:, 4] = np.abs(rng.normal(1.6, 0.2, n)) + 0.1
    b[:, 5] = np.abs(rng.normal(1.56, 0.2, n)) + 0.1
    return b.astype(np.float32)


def greedy_from_iou(iou, thresh):
    n = iou.shape[0]
    removed = np.zeros(n, bool)
    keep = []
    margin = np.inf
    for i in range(n):
        if removed[i]:
            continue
        keep.append(i)
        row = iou[i, i + 1:]
        if row.size:
            margin = min(margin, float(np.abs(row.astype(np.float64) - np.float64(np.float32(thresh))).min()))
        removed[i + 1:] |= row > np.float32(thresh)
    return np.asarray(keep, np.int64), margin


def edge_case_boxes():
    """Hand-made degenerate inputs the reference arithmetic must be followed through."""
    rows = [
        [0, 0, 0, 4, 2, 1.5, 0.0],          # base
        [0, 0, 0, 4, 2, 1.5, 0.0],          # identical
        [0, 0, 0, 4, 2, 1.5, np.pi / 2],    # same centre, rotated 90 deg
        [4, 0, 0, 4, 2, 1.5, 0.0],          # touching edge (shares x = 2)
        [2, 1, 0, 4, 2, 1.5, 0.0],          # half overlap
        [0.5, 0.25, 0, 1, 0.5, 1, 0.3],     # fully contained
        [100, 100, 0, 4, 2, 1.5, 1.0],      # far away
        [0, 0, 0, 0, 0, 0, 0.0],            # zero size
        [0, 0, 0, 4, 2, 1.5, np.pi],        # rotated 180 deg
        [0.005, 0, 0, 4, 2, 1.5, 1e-4],     # nearly identical (inside MARGIN)
        [1, 1, 0, 2, 2, 2, np.pi / 4],      # diamond
        [-3, 0.5, 0, 4, 2, 1.5, -2.5],
    ]
    return np.asarray(rows, np.float32)


def main():
    assert loader.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    rng = np.random.default_rng(20240607)
    out = {}
    cases = [
        # name, n, spread, clustered, thresh, store_matrix
        ("dense200", 200, 6.0, 0, 0.2, True),
        ("centerpoint500", 500, 50.0, 40, 0.2, False),
        ("kitti1000", 1000, 30.0, 0, 0.01, False),
        ("waymo700", 700, 40.0, 60, 0.7, False),
        ("tile64", 64, 5.0, 0, 0.3, True),
        ("tile65", 65, 5.0, 0, 0.3, True),
        ("tile129", 129, 8.0, 0, 0.1, True),
        ("single", 1, 1.0, 0, 0.5, True),
    ]
    names = []
    for name, n, spread, clustered, thresh, store in cases:
        boxes = synth_boxes(rng, n, spread, clustered)
        iou = loader.ref_boxes_iou_bev(boxes, boxes)
        keep, margin = greedy_from_iou(iou, thresh)
        out[name + "_boxes"] = boxes
        out[name + "_thresh"] = np.float32(thresh)
        out[name + "_keep"] = keep
        out[name + "_margin"] = np.float64(margin)
        if store:
            out[name + "_iou"] = iou
        names.append(name)
        print("%-16s n=%4d kept=%4d margin=%.3e overlap_frac=%.3f" % (name, n, keep.size, margin, (iou > 0).mean()))
    eb = edge_case_boxes()
    out["edge_boxes"] = eb
    out["edge_iou"] = loader.ref_boxes_iou_bev(eb, eb)
    out["edge_aligned_iou"] = loader.ref_boxes_aligned_iou_bev(eb, eb[::-1].copy())
    # two different sets, rectangular matrix
    a = synth_boxes(rng, 70, 8.0)
    b = synth_boxes(rng, 45, 8.0)
    out["rect_a"], out["rect_b"] = a, b
    out["rect_iou"] = loader.ref_boxes_iou_bev(a, b)
    out["case_names"] = np.asarray(names)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "iou3d_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
