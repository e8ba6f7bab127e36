"""GPU: pvx_nms_static (the one entry point that allocates nothing: its suppression mask lives in a static device array
inside the library) against pvx_nms (caller workspace) through the C ABI.  Bar: keep lists and counts bit-exact, on two
calls in a row, the second with a smaller input (the array keeps the first call's contents); also the rotated and the
axis-aligned form, the size limit, and that the call makes no device allocation."""
import numpy as np
import pytest
import torch

from openpcdet_amd import _lib

pytestmark = pytest.mark.gpu


def _rand_boxes(rng, n, spread):
    return np.c_[rng.uniform(-spread, spread, (n, 2)), rng.uniform(-1, 1, n), rng.uniform(0.3, 5, (n, 3)),
                 rng.uniform(-np.pi, np.pi, n)].astype(np.float32)


def _nms_ws(L, boxes, thresh, normal):
    n = boxes.shape[0]
    keep = torch.full((max(n, 1),), -7, dtype=torch.int64, device=boxes.device)
    num = torch.full((1,), -7, dtype=torch.int32, device=boxes.device)
    ws = _lib.workspace(L.pvx_nms_workspace_bytes(n, 1), boxes.device)
    _lib.check(L.pvx_nms(_lib.ptr(boxes), n, thresh, normal, _lib.ptr(keep), _lib.ptr(num), _lib.ptr(ws), ws.numel(),
                         _lib.stream(boxes.device)), "pvx_nms")
    return keep, num


def _nms_static(L, boxes, thresh, normal):
    n = boxes.shape[0]
    keep = torch.full((max(n, 1),), -7, dtype=torch.int64, device=boxes.device)
    num = torch.full((1,), -7, dtype=torch.int32, device=boxes.device)
    rc = L.pvx_nms_static(_lib.ptr(boxes), n, thresh, normal, _lib.ptr(keep), _lib.ptr(num), _lib.stream(boxes.device))
    return rc, keep, num


def test_nms_static_matches_workspace_entry_point(cuda_device):
    L = _lib.lib()
    rng = np.random.default_rng(77)
    # two calls in a row on the same stream, the second smaller: stale words of the first call stay in the static array
    for normal in (0, 1):
        for n, spread, thresh in ((4096, 40.0, 0.3), (777, 9.0, 0.1), (4095, 25.0, 0.7), (64, 3.0, 0.2), (1, 1.0, 0.5)):
            boxes = torch.from_numpy(_rand_boxes(rng, n, spread)).to(cuda_device)
            rc, keep_s, num_s = _nms_static(L, boxes, thresh, normal)
            assert rc == _lib.PVX_OK
            keep_w, num_w = _nms_ws(L, boxes, thresh, normal)
            torch.cuda.synchronize()
            k = int(num_w.item())
            assert int(num_s.item()) == k and 0 < k <= n
            assert torch.equal(keep_s[:k], keep_w[:k]), (normal, n)


def test_nms_static_limits_and_no_allocation(cuda_device):
    L = _lib.lib()
    rng = np.random.default_rng(78)
    big = torch.from_numpy(_rand_boxes(rng, 4097, 30.0)).to(cuda_device)
    rc, _, _ = _nms_static(L, big, 0.5, 0)
    assert rc == -4                                         # PVX_ERR_RANGE: beyond PVX_NMS_STATIC_MAX
    boxes = torch.from_numpy(_rand_boxes(rng, 1000, 12.0)).to(cuda_device)
    keep = torch.empty(1000, dtype=torch.int64, device=cuda_device)
    num = torch.zeros(1, dtype=torch.int32, device=cuda_device)
    rc = L.pvx_nms_static(_lib.ptr(boxes), 0, 0.5, 0, _lib.ptr(keep), _lib.ptr(num), _lib.stream(cuda_device))
    torch.cuda.synchronize()
    assert rc == _lib.PVX_OK and int(num.item()) == 0       # empty input
    L.pvx_nms_static(_lib.ptr(boxes), 1000, 0.5, 0, _lib.ptr(keep), _lib.ptr(num), _lib.stream(cuda_device))   # resolves the symbol
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(cuda_device)
    for _ in range(20):
        assert L.pvx_nms_static(_lib.ptr(boxes), 1000, 0.5, 0, _lib.ptr(keep), _lib.ptr(num), _lib.stream(cuda_device)) == 0
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(cuda_device)
    assert free1 == free0                                   # device memory in use did not move


def test_nms_static_golden_keep_lists(cuda_device):
    """the reference-derived keep lists of tests/golden/iou3d_golden.npz (the reference's own CPU IoU + its greedy rule, made by
    tests/golden/make_iou3d_golden.py) through pvx_nms_static: bit-exact, every case, back to back on one stream"""
    import os
    L = _lib.lib()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "iou3d_golden.npz"))
    names = [str(s) for s in g["case_names"]]
    assert len(names) == 8
    for name in names + names[::-1]:        # a second pass in the other order: larger-then-smaller and smaller-then-larger neighbours
        boxes = torch.from_numpy(g[name + "_boxes"]).to(cuda_device)
        want = g[name + "_keep"]
        rc, keep, num = _nms_static(L, boxes, float(g[name + "_thresh"]), 0)
        torch.cuda.synchronize()
        assert rc == _lib.PVX_OK
        assert int(num.item()) == len(want), name
        assert np.array_equal(keep[:len(want)].cpu().numpy(), want), name
