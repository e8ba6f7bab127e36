"""GPU parity: the gfx950 IoU / NMS kernels through the C ABI vs the reference-pinned oracle and the golden
vectors made by the reference's own iou3d_cpu.cpp.  Bar: NMS keep lists bit-exact (every fixture's recorded decision margin min|iou - thresh| is > 1e-5);
IoU / overlap values are floating point: |diff| <= 1e-5 + 1e-5*|ref| (IOU_TOL below), and >= 97 % of the entries must
be bit-identical -- the only non-identical operation is sin/cos/atan2 rounding (csrc/iou_core.h header)."""
import os

import numpy as np
import pytest
import torch

from oracle import loader
from openpcdet_amd.ops.iou3d_nms import iou3d_nms_utils as U
from openpcdet_amd.ops.iou3d_nms import iou3d_nms_cuda

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "iou3d_golden.npz"))


def _rand_boxes(rng, n, spread):
    return np.c_[rng.uniform(-spread, spread, (n, 2)), rng.uniform(-1, 1, n), rng.uniform(0.3, 5, (n, 3)),
                 rng.uniform(-np.pi, np.pi, n)].astype(np.float32)


IOU_TOL = dict(rtol=1e-5, atol=1e-5)


def _check_values(got, want, what=""):
    got = np.ascontiguousarray(got, np.float32)
    want = np.ascontiguousarray(want, np.float32)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, err_msg=what, **IOU_TOL)
    assert np.array_equal(got == 0, want == 0), what + ": zero pattern differs"
    same = (got.view(np.uint32) == want.view(np.uint32)).mean()
    assert same >= 0.97, "%s: only %.4f of entries bit-identical" % (what, same)


def test_iou_matrix_golden(G, cuda_device):
    for name in ("dense200", "tile64", "tile65", "tile129", "single"):
        boxes = torch.from_numpy(G[name + "_boxes"]).to(cuda_device)
        got = U.boxes_iou_bev(boxes, boxes).cpu().numpy()
        _check_values(got, G[name + "_iou"], name)
    a = torch.from_numpy(G["rect_a"]).to(cuda_device)
    b = torch.from_numpy(G["rect_b"]).to(cuda_device)
    _check_values(U.boxes_iou_bev(a, b).cpu().numpy(), G["rect_iou"], "rect")


def test_iou_edge_cases(G, cuda_device):
    eb = torch.from_numpy(G["edge_boxes"]).to(cuda_device)
    got = U.boxes_iou_bev(eb, eb).cpu().numpy()
    np.testing.assert_allclose(got, G["edge_iou"], **IOU_TOL)


def test_overlap_paired_aligned_3d(cuda_device):
    rng = np.random.default_rng(3)
    a, b = _rand_boxes(rng, 333, 6.0), _rand_boxes(rng, 333, 6.0)
    ta, tb = torch.from_numpy(a).to(cuda_device), torch.from_numpy(b).to(cuda_device)
    ov = torch.zeros((333, 1), device=cuda_device)
    iou3d_nms_cuda.paired_boxes_overlap_bev_gpu(ta, tb, ov)
    want = loader.boxes_bev_paired(a, b, 1)
    _check_values(ov.cpu().numpy()[:, 0], want, "paired")
    ovm = torch.zeros((333, 333), device=cuda_device)
    iou3d_nms_cuda.boxes_overlap_bev_gpu(ta, tb, ovm)
    _check_values(ovm.cpu().numpy(), loader.boxes_bev_matrix(a, b, 1), "overlap matrix")
    # 3D IoU wrappers: shapes and agreement between the matrix diagonal and the paired forms
    m = U.boxes_iou3d_gpu(ta, tb)
    p = U.paired_boxes_iou3d_gpu(ta, tb)
    al = U.boxes_aligned_iou3d_gpu(ta, tb)
    assert m.shape == (333, 333) and p.shape == (333,) and al.shape == (333, 1)
    assert torch.equal(torch.diagonal(m), p) and torch.equal(al.view(-1), p)
    assert float(m.max()) <= 1.0 + 1e-5 and float(m.min()) >= 0.0


def test_nms_golden_keep_lists(G, cuda_device):
    for name in G["case_names"]:
        name = str(name)
        boxes = torch.from_numpy(G[name + "_boxes"]).to(cuda_device)
        n = boxes.shape[0]
        scores = torch.arange(n, 0, -1, device=cuda_device, dtype=torch.float32)   # already sorted
        sel, none = U.nms_gpu(boxes, scores, float(G[name + "_thresh"]))
        assert none is None
        assert sel.dtype == torch.int64
        assert np.array_equal(sel.cpu().numpy(), G[name + "_keep"]), name


def test_nms_reference_signature_cpu_keep(G, cuda_device):
    boxes = torch.from_numpy(G["centerpoint500_boxes"]).to(cuda_device)
    keep = torch.LongTensor(boxes.size(0))
    num = iou3d_nms_cuda.nms_gpu(boxes, keep, float(G["centerpoint500_thresh"]))
    assert num == G["centerpoint500_keep"].size
    assert np.array_equal(keep[:num].numpy(), G["centerpoint500_keep"])


def test_nms_random_vs_oracle_sizes(cuda_device):
    rng = np.random.default_rng(11)
    for n, spread, thresh in ((0, 1, 0.5), (1, 1, 0.5), (63, 4, 0.3), (64, 4, 0.3), (127, 5, 0.1), (500, 12, 0.2),
                              (1500, 25, 0.01), (4096, 60, 0.7)):
        b = _rand_boxes(rng, n, spread)
        s = rng.permutation(n).astype(np.float32)   # distinct scores
        order = np.argsort(-s, kind="stable")
        want, margin = loader.nms(b[order], thresh)
        sel, _ = U.nms_gpu(torch.from_numpy(b).to(cuda_device), torch.from_numpy(s).to(cuda_device), thresh)
        assert np.array_equal(sel.cpu().numpy(), order[want]), "n=%d margin=%g" % (n, margin)


def test_nms_pre_maxsize_and_normal(cuda_device):
    rng = np.random.default_rng(5)
    b = _rand_boxes(rng, 900, 10.0)
    s = rng.permutation(900).astype(np.float32)
    order = np.argsort(-s, kind="stable")[:300]
    want, _ = loader.nms(b[order], 0.25)
    sel, _ = U.nms_gpu(torch.from_numpy(b).to(cuda_device), torch.from_numpy(s).to(cuda_device), 0.25, pre_maxsize=300,
                       extra_kwarg_ignored=1)
    assert np.array_equal(sel.cpu().numpy(), order[want])
    order = np.argsort(-s, kind="stable")
    want, _ = loader.nms(b[order], 0.25, normal=True)
    sel, _ = U.nms_normal_gpu(torch.from_numpy(b).to(cuda_device), torch.from_numpy(s).to(cuda_device), 0.25)
    assert np.array_equal(sel.cpu().numpy(), order[want])


def test_nms_batched_segments(cuda_device):
    """6 heads x 2 frames, ragged counts incl. an empty segment, one launch."""
    rng = np.random.default_rng(21)
    S, cap = 12, 500
    boxes = np.zeros((S, cap, 9), np.float32)
    scores = np.zeros((S, cap), np.float32)
    counts = np.array([500, 499, 0, 1, 64, 65, 128, 300, 17, 500, 256, 2], np.int32)
    for s in range(S):
        boxes[s, :, :7] = _rand_boxes(rng, cap, 8.0 + s)
        scores[s] = rng.permutation(cap).astype(np.float32)
    sel, num = U.nms_batched(torch.from_numpy(boxes).to(cuda_device), torch.from_numpy(scores).to(cuda_device),
                             torch.from_numpy(counts).to(cuda_device), 0.2)
    sel, num = sel.cpu().numpy(), num.cpu().numpy()
    for s in range(S):
        c = counts[s]
        order = np.argsort(-scores[s, :c], kind="stable")
        want, _ = loader.nms(boxes[s, :c, :7][order], 0.2)
        assert num[s] == want.size, s
        assert np.array_equal(sel[s, :num[s]], order[want]), s


def test_nms_idempotent_full_size(cuda_device):
    """Size-independent property at the largest configured pre-NMS size (PV-RCNN++ 9000):
    NMS of the kept set keeps everything, and no kept pair exceeds the threshold."""
    rng = np.random.default_rng(33)
    b = torch.from_numpy(_rand_boxes(rng, 9000, 70.0)).to(cuda_device)
    s = torch.from_numpy(rng.permutation(9000).astype(np.float32)).to(cuda_device)
    sel, _ = U.nms_gpu(b, s, 0.8)
    kept = b[sel]
    sel2, _ = U.nms_gpu(kept, s[sel], 0.8)
    assert torch.equal(sel2, torch.arange(sel.numel(), device=cuda_device))
    iou = U.boxes_iou_bev(kept[:2000], kept[:2000])
    iou = torch.triu(iou, diagonal=1)
    assert float(iou.max()) <= 0.8


def _select_oracle(boxes, scores, count, thresh, normal, score_thresh, pre_max, post_max):
    """model_nms_utils.py:6-25 in numpy over the oracle NMS: indices into the segment's own boxes"""
    idx = np.arange(count)
    sc = scores[:count]
    if score_thresh is not None:
        idx = idx[sc >= score_thresh]
    order = idx[np.argsort(-scores[idx], kind="stable")]          # descending score, ties by ascending index
    if pre_max is not None:
        order = order[:pre_max]
    keep, margin = loader.nms(boxes[order][:, :7], thresh, normal=normal)
    return order[keep[:post_max]], margin


@pytest.mark.parametrize("cap,S", [(500, 7), (64, 3), (5000, 2), (16384, 1)])
def test_nms_select_vs_oracle(cuda_device, cap, S):
    """pvx_nms_select_batched (threshold + top-k sort + gather + NMS + post-max mapping on the device) vs the numpy
    restatement of class_agnostic_nms over the oracle NMS: bit-exact index lists.  Ragged counts with an empty segment,
    tied scores (ties resolve to the lower index), 9-wide boxes, LDS sort sizes below and above the 32 KiB opt-in."""
    rng = np.random.default_rng(100 + cap)
    boxes = np.zeros((S, cap, 9), np.float32)
    scores = np.zeros((S, cap), np.float32)
    counts = np.array(([cap, 0, cap - 1, 1, 65, cap // 2, 17] * 3)[:S], np.int32)
    for s in range(S):
        boxes[s, :, :7] = _rand_boxes(rng, cap, 0.35 * np.sqrt(cap) + s)
        boxes[s, :, 7:] = rng.normal(size=(cap, 2))
        scores[s] = (rng.integers(0, max(cap // 3, 2), cap) / max(cap // 3, 2)).astype(np.float32)    # many ties
    tb, ts, tc = (torch.from_numpy(x).to(cuda_device) for x in (boxes, scores, counts))
    cases = ((False, None, None, None), (False, 0.3, cap // 2 + 1, 83), (True, 0.1, None, 40), (False, 2.0, None, 10))
    if cap > 6000:
        cases = cases[1:2]                              # the CPU checker is O(n^2): one configuration at the largest size
    for normal, score_thresh, pre_max, post_max in cases:
        sel, num = U.nms_select(tb, ts, tc, 0.2, normal=normal, score_thresh=score_thresh, pre_max=pre_max, post_max=post_max)
        post = cap if post_max is None else post_max
        assert sel.shape == (S, post) and sel.dtype == torch.int64 and num.dtype == torch.int32
        sel, num = sel.cpu().numpy(), num.cpu().numpy()
        for s in range(S):
            want, margin = _select_oracle(boxes[s], scores[s], counts[s], 0.2, normal, score_thresh, pre_max, post)
            assert num[s] == want.size, (s, normal, score_thresh, margin)
            assert np.array_equal(sel[s, :num[s]], want), (s, normal, score_thresh, margin)
            if num[s] < post and want.size:
                assert np.all(sel[s, num[s]:] == want[0])
    # no counts array = every box valid; second call on the same buffers with other data (nothing stale is read)
    sel, num = U.nms_select(tb[:, :, :7].contiguous(), ts, None, 0.2, post_max=50)
    for s in range(S):
        want, _ = _select_oracle(boxes[s], scores[s], cap, 0.2, False, None, None, 50)
        assert np.array_equal(sel[s, :int(num[s])].cpu().numpy(), want)


@pytest.mark.parametrize("cap,spread,thresh", [(500, 9.0, 0.2), (1000, 60.0, 0.5), (1024, 3.0, 0.1), (1024, 200.0, 0.7),
                                               (130, 5.0, -1.0), (1025, 12.0, 0.2)])
def test_nms_many_small_problems(cuda_device, cap, spread, thresh):
    """16 or more problems of at most 1024 boxes go through the one-kernel form (mask and scan fused, IoUs only for rows that are kept);
    anything else through the tiled mask + scan.  Both must give the oracle's keep lists, and each other's: every segment is also run
    alone (one problem = the tiled form).  Spreads from 'everything overlaps' to 'nothing does' (kept rows per 64-box chunk from 1 to
    64: the column passes split), a negative threshold (every pair suppresses), ragged counts, rotated and axis-aligned."""
    rng = np.random.default_rng(700 + cap + int(spread))
    S = 18
    boxes = np.zeros((S, cap, 7), np.float32)
    scores = np.zeros((S, cap), np.float32)
    counts = np.array(([cap, cap - 1, 0, 1, 64, 65, cap // 2, 129, 63] * 2)[:S], np.int32)
    for s in range(S):
        boxes[s] = _rand_boxes(rng, cap, spread * (1 + 0.1 * s))
        scores[s] = rng.permutation(cap).astype(np.float32)
    tb, ts, tc = (torch.from_numpy(x).to(cuda_device) for x in (boxes, scores, counts))
    for normal in (False, True):
        sel, num = U.nms_select(tb, ts, tc, thresh, normal=normal, post_max=cap)
        sel2, num2 = U.nms_batched(tb, ts, tc, thresh) if not normal else (None, None)
        sel, num = sel.cpu().numpy(), num.cpu().numpy()
        for s in range(S):
            c = int(counts[s])
            order = np.argsort(-scores[s, :c], kind="stable")
            want, margin = loader.nms(boxes[s, :c][order], thresh, normal=normal)
            assert num[s] == want.size and np.array_equal(sel[s, :num[s]], order[want]), (s, normal, margin)
            if c and s % 3 == 0:
                alone, n_alone = U.nms_select(tb[s:s + 1], ts[s:s + 1], tc[s:s + 1], thresh, normal=normal, post_max=cap)
                assert int(n_alone[0]) == num[s] and np.array_equal(alone[0, :num[s]].cpu().numpy(), sel[s, :num[s]]), (s, normal)
            if sel2 is not None:
                assert int(num2[s]) == num[s] and np.array_equal(sel2[s, :num[s]].cpu().numpy(), sel[s, :num[s]]), s


def test_nms_select_limits(cuda_device):
    from openpcdet_amd import _lib
    L = _lib.lib()
    b = torch.zeros((1, 16385, 7), device=cuda_device)
    s = torch.zeros((1, 16385), device=cuda_device)
    sel = torch.zeros((1, 10), dtype=torch.int64, device=cuda_device)
    num = torch.zeros(1, dtype=torch.int32, device=cuda_device)
    ws = torch.zeros(1 << 20, dtype=torch.uint8, device=cuda_device)
    rc = L.pvx_nms_select_batched(_lib.ptr(b), 7, _lib.ptr(s), None, 1, 16385, float("-inf"), 0, 0.5, 0, 10, _lib.ptr(sel), _lib.ptr(num),
                                  _lib.ptr(ws), ws.numel(), None)
    assert rc == -4                                   # PVX_ERR_RANGE: more keys than one CU's LDS sorts
    rc = L.pvx_nms_select_batched(_lib.ptr(b), 7, _lib.ptr(s), None, 1, 4096, float("-inf"), 0, 0.5, 0, 10, _lib.ptr(sel), _lib.ptr(num),
                                  _lib.ptr(ws), ws.numel(), None)
    assert rc == -2                                   # PVX_ERR_WORKSPACE: the mask alone needs 2 MiB
    rc = L.pvx_nms_select_batched(_lib.ptr(b), 6, _lib.ptr(s), None, 1, 64, float("-inf"), 0, 0.5, 0, 10, _lib.ptr(sel), _lib.ptr(num),
                                  _lib.ptr(ws), ws.numel(), None)
    assert rc == -1                                   # PVX_ERR_INVALID: rows narrower than a box


def test_nms_above_65536_boxes(cuda_device):
    """the reference has no 65536-box limit.  70000 boxes = 35000 well-separated boxes + an exact lower-scored copy of each:
    the kept set is known without an O(n^2) CPU check -- every original, in score order, and no copy."""
    rng = np.random.default_rng(41)
    m = 35000
    gx, gy = np.meshgrid(np.arange(200), np.arange(175))
    centers = np.stack([gx.ravel()[:m] * 6.0, gy.ravel()[:m] * 6.0], 1)
    base = np.c_[centers, rng.uniform(-1, 1, m), rng.uniform(1.5, 4.0, (m, 2)), rng.uniform(1, 2, m), rng.uniform(-3, 3, m)].astype(np.float32)
    boxes = np.concatenate([base, base])
    scores = np.concatenate([rng.permutation(m) + m, rng.permutation(m)]).astype(np.float32)     # every original outranks every copy
    sel, _ = U.nms_gpu(torch.from_numpy(boxes).to(cuda_device), torch.from_numpy(scores).to(cuda_device), 0.5)
    sel = sel.cpu().numpy()
    assert sel.size == m and sel.max() < m
    assert np.array_equal(sel, np.argsort(-scores[:m], kind="stable"))
