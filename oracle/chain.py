"""TEST INFRASTRUCTURE ONLY: the CPU oracle chained end to end -- voxelize -> mean VFE -> the sparse backbone's convs
(+ eval BatchNorm, ReLU, residual adds) -> dense(), and the greedy NMS of the detection heads -- on liboracle.so.
Used by tests/ as the checker and by bench.py's `cpu_baseline` leg as the timed CPU port; never by the product path."""
import time

import numpy as np

from . import loader


def _bn_affine(m):
    s = (m.weight / (m.running_var + m.eps).sqrt()).detach().float().cpu().numpy()
    t = (m.bias - m.running_mean * (m.weight / (m.running_var + m.eps).sqrt())).detach().float().cpu().numpy()
    return s, t


def backbone_chain(net, frames, cfg, timings=None):
    """net: a VoxelBackBone8x / VoxelResBackBone8x module tree (weights are read off it, nothing of it is executed).
    frames: list of [n, C] float32 arrays.  -> (per-stage [(coords, feats)], dense [B, C, D, H, W])."""
    t0 = time.perf_counter()
    feats, coords = [], []
    for b, pts in enumerate(frames):
        vox, czyx, num, _ = loader.voxelize_hard(pts, cfg["voxel"], cfg["range"], cfg["T"], cfg["max_voxels"])
        feats.append(loader.mean_vfe(vox, num))
        coords.append(np.concatenate([np.full((czyx.shape[0], 1), b, np.int32), czyx], 1))
    x, c = np.concatenate(feats), np.concatenate(coords)
    t1 = time.perf_counter()
    shape, B = list(net.sparse_shape), len(frames)
    books = {}
    t_rule = [0.0]

    def bn(f, m):
        s, t = _bn_affine(m)
        return f * s + t

    def conv(f, cv):
        nonlocal c, shape
        w = cv.weight.detach().float().cpu().numpy()
        b = cv.bias.detach().float().cpu().numpy() if cv.bias is not None else None
        ta = time.perf_counter()
        if cv.subm:
            key = tuple(shape)
            if key not in books:   # one rulebook per site set (the reference builds 'subm1' and 'res1' separately)
                books[key] = loader.subm_rulebook(c, shape, cv.kernel_size)
            nbr = books[key]
        else:
            oc, nbr, oshape = loader.sparse_rulebook(c, shape, cv.kernel_size, cv.stride, cv.padding)
            c, shape = oc, oshape
        t_rule[0] += time.perf_counter() - ta
        return loader.sparse_conv_fwd(f, w, b, nbr)

    outs = []
    seqs = [net.conv_input] + [getattr(net, n) for n in ("conv1", "conv2", "conv3", "conv4")] + [net.conv_out]
    for si, seq in enumerate(seqs):
        items = [seq] if si in (0, 5) else list(seq._modules.values())
        for it in items:
            if hasattr(it, "conv1") and hasattr(it, "bn2"):    # SparseBasicBlock
                h = np.maximum(bn(conv(x, it.conv1), it.bn1), 0)
                x = np.maximum(bn(conv(h, it.conv2), it.bn2) + x, 0)
            else:
                mods = list(it._modules.values())
                x = np.maximum(bn(conv(x, mods[0]), mods[1]), 0)
        outs.append((c.copy(), x.copy()))
    t2 = time.perf_counter()
    d = loader.dense(x, c, B, shape)
    t3 = time.perf_counter()
    if timings is not None:
        timings.update(voxelize=t1 - t0, rulebook=t_rule[0], conv=t2 - t1 - t_rule[0], dense=t3 - t2)
    return outs, d


def heads_nms(boxes, scores, counts, thresh, post_max):
    """boxes [S, cap, >=7], scores [S, cap], counts [S] -> list of kept index arrays (descending score), per segment."""
    out = []
    for s in range(boxes.shape[0]):
        n = int(counts[s])
        order = np.argsort(-scores[s, :n], kind="stable")
        keep, _ = loader.nms(boxes[s, order, :7], thresh)
        out.append(order[keep][:post_max])
    return out
