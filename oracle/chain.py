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


def voxelnext_chain(net, x, c, batch):
    """VoxelResBackBone8xVoxelNeXt.forward (pcdet/models/backbones_3d/spconv_backbone_voxelnext.py:166-225) on the oracle:
    x [N, C] f32 voxel features, c [N, 4] i32 (b, z, y, x).  -> dict(x_conv1..6 = (coords, feats), x_conv4_cat, bev = (coords [M, 3],
    feats) after bev_out, out = (coords [M', 3], feats) after conv_out + shared_conv)."""
    shape = list(net.sparse_shape)
    books = {}
    state = {"c": c, "shape": shape}

    def bn(f, m):
        s, t = _bn_affine(m)
        return f * s + t

    def conv(f, cv):
        w = cv.weight.detach().float().cpu().numpy()
        b = cv.bias.detach().float().cpu().numpy() if cv.bias is not None else None
        k, s, p = list(cv.kernel_size), list(cv.stride), list(cv.padding)
        if len(k) == 2:       # 2-D conv = the D = 1, kz = 1 case
            k, s, p = [1] + k, [1] + s, [0] + p
            w = w.reshape(w.shape[0], 1, *w.shape[1:])
        if cv.subm:
            key = (tuple(state["shape"]), tuple(k))
            if key not in books:
                books[key] = loader.subm_rulebook(state["c"], state["shape"], k)
            nbr = books[key]
        else:
            state["c"], nbr, state["shape"] = loader.sparse_rulebook(state["c"], state["shape"], k, s, p)
        return loader.sparse_conv_fwd(f, w, b, nbr)

    def run(seq_items, x):
        for it in seq_items:
            if hasattr(it, "conv1") and hasattr(it, "bn2"):
                h = np.maximum(bn(conv(x, it.conv1), it.bn1), 0)
                x = np.maximum(bn(conv(h, it.conv2), it.bn2) + x, 0)
            else:
                mods = list(it._modules.values())
                x = np.maximum(bn(conv(x, mods[0]), mods[1]), 0)
        return x

    out = {}
    x = run([net.conv_input], x)
    for i in range(1, 7):
        x = run(list(getattr(net, "conv%d" % i)._modules.values()), x)
        out["x_conv%d" % i] = (state["c"].copy(), x.copy())
        out["shape%d" % i] = list(state["shape"])
    c4, c5, c6 = [out["x_conv%d" % i][0].copy() for i in (4, 5, 6)]
    c5[:, 1:] *= 2
    c6[:, 1:] *= 4
    cat_c = np.concatenate([c4, c5, c6])
    cat_f = np.concatenate([out["x_conv%d" % i][1] for i in (4, 5, 6)])
    out["x_conv4_cat"] = (cat_c, cat_f)
    uniq, inv = np.unique(cat_c[:, [0, 2, 3]], axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    bev = np.zeros((uniq.shape[0], cat_f.shape[1]), np.float64)
    np.add.at(bev, inv, cat_f.astype(np.float64))
    bev = bev.astype(np.float32)
    out["bev"] = (uniq.astype(np.int32), bev)
    state["shape"] = [1] + out["shape4"][1:]
    state["c"] = np.concatenate([uniq[:, :1], np.zeros_like(uniq[:, :1]), uniq[:, 1:]], 1).astype(np.int32)
    y = run([net.conv_out], bev)
    y = run([net.shared_conv], y)
    out["out"] = (state["c"][:, [0, 2, 3]].copy(), y)
    return out


def heads_nms(boxes, scores, counts, thresh, post_max):
    """boxes [S, cap, >=7], scores [S, cap], counts [S] -> list of kept index arrays (descending score), per segment."""
    out = []
    for s in range(boxes.shape[0]):
        n = int(counts[s])
        order = np.argsort(-scores[s, :n], kind="stable")
        keep, _ = loader.nms(boxes[s, order, :7], thresh)
        out.append(order[keep][:post_max])
    return out
