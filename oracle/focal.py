"""TEST INFRASTRUCTURE ONLY: CPU (numpy) restatement of the focal sparse convolution's site-set surgery,
pcdet/models/backbones_3d/focal_sparse_conv/focal_sparse_utils.py:39-147 and focal_sparse_conv.py:115-224 (lidar-only, eval),
with the convolutions done by liboracle.so.  Parity unpinned: the reference ships no fixture for this layer and its own
modules cannot be imported here (spconv is absent); this follows the source text line by line.
Unspecified choices of the reference (unstable sort ties, which duplicate's coordinates survive) are fixed as: lower row
first, last row in merge order."""
import numpy as np

from . import loader


def merge_key(idx):
    """:48,71 -- z * max(y) * max(x) + y * max(x) + x over THIS subset's maxima"""
    idx = idx.astype(np.int64)
    m1, m2 = idx[:, 1].max(), idx[:, 2].max()
    return idx[:, 0] * m1 * m2 + idx[:, 1] * m2 + idx[:, 2]


def check_repeat(features, indices, features_add=None, sort_first=True, flip_first=True):
    """:56-86"""
    if len(indices) == 0:
        return features, indices, features_add
    if sort_first:                                                           # :39-54
        ind = np.argsort(merge_key(indices[:, 1:]), kind="stable")
        features, indices = features[ind], indices[ind]
        if features_add is not None:
            features_add = features_add[ind]
    if flip_first:
        features, indices = features[::-1], indices[::-1]
    if features_add is not None:
        features_add = features_add[::-1]
    key = merge_key(indices[:, 1:])
    start = np.r_[True, key[1:] != key[:-1]]                                 # unique_consecutive
    if start.sum() < len(key):
        inverse = np.cumsum(start) - 1
        n = int(start.sum())
        merged = np.zeros((n, features.shape[1]), np.float32)
        np.add.at(merged, inverse, features)
        last = np.zeros(n, np.int64)
        last[inverse] = np.arange(len(key))                                  # sequential scatter: the last row of each run stays
        features, indices = merged, indices[last]
        if features_add is not None:
            s = np.zeros(n, np.float32)
            np.add.at(s, inverse, features_add)
            features_add = s / np.bincount(inverse, minlength=n).astype(np.float32)
    return features, indices, features_add


def sigmoid(v):
    return (1.0 / (1.0 + np.exp(-v.astype(np.float64)))).astype(np.float32)


def split_voxels(feats, coords, shape, b, imps, offsets, mask_multi, topk, threshold):
    """:89-147 -> fore feats / coords, back feats / coords, kernel mask, decision margin"""
    rows = coords[:, 0] == b
    c0, f0 = coords[rows], feats[rows].copy()
    mv, mk_all = sigmoid(imps[rows, -1]), sigmoid(imps[rows, :-1])
    if mask_multi:
        f0 = f0 * mv[:, None]
    if topk:
        order = np.argsort(-mv, kind="stable")
        cut = int(len(mv) * threshold)
        fore, back = order[:cut], order[cut:]
        margin = float(mv[order[cut - 1]] - mv[order[cut]]) if 0 < cut < len(mv) else 1.0      # the gap at the cut decides
    else:
        fore, back = mv > threshold, mv <= threshold
        live = imps[rows, -1] != 0                  # an importance of exactly 0 (all-zero receptive field) is exact on any device
        margin = float(np.min(np.abs(mv[live] - threshold))) if live.any() else 1.0
    ff, fc = f0[fore], c0[fore]
    mk = mk_all[fore]
    chosen = mk >= threshold
    live = imps[rows, :-1][fore] != 0
    if live.any():
        margin = min(margin, float(np.min(np.abs(mk[live] - threshold))))
    cells = (fc[:, None, 1:] + offsets[None, :, :])[chosen]
    mk = mk[chosen]
    inside = (cells > 0).all(1) & (cells[:, 0] < shape[0]) & (cells[:, 1] < shape[1]) & (cells[:, 2] < shape[2])     # :130-131
    cells, mk = cells[inside], mk[inside]
    cells = np.concatenate([np.full((len(cells), 1), b, cells.dtype), cells], 1)
    f = np.concatenate([ff, np.zeros((len(cells), f0.shape[1]), np.float32)])
    c = np.concatenate([fc, cells])
    m = np.concatenate([np.ones(len(ff), np.float32), mk])
    f, c, m = check_repeat(f, c, m)
    return f, c, f0[back], c0[back], m, margin


def focal_conv_sites(feats, coords, shape, batch, imps, ksize=3, mask_multi=False, topk=False, threshold=0.5, skip_mask_kernel=False):
    """focal_sparse_conv.py:134-160,171-197,213-215: -> (features, coords) the layer's own conv runs on, decision margin"""
    s = ksize // 2
    offsets = np.array([[i, j, k] for i in range(-s, s + 1) for j in range(-s, s + 1) for k in range(-s, s + 1)
                        if (i, j, k) != (0, 0, 0)], coords.dtype)
    F, C, margins = [], [], []
    fore_f, fore_c, back_f, back_c, masks = [], [], [], [], []
    for b in range(batch):
        ff, fc, bf, bc, m, margin = split_voxels(feats, coords, shape, b, imps, offsets, mask_multi, topk, threshold)
        fore_f.append(ff); fore_c.append(fc); back_f.append(bf); back_c.append(bc); masks.append(m); margins.append(margin)
    ff, fc, mk = np.concatenate(fore_f), np.concatenate(fore_c), np.concatenate(masks)
    if not skip_mask_kernel:
        ff = ff * mk[:, None]
    f = np.concatenate([ff, np.concatenate(back_f)])
    c = np.concatenate([fc, np.concatenate(back_c)])
    for b in range(batch):
        rows = c[:, 0] == b
        fb, cb, _ = check_repeat(f[rows], c[rows], flip_first=False)
        F.append(fb); C.append(cb)
    return np.concatenate(F).astype(np.float32), np.concatenate(C).astype(np.int32), min(margins)


def bn_eval(f, m):
    s = (m.weight / (m.running_var + m.eps).sqrt()).detach().numpy()
    return f * s + (m.bias.detach().numpy() - m.running_mean.detach().numpy() * s)


def conv_w(cv):
    return cv.weight.detach().numpy().reshape(cv.out_channels, -1, cv.in_channels)


def focal_layer(layer, feats, coords, shape, batch):
    """whole FocalSparseConv.forward (eval) from a CPU copy of the module -> (features, coords, decision margin)"""
    k3 = (3, 3, 3)
    book = loader.subm_rulebook(coords, shape, k3)
    xp = feats
    if layer.conv_enlarge is not None:
        xp = np.maximum(bn_eval(loader.sparse_conv_fwd(feats, conv_w(layer.conv_enlarge[0]), None, book), layer.conv_enlarge[1]), 0)
    imps = loader.sparse_conv_fwd(xp, conv_w(layer.conv_imp), None, book)
    ks = int(layer.conv.kernel_size[0])
    f, c, margin = focal_conv_sites(feats, coords, shape, batch, imps, ksize=ks, mask_multi=layer.mask_multi, topk=layer.topk,
                                    threshold=layer.threshold, skip_mask_kernel=layer.skip_mask_kernel)
    out = loader.sparse_conv_fwd(f, conv_w(layer.conv), None, loader.subm_rulebook(c, shape, (ks, ks, ks)))
    return np.maximum(bn_eval(out, layer.bn1), 0), c, margin
