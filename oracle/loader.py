"""TEST INFRASTRUCTURE ONLY: ctypes access to oracle/liboracle.so (our C restatements) and, when it has been
built in the container that holds /root/reference, oracle/_ref/libiou3d_ref.so (the reference's own CPU IoU)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libiou3d_ref.so")

_fp = ctypes.POINTER(ctypes.c_float)
_ip = ctypes.POINTER(ctypes.c_int)
_lp = ctypes.POINTER(ctypes.c_int64)

_orc = None
_ref = None


def build(with_ref=True):
    """(Re)build the oracle library; the reference library only where /root/reference exists."""
    subprocess.run(["make", "-C", HERE, "liboracle.so"], check=True, capture_output=True)
    if with_ref and os.path.isdir("/root/reference"):
        subprocess.run(["make", "-C", HERE, "ref"], check=True, capture_output=True)


def oracle():
    global _orc
    if _orc is None:
        if not os.path.exists(ORACLE_SO):
            build(with_ref=False)
        _orc = ctypes.CDLL(ORACLE_SO)
        _orc.orc_box_overlap.restype = ctypes.c_float
        _orc.orc_iou_bev.restype = ctypes.c_float
        _orc.orc_iou_normal.restype = ctypes.c_float
    return _orc


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    """The reference's own iou3d_cpu.cpp (needs torch's shared libraries in the process)."""
    global _ref
    if _ref is None:
        import torch  # noqa: F401  (loads libc10/libtorch_cpu the .so links against)
        _ref = ctypes.CDLL(REF_SO)
    return _ref


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def fptr(a):
    return a.ctypes.data_as(_fp)


# ---------------------------------------------------------------------------------------------------
# iou3d
# ---------------------------------------------------------------------------------------------------
def boxes_bev_matrix(a, b, mode=0):
    """mode 0: IoU, 1: overlap area.  (N,7),(M,7) -> (N,M) float32."""
    a, b = f32(a), f32(b)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    oracle().orc_boxes_bev_matrix(fptr(a), a.shape[0], fptr(b), b.shape[0], fptr(out), int(mode))
    return out


def boxes_bev_paired(a, b, mode=1):
    a, b = f32(a), f32(b)
    out = np.zeros((a.shape[0],), np.float32)
    oracle().orc_boxes_bev_paired(fptr(a), fptr(b), a.shape[0], fptr(out), int(mode))
    return out


def nms(boxes_sorted, thresh, normal=False):
    """Greedy NMS on score-sorted boxes -> (keep int64 array, decision margin)."""
    b = f32(boxes_sorted)
    n = b.shape[0]
    keep = np.zeros((max(n, 1),), np.int64)
    margin = ctypes.c_double(0)
    k = oracle().orc_nms(fptr(b), n, ctypes.c_float(thresh), int(bool(normal)), keep.ctypes.data_as(_lp),
                         ctypes.byref(margin))
    return keep[:k].copy(), margin.value


def ref_boxes_iou_bev(a, b):
    a, b = f32(a), f32(b)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    ref().ref_boxes_iou_bev_cpu(fptr(a), a.shape[0], fptr(b), b.shape[0], fptr(out))
    return out


def ref_boxes_aligned_iou_bev(a, b):
    a, b = f32(a), f32(b)
    out = np.zeros((a.shape[0], 1), np.float32)
    ref().ref_boxes_aligned_iou_bev_cpu(fptr(a), fptr(b), a.shape[0], fptr(out))
    return out[:, 0]


# ---------------------------------------------------------------------------------------------------
# voxelization
# ---------------------------------------------------------------------------------------------------
def voxel_grid_size(vsize_xyz, range_xyz):
    v, r = f32(vsize_xyz), f32(range_xyz)
    g = np.zeros(3, np.int32)
    oracle().orc_voxel_grid_size(fptr(v), fptr(r), g.ctypes.data_as(_ip))
    return g.tolist()


def voxelize_hard(points, vsize_xyz, range_xyz, T, max_voxels):
    """-> voxels [M,T,C], coords_zyx [M,3], num_per_voxel [M], point_to_voxel [N]"""
    p = f32(points)
    n, c = p.shape
    v, r = f32(vsize_xyz), f32(range_xyz)
    voxels = np.zeros((max_voxels, T, c), np.float32)
    coords = np.zeros((max_voxels, 3), np.int32)
    num = np.zeros((max_voxels,), np.int32)
    p2v = np.zeros((max(n, 1),), np.int32)
    m = oracle().orc_voxelize_hard(fptr(p), n, c, fptr(v), fptr(r), int(T), int(max_voxels), fptr(voxels),
                                   coords.ctypes.data_as(_ip), num.ctypes.data_as(_ip), p2v.ctypes.data_as(_ip))
    return voxels[:m].copy(), coords[:m].copy(), num[:m].copy(), p2v[:n].copy()


def mean_vfe(voxels, num):
    vx = f32(voxels)
    m, T, c = vx.shape
    nm = np.ascontiguousarray(num, np.int32)
    out = np.zeros((m, c), np.float32)
    oracle().orc_mean_vfe(fptr(vx), nm.ctypes.data_as(_ip), m, T, c, fptr(out))
    return out


# ---------------------------------------------------------------------------------------------------
# sparse convolution
# ---------------------------------------------------------------------------------------------------
def _i3(v):
    if isinstance(v, int):
        v = (v, v, v)
    return (ctypes.c_int * 3)(*[int(x) for x in v])


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def conv_out_shape(in_shape, ksize, stride, padding, dilation=1):
    out = (ctypes.c_int * 3)()
    oracle().orc_conv_out_shape(_i3(in_shape), _i3(ksize), _i3(stride), _i3(padding), _i3(dilation), out)
    return [out[0], out[1], out[2]]


def subm_rulebook(coords, shape, ksize, dilation=1):
    """coords [n,4] (b,z,y,x) -> nbr [K,n] int32"""
    c = _i32(coords)
    n = c.shape[0]
    ks = _i3(ksize)
    K = ks[0] * ks[1] * ks[2]
    nbr = np.full((K, max(n, 1)), -1, np.int32)
    oracle().orc_subm_rulebook(c.ctypes.data_as(_ip), n, _i3(shape), ks, _i3(dilation), nbr.ctypes.data_as(_ip))
    return nbr[:, :n].copy() if n else np.zeros((K, 0), np.int32)


def sparse_rulebook(coords, in_shape, ksize, stride, padding, dilation=1):
    """strided conv -> (out_coords [n_out,4] ascending key, nbr [K,n_out], out_shape)"""
    c = _i32(coords)
    n = c.shape[0]
    ks = _i3(ksize)
    K = ks[0] * ks[1] * ks[2]
    args = (c.ctypes.data_as(_ip), n, _i3(in_shape), ks, _i3(stride), _i3(padding), _i3(dilation))
    n_out = oracle().orc_sparse_rulebook(*args, None, None)
    oc = np.zeros((max(n_out, 1), 4), np.int32)
    nbr = np.full((K, max(n_out, 1)), -1, np.int32)
    got = oracle().orc_sparse_rulebook(*args, oc.ctypes.data_as(_ip), nbr.ctypes.data_as(_ip))
    assert got == n_out
    return oc[:n_out].copy(), np.ascontiguousarray(nbr[:, :n_out]), conv_out_shape(in_shape, ksize, stride, padding, dilation)


def inverse_rulebook(nbr_fwd, n_in_fwd):
    f = _i32(nbr_fwd)
    K, n_out = f.shape
    inv = np.full((K, max(n_in_fwd, 1)), -1, np.int32)
    oracle().orc_inverse_rulebook(f.ctypes.data_as(_ip), K, n_out, n_in_fwd, inv.ctypes.data_as(_ip))
    return np.ascontiguousarray(inv[:, :n_in_fwd])


def sparse_conv_fwd(feat, weight, bias, nbr):
    """feat [n_in,cin] f32, weight [cout,K,cin] (or [cout,kz,ky,kx,cin]) f32, nbr [K,n_out] -> [n_out,cout] f32"""
    f = f32(feat)
    w = f32(weight)
    cout, cin = w.shape[0], w.shape[-1]
    w = w.reshape(cout, -1, cin)
    K = w.shape[1]
    nb = _i32(nbr)
    assert nb.shape[0] == K
    n_out = nb.shape[1]
    out = np.zeros((max(n_out, 1), cout), np.float32)
    b = f32(bias) if bias is not None else None
    oracle().orc_sparse_conv_fwd(fptr(f), cin, fptr(w), fptr(b) if b is not None else None, nb.ctypes.data_as(_ip), K, n_out,
                                 cout, fptr(out))
    return out[:n_out].copy()


def count_pairs(nbr):
    nb = _i32(nbr)
    oracle().orc_count_pairs.restype = ctypes.c_longlong
    return int(oracle().orc_count_pairs(nb.ctypes.data_as(_ip), nb.shape[0], nb.shape[1]))


def dense(feat, coords, batch, shape):
    f = f32(feat)
    c = _i32(coords)
    n, ch = f.shape
    out = np.zeros((batch, ch, shape[0], shape[1], shape[2]), np.float32)
    oracle().orc_dense(fptr(f), c.ctypes.data_as(_ip), n, ch, batch, _i3(shape), fptr(out))
    return out


# ---- VectorPool (vector_pool_oracle.c) -------------------------------------------------------------------------------------
def vp_three_nn(support_xyz, xyz_cnt, new_xyz, centers, new_cnt, max_dist, nsample, neighbor_type):
    """-> dist2 [M, G, 3] f32, idx [M, G, 3] i32, summed neighbour-list lengths"""
    s, q, c = f32(support_xyz), f32(new_xyz), f32(centers)
    xc, nc = _i32(xyz_cnt), _i32(new_cnt)
    M, G = c.shape[0], c.shape[1]
    d2 = np.zeros((M, G, 3), np.float32)
    idx = np.zeros((M, G, 3), np.int32)
    fn = oracle().orc_vp_three_nn
    fn.restype = ctypes.c_longlong
    total = fn(fptr(s), xc.ctypes.data_as(_ip), fptr(q), fptr(c), nc.ctypes.data_as(_ip), len(xc), M, G, ctypes.c_float(max_dist),
               int(nsample), int(neighbor_type), fptr(d2), idx.ctypes.data_as(_ip))
    return d2, idx, int(total)


def vector_pool(support_xyz, support_feat, xyz_cnt, new_xyz, new_cnt, ceach, grids, max_dist, use_xyz, nsample, neighbor_type, pooling, cap):
    """raw sums (not yet divided by the counts) -> new_features [M, G*ceach], new_local_xyz [M, G*3], cnt [M, G], grouped [min(cum, cap), 3], cum"""
    s, f, q = f32(support_xyz), f32(support_feat), f32(new_xyz)
    xc, nc = _i32(xyz_cnt), _i32(new_cnt)
    M, cin = q.shape[0], f.shape[1]
    G = grids[0] * grids[1] * grids[2]
    nf = np.zeros((M, G * ceach), np.float32)
    nl = np.zeros((M, G * 3), np.float32)
    cnt = np.zeros((M, G), np.int32)
    grouped = np.zeros((max(cap, 1), 3), np.int32)
    fn = oracle().orc_vector_pool
    fn.restype = ctypes.c_longlong
    cum = int(fn(fptr(s), fptr(f), xc.ctypes.data_as(_ip), fptr(q), nc.ctypes.data_as(_ip), len(xc), M, cin, int(ceach), int(grids[0]),
                 int(grids[1]), int(grids[2]), ctypes.c_float(max_dist), int(use_xyz), int(nsample), int(neighbor_type), int(pooling),
                 fptr(nf), fptr(nl), cnt.ctypes.data_as(_ip), grouped.ctypes.data_as(_ip), ctypes.c_longlong(cap)))
    return nf, nl, cnt, grouped[:min(cum, cap)].copy(), cum


def vector_pool_grad(grad_new, cnt, grouped, n_support, cin, ceach):
    g = f32(grad_new)
    c, gr = _i32(cnt), _i32(grouped)
    out = np.zeros((n_support, cin), np.float32)
    oracle().orc_vector_pool_grad(fptr(g), c.ctypes.data_as(_ip), gr.ctypes.data_as(_ip), ctypes.c_longlong(len(gr)), int(cin), int(ceach),
                                  int(c.shape[1]), fptr(out))
    return out
