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
