"""Device voxelizer: the host-side mirror of the reference's voxelizer interfaces over pvx_voxelize_hard.

* ``PointToVoxel``          -- spconv 2.x torch-tensor voxelizer signature [spconv] (named by BASELINE.json config 1)
* ``Point2VoxelCPU3d``      -- the class the reference instantiates (data_processor.py:26-43) with the same ctor
                               keywords and ``point_to_voxel(tv_tensor)`` -> 3 tensors with ``.numpy()``
* ``VoxelGeneratorWrapper`` -- same name / ``generate(points)`` contract as data_processor.py:16-61
* ``mean_vfe``              -- MeanVFE arithmetic on an existing padded voxel tensor
"""
import ctypes

import numpy as np
import torch

from . import _lib


def grid_size(vsize_xyz, coors_range_xyz):
    v = (ctypes.c_float * 3)(*[float(x) for x in vsize_xyz])
    r = (ctypes.c_float * 6)(*[float(x) for x in coors_range_xyz])
    g = (ctypes.c_int * 3)()
    _lib.check(_lib.lib().pvx_voxel_grid_size(v, r, g), "pvx_voxel_grid_size")
    return [g[0], g[1], g[2]]


class PointToVoxel:
    """voxels, zyx indices and per-voxel counts of one frame, computed on the GPU the points live on."""

    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_voxels, max_num_points_per_voxel,
                 device=None):
        self.vsize = [float(x) for x in vsize_xyz]
        self.range = [float(x) for x in coors_range_xyz]
        self.num_point_features = int(num_point_features)
        self.max_num_voxels = int(max_num_voxels)
        self.max_num_points_per_voxel = int(max_num_points_per_voxel)
        self.device = device
        self.grid_size = grid_size(self.vsize, self.range)
        self._v = (ctypes.c_float * 3)(*self.vsize)
        self._r = (ctypes.c_float * 6)(*self.range)

    def run(self, points, want_voxels=True, want_mean=False, want_point_map=False):
        """points [N, C] float32 on the GPU -> dict of padded-capacity device tensors + 'num_voxels' (device int32[1]).
        No host synchronisation; rows >= num_voxels are undefined."""
        assert points.is_cuda and points.dtype == torch.float32 and points.dim() == 2
        assert points.shape[1] == self.num_point_features
        points = points.contiguous()
        n, c = points.shape
        dev = points.device
        M, T = self.max_num_voxels, self.max_num_points_per_voxel
        L = _lib.lib()
        out = {
            "coords": torch.empty((M, 3), dtype=torch.int32, device=dev),
            "num_points": torch.empty((M,), dtype=torch.int32, device=dev),
            "num_voxels": torch.zeros((1,), dtype=torch.int32, device=dev),
            "voxels": torch.empty((M, T, c), dtype=torch.float32, device=dev) if want_voxels else None,
            "mean": torch.empty((M, c), dtype=torch.float32, device=dev) if want_mean else None,
            "point_to_voxel": torch.empty((n,), dtype=torch.int32, device=dev) if want_point_map else None,
        }
        with torch.cuda.device(dev):
            ws = _lib.workspace(L.pvx_voxelize_hard_workspace_bytes(n, M, T), dev)
            _lib.check(L.pvx_voxelize_hard(_lib.ptr(points), n, c, self._v, self._r, T, M, _lib.ptr(out["voxels"]),
                                           _lib.ptr(out["coords"]), _lib.ptr(out["num_points"]), _lib.ptr(out["mean"]),
                                           _lib.ptr(out["point_to_voxel"]), _lib.ptr(out["num_voxels"]), _lib.ptr(ws),
                                           ws.numel(), _lib.stream()), "pvx_voxelize_hard")
        return out

    def run_batch(self, clouds, want_point_map=False):
        """clouds [F, N, C] float32 on the GPU (shorter clouds padded with NaN rows) -> dict of [F, ...] padded-capacity
        tensors + 'num_voxels' (device int32[F]): F independent voxelizations in one set of launches
        (pvx_voxelize_hard_batch; the DOUBLE_FLIP path of data_processor.py:225-242).  No host synchronisation."""
        assert clouds.is_cuda and clouds.dtype == torch.float32 and clouds.dim() == 3
        assert clouds.shape[2] == self.num_point_features
        clouds = clouds.contiguous()
        F, n, c = clouds.shape
        dev = clouds.device
        M, T = self.max_num_voxels, self.max_num_points_per_voxel
        L = _lib.lib()
        out = {
            "coords": torch.empty((F, M, 3), dtype=torch.int32, device=dev),
            "num_points": torch.empty((F, M), dtype=torch.int32, device=dev),
            "num_voxels": torch.zeros((F,), dtype=torch.int32, device=dev),
            "voxels": torch.empty((F, M, T, c), dtype=torch.float32, device=dev),
            "point_to_voxel": torch.empty((F, n), dtype=torch.int32, device=dev) if want_point_map else None,
        }
        with torch.cuda.device(dev):
            ws = _lib.workspace(L.pvx_voxelize_batch_workspace_bytes(F, n, M, T), dev)
            _lib.check(L.pvx_voxelize_hard_batch(_lib.ptr(clouds), F, n, c, self._v, self._r, T, M, _lib.ptr(out["voxels"]),
                                                 _lib.ptr(out["coords"]), _lib.ptr(out["num_points"]),
                                                 _lib.ptr(out["point_to_voxel"]), _lib.ptr(out["num_voxels"]), _lib.ptr(ws),
                                                 ws.numel(), _lib.stream()), "pvx_voxelize_hard_batch")
        return out

    def __call__(self, points):
        """spconv-style result: (voxels [M,T,C], indices [M,3] zyx, num_per_voxel [M]) sliced to the true M
        (one host read of M, like the reference's `[:voxel_num]` views)."""
        o = self.run(points, want_voxels=True)
        m = int(o["num_voxels"].item())
        return o["voxels"][:m], o["coords"][:m], o["num_points"][:m]

    def generate_voxel_with_id(self, points):
        o = self.run(points, want_voxels=True, want_point_map=True)
        m = int(o["num_voxels"].item())
        return o["voxels"][:m], o["coords"][:m], o["num_points"][:m], o["point_to_voxel"]


class _TvTensor:
    """Minimal cumm.tensorview.Tensor look-alike: wraps an array, `.numpy()` returns a copy (data_processor.py:57)."""

    def __init__(self, arr):
        self._arr = arr

    def numpy(self):
        a = self._arr
        return a.detach().cpu().numpy().copy() if isinstance(a, torch.Tensor) else np.array(a, copy=True)

    def torch(self):
        return self._arr


class tensorview:
    """`from cumm import tensorview as tv` stand-in: tv.from_numpy / tv.Tensor (data_processor.py:9-13, 54)."""
    Tensor = _TvTensor

    @staticmethod
    def from_numpy(arr):
        return _TvTensor(arr)


class Point2VoxelCPU3d:
    """Same constructor keywords and method as spconv.utils.Point2VoxelCPU3d as used at data_processor.py:36-43, 55;
    the work runs on the current GPU (despite the historical name) -- use it from a process that may touch the GPU."""

    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel, max_num_voxels):
        self._impl = PointToVoxel(vsize_xyz, coors_range_xyz, num_point_features, max_num_voxels, max_num_points_per_voxel)

    def point_to_voxel(self, tv_points):
        pts = tv_points.torch() if isinstance(tv_points, _TvTensor) else tv_points
        if isinstance(pts, np.ndarray):
            pts = torch.from_numpy(np.ascontiguousarray(pts, np.float32))
        v, c, n = self._impl(pts.cuda())
        return _TvTensor(v), _TvTensor(c), _TvTensor(n)


class VoxelGenerator:
    """spconv 1.0's `spconv.utils.VoxelGenerator` as data_processor.py:23-35,47,52 uses it: keyword constructor without the
    point-feature count (taken from the first cloud), `generate(points ndarray)` -> (voxels, coordinates, num_points) ndarrays.
    Not exported from the `spconv.utils` stand-in unless `install_as_spconv(v1_voxel_generators=True)` asks for it: the
    reference tries the v1 names first and would otherwise leave its 2.x branch."""

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000):
        self._args = (list(voxel_size), list(point_cloud_range), int(max_num_points), int(max_voxels))
        self._impl = None

    def _run(self, points):
        pts = torch.from_numpy(np.ascontiguousarray(points, np.float32)) if isinstance(points, np.ndarray) else points
        if pts.dim() != 2:
            raise ValueError("points must be [N, C]")
        if self._impl is None or self._impl[0] != pts.shape[1]:
            vs, rng, t, m = self._args
            self._impl = (pts.shape[1], PointToVoxel(vs, rng, pts.shape[1], m, t))
        v, c, n = self._impl[1](pts.cuda())
        return v.cpu().numpy(), c.cpu().numpy(), n.cpu().numpy()

    def generate(self, points, max_voxels=None):
        if max_voxels is not None and int(max_voxels) != self._args[3]:
            self._args = self._args[:3] + (int(max_voxels),)
            self._impl = None
        return self._run(points)


class VoxelGeneratorV2(VoxelGenerator):
    """spconv 1.2's variant: `generate` returns the dict data_processor.py:48-50 unpacks."""

    def generate(self, points, max_voxels=None):
        voxels, coordinates, num_points = super().generate(points, max_voxels)
        return {"voxels": voxels, "coordinates": coordinates, "num_points_per_voxel": num_points,
                "voxel_num": int(voxels.shape[0])}


class VoxelGeneratorWrapper:
    """data_processor.py:16-61: generate(points ndarray) -> (voxels, coordinates, num_points) ndarrays."""

    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel, max_num_voxels):
        self._voxel_generator = Point2VoxelCPU3d(vsize_xyz=vsize_xyz, coors_range_xyz=coors_range_xyz,
                                                 num_point_features=num_point_features,
                                                 max_num_points_per_voxel=max_num_points_per_voxel,
                                                 max_num_voxels=max_num_voxels)

    def generate(self, points):
        tv_voxels, tv_coordinates, tv_num_points = self._voxel_generator.point_to_voxel(tensorview.from_numpy(points))
        return tv_voxels.numpy(), tv_coordinates.numpy(), tv_num_points.numpy()


def mean_vfe(voxels, voxel_num_points):
    """voxels [M,T,C] f32, voxel_num_points [M] (any int/float dtype) -> [M,C]   (mean_vfe.py:26-31)."""
    assert voxels.is_cuda and voxels.dim() == 3
    voxels = voxels.contiguous().float()
    num = voxel_num_points.to(torch.int32).contiguous()
    m, t, c = voxels.shape
    out = torch.empty((m, c), dtype=torch.float32, device=voxels.device)
    with torch.cuda.device(voxels.device):
        _lib.check(_lib.lib().pvx_mean_vfe(_lib.ptr(voxels), _lib.ptr(num), m, t, c, _lib.ptr(out), _lib.stream()),
                   "pvx_mean_vfe")
    return out
