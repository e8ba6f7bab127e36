"""SparseConvTensor: the container type every spconv consumer in pcdet builds and passes around
(constructed at pcdet/models/backbones_3d/spconv_backbone.py:144-149,278-283,
spconv_backbone_voxelnext.py:158-163,179-184, spconv_backbone_2d.py:171-176).

Plain mutable attributes on purpose: callers assign and mutate `.indices` in place
(spconv_backbone_voxelnext.py:194-197), so nothing derived from the coordinates is trusted unless the
tensor's identity and version counter still match (see `_site_index`).
"""
from enum import Enum

import numpy as np
import torch

from ... import _lib


class ConvAlgo(Enum):
    """spconv.core.ConvAlgo (imported by libraries/3DSparseConvolution/tool/centerpoint-export/exptool.py:27-29);
    accepted and ignored: there is one algorithm here, the output-stationary implicit GEMM."""
    Native = 0
    MaskImplicitGemm = 1
    MaskSplitImplicitGemm = 2


class IndiceData:
    """What one `indice_key` caches: the rulebook of a conv plus what its inverse needs."""

    def __init__(self, nbr, n_in, n_out, out_indices, in_indices, out_spatial_shape, in_spatial_shape, ksize, stride,
                 padding, dilation, subm, out_index_cache=None):
        self.nbr = nbr                      # [K, n_out] int32 (device)
        self.n_in, self.n_out = n_in, n_out
        self.out_indices, self.in_indices = out_indices, in_indices
        self.out_spatial_shape, self.in_spatial_shape = out_spatial_shape, in_spatial_shape
        self.ksize, self.stride, self.padding, self.dilation = ksize, stride, padding, dilation
        self.subm = subm
        self.out_index_cache = out_index_cache
        self.nbr_inv = None                 # built lazily by SparseInverseConv

    # names spconv exposes on its own IndiceData, for code that pokes at them
    @property
    def indice_pairs(self):
        return self.nbr

    @property
    def is_subm(self):
        return self.subm


class _SiteIndexCache:
    def __init__(self, indices, index, perm, n_unique):
        self.ref = indices
        self.version = indices._version
        self.ptr = indices.data_ptr()
        self.rows = indices.shape[0]
        self.index, self.perm, self.n_unique = index, perm, n_unique

    def valid_for(self, indices):
        return (self.ref is indices and self.version == indices._version and self.ptr == indices.data_ptr()
                and self.rows == indices.shape[0])


def _shape_list(spatial_shape):
    if isinstance(spatial_shape, torch.Tensor):
        spatial_shape = spatial_shape.tolist()
    return [int(x) for x in np.asarray(spatial_shape).reshape(-1).tolist()]


class _DenseFunction(torch.autograd.Function):
    """features [N, C] at coords [N, 4] (b, z, y, x) -> dense [B, C, D, H, W] through pvx_sparse_to_dense; backward = the rows
    of the dense gradient at the same cells (what autograd derives for the reference's indexed assignment)."""

    @staticmethod
    def forward(ctx, feats, coords, batch_size, shape3, sorted_index=None):
        f = feats.contiguous()
        L = _lib.lib()
        dev = f.device
        c = f.shape[1]
        out = torch.empty((batch_size, c, *shape3), dtype=f.dtype, device=dev)
        with torch.cuda.device(dev):
            rc = -4
            if sorted_index is not None and f.dtype in (torch.bfloat16, torch.float16) and f.shape[0] > 0:
                # rows in key order with their site index at hand (the output of a strided conv): one pass, no zero-fill;
                # PVX_ERR_RANGE (-4) = a shape the line form does not take
                rc = L.pvx_sparse_to_dense_sorted(_lib.ptr(f), _lib.ptr(sorted_index), f.shape[0], None, c, _lib.DTYPE_CODE[f.dtype],
                                                  batch_size, _lib.int3(list(shape3)), _lib.ptr(out), _lib.stream())
                if rc != -4:
                    _lib.check(rc, "pvx_sparse_to_dense_sorted")
            if rc == -4:
                _lib.check(L.pvx_sparse_to_dense(_lib.ptr(f), _lib.ptr(coords), f.shape[0], None, c, _lib.DTYPE_CODE[f.dtype],
                                                 batch_size, _lib.int3(list(shape3)), _lib.ptr(out), _lib.stream()),
                           "pvx_sparse_to_dense")
        ctx.save_for_backward(coords)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (coords,) = ctx.saved_tensors
        ix = coords.long()
        g = grad_out[ix[:, 0], :, ix[:, 1], ix[:, 2], ix[:, 3]]
        return g.contiguous(), None, None, None, None


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None, voxel_num=None, indice_dict=None,
                 benchmark=False, permanent_thrust_allocator=False, enable_timer=False, force_algo=None):
        """features [N, C]; indices [N, 1 + ndim] int32 (b, z, y, x) or (b, y, x); spatial_shape list / ndarray
        (an np.int64 array arrives from `grid_size[::-1] + [1, 0, 0]`, spconv_backbone.py:194)."""
        self._features = features
        self.indices = indices
        self.spatial_shape = _shape_list(spatial_shape)
        self.batch_size = int(batch_size)
        self.indice_dict = indice_dict if indice_dict is not None else {}
        self.grid = grid
        self.voxel_num = voxel_num
        self.benchmark = benchmark
        self.benchmark_record = {}
        self.thrust_allocator = None
        self.force_algo = force_algo
        self.int8_scale = None
        self._index_holder = [None]   # shared by replace_feature() copies: one site index per coordinate set
        assert features.dim() == 2 and indices.dim() == 2, "features must be [N, C], indices [N, 1 + ndim]"
        assert indices.shape[1] == len(self.spatial_shape) + 1, "indices must have 1 + ndim columns"
        assert indices.dtype == torch.int32, "indices must be int32 (pass coords.int())"
        assert features.shape[0] == indices.shape[0]

    # ---- features -------------------------------------------------------------------------------------
    @property
    def features(self):
        return self._features

    @features.setter
    def features(self, value):   # spconv 1.x style assignment still works (spconv_utils.py:36-38)
        self._features = value

    def replace_feature(self, feature):
        """New tensor object sharing indices, rulebook cache and site index (spconv 2.x API, spconv_utils.py:32-35)."""
        new = SparseConvTensor.__new__(SparseConvTensor)
        new.__dict__.update(self.__dict__)
        new._features = feature
        return new

    def shadow_copy(self):
        return self.replace_feature(self._features)

    @property
    def ndim(self):
        return len(self.spatial_shape)

    @property
    def spatial_size(self):
        return int(np.prod(self.spatial_shape))

    @property
    def is_quantized(self):
        return False

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key)

    @classmethod
    def from_dense(cls, x):
        """x [B, *spatial, C] channels-last dense tensor -> sparse tensor of its non-zero sites."""
        mask = (x != 0).any(-1)
        idx = mask.nonzero().int().contiguous()
        feats = x[mask]
        return cls(feats, idx, list(x.shape[1:-1]), x.shape[0])

    # ---- coordinates as the kernels want them: [N, 4] (b, z, y, x), 2-D tensors get z = 0 ---------------
    def _coords4(self):
        idx = self.indices
        if idx.shape[1] == 4:
            return idx if idx.is_contiguous() else idx.contiguous()
        assert idx.shape[1] == 3, "only 2-D and 3-D sparse tensors are supported"
        out = torch.zeros((idx.shape[0], 4), dtype=torch.int32, device=idx.device)
        out[:, 0] = idx[:, 0]
        out[:, 2:] = idx[:, 1:]
        return out

    def _shape3(self):
        s = self.spatial_shape
        return [1] + list(s) if len(s) == 2 else list(s)

    def _site_index(self):
        """(index, perm) of the current coordinates; rebuilt whenever `.indices` was replaced or written to."""
        c = self._index_holder[0]
        if c is not None and c.valid_for(self.indices):
            return c.index, c.perm
        L = _lib.lib()
        dev = self.indices.device
        shape = _lib.int3(self._shape3())
        nbytes = L.pvx_site_index_bytes(self.batch_size, shape)
        if nbytes == 0:
            raise ValueError("spatial_shape %s x batch %d is outside what the site index supports" %
                             (self.spatial_shape, self.batch_size))
        coords = self._coords4()
        n = coords.shape[0]
        with torch.cuda.device(dev):
            index = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            perm = torch.full((max(n, 1),), -1, dtype=torch.int32, device=dev)
            n_unique = torch.zeros(1, dtype=torch.int32, device=dev)
            ws = _lib.workspace(L.pvx_site_index_workspace_bytes(self.batch_size, shape), dev)
            _lib.check(L.pvx_site_index_build(_lib.ptr(coords), n, None, self.batch_size, shape, _lib.ptr(index),
                                              _lib.ptr(perm), _lib.ptr(n_unique), _lib.ptr(ws), ws.numel(),
                                              _lib.stream()), "pvx_site_index_build")
        self._index_holder[0] = _SiteIndexCache(self.indices, index, perm, n_unique)
        return index, perm

    def _seed_site_index(self, index, perm, n_unique):
        self._index_holder[0] = _SiteIndexCache(self.indices, index, perm, n_unique)

    # ---- dense() --------------------------------------------------------------------------------------
    def dense(self, channels_first=True):
        """[B, C, *spatial] (height_compression.py:22); channels_first=False gives [B, *spatial, C].  Differentiable with
        respect to the features, as spconv's: the backward is the gather of the dense gradient at the active sites."""
        f = self._features
        if not f.is_cuda:
            raise RuntimeError("SparseConvTensor.dense() runs on the GPU only (no CPU fallback)")
        shape3 = self._shape3()
        cache = self._index_holder[0]
        sorted_index = None
        if cache is not None and cache.valid_for(self.indices) and cache.perm is None and len(self.spatial_shape) == 3:
            sorted_index = cache.index          # seeded by a strided conv: row i is the site of rank i
        out = _DenseFunction.apply(f, self._coords4(), self.batch_size, tuple(shape3), sorted_index)
        if len(self.spatial_shape) == 2:
            out = out[:, :, 0]
        if not channels_first:
            out = out.permute(0, *range(2, out.dim()), 1).contiguous()
        return out
