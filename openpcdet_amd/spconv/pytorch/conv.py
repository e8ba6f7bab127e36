"""Sparse convolution modules with spconv's names and constructor signatures (boundary B1):
SubMConv3d / SparseConv3d / SparseInverseConv3d and their 2-D forms, all deriving from ``SparseConvolution``
(the class pcdet/utils/spconv_utils.py:23 isinstance-checks and exptool.py:107-129 reads attributes from).

``weight`` is a Parameter of shape [Cout, *kernel_size, Cin] (detector3d_template.py:357-368, funcs.py:46-63).
The forward pass is three C-ABI calls: site index -> rulebook -> implicit-GEMM; rulebooks are cached per
``indice_key`` in ``SparseConvTensor.indice_dict`` exactly like spconv does.
"""
import math

import numpy as np
import torch
from torch import nn

from ... import _lib
from .core import IndiceData, SparseConvTensor
from .modules import SparseModule


def _ntuple(v, ndim):
    if isinstance(v, (int, np.integer)):
        return [int(v)] * ndim
    v = [int(x) for x in v]
    assert len(v) == ndim, "expected %d values, got %s" % (ndim, v)
    return v


def _to3(v, ndim, fill):
    return list(v) if ndim == 3 else [fill] + list(v)


class _SparseConvFunction(torch.autograd.Function):
    """Forward on the HIP implicit-GEMM kernel; backward = the same kernel over the inverse rulebook with the weight
    transposed (input gradient) + pvx_sparse_conv_wgrad (weight gradient) + a column sum (bias gradient).
    `back` = (callable returning the rulebook [K, n_in] that maps input rows to output rows, flip): for a submanifold conv
    the forward rulebook itself serves, read with the kernel offsets reversed (coords[o] = coords[i] + off(k) <=>
    coords[i] = coords[o] + off(K-1-k)), so nothing extra is built."""

    @staticmethod
    def forward(ctx, features, weight, bias, module, nbr, n_out, back):
        ctx.module, ctx.nbr, ctx.n_out, ctx.back = module, nbr, n_out, back
        ctx.has_bias = bias is not None
        ctx.save_for_backward(features, weight)
        return module._launch(features, weight, bias, nbr, n_out)

    @staticmethod
    def backward(ctx, grad_out):
        features, weight = ctx.saved_tensors
        module = ctx.module
        g = grad_out.to(features.dtype).contiguous()
        grad_feat = grad_w = grad_b = None
        if ctx.needs_input_grad[0]:
            nbr_back, flip = ctx.back
            grad_feat = module._dgrad(g, weight, nbr_back(), flip, features.shape[0])
        if ctx.needs_input_grad[1]:
            grad_w = module._wgrad(features, g, ctx.nbr, ctx.n_out).view_as(weight).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            grad_b = g.float().sum(0).to(module.bias.dtype)
        return grad_feat, grad_w, grad_b, None, None, None, None


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, subm=False, output_padding=0, transposed=False, inverse=False, indice_key=None, algo=None,
                 fp32_accum=None, record_voxel_count=False, act_type=None, act_alpha=0, act_beta=0, large_kernel_fast_algo=False,
                 name=None):
        super().__init__(name=name)
        assert ndim in (2, 3), "2-D and 3-D sparse convolutions only"
        assert groups == 1, "groups != 1 is not supported (no caller in pcdet uses it)"
        self.ndim = ndim
        self.in_channels, self.out_channels = int(in_channels), int(out_channels)
        self.kernel_size = _ntuple(kernel_size, ndim)
        self.stride = _ntuple(stride, ndim)
        self.padding = _ntuple(padding, ndim)
        self.dilation = _ntuple(dilation, ndim)
        self.output_padding = _ntuple(output_padding, ndim)
        self.groups = groups
        self.subm, self.transposed, self.inverse = bool(subm), bool(transposed), bool(inverse)
        self.indice_key = indice_key
        self.algo = algo
        self.fp32_accum = fp32_accum
        self.conv1x1 = int(np.prod(self.kernel_size)) == 1
        if self.transposed:
            raise NotImplementedError("SparseConvTranspose is not part of the hot path (no caller in pcdet)")
        self.weight = nn.Parameter(torch.empty(self.out_channels, *self.kernel_size, self.in_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(self.out_channels))
        else:
            self.register_parameter("bias", None)
        self._prepared = None   # (key, tensor) cache of the kernel-ready weight
        self.reset_parameters()

    def extra_repr(self):
        s = "{in_channels}, {out_channels}, kernel_size={kernel_size}, stride={stride}, padding={padding}, dilation={dilation}"
        if self.bias is None:
            s += ", bias=False"
        s += ", subm={subm}, indice_key={indice_key}"
        return s.format(**self.__dict__)

    def reset_parameters(self):
        # spconv initialises like nn.Conv: kaiming-uniform(a = sqrt 5) with fan_in = Cin * prod(k)
        fan_in = self.in_channels * int(np.prod(self.kernel_size))
        bound = math.sqrt(6.0 / ((1 + 5.0) * fan_in))
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                b = 1.0 / math.sqrt(fan_in)
                self.bias.uniform_(-b, b)

    # ---- weights in kernel order ---------------------------------------------------------------------------
    def _padded_cin(self, dtype):
        """input channels the forward kernel is given: a narrow first layer (the 4- or 5-channel voxel means) is zero-padded to
        16 channels when that puts it on the MFMA kernels instead of the scalar one (the products with the padding are zeros)"""
        cin = self.in_channels
        if cin < 16 and dtype != torch.float32:
            L = _lib.lib()
            code = _lib.DTYPE_CODE[dtype]
            if not L.pvx_sparse_conv_uses_mfma(cin, self.out_channels, code) and L.pvx_sparse_conv_uses_mfma(16, self.out_channels, code):
                return 16
        return cin

    def _prepared_weight(self, weight, dtype):
        key = (weight.data_ptr(), weight._version, weight.dtype, dtype, weight.device)
        if self._prepared is not None and self._prepared[0] == key:
            return self._prepared[1]
        L = _lib.lib()
        w = weight.detach().to(dtype)
        cin = self._padded_cin(dtype)
        if cin != self.in_channels:
            w = torch.nn.functional.pad(w, (0, cin - self.in_channels))
        w = w.contiguous()
        kvol = int(np.prod(self.kernel_size))
        code = _lib.DTYPE_CODE[dtype]
        nbytes = L.pvx_sparse_conv_weight_bytes(self.out_channels, kvol, cin, code)
        out = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
        with torch.cuda.device(weight.device):
            _lib.check(L.pvx_sparse_conv_prepare_weight(_lib.ptr(w), self.out_channels, kvol, cin, code,
                                                        _lib.ptr(out), _lib.stream()), "pvx_sparse_conv_prepare_weight")
        self._prepared = (key, out)
        return out

    def _launch(self, features, weight, bias, nbr, n_out):
        L = _lib.lib()
        dev = features.device
        dtype = features.dtype
        kvol = int(np.prod(self.kernel_size))
        pw = self._prepared_weight(weight, dtype)
        cin = self._padded_cin(dtype)
        if cin != self.in_channels:
            features = torch.nn.functional.pad(features, (0, cin - self.in_channels))
        b32 = bias.detach().float().contiguous() if bias is not None else None
        out = torch.empty((n_out, self.out_channels), dtype=dtype, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.pvx_sparse_conv_fwd(_lib.ptr(features), features.shape[0], cin, _lib.ptr(pw),
                                             _lib.ptr(b32), _lib.ptr(nbr), kvol, n_out, None, self.out_channels,
                                             _lib.DTYPE_CODE[dtype], None, None, None, 0, _lib.ptr(out), _lib.stream()),
                       "pvx_sparse_conv_fwd")
        return out

    # ---- backward ------------------------------------------------------------------------------------------
    def _dgrad(self, grad_out, weight, nbr_back, flip, n_in):
        """grad_in[i] = sum_k W[:, k, :]^T grad_out[o(k, i)]: a forward conv Cout -> Cin over the inverse rulebook"""
        L = _lib.lib()
        dev, dtype = grad_out.device, grad_out.dtype
        kvol = int(np.prod(self.kernel_size))
        w = weight.detach().reshape(self.out_channels, kvol, self.in_channels)
        if flip:
            w = w.flip(1)
        w = w.permute(2, 1, 0).to(dtype).contiguous()            # [Cin, K, Cout]
        code = _lib.DTYPE_CODE[dtype]
        pw = torch.empty(L.pvx_sparse_conv_weight_bytes(self.in_channels, kvol, self.out_channels, code), dtype=torch.uint8, device=dev)
        out = torch.empty((n_in, self.in_channels), dtype=dtype, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.pvx_sparse_conv_prepare_weight(_lib.ptr(w), self.in_channels, kvol, self.out_channels, code, _lib.ptr(pw),
                                                        _lib.stream()), "pvx_sparse_conv_prepare_weight")
            _lib.check(L.pvx_sparse_conv_fwd(_lib.ptr(grad_out), grad_out.shape[0], self.out_channels, _lib.ptr(pw), None,
                                             _lib.ptr(nbr_back), kvol, n_in, None, self.in_channels, code, None, None, None, 0,
                                             _lib.ptr(out), _lib.stream()), "pvx_sparse_conv_fwd (dgrad)")
        return out

    def _wgrad(self, features, grad_out, nbr, n_out):
        L = _lib.lib()
        dev = features.device
        kvol = int(np.prod(self.kernel_size))
        gw = torch.empty((self.out_channels, kvol, self.in_channels), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            ws = _lib.workspace(L.pvx_sparse_conv_wgrad_workspace_bytes(n_out, kvol, self.in_channels, self.out_channels), dev)
            _lib.check(L.pvx_sparse_conv_wgrad(_lib.ptr(features), features.shape[0], self.in_channels, _lib.ptr(grad_out),
                                               _lib.ptr(nbr), kvol, n_out, None, self.out_channels,
                                               _lib.DTYPE_CODE[features.dtype], _lib.ptr(gw), _lib.ptr(ws), ws.numel(),
                                               _lib.stream()), "pvx_sparse_conv_wgrad")
        return gw

    # ---- rulebooks ---------------------------------------------------------------------------------------------
    def _build_subm(self, x):
        L = _lib.lib()
        dev = x.indices.device
        index, perm = x._site_index()
        coords = x._coords4()
        n = coords.shape[0]
        kvol = int(np.prod(self.kernel_size))
        nbr = torch.empty((kvol, n), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.pvx_subm_rulebook(_lib.ptr(coords), n, None, x.batch_size, _lib.int3(x._shape3()), _lib.ptr(index),
                                           _lib.ptr(perm), _lib.int3(_to3(self.kernel_size, self.ndim, 1)),
                                           _lib.int3(_to3(self.dilation, self.ndim, 1)), _lib.ptr(nbr), _lib.stream()),
                       "pvx_subm_rulebook")
        return IndiceData(nbr, n, n, x.indices, x.indices, x.spatial_shape, x.spatial_shape, self.kernel_size, self.stride,
                          self.padding, self.dilation, True)

    def _build_strided(self, x):
        L = _lib.lib()
        dev = x.indices.device
        in_index, in_perm = x._site_index()
        coords = x._coords4()
        n_in = coords.shape[0]
        k3 = _lib.int3(_to3(self.kernel_size, self.ndim, 1))
        s3 = _lib.int3(_to3(self.stride, self.ndim, 1))
        p3 = _lib.int3(_to3(self.padding, self.ndim, 0))
        d3 = _lib.int3(_to3(self.dilation, self.ndim, 1))
        in3 = _lib.int3(x._shape3())
        out3 = (_lib.ctypes.c_int * 3)()
        _lib.check(L.pvx_conv_out_shape(in3, k3, s3, p3, d3, out3), "pvx_conv_out_shape")
        out_shape3 = [out3[0], out3[1], out3[2]]
        kvol = int(np.prod(self.kernel_size))
        with torch.cuda.device(dev):
            nbytes = L.pvx_site_index_bytes(x.batch_size, out3)
            if nbytes == 0:
                raise ValueError("output grid %s is outside what the site index supports" % out_shape3)
            out_index = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            n_out_dev = torch.zeros(1, dtype=torch.int32, device=dev)
            ws = _lib.workspace(L.pvx_site_index_workspace_bytes(x.batch_size, out3), dev)
            _lib.check(L.pvx_conv_out_index(_lib.ptr(coords), n_in, None, x.batch_size, in3, k3, s3, p3, d3,
                                            _lib.ptr(out_index), _lib.ptr(n_out_dev), _lib.ptr(ws), ws.numel(),
                                            _lib.stream()), "pvx_conv_out_index")
            n_out = int(n_out_dev.item())   # the one host read a data-dependent output SHAPE needs (spconv does the same)
            coords_out = torch.empty((n_out, 4), dtype=torch.int32, device=dev)
            nbr = torch.empty((kvol, n_out), dtype=torch.int32, device=dev)
            _lib.check(L.pvx_conv_out_rulebook(_lib.ptr(out_index), n_out, None, x.batch_size, in3, k3, s3, p3, d3,
                                               _lib.ptr(in_index), _lib.ptr(in_perm), _lib.ptr(coords_out), _lib.ptr(nbr),
                                               _lib.stream()), "pvx_conv_out_rulebook")
        out_indices = coords_out if self.ndim == 3 else coords_out[:, [0, 2, 3]].contiguous()
        out_shape = out_shape3 if self.ndim == 3 else out_shape3[1:]
        return IndiceData(nbr, n_in, n_out, out_indices, x.indices, out_shape, x.spatial_shape, self.kernel_size,
                          self.stride, self.padding, self.dilation, False, out_index_cache=(out_index, n_out_dev))

    def _inverse_of(self, x, data):
        if data.nbr_inv is None:
            L = _lib.lib()
            dev = x.indices.device
            kvol = data.nbr.shape[0]
            inv = torch.empty((kvol, data.n_in), dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(L.pvx_inverse_rulebook(_lib.ptr(data.nbr), kvol, data.n_out, data.n_in, _lib.ptr(inv),
                                                  _lib.stream()), "pvx_inverse_rulebook")
            data.nbr_inv = inv
        return data.nbr_inv

    # ---- forward -------------------------------------------------------------------------------------------
    def forward(self, input):
        assert isinstance(input, SparseConvTensor), "expected a SparseConvTensor"
        feats = input.features
        if not feats.is_cuda:
            raise RuntimeError("sparse convolution runs on the GPU only (there is no CPU fallback)")
        assert feats.shape[1] == self.in_channels, "channel mismatch: %d vs %d" % (feats.shape[1], self.in_channels)
        if torch.is_autocast_enabled():
            feats = feats.to(torch.get_autocast_dtype('cuda'))
        if feats.dtype not in _lib.DTYPE_CODE:
            raise TypeError("unsupported feature dtype %s" % feats.dtype)
        feats = feats.contiguous()

        data = input.find_indice_pair(self.indice_key)
        if self.inverse:
            assert data is not None and not data.subm, "SparseInverseConv needs the indice_key of an earlier SparseConv"
            assert data.n_out == feats.shape[0], "inverse conv input does not match the cached rulebook"
            assert list(data.ksize) == list(self.kernel_size)
            nbr, n_out = self._inverse_of(input, data), data.n_in
            out_indices, out_shape = data.in_indices, data.in_spatial_shape
            seed = None
            back = (lambda d=data: d.nbr, False)
        elif self.subm:
            if self.conv1x1 and data is None:
                data = "identity"
            if data is None:
                data = self._build_subm(input)
                if self.indice_key is not None:
                    input.indice_dict[self.indice_key] = data
            elif data != "identity":
                assert data.subm and data.n_out == feats.shape[0], "cached rulebook does not match this tensor"
            if data == "identity":   # 1x1 submanifold conv = a plain row-wise linear map (VoxelNeXtHead)
                n = feats.shape[0]
                nbr = torch.arange(n, dtype=torch.int32, device=feats.device).view(1, n)
                n_out = n
            else:
                nbr, n_out = data.nbr, data.n_out
            out_indices, out_shape = input.indices, input.spatial_shape
            seed = None
            back = (lambda n=nbr: n, True)
        else:
            if data is None:
                data = self._build_strided(input)
                if self.indice_key is not None:
                    input.indice_dict[self.indice_key] = data
            else:
                assert not data.subm and data.n_in == feats.shape[0], "cached rulebook does not match this tensor"
            nbr, n_out = data.nbr, data.n_out
            out_indices, out_shape = data.out_indices, data.out_spatial_shape
            seed = data.out_index_cache
            back = (lambda d=data, x=input: self._inverse_of(x, d), False)

        needs_grad = torch.is_grad_enabled() and (feats.requires_grad or self.weight.requires_grad)
        if needs_grad:
            out_feats = _SparseConvFunction.apply(feats, self.weight, self.bias, self, nbr, n_out, back)
        else:
            out_feats = self._launch(feats, self.weight, self.bias, nbr, n_out)

        if self.subm and not self.inverse:
            out = input.replace_feature(out_feats)
        else:
            out = SparseConvTensor(out_feats, out_indices, out_shape, input.batch_size, indice_dict=input.indice_dict,
                                   benchmark=input.benchmark)
            if seed is not None:
                out._seed_site_index(seed[0], None, seed[1])
        return out


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, algo=None, fp32_accum=None, name=None, **kwargs):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, True,
                         indice_key=indice_key, algo=algo, fp32_accum=fp32_accum, name=name)


class SubMConv2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, algo=None, fp32_accum=None, name=None, **kwargs):
        super().__init__(2, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, True,
                         indice_key=indice_key, algo=algo, fp32_accum=fp32_accum, name=name)


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, algo=None, fp32_accum=None, name=None, **kwargs):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         indice_key=indice_key, algo=algo, fp32_accum=fp32_accum, name=name)


class SparseConv2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, algo=None, fp32_accum=None, name=None, **kwargs):
        super().__init__(2, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         indice_key=indice_key, algo=algo, fp32_accum=fp32_accum, name=name)


class SparseInverseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key, bias=True, algo=None, fp32_accum=None,
                 name=None, **kwargs):
        super().__init__(3, in_channels, out_channels, kernel_size, bias=bias, inverse=True, indice_key=indice_key,
                         algo=algo, fp32_accum=fp32_accum, name=name)


class SparseInverseConv2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key, bias=True, algo=None, fp32_accum=None,
                 name=None, **kwargs):
        super().__init__(2, in_channels, out_channels, kernel_size, bias=bias, inverse=True, indice_key=indice_key,
                         algo=algo, fp32_accum=fp32_accum, name=name)
