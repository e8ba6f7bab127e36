"""scatter_mean / scatter_max along dim 0 -- the two torch_scatter functions the reference's dynamic VFEs import
(pcdet/models/backbones_3d/vfe/dynamic_pillar_vfe.py:5-9,40,105; dynamic_mean_vfe.py:65).  Same call shape
(`scatter_mean(src, index, dim=0)`, `scatter_max(src, index, dim=0) -> (values, argmax)`), backed by
pvx_scatter_mean / pvx_scatter_max; both are differentiable (PFN weights train through scatter_max).
`index` may be int32 or int64; negative entries are skipped.  `dim_size` avoids the host read of index.max()."""
import torch

from . import _lib


def _prep(src, index, dim, dim_size):
    assert dim == 0, "only dim=0 is used by pcdet and implemented here"
    assert src.is_cuda and index.is_cuda and index.dim() == 1 and index.shape[0] == src.shape[0]
    n = src.shape[0]
    if dim_size is None:
        dim_size = int(index.max().item()) + 1 if n > 0 else 0
    src2 = src.reshape(n, -1).float().contiguous()
    return src2, index.to(torch.int32).contiguous(), n, src2.shape[1], int(dim_size)


class _ScatterMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, index, m):
        L = _lib.lib()
        n, c = src.shape
        out = torch.empty((m, c), dtype=torch.float32, device=src.device)
        ws = _lib.workspace(L.pvx_scatter_mean_workspace_bytes(m, c), src.device)
        with torch.cuda.device(src.device):
            _lib.check(L.pvx_scatter_mean(_lib.ptr(src), _lib.ptr(index), n, None, c, m, _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                          _lib.stream()), "pvx_scatter_mean")
        ctx.save_for_backward(index)
        ctx.m = m
        return out

    @staticmethod
    def backward(ctx, grad):
        index, = ctx.saved_tensors
        idx = index.long().clamp(min=0)
        cnt = torch.bincount(idx[index >= 0], minlength=ctx.m).clamp(min=1).to(grad.dtype)
        g = (grad / cnt.unsqueeze(1))[idx]
        g = torch.where((index >= 0).unsqueeze(1), g, torch.zeros_like(g))
        return g, None, None


class _ScatterMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, index, m):
        L = _lib.lib()
        n, c = src.shape
        out = torch.empty((m, c), dtype=torch.float32, device=src.device)
        arg = torch.empty((m, c), dtype=torch.int32, device=src.device)
        with torch.cuda.device(src.device):
            _lib.check(L.pvx_scatter_max(_lib.ptr(src), _lib.ptr(index), n, None, c, m, _lib.ptr(out), _lib.ptr(arg), _lib.stream()),
                       "pvx_scatter_max")
        ctx.save_for_backward(arg)
        ctx.n = n
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, grad, _):
        arg, = ctx.saved_tensors
        g = grad.new_zeros((ctx.n + 1, grad.shape[1]))          # row n collects the empty segments
        g.scatter_(0, arg.long(), grad)
        return g[:ctx.n], None, None


def scatter_mean(src, index, dim=0, out=None, dim_size=None):
    assert out is None
    s, i, n, c, m = _prep(src, index, dim, dim_size)
    r = _ScatterMean.apply(s, i, m)
    return r.reshape((m,) + tuple(src.shape[1:])).to(src.dtype)


def scatter_max(src, index, dim=0, out=None, dim_size=None):
    assert out is None
    s, i, n, c, m = _prep(src, index, dim, dim_size)
    v, a = _ScatterMax.apply(s, i, m)
    shape = (m,) + tuple(src.shape[1:])
    return v.reshape(shape).to(src.dtype), a.reshape(shape).long()
