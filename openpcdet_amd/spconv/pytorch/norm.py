"""BatchNorm1d (+ residual) + ReLU over a sparse tensor's feature matrix in one fused op for the TRAINING step: what
`norm_fn(C) -> nn.ReLU()` (pcdet/models/backbones_3d/spconv_backbone.py:21-25) and `bn -> += identity -> relu` (:52-64) compute,
on csrc/batchnorm.hip.  The modules stay plain `nn.BatchNorm1d` / `nn.ReLU` (state-dict keys, `.eval()`, SyncBatchNorm conversion
all unchanged); `SparseSequential` and `SparseBasicBlock` call `bn_act` where they see the pattern.  Anything the kernels do not
take -- eval mode, SyncBatchNorm, momentum=None, CPU tensors, odd channel counts -- goes through the modules themselves."""
import os

import torch
from torch import nn
from torch.autograd import Function

from ... import _lib

_ENABLED = os.environ.get("PVX_FUSED_BN", "1") != "0"       # A/B switch


class _BNAct(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, residual, eps, momentum, relu):
        L = _lib.lib()
        x = x.contiguous()
        n, c = x.shape
        code = _lib.DTYPE_CODE[x.dtype]
        dev = x.device
        if residual is not None:
            residual = residual.to(x.dtype).contiguous()
        y = torch.empty_like(x)
        mean = torch.empty(c, dtype=torch.float32, device=dev)
        invstd = torch.empty(c, dtype=torch.float32, device=dev)
        w = weight.detach().float().contiguous() if weight is not None else None
        b = bias.detach().float().contiguous() if bias is not None else None
        with torch.cuda.device(dev):
            ws = _lib.workspace(L.pvx_bn_act_workspace_bytes(c), dev)
            _lib.check(L.pvx_bn_act_fwd(_lib.ptr(x), n, c, code, _lib.ptr(w), _lib.ptr(b), float(eps), float(momentum),
                                        _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(residual), int(relu), _lib.ptr(y),
                                        _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(ws), ws.numel(), _lib.stream()), "pvx_bn_act_fwd")
        ctx.save_for_backward(x, y, w, mean, invstd)
        ctx.relu, ctx.has_res = bool(relu), residual is not None
        ctx.param_dtypes = (weight.dtype if weight is not None else None, bias.dtype if bias is not None else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, w, mean, invstd = ctx.saved_tensors
        L = _lib.lib()
        n, c = x.shape
        dev = x.device
        dy = dy.to(x.dtype).contiguous()
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_res else None
        dgamma = torch.empty(c, dtype=torch.float32, device=dev)
        dbeta = torch.empty(c, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            ws = _lib.workspace(L.pvx_bn_act_workspace_bytes(c), dev)
            _lib.check(L.pvx_bn_act_bwd(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(y), n, c, _lib.DTYPE_CODE[x.dtype], _lib.ptr(w), _lib.ptr(mean),
                                        _lib.ptr(invstd), int(ctx.relu), _lib.ptr(dx), _lib.ptr(dres), _lib.ptr(dgamma), _lib.ptr(dbeta),
                                        _lib.ptr(ws), ws.numel(), _lib.stream()), "pvx_bn_act_bwd")
        wd, bd = ctx.param_dtypes
        return (dx, dgamma.to(wd) if wd is not None else None, dbeta.to(bd) if bd is not None else None, None, None, dres, None, None, None)


class _SyncBNAct(Function):
    """the same op with statistics over every rank of `group` (torch.nn.SyncBatchNorm's semantics, torch/nn/modules/_functions.py):
    local column sums + row count -> one all-reduce (RCCL) -> finish and apply; no host synchronisation"""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, residual, eps, momentum, relu, group):
        import torch.distributed as dist
        L = _lib.lib()
        x = x.contiguous()
        n, c = x.shape
        code = _lib.DTYPE_CODE[x.dtype]
        dev = x.device
        if residual is not None:
            residual = residual.to(x.dtype).contiguous()
        y = torch.empty_like(x)
        mean = torch.empty(c, dtype=torch.float32, device=dev)
        invstd = torch.empty(c, dtype=torch.float32, device=dev)
        sums = torch.empty(2 * c + 1, dtype=torch.float64, device=dev)
        w = weight.detach().float().contiguous() if weight is not None else None
        b = bias.detach().float().contiguous() if bias is not None else None
        with torch.cuda.device(dev):
            ws = _lib.workspace(L.pvx_bn_act_workspace_bytes(c), dev)
            args = (_lib.ptr(x), n, c, code, _lib.ptr(w), _lib.ptr(b), float(eps), float(momentum), _lib.ptr(running_mean),
                    _lib.ptr(running_var), _lib.ptr(residual), int(relu), _lib.ptr(y), _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(sums))
            _lib.check(L.pvx_bn_act_fwd_sync(*args, 0, 1, _lib.ptr(ws), ws.numel(), _lib.stream()), "pvx_bn_act_fwd_sync")
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
            _lib.check(L.pvx_bn_act_fwd_sync(*args, 0, 2, _lib.ptr(ws), ws.numel(), _lib.stream()), "pvx_bn_act_fwd_sync")
        ctx.save_for_backward(x, y, w, mean, invstd)
        ctx.relu, ctx.has_res, ctx.group = bool(relu), residual is not None, group
        ctx.param_dtypes = (weight.dtype if weight is not None else None, bias.dtype if bias is not None else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        import torch.distributed as dist
        x, y, w, mean, invstd = ctx.saved_tensors
        L = _lib.lib()
        n, c = x.shape
        dev = x.device
        dy = dy.to(x.dtype).contiguous()
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_res else None
        dgamma = torch.empty(c, dtype=torch.float32, device=dev)
        dbeta = torch.empty(c, dtype=torch.float32, device=dev)
        sums = torch.empty(2 * c + 1, dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            ws = _lib.workspace(L.pvx_bn_act_workspace_bytes(c), dev)
            args = (_lib.ptr(dy), _lib.ptr(x), _lib.ptr(y), n, c, _lib.DTYPE_CODE[x.dtype], _lib.ptr(w), _lib.ptr(mean), _lib.ptr(invstd),
                    int(ctx.relu), _lib.ptr(dx), _lib.ptr(dres), _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(sums))
            _lib.check(L.pvx_bn_act_bwd_sync(*args, 0, 1, _lib.ptr(ws), ws.numel(), _lib.stream()), "pvx_bn_act_bwd_sync")
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=ctx.group)
            _lib.check(L.pvx_bn_act_bwd_sync(*args, 0, 2, _lib.ptr(ws), ws.numel(), _lib.stream()), "pvx_bn_act_bwd_sync")
        wd, bd = ctx.param_dtypes
        return (dx, dgamma.to(wd) if wd is not None else None, dbeta.to(bd) if bd is not None else None, None, None, dres, None, None, None,
                None)


def _sync_group(bn):
    """the process group a SyncBatchNorm module reduces over, or None when it would not synchronise (torch falls back to plain
    batch_norm then: not training, no process group; a group of ONE rank is handled by bn_act without a collective)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    return bn.process_group if bn.process_group is not None else dist.group.WORLD


def fusable(bn, x):
    if type(bn) is nn.SyncBatchNorm:
        # decided from facts every rank of the group shares (module, mode, dtype, channels, group) and never from this rank's row
        # count: the fused path issues ONE all-reduce where torch's SyncBatchNorm issues an all-gather, so two ranks that chose
        # differently (a frame with a single active row at some level, PartA2's fake_sparse_idx) would hang or mix buffers.
        # The kernels take n == 1, and n == 0 as "zero sums, nothing to apply".
        return (_ENABLED and bn.training and bn.momentum is not None and x.is_cuda and x.dim() == 2
                and x.dtype in _lib.DTYPE_CODE and _sync_group(bn) is not None
                and bool(_lib.lib().pvx_bn_act_supported(x.shape[1], _lib.DTYPE_CODE[x.dtype])))
    return (_ENABLED and type(bn) is nn.BatchNorm1d and bn.training and bn.momentum is not None and x.is_cuda and x.dim() == 2
            and x.shape[0] > 1 and x.dtype in _lib.DTYPE_CODE and bool(_lib.lib().pvx_bn_act_supported(x.shape[1], _lib.DTYPE_CODE[x.dtype])))


def bn_act(x, bn, residual=None, relu=True):
    """relu?(bn(x) (+ residual)) for a [N, C] feature matrix; one fused forward and one fused backward when `fusable`"""
    if not fusable(bn, x):
        y = bn(x)
        if residual is not None:
            y = y + residual
        return torch.relu(y) if relu else y
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    args = (x, bn.weight if bn.affine else None, bn.bias if bn.affine else None, rm, rv, residual, bn.eps, bn.momentum, relu)
    if type(bn) is nn.SyncBatchNorm:
        group = _sync_group(bn)
        import torch.distributed as dist
        if dist.get_world_size(group) > 1:
            return _SyncBNAct.apply(*args, group)
        # a group of one rank: the statistics are this rank's own, no collective (torch.nn.SyncBatchNorm takes its plain batch_norm
        # path in that case too: need_sync = world_size > 1); the choice depends on the group only, so every rank makes the same one
        if x.shape[0] <= 1:
            y = bn(x)
            if residual is not None:
                y = y + residual
            return torch.relu(y) if relu else y
    return _BNAct.apply(*args)
