"""SparseModule / SparseSequential (spconv.pytorch.modules [spconv]; used at
pcdet/models/backbones_3d/spconv_backbone.py:21-25,30 and throughout the sparse backbones)."""
from collections import OrderedDict

import torch
from torch import nn

from .core import SparseConvTensor


def is_spconv_module(module):
    return isinstance(module, SparseModule)


def is_sparse_conv(module):
    from .conv import SparseConvolution
    return isinstance(module, SparseConvolution)


class SparseModule(nn.Module):
    """Marker base: a module whose forward takes and returns a SparseConvTensor."""

    def __init__(self, name=None):
        super().__init__()
        self.name = name


class SparseSequential(SparseModule):
    """nn.Sequential that hands SparseConvTensors to sparse modules and `.features` to everything else
    (BatchNorm1d, ReLU ... at spconv_backbone.py:21-25)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError("index {} is out of range".format(idx))
        if idx < 0:
            idx += len(self)
        it = iter(self._modules.values())
        for _ in range(idx):
            next(it)
        return next(it)

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def forward(self, input):
        from . import norm
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            module = mods[i]
            if is_spconv_module(module):
                assert isinstance(input, SparseConvTensor)
                input = module(input)
            elif isinstance(input, SparseConvTensor):
                if input.indices.shape[0] != 0:
                    # BatchNorm1d -> ReLU in training mode: one fused op (csrc/batchnorm.hip) instead of six torch launches
                    if (isinstance(module, (nn.BatchNorm1d, nn.SyncBatchNorm)) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                            and norm.fusable(module, input.features)):
                        input = input.replace_feature(norm.bn_act(input.features, module, relu=True))
                        i += 1
                    else:
                        input = input.replace_feature(module(input.features))
            else:
                input = module(input)
            i += 1
        return input


class ToDense(SparseModule):
    def forward(self, x):
        return x.dense()


class RemoveGrid(SparseModule):
    def forward(self, x):
        x.grid = None
        return x
