"""`import spconv.pytorch as spconv` stand-in (the one import gate is pcdet/utils/spconv_utils.py:3-10).
Every symbol pcdet/ and tools/ use from that namespace (boundary B1) lives here."""
from . import conv, core, modules, ops, utils  # noqa: F401
from .conv import (SparseConv2d, SparseConv3d, SparseConvolution, SparseInverseConv2d, SparseInverseConv3d,  # noqa: F401
                   SubMConv2d, SubMConv3d)
from .core import ConvAlgo, SparseConvTensor  # noqa: F401
from .modules import SparseModule, SparseSequential, ToDense  # noqa: F401
