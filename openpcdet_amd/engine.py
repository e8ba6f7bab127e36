"""Sync-free execution plan for the whole hot path of one batch of LiDAR frames:

    raw points --voxelize + mean VFE--> sparse backbone (VoxelBackBone8x / VoxelResBackBone8x) --dense()--> BEV
    decoded boxes --score sort--> batched rotated NMS

The module API in `openpcdet_amd.spconv` mirrors spconv call for call and therefore reads one integer back per
strided conv (a tensor SHAPE has to be known on the host).  This plan is what a deployment runs instead -- the role
the reference gives to its SCN engine (libraries/3DSparseConvolution/libspconv/include/spconv/engine.hpp:58-104,
fed by tool/centerpoint-export/export-scn.py): every buffer is allocated once at the configured CAPACITY, every
data-dependent row count stays on the device (the `n_dev` arguments of the C ABI), eval-mode BatchNorm is folded into
the conv epilogue exactly as the reference's exporter folds it (centerpoint-export/funcs.py:46-63,145-168), the two
rulebooks the reference builds twice for one site set ('subm1' and 'res1', spconv_backbone.py:197-206) are built once,
and the whole sequence of C-ABI launches contains no allocation and no host synchronisation, so it can be captured
in a HIP graph and replayed.

The arithmetic is the module path's: the same kernels, the same rulebooks; tests/test_gpu_engine.py checks one
against the other.
"""
import ctypes
import os

import numpy as np
import torch
from torch import nn

from . import _lib
from .backbones import SparseBasicBlock
from .spconv.pytorch.conv import SparseConvolution
from .spconv.pytorch.modules import SparseSequential


def _fold_bn(bn, cout, device):
    """eval-mode BatchNorm1d as y = x * scale + shift (f32)."""
    if bn is None:
        return None, None
    assert isinstance(bn, nn.BatchNorm1d) and bn.num_features == cout
    var = bn.running_var.detach().float()
    mean = bn.running_mean.detach().float()
    g = bn.weight.detach().float() if bn.affine else torch.ones_like(var)
    b = bn.bias.detach().float() if bn.affine else torch.zeros_like(var)
    scale = g / torch.sqrt(var + bn.eps)
    shift = b - mean * scale
    return scale.to(device).contiguous(), shift.to(device).contiguous()


class _ConvSpec:
    """one fused conv launch: conv (+bias) -> folded BN -> (+residual) -> ReLU"""

    def __init__(self, conv, bn, relu, residual):
        self.conv, self.bn, self.relu, self.residual = conv, bn, relu, residual


def _flatten_backbone(backbone):
    """-> list of stages; a stage is a list of ('conv', _ConvSpec) / ('save',) / ('block', spec1, spec2) items in execution
    order, recognised from the module tree the reference's backbones (and backbones.py) build."""
    def seq_conv(seq):
        mods = list(seq._modules.values())
        assert isinstance(mods[0], SparseConvolution)
        bn = mods[1] if len(mods) > 1 and isinstance(mods[1], nn.BatchNorm1d) else None
        relu = any(isinstance(m, nn.ReLU) for m in mods[1:])
        return _ConvSpec(mods[0], bn, relu, False)

    stages = [[('conv', seq_conv(backbone.conv_input))]]
    for name in ('conv1', 'conv2', 'conv3', 'conv4'):
        items = []
        for m in getattr(backbone, name)._modules.values():
            if isinstance(m, SparseBasicBlock):
                assert m.downsample is None
                items.append(('block', _ConvSpec(m.conv1, m.bn1, True, False), _ConvSpec(m.conv2, m.bn2, True, True)))
            elif isinstance(m, SparseSequential):
                items.append(('conv', seq_conv(m)))
            else:
                raise TypeError("unrecognised backbone layer %r" % type(m))
        stages.append(items)
    stages.append([('conv', seq_conv(backbone.conv_out))])
    return stages


class _Level:
    """one resolution level: coordinates, live row count, site index, submanifold rulebooks"""

    def __init__(self, shape3, cap, batch, device, with_perm):
        L = _lib.lib()
        self.shape3, self.cap = list(shape3), int(cap)
        self.c_shape = _lib.int3(shape3)
        self.coords = torch.zeros((cap, 4), dtype=torch.int32, device=device)
        self.n_dev = torch.zeros(1, dtype=torch.int32, device=device)
        nbytes = L.pvx_site_index_bytes(batch, self.c_shape)
        if nbytes == 0:
            raise ValueError("grid %s x batch %d is outside what the site index supports" % (shape3, batch))
        # zero-filled ONCE: the plan's builders are the *_clean forms and every forward leaves the buffer zero again (pvx_site_index_unset)
        self.index = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        self.perm = torch.empty(cap, dtype=torch.int32, device=device) if with_perm else None
        self.ws = _lib.workspace(L.pvx_site_index_workspace_bytes(batch, self.c_shape), device)
        self.subm_nbr = {}    # (ksize tuple, dilation tuple) -> [K, cap] int32
        self.bufs = {}        # channel count -> list of feature buffers


class HotPathEngine:
    def __init__(self, backbone, batch_size, point_cloud_range, voxel_size, max_points_per_voxel, max_voxels_per_frame,
                 num_point_features, max_points_per_frame, dtype=torch.bfloat16, device="cuda", level_capacity_factor=None,
                 want_dense=True, sort_input=True):
        """backbone: a VoxelBackBone8x / VoxelResBackBone8x (this package's or a state-dict-compatible tree) in eval mode.
        level_capacity_factor: row capacity of each strided conv's output relative to its input's capacity.  None (the
        default) sizes every level for the worst case, prod_j ceil(k_j / s_j) outputs per input site (8 for k3 s2),
        capped by the number of grid cells -- never overflows, costs memory and idle workgroups.  Dense LiDAR frames give
        ~1x, isolated voxels up to ~3.4x; with an explicit factor, rows beyond the capacity are dropped and
        `overflowed()` reports it.
        sort_input: move the voxels from the voxelizer's first-come order into key order before the first conv
        (pvx_sort_sites; the level-0 tiles then walk neighbouring cells).  `stage_output` still reports level-0 rows in
        first-come order.  Needs 16-byte-multiple feature rows, otherwise it is switched off."""
        assert not backbone.training, "the plan folds BatchNorm running statistics: call backbone.eval() first"
        assert dtype in _lib.DTYPE_CODE
        self.L = _lib.lib()
        self.device = torch.device(device)
        self.dtype, self.code = dtype, _lib.DTYPE_CODE[dtype]
        self.B = int(batch_size)
        self.C = int(num_point_features)
        self.T = int(max_points_per_voxel)
        self.max_voxels = int(max_voxels_per_frame)
        self.n_pts_cap = int(max_points_per_frame)
        self.vsize = (ctypes.c_float * 3)(*[float(v) for v in voxel_size])
        self.range = (ctypes.c_float * 6)(*[float(v) for v in point_cloud_range])
        self.want_dense = want_dense
        self.sort_input = bool(sort_input)
        dev = self.device
        with torch.cuda.device(dev):
            self._build(backbone, level_capacity_factor)

    # layers that share a rulebook with enough launches to pay for its book (55 us): the five 16 -> 16 layers (bound by rulebook
    # bytes) and the four 32 -> 32 layers (the rolling kernel form needs the book's window spans a tile ahead)
    # (32, 64): its book form measured slower (282 vs 254 us); (64, 128), (128, 128): streaming kernels, which take only the window spans
    # from the book (PVX_ENGINE_SPAN_BOOKS=0: without)
    TILEBOOK_SHAPES = {(16, 16), (16, 32), (32, 32), (64, 64)} | ({(64, 128), (128, 128)} if os.environ.get("PVX_ENGINE_SPAN_BOOKS", "1") != "0" else set())

    def _wants_tilebook(self, cin, cout):
        if self.tilebook_mode == "0":
            return False
        if self.tilebook_mode == "all":
            return True
        if (cin, cout) == (32, 32):                    # only for the rolling kernel form (its books have 896-row windows)
            return _lib.lib().pvx_tilebook_rcap(32, 32, 27, self.code) >= 896
        return (cin, cout) in self.TILEBOOK_SHAPES

    # ------------------------------------------------------------------------------------------------------------
    def _feat_buf(self, level, channels, avoid=()):
        """a [cap, channels] buffer of `level` that is none of `avoid` (buffers are recycled within a level)"""
        pool = level.bufs.setdefault(channels, [])
        for b in pool:
            if all(b is not a for a in avoid):
                return b
        b = torch.zeros((level.cap, channels), dtype=self.dtype, device=self.device)
        pool.append(b)
        return b

    def _prep_weight(self, conv, cin_pad):
        """kernel-ready weight, input channels zero-padded to cin_pad"""
        L = self.L
        cout, cin = conv.out_channels, conv.in_channels
        kvol = int(np.prod(conv.kernel_size))
        w = conv.weight.detach().to(self.device).float().reshape(cout, kvol, cin)
        if cin_pad != cin:
            w = torch.nn.functional.pad(w, (0, cin_pad - cin))
        w = w.to(self.dtype).contiguous()
        out = torch.empty(L.pvx_sparse_conv_weight_bytes(cout, kvol, cin_pad, self.code), dtype=torch.uint8, device=self.device)
        _lib.check(L.pvx_sparse_conv_prepare_weight(_lib.ptr(w), cout, kvol, cin_pad, self.code, _lib.ptr(out), _lib.stream()),
                   "pvx_sparse_conv_prepare_weight")
        return out

    def _build(self, backbone, factor):
        L, dev, B = self.L, self.device, self.B
        stages = _flatten_backbone(backbone)
        shape0 = [int(s) for s in backbone.sparse_shape]
        grid = (ctypes.c_int * 3)()
        _lib.check(L.pvx_voxel_grid_size(self.vsize, self.range, grid), "pvx_voxel_grid_size")
        assert [grid[2] + 1, grid[1], grid[0]] == shape0, "voxel grid %s does not match the backbone's sparse_shape %s" % (
            list(grid), shape0)

        # input features: the first conv's input channels padded so that the MFMA kernel can take it
        first = stages[0][0][1].conv
        assert first.in_channels == self.C
        self.cin0 = self.C
        if self.dtype != torch.float32 and L.pvx_sparse_conv_uses_mfma(16, first.out_channels, self.code) and self.C <= 16:
            self.cin0 = 16

        row_bytes0 = self.cin0 * torch.empty((), dtype=self.dtype).element_size()
        self.sort_input = self.sort_input and row_bytes0 % 16 == 0
        lvl = _Level(shape0, B * self.max_voxels, B, dev, with_perm=not self.sort_input)
        self.levels = [lvl]
        self.feat0 = torch.zeros((lvl.cap, self.cin0), dtype=self.dtype, device=dev)
        # what the voxelizer writes: the level's own rows, or first-come staging rows that pvx_sort_sites moves into key order
        self.vox_coords, self.vox_feat = lvl.coords, self.feat0
        if self.sort_input:
            self.vox_coords, self.vox_feat = torch.zeros_like(lvl.coords), torch.zeros_like(self.feat0)
            self.rank0 = torch.full((lvl.cap,), -1, dtype=torch.int32, device=dev)
        self.points = torch.full((B, self.n_pts_cap, self.C), float('nan'), dtype=torch.float32, device=dev)
        self.num_voxels = torch.zeros(B, dtype=torch.int32, device=dev)
        self.vox_ws = _lib.workspace(L.pvx_voxelize_batch_workspace_bytes(B, self.n_pts_cap, self.max_voxels, self.T), dev)

        self.ops = []          # (name, callable(stream))
        self._books = {}       # (rulebook address, window rows) -> tile book buffer
        self._fused_books = {} # rulebook address -> (index in self.ops, arguments, window rows) of the launch that writes rulebook + book
        self._subm_ops = {}    # rulebook address -> (index in self.ops, arguments) of the launch that builds it, while it has no book
        # which (cin, cout) layers read their rulebook through a tile book: building one costs about as much as one 16 -> 16
        # layer saves, so it pays where several rulebook-bound layers share it -- the five 16 -> 16 layers of level 0 (104 -> 68 us
        # each at B = 16); 32 -> 32 and 64 -> 64 gain 0-3 %, less than their book costs (profiles/r02_tilebook.txt).
        # PVX_ENGINE_TILEBOOK=0 / all: none / every shape that has the form (A/B switch).
        self.tilebook_mode = os.environ.get("PVX_ENGINE_TILEBOOK", "1")
        # the level indices are filled with zeros once and un-set after use (25 MB of scattered stores instead of a 340 MB fill per forward at
        # level 0 of the nuScenes grid, batch 16); PVX_ENGINE_CLEAN_INDEX=0: the builders that fill (A/B switch)
        self.clean_index = os.environ.get("PVX_ENGINE_CLEAN_INDEX", "1") != "0"
        # a submanifold rulebook all of whose layers read it through ONE tile book is stored as that book + its far entries only
        # (pvx_subm_book / pvx_sparse_conv_fwd_book); PVX_ENGINE_FAR_ONLY=0: full int32 rulebooks (A/B switch)
        self.far_only = os.environ.get("PVX_ENGINE_FAR_ONLY", "1") != "0"
        self.schedule = os.environ.get("PVX_ENGINE_SCHEDULE", "1") != "0"
        self.conv_records = [] # per conv launch: dict(name, cin, cout, kvol, nbr, level_out) for flop accounting
        keep = []              # keeps ctypes arrays / tensors alive
        self._keep = keep
        cur, cur_feat, cin_cur = lvl, self.feat0, self.cin0
        self.stage_outputs = []

        def add_subm_rulebook(level, conv):
            key = (tuple(conv.kernel_size), tuple(conv.dilation))
            if key in level.subm_nbr:
                return level.subm_nbr[key]
            kvol = int(np.prod(conv.kernel_size))
            nbr = torch.empty((kvol, level.cap), dtype=torch.int32, device=dev)
            level.subm_nbr[key] = nbr
            k3, d3 = _lib.int3(conv.kernel_size), _lib.int3(conv.dilation)
            keep.extend([k3, d3])
            args = (_lib.ptr(level.coords), level.cap, _lib.ptr(level.n_dev), B, level.c_shape, _lib.ptr(level.index),
                    _lib.ptr(level.perm), k3, d3, _lib.ptr(nbr))
            self.ops.append(("subm_rulebook", lambda st, a=args: _lib.check(L.pvx_subm_rulebook(*a, st), "pvx_subm_rulebook")))
            if tuple(conv.kernel_size) == (3, 3, 3):       # a tile book wanted later is written by this very launch (add_conv)
                self._subm_ops[nbr.data_ptr()] = (len(self.ops) - 1, args)
            return nbr

        def add_conv(spec, level_in, level_out, feat_in, cin, nbr, residual_buf, avoid):
            conv = spec.conv
            cout = conv.out_channels
            kvol = int(np.prod(conv.kernel_size))
            pw = self._prep_weight(conv, cin)
            bias = conv.bias.detach().float().to(dev).contiguous() if conv.bias is not None else None
            scale, shift = _fold_bn(spec.bn, cout, dev)
            out = self._feat_buf(level_out, cout, avoid=avoid)
            keep.extend([pw, bias, scale, shift])
            # tile book of this rulebook for this kernel's window size: built once per (rulebook, window) on the geometry stream,
            # shared by every conv layer that reads the rulebook with that window (csrc/sparse_conv_slab_tb.h)
            rcap = L.pvx_tilebook_rcap(cin, cout, kvol, self.code) if self._wants_tilebook(cin, cout) else 0
            book = None
            if rcap > 0:
                bkey = (nbr.data_ptr(), rcap)
                book = self._books.get(bkey)
                if book is None:
                    book = torch.empty(L.pvx_tilebook_bytes(level_out.cap), dtype=torch.uint8, device=dev)
                    self._books[bkey] = book
                    fuse = self._subm_ops.pop(nbr.data_ptr(), None) if os.environ.get("PVX_ENGINE_FUSED_BOOK", "1") != "0" else None
                    if fuse is not None:
                        # the submanifold rulebook's own launch writes the book too (pvx_subm_rulebook_tb): no second pass over nbr
                        at, ra = fuse
                        af = ra[:7] + (ra[8], ra[9], rcap, _lib.ptr(book))
                        self.ops[at] = ("subm_rulebook", lambda st, a=af: _lib.check(L.pvx_subm_rulebook_tb(*a, st), "pvx_subm_rulebook_tb"))
                        self._fused_books[nbr.data_ptr()] = (at, af, rcap)
                    else:
                        ab = (_lib.ptr(nbr), kvol, level_out.cap, _lib.ptr(level_out.n_dev), level_in.cap, rcap, _lib.ptr(book))
                        self.ops.append(("tilebook", lambda st, a=ab: _lib.check(L.pvx_tilebook_build(*a, st), "pvx_tilebook_build")))
                    # tiles dealt to the kernel's workgroups by cost, for the kernels that walk such a schedule (after the book's builder,
                    # which runs earlier in self.ops whichever branch built it)
                    if self.schedule and L.pvx_tilebook_schedule(None, level_out.cap, None, cin, cout, kvol, self.code, None) == 1:
                        asd = (_lib.ptr(book), level_out.cap, _lib.ptr(level_out.n_dev), cin, cout, kvol, self.code)
                        self.ops.append(("tile_schedule", lambda st, a=asd: _lib.check(L.pvx_tilebook_schedule(*a, st), "pvx_tilebook_schedule")))
            args = (_lib.ptr(feat_in), level_in.cap, cin, _lib.ptr(pw), _lib.ptr(bias), _lib.ptr(nbr), kvol, level_out.cap,
                    _lib.ptr(level_out.n_dev), cout, self.code, _lib.ptr(scale), _lib.ptr(shift),
                    _lib.ptr(residual_buf) if residual_buf is not None else None, 1 if spec.relu else 0, _lib.ptr(out),
                    _lib.ptr(book), rcap)
            name = "conv%d_%dx%d_k%d" % (len(self.conv_records), cin, cout, kvol)
            self.ops.append((name, lambda st, a=args: _lib.check(L.pvx_sparse_conv_fwd_tb(*a, st), "pvx_sparse_conv_fwd_tb")))
            self.conv_records.append(dict(name=name, cin=cin, cin_real=conv.in_channels, cout=cout, kvol=kvol, nbr=nbr,
                                          level_in=level_in, level_out=level_out, subm=conv.subm, residual=residual_buf is not None,
                                          op=len(self.ops) - 1, args=args, rcap=rcap, book=book, far_only=False))
            return out, cout

        # site index of the input level
        self._n_unique0 = torch.zeros(1, dtype=torch.int32, device=dev)
        a0 = (_lib.ptr(self.vox_coords), lvl.cap, _lib.ptr(lvl.n_dev), B, lvl.c_shape, _lib.ptr(lvl.index), _lib.ptr(lvl.perm),
              _lib.ptr(self._n_unique0), _lib.ptr(lvl.ws), lvl.ws.numel())
        build0 = L.pvx_site_index_build_clean if self.clean_index else L.pvx_site_index_build
        self.ops.append(("site_index", lambda st, a=a0: _lib.check(build0(*a, st), "pvx_site_index_build")))

        def add_unset(level, coords, name="index_unset"):
            if self.clean_index:
                au = (_lib.ptr(coords), level.cap, _lib.ptr(level.n_dev), B, level.c_shape, _lib.ptr(level.index))
                self.ops.append((name, lambda st, a=au: _lib.check(L.pvx_site_index_unset(*a, st), "pvx_site_index_unset")))

        if self.sort_input:
            a0s = (_lib.ptr(self.vox_coords), lvl.cap, _lib.ptr(lvl.n_dev), B, lvl.c_shape, _lib.ptr(lvl.index),
                   _lib.ptr(self.vox_feat), row_bytes0, _lib.ptr(lvl.coords), _lib.ptr(self.feat0), _lib.ptr(self.rank0))
            self.ops.append(("sort_sites", lambda st, a=a0s: _lib.check(L.pvx_sort_sites(*a, st), "pvx_sort_sites")))

        for stage in stages:
            for item in stage:
                if item[0] == 'conv':
                    spec = item[1]
                    conv = spec.conv
                    assert not conv.inverse
                    if conv.subm:
                        nbr = add_subm_rulebook(cur, conv)
                        cur_feat, cin_cur = add_conv(spec, cur, cur, cur_feat, cin_cur, nbr, None, avoid=(cur_feat,))
                    else:
                        k3, s3, p3, d3 = (_lib.int3(conv.kernel_size), _lib.int3(conv.stride), _lib.int3(conv.padding),
                                          _lib.int3(conv.dilation))
                        out3 = (ctypes.c_int * 3)()
                        _lib.check(L.pvx_conv_out_shape(cur.c_shape, k3, s3, p3, d3, out3), "pvx_conv_out_shape")
                        cells = B * out3[0] * out3[1] * out3[2]
                        worst = int(np.prod([-(-k // s) for k, s in zip(conv.kernel_size, conv.stride)]))
                        f = worst if factor is None else min(float(factor), worst)
                        nxt = _Level([out3[0], out3[1], out3[2]], max(1, min(int(cur.cap * f), cells)), B, dev, with_perm=False)
                        kvol = int(np.prod(conv.kernel_size))
                        nbr = torch.empty((kvol, nxt.cap), dtype=torch.int32, device=dev)
                        keep.extend([k3, s3, p3, d3, nbr])
                        a1 = (_lib.ptr(cur.coords), cur.cap, _lib.ptr(cur.n_dev), B, cur.c_shape, k3, s3, p3, d3,
                              _lib.ptr(nxt.index), _lib.ptr(nxt.n_dev), _lib.ptr(nxt.ws), nxt.ws.numel())
                        a2 = (_lib.ptr(nxt.index), nxt.cap, _lib.ptr(nxt.n_dev), B, cur.c_shape, k3, s3, p3, d3,
                              _lib.ptr(cur.index), _lib.ptr(cur.perm), _lib.ptr(nxt.coords), _lib.ptr(nbr))
                        out_index = L.pvx_conv_out_index_clean if self.clean_index else L.pvx_conv_out_index
                        self.ops.append(("conv_out_index", lambda st, a=a1: _lib.check(out_index(*a, st), "pvx_conv_out_index")))
                        # a strided conv whose kernel has the tile-book form: the rulebook launch writes the book too, and (one reader)
                        # only the book's far entries of the int32 rulebook
                        rcap_s = 0
                        if tuple(conv.kernel_size) == (3, 3, 3) and conv.dilation[2] == 1 and self._wants_tilebook(cin_cur, conv.out_channels) \
                                and os.environ.get("PVX_ENGINE_FUSED_BOOK", "1") != "0":
                            rcap_s = L.pvx_tilebook_rcap(cin_cur, conv.out_channels, kvol, self.code)
                        if rcap_s > 0:
                            book_s = torch.empty(L.pvx_tilebook_bytes(nxt.cap), dtype=torch.uint8, device=dev)
                            self._books[(nbr.data_ptr(), rcap_s)] = book_s
                            far_s = self.far_only and bool(L.pvx_sparse_conv_book_form(cur.cap, cin_cur, conv.out_channels, kvol, nxt.cap,
                                                                                       self.code, rcap_s))
                            a2b = a2 + (cur.cap, rcap_s, _lib.ptr(book_s), 1 if far_s else 0)
                            self.ops.append(("conv_out_rulebook",
                                             lambda st, a=a2b: _lib.check(L.pvx_conv_out_book(*a, st), "pvx_conv_out_book")))
                        else:
                            self.ops.append(("conv_out_rulebook",
                                             lambda st, a=a2: _lib.check(L.pvx_conv_out_rulebook(*a, st), "pvx_conv_out_rulebook")))
                        # the input level's index has had its last reader (its submanifold rulebooks were built before this conv)
                        add_unset(cur, cur.coords)     # (level 0: the rows in key order after pvx_sort_sites mark the same cells)
                        cur_feat, cin_cur = add_conv(spec, cur, nxt, cur_feat, cin_cur, nbr, None, avoid=())
                        if rcap_s > 0 and far_s:
                            r = self.conv_records[-1]
                            assert r["rcap"] == rcap_s and r["book"] is book_s
                            r["far_only"] = True
                            self.ops[r["op"]] = (r["name"], lambda st, a=r["args"]: _lib.check(L.pvx_sparse_conv_fwd_book(*a, st),
                                                                                               "pvx_sparse_conv_fwd_book"))
                        self.levels.append(nxt)
                        cur = nxt
                else:   # residual block
                    _, s1, s2 = item
                    nbr = add_subm_rulebook(cur, s1.conv)
                    identity = cur_feat
                    mid, c_mid = add_conv(s1, cur, cur, cur_feat, cin_cur, nbr, None, avoid=(identity,))
                    cur_feat, cin_cur = add_conv(s2, cur, cur, mid, c_mid, nbr, identity, avoid=(identity, mid))
            self.stage_outputs.append((cur, cur_feat, cin_cur))
            # a stage output handed to the caller must survive the following launches of the same level
            cur.bufs[cin_cur] = [b for b in cur.bufs.get(cin_cur, []) if b is not cur_feat]
        self.out_level, self.out_feat, self.out_channels = cur, cur_feat, cin_cur
        if self.want_dense:
            self.dense = torch.empty((B, self.out_channels, *cur.shape3), dtype=self.dtype, device=dev)
            ad = (_lib.ptr(cur_feat), _lib.ptr(cur.coords), cur.cap, _lib.ptr(cur.n_dev), self.out_channels, self.code, B,
                  cur.c_shape, _lib.ptr(self.dense))
            # the strided chain leaves its rows in key order: the line form writes the tensor once, zeros included (PVX_ENGINE_DENSE_LINES=0:
            # zero-fill + scatter, the A/B switch and the fall-back for shapes the line form does not take)
            w, words = cur.shape3[2], self.out_channels // 2
            lines_ok = (os.environ.get("PVX_ENGINE_DENSE_LINES", "1") != "0" and self.dtype in (torch.bfloat16, torch.float16)
                        and self.out_channels % 2 == 0 and w % 4 == 0 and (w + w * ((words // (4 if words % 4 == 0 else 2 if words % 2 == 0 else 1)) | 1)) * 4 <= 64 * 1024 - 16
                        and len(self.levels) > 1)
            if lines_ok:
                al = (_lib.ptr(cur_feat), _lib.ptr(cur.index), cur.cap, _lib.ptr(cur.n_dev), self.out_channels, self.code, B,
                      cur.c_shape, _lib.ptr(self.dense))
                self.ops.append(("dense", lambda st, a=al: _lib.check(L.pvx_sparse_to_dense_sorted(*a, st),
                                                                      "pvx_sparse_to_dense_sorted")))
            else:
                self.ops.append(("dense", lambda st, a=ad: _lib.check(L.pvx_sparse_to_dense(*a, st), "pvx_sparse_to_dense")))
        # the last level's index is read by the line-form dense pass on the main stream: its un-set follows there
        add_unset(cur, cur.coords, name="index_unset_main")
        def book_form(r):
            return bool(L.pvx_sparse_conv_book_form(r["level_in"].cap, r["cin"], r["cout"], r["kvol"], r["level_out"].cap, self.code, r["rcap"]))

        if self.far_only:
            # rulebooks written together with their book (pvx_subm_rulebook_tb) whose every reader uses that very book
            for ptr, (at, af, rcap) in self._fused_books.items():
                readers = [r for r in self.conv_records if r["nbr"].data_ptr() == ptr]
                if readers and all(r["rcap"] == rcap and r["book"] is self._books[(ptr, rcap)] and book_form(r) for r in readers):
                    self.ops[at] = ("subm_rulebook", lambda st, a=af: _lib.check(L.pvx_subm_book(*a, st), "pvx_subm_book"))
                    for r in readers:
                        r["far_only"] = True
                        self.ops[r["op"]] = (r["name"], lambda st, a=r["args"]: _lib.check(L.pvx_sparse_conv_fwd_book(*a, st),
                                                                                           "pvx_sparse_conv_fwd_book"))
        torch.cuda.synchronize(dev)

    # ------------------------------------------------------------------------------------------------------------
    def load_points(self, frames):
        """frames: list of B [n_b, C] float32 tensors (n_b <= max_points_per_frame); unused rows are NaN = dropped."""
        assert len(frames) == self.B
        for b, f in enumerate(frames):
            n = f.shape[0]
            assert n <= self.n_pts_cap and f.shape[1] == self.C
            self.points[b, :n].copy_(f, non_blocking=True)
            if n < self.n_pts_cap:
                self.points[b, n:].fill_(float('nan'))

    def voxelize(self, st=None):
        """all B frames in one launch sequence: voxels of frame b follow those of frame b-1, batch index in column 0"""
        L = self.L
        st = _lib.stream() if st is None else st
        lvl = self.levels[0]
        _lib.check(L.pvx_voxelize_mean_batch(_lib.ptr(self.points), self.B, self.n_pts_cap, self.C, self.vsize, self.range,
                                             self.T, self.max_voxels, lvl.cap, _lib.ptr(self.vox_coords), _lib.ptr(self.vox_feat),
                                             self.cin0, self.code, _lib.ptr(self.num_voxels), _lib.ptr(lvl.n_dev),
                                             _lib.ptr(self.vox_ws), self.vox_ws.numel(), st), "pvx_voxelize_mean_batch")

    def _begin_ops(self):
        """the clean-index protocol (level indices zero between forwards) holds only if every forward enqueued ALL its launches, the
        un-sets included: a forward that raised half-way leaves marked words behind, so the next one zero-fills the indices first"""
        if self.clean_index and getattr(self, "_ops_open", False):
            torch.cuda.synchronize(self.device)
            for lvl in self.levels:
                lvl.index.zero_()
            torch.cuda.synchronize(self.device)
        self._ops_open = True

    def _end_ops(self):
        self._ops_open = False

    def backbone(self, st=None, hook=None):
        st = _lib.stream() if st is None else st
        self._begin_ops()
        if hook is None:
            for _, fn in self.ops:
                fn(st)
        else:
            for name, fn in self.ops:
                hook(name, lambda: fn(st))
        self._end_ops()

    _GEOMETRY = ("site_index", "sort_sites", "subm_rulebook", "conv_out_index", "conv_out_rulebook", "tilebook", "tile_schedule", "index_unset")

    def voxelize_ahead(self, stream, load=None, after=None):
        """Voxelize the NEXT batch of this plan on `stream` while another plan's rulebooks are being built: waits for this
        plan's previous forward (whose convs read the level-0 rows and counts), calls `load()` (which copies the frames into
        `self.points` on `stream`), voxelizes, and records the event that `forward(prevoxelized=True)` waits for.  The
        voxelizer is bound by atomic round trips and the rulebook kernels by bandwidth, so the two overlap well."""
        with torch.cuda.device(self.device):
            self._geo_stream()
            if getattr(self, "_vox_done", None) is None:
                self._vox_done = torch.cuda.Event()
            with torch.cuda.stream(stream):
                stream.wait_event(self._done)
                if after is not None:
                    stream.wait_event(after)      # e.g. another plan's `mark` event: start beside a chosen conv layer
                if load is not None:
                    load()
                self.voxelize(stream.cuda_stream)
                self._vox_done.record(stream)

    def forward(self, overlap=True, pipelined=False, prevoxelized=False, mark_after=None, mark2_after=None):
        """voxelize the loaded frames and run the backbone; returns the dense BEV tensor [B, C*D, H, W] (a view of a
        static buffer) or None.  No host synchronisation.

        overlap: the GEOMETRY of a frame (voxel coordinates, site indices, rulebooks of every level) depends on coordinates
        only, never on features, so it runs ahead on its own HIP stream while the convolutions of the earlier levels occupy
        the main stream; a conv waits (stream event) for the last geometry launch that precedes it in plan order.  The conv
        kernels are latency-bound and leave HBM mostly idle, which is what the bandwidth-bound
        geometry kernels use.
        pipelined: for a serving loop that alternates TWO plans (frames i, i+2, ... on one; i+1, i+3, ... on the other): the
        geometry stream then does not wait for the main stream (so the geometry of frame i+1 runs beside the convolutions of
        frame i) but only for this plan's own previous forward; the caller loads `points` on `geometry_stream()`.
        prevoxelized: `voxelize_ahead()` has already voxelized this batch (on a third stream, beside the previous batch's
        rulebook kernels); the geometry stream only waits for it."""
        with torch.cuda.device(self.device):
            main = torch.cuda.current_stream(self.device)
            if not overlap:
                st = main.cuda_stream
                self.voxelize(st)
                self.backbone(st)
            else:
                geo = self._geo_stream()
                g, m = geo.cuda_stream, main.cuda_stream
                if prevoxelized:
                    geo.wait_event(self._vox_done)        # voxelize_ahead() ran after this plan's previous forward
                else:
                    if pipelined:
                        geo.wait_event(self._done)        # the convs of this plan's previous batch read the rulebooks
                    else:
                        geo.wait_stream(main)             # the frames were loaded on the main stream
                    self.voxelize(g)
                pending = self._geo_events[0]
                pending.record(geo)
                self._begin_ops()
                i = 1
                n_conv = 0
                if mark_after is not None and getattr(self, "mark", None) is None:
                    self.mark = torch.cuda.Event()
                if mark2_after and getattr(self, "mark2", None) is None:
                    self.mark2 = torch.cuda.Event()      # a second marker for another consumer (the NMS stream)
                for name, fn in self.ops:
                    if name in self._GEOMETRY:
                        fn(g)
                        pending = self._geo_events[i]
                        pending.record(geo)
                        i += 1
                    else:
                        if pending is not None:
                            main.wait_event(pending)
                            pending = None
                        rec = getattr(self, "conv_events", None)
                        if rec is not None and name.startswith("conv"):     # measurement only: the launch's duration INSIDE the overlapped schedule
                            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            a.record(main)
                            fn(m)
                            b.record(main)
                            rec.append((name, a, b))
                        else:
                            fn(m)
                        if (mark_after is not None or mark2_after) and name.startswith("conv"):
                            n_conv += 1
                            if n_conv == mark_after:
                                self.mark.record(main)    # the mark_after-th conv launch of this forward has been enqueued
                            if n_conv == mark2_after:
                                self.mark2.record(main)
                if pending is not None:
                    main.wait_event(pending)
                self._end_ops()
                self._done.record(main)
        if self.want_dense:
            d = self.dense
            return d.view(d.shape[0], d.shape[1] * d.shape[2], d.shape[3], d.shape[4])
        return None

    def _geo_stream(self, preset=None):
        if getattr(self, "_geo_events", None) is None:
            n = 1 + sum(1 for name, _ in self.ops if name in self._GEOMETRY)
            self._geo_events = [torch.cuda.Event() for _ in range(n)]
            self._done = torch.cuda.Event()
            self._done.record(torch.cuda.current_stream(self.device))
            self._geo = None
        if preset is not None:
            self._geo = preset
        elif self._geo is None:
            self._geo = torch.cuda.Stream(device=self.device)
        return self._geo

    def geometry_stream(self):
        with torch.cuda.device(self.device):
            return self._geo_stream()

    def use_geometry_stream(self, stream):
        """run the geometry launches on `stream` instead of a stream of the plan's own (two alternating plans share one: a process has
        four hardware queues, and a schedule should not need more streams than that); call before the first forward(overlap=True)"""
        with torch.cuda.device(self.device):
            self._geo_stream(preset=stream)

    # ---- results (these DO synchronise: for tests / reporting, not for the hot loop) ---------------------------------
    def level_rows(self):
        return [int(l.n_dev.item()) for l in self.levels]

    def overflowed(self):
        """True when some level produced more rows than its capacity (rows were dropped)."""
        return any(int(l.n_dev.item()) > l.cap for l in self.levels)

    def stage_output(self, i):
        """(indices [n, 4], features [n, C]) of stage i (0 = conv_input, 1..4 = x_conv1..4, 5 = conv_out)"""
        lvl, feat, _ = self.stage_outputs[i]
        n = min(int(lvl.n_dev.item()), lvl.cap)
        if self.sort_input and lvl is self.levels[0]:      # level-0 rows live in key order: report them first-come, as spconv would
            r = self.rank0[:n].long()
            return lvl.coords[r], feat[r]
        return lvl.coords[:n], feat[:n]

    def conv_flops(self):
        """useful flops of one forward = sum over conv launches of 2 * P * Cin * Cout with P the number of valid
        (input, output, offset) pairs of the launch's rulebook restricted to live rows."""
        total, per = 0, []
        for r in self.conv_records:
            lvl = r["level_out"]
            n = min(int(lvl.n_dev.item()), lvl.cap)
            if r["far_only"]:    # the int32 rulebook holds the far entries only: count the book's codes (0xFFFF = no neighbour)
                tiles = -(-lvl.cap // 256)                           # the tile blocks come first, the schedule area follows them
                codes = r["book"][:tiles * 866 * 16].view(torch.int16).view(-1, 866 * 8)[:, 16:].reshape(-1, 27, 256)
                live = (torch.arange(codes.shape[0] * 256, device=codes.device) < n).view(-1, 1, 256)
                pairs = int(((codes != -1) & live).sum().item())
            else:
                pairs = int((r["nbr"][:, :n] >= 0).sum().item())
            f = 2 * pairs * r["cin_real"] * r["cout"]
            per.append((r["name"], n, pairs, f))
            total += f
        return total, per


class BatchedNMS:
    """All (frame, head) NMS problems of a batch in one launch with static buffers (SURVEY.md H7;
    pcdet/models/dense_heads/center_head.py:332-384 loops them in Python around
    pcdet/models/model_utils/model_nms_utils.py:6-25)."""

    def __init__(self, segments, cap, thresh, post_max, device="cuda", normal=False):
        self.S, self.cap, self.thresh, self.post_max, self.normal = int(segments), int(cap), float(thresh), int(post_max), int(normal)
        dev = torch.device(device)
        self.device = dev
        L = _lib.lib()
        from .ops.iou3d_nms import iou3d_nms_utils
        self._select = iou3d_nms_utils.nms_select
        # problems too large for the in-LDS sort (and PVX_NMS_TORCH_FRONT=1, a diagnostic) sort and gather with torch
        self.fused = self.cap <= iou3d_nms_utils.NMS_SELECT_MAX and not os.environ.get("PVX_NMS_TORCH_FRONT")
        self.num = torch.zeros(self.S, dtype=torch.int32, device=dev)
        if self.fused:
            self.sel = torch.zeros((self.S, self.post_max), dtype=torch.int64, device=dev)
            self.ws = _lib.workspace(L.pvx_nms_select_workspace_bytes(self.cap, self.S), dev)
        else:
            self.sorted_boxes = torch.zeros((self.S, self.cap, 7), dtype=torch.float32, device=dev)
            self.keep = torch.zeros((self.S, self.cap), dtype=torch.int64, device=dev)
            self.ws = _lib.workspace(L.pvx_nms_workspace_bytes(self.cap, self.S), dev)
            self._idx = torch.arange(self.cap, device=dev).unsqueeze(0)

    def __call__(self, boxes, scores, counts):
        """boxes [S, cap, >=7], scores [S, cap], counts [S] int32 -> (sel [S, post_max] int64 indices into each
        segment's boxes in descending score order, num [S] int32 = min(kept, post_max)); no host sync.  Sort, gather,
        NMS and the selection run in pvx_nms_select_batched's three launches into buffers owned here."""
        if self.fused:
            return self._select(boxes, scores, counts, self.thresh, self.normal, post_max=self.post_max,
                                out=(self.sel, self.num), ws=self.ws)
        S, cap = self.S, self.cap
        masked = scores.masked_fill(self._idx >= counts.view(-1, 1), float('-inf'))   # scalar as a kernel argument: no H2D copy
        order = masked.sort(dim=1, descending=True, stable=True)[1]
        torch.gather(boxes[..., :7], 1, order.unsqueeze(-1).expand(S, cap, 7), out=self.sorted_boxes)
        L = _lib.lib()
        with torch.cuda.device(self.device):
            _lib.check(L.pvx_nms_batched(_lib.ptr(self.sorted_boxes), _lib.ptr(counts), S, cap, self.thresh, self.normal,
                                         _lib.ptr(self.keep), _lib.ptr(self.num), _lib.ptr(self.ws), self.ws.numel(),
                                         _lib.stream()), "pvx_nms_batched")
        k = self.keep[:, :self.post_max]
        valid = self._idx[:, :self.post_max] < self.num.view(-1, 1)
        sel = torch.gather(order, 1, torch.where(valid, k, torch.zeros_like(k)))
        return sel, torch.clamp(self.num, max=self.post_max)
