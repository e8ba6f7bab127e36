"""The world augmentations of the reference's DataAugmentor on the GPU:
`pcdet/datasets/augmentor/data_augmentor.py:61-119,140-161,295-324` with `augmentor_utils.py:8-91`.

`DataAugmentorGPU(augmentor_configs)` takes the same AUG_CONFIG_LIST (dicts or EasyDicts, DISABLE_AUG_LIST honoured) and
`forward(data_dict)` rewrites `points` and `gt_boxes` and records `flip_x` / `flip_y` / `noise_rot` / `noise_scale` /
`noise_translate` like the reference.  The random numbers are drawn on the host from numpy's global generator with the very
calls, in the very order, of the reference -- seeded alike, both draw the same transformation -- and the whole list is then
applied in ONE pass over the points (`pvx_points_transform`) and one over the boxes (`pvx_boxes_transform`), instead of one
numpy pass per augmentation.

Steps that are not point-level world transforms (gt_sampling with its database and box collision test, the image steps, the
local per-box noise) have no kernel here: the constructor raises for them unless `host_steps` maps their NAME to a callable
(e.g. the bound method of a CPU DataAugmentor), which then runs at that place in the queue on host arrays."""
import ctypes

import numpy as np
import torch

from . import _lib

FLIP_X, FLIP_Y, ROTATE, SCALE, TRANSLATE, LIMIT_HEADING = range(6)
MAX_OPS = 16


class PointOp(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("a", ctypes.c_float), ("b", ctypes.c_float), ("c", ctypes.c_float)]


def _get(cfg, name, default=None):
    return cfg.get(name, default) if isinstance(cfg, dict) else getattr(cfg, name, default)


def _pack(ops):
    if len(ops) > MAX_OPS:
        raise ValueError("at most %d operations per call" % MAX_OPS)
    arr = (PointOp * max(len(ops), 1))()
    for i, (kind, a, b, c) in enumerate(ops):
        arr[i] = PointOp(int(kind), float(a), float(b), float(c))
    return arr


def rotate_op(angle):
    """(ROTATE, cos, sin, angle) with the float32 cosine / sine rotate_points_along_z computes (common_utils.py:43-47)"""
    a = torch.tensor([float(angle)], dtype=torch.float64).float()
    return (ROTATE, float(torch.cos(a)[0]), float(torch.sin(a)[0]), float(a[0]))


def limit_heading_op():
    """data_augmentor.py:309-311: heading into [-pi, pi)"""
    return (LIMIT_HEADING, float(np.float32(2 * np.pi)), 0.5, 0.0)


def transform_points(points, ops):
    """apply `ops` [(kind, a, b, c), ...] in order, in place, to columns 0..2 of the [N, 3 + C] float32 device tensor"""
    assert points.is_cuda and points.dtype == torch.float32 and points.dim() == 2 and points.is_contiguous()
    with torch.cuda.device(points.device):
        _lib.check(_lib.lib().pvx_points_transform(_lib.ptr(points), points.shape[0], points.shape[1], _pack(ops), len(ops),
                                                   _lib.stream()), "pvx_points_transform")
    return points


def transform_boxes(boxes, ops):
    """apply `ops` in order, in place, to the [N, 7 (+2 velocity ...)] float32 device tensor of boxes"""
    assert boxes.is_cuda and boxes.dtype == torch.float32 and boxes.dim() == 2 and boxes.is_contiguous()
    with torch.cuda.device(boxes.device):
        _lib.check(_lib.lib().pvx_boxes_transform(_lib.ptr(boxes), boxes.shape[0], boxes.shape[1], _pack(ops), len(ops),
                                                  _lib.stream()), "pvx_boxes_transform")
    return boxes


class DataAugmentorGPU:
    def __init__(self, augmentor_configs, device="cuda", host_steps=None):
        self.device = torch.device(device)
        cfgs = augmentor_configs if isinstance(augmentor_configs, list) else _get(augmentor_configs, 'AUG_CONFIG_LIST')
        disabled = [] if isinstance(augmentor_configs, list) else (_get(augmentor_configs, 'DISABLE_AUG_LIST') or [])
        host_steps = host_steps or {}
        self.queue = []                                  # (draw function or host callable, config, is_host)
        for cfg in cfgs:
            name = _get(cfg, 'NAME')
            if name in disabled:                         # data_augmentor.py:26-28
                continue
            if name in host_steps:
                self.queue.append((host_steps[name], cfg, True))
            elif name in ('random_world_flip', 'random_world_rotation', 'random_world_scaling', 'random_world_translation'):
                self.queue.append((getattr(self, '_draw_' + name[len('random_world_'):]), cfg, False))
            else:
                raise NotImplementedError("DataAugmentorGPU has no step %r (pass it in host_steps to run it on the host)" % name)

    # the draws: same numpy calls in the same order as the reference, so a seeded run picks the same transformation
    def _draw_flip(self, data_dict, config):
        ops = []
        for axis in _get(config, 'ALONG_AXIS_LIST'):                                            # data_augmentor.py:65-70
            assert axis in ['x', 'y']
            enable = np.random.choice([False, True], replace=False, p=[0.5, 0.5])               # augmentor_utils.py:15-16
            data_dict['flip_%s' % axis] = enable
            if enable:
                ops.append((FLIP_X if axis == 'x' else FLIP_Y, 0.0, 0.0, 0.0))
        return ops

    def _draw_rotation(self, data_dict, config):
        rot_range = _get(config, 'WORLD_ROT_ANGLE')                                             # data_augmentor.py:85-87
        if not isinstance(rot_range, (list, tuple)):
            rot_range = [-rot_range, rot_range]
        noise = np.random.uniform(rot_range[0], rot_range[1])                                   # augmentor_utils.py:58-59
        data_dict['noise_rot'] = noise
        return [rotate_op(noise)]

    def _draw_scaling(self, data_dict, config):
        lo, hi = _get(config, 'WORLD_SCALE_RANGE')
        if hi - lo < 1e-3:                                                                      # augmentor_utils.py:82-83
            return []
        noise = np.random.uniform(lo, hi)
        data_dict['noise_scale'] = noise
        return [(SCALE, noise, 0.0, 0.0)]

    def _draw_translation(self, data_dict, config):
        std = _get(config, 'NOISE_TRANSLATE_STD')                                               # data_augmentor.py:143-149
        assert len(std) == 3
        t = np.array([np.random.normal(0, std[0], 1), np.random.normal(0, std[1], 1), np.random.normal(0, std[2], 1)],
                     dtype=np.float32).T
        data_dict['noise_translate'] = t
        return [(TRANSLATE, t[0, 0], t[0, 1], t[0, 2])]

    def _apply(self, data_dict, ops):
        """points and gt_boxes (device tensors by now) through `ops`"""
        for i in range(0, len(ops), MAX_OPS):
            transform_points(data_dict['points'], ops[i:i + MAX_OPS])
            if data_dict.get('gt_boxes') is not None:
                transform_boxes(data_dict['gt_boxes'], ops[i:i + MAX_OPS])

    def _to_device(self, data_dict):
        for key in ('points', 'gt_boxes'):
            v = data_dict.get(key)
            if v is not None and not torch.is_tensor(v):
                data_dict[key] = torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(self.device)
            elif v is not None:
                data_dict[key] = v.to(self.device, torch.float32).contiguous()

    def forward(self, data_dict):
        """data_augmentor.py:295-324.  `points` comes back as a device tensor (the voxelizer's input); `gt_boxes` comes back as
        it came: a numpy array if it was one (its consumers are host code in the reference), else a device tensor."""
        if 'roi_boxes' in data_dict:
            raise NotImplementedError("roi_boxes (MPPNet's multi-frame proposals, data_augmentor.py:71-76) are not handled here")
        host_boxes = isinstance(data_dict.get('gt_boxes'), np.ndarray)
        ops = []
        on_device = False
        for step, cfg, is_host in self.queue:
            if is_host:
                if on_device:                            # flush what was drawn so far, hand host arrays to the host step
                    self._apply(data_dict, ops)
                    ops = []
                    for key in ('points', 'gt_boxes'):
                        if torch.is_tensor(data_dict.get(key)):
                            data_dict[key] = data_dict[key].cpu().numpy()
                    on_device = False
                data_dict = step(data_dict=data_dict)
            else:
                if not on_device:
                    self._to_device(data_dict)
                    on_device = True
                ops += step(data_dict, cfg)
        if not on_device:
            self._to_device(data_dict)
        if data_dict.get('gt_boxes') is not None:
            ops.append(limit_heading_op())                                                      # :309-311
        self._apply(data_dict, ops)
        data_dict.pop('road_plane', None)                                                       # :314-315
        if 'gt_boxes_mask' in data_dict:                                                        # :316-323
            mask = np.asarray(data_dict.pop('gt_boxes_mask'))
            data_dict['gt_boxes'] = data_dict['gt_boxes'][torch.as_tensor(mask, device=data_dict['gt_boxes'].device)]
            data_dict['gt_names'] = data_dict['gt_names'][mask]
            if 'gt_boxes2d' in data_dict:
                data_dict['gt_boxes2d'] = data_dict['gt_boxes2d'][mask]
        if host_boxes:
            data_dict['gt_boxes'] = data_dict['gt_boxes'].cpu().numpy()
        return data_dict
