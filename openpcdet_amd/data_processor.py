"""The point-level part of the reference's DataProcessor on the GPU: range mask, shuffle, voxelization straight from the frame's
point buffer -- what `pcdet/datasets/processor/data_processor.py:81-112,200-247` does with numpy in DataLoader workers before
the padded voxel tensors are pickled to the training process (32 MB per frame).  SURVEY.md section 8 row f3.

`DataProcessorGPU(processor_configs, point_cloud_range, training, num_point_features)` takes the same config list (dicts or
EasyDicts with NAME, and per step REMOVE_OUTSIDE_BOXES / SHUFFLE_ENABLED / VOXEL_SIZE / MAX_POINTS_PER_VOXEL /
MAX_NUMBER_OF_VOXELS) and `forward(data_dict)` rewrites `points` and adds `voxels`, `voxel_coords`, `voxel_num_points` as device
tensors.  The mask and the shuffle are one HIP pass (`pvx_points_prepare`, csrc/point_prep.hip: gather by a permutation + stable
compaction); the world augmentations that precede them in the reference's `prepare_data` are in data_augmentor.py.  Steps without
a point-level meaning here (image steps, corner-count box filtering) raise NotImplementedError."""
import ctypes

import numpy as np
import torch

from . import _lib, voxelize


def _get(cfg, name, default=None):
    return cfg.get(name, default) if isinstance(cfg, dict) else getattr(cfg, name, default)


def prepare_points(points, perm=None, range_xy=None, shuffle_seed=0):
    """rows points[perm] (perm: int64 device permutation; None = the keyed bijection `shuffle_seed` selects, computed in the kernel,
    or the identity for seed 0) inside x0 <= x <= x1, y0 <= y <= y1 (range_xy, None =
    all), in order: `pvx_points_prepare`, one gather + stable compaction for mask_points_by_range (common_utils.py:78-81) and
    shuffle_points (data_processor.py:104-112).  One host read of the count, as `points[mask]` has."""
    assert points.is_cuda and points.dtype == torch.float32 and points.dim() == 2 and points.is_contiguous()
    n, c = points.shape
    if perm is not None:
        assert perm.dtype == torch.int64 and perm.is_cuda and perm.is_contiguous() and perm.numel() == n
    L = _lib.lib()
    out = torch.empty_like(points)
    count = torch.empty(1, dtype=torch.int32, device=points.device)
    ws_bytes = L.pvx_points_prepare_workspace_bytes(n)
    ws = _lib.workspace(ws_bytes, points.device)
    rng = (ctypes.c_float * 4)(*[float(v) for v in range_xy]) if range_xy is not None else None
    with torch.cuda.device(points.device):
        _lib.check(L.pvx_points_prepare(_lib.ptr(points), n, c, _lib.ptr(perm), int(shuffle_seed) & (2 ** 64 - 1), rng, _lib.ptr(out), _lib.ptr(count), _lib.ptr(ws),
                                        ws_bytes, _lib.stream()), "pvx_points_prepare")
    if range_xy is None:
        return out                                     # nothing dropped: no host read
    return out[:int(count.item())]


class DataProcessorGPU:
    def __init__(self, processor_configs, point_cloud_range, training, num_point_features, device="cuda", generator=None):
        self.point_cloud_range = np.asarray(point_cloud_range, np.float32)
        self.training = bool(training)
        self.mode = 'train' if training else 'test'
        self.num_point_features = int(num_point_features)
        self.device = torch.device(device)
        self.generator = generator                     # torch.Generator on `device` for reproducible shuffles
        self.grid_size = self.voxel_size = None
        self.voxel_generator = None
        self.queue = []
        for cfg in processor_configs:
            name = _get(cfg, 'NAME')
            step = getattr(self, name, None)
            if step is None:
                raise NotImplementedError("DataProcessorGPU has no step %r" % name)
            self.queue.append((step, cfg))
            if name in ('transform_points_to_voxels', 'transform_points_to_voxels_placeholder'):
                vs = [float(v) for v in _get(cfg, 'VOXEL_SIZE')]
                g = (self.point_cloud_range[3:6] - self.point_cloud_range[0:3]) / np.array(vs)     # data_processor.py:115-118
                self.grid_size = np.round(g).astype(np.int64)
                self.voxel_size = vs

    def _shuffle_args(self, n):
        """how to shuffle n rows: with a torch generator, a drawn permutation (torch.randperm: random keys + a device sort,
        160 us for 300k rows); without, a seed drawn from numpy's global generator -- the one the reference's
        np.random.permutation (data_processor.py:107-110) uses -- for the kernel's own keyed bijection (no array, no sort)"""
        if self.generator is not None:
            return dict(perm=torch.randperm(n, device=self.device, generator=self.generator))
        return dict(shuffle_seed=int(np.random.randint(1, 2 ** 63 - 1, dtype=np.int64)))

    # data_processor.py:81-102
    def mask_points_and_boxes_outside_range(self, data_dict, config, shuffle=False):
        """shuffle=True folds the shuffle_points step that follows into the same pass: a uniform permutation of ALL rows followed
        by the order-keeping mask leaves the kept rows in uniformly random order, which is what shuffling after the mask does"""
        pts = data_dict.get('points')
        if pts is not None:
            r = self.point_cloud_range
            data_dict['points'] = prepare_points(pts, range_xy=(r[0], r[1], r[3], r[4]),
                                                  **(self._shuffle_args(pts.shape[0]) if shuffle else {}))
        boxes = data_dict.get('gt_boxes')
        if boxes is not None and _get(config, 'REMOVE_OUTSIDE_BOXES', False) and self.training:
            if not _get(config, 'USE_CENTER_TO_FILTER', True):
                raise NotImplementedError("corner-count filtering of gt boxes (box_utils.py:93-120) stays on the CPU")
            b = torch.as_tensor(boxes)
            r = self.point_cloud_range
            keep = ((b[:, :3] >= torch.tensor(r[:3], dtype=b.dtype, device=b.device)) &
                    (b[:, :3] <= torch.tensor(r[3:], dtype=b.dtype, device=b.device))).all(1)
            data_dict['gt_boxes'] = boxes[keep.cpu().numpy()] if isinstance(boxes, np.ndarray) else boxes[keep]
        return data_dict

    # data_processor.py:104-112
    def shuffle_points(self, data_dict, config):
        enabled = _get(config, 'SHUFFLE_ENABLED')
        if enabled[self.mode]:
            pts = data_dict['points']
            data_dict['points'] = prepare_points(pts, **self._shuffle_args(pts.shape[0]))
        return data_dict

    # data_processor.py:114-122
    def transform_points_to_voxels_placeholder(self, data_dict, config):
        return data_dict

    # data_processor.py:124-138
    @staticmethod
    def double_flip(points):
        """-> (y-flipped, x-flipped, xy-flipped) copies of a [N, C] device tensor"""
        y, x, xy = points.clone(), points.clone(), points.clone()
        y[:, 1] = -y[:, 1]
        x[:, 0] = -x[:, 0]
        xy[:, 0] = -xy[:, 0]
        xy[:, 1] = -xy[:, 1]
        return y, x, xy

    # data_processor.py:200-247
    def transform_points_to_voxels(self, data_dict, config):
        if self.voxel_generator is None:
            self.voxel_generator = voxelize.PointToVoxel(self.voxel_size, self.point_cloud_range.tolist(), self.num_point_features,
                                                         int(_get(config, 'MAX_NUMBER_OF_VOXELS')[self.mode]),
                                                         int(_get(config, 'MAX_POINTS_PER_VOXEL')))
        points = data_dict['points']
        lead = 0 if data_dict.get('use_lead_xyz', True) else 3       # :222-223 drops xyz from the voxel rows
        if _get(config, 'DOUBLE_FLIP', False):
            # the cloud and its three flipped copies (:225-242) are four frames of ONE batched voxelization; the lists keep the
            # reference's order [original, yflip, xflip, xyflip]
            out = self.voxel_generator.run_batch(torch.stack((points,) + self.double_flip(points)))
            counts = out['num_voxels'].tolist()                      # one host read for the four sizes
            data_dict['voxels'] = [out['voxels'][f, :n, :, lead:] for f, n in enumerate(counts)]
            data_dict['voxel_coords'] = [out['coords'][f, :n] for f, n in enumerate(counts)]
            data_dict['voxel_num_points'] = [out['num_points'][f, :n] for f, n in enumerate(counts)]
            return data_dict
        out = self.voxel_generator.run(points)
        n = int(out['num_voxels'].item())              # the reference hands on exactly-sized arrays
        data_dict['voxels'] = out['voxels'][:n, :, lead:]
        data_dict['voxel_coords'] = out['coords'][:n]
        data_dict['voxel_num_points'] = out['num_points'][:n]
        return data_dict

    def forward(self, data_dict):
        pts = data_dict.get('points')
        if pts is not None and not torch.is_tensor(pts):
            data_dict['points'] = torch.from_numpy(np.ascontiguousarray(pts, np.float32)).to(self.device)
        elif pts is not None:
            data_dict['points'] = pts.to(self.device, torch.float32).contiguous()
        q = self.queue
        i = 0
        while i < len(q):
            step, cfg = q[i]
            nxt = q[i + 1] if i + 1 < len(q) else None
            if (step == self.mask_points_and_boxes_outside_range and nxt is not None and nxt[0] == self.shuffle_points
                    and _get(nxt[1], 'SHUFFLE_ENABLED')[self.mode]):
                data_dict = step(data_dict, cfg, shuffle=True)       # mask + shuffle: one gather / compaction pass
                i += 2
                continue
            data_dict = step(data_dict, cfg)
            i += 1
        return data_dict
