"""The point-level part of the reference's DataProcessor on the GPU: range mask, shuffle, voxelization straight from the frame's
point buffer -- what `pcdet/datasets/processor/data_processor.py:81-112,200-247` does with numpy in DataLoader workers before
the padded voxel tensors are pickled to the training process (32 MB per frame).  SURVEY.md section 8 row f3.

`DataProcessorGPU(processor_configs, point_cloud_range, training, num_point_features)` takes the same config list (dicts or
EasyDicts with NAME, and per step REMOVE_OUTSIDE_BOXES / SHUFFLE_ENABLED / VOXEL_SIZE / MAX_POINTS_PER_VOXEL /
MAX_NUMBER_OF_VOXELS) and `forward(data_dict)` rewrites `points` and adds `voxels`, `voxel_coords`, `voxel_num_points` as device
tensors.  Steps without a point-level meaning here (gt box filtering beyond the centre test, image steps) are passed through to an
optional CPU fallback processor."""
import numpy as np
import torch

from . import voxelize


def _get(cfg, name, default=None):
    return cfg.get(name, default) if isinstance(cfg, dict) else getattr(cfg, name, default)


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

    # data_processor.py:81-102
    def mask_points_and_boxes_outside_range(self, data_dict, config):
        pts = data_dict.get('points')
        if pts is not None:
            r = self.point_cloud_range
            mask = (pts[:, 0] >= float(r[0])) & (pts[:, 0] <= float(r[3])) & (pts[:, 1] >= float(r[1])) & (pts[:, 1] <= float(r[4]))
            data_dict['points'] = pts[mask]
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
            data_dict['points'] = pts[torch.randperm(pts.shape[0], device=pts.device, generator=self.generator)]
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
        for step, cfg in self.queue:
            data_dict = step(data_dict, cfg)
        return data_dict
