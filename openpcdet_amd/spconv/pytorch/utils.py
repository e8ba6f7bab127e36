"""spconv.pytorch.utils: the torch-tensor voxelizer (BASELINE.json config 1 names PointToVoxel)."""
from ...voxelize import PointToVoxel  # noqa: F401
