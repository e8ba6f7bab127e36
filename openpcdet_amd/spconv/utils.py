"""spconv.utils: the voxelizer classes pcdet/datasets/processor/data_processor.py:19-43 looks for."""
from ..voxelize import Point2VoxelCPU3d  # noqa: F401
from ..voxelize import Point2VoxelCPU3d as Point2VoxelGPU3d  # noqa: F401
