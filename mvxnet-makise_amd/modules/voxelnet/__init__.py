from .VoxelNet import VoxelNet  # noqa: F401
from .Loss import VoxelLoss  # noqa: F401
