from .anchors import create_anchors_3d_range, generate_task_anchors
from .pool import PoolFrames, DeviceSweepLoader

__all__ = ["create_anchors_3d_range", "generate_task_anchors", "PoolFrames", "DeviceSweepLoader"]
