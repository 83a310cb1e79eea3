from .anchors import create_anchors_3d_range, generate_task_anchors
from .pool import PoolFrames, DeviceSweepLoader, CameraLidarSweepLoader
from .file_loader import FileSweepLoader, SweepFileReader
from .camera_files import CameraLidarFileLoader, ImageAugTest
from .pipelines import (PIPELINES, Compose, LoadPointCloudFromFile, LoadPointCloudAnnotations, Preprocess,
                        Voxelization, AssignTarget, Reformat, SweepDataset, collate_device)

__all__ = ["create_anchors_3d_range", "generate_task_anchors", "PoolFrames", "DeviceSweepLoader", "CameraLidarSweepLoader", "FileSweepLoader", "SweepFileReader", "CameraLidarFileLoader", "ImageAugTest", "PIPELINES",
           "Compose", "LoadPointCloudFromFile", "LoadPointCloudAnnotations", "Preprocess", "Voxelization",
           "AssignTarget", "Reformat", "SweepDataset", "collate_device"]
