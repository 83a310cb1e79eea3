from .registry import (BACKBONES, DETECTORS, ESTIMATORS, HEADS, LOSSES, NECKS, READERS,
                       ROI_EXTRACTORS, SHARED_HEADS)
from .builder import build_backbone, build_detector, build_head, build_neck, build_reader
from .readers import VoxelFeatureExtractorV3
from .backbones import FPNSpMiddleResNetFHD, SparseTensor
from .necks import RPN
from .bbox_heads import Head, MultiGroupHead
from .detectors import FPNVoxelNet, VoxelNet
from .box_coder import GroundBox3dCoderTorch, build_box_coder
from .bevfusion_camera import ConvFuser, DepthLSSTransform, GeneralizedLSSFPN, LSSViewTransform
from .transfusion_head import TransFusionHead
from .swin import SwinTransformer
from .bevfusion_model import BEVFusion, BEVFusionCameraLidar

__all__ = ["READERS", "BACKBONES", "NECKS", "HEADS", "DETECTORS", "build_detector",
           "build_reader", "build_backbone", "build_neck", "build_head", "build_box_coder"]
