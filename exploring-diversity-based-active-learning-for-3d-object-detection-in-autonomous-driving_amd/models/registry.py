"""Model registries, same names as the reference (det3d/models/registry.py:1-11)."""
from ..utils import Registry

READERS = Registry("reader")
BACKBONES = Registry("backbone")
NECKS = Registry("neck")
ROI_EXTRACTORS = Registry("roi_extractor")
SHARED_HEADS = Registry("shared_head")
HEADS = Registry("head")
LOSSES = Registry("loss")
DETECTORS = Registry("detector")
ESTIMATORS = Registry("estimator")
