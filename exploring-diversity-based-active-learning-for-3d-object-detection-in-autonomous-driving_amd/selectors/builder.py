"""``build_selector(cfg)`` (reference det3d/selectors/builder.py:8-9)."""
from ..utils import build_from_cfg
from .registry import SELECTORS


def build_selector(cfg):
    return build_from_cfg(cfg, SELECTORS)
