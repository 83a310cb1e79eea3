"""Selector plugins; same registry names and constructor kwargs as the reference
(det3d/selectors/__init__.py:1-24)."""
from .registry import SELECTORS
from .builder import build_selector
from .base_selector import BaseSelector
from .random_selector import RandomSelector
from .map_selectors import (SpatialSelector, TemporalSelector, EuSpatialSelector,
                            SpatialTemporalSelector)
from .feature_selectors import (FeatureSelector, SpatialFeatureSelector,
                                SpatialTemporalFeatureSelector)

from .uncertainty_selectors import EntropySelector, BadgeSelector, UWESelector, PPALSelector
from .cald_selector import CaldSelector

__all__ = ["EntropySelector", "BadgeSelector", "UWESelector", "PPALSelector", "CaldSelector", "BaseSelector", "RandomSelector", "SpatialSelector", "EuSpatialSelector",
           "TemporalSelector", "SpatialTemporalSelector", "FeatureSelector",
           "SpatialFeatureSelector", "SpatialTemporalFeatureSelector",
           "SELECTORS", "build_selector"]
