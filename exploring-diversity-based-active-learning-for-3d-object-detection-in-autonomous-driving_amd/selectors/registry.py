"""``SELECTORS`` registry (reference det3d/selectors/registry.py:1-4)."""
from ..utils import Registry

SELECTORS = Registry("selector")
