from .registry import Registry, build_from_cfg
from .config import Config, ConfigDict
from . import fileio
from .fileio import load, dump

__all__ = ["Registry", "build_from_cfg", "Config", "ConfigDict", "fileio", "load", "dump"]
