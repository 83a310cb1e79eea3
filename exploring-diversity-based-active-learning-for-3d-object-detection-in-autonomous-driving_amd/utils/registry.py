"""Name -> class registries with ``type=`` dispatch.

Mirrors the plugin contract of the reference (det3d/utils/registry.py:6-78):
``@REG.register_module`` on a class, ``build_from_cfg(cfg, REG, default_args)``
pops ``type`` (a registered name or a class), fills missing kwargs from
``default_args`` and instantiates.  Error types match the reference
(KeyError for duplicates / unknown names, TypeError for non-classes).
"""
import inspect


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    def __repr__(self):
        return f"{type(self).__name__}(name={self._name}, items={list(self._module_dict)})"

    def __contains__(self, key):
        return key in self._module_dict

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def register_module(self, cls):
        if not inspect.isclass(cls):
            raise TypeError(f"module must be a class, but got {type(cls)}")
        key = cls.__name__
        if key in self._module_dict:
            raise KeyError(f"{key} is already registered in {self._name}")
        self._module_dict[key] = cls
        return cls


def build_from_cfg(cfg, registry, default_args=None):
    assert isinstance(cfg, dict) and "type" in cfg
    assert default_args is None or isinstance(default_args, dict)
    kwargs = dict(cfg)
    obj_type = kwargs.pop("type")
    if isinstance(obj_type, str):
        cls = registry.get(obj_type)
        if cls is None:
            raise KeyError(f"{obj_type} is not in the {registry.name} registry")
    elif inspect.isclass(obj_type):
        cls = obj_type
    else:
        raise TypeError(f"type must be a str or valid type, but got {type(obj_type)}")
    for k, v in (default_args or {}).items():
        kwargs.setdefault(k, v)
    return cls(**kwargs)
