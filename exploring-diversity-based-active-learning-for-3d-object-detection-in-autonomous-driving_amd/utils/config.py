"""Python-file configs -> attribute dicts.

Same contract as the reference's det3d/torchie/utils/config.py:55-145: a
``.py`` config is executed as a module, its public globals become the config
dict; ``_base_`` lists files to inherit from; nested dicts allow attribute
access.  json/yaml configs load through ``fileio``.
"""
import os.path as osp
import runpy

from . import fileio

BASE_KEY = "_base_"
DELETE_KEY = "_delete_"


class ConfigDict(dict):
    """dict with attribute access (missing attribute -> AttributeError)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(f"'ConfigDict' object has no attribute '{name}'")

    def __setattr__(self, name, value):
        self[name] = value

    def update(self, *args, **kwargs):
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def copy(self):
        return ConfigDict(self)


class Config:
    @staticmethod
    def _file2dict(filename):
        filename = osp.abspath(osp.expanduser(filename))
        if not osp.isfile(filename):
            raise FileNotFoundError(f'file "{filename}" does not exist')
        if filename.endswith(".py"):
            ns = runpy.run_path(filename)
            cfg = {k: v for k, v in ns.items() if not k.startswith("__")}
        elif filename.endswith((".yml", ".yaml", ".json")):
            cfg = fileio.load(filename)
        else:
            raise IOError("Only py/yml/yaml/json type are supported now!")
        if BASE_KEY in cfg:
            bases = cfg.pop(BASE_KEY)
            bases = bases if isinstance(bases, list) else [bases]
            merged = {}
            for b in bases:
                c = Config._file2dict(osp.join(osp.dirname(filename), b))
                if merged.keys() & c.keys():
                    raise KeyError("Duplicate key is not allowed among bases")
                merged.update(c)
            Config._merge_a_into_b(cfg, merged)
            cfg = merged
        return cfg

    @staticmethod
    def _merge_a_into_b(a, b):
        for k, v in a.items():
            if isinstance(v, dict) and k in b and not v.pop(DELETE_KEY, False):
                if not isinstance(b[k], dict):
                    raise TypeError(f"Cannot inherit key {k} from base!")
                Config._merge_a_into_b(v, b[k])
            else:
                b[k] = v

    @staticmethod
    def fromfile(filename):
        return Config(Config._file2dict(filename), filename=filename)

    def __init__(self, cfg_dict=None, filename=None):
        cfg_dict = {} if cfg_dict is None else cfg_dict
        if not isinstance(cfg_dict, dict):
            raise TypeError(f"cfg_dict must be a dict, but got {type(cfg_dict)}")
        import types
        clean = {k: v for k, v in cfg_dict.items()
                 if not isinstance(v, (types.ModuleType, types.FunctionType, type))}
        object.__setattr__(self, "_cfg_dict", ConfigDict(clean))
        object.__setattr__(self, "_filename", filename)

    @property
    def filename(self):
        return self._filename

    def __repr__(self):
        return f"Config (path: {self._filename}): {dict.__repr__(self._cfg_dict)}"

    def __len__(self):
        return len(self._cfg_dict)

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setattr__(self, name, value):
        self._cfg_dict[name] = value

    def __setitem__(self, name, value):
        self._cfg_dict[name] = value

    def __contains__(self, name):
        return name in self._cfg_dict

    def __iter__(self):
        return iter(self._cfg_dict)
