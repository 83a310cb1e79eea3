"""json / yaml / pickle load+dump keyed by file extension.

Wire formats follow the reference's det3d/torchie/fileio/io.py:15-80 and its
handlers: json via ``json.dump`` (kwargs such as ``indent`` pass through),
pickle with protocol 2 by default, yaml via safe loader.
"""
import json
import pickle
from pathlib import Path

import yaml


def _fmt(file, file_format):
    if isinstance(file, Path):
        file = str(file)
    if file_format is None and isinstance(file, str):
        file_format = file.split(".")[-1]
    if file_format == "pickle":
        file_format = "pkl"
    if file_format == "yml":
        file_format = "yaml"
    if file_format not in ("json", "yaml", "pkl"):
        raise TypeError(f"Unsupported format: {file_format}")
    return file, file_format


def load(file, file_format=None, **kwargs):
    file, fmt = _fmt(file, file_format)
    if isinstance(file, str):
        with open(file, "rb" if fmt == "pkl" else "r") as f:
            return load(f, fmt, **kwargs)
    if not hasattr(file, "read"):
        raise TypeError('"file" must be a filepath str or a file-object')
    if fmt == "json":
        return json.load(file)
    if fmt == "yaml":
        return yaml.load(file, Loader=yaml.SafeLoader)
    return pickle.load(file, **kwargs)


def dump(obj, file=None, file_format=None, **kwargs):
    if file is None:
        if file_format is None:
            raise ValueError("file_format must be specified since file is None")
        _, fmt = _fmt("x." + file_format, None)
        if fmt == "json":
            return json.dumps(obj, **kwargs)
        if fmt == "yaml":
            return yaml.dump(obj, **kwargs)
        kwargs.setdefault("protocol", 2)
        return pickle.dumps(obj, **kwargs)
    file, fmt = _fmt(file, file_format)
    if isinstance(file, str):
        with open(file, "wb" if fmt == "pkl" else "w") as f:
            return dump(obj, f, fmt, **kwargs)
    if not hasattr(file, "write"):
        raise TypeError('"file" must be a filename str or a file-object')
    if fmt == "json":
        json.dump(obj, file, **kwargs)
    elif fmt == "yaml":
        yaml.dump(obj, file, **kwargs)
    else:
        kwargs.setdefault("protocol", 2)
        pickle.dump(obj, file, **kwargs)
