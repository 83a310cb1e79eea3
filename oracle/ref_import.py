"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Import pieces of the *reference* python package from /root/reference so that
golden vectors can be generated in this container.  The
reference never travels to the GPU box, so this module is used only by the
``oracle/gen_golden_*.py`` scripts; the fixtures they write under
``tests/golden/`` are what the test-suite consumes.

The reference cannot be imported as-is (missing addict / torchvision /
terminaltables / numba / spconv ...).  Those are ordinary ModuleNotFoundErrors,
not permission denials, so trivial stand-ins are registered in ``sys.modules``
before import.  The stand-ins carry no algorithmic content: an attribute dict,
empty modules, and identity decorators.
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("AL3D_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "det3d", "selectors"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


class _AttrDict(dict):
    """Stand-in for addict.Dict (only attribute access is needed)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _identity_decorator(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda f: f


_installed = False


def install_standins():
    """Register the stand-in modules (idempotent)."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    if "addict" not in sys.modules:
        _mod("addict", Dict=_AttrDict)
    if "torchvision" not in sys.modules:
        tv = _mod("torchvision")
        tv.models = _mod("torchvision.models")
        tv.models.resnet = _mod("torchvision.models.resnet")
    if "terminaltables" not in sys.modules:
        _mod("terminaltables", AsciiTable=object)
    if "numba" not in sys.modules:
        nb = _mod("numba", jit=_identity_decorator, njit=_identity_decorator,
                  prange=range)
        nb.cuda = _mod("numba.cuda", test=lambda *a, **k: None,
                       jit=_identity_decorator)
    _installed = True


def import_selectors():
    """Return the reference ``det3d.selectors`` sub-modules that run on CPU.

    ``det3d/selectors/__init__.py`` imports every selector (and through them the
    CUDA-only model stack); pre-seeding an empty package object for it lets the
    individual selector modules be imported one by one.
    """
    install_standins()
    import det3d  # noqa: F401  (version only)
    import det3d.torchie  # noqa: F401
    if "det3d.selectors" not in sys.modules or not hasattr(
            sys.modules["det3d.selectors"], "_al3d_seeded"):
        p = _pkg("det3d.selectors",
                 os.path.join(REFERENCE_ROOT, "det3d", "selectors"))
        p._al3d_seeded = True
    # the feature-family selectors take exactly one name from this module
    if "det3d.torchie.apis.train" not in sys.modules:
        if "det3d.torchie.apis" not in sys.modules:
            _pkg("det3d.torchie.apis",
                 os.path.join(REFERENCE_ROOT, "det3d", "torchie", "apis"))

        def example_to_device(example, device=None, non_blocking=False):
            return example

        _mod("det3d.torchie.apis.train", example_to_device=example_to_device)
    out = {}
    for name in ("base_selector", "spatial_temporal_selector", "spatial_selector",
                 "temporal_selector", "random_selector",
                 "euclidean_spatial_selector", "feature_selector",
                 "spatial_temporal_feature_selector", "spatial_feature_selector",
                 "entropy_selector"):
        try:
            out[name] = importlib.import_module(f"det3d.selectors.{name}")
        except Exception as e:  # ordinary import errors are reported, not hidden
            out[name] = e
    return out


def import_point_cloud_ops():
    install_standins()
    for pkg in ("det3d.ops", "det3d.ops.point_cloud"):
        if pkg not in sys.modules:
            _pkg(pkg, os.path.join(REFERENCE_ROOT, *pkg.split(".")))
    return importlib.import_module("det3d.ops.point_cloud.point_cloud_ops")


def import_box_torch_ops():
    install_standins()
    for pkg in ("det3d.core", "det3d.core.bbox", "det3d.ops", "det3d.ops.nms"):
        if pkg not in sys.modules:
            _pkg(pkg, os.path.join(REFERENCE_ROOT, *pkg.split(".")))
    return importlib.import_module("det3d.core.bbox.box_torch_ops")


def import_box_np_ops():
    install_standins()
    if "spconv" not in sys.modules:
        # box_np_ops only *names* two spconv.utils helpers at import time (box_np_ops.py:10);
        # the anchor / corner functions used for golden vectors never call them.
        sp = _mod("spconv")
        sp.utils = _mod("spconv.utils", rbbox_intersection=None, rbbox_iou=None)
    for pkg in ("det3d.core", "det3d.core.bbox", "det3d.ops", "det3d.ops.nms"):
        if pkg not in sys.modules:
            _pkg(pkg, os.path.join(REFERENCE_ROOT, *pkg.split(".")))
    return importlib.import_module("det3d.core.bbox.box_np_ops")


def import_rpn():
    """Reference dense neck (det3d/models/necks/rpn.py) on CPU torch."""
    install_standins()
    import det3d.torchie  # noqa: F401
    for pkg in ("det3d.models", "det3d.models.necks", "det3d.ops"):
        if pkg not in sys.modules:
            _pkg(pkg, os.path.join(REFERENCE_ROOT, *pkg.split(".")))
    if "det3d.ops.syncbn" not in sys.modules:
        import torch
        _mod("det3d.ops.syncbn", DistributedSyncBN=torch.nn.BatchNorm2d)
    return importlib.import_module("det3d.models.necks.rpn")
