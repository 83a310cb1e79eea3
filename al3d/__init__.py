"""Importable alias for the hyphenated package directory.

The product lives in
``exploring-diversity-based-active-learning-for-3d-object-detection-in-autonomous-driving_amd/``
(a name python cannot import); ``import al3d`` resolves sub-modules from there.
"""
import os as _os

_real = _os.path.join(
    _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
    "exploring-diversity-based-active-learning-for-3d-object-detection-in-autonomous-driving_amd",
)
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
