"""TEST INFRASTRUCTURE ONLY.  Cross-check vectors for the rotated-box geometry of row a10 (rotated NMS).

The reference's test path suppresses with ``rotate_nms_cc`` -> ``rotate_non_max_suppression_cpu`` (det3d/ops/nms/nms_cpu.h:73-168,
compiled, boost::geometry: neither boost nor the built extension exists here, so that path stays unpinned).  The SAME tree
holds a second, self-contained implementation of the same geometry: det3d/ops/nms/nms_gpu.py:183-420 -- ``rbbox_to_corners``,
``quadrilateral_intersection``, ``sort_vertex_in_convex_polygon``, ``area``, ``inter``, ``devRotateIoU`` -- written as
numba.cuda DEVICE functions in Python.  numba is absent (an ordinary ModuleNotFoundError); with stand-ins that carry no
algorithm -- ``cuda.jit`` / ``numba.jit`` return the Python function unchanged, ``cuda.local.array`` hands out a zeroed numpy
array of the requested shape and dtype, ``numba.float32`` is numpy's, the compiled ``det3d.ops.nms.nms`` is an empty placeholder
that is never called -- those functions run here as plain Python in float32 arrays.  Their outputs on seeded box pairs (general
overlaps, identical boxes, containment, axis-aligned, shared edges, disjoint) are committed as
tests/golden/rotated_iou_pairs.npz: corners, intersection area and IoU of every pair.  tests/test_rotated_iou_golden.py holds
the oracle's geometry (al3d_oracle_rbox_pair = what al3d_oracle_rotate_nms uses) against them; on degenerate pairs (coincident
edges) the reference's vertex collection is unstable -- see that test's docstring -- and known answers are used instead.
Run in the build container only (needs /root/reference).
"""
import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.environ.get("AL3D_REFERENCE_ROOT", "/root/reference")


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


def _identity_decorator(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda f: f


def import_reference():
    if not os.path.isdir(os.path.join(ROOT, "det3d")):
        raise RuntimeError(f"reference tree not found at {ROOT}")
    local = types.SimpleNamespace(array=lambda shape, dtype: np.zeros(shape, dtype=dtype))
    cuda = _mod("numba.cuda", jit=_identity_decorator, local=local)
    _mod("numba", jit=_identity_decorator, njit=_identity_decorator, cuda=cuda, float32=np.float32, int32=np.int32)
    _pkg("det3d", os.path.join(ROOT, "det3d"))
    _pkg("det3d.utils", os.path.join(ROOT, "det3d", "utils"))
    _pkg("det3d.utils.buildtools", os.path.join(ROOT, "det3d", "utils", "buildtools"))
    _mod("det3d.utils.buildtools.pybind11_build", load_pb11=None)
    _pkg("det3d.ops", os.path.join(ROOT, "det3d", "ops"))
    _pkg("det3d.ops.nms", os.path.join(ROOT, "det3d", "ops", "nms"))
    _mod("det3d.ops.nms.nms", non_max_suppression=None)        # the compiled extension: a name, never called
    return importlib.import_module("det3d.ops.nms.nms_gpu")


def pairs(rng):
    out = []
    for _ in range(1200):                                       # general overlaps of car / truck / pedestrian sized boxes
        a = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(0.4, 5), rng.uniform(0.4, 11), rng.uniform(-np.pi, np.pi)])
        b = a + np.array([rng.normal(0, 1.2), rng.normal(0, 1.2), rng.normal(0, 0.4), rng.normal(0, 0.8), rng.normal(0, 0.5)])
        b[2:4] = np.abs(b[2:4]) + 0.2
        out.append((a, b))
    for _ in range(150):                                        # the same box twice, and a box inside another
        a = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(1, 4), rng.uniform(1, 8), rng.uniform(-np.pi, np.pi)])
        out.append((a, a.copy()))
        b = a.copy(); b[2:4] *= rng.uniform(0.2, 0.9)
        out.append((a, b))
    for _ in range(150):                                        # axis-aligned and right-angle pairs, shared edges
        a = np.array([rng.integers(-3, 4), rng.integers(-3, 4), 2.0, 4.0, rng.integers(0, 4) * np.pi / 2])
        b = np.array([a[0] + rng.integers(-2, 3), a[1] + rng.integers(-4, 5), 2.0, 4.0, rng.integers(0, 4) * np.pi / 2])
        out.append((a, b))
    for _ in range(100):                                        # far apart
        a = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), 2.0, 4.5, rng.uniform(-np.pi, np.pi)])
        b = a + np.array([30.0, -20.0, 0, 0, 1.0])
        out.append((a, b))
    return out


def main():
    ng = import_reference()
    rng = np.random.default_rng(2024)
    ps = pairs(rng)
    n = len(ps)
    A, B = np.zeros((n, 5), np.float32), np.zeros((n, 5), np.float32)
    ca, cb = np.zeros((n, 8), np.float32), np.zeros((n, 8), np.float32)
    inter, iou = np.zeros(n, np.float32), np.zeros(n, np.float32)
    for i, (a, b) in enumerate(ps):
        A[i], B[i] = a.astype(np.float32), b.astype(np.float32)
        ng.rbbox_to_corners(ca[i], A[i])                         # interleaved x0, y0, x1, y1, ...
        ng.rbbox_to_corners(cb[i], B[i])
        with np.errstate(all="ignore"):
            try:
                inter[i] = ng.inter(A[i], B[i])
                iou[i] = ng.devRotateIoU(A[i], B[i])
            except IndexError:
                # the reference collects up to 8 + 16 candidate vertices in a 16-float (8-point) local array: coincident
                # edges (the same box twice, shared edges) overflow it -- on the GPU a silent out-of-bounds write, here an
                # IndexError.  Such pairs carry NaN and are not compared.
                inter[i] = iou[i] = np.nan
    out = os.path.join(os.path.dirname(HERE), "tests", "golden", "rotated_iou_pairs.npz")
    np.savez_compressed(out, a=A, b=B, corners_a=ca, corners_b=cb, inter=inter, iou=iou)
    print("wrote", out, n, "pairs; overlapping:", int((inter > 0).sum()), "overflowed in the reference:", int(np.isnan(inter).sum()),
          "iou range", float(np.nanmin(iou)), float(np.nanmax(iou)))


if __name__ == "__main__":
    main()
