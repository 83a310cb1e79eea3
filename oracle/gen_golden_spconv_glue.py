"""TEST INFRASTRUCTURE ONLY.  Golden vectors from the pure-Python glue of the reference's vendored spconv
(bevfusion/mmdet3d/ops/spconv), the two pieces of it that run without the compiled extension:

  * ``structure.py:5-63``: ``scatter_nd`` and ``SparseConvTensor.dense()`` (pure torch; imported BY FILE), followed by the
    ``[N, C, D, H, W] -> [N, C * D, H, W]`` view of the encoder's last line (det3d/models/backbones/scn.py:387-390): pins
    the layout ``al3d_sp_to_dense_nhwc`` writes (channel = c * D + z);
  * ``ops.py:19-32``: ``get_conv_output_size`` for the encoder's strided geometries (imported behind a content-free
    ``sparse_conv_ext`` module object -- ops.py does ``from . import sparse_conv_ext`` at module level and the function
    never touches it): pins the encoder's output shapes.

Nothing of the compiled rulebook / indice convolution is run (spconv 1.2.1 is absent, the vendored 1.0 needs CUDA headers):
the sparse convolution itself stays "parity unpinned" (DESIGN 3).  Run in the build container only (needs /root/reference);
writes tests/golden/spconv_glue.npz.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.environ.get("AL3D_REFERENCE_ROOT", "/root/reference")
SP = os.path.join(ROOT, "bevfusion", "mmdet3d", "ops", "spconv")


def _load(name, path, package=None):
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=None)
    mod = importlib.util.module_from_spec(spec)
    if package:
        mod.__package__ = package
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    structure = _load("al3d_ref_spconv_structure", os.path.join(SP, "structure.py"))
    # ops.py: `from . import sparse_conv_ext` -> a package shell holding an EMPTY module of that name
    pkg = types.ModuleType("al3d_ref_spconv_pkg")
    pkg.__path__ = [SP]
    sys.modules["al3d_ref_spconv_pkg"] = pkg
    sys.modules["al3d_ref_spconv_pkg.sparse_conv_ext"] = types.ModuleType("al3d_ref_spconv_pkg.sparse_conv_ext")
    ops = _load("al3d_ref_spconv_pkg.ops", os.path.join(SP, "ops.py"), package="al3d_ref_spconv_pkg")

    rng = np.random.default_rng(41)
    store = {}
    # ---- dense(): two cases (the encoder's last level is [N, 128, 2, H, W]; a tiny odd one)
    for case, (B, C, D, H, W, n) in {"small": (2, 8, 2, 6, 5, 23), "odd": (3, 4, 3, 5, 7, 61)}.items():
        cells = rng.choice(B * D * H * W, size=n, replace=False)
        b, r = np.divmod(cells, D * H * W)
        z, r = np.divmod(r, H * W)
        y, x = np.divmod(r, W)
        idx = np.stack([b, z, y, x], 1).astype(np.int32)
        feats = rng.normal(size=(n, C)).astype(np.float32)
        t = structure.SparseConvTensor(torch.from_numpy(feats), torch.from_numpy(idx), [D, H, W], B)
        dense = t.dense()                                            # [B, C, D, H, W]
        N_, C_, D_, H_, W_ = dense.shape
        bev = dense.view(N_, C_ * D_, H_, W_)                        # scn.py:387-390
        store[f"{case}.features"], store[f"{case}.indices"] = feats, idx
        store[f"{case}.shape"] = np.array([B, C, D, H, W])
        store[f"{case}.dense"] = dense.numpy()
        store[f"{case}.bev"] = bev.numpy()
        # scatter_nd itself, channels last
        store[f"{case}.scatter_nd"] = structure.scatter_nd(torch.from_numpy(idx).long(), torch.from_numpy(feats),
                                                           [B, D, H, W, C]).numpy()
    # ---- get_conv_output_size: the strided layers of FPNSpMiddleResNetFHD (scn.py:331-369) and of BEVFusion's
    # SparseEncoder at voxelnet_0p075 (sparse_encoder.py:57-130), in (z, y, x) order
    geoms = [((41, 1024, 1024), (3, 3, 3), (2, 2, 2), (1, 1, 1)), ((21, 512, 512), (3, 3, 3), (2, 2, 2), (1, 1, 1)),
             ((11, 256, 256), (3, 3, 3), (2, 2, 2), (0, 1, 1)), ((5, 128, 128), (3, 1, 1), (2, 1, 1), (0, 0, 0)),
             ((41, 1440, 1440), (3, 3, 3), (2, 2, 2), (1, 1, 1)), ((21, 720, 720), (3, 3, 3), (2, 2, 2), (1, 1, 1)),
             ((11, 360, 360), (3, 3, 3), (2, 2, 2), (0, 1, 1)), ((5, 180, 180), (3, 1, 1), (2, 1, 1), (0, 0, 0)),
             ((7, 9, 10), (3, 3, 3), (2, 2, 2), (1, 1, 1)), ((4, 5, 6), (3, 3, 3), (2, 2, 2), (0, 1, 1))]
    store["conv.geoms"] = np.array([[*i, *k, *s, *p] for i, k, s, p in geoms])
    store["conv.out"] = np.array([ops.get_conv_output_size(list(i), list(k), list(s), list(p), [1, 1, 1]) for i, k, s, p in geoms])
    out = os.path.join(os.path.dirname(HERE), "tests", "golden", "spconv_glue.npz")
    np.savez_compressed(out, **store)
    print("wrote", out, {k: v.shape for k, v in store.items()})
    print(store["conv.out"].tolist())


if __name__ == "__main__":
    main()
