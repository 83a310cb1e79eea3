"""CPU suite: the encoder's output shapes against the reference's own ``get_conv_output_size``
(bevfusion/mmdet3d/ops/spconv/ops.py:19-32, run in the build container by oracle/gen_golden_spconv_glue.py)."""
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_encoder_output_shapes_equal_the_reference_function():
    from al3d.models.backbones import _SparseEncoderBase
    g = np.load(os.path.join(G, "spconv_glue.npz"))
    for row, want in zip(g["conv.geoms"], g["conv.out"]):
        i, k, s, p = [list(map(int, row[3 * j:3 * j + 3])) for j in range(4)]
        assert _SparseEncoderBase._out_shape(i, k, s, p) == list(map(int, want)), (i, k, s, p)


def test_the_shipped_encoders_walk_the_reference_shapes():
    """FPNSpMiddleResNetFHD's four strided layers (det3d/models/backbones/scn.py:331-369) from the CBGS grid and from
    BEVFusion's 0.075 m grid: the chain of output shapes == the reference function's chain."""
    from al3d.models.backbones import FPNSpMiddleResNetFHD, _SpConvParams
    g = np.load(os.path.join(G, "spconv_glue.npz"))
    table = {tuple(int(v) for v in r): [int(v) for v in o] for r, o in zip(g["conv.geoms"], g["conv.out"])}
    m = FPNSpMiddleResNetFHD(num_input_features=5)
    for start in ([41, 1024, 1024], [41, 1440, 1440]):
        shape = list(start)
        for seq in m._stages():
            for mod in seq.children():
                if isinstance(mod, _SpConvParams) and not mod.subm:
                    key = (*shape, *mod.kernel_size, *mod.stride, *mod.padding)
                    assert key in table, key
                    shape = m._out_shape(shape, mod.kernel_size, mod.stride, mod.padding)
                    assert shape == table[key]
        assert shape[0] == 2
