"""GPU suite: the remaining dense BEVFusion modules and the spconv glue against goldens produced by the reference's own
code (oracle/gen_golden_bevfusion_second.py, oracle/gen_golden_spconv_glue.py):

  * SECOND + SECONDFPN (the decoder; second.py:12-97, necks/second.py:12-99) == this build's RPN after
    convert_decoder_state_dict;
  * GeneralizedLSSFPN (generalized_lss.py:13-110) with the swint configs' upsample_cfg (align_corners false);
  * SparseConvTensor.dense() + the [N, C * D, H, W] view (structure.py:5-63, scn.py:387-390) == al3d_sp_to_dense_nhwc.

The module fixtures were generated behind mmcv.cnn FACTORY STAND-INS (the ``standin`` entry of each file says which): what
they pin is the reference classes' structure and forward code on torch's own layers.  Parameters are regenerated here by the
generator's per-tensor seeded rule and checked against the fixture's digest."""
import hashlib
import os
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def seeded_state(keys, shapes, seed):
    """oracle/gen_golden_bevfusion_second.py::seeded_state_ restated on (name, shape) lists -> (state dict, digest)."""
    sd, h = {}, hashlib.sha256()
    for name, shp in zip(keys, shapes):
        name, shp = str(name), eval(str(shp))
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.zeros(shp, dtype=torch.int64)
            continue
        g = torch.Generator().manual_seed(int(seed) + zlib.crc32(name.encode()))
        if len(shp) >= 2:
            v = torch.randn(shp, generator=g) * (2.0 / float(np.prod(shp[1:]))) ** 0.5
        elif name.endswith("running_var") or name.endswith("weight"):
            v = torch.rand(shp, generator=g) + 0.5
        else:
            v = torch.randn(shp, generator=g) * 0.1
        sd[name] = v
        h.update(name.encode() + v.numpy().tobytes())
    return sd, h.hexdigest()


def test_second_and_secondfpn_match_the_reference_modules():
    from al3d.models.bevfusion_compat import convert_decoder_state_dict
    from al3d.models.necks import RPN
    g = np.load(os.path.join(G, "bevfusion_second.npz"))
    assert "factory stand-ins" in str(g["standin"])
    sd = {}
    for part, keys, shapes, seed, digest in (("backbone", g["keys_backbone"], g["shapes_backbone"], g["seeds"][0], g["digest"][0]),
                                             ("neck", g["keys_neck"], g["shapes_neck"], g["seeds"][1], g["digest"][1])):
        part_sd, dig = seeded_state(keys, shapes, seed)
        assert dig == str(digest), part                         # the same parameters the reference classes ran with
        sd.update({f"decoder.{part}.{k}": v for k, v in part_sd.items()})
    rpn = RPN(layer_nums=[5, 5], ds_layer_strides=[1, 2], ds_num_filters=[128, 256], us_layer_strides=[1, 2],
              us_num_filters=[256, 256], num_input_features=256)
    missing, unexpected = rpn.load_state_dict(convert_decoder_state_dict(sd, neck_prefix=""), strict=True)
    rpn = rpn.to(DEV).eval()
    x = torch.from_numpy(g["x"]).to(DEV)                        # [N, C, H = x, W = y] there -> [N, H = y, W = x, C] here
    with torch.no_grad():
        out = rpn(x.permute(0, 3, 2, 1).contiguous())
    ref = torch.from_numpy(g["out"]).to(DEV).permute(0, 3, 2, 1)
    assert out.shape == ref.shape
    err = float((out - ref).abs().max())
    assert err <= 1e-4 * float(ref.abs().max()) + 1e-5, err    # eleven fp32 convolutions deep


def test_generalized_lss_fpn_matches_the_reference_module():
    from al3d.models.bevfusion_camera import GeneralizedLSSFPN
    g = np.load(os.path.join(G, "bevfusion_lss_fpn.npz"))
    assert "factory stand-ins" in str(g["standin"])
    sd, dig = seeded_state(g["keys"], g["shapes"], g["seeds"][0])
    assert dig == str(g["digest"][0])
    fpn = GeneralizedLSSFPN([192, 384, 768], 256, 3, upsample_cfg=dict(mode="bilinear", align_corners=False))
    fpn.load_state_dict(sd, strict=True)
    fpn = fpn.to(DEV).eval()
    ins = [torch.from_numpy(g[f"in{i}"]).to(DEV).permute(0, 2, 3, 1).contiguous() for i in range(3)]
    with torch.no_grad():
        outs = fpn(ins)
    assert len(outs) == 2
    for i, o in enumerate(outs):
        ref = torch.from_numpy(g[f"out{i}"]).to(DEV).permute(0, 2, 3, 1)
        assert o.shape == ref.shape
        err = float((o - ref).abs().max())
        assert err <= 1e-4 * float(ref.abs().max()) + 1e-5, (i, err)
    # the class default (align_corners = True) is a different map: the configs' setting matters
    dflt = GeneralizedLSSFPN([192, 384, 768], 256, 3)
    dflt.load_state_dict(sd, strict=True)
    with torch.no_grad():
        other = dflt.to(DEV).eval()(ins)[0]
    assert float((other - outs[0]).abs().max()) > 1e-3


@pytest.mark.parametrize("case", ["small", "odd"])
def test_dense_scatter_matches_the_reference_sparse_tensor(case):
    """al3d_sp_to_dense_nhwc == SparseConvTensor.dense() followed by the encoder's [N, C * D, H, W] view (channel = c D + z),
    bit for bit (a scatter), in this build's channels-last layout; scatter_nd itself is the [B, D, H, W, C] form of it."""
    from al3d.models.backbones import SparseTensor, _SparseEncoderBase
    g = np.load(os.path.join(G, "spconv_glue.npz"))
    B, C, D, H, W = [int(v) for v in g[f"{case}.shape"]]
    sp = SparseTensor(torch.from_numpy(g[f"{case}.features"]).to(DEV), torch.from_numpy(g[f"{case}.indices"]).to(DEV), [D, H, W], B)
    got = _SparseEncoderBase.dense_nhwc(sp).cpu().numpy()       # [B, H, W, C * D]
    want = np.transpose(g[f"{case}.bev"], (0, 2, 3, 1))
    assert got.shape == want.shape and np.array_equal(got.view(np.int32), want.view(np.int32))
    nd = np.transpose(g[f"{case}.scatter_nd"], (0, 2, 3, 4, 1)).reshape(B, H, W, C * D)      # [B, D, H, W, C] -> c D + z
    assert np.array_equal(got, nd)
