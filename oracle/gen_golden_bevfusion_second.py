"""TEST INFRASTRUCTURE ONLY.  Golden vectors for the three remaining dense BEVFusion modules of configs[3] / configs[4],
produced by the reference's own classes run on the CPU in this container with seeded parameters:

  * ``SECOND`` (bevfusion/mmdet3d/models/backbones/second.py:12-97) and ``SECONDFPN`` (necks/second.py:12-99) with the
    decoder settings of configs/nuscenes/det/transfusion/secfpn/default.yaml:2-28 -- this build's ``RPN`` after
    ``convert_decoder_state_dict``;
  * ``GeneralizedLSSFPN`` (necks/generalized_lss.py:13-110) with the swint settings (in_channels [192, 384, 768], 256 out).

Their files import ``mmcv.cnn`` / ``mmcv.runner`` / the mmdet registries at module level (absent here: ordinary
ModuleNotFoundErrors).  The stand-ins registered first are FACTORY STAND-INS, labelled as such in the fixtures
(``standin`` entries): ``build_conv_layer`` / ``build_norm_layer`` / ``build_upsample_layer`` return the torch layer their
``type`` names (Conv2d / BatchNorm2d with the cfg's eps and momentum / ConvTranspose2d, remaining cfg entries as keyword
arguments -- what mmcv's registries resolve those names to), ``ConvModule`` is conv (no bias when a norm follows) -> norm ->
ReLU with mmcv's attribute names (``conv``, ``bn``, ``activate``), ``BaseModule`` is ``nn.Module``, ``auto_fp16`` the
identity.  The forward passes -- block structure, strides, paddings, the top-down interpolate / cat order -- are the
reference's code.  Run in the build container only (needs /root/reference); writes tests/golden/bevfusion_second.npz and
bevfusion_lss_fpn.npz.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.environ.get("AL3D_REFERENCE_ROOT", "/root/reference")
BEV = os.path.join(ROOT, "bevfusion")
STANDIN = ("mmcv.cnn factory stand-ins: build_conv_layer -> nn.Conv2d, build_norm_layer -> nn.BatchNorm2d(eps, momentum), "
           "build_upsample_layer(deconv) -> nn.ConvTranspose2d, ConvModule = conv(bias=False) + BatchNorm2d + ReLU; "
           "mmcv.runner.BaseModule = nn.Module, auto_fp16 = identity; registries return the class")


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


class _Registry:
    def register_module(self, *a, **k):
        return lambda cls: cls


def build_conv_layer(cfg, *args, **kwargs):
    cfg = dict(cfg or dict(type="Conv2d"))
    assert cfg.pop("type") in ("Conv2d", "Conv"), cfg
    return nn.Conv2d(*args, **kwargs, **cfg)


def build_norm_layer(cfg, num_features, postfix=""):
    cfg = dict(cfg)
    assert cfg.pop("type") in ("BN", "BN2d"), cfg
    cfg.pop("requires_grad", None)
    return "bn" + str(postfix), nn.BatchNorm2d(num_features, **cfg)


def build_upsample_layer(cfg, *args, **kwargs):
    cfg = dict(cfg)
    assert cfg.pop("type") == "deconv", cfg
    return nn.ConvTranspose2d(*args, **kwargs, **cfg)


class ConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, conv_cfg=None, norm_cfg=None,
                 act_cfg=dict(type="ReLU"), inplace=True, **kw):
        super().__init__()
        assert not kw and (act_cfg is None or act_cfg["type"] == "ReLU")
        self.conv = build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                     bias=norm_cfg is None)
        self.bn = build_norm_layer(norm_cfg, out_channels)[1] if norm_cfg is not None else None
        self.activate = nn.ReLU(inplace=inplace) if act_cfg is not None else None

    def forward(self, x):
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        return self.activate(x) if self.activate is not None else x


class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg


def auto_fp16(*a, **k):
    return lambda fn: fn


def import_reference():
    if not os.path.isdir(BEV):
        raise RuntimeError(f"reference tree not found at {BEV}")
    mmcv = _mod("mmcv")
    mmcv.cnn = _mod("mmcv.cnn", build_conv_layer=build_conv_layer, build_norm_layer=build_norm_layer,
                    build_upsample_layer=build_upsample_layer, ConvModule=ConvModule)
    mmcv.runner = _mod("mmcv.runner", BaseModule=BaseModule, auto_fp16=auto_fp16)
    _mod("mmdet")
    _mod("mmdet.models", BACKBONES=_Registry(), NECKS=_Registry())
    _mod("mmdet.models.builder", BACKBONES=_Registry(), NECKS=_Registry())
    _pkg("mmdet3d", os.path.join(BEV, "mmdet3d"))
    _pkg("mmdet3d.models", os.path.join(BEV, "mmdet3d", "models"))
    _pkg("mmdet3d.models.backbones", os.path.join(BEV, "mmdet3d", "models", "backbones"))
    _pkg("mmdet3d.models.necks", os.path.join(BEV, "mmdet3d", "models", "necks"))
    second = importlib.import_module("mmdet3d.models.backbones.second")
    fpn = importlib.import_module("mmdet3d.models.necks.second")
    lss = importlib.import_module("mmdet3d.models.necks.generalized_lss")
    return second.SECOND, fpn.SECONDFPN, lss.GeneralizedLSSFPN


def seeded_state_(mod, seed):
    """Every floating-point entry of the state dict from a per-tensor seeded generator (seed + crc32(name)): conv kernels
    ~ N(0, 2 / fan_in), BN weights / variances ~ U(0.5, 1.5), the rest ~ N(0, 0.1).  tests/test_bevfusion_second_golden_gpu.py
    holds the same rule, so the fixtures carry inputs, outputs and a digest of the parameters instead of 18 MB of weights."""
    import hashlib
    import zlib
    sd = mod.state_dict()
    h = hashlib.sha256()
    for name in sorted(sd):
        t = sd[name]
        if not t.dtype.is_floating_point:
            continue
        g = torch.Generator().manual_seed(int(seed) + zlib.crc32(name.encode()))
        if t.dim() >= 2:
            v = torch.randn(t.shape, generator=g) * (2.0 / float(np.prod(t.shape[1:]))) ** 0.5
        elif name.endswith("running_var") or name.endswith("weight"):
            v = torch.rand(t.shape, generator=g) + 0.5
        else:
            v = torch.randn(t.shape, generator=g) * 0.1
        t.copy_(v)
        h.update(name.encode() + v.numpy().tobytes())
    mod.load_state_dict(sd)
    return mod.eval(), h.hexdigest()


def main():
    SECOND, SECONDFPN, GeneralizedLSSFPN = import_reference()
    gold = os.path.join(os.path.dirname(HERE), "tests", "golden")
    g = torch.Generator().manual_seed(77)
    # ---- decoder: SECOND + SECONDFPN (secfpn/default.yaml:2-28); maps are [N, C, H = x, W = y] there
    bn = dict(type="BN", eps=1.0e-3, momentum=0.01)
    backbone, d0 = seeded_state_(SECOND(in_channels=256, out_channels=[128, 256], layer_nums=[5, 5], layer_strides=[1, 2],
                                        norm_cfg=bn, conv_cfg=dict(type="Conv2d", bias=False)), 1)
    neck, d1 = seeded_state_(SECONDFPN(in_channels=[128, 256], out_channels=[256, 256], upsample_strides=[1, 2], norm_cfg=bn,
                                       upsample_cfg=dict(type="deconv", bias=False), use_conv_for_no_stride=True), 2)
    x = torch.randn(1, 256, 12, 8, generator=g)
    with torch.no_grad():
        feats = backbone(x)
        out = neck(feats)[0]
    store = {"standin": np.array(STANDIN), "x": x.numpy(), "out": out.numpy(), "seeds": np.array([1, 2]),
             "digest": np.array([d0, d1]), "keys_backbone": np.array(sorted(backbone.state_dict())),
             "keys_neck": np.array(sorted(neck.state_dict())),
             "shapes_backbone": np.array([str(tuple(backbone.state_dict()[k].shape)) for k in sorted(backbone.state_dict())]),
             "shapes_neck": np.array([str(tuple(neck.state_dict()[k].shape)) for k in sorted(neck.state_dict())])}
    np.savez_compressed(os.path.join(gold, "bevfusion_second.npz"), **store)
    print("wrote bevfusion_second.npz", out.shape, [f.shape for f in feats])
    # ---- camera neck: GeneralizedLSSFPN on Swin-T's three last stages, upsample_cfg of the swint configs
    # (configs/nuscenes/det/transfusion/secfpn/camera+lidar/default.yaml:5-18: align_corners false)
    fpn, d2 = seeded_state_(GeneralizedLSSFPN(in_channels=[192, 384, 768], out_channels=256, start_level=0, num_outs=3,
                                              norm_cfg=dict(type="BN2d", requires_grad=True),
                                              act_cfg=dict(type="ReLU", inplace=True),
                                              upsample_cfg=dict(mode="bilinear", align_corners=False)), 3)
    ins = [torch.randn(1, 192, 8, 12, generator=g), torch.randn(1, 384, 4, 6, generator=g), torch.randn(1, 768, 2, 3, generator=g)]
    with torch.no_grad():
        outs = fpn([t.clone() for t in ins])
    sd = fpn.state_dict()
    store = {"standin": np.array(STANDIN), "seeds": np.array([3]), "digest": np.array([d2]), "keys": np.array(sorted(sd)),
             "shapes": np.array([str(tuple(sd[k].shape)) for k in sorted(sd)])}
    for i, t in enumerate(ins):
        store[f"in{i}"] = t.numpy()
    for i, t in enumerate(outs):
        store[f"out{i}"] = t.numpy()
    np.savez_compressed(os.path.join(gold, "bevfusion_lss_fpn.npz"), **store)
    print("wrote bevfusion_lss_fpn.npz", [o.shape for o in outs])


if __name__ == "__main__":
    main()
