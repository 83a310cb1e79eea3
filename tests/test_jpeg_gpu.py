"""GPU suite: the device half of the split JPEG decoder (csrc/jpeg.hip) == the installed Pillow's decode, byte for byte, and
the camera file loader fed JPEGs through it == the same loader decoding with Pillow."""
import ctypes
import io

import numpy as np
import pytest
import torch

from gen_golden_bevfusion_loading import synth_image
from test_jpeg_host import CASES, _encode, _pil, host_decode

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def device_decode(items):
    """items: [(info, quant, coefs)] of ONE geometry -> [n, H, W, 3] uint8 through al3d_jpeg_idct_rgb_u8."""
    from al3d import lib
    from al3d.selector_ops import _ptr, _stream
    info = np.ascontiguousarray(items[0][0], dtype=np.int32)
    n = len(items)
    coefs = torch.from_numpy(np.stack([it[2] for it in items])).to(DEV)
    quant = torch.from_numpy(np.stack([it[1] for it in items]).view(np.int16)).to(DEV)
    ip = info.ctypes.data_as(ctypes.c_void_p)
    ws = torch.empty(max(int(lib.load().al3d_jpeg_workspace_bytes(ip, n)), 16), dtype=torch.uint8, device=DEV)
    out = torch.empty((n, int(info[1]), int(info[0]), 3), dtype=torch.uint8, device=DEV)
    lib.call("al3d_jpeg_idct_rgb_u8", _ptr(coefs), _ptr(quant), ip, n, _ptr(out), _ptr(ws), _stream())
    return out.cpu().numpy()


@pytest.mark.parametrize("case", range(len(CASES)))
def test_device_decode_equals_pillow(case):
    c = CASES[case]
    datas = [_encode(synth_image(10 + case, *c["hw"]), **c["kw"]),
             _encode(synth_image(90 + case, *c["hw"]), **dict(c["kw"], quality=min(95, c["kw"]["quality"] + 7)))]
    got = device_decode([host_decode(d) for d in datas])         # two images, two sets of quantisation tables, one launch
    for k, d in enumerate(datas):
        assert np.array_equal(got[k], _pil(d)), (case, k)


def test_device_decode_grayscale_and_extreme_content():
    from PIL import Image
    g = synth_image(3, 75, 101)[..., 0]
    buf = io.BytesIO()
    Image.fromarray(g).save(buf, format="JPEG", quality=80)
    assert np.array_equal(device_decode([host_decode(buf.getvalue())])[0], _pil(buf.getvalue()))
    rng = np.random.default_rng(2)
    hard = (rng.integers(0, 2, (64, 96, 3)) * 255).astype(np.uint8)          # saturated noise: clamps on every path
    hard[:, 48:] = np.array([255, 0, 255], np.uint8)
    for sub in (0, 1, 2):
        d = _encode(hard, quality=100, subsampling=sub)
        assert np.array_equal(device_decode([host_decode(d)])[0], _pil(d)), sub


def test_camera_file_loader_split_jpeg_equals_pillow(tmp_path):
    """CameraLidarFileLoader on a pool whose camera frames are JPEGs: jpeg='split' (entropy decoding on host threads,
    the rest on the device) hands the detector the SAME normalised images, bit for bit, as jpeg='pil'; a batch holding
    a frame the split decoder does not take (progressive) is decoded by Pillow as a whole."""
    from PIL import Image
    from al3d.datasets import CameraLidarFileLoader
    rng = np.random.default_rng(4)
    infos = []
    for s in range(3):
        pts = rng.normal(0, 10, (300, 5)).astype(np.float32)
        pts.tofile(tmp_path / f"k{s}.bin")
        cams = {}
        for k, nm in enumerate(["CAM_FRONT", "CAM_BACK"]):
            Image.fromarray(synth_image(7 * s + k, 225, 400)).save(tmp_path / f"s{s}_{nm}.jpg", quality=80,
                                                                   progressive=(s == 2 and k == 1))
            cams[nm] = dict(data_path=f"s{s}_{nm}.jpg", sensor2lidar_rotation=np.eye(3), sensor2lidar_translation=np.zeros(3),
                            camera_intrinsics=np.array([[300.0, 0, 200], [0, 300.0, 112], [0, 0, 1]]))
        infos.append(dict(token=f"t{s}", lidar_path=f"k{s}.bin", timestamp=1_000_000 * s, sweeps=[], cams=cams))
    vox = dict(range=[-54.0, -54.0, -5.0, 54.0, 54.0, 3.0], voxel_size=[0.075, 0.075, 0.2], max_points_in_voxel=10,
               max_voxel_num=120000)
    outs = {}
    for mode in ("pil", "split"):
        loader = CameraLidarFileLoader(infos, vox, None, batch_size=1, device=DEV, root=str(tmp_path), image_size=(64, 176),
                                       threads=2, decode_threads=2, jpeg=mode)
        outs[mode] = [ex["img"].cpu().numpy() for ex in loader]
        if mode == "split":
            assert loader.images_split == 4 and loader.images_decoded == 6     # the third sample fell back as a whole
        else:
            assert loader.images_split == 0
    for a, b in zip(outs["pil"], outs["split"]):
        assert np.array_equal(a.view(np.int32), b.view(np.int32))
