"""CPU suite: the BEVFusion data-path oracle (oracle/bevfusion_loading.py) against the reference's own classes (golden
vectors from oracle/gen_golden_bevfusion_loading.py: ImageAug3D, LoadPointsFromMultiSweeps) and against the installed Pillow
(the reference's resize IS Pillow's), and the library's host-side filter tables against the oracle's."""
import ctypes
import hashlib
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import bevfusion_loading as BL  # noqa: E402
from gen_golden_bevfusion_loading import synth_image  # noqa: E402

G = os.path.join(HERE, "golden")


def test_pil_resize_restatement_equals_the_installed_pillow():
    """Resample.c restated in numpy == ``Image.resize`` (default filter = BICUBIC, and BILINEAR), down- and up-scaling,
    odd sizes: bit for bit.  Pillow is the reference's own dependency (pinned 8.4.0; the algorithm has not changed)."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(0)
    for (H, W, oh, ow) in [(90, 160, 43, 76), (37, 53, 80, 111), (225, 400, 108, 192), (64, 64, 64, 30)]:
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        assert np.array_equal(BL.pil_resize(img, ow, oh), np.asarray(Image.fromarray(img).resize((ow, oh))))
        assert np.array_equal(BL.pil_resize(img, ow, oh, 2), np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR)))


@pytest.mark.parametrize("case", ["small", "odd", "full"])
def test_image_aug_oracle_matches_the_reference_class(case):
    """oracle.image_aug_test == the reference's ImageAug3D (is_train=False) on the same seeded frames: cropped 8-bit image
    (sha256; the full arrays for the small cases) and the 4 x 4 img_aug_matrix."""
    g = np.load(os.path.join(G, "bevfusion_image_aug.npz"))
    h, w = g[f"{case}.hw"]
    final_dim = tuple(int(v) for v in g[f"{case}.final_dim"])
    for k, seed in enumerate(g[f"{case}.seeds"]):
        u8, m = BL.image_aug_test(synth_image(int(seed), int(h), int(w)), final_dim)
        assert hashlib.sha256(np.ascontiguousarray(u8).tobytes()).hexdigest() == str(g[f"{case}.sha256"][k])
        assert np.array_equal(m, g[f"{case}.matrix"][k])
        if case != "full":
            assert np.array_equal(u8, g[f"{case}.out_u8"][k])
        else:
            assert np.array_equal(u8[:8, :8], g[f"{case}.out_u8_corner"][k])


def _golden_sweeps(case):
    g = np.load(os.path.join(G, "bevfusion_sweeps.npz"))
    n = int(g[f"{case}.nsweeps"][0])
    sweeps = [dict(points=g[f"{case}.sweep{i}"], timestamp=int(g[f"{case}.sweep{i}.ts"][0]),
                   sensor2lidar_rotation=g[f"{case}.sweep{i}.R"], sensor2lidar_translation=g[f"{case}.sweep{i}.t"])
              for i in range(n)]
    return g[f"{case}.key"], sweeps, int(g[f"{case}.ts"][0]), g[f"{case}.out"]


@pytest.mark.parametrize("case", ["five", "eleven", "none"])
def test_sweep_merge_oracle_matches_the_reference_class(case):
    """oracle.merge_sweeps == the reference's LoadPointsFromMultiSweeps (sweeps_num 9, pad_empty_sweeps, remove_close,
    test_mode): five sweeps, eleven (the first nine are used), none (nine filtered copies of the key frame): bit for bit."""
    key, sweeps, ts, want = _golden_sweeps(case)
    got = BL.merge_sweeps(key, sweeps, ts)
    assert got.shape == want.shape and np.array_equal(got.view(np.int32), want.view(np.int32))


@pytest.mark.parametrize("case", ["five", "eleven", "none"])
def test_points_range_filter_oracle_matches_the_reference_class(case):
    """oracle.points_range_filter == the reference's PointsRangeFilter on the merged cloud (strict bounds: the range's
    x_max is exactly one point's x, and that point goes): same rows, same order, bit for bit."""
    g = np.load(os.path.join(G, "bevfusion_sweeps.npz"))
    key, sweeps, ts, merged = _golden_sweeps(case)
    rg, want = g[f"{case}.range"], g[f"{case}.out_range"]
    assert (merged[:, 0] == rg[3]).any()
    got = BL.points_range_filter(BL.merge_sweeps(key, sweeps, ts), rg)
    assert got.shape == want.shape and np.array_equal(got.view(np.int32), want.view(np.int32))


def test_library_filter_tables_equal_the_oracle():
    """al3d_image_resample_coeffs (a host function of the library: no GPU needed) == the numpy restatement."""
    from al3d import lib
    L = lib.load()
    for (i, o) in [(1600, 768), (900, 432), (53, 111), (160, 76), (64, 64)]:
        for f in (2, 3):
            ks = L.al3d_image_resample_ksize(i, o, f)
            b = np.zeros((o, 2), np.int32)
            c = np.zeros((o, ks), np.int32)
            lib.call("al3d_image_resample_coeffs", i, o, f, b.ctypes.data_as(ctypes.c_void_p), c.ctypes.data_as(ctypes.c_void_p))
            rb, rc = BL.resample_coeffs(i, o, f)
            assert np.array_equal(b, rb) and np.array_equal(c, rc), (i, o, f)
    assert L.al3d_image_resample_ksize(0, 5, 3) < 0 and L.al3d_image_resample_ksize(5, 5, 7) < 0


def test_camera_matrices_follow_get_data_info():
    """lidar2image = K @ lidar2camera, camera2lidar as get_data_info builds them: a lidar point projected through
    lidar2image lands where the explicit inverse transform + intrinsics put it."""
    from al3d.datasets.camera_files import camera_matrices
    rng = np.random.default_rng(4)
    yaw = 0.7
    R = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1.0]]) @ np.array([[0, 0, 1.0], [-1, 0, 0], [0, -1, 0]])
    cam = dict(sensor2lidar_rotation=R, sensor2lidar_translation=np.array([1.5, -0.3, 1.6]),
               camera_intrinsics=np.array([[1260.0, 0, 800], [0, 1260.0, 450], [0, 0, 1]]))
    l2i, K, c2l = camera_matrices(cam)
    ol2i, oK, oc2l = BL.camera_matrices(cam)
    assert np.array_equal(l2i, ol2i) and np.array_equal(K, oK) and np.array_equal(c2l, oc2l)
    p = np.array([12.0, 3.0, 0.5])
    pc = R.T @ (p - cam["sensor2lidar_translation"])
    uvw = cam["camera_intrinsics"] @ pc
    got = l2i[:3, :3].astype(np.float64) @ p + l2i[:3, 3]
    assert np.allclose(got, uvw, rtol=1e-5)
