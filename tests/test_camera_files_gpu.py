"""GPU suite: configs[4] from files -- the device image pipeline and the BEVFusion sweep rule against the reference's own
classes (golden vectors: ImageAug3D, LoadPointsFromMultiSweeps; oracle/gen_golden_bevfusion_loading.py), the file-fed
camera+lidar loader end to end, and the CLI on an mmdet3d-format pool written to a temp dir."""
import hashlib
import json
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bevfusion_loading as BL  # noqa: E402
from gen_golden_bevfusion_loading import synth_image  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = os.path.join(HERE, "golden")


@pytest.mark.parametrize("case", ["small", "odd", "full"])
def test_device_image_aug_equals_the_reference_class(case):
    """al3d_image_aug_normalize_u8: the 8-bit crop == the reference's ImageAug3D output (PIL bicubic resize + crop) bit for
    bit, the float32 output == ImageNormalize's two float32 operations on it bit for bit, img_aug_matrix == the reference's."""
    from al3d.datasets import ImageAugTest
    g = np.load(os.path.join(G, "bevfusion_image_aug.npz"))
    h, w = [int(v) for v in g[f"{case}.hw"]]
    final_dim = tuple(int(v) for v in g[f"{case}.final_dim"])
    frames = np.stack([synth_image(int(s), h, w) for s in g[f"{case}.seeds"]])
    aug = ImageAugTest((w, h), final_dim, device=DEV)
    out, u8 = aug(torch.from_numpy(frames).to(DEV), want_u8=True)
    torch.cuda.synchronize()
    u8, out = u8.cpu().numpy(), out.cpu().numpy()
    for k in range(len(frames)):
        assert hashlib.sha256(np.ascontiguousarray(u8[k]).tobytes()).hexdigest() == str(g[f"{case}.sha256"][k])
        assert np.array_equal(aug.matrix, g[f"{case}.matrix"][k])
    want = BL.image_normalize(u8)
    assert np.array_equal(out.view(np.int32), want.view(np.int32))
    t = (torch.from_numpy(u8).float() / 255 - torch.tensor([0.485, 0.456, 0.406])) / torch.tensor([0.229, 0.224, 0.225])
    assert np.array_equal(out.view(np.int32), t.numpy().view(np.int32))          # == torch's float32 expression too


def test_image_aug_rejects_what_it_does_not_build():
    from al3d import lib
    from al3d.datasets import ImageAugTest
    with pytest.raises(lib.Al3dError):
        ImageAugTest((400, 225), (256, 704), device=DEV)                    # the crop would leave the resized frame
    aug = ImageAugTest((400, 225), (64, 176), device=DEV)
    with pytest.raises(lib.Al3dError):
        aug(torch.zeros((1, 225, 400, 3), dtype=torch.float32, device=DEV))  # not 8-bit
    assert aug(torch.zeros((0, 225, 400, 3), dtype=torch.uint8, device=DEV)).shape == (0, 64, 176, 3)


def _write_pool_from_golden(tmp_path, cases):
    """An mmdet3d-format pool whose lidar files are the golden sweep inputs and whose cameras are PNGs of seeded frames."""
    from PIL import Image
    g = np.load(os.path.join(G, "bevfusion_sweeps.npz"))
    infos, want = [], []
    names = ["CAM_FRONT", "CAM_BACK"]
    for ci, case in enumerate(cases):
        n = int(g[f"{case}.nsweeps"][0])
        g[f"{case}.key"].tofile(tmp_path / f"{case}_key.bin")
        sweeps = []
        for i in range(n):
            g[f"{case}.sweep{i}"].tofile(tmp_path / f"{case}_s{i}.bin")
            sweeps.append(dict(data_path=f"{case}_s{i}.bin", timestamp=int(g[f"{case}.sweep{i}.ts"][0]),
                               sensor2lidar_rotation=g[f"{case}.sweep{i}.R"], sensor2lidar_translation=g[f"{case}.sweep{i}.t"]))
        cams = {}
        for k, nm in enumerate(names):
            Image.fromarray(synth_image(10 * ci + k, 225, 400)).save(tmp_path / f"{case}_{nm}.png")
            yaw = 0.4 + k
            R = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1.0]]) @ \
                np.array([[0, 0, 1.0], [-1, 0, 0], [0, -1, 0]])
            cams[nm] = dict(data_path=f"{case}_{nm}.png", sensor2lidar_rotation=R,
                            sensor2lidar_translation=np.array([1.0 + k, 0.2, 1.5]),
                            camera_intrinsics=np.array([[300.0, 0, 200], [0, 300.0, 112], [0, 0, 1]]))
        infos.append(dict(token=f"tok_{case}", lidar_path=f"{case}_key.bin", timestamp=int(g[f"{case}.ts"][0]), sweeps=sweeps,
                          cams=cams))
        want.append(g[f"{case}.out"])
    return infos, want, names


def test_camera_lidar_file_loader_end_to_end(tmp_path):
    """CameraLidarFileLoader on files: per sample the merged cloud == the reference's LoadPointsFromMultiSweeps output (five
    sweeps; eleven listed, nine used; none: nine filtered copies of the key frame) bit for bit; the images == the oracle's
    resize / crop / normalise of the decoded PNGs bit for bit; the matrices == get_data_info's; lidar_aug_matrix = I; the
    lidar keys are the voxelizer's of those clouds."""
    from al3d.datasets import CameraLidarFileLoader
    from al3d.datasets.camera_files import camera_matrices
    from PIL import Image
    cases = ["five", "eleven", "none"]
    infos, want, names = _write_pool_from_golden(tmp_path, cases)
    vox = dict(range=[-54.0, -54.0, -5.0, 54.0, 54.0, 3.0], voxel_size=[0.075, 0.075, 0.2], max_points_in_voxel=10,
               max_voxel_num=120000)
    loader = CameraLidarFileLoader(infos, vox, None, batch_size=2, device=DEV, root=str(tmp_path), image_size=(64, 176),
                                   threads=2, decode_threads=2)
    seen = 0
    for ex in loader:
        B = len(ex["metadata"])
        assert ex["img"].shape == (B, 2, 64, 176, 3) and ex["img"].dtype == torch.float32
        for k in range(B):
            i = ex["metadata"][k]["index"]
            got = ex["points"][k].cpu().numpy()
            # default point_cloud_range = the voxelizer's range: the reference pipeline's PointsRangeFilter (the oracle's
            # filter is pinned by the reference class in tests/test_image_ops.py; it drops one 5-sigma point of "five")
            ref = BL.points_range_filter(want[i], vox["range"])
            assert got.shape == ref.shape and np.array_equal(got.view(np.int32), ref.view(np.int32)), cases[i]
            for c, nm in enumerate(names):
                frame = np.asarray(Image.open(tmp_path / infos[i]["cams"][nm]["data_path"]).convert("RGB"))
                u8, m = BL.image_aug_test(frame, (64, 176))
                assert np.array_equal(ex["img"][k, c].cpu().numpy().view(np.int32), BL.image_normalize(u8).view(np.int32))
                assert np.array_equal(ex["img_aug_matrix"][k, c].cpu().numpy(), m)
                l2i, K, c2l = camera_matrices(infos[i]["cams"][nm])
                assert np.array_equal(ex["lidar2image"][k, c].cpu().numpy(), l2i)
                assert np.array_equal(ex["camera_intrinsics"][k, c].cpu().numpy(), K)
                assert np.array_equal(ex["camera2lidar"][k, c].cpu().numpy(), c2l)
            assert torch.equal(ex["lidar_aug_matrix"][k].cpu(), torch.eye(4))
        assert int(ex["num_voxels"].sum()) == ex["coordinates"].shape[0] > 0
        seen += B
    assert seen == 3 and loader.images_decoded == 6


@pytest.mark.parametrize("case", ["five", "eleven", "none"])
def test_camera_lidar_file_loader_applies_the_points_range_filter(tmp_path, case):
    """the test pipeline's PointsRangeFilter (default.yaml:233-235) sits between the sweep merge and everything
    that reads ``points`` (the voxelizer AND the view transform's lidar depth image).  The loader's merged, filtered cloud
    == the reference's LoadPointsFromMultiSweeps -> PointsRangeFilter output bit for bit (golden from the reference classes;
    the range's x_max is exactly one merged point's x: strict bound); with the filter off the unfiltered golden comes back."""
    from al3d.datasets import CameraLidarFileLoader
    g = np.load(os.path.join(G, "bevfusion_sweeps.npz"))
    infos, want, _ = _write_pool_from_golden(tmp_path, [case])
    rg = g[f"{case}.range"]
    vox = dict(range=[-54.0, -54.0, -5.0, 54.0, 54.0, 3.0], voxel_size=[0.075, 0.075, 0.2], max_points_in_voxel=10,
               max_voxel_num=120000)
    for pr, ref in ((rg.tolist(), g[f"{case}.out_range"]), (None, want[0])):
        loader = CameraLidarFileLoader(infos, vox, None, batch_size=1, device=DEV, root=str(tmp_path), image_size=(64, 176),
                                       threads=2, decode_threads=2, point_cloud_range=pr)
        ex = next(iter(loader))
        got = ex["points"][0].cpu().numpy()
        assert got.shape == ref.shape and np.array_equal(got.view(np.int32), ref.view(np.int32)), (case, pr)
    # default: the voxelizer's range
    assert CameraLidarFileLoader(infos, vox, None, batch_size=1, device=DEV, root=str(tmp_path), image_size=(64, 176),
                                 threads=2, decode_threads=2).point_range == vox["range"]


def test_cli_sweeps_a_camera_lidar_pool_from_files(tmp_path):
    """tools/active_select.py on BASELINE configs[4] WITHOUT --synthetic-scenes: an mmdet3d-format pool on disk (lidar .bin
    files, six JPEG cameras per sample, infos pkl) -> CameraLidarFileLoader -> the registered BEVFusion detector ->
    SpatialTemporalFeatureSelector: the two-invocation flow writes a selection that obeys the budget."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from write_synthetic_pool import write_camera_lidar_pool
    data = tmp_path / "data" / "nuScenes"
    infos, logs = write_camera_lidar_pool(str(data), scenes=1, base=2, frames_per_scene=40)
    infos = infos[:12]
    with open(data / "infos_train_10sweeps_withvelo.pkl", "wb") as f:
        pickle.dump(infos, f)
    cfg = tmp_path / "camera_lidar_files.py"
    cfg.write_text(f'''_base_ = "{os.path.join(ROOT, "examples", "active", "bevfusion_camera_lidar_spatial_temporal_feature.py")}"
data_root = "{data}"
selector = dict(logs_file="{data / "log.json"}", distance_store_file=None, buffer_path="{tmp_path / "feature_pred.pt"}",
                buffer_file="{tmp_path / "buffer.json"}", infos_origin="{data / "infos_train_10sweeps_withvelo.pkl"}")
''')
    cmd = [sys.executable, os.path.join(ROOT, "tools", "active_select.py"), "--config", str(cfg), "--budget", "5", "--pred",
           "--batch", "4"]
    env = dict(os.environ, PYTHONPATH=ROOT)
    for _ in range(2):
        r = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
    out = json.load(open(tmp_path / "buffer.json"))
    assert list(out) == ["0", "5"] and len(out["5"]) >= 2 and all(0 <= i < 12 for i in out["5"])
    cost = sum(0.12 + 0.04 * len(infos[i]["gt_names"]) for i in out["5"])
    assert cost <= 5
