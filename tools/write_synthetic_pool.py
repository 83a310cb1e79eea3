#!/usr/bin/env python3
"""Write a synthetic nuScenes-shaped pool to disk (tmpfs by default) in the reference's file formats:
``<root>/samples|sweeps/LIDAR_TOP/*.pcd.bin`` (float32 x,y,z,intensity,ring rows) + ``infos.pkl``
(nusc_common.py:410-419 schema with ``sweeps[i]{lidar_path, transform_matrix, time_lag}``) + ``log.json``.

    python tools/write_synthetic_pool.py --root /dev/shm/al3d_pool --scenes 16 [--base 16]

``--base`` distinct 10-sweep frames are written (10 files each, ~0.5 MB per file); the pool's frames reuse them
round-robin with their own per-sweep rigid transforms, so every frame still costs ten file reads of full size
(what `bench.py --from-files` measures) without needing scenes x 40 x 5 MB of tmpfs."""
import argparse
import json
import os
import pickle
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def write_pool(root, scenes, base=16, nsweeps=10, seed=0):
    from al3d import synthetic
    os.makedirs(os.path.join(root, "samples", "LIDAR_TOP"), exist_ok=True)
    os.makedirs(os.path.join(root, "sweeps", "LIDAR_TOP"), exist_ok=True)
    base_files = []
    for b in range(base):
        pts = synthetic.make_point_cloud(1000 + b, nsweeps=nsweeps)
        files = []
        for s in range(nsweeps):
            sel = pts[np.abs(pts[:, 4] - np.float32(0.05 * s)) < 1e-4]
            raw = np.zeros((len(sel), 5), dtype=np.float32)
            raw[:, :4] = sel[:, :4]
            raw[:, 0] += np.float32(0.2 * s)             # back into the sweep's own frame (the generator shifted it)
            raw[:, 4] = np.arange(len(sel)) % 32         # ring index column (dropped by the loader)
            rel = os.path.join("samples" if s == 0 else "sweeps", "LIDAR_TOP", f"base{b:03d}_s{s}.pcd.bin")
            raw.tofile(os.path.join(root, rel))
            files.append(rel)
        base_files.append(files)
    infos, logs = synthetic.make_pool(scenes, seed=seed)
    rng = np.random.default_rng(seed + 17)
    for i, info in enumerate(infos):
        files = base_files[i % base]
        info["lidar_path"] = files[0]
        sweeps = []
        for s in range(1, nsweeps):
            yaw = rng.normal(0.0, 0.01)
            T = np.eye(4)
            T[:2, :2] = [[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]]
            T[:3, 3] = [-0.2 * s + rng.normal(0, 0.01), rng.normal(0, 0.01), rng.normal(0, 0.002)]
            sweeps.append({"lidar_path": files[s], "sample_data_token": f"sd{i:06d}_{s}", "transform_matrix": T,
                           "time_lag": 0.05 * s})
        info["sweeps"] = sweeps
    with open(os.path.join(root, "infos.pkl"), "wb") as f:
        pickle.dump(infos, f)
    with open(os.path.join(root, "log.json"), "w") as f:
        json.dump(logs, f)
    return infos, logs


def write_camera_lidar_pool(root, scenes, base=4, sweeps=9, image_hw=(900, 1600), cams=6, fmt="jpg", seed=0,
                            frames_per_scene=40):
    """The same pool in the mmdet3d / BEVFusion info schema (bevfusion/tools/data_converter/nuscenes_converter.py):
    ``lidar_path``, ``timestamp`` (microseconds), ``sweeps[k]{data_path, timestamp, sensor2lidar_rotation,
    sensor2lidar_translation}``, ``cams{name: {data_path, sensor2lidar_rotation, sensor2lidar_translation,
    camera_intrinsics}}`` next to the keys the selectors read, plus ``base`` distinct sets of ``cams`` camera frames
    (``fmt`` jpg / png, written with PIL) reused round-robin."""
    from PIL import Image
    from al3d import synthetic
    infos, logs = write_pool(root, scenes, base=base, nsweeps=sweeps + 1, seed=seed)
    names = ["CAM_FRONT", "CAM_FRONT_RIGHT", "CAM_FRONT_LEFT", "CAM_BACK", "CAM_BACK_LEFT", "CAM_BACK_RIGHT"][:cams]
    H, W = image_hw
    rng = np.random.default_rng(seed + 91)
    for b in range(base):
        for n in names:
            os.makedirs(os.path.join(root, "samples", n), exist_ok=True)
            yy, xx = np.mgrid[0:H, 0:W]
            ph = rng.uniform(0, 6.28, 3)
            img = np.stack([(np.sin(xx / 23.0 + ph[0]) + np.cos(yy / 17.0)) * 50 + 128,
                            ((xx // 64 + yy // 48) % 2) * 180 + 30, (np.cos((xx + yy) / 31.0 + ph[2])) * 90 + 128], -1)
            img = np.clip(img + rng.normal(0, 12, img.shape), 0, 255).astype(np.uint8)
            Image.fromarray(img).save(os.path.join(root, "samples", n, f"base{b:03d}.{fmt}"))
    K4, cam2lidar, _, _, _, _ = synthetic.camera_setup(1, cams, seed + 9, image_hw)
    for i, info in enumerate(infos):
        ts = int(round(info["timestamp"] * 1e6))
        info["timestamp"] = ts
        for k, sw in enumerate(info["sweeps"]):
            T = np.asarray(sw.pop("transform_matrix"), dtype=np.float64)
            sw["data_path"] = sw.pop("lidar_path")
            sw["sensor2lidar_rotation"], sw["sensor2lidar_translation"] = T[:3, :3].copy(), T[:3, 3].copy()
            sw["timestamp"] = ts - int(round(sw.pop("time_lag") * 1e6))
        info["cams"] = {}
        for c, n in enumerate(names):
            c2l = cam2lidar[0, c].double().numpy()
            info["cams"][n] = {"data_path": os.path.join("samples", n, f"base{i % base:03d}.{fmt}"),
                               "sensor2lidar_rotation": c2l[:3, :3].copy(), "sensor2lidar_translation": c2l[:3, 3].copy(),
                               "camera_intrinsics": K4[0, c, :3, :3].double().numpy()}
    with open(os.path.join(root, "infos.pkl"), "wb") as f:
        pickle.dump(infos, f)
    return infos, logs


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", default="/dev/shm/al3d_pool")
    ap.add_argument("--scenes", type=int, default=16)
    ap.add_argument("--base", type=int, default=16)
    ap.add_argument("--cameras", action="store_true", help="mmdet3d / BEVFusion info schema with six camera frames per sample")
    a = ap.parse_args()
    infos, _ = write_camera_lidar_pool(a.root, a.scenes, a.base) if a.cameras else write_pool(a.root, a.scenes, a.base)
    size = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(a.root) for f in fs)
    print(f"wrote {len(infos)} frames ({a.base} distinct 10-sweep file sets, {size / 2**20:.1f} MiB) under {a.root}")
