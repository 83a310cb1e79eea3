"""Host-side reading of real nuScenes 10-sweep frames.

Rules of the reference loader (det3d/datasets/pipelines/loading.py:17-63,98-126): ``.bin`` files
are float32 x,y,z,intensity,ring -> first four columns; sweeps (not the key frame) drop points
with |x|<1 and |y|<1 in their own frame, are moved by ``transform_matrix`` (float64 4x4, result
stored back in float32) and get a time-lag column; output ``[P,5]`` float32.  The reference picks
the nsweeps-1 sweeps in a fresh random order per sample (so its sweep is not reproducible, SURVEY
D8); here the order is the list order unless an ``rng`` is passed.
"""
import numpy as np


def read_file(path, num_point_feature=4):
    pts = np.fromfile(path, dtype=np.float32)
    pts = pts[: pts.shape[0] - pts.shape[0] % 5]
    return pts.reshape(-1, 5)[:, :num_point_feature]


def read_sweep(sweep, min_distance=1.0):
    pts = read_file(str(sweep["lidar_path"])).T              # [4, p]
    close = (np.abs(pts[0]) < min_distance) & (np.abs(pts[1]) < min_distance)
    pts = pts[:, ~close]
    if sweep.get("transform_matrix") is not None:
        n = pts.shape[1]
        pts[:3, :] = np.asarray(sweep["transform_matrix"]).dot(np.vstack((pts[:3, :], np.ones(n))))[:3, :]
    times = sweep["time_lag"] * np.ones((1, pts.shape[1]))
    return pts.T, times.T


def load_frame_points(info, nsweeps=10, root=None, rng=None):
    import os
    def p(x):
        return x if root is None or os.path.isabs(str(x)) else os.path.join(root, str(x))
    pts = read_file(p(info["lidar_path"]))
    plist, tlist = [pts], [np.zeros((pts.shape[0], 1))]
    assert nsweeps - 1 <= len(info["sweeps"]), \
        f"nsweeps {nsweeps} should not greater than list length {len(info['sweeps'])}."
    order = range(nsweeps - 1) if rng is None else rng.choice(len(info["sweeps"]), nsweeps - 1, replace=False)
    for i in order:
        sw = dict(info["sweeps"][i])
        sw["lidar_path"] = p(sw["lidar_path"])
        ps, ts = read_sweep(sw)
        plist.append(ps)
        tlist.append(ts)
    points = np.concatenate(plist, axis=0)
    times = np.concatenate(tlist, axis=0).astype(points.dtype)
    return np.hstack([points, times]).astype(np.float32)
