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


# ---------------------------------------------------------------- device path (a1 on the GPU)
def _raw_rows(path):
    """The reference's read_file without the column cut: whole 5-float rows of a ``.bin``."""
    pts = np.fromfile(path, dtype=np.float32)
    return pts[: pts.shape[0] - pts.shape[0] % 5].reshape(-1, 5)


def merge_sweeps_device(files, xforms, time_lags, device, min_distance=1.0, pinned=None):
    """Raw ``[p,5]`` float32 arrays (file 0 = key frame), per-file 4x4 float64 transforms (or
    None) and time lags -> the combined ``[P,5]`` float32 cloud as a device tensor.

    One pinned staging buffer + one async H2D per frame; remove_close / transform / time column /
    compaction run in ``al3d_merge_sweeps_f32``.  ``pinned`` may be a reusable pinned uint8 tensor."""
    import torch
    from .. import lib
    device = torch.device(device)
    counts = [int(np.asarray(f).reshape(-1, 5).shape[0]) for f in files]
    total, nf = int(sum(counts)), len(files)
    # staging layout: raw rows | file_off (i64) | xform (f64 [nf,12]) | time_lag (f64) | has_xform (u8)
    o_off = total * 20
    o_off += (-o_off) % 8
    o_xf = o_off + 8 * (nf + 1)
    o_tl = o_xf + 96 * nf
    o_has = o_tl + 8 * nf
    nbytes = o_has + nf
    if pinned is None or pinned.numel() < nbytes:
        pinned = torch.empty(max(nbytes, 1), dtype=torch.uint8).pin_memory()
    host = pinned.numpy()
    raw = host[: total * 20].view(np.float32).reshape(total, 5)
    r = 0
    for f, c in zip(files, counts):
        raw[r:r + c] = np.asarray(f, dtype=np.float32).reshape(-1, 5)
        r += c
    off = host[o_off:o_xf].view(np.int64)
    off[0] = 0
    off[1:] = np.cumsum(counts)
    xf = host[o_xf:o_tl].view(np.float64).reshape(nf, 12)
    has = host[o_has:o_has + nf]
    for i, t in enumerate(xforms):
        has[i] = 0 if t is None else 1
        xf[i] = 0.0 if t is None else np.asarray(t, dtype=np.float64)[:3, :].reshape(12)
    host[o_tl:o_has].view(np.float64)[:] = np.asarray(time_lags, dtype=np.float64)
    dev = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
    dev[:nbytes].copy_(pinned[:nbytes], non_blocking=True)
    out = torch.empty((max(total, 1), 5), dtype=torch.float32, device=device)
    cnt = torch.zeros(1, dtype=torch.int32, device=device)
    ws = torch.empty(lib.load().al3d_merge_sweeps_workspace_bytes(total), dtype=torch.uint8, device=device)
    base = dev.data_ptr()
    lib.call("al3d_merge_sweeps_f32", base, base + o_off, nf, total, base + o_xf, base + o_has, base + o_tl,
             float(min_distance), out.data_ptr(), cnt.data_ptr(), ws.data_ptr(),
             torch.cuda.current_stream(device).cuda_stream)
    return out[: int(cnt.item())]


def load_frame_points_device(info, device, nsweeps=10, root=None, rng=None, pinned=None):
    """``load_frame_points`` with the arithmetic on the device (same argument meaning)."""
    import os

    def p(x):
        return x if root is None or os.path.isabs(str(x)) else os.path.join(root, str(x))
    assert nsweeps - 1 <= len(info["sweeps"]), \
        f"nsweeps {nsweeps} should not greater than list length {len(info['sweeps'])}."
    order = range(nsweeps - 1) if rng is None else rng.choice(len(info["sweeps"]), nsweeps - 1, replace=False)
    files, xforms, lags = [_raw_rows(p(info["lidar_path"]))], [None], [0.0]
    for i in order:
        sw = info["sweeps"][i]
        files.append(_raw_rows(p(sw["lidar_path"])))
        xforms.append(sw.get("transform_matrix"))
        lags.append(float(sw["time_lag"]))
    return merge_sweeps_device(files, xforms, lags, device, pinned=pinned)
