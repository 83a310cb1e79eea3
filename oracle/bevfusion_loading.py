"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU restatement (numpy) of the test branch of BEVFusion's data pipeline, the checker for csrc/images.hip, the BEVFusion rule
of csrc/sweeps.hip and ``al3d.datasets.camera_files``:

* ``pil_resize``: Pillow's convolution resize on 8-bit RGB (``Image.resize(size)`` = BICUBIC; Pillow 8.4.0 pinned by
  bevfusion/README.md:71).  Pillow is not part of /root/reference; this restates its published algorithm
  (src/libImaging/Resample.c: ``precompute_coeffs``, ``normalize_coeffs_8bpc``, horizontal then vertical 8bpc passes with an
  8-bit intermediate image) and tests/test_image_ops.py pins it against the Pillow installed in the container, bit for bit.
* ``image_aug_test``: ImageAug3D with ``is_train=False`` (bevfusion/mmdet3d/datasets/pipelines/transforms_3d.py:37-62 the
  parameters, :64-96 the transform and its 4 x 4 matrix), ``image_normalize`` = ImageNormalize (:903-920: torchvision
  ToTensor + Normalize written out in float32).  Pinned by tests/golden/bevfusion_image_aug.npz, generated from the
  reference's own ImageAug3D class (oracle/gen_golden_bevfusion_loading.py).
* ``merge_sweeps``: LoadPointsFromMultiSweeps (loading.py:176-237) in list order (the reference draws a random subset when
  more than ``sweeps_num`` sweeps exist and test_mode is unset: not reproducible, as det3d's loader).  Pinned by
  tests/golden/bevfusion_sweeps.npz from the reference class.
* ``camera_matrices``: NuScenesDataset.get_data_info (bevfusion/mmdet3d/datasets/nuscenes_dataset.py:233-275).
"""
import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _bilinear(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def resample_coeffs(in_size, out_size, filt=3):
    """(bounds [out,2] i32, coeffs [out,ksize] i32): Resample.c precompute_coeffs + normalize_coeffs_8bpc."""
    fn, fsupport = (_bicubic, 2.0) if filt == 3 else (_bilinear, 1.0)
    in0, in1 = 0.0, float(np.float32(in_size))
    scale = filterscale = (in1 - in0) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = fsupport * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    coeffs = np.zeros((out_size, ksize), np.int32)
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = np.zeros(ksize, np.float64)
        ww = 0.0
        for x in range(xmax):
            w = fn((x + xmin - center + 0.5) * ss)
            k[x] = w
            ww += w
        if ww != 0.0:
            k[:xmax] /= ww
        bounds[xx] = (xmin, xmax)
        for x in range(ksize):
            v = k[x] * (1 << PRECISION_BITS)
            coeffs[xx, x] = int(-0.5 + v) if k[x] < 0 else int(0.5 + v)
    return bounds, coeffs


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def pil_resize(img, out_w, out_h, filt=3):
    """img [H,W,3] u8 -> [out_h,out_w,3] u8: horizontal pass over the rows the vertical pass needs, then vertical."""
    H, W, _ = img.shape
    hb, hk = resample_coeffs(W, out_w, filt)
    vb, vk = resample_coeffs(H, out_h, filt)
    first, last = int(vb[0, 0]), int(vb[-1, 0] + vb[-1, 1])
    src = img[first:last].astype(np.int64)
    temp = np.empty((last - first, out_w, 3), np.uint8)
    for xx in range(out_w):
        x0, n = hb[xx]
        acc = (1 << (PRECISION_BITS - 1)) + (src[:, x0:x0 + n, :] * hk[xx, :n, None].astype(np.int64)).sum(1)
        temp[:, xx] = _clip8(acc)
    out = np.empty((out_h, out_w, 3), np.uint8)
    t64 = temp.astype(np.int64)
    for yy in range(out_h):
        y0, n = vb[yy]
        y0 -= first
        acc = (1 << (PRECISION_BITS - 1)) + (t64[y0:y0 + n] * vk[yy, :n, None, None].astype(np.int64)).sum(0)
        out[yy] = _clip8(acc)
    return out


def image_aug_params(ori_w, ori_h, final_dim=(256, 704), resize_lim=(0.48, 0.48), bot_pct_lim=(0.0, 0.0)):
    """ImageAug3D.sample_augmentation with is_train=False: (resize, resize_dims (W, H), crop (x0, y0, x1, y1))."""
    fH, fW = final_dim
    resize = np.mean(resize_lim)
    resize_dims = (int(ori_w * resize), int(ori_h * resize))
    newW, newH = resize_dims
    crop_h = int((1 - np.mean(bot_pct_lim)) * newH) - fH
    crop_w = int(max(0, newW - fW) / 2)
    return resize, resize_dims, (crop_w, crop_h, crop_w + fW, crop_h + fH)


def image_aug_matrix(resize, crop):
    """The 4 x 4 ``img_aug_matrix`` of the test branch (no flip, rotate 0), float32 like the reference's torch code."""
    rot = np.eye(2, dtype=np.float32) * np.float32(resize)
    tran = np.zeros(2, np.float32) - np.array(crop[:2], np.float32)
    theta = np.float32(0.0 / 180 * np.pi)
    A = np.array([[np.cos(theta), np.sin(theta)], [-np.sin(theta), np.cos(theta)]], dtype=np.float32)
    b = np.array([crop[2] - crop[0], crop[3] - crop[1]], np.float32) / 2
    b = A @ (-b) + b
    rot = A @ rot
    tran = A @ tran + b
    m = np.eye(4, dtype=np.float32)
    m[:2, :2] = rot
    m[:2, 3] = tran
    return m


def image_normalize(u8, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """[..., 3] u8 -> float32 ((x / 255) - mean) / std, channels last."""
    x = u8.astype(np.float32) / np.float32(255.0)
    return (x - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)


def image_aug_test(img, final_dim=(256, 704), resize_lim=(0.48, 0.48), bot_pct_lim=(0.0, 0.0)):
    """[H,W,3] u8 -> (cropped u8 [fH,fW,3], img_aug_matrix)."""
    H, W, _ = img.shape
    resize, dims, crop = image_aug_params(W, H, final_dim, resize_lim, bot_pct_lim)
    r = pil_resize(img, dims[0], dims[1])
    return r[crop[1]:crop[3], crop[0]:crop[2]], image_aug_matrix(resize, crop)


def merge_sweeps(key_points, sweeps, ts, sweeps_num=9, pad_empty_sweeps=True, remove_close=True, radius=1.0):
    """key_points [P,5] f32; sweeps: list of dict(points [p,5] f32, timestamp, sensor2lidar_rotation [3,3] f64,
    sensor2lidar_translation [3] f64); ts: the key frame's timestamp (microseconds).  -> [N,5] f32."""
    def not_close(p):
        return p[~((np.abs(p[:, 0]) < radius) & (np.abs(p[:, 1]) < radius))]
    pts = key_points.astype(np.float32).copy()
    pts[:, 4] = 0
    out = [pts]
    ts = ts / 1e6
    if pad_empty_sweeps and len(sweeps) == 0:
        for _ in range(sweeps_num):
            out.append(not_close(pts) if remove_close else pts)
    else:
        for sw in sweeps[:sweeps_num]:
            p = sw["points"].astype(np.float32).copy()
            if remove_close:
                p = not_close(p)
            p[:, :3] = p[:, :3] @ np.asarray(sw["sensor2lidar_rotation"]).T
            p[:, :3] += np.asarray(sw["sensor2lidar_translation"])
            p[:, 4] = ts - sw["timestamp"] / 1e6
            out.append(p)
    return np.concatenate(out, 0)


def points_range_filter(points, point_cloud_range):
    """PointsRangeFilter (transforms_3d.py:503-525 on base_points.py:208-232): rows STRICTLY inside the float32 range."""
    r = np.asarray(point_cloud_range, dtype=np.float32)
    p = points
    keep = (p[:, 0] > r[0]) & (p[:, 1] > r[1]) & (p[:, 2] > r[2]) & (p[:, 0] < r[3]) & (p[:, 1] < r[4]) & (p[:, 2] < r[5])
    return p[keep]


def camera_matrices(cam):
    """One camera's (lidar2image, camera_intrinsics, camera2lidar) 4 x 4 float32 as get_data_info builds them."""
    l2c_r = np.linalg.inv(cam["sensor2lidar_rotation"])
    l2c_t = cam["sensor2lidar_translation"] @ l2c_r.T
    l2c = np.eye(4).astype(np.float32)
    l2c[:3, :3] = l2c_r.T
    l2c[3, :3] = -l2c_t
    K = np.eye(4).astype(np.float32)
    K[:3, :3] = cam["camera_intrinsics"]
    l2i = K @ l2c.T
    c2l = np.eye(4).astype(np.float32)
    c2l[:3, :3] = cam["sensor2lidar_rotation"]
    c2l[:3, 3] = cam["sensor2lidar_translation"]
    return l2i, K, c2l
