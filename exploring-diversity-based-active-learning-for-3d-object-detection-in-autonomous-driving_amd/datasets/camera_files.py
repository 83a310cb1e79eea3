"""configs[4] from files: BEVFusion camera+lidar samples read from an mmdet3d-format nuScenes pool.

The test branch of the reference's pipeline (bevfusion/configs/nuscenes/default.yaml:186-243):
``LoadMultiViewImageFromFiles`` (loading.py:19-83) -> ``LoadPointsFromFile`` / ``LoadPointsFromMultiSweeps``
(loading.py:84-237; sweeps_num 9, pad_empty_sweeps, remove_close) -> ``PointsRangeFilter`` (transforms_3d.py:503-525:
merged points strictly inside ``point_cloud_range``; it is what keeps far / high returns out of the view transform's
lidar depth image, base.py:213-262) -> ``ImageAug3D`` (transforms_3d.py:26-122, is_train
False: resize 0.48, bottom crop to 256 x 704, the ``img_aug_matrix`` it emits) -> ``GlobalRotScaleTrans`` (identity in test
mode: ``lidar_aug_matrix`` = I) -> ``ImageNormalize`` (transforms_3d.py:903-920), with the calibration matrices of
``NuScenesDataset.get_data_info`` (nuscenes_dataset.py:233-275).

Here: the lidar side streams through the native reader pool and ``al3d_merge_sweeps_batch_rule_f32`` (rule 1: BEVFusion's
two-step float64 -> float32 transform).  Camera frames (round 5): a baseline JPEG is SPLIT -- its Huffman entropy decoding
runs on host threads (``al3d_jpeg_entropy_decode``, straight into a pinned buffer), the quantised coefficients are uploaded
and the device does the inverse DCT, chroma upsampling and colour conversion (``al3d_jpeg_idct_rgb_u8``: the bytes Pillow's
decoder gives, tests/test_jpeg_gpu.py); any other file (PNG, progressive JPEG, ...) and any batch that mixes geometries is
decoded by Pillow on the same threads (imported lazily) and uploaded as 8-bit RGB.  Either way the frames are then resized /
cropped / normalised by ``al3d_image_aug_normalize_u8``
(PIL's bicubic resize restated as an integer kernel: the same bytes as ``Image.resize``) into the channels-last float32
layout the token kernels read.  Sweep order = list order (the reference draws a random subset of nine when a sample lists
more and test_mode is unset: not reproducible, like det3d's loader).
"""
import ctypes
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from .. import lib
from .file_loader import FileSweepLoader


def resample_tables(in_size, out_size, filt=3):
    """PIL's filter windows for one axis: (bounds [out,2] i32, coeffs [out,ksize] i32) from the library's host function."""
    L = lib.load()
    ks = int(L.al3d_image_resample_ksize(int(in_size), int(out_size), int(filt)))
    if ks <= 0:
        raise lib.Al3dError(f"resample_tables: bad sizes {in_size} -> {out_size}")
    b = np.zeros((out_size, 2), np.int32)
    c = np.zeros((out_size, ks), np.int32)
    lib.call("al3d_image_resample_coeffs", int(in_size), int(out_size), int(filt), b.ctypes.data_as(ctypes.c_void_p),
             c.ctypes.data_as(ctypes.c_void_p))
    return b, c


class ImageAugTest:
    """ImageAug3D with ``is_train=False`` + ImageNormalize for frames of one size (W, H): the resize / crop parameters
    (transforms_3d.py:52-61), the 4 x 4 ``img_aug_matrix`` (:64-96, :107-118) and the device call."""

    def __init__(self, ori_size, final_dim=(256, 704), resize_lim=(0.48, 0.48), bot_pct_lim=(0.0, 0.0),
                 mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), device="cuda"):
        W, H = int(ori_size[0]), int(ori_size[1])
        fH, fW = int(final_dim[0]), int(final_dim[1])
        resize = np.mean(resize_lim)
        newW, newH = int(W * resize), int(H * resize)
        crop_h = int((1 - np.mean(bot_pct_lim)) * newH) - fH
        crop_w = int(max(0, newW - fW) / 2)
        if crop_h < 0 or crop_w + fW > newW:
            raise lib.Al3dError(f"ImageAugTest: final_dim {final_dim} does not fit the resized frame {newW} x {newH} "
                                "(PIL would pad the crop with black: not built)")
        self.W, self.H, self.fH, self.fW, self.newW, self.newH = W, H, fH, fW, newW, newH
        self.resize, self.crop = resize, (crop_w, crop_h, crop_w + fW, crop_h + fH)
        # post-homography of the test branch: rotation = resize * I, translation = -crop[:2]; flip off, rotate 0 (the
        # reference's float32 torch arithmetic: A = I, b = A(-b) + b = 0)
        m = np.eye(4, dtype=np.float32)
        m[0, 0] = m[1, 1] = np.float32(resize)
        m[0, 3], m[1, 3] = np.float32(-crop_w), np.float32(-crop_h)
        self.matrix = m
        self.device = torch.device(device)
        hb, hk = resample_tables(W, newW)
        vb, vk = resample_tables(H, newH)
        self.row_first = int(vb[crop_h, 0])
        self.row_count = int(vb[crop_h + fH - 1, 0] + vb[crop_h + fH - 1, 1]) - self.row_first
        self._tab = [torch.from_numpy(a).to(self.device) for a in (hb, hk, vb, vk)]
        self._ks = (hk.shape[1], vk.shape[1])
        self._mean = (ctypes.c_float * 3)(*[float(v) for v in mean])
        self._std = (ctypes.c_float * 3)(*[float(v) for v in std])

    def __call__(self, frames_u8, want_u8=False):
        """frames_u8 [n, H, W, 3] uint8 device tensor -> [n, fH, fW, 3] float32 (and the 8-bit crop when asked)."""
        from ..selector_ops import _ptr, _stream
        x = frames_u8
        if x.dtype != torch.uint8 or not x.is_cuda or x.dim() != 4 or tuple(x.shape[1:]) != (self.H, self.W, 3):
            raise lib.Al3dError(f"ImageAugTest: expected a uint8 device tensor [n, {self.H}, {self.W}, 3]")
        x = x.contiguous()
        n = x.shape[0]
        out = torch.empty((n, self.fH, self.fW, 3), dtype=torch.float32, device=x.device)
        u8 = torch.empty((n, self.fH, self.fW, 3), dtype=torch.uint8, device=x.device) if want_u8 else None
        ws = torch.empty(max(int(lib.load().al3d_image_aug_workspace_bytes(n, self.row_count, self.fW)), 1),
                         dtype=torch.uint8, device=x.device)
        hb, hk, vb, vk = self._tab
        lib.call("al3d_image_aug_normalize_u8", _ptr(x), n, self.H, self.W, self.newH, self.newW, self.crop[0], self.crop[1],
                 self.fH, self.fW, _ptr(hb), _ptr(hk), self._ks[0], _ptr(vb), _ptr(vk), self._ks[1], self._mean, self._std,
                 self.row_first, self.row_count, _ptr(ws), _ptr(out), _ptr(u8), _stream())
        return (out, u8) if want_u8 else out


def camera_matrices(cam):
    """(lidar2image, camera_intrinsics, camera2lidar) 4 x 4 float32 of one ``info["cams"][name]`` entry, built as
    NuScenesDataset.get_data_info does (nuscenes_dataset.py:241-275)."""
    l2c_r = np.linalg.inv(cam["sensor2lidar_rotation"])
    l2c_t = cam["sensor2lidar_translation"] @ l2c_r.T
    l2c = np.eye(4).astype(np.float32)
    l2c[:3, :3] = l2c_r.T
    l2c[3, :3] = -l2c_t
    K = np.eye(4).astype(np.float32)
    K[:3, :3] = cam["camera_intrinsics"]
    c2l = np.eye(4).astype(np.float32)
    c2l[:3, :3] = cam["sensor2lidar_rotation"]
    c2l[:3, 3] = cam["sensor2lidar_translation"]
    return K @ l2c.T, K, c2l


def _decode_rgb(path):
    """One camera frame -> [H, W, 3] uint8 (PIL: ``Image.open`` like the reference's loader; RGB)."""
    from PIL import Image                                    # lazy: only real camera pools need a decoder
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))


def jpeg_header(data):
    """(info [32] int32, quant [3, 64] uint16) of a baseline JPEG in memory, or None when the split decoder does not take it."""
    info = np.zeros(32, np.int32)
    quant = np.zeros((3, 64), np.uint16)
    rc = lib.load().al3d_jpeg_header(data.ctypes.data_as(ctypes.c_void_p), int(data.size), info.ctypes.data_as(ctypes.c_void_p),
                                     quant.ctypes.data_as(ctypes.c_void_p))
    return (info, quant) if rc == 0 else None


def _decode_jpeg_into(path, want_info, coef_ptr, coef_blocks, quant_out):
    """Worker-thread job: entropy-decode one camera frame into its slice of the batch's pinned coefficient buffer.
    Returns True, or the decoded RGB array (Pillow) when the file is not a baseline JPEG of the batch's geometry."""
    data = np.fromfile(path, dtype=np.uint8)
    hdr = jpeg_header(data) if data.size > 4 and data[0] == 0xff and data[1] == 0xd8 else None
    if hdr is not None and np.array_equal(hdr[0][:24], want_info[:24]):
        rc = lib.load().al3d_jpeg_entropy_decode(data.ctypes.data_as(ctypes.c_void_p), int(data.size), ctypes.c_void_p(coef_ptr),
                                                 int(coef_blocks))
        if rc == 0:
            quant_out[:] = hdr[1]
            return True
    return _decode_rgb(path)


class CameraLidarFileLoader(FileSweepLoader):
    """Batches of BEVFusion ``example`` dicts (``al3d.models.bevfusion_model.CAMERA_KEYS`` + the lidar keys) from an
    mmdet3d-format pool: ``infos[i]`` carries ``lidar_path``, ``timestamp``, ``sweeps[k]{data_path, timestamp,
    sensor2lidar_rotation, sensor2lidar_translation}`` and ``cams{name: {data_path, sensor2lidar_rotation,
    sensor2lidar_translation, camera_intrinsics}}`` (bevfusion/tools/data_converter/nuscenes_converter.py's schema)."""

    rule = 1                                                 # BEVFusion's two-step float64 -> float32 sweep transform

    def __init__(self, infos, voxel_cfg, anchors, batch_size=4, device="cuda", sweeps_num=9, root=None, threads=8,
                 indices=None, depth=2, min_distance=1.0, image_size=(256, 704), resize_lim=(0.48, 0.48),
                 bot_pct_lim=(0.0, 0.0), mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), decode_threads=None,
                 pad_empty_sweeps=True, point_cloud_range="voxel", jpeg="split"):
        super().__init__(infos, voxel_cfg, anchors, batch_size=batch_size, device=device, nsweeps=sweeps_num + 1, root=root,
                         threads=threads, indices=indices, depth=depth, min_distance=min_distance)
        self.sweeps_num, self.pad_empty_sweeps = int(sweeps_num), bool(pad_empty_sweeps)
        # PointsRangeFilter: the pipeline's ${point_cloud_range} is the voxelizer's range in every shipped config
        # (default.yaml:233-235 / voxelnet_0p075: [-54, -54, -5, 54, 54, 3]); None switches the filter off
        if isinstance(point_cloud_range, str):
            point_cloud_range = list(voxel_cfg["range"] if isinstance(voxel_cfg, dict) else voxel_cfg.range)
        self.point_range = None if point_cloud_range is None else [float(v) for v in point_cloud_range]
        if self.point_range is not None and len(self.point_range) != 6:
            raise lib.Al3dError("CameraLidarFileLoader: point_cloud_range = [x_min, y_min, z_min, x_max, y_max, z_max]")
        self.image_size, self._aug_cfg = tuple(image_size), (tuple(resize_lim), tuple(bot_pct_lim), tuple(mean), tuple(std))
        self._aug = {}
        from .file_loader import usable_cores
        self._pool = ThreadPoolExecutor(max_workers=int(decode_threads or max(2, usable_cores() - 2)))
        self._img_pinned = {}
        self.images_decoded = 0
        # jpeg = "split": baseline JPEGs are entropy-decoded on the host and finished on the device; "pil": everything by Pillow
        if jpeg not in ("split", "pil"):
            raise lib.Al3dError("CameraLidarFileLoader: jpeg = 'split' or 'pil'")
        self.jpeg = jpeg
        self._jpeg_info = None                               # geometry of the pool's frames (from the first one that parses)
        self._coef_pinned = {}
        self.images_split = 0

    # ---- lidar side: LoadPointsFromFile + LoadPointsFromMultiSweeps (test branch)
    def _frame_files(self, info):
        key = self._path(info["lidar_path"])
        paths, xforms, lags, keys = [key], [None], [0.0], [1]
        sweeps = info.get("sweeps", [])
        ts = info["timestamp"] / 1e6
        if self.pad_empty_sweeps and len(sweeps) == 0:
            for _ in range(self.sweeps_num):                 # the key frame again, remove_close applied, time 0
                paths.append(key); xforms.append(None); lags.append(0.0); keys.append(0)
        else:
            for sw in sweeps[:self.sweeps_num]:
                m = np.zeros((3, 4), np.float64)
                m[:, :3] = np.asarray(sw["sensor2lidar_rotation"], np.float64)
                m[:, 3] = np.asarray(sw["sensor2lidar_translation"], np.float64)
                paths.append(self._path(sw["data_path"]))
                xforms.append(m)
                lags.append(ts - sw["timestamp"] / 1e6)
                keys.append(0)
        return paths, xforms, lags, keys

    # ---- camera side
    def _start(self, b):
        st = super()._start(b)
        ids = self.indices[b * self.batch_size:(b + 1) * self.batch_size]
        paths = [[self._path(c["data_path"]) for c in self.infos[i]["cams"].values()] for i in ids]
        nimg = sum(len(p) for p in paths)
        info = None
        if self.jpeg == "split" and nimg:
            if self._jpeg_info is None:
                first = paths[0][0]
                if first.lower().endswith((".jpg", ".jpeg")):
                    hdr = jpeg_header(np.fromfile(first, dtype=np.uint8))
                    self._jpeg_info = hdr[0] if hdr is not None else False
                else:
                    self._jpeg_info = False
            info = self._jpeg_info if self._jpeg_info is not False else None
        if info is None:
            st.extra = dict(mode="pil", jobs=[[self._pool.submit(_decode_rgb, p) for p in cams] for cams in paths])
            return st
        # split path: every job writes its coefficients into its slice of the slot's pinned buffer (free: the parent has
        # just waited for the slot's previous upload, and ours is recorded on the same slot below)
        nb = int(info[20])
        held = self._coef_pinned.get(st.slot)
        if held is not None and held[2] is not None:
            held[2].synchronize()
        if held is None or held[0].shape[0] < nimg or held[0].shape[1] != nb:
            held = [torch.empty((nimg, nb, 64), dtype=torch.int16).pin_memory(),
                    torch.empty((nimg, 3, 64), dtype=torch.int16).pin_memory(), None]
            self._coef_pinned[st.slot] = held
        coefs, quant = held[0], held[1]
        qv = quant.numpy().view(np.uint16)
        jobs, k = [], 0
        for cams in paths:
            row = []
            for pth in cams:
                row.append(self._pool.submit(_decode_jpeg_into, pth, info, coefs[k].data_ptr(), nb, qv[k]))
                k += 1
            jobs.append(row)
        st.extra = dict(mode="split", jobs=jobs, info=info, nimg=nimg)
        return st

    def _abandon(self, st):
        """Iteration ended before this batch was finished: cancel the decode jobs that have not started, drop the rest."""
        for cam_jobs in (getattr(st, "extra", None) or {}).get("jobs", []):
            for f in cam_jobs:
                f.cancel()
        st.extra = None

    def _finish(self, st):
        ex = super()._finish(st)
        extra, st.extra = st.extra, None
        jobs = extra["jobs"]
        results = [[f.result() for f in cam_jobs] for cam_jobs in jobs]
        B, N = len(results), len(results[0])
        if any(len(fr) != N for fr in results):
            raise lib.Al3dError("CameraLidarFileLoader: every sample of a batch must list the same number of cameras")
        split = extra["mode"] == "split" and all(r is True for fr in results for r in fr)
        if split:
            # coefficients -> device -> RGB bytes (al3d_jpeg_idct_rgb_u8: what Pillow's decoder gives)
            from ..selector_ops import _ptr, _stream
            info, nimg = extra["info"], extra["nimg"]
            W, H = int(info[0]), int(info[1])
            held = self._coef_pinned[st.slot]
            dcoef = held[0][:nimg].to(self.device, non_blocking=True)
            dquant = held[1][:nimg].to(self.device, non_blocking=True)
            held[2] = torch.cuda.Event()
            held[2].record(torch.cuda.current_stream(self.device))
            L = lib.load()
            ws = torch.empty(max(int(L.al3d_jpeg_workspace_bytes(info.ctypes.data_as(ctypes.c_void_p), nimg)), 16),
                             dtype=torch.uint8, device=self.device)
            dev_u8 = torch.empty((nimg, H, W, 3), dtype=torch.uint8, device=self.device)
            lib.call("al3d_jpeg_idct_rgb_u8", _ptr(dcoef), _ptr(dquant), info.ctypes.data_as(ctypes.c_void_p), nimg, _ptr(dev_u8),
                     _ptr(ws), _stream())
            self.images_split += nimg
        else:
            # Pillow's pixels (a batch with any frame the split decoder did not take is finished by Pillow as a whole)
            frames = []
            for fr, cams in zip(results, [list(self.infos[i]["cams"].values()) for i in st.ids]):
                frames.append([a if a is not True else _decode_rgb(self._path(c["data_path"])) for a, c in zip(fr, cams)])
            H, W = frames[0][0].shape[:2]
            # pinned staging, one buffer per in-flight slot of the parent's ring (refilled only after its upload has left)
            need = B * N * H * W * 3
            held = self._img_pinned.get(st.slot)
            if held is not None and held[1] is not None:
                held[1].synchronize()
            if held is None or held[0].numel() < need:
                held = [torch.empty(need, dtype=torch.uint8).pin_memory(), None]
                self._img_pinned[st.slot] = held
            host = held[0][:need].view(B * N, H, W, 3)
            hv = host.numpy()
            for bi, fr in enumerate(frames):
                for ci, a in enumerate(fr):
                    if a.shape != (H, W, 3):
                        raise lib.Al3dError(f"CameraLidarFileLoader: camera frames of one batch differ in size ({a.shape} vs {(H, W, 3)})")
                    hv[bi * N + ci] = a
            dev_u8 = host.to(self.device, non_blocking=True)
            held[1] = torch.cuda.Event()
            held[1].record(torch.cuda.current_stream(self.device))
        self.images_decoded += B * N
        aug = self._aug.get((W, H))
        if aug is None:
            rl, bp, mean, std = self._aug_cfg
            aug = self._aug[(W, H)] = ImageAugTest((W, H), self.image_size, rl, bp, mean, std, device=self.device)
        img = aug(dev_u8)
        l2i, K, c2l = [], [], []
        for i in st.ids:
            mats = [camera_matrices(c) for c in self.infos[i]["cams"].values()]
            l2i.append(np.stack([m[0] for m in mats]))
            K.append(np.stack([m[1] for m in mats]))
            c2l.append(np.stack([m[2] for m in mats]))

        def dev(a):
            return torch.from_numpy(np.ascontiguousarray(np.stack(a), dtype=np.float32)).to(self.device)
        ex["img"] = img.view(B, N, *img.shape[1:])
        ex["lidar2image"], ex["camera_intrinsics"], ex["camera2lidar"] = dev(l2i), dev(K), dev(c2l)
        # digest of the host bytes every calibration tensor of this batch was uploaded from (the view transform's plan cache
        # compares it on the host: models/bevfusion_camera.py)
        import hashlib
        hk = hashlib.sha1()
        for a in (l2i, K, c2l):
            hk.update(np.ascontiguousarray(np.stack(a), dtype=np.float32).tobytes())
        hk.update(aug.matrix.tobytes() + bytes([B, N]))
        ex["calib_key"] = hk.digest()
        ex["img_aug_matrix"] = torch.from_numpy(aug.matrix).to(self.device).expand(B, N, 4, 4).contiguous()
        ex["lidar_aug_matrix"] = torch.eye(4, device=self.device).expand(B, 4, 4).contiguous()
        off = ex["point_offsets"].cpu().tolist()             # per-sample clouds for the view transform's depth image
        ex["points"] = [ex["points"][off[k]:off[k + 1]] for k in range(B)]
        return ex
