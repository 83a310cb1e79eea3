"""Device-resident unlabeled pool + sweep loader.

The reference feeds the sweep through 8 DataLoader worker processes that read ten ``.bin``
sweeps per frame, voxelize with numpy/numba and regenerate the anchors for every sample
(SURVEY 3.2).  Here the pool's point clouds live in HBM (288 GB holds the whole nuScenes
train split), voxelization runs on device per batch, and anchors are constants.

``PoolFrames`` is the storage: a list of ``[P_i,5]`` float32 device tensors.  ``from_synthetic``
builds nuScenes-shaped frames (there is no dataset offline): a handful of ring-scan base clouds
from ``synthetic.make_point_cloud`` rotated/shifted per frame so every frame is distinct.
``from_files`` reads real 10-sweep frames through the reference's loading rules (a1).
"""
import math

import numpy as np
import torch

from .. import synthetic
from ..detector_ops import Voxelizer


class PoolFrames:
    def __init__(self, frames, tokens=None):
        self.frames = frames
        self.tokens = tokens or [f"frame{i:06d}" for i in range(len(frames))]
        self.flat = None            # optional: all frames back to back in one tensor (see pack())
        self.offsets = None         # python list, len(frames) + 1

    def pack(self):
        """Store the frames back to back in one tensor (``frames`` become views of it) so that a batch
        of consecutive frames is a slice instead of a ``torch.cat`` copy (160 MB per 32-frame batch)."""
        if self.flat is None and len(self.frames) > 0:
            counts = [int(f.shape[0]) for f in self.frames]
            self.offsets = [0]
            for c in counts:
                self.offsets.append(self.offsets[-1] + c)
            self.flat = torch.cat(self.frames, dim=0)
            self.frames = [self.flat[self.offsets[i]:self.offsets[i + 1]] for i in range(len(counts))]
        return self

    def __len__(self):
        return len(self.frames)

    @classmethod
    def from_synthetic(cls, num_frames, device, num_base=16, seed=0, nsweeps=10):
        base = [torch.from_numpy(synthetic.make_point_cloud(1000 + b, nsweeps=nsweeps)).to(device)
                for b in range(min(num_base, max(1, num_frames)))]
        g = torch.Generator(device="cpu").manual_seed(seed)
        yaw = torch.rand(num_frames, generator=g) * 2 * math.pi
        shift = torch.randn(num_frames, 2, generator=g) * 0.5
        frames = []
        for i in range(num_frames):
            p = base[i % len(base)].clone()
            c, s = math.cos(float(yaw[i])), math.sin(float(yaw[i]))
            x, y = p[:, 0].clone(), p[:, 1].clone()
            p[:, 0] = c * x - s * y + float(shift[i, 0])
            p[:, 1] = s * x + c * y + float(shift[i, 1])
            frames.append(p.contiguous())
        return cls(frames).pack()

    @classmethod
    def from_files(cls, infos, device, nsweeps=10, root=None):
        """Real nuScenes frames: raw sweep files -> pinned staging -> device merge kernel (a1)."""
        from .nusc_files import load_frame_points_device
        frames = [load_frame_points_device(info, device, nsweeps=nsweeps, root=root) for info in infos]
        return cls(frames, tokens=[str(i.get("token", f"frame{k:06d}")) for k, i in enumerate(infos)]).pack()

    @classmethod
    def from_numpy(cls, arrays, device):
        return cls([torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
                    for a in arrays])


class DeviceSweepLoader:
    """Iterates the pool in dataset order (or a rank's shard) yielding ``example`` dicts with the
    keys the detector reads (voxelnet.py:84-97, mg_head.py:710-724): ``voxel_features``,
    ``coordinates``, ``num_points``, ``num_voxels``, ``shape``, ``anchors``, ``metadata``."""

    def __init__(self, pool, voxel_cfg, anchors, batch_size=4, indices=None, device="cuda"):
        self.pool = pool
        self.batch_size = int(batch_size)
        self.device = torch.device(device)
        self.indices = list(range(len(pool))) if indices is None else list(indices)
        self.voxelizer = Voxelizer(voxel_cfg["range"], voxel_cfg["voxel_size"],
                                   voxel_cfg["max_points_in_voxel"], voxel_cfg["max_voxel_num"],
                                   max_batch=self.batch_size, device=self.device)
        # anchors=None: embedding-only models (no detection head)
        self.anchors = [torch.as_tensor(a, dtype=torch.float32, device=self.device) for a in (anchors or [])]
        self.dataset = pool          # len(loader.dataset) like a torch DataLoader
        self.sampler = self.indices

    def __len__(self):
        return (len(self.indices) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        gs = self.voxelizer.grid_size
        for s in range(0, len(self.indices), self.batch_size):
            ids = self.indices[s:s + self.batch_size]
            frames = [self.pool.frames[i] for i in ids]
            off = torch.tensor([0] + list(np.cumsum([f.shape[0] for f in frames])), dtype=torch.int64,
                               device=self.device)
            consecutive = all(b == a + 1 for a, b in zip(ids[:-1], ids[1:]))
            if self.pool.flat is not None and consecutive:
                pts = self.pool.flat[self.pool.offsets[ids[0]]:self.pool.offsets[ids[-1] + 1]]
            else:
                pts = torch.cat(frames, dim=0) if len(frames) > 1 else frames[0]
            v = self.voxelizer(pts, off)
            B = len(ids)
            yield {
                "voxel_features": v["feat"], "coordinates": v["coords"], "num_points": v["num_points"],
                "num_voxels": v["num_voxels"], "voxel_cap": v["voxel_cap"],
                "shape": np.tile(np.asarray(gs, dtype=np.int64)[None], (B, 1)),
                "anchors": self.anchors,
                "metadata": [{"token": self.pool.tokens[i], "index": i} for i in ids],
            }


class CameraLidarSweepLoader(DeviceSweepLoader):
    """``DeviceSweepLoader`` + the camera side of a BEVFusion sample (``al3d.models.bevfusion_model.CAMERA_KEYS``): channels-
    last images ``img [B, N, H, W, 3]``, the per-sample point clouds (the view transform rasterises them into the depth
    image), and the 4 x 4 matrices ``lidar2image``, ``camera_intrinsics``, ``camera2lidar``, ``img_aug_matrix`` ``[B, N, 4, 4]``
    and ``lidar_aug_matrix [B, 4, 4]`` (bevfusion/mmdet3d/models/fusion_models/bevfusion.py:165-205's arguments).

    ``images``: ``[M, N, H, W, 3]`` float32 device tensor, frame i uses ``images[i % M]`` (a real pool would hold M = frames);
    ``calib``: dict of the five matrices with a leading dimension of 1 (one rig) or M.  Both default to seeded synthetic
    content (``synthetic.camera_setup``; there is no nuScenes camera data offline)."""

    def __init__(self, pool, voxel_cfg, anchors, batch_size=4, indices=None, device="cuda", image_size=(256, 704),
                 num_cameras=6, images=None, calib=None, num_image_base=4, seed=0):
        super().__init__(pool, voxel_cfg, anchors, batch_size, indices, device)
        self.image_size, self.num_cameras = tuple(image_size), int(num_cameras)
        if images is None:
            g = torch.Generator().manual_seed(seed)
            images = torch.randn(num_image_base, self.num_cameras, *self.image_size, 3, generator=g)
        self.images = images.to(self.device, dtype=torch.float32).contiguous()
        if calib is None:
            K, cam2lidar, lidar2image, img_aug, lidar_aug, _ = synthetic.camera_setup(1, self.num_cameras, seed + 9, self.image_size)
            calib = dict(camera_intrinsics=K, camera2lidar=cam2lidar, lidar2image=lidar2image, img_aug_matrix=img_aug,
                         lidar_aug_matrix=lidar_aug)
        # digests of the host copies, per calibration row: a batch's ``calib_key`` lets the view transform decide on the
        # host whether its cached pooling plan still applies (no device comparison, no stream synchronisation)
        import hashlib
        host = {k: v.detach().to("cpu", torch.float32).contiguous() for k, v in calib.items()}
        self._calib_digest = {k: [hashlib.sha1(v[r].numpy().tobytes()).digest() for r in range(v.shape[0])]
                              for k, v in host.items()}
        self.calib = {k: v.to(self.device) for k, v in host.items()}

    def __iter__(self):
        M = self.images.shape[0]
        for ex in super().__iter__():
            ids = [m["index"] for m in ex["metadata"]]
            sel = torch.as_tensor([i % M for i in ids], device=self.device)
            ex["img"] = self.images[sel]
            ex["points"] = [self.pool.frames[i] for i in ids]
            key = []
            for k, v in self.calib.items():
                rows = torch.as_tensor([i % v.shape[0] for i in ids], device=self.device)
                ex[k] = v[rows]
                key.append(b"".join(self._calib_digest[k][i % v.shape[0]] for i in ids))
            ex["calib_key"] = b"|".join(key)
            yield ex
