"""Deterministic synthetic nuScenes-shaped inputs.

No dataset exists offline, so parity fixtures, tests and ``bench.py`` all draw
from these generators.  The *schema* is the reference's info-pkl schema
(det3d/datasets/nuscenes/nusc_common.py:410-419): the selectors read only
``car_from_global``, ``cam_front_path`` and ``gt_names``.
"""
import math

import numpy as np

LOCATIONS = ("boston-seaport", "singapore-onenorth", "singapore-queenstown",
             "singapore-hollandvillage")
FRAMES_PER_SCENE = 40


def make_pool(num_scenes, seed=0, frames_per_scene=FRAMES_PER_SCENE,
              step_m=4.0, jitter=0.2, yaw_jitter=0.02, max_boxes=69,
              extent=2000.0):
    """Return ``(infos, logs)`` for a pool of ``num_scenes * frames_per_scene`` frames.

    One logfile per scene, 4 map locations round-robin; each ego track starts
    at U(0, extent)^2 with heading U(0, 2pi) and advances ``step_m`` per frame
    plus N(0, jitter) noise; ``n_boxes ~ U{0..max_boxes}``.
    """
    rng = np.random.default_rng(seed)
    infos, logs = [], []
    for s in range(num_scenes):
        logfile = f"n{s % 16:03d}-2018-{1 + s % 12:02d}-{1 + s % 28:02d}-{s:06d}"
        logs.append({"logfile": logfile, "location": LOCATIONS[s % len(LOCATIONS)],
                     "token": f"log{s:06d}"})
        pos = rng.uniform(0.0, extent, size=2)
        heading = rng.uniform(0.0, 2.0 * math.pi)
        for f in range(frames_per_scene):
            yaw = heading + rng.normal(0.0, yaw_jitter)
            if f > 0:
                pos = pos + step_m * np.array([math.cos(heading), math.sin(heading)]) \
                    + rng.normal(0.0, jitter, size=2)
            c, s_ = math.cos(yaw), math.sin(yaw)
            global_from_car = np.eye(4)
            global_from_car[:3, :3] = np.array([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]])
            global_from_car[:3, 3] = [pos[0], pos[1], 0.0]
            car_from_global = np.linalg.inv(global_from_car)
            n_boxes = int(rng.integers(0, max_boxes + 1))
            ts = 1_530_000_000_000_000 + (s * frames_per_scene + f) * 500_000
            infos.append({
                "token": f"tok{s:05d}_{f:03d}",
                "lidar_path": f"samples/LIDAR_TOP/{logfile}__LIDAR_TOP__{ts}.pcd.bin",
                "cam_front_path": f"samples/CAM_FRONT/{logfile}__CAM_FRONT__{ts}.jpg",
                "car_from_global": car_from_global,
                "timestamp": ts * 1e-6,
                "gt_names": np.array(["car"] * n_boxes, dtype="<U32"),
                "gt_boxes": np.zeros((n_boxes, 9), dtype=np.float32),
            })
    return infos, logs


def pool_arrays(infos):
    """Pull the three things the selectors read into flat arrays.

    Returns ``car_from_global [N,4,4] f64``, ``run_id [N] i64`` (consecutive
    runs of equal logfile, reference spatial_temporal_selector.py:114-129) and
    ``n_boxes [N] i64``.
    """
    cfg = np.stack([np.asarray(i["car_from_global"], dtype=np.float64) for i in infos])
    run_id = np.zeros(len(infos), dtype=np.int64)
    prev, rid = None, -1
    for k, info in enumerate(infos):
        lf = info["cam_front_path"].split("/")[-1].split("__")[0]
        if lf != prev:
            rid += 1
            prev = lf
        run_id[k] = rid
    n_boxes = np.array([i["gt_names"].shape[0] for i in infos], dtype=np.int64)
    return cfg, run_id, n_boxes


def make_embeddings(n, c=512, seed=0, scale=1.0):
    """Seeded stand-in for the swept ``[N,512]`` BEV embeddings (post-ReLU GAP => >= 0)."""
    rng = np.random.default_rng(seed)
    base = np.abs(rng.normal(0.0, 1.0, size=(n, c))).astype(np.float32)
    return (base * np.float32(scale)).astype(np.float32)


def seeded_init_(module, seed=0):
    """Fill every parameter/buffer of ``module`` from a per-tensor seeded generator.

    Independent of construction order and of the global RNG: tensor ``name`` uses seed
    ``seed + crc32(name)``.  Conv / linear weights ~ N(0, 1/fan_in), biases and BN means
    ~ N(0, 0.1), BN scales and variances ~ U(0.5, 1.5) so that BatchNorm is not the
    identity.  There are no checkpoints offline; bench.py and the
    parity fixtures both use this.
    """
    import zlib

    import torch
    with torch.no_grad():
        for name, t in sorted(module.state_dict().items()):
            if not t.dtype.is_floating_point:
                continue
            g = torch.Generator().manual_seed(int(seed) + zlib.crc32(name.encode()))
            if t.dim() >= 2:
                fan_in = int(np.prod(t.shape[1:])) if t.dim() > 2 else int(t.shape[1])
                if "deblocks" in name and t.dim() == 4 and t.shape[2] == 2:
                    fan_in = int(t.shape[0])          # ConvTranspose2d [Cin,Cout,2,2]
                if t.dim() == 5:                       # spconv layout [kz,ky,kx,Cin,Cout]
                    fan_in = int(np.prod(t.shape[:4]))
                v = torch.randn(t.shape, generator=g) / float(max(fan_in, 1)) ** 0.5
            elif name.endswith("running_var") or (name.endswith("weight") and t.dim() == 1):
                v = torch.rand(t.shape, generator=g) + 0.5
            else:
                v = torch.randn(t.shape, generator=g) * 0.1
            t.copy_(v.to(t.dtype))
    return module


def make_point_cloud(frame_index, nsweeps=10, beams=32, azimuths=1085,
                     n_boxes=30, max_range=54.0):
    """Synthetic 10-sweep lidar frame: ``[P,5] f32`` (x, y, z, intensity, dt).

    32-beam x 1085-azimuth ring scan hitting a ground plane at z=-1.84 or one
    of ``n_boxes`` random upright boxes, whichever is nearer; fixed sweep order;
    seed = frame index.  Points closer than 1 m are dropped like the
    reference's ``remove_close`` (det3d/datasets/pipelines/loading.py:33-41).
    """
    rng = np.random.default_rng(1000003 * 7 + int(frame_index))
    az = np.linspace(0.0, 2.0 * np.pi, azimuths, endpoint=False)
    el = np.deg2rad(np.linspace(-30.67, 10.67, beams))
    bx = rng.uniform(-45.0, 45.0, size=(n_boxes, 2))
    bsz = rng.uniform(0.6, 5.0, size=(n_boxes, 2))
    bh = rng.uniform(0.8, 3.2, size=n_boxes)
    sweeps = []
    for s in range(nsweeps):
        a = (az + rng.uniform(0, 2 * np.pi / azimuths))[None, :]
        e = el[:, None] + rng.normal(0.0, 2e-4, size=(beams, 1))
        dx, dy, dz = np.cos(e) * np.cos(a), np.cos(e) * np.sin(a), np.sin(e) * np.ones_like(a)
        # ground hit
        with np.errstate(divide="ignore", invalid="ignore"):
            t = np.where(dz < -1e-3, -1.84 / dz, np.inf)
        # box hits: slab test in xy, accept if z within box height
        for b in range(n_boxes):
            lo = bx[b] - 0.5 * bsz[b]
            hi = bx[b] + 0.5 * bsz[b]
            with np.errstate(divide="ignore", invalid="ignore"):
                tx1, tx2 = lo[0] / dx, hi[0] / dx
                ty1, ty2 = lo[1] / dy, hi[1] / dy
            tn = np.maximum(np.minimum(tx1, tx2), np.minimum(ty1, ty2))
            tf = np.minimum(np.maximum(tx1, tx2), np.maximum(ty1, ty2))
            zhit = tn * dz
            ok = (tn > 0) & (tn <= tf) & (zhit > -1.84) & (zhit < -1.84 + bh[b])
            t = np.where(ok & (tn < t), tn, t)
        t = t + rng.normal(0.0, 0.02, size=t.shape)
        keep = np.isfinite(t) & (t < max_range * 1.4) & (t > 0)
        x, y, z = (t * dx)[keep], (t * dy)[keep], (t * dz)[keep]
        # ego motion between sweeps: shift along +x by 0.2 m per 0.05 s
        x = x - 0.2 * s
        inten = rng.uniform(0.0, 255.0, size=x.shape)
        dt = np.full(x.shape, 0.05 * s)
        pts = np.stack([x, y, z, inten, dt], axis=1).astype(np.float32)
        close = (np.abs(pts[:, 0]) < 1.0) & (np.abs(pts[:, 1]) < 1.0)
        sweeps.append(pts[~close])
    return np.concatenate(sweeps, axis=0)


def seed_modules_(mod, seed):
    """Seeded He-style weights / plausible BN statistics for every Conv2d / BatchNorm2d / Linear / LayerNorm of ``mod``
    (the BEVFusion camera-branch modules have no checkpoint offline); returns ``mod.eval()``."""
    import torch
    from torch import nn
    g = torch.Generator().manual_seed(seed)
    for m in mod.modules():
        if isinstance(m, nn.Conv2d):
            fan = m.in_channels * m.kernel_size[0] * m.kernel_size[1]
            m.weight.data = torch.randn(m.weight.shape, generator=g) * (2.0 / fan) ** 0.5
            if m.bias is not None:
                m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.1
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data = torch.rand(m.weight.shape, generator=g) + 0.5
            m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.1
            m.running_mean.data = torch.randn(m.running_mean.shape, generator=g) * 0.1
            m.running_var.data = torch.rand(m.running_var.shape, generator=g) + 0.5
        elif isinstance(m, nn.Linear):
            m.weight.data = torch.randn(m.weight.shape, generator=g) * (1.0 / m.in_features) ** 0.5
            if m.bias is not None:
                m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.1
        elif isinstance(m, nn.LayerNorm):
            m.weight.data = torch.rand(m.weight.shape, generator=g) + 0.5
            m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.1
    for name, prm in mod.named_parameters():
        if name.endswith("relative_position_bias_table"):
            prm.data = torch.randn(prm.shape, generator=g) * 0.5
    return mod.eval()


def camera_setup(B, N, seed, image_size):
    """Plausible nuScenes-like camera matrices: N cameras looking around the ego vehicle."""
    import torch
    g = torch.Generator().manual_seed(seed)
    iH, iW = image_size
    K = torch.eye(4).repeat(B, N, 1, 1)
    K[..., 0, 0] = K[..., 1, 1] = 0.48 * iW
    K[..., 0, 2], K[..., 1, 2] = iW / 2.0, iH / 2.0
    cam2lidar = torch.eye(4).repeat(B, N, 1, 1)
    for n in range(N):
        yaw = 2 * np.pi * n / N
        # camera axes (x right, y down, z forward) expressed in the lidar frame
        fwd = torch.tensor([np.cos(yaw), np.sin(yaw), 0.0])
        right = torch.tensor([np.sin(yaw), -np.cos(yaw), 0.0])
        down = torch.tensor([0.0, 0.0, -1.0])
        cam2lidar[:, n, :3, :3] = torch.stack([right, down, fwd], 1).float()
        cam2lidar[:, n, :3, 3] = torch.tensor([0.5 * np.cos(yaw), 0.5 * np.sin(yaw), 1.5]).float()
    lidar2cam = torch.inverse(cam2lidar)
    lidar2image = K.matmul(lidar2cam)
    img_aug = torch.eye(4).repeat(B, N, 1, 1)
    img_aug[..., 0, 0] = img_aug[..., 1, 1] = 0.9
    img_aug[..., 0, 3], img_aug[..., 1, 3] = 3.0, -2.0
    lidar_aug = torch.eye(4).repeat(B, 1, 1)
    ang = 0.05
    lidar_aug[:, 0, 0], lidar_aug[:, 0, 1], lidar_aug[:, 1, 0], lidar_aug[:, 1, 1] = np.cos(ang), -np.sin(ang), np.sin(ang), np.cos(ang)
    lidar_aug[:, :3, 3] = torch.tensor([0.3, -0.2, 0.05])
    points = [torch.cat([(torch.rand(4000, 2, generator=g) - 0.5) * 90.0, torch.rand(4000, 1, generator=g) * 4.0 - 2.0,
                         torch.rand(4000, 2, generator=g)], 1) for _ in range(B)]
    return K, cam2lidar, lidar2image, img_aug, lidar_aug, points
