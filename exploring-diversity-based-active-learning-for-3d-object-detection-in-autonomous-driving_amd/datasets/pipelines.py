"""Pipeline plug-ins (reference ``det3d/datasets/registry.py`` PIPELINES and
``det3d/datasets/pipelines/{loading,preprocess,formating}.py``), device-backed.

Same registry name, constructor kwargs and ``__call__(res, info) -> (res, info)`` contract as the
reference stages of the *val* pipeline; the arithmetic runs in libal3d_hip.so and the arrays they
put into ``res`` are device tensors:

* ``LoadPointCloudFromFile(dataset="NuScenesDataset")`` -- ``res["lidar"]["points" | "times" |
  "combined"]`` from the key frame + ``nsweeps-1`` sweeps (loading.py:73-126) via
  ``al3d_merge_sweeps_f32``; the sweep order comes from ``np.random.choice`` like the reference.
* ``Voxelization(cfg=dict(range, voxel_size, max_points_in_voxel, max_voxel_num))`` --
  ``res["lidar"]["voxels"] = dict(voxels, coordinates, num_points, num_voxels, shape)``
  (preprocess.py:259-304) via ``al3d_voxelize_mean_f32``; additionally ``voxel_features`` (the mean
  VFE the detector's reader would compute, voxel_encoder.py:206-211).
* ``AssignTarget(cfg=...)`` (val branch) -- the constant anchors per task
  (preprocess.py:346-378,426-431), generated once and cached on the device.
* ``Reformat`` -- the ``example`` dict the detector reads (formating.py).
``collate_device`` is the device analogue of ``collate_kitti`` (collate.py:90-150).
"""
import numpy as np
import torch

from ..utils import Registry, build_from_cfg
from .anchors import generate_task_anchors

PIPELINES = Registry("pipeline")


def _get(cfg, key):
    return cfg[key] if isinstance(cfg, dict) else getattr(cfg, key)


class Compose:
    def __init__(self, transforms):
        self.transforms = [build_from_cfg(t, PIPELINES) if isinstance(t, dict) else t for t in transforms]

    def __call__(self, res, info):
        for t in self.transforms:
            res, info = t(res, info)
            if res is None:
                return None
        return res, info


@PIPELINES.register_module
class LoadPointCloudFromFile:
    def __init__(self, dataset="KittiDataset", device="cuda", root=None, **kwargs):
        self.type = dataset
        self.device = device
        self.root = root

    def __call__(self, res, info):
        res["type"] = self.type
        if self.type != "NuScenesDataset":
            raise NotImplementedError("only the NuScenesDataset branch is on the sweep path")
        from .nusc_files import load_frame_points_device
        nsweeps = res["lidar"]["nsweeps"]
        combined = load_frame_points_device(info, self.device, nsweeps=nsweeps, root=self.root, rng=np.random)
        res["lidar"]["points"] = combined[:, :4]
        res["lidar"]["times"] = combined[:, 4:5]
        res["lidar"]["combined"] = combined
        return res, info


@PIPELINES.register_module
class LoadPointCloudAnnotations:
    """Val-mode pass-through: the sweep never reads annotations (loading.py:163-199 fills
    ``res["lidar"]["annotations"]`` for training); kept so reference pipelines load unchanged."""

    def __init__(self, with_bbox=True, **kwargs):
        pass

    def __call__(self, res, info):
        return res, info


@PIPELINES.register_module
class Preprocess:
    """Val branch of preprocess.py:33-257: NuScenes points become the 5-column ``combined`` cloud;
    no shuffling, no filtering, no augmentation."""

    def __init__(self, cfg=None, **kwargs):
        self.mode = _get(cfg, "mode") if cfg is not None else "val"
        if self.mode == "train":
            raise NotImplementedError("al3d implements the inference sweep, not training")

    def __call__(self, res, info):
        res["mode"] = self.mode
        if res.get("type") in ("NuScenesDataset", "LyftDataset") and res["lidar"].get("combined") is not None:
            res["lidar"]["points"] = res["lidar"]["combined"]
        return res, info


@PIPELINES.register_module
class Voxelization:
    def __init__(self, **kwargs):
        cfg = kwargs.get("cfg", None)
        self.range = _get(cfg, "range")
        self.voxel_size = _get(cfg, "voxel_size")
        self.max_points_in_voxel = _get(cfg, "max_points_in_voxel")
        self.max_voxel_num = _get(cfg, "max_voxel_num")
        self.device = kwargs.get("device", "cuda")
        self._vox = None

    def _voxelizer(self, device):
        from ..detector_ops import Voxelizer
        if self._vox is None or self._vox.device != torch.device(device):
            self._vox = Voxelizer(self.range, self.voxel_size, self.max_points_in_voxel, self.max_voxel_num,
                                  max_batch=1, device=device)
        return self._vox

    def __call__(self, res, info):
        if res.get("mode", "val") == "train":
            raise NotImplementedError("al3d implements the inference sweep, not training")
        pts = res["lidar"].get("combined")
        if pts is None:
            pts = res["lidar"]["points"]
        if not isinstance(pts, torch.Tensor):
            pts = torch.as_tensor(np.ascontiguousarray(pts, dtype=np.float32))
        pts = pts.to(self.device).contiguous()
        vox = self._voxelizer(pts.device)
        off = torch.tensor([0, pts.shape[0]], dtype=torch.int64, device=pts.device)
        v = vox(pts, off, want_voxels=True)
        res["lidar"]["voxels"] = dict(
            voxels=v["voxels"],
            coordinates=v["coords"][:, 1:],                 # (z, y, x) like the reference
            num_points=v["num_points"].to(torch.int64),
            num_voxels=np.array([v["feat"].shape[0]], dtype=np.int64),
            shape=vox.grid_size,
            voxel_features=v["feat"],
        )
        return res, info


@PIPELINES.register_module
class AssignTarget:
    def __init__(self, **kwargs):
        cfg = kwargs["cfg"]
        ta = _get(cfg, "target_assigner")
        self.tasks = _get(ta, "tasks")
        self.anchor_generators = _get(ta, "anchor_generators")
        self.out_size_factor = _get(cfg, "out_size_factor")
        self.device = kwargs.get("device", "cuda")
        self._cache = {}

    def __call__(self, res, info):
        if res.get("mode", "val") == "train":
            raise NotImplementedError("al3d implements the inference sweep, not training")
        grid = np.asarray(res["lidar"]["voxels"]["shape"])
        fm = [*(grid[:2] // self.out_size_factor), 1][::-1]          # [1, H, W]
        key = tuple(int(v) for v in fm)
        if key not in self._cache:
            self._cache[key] = [torch.as_tensor(a, dtype=torch.float32, device=self.device)
                                for a in generate_task_anchors(self.tasks, self.anchor_generators, list(key))]
        res["lidar"]["targets"] = dict(anchors=self._cache[key])
        return res, info


@PIPELINES.register_module
class Reformat:
    def __init__(self, **kwargs):
        pass

    def __call__(self, res, info):
        v = res["lidar"]["voxels"]
        example = dict(metadata=res.get("metadata", {}), points=res["lidar"].get("points"),
                       voxels=v["voxels"], shape=v["shape"], num_points=v["num_points"],
                       num_voxels=v["num_voxels"], coordinates=v["coordinates"],
                       voxel_features=v.get("voxel_features"))
        if "targets" in res["lidar"]:
            example["anchors"] = res["lidar"]["targets"]["anchors"]
        return example, info


class SweepDataset:
    """Minimal val-mode dataset (reference NuScenesDataset.get_sensor_data,
    det3d/datasets/nuscenes/nuscenes.py:139-170): ``dataset[i]`` runs the pipeline on ``infos[i]``."""

    def __init__(self, infos, pipeline, nsweeps=10, class_names=None, **kwargs):
        self.infos = infos
        self.nsweeps = nsweeps
        self.class_names = class_names
        self.pipeline = pipeline if isinstance(pipeline, Compose) else Compose(pipeline)

    def __len__(self):
        return len(self.infos)

    def __getitem__(self, idx):
        info = self.infos[idx]
        res = {"lidar": {"type": "lidar", "points": None, "nsweeps": self.nsweeps, "annotations": None},
               "metadata": {"image_prefix": None, "num_point_features": 5, "token": info.get("token")},
               "calib": None, "cam": {}, "mode": "val"}
        data, _ = self.pipeline(res, info)
        return data


def collate_device(batch_list):
    """Device analogue of ``collate_kitti``: concatenate per-sample voxels and prepend the batch
    index to ``coordinates`` (``[sum M, 4]`` i32 ``(b, z, y, x)``); anchors are shared constants."""
    out = {}
    coords = []
    for b, ex in enumerate(batch_list):
        c = ex["coordinates"].to(torch.int32)
        coords.append(torch.cat([torch.full((c.shape[0], 1), b, dtype=torch.int32, device=c.device), c], dim=1))
    out["coordinates"] = torch.cat(coords, dim=0).contiguous()
    for k in ("voxels", "num_points", "voxel_features"):
        if batch_list[0].get(k) is not None:
            out[k] = torch.cat([ex[k] for ex in batch_list], dim=0)
    out["num_voxels"] = np.concatenate([np.asarray(ex["num_voxels"]) for ex in batch_list])
    out["voxel_cap"] = int(out["num_voxels"].max()) if len(out["num_voxels"]) else 0   # frames are concatenated in order
    out["shape"] = np.stack([np.asarray(ex["shape"]) for ex in batch_list])
    out["metadata"] = [ex.get("metadata") for ex in batch_list]
    if "anchors" in batch_list[0]:
        out["anchors"] = batch_list[0]["anchors"]
    return out
