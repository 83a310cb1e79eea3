"""Selector base class + the device greedy driver shared by every map selector.

Interface mirrors the reference's ``BaseSelector``
(det3d/selectors/base_selector.py:12-86): same constructor arguments, the
``selected_index`` dict keyed by the cumulative budget string, ``dump_file()``
writing the buffer json and ``infos_<cum>.pkl`` on rank 0 only.
"""
import functools
import logging
import os
import random
from typing import Dict, List, Optional

import numpy as np
import torch

from ..utils.fileio import dump, load
from .registry import SELECTORS


def _rank():
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return torch.distributed.get_rank()
    return 0


def master_only(func):
    """Run on rank 0 only (reference det3d/torchie/trainer/utils.py:36-43)."""
    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        if _rank() == 0:
            return func(*args, **kwargs)
    return wrapper


def save_npy_atomic(path, array) -> None:
    """``np.save`` through a temporary file + ``os.replace``: a reader (the next round's
    ``os.path.exists`` + ``np.load``) never sees a torn map.  Same name rule as ``np.save``
    (``.npy`` appended when missing)."""
    import os
    import numpy as np
    if not str(path).endswith(".npy"):
        path = str(path) + ".npy"
    tmp = f"{path}.tmp{os.getpid()}"
    with open(tmp, "wb") as f:
        np.save(f, array)
    os.replace(tmp, path)


def logfile_of(info) -> str:
    """``cam_front_path`` basename up to the first ``__`` (spatial_temporal_selector.py:80)."""
    return info["cam_front_path"].split("/")[-1].split("__")[0]


@SELECTORS.register_module
class BaseSelector(object):
    def __init__(
            self,
            budget: int,
            buffer_file: str,
            dump_file_name: Optional[str] = None,
            infos_origin: List[Dict] = [],
            detector: Optional[torch.nn.Module] = None,
            dataloader=None,
            logger: Optional[logging.Logger] = None,
            pred: bool = False,
            cost_b: float = 0.04,
            cost_f: float = 0.12,
    ) -> None:
        super().__init__()
        self.budget = budget
        self.buffer_file = buffer_file
        self.dump_file_name = buffer_file if dump_file_name is None else dump_file_name
        self.buffer = load(buffer_file)
        self.detector = detector
        self.dataloader = dataloader
        self.selected_index = {}
        self.infos_file = infos_origin
        self.infos_origin = load(infos_origin)
        self.current_budget = str(self.budget + int(self.get_max_key()))
        self.logger = logger if logger is not None else logging.getLogger(__file__)
        self.pred = pred
        self.cost_b = cost_b
        self.cost_f = cost_f

    # ------------------------------------------------------------ reference API
    def get_max_key(self):
        return str(max(int(key) for key in self.buffer.keys()))

    def select_samples(self, **kwargs) -> None:
        return

    @master_only
    def dump_file(self) -> None:
        self.buffer.update(self.selected_index)
        dump(self.buffer, self.dump_file_name)
        self.logger.info(f"update the buffer, and save as {self.dump_file_name}")
        ext = os.path.splitext(self.infos_file)[-1]
        replace_path = self.infos_file.replace(ext, f"_{self.current_budget}{ext}")
        infos_sampled = [self.infos_origin[i] for i in self.buffer[str(self.current_budget)]]
        dump(infos_sampled, replace_path)
        self.logger.info(f"sample the {self.current_budget} infos and save as {replace_path}")

    def get_selected_samples(self):
        return self.selected_index

    def get_cost_amount(self):
        cost = 0
        sampled_frames = [self.infos_origin[i] for i in self.buffer[self.get_max_key()]]
        cost += self.cost_f * len(sampled_frames)
        for anno in sampled_frames:
            cost += anno["gt_names"].shape[0] * self.cost_b
        return cost

    # ------------------------------------------------------------ device helpers
    def _device(self, kwargs) -> torch.device:
        """Device the selection runs on: ``local_rank`` kwarg (as tools/active_select.py:162
        passes it) or the current device.  A GPU is required -- there is no CPU path."""
        if not torch.cuda.is_available():
            raise RuntimeError("al3d selectors need a ROCm device; no CPU fallback exists")
        if "local_rank" in kwargs and kwargs["local_rank"] is not None:
            return torch.device("cuda", int(kwargs["local_rank"]))
        return torch.device("cuda", torch.cuda.current_device())

    def _greedy(self, distance_map, device, *, seed_map=None, check_seeded=False,
                order="sampled+selected"):
        """Greedy k-center under the cost budget on a device-resident map
        (spatial_temporal_selector.py:157-193 / feature_selector.py:142-172)."""
        from .. import selector_ops as ops
        n = len(self.infos_origin)
        sampled = list(self.buffer[self.get_max_key()])
        if len(sampled) > 0:
            first = -1
        else:
            if order == "selected+sampled":
                # feature_selector.py:143-148: torch.stack([]) on an empty buffer
                raise RuntimeError("stack expects a non-empty TensorList")
            first = random.choice(range(n))
        start_cost = float(self.get_cost_amount())
        box_cost = torch.tensor(
            [info["gt_names"].shape[0] * self.cost_b for info in self.infos_origin],
            dtype=torch.float64, device=device)
        status, picks = ops.greedy_kcenter(
            distance_map, sampled, first, box_cost, self.cost_f, start_cost,
            float(int(self.current_budget)), seed_map=seed_map, check_seeded=check_seeded)
        if status == -1:
            # the reference's duplicate-pick assert (spatial_temporal_selector.py:182)
            raise AssertionError("id has been selected")
        if status != 0:
            raise RuntimeError(f"greedy k-center failed with status {status}")
        self.logger.info(f"selected {len(picks)} new frames")
        if order == "sampled+selected":
            self.selected_index[self.current_budget] = sampled + picks
        else:
            self.selected_index[self.current_budget] = picks + sampled
        return picks

    # ------------------------------------------------------------ metadata (host)
    def _ego_xy(self) -> np.ndarray:
        """Ego XY per frame, the reference's own numpy expression
        (spatial_temporal_selector.py:83-89); metadata parsing stays on the host."""
        locs = []
        for info in self.infos_origin:
            cal = info["car_from_global"]
            location = -(cal[:3, 3].T @ cal[:3, :3])
            locs.append(location[:2])
        return np.stack(locs)

    def _run_ids(self) -> np.ndarray:
        """Consecutive-run id per frame (spatial_temporal_selector.py:112-129)."""
        ids, prev, rid = [], None, -1
        for info in self.infos_origin:
            lf = logfile_of(info)
            if lf != prev:
                rid += 1
                prev = lf
            ids.append(rid)
        return np.asarray(ids, dtype=np.int64)

    def _max_temporal_distance(self) -> int:
        """Longest run, ignoring the last one -- the reference only updates it on a
        logfile change (spatial_temporal_selector.py:117-129)."""
        best, count, prev = 0, 0, logfile_of(self.infos_origin[0])
        for info in self.infos_origin:
            lf = logfile_of(info)
            if lf == prev:
                count += 1
            else:
                prev = lf
                if count > best:
                    best = count
                count = 1
        return best

    def prefetch_spatial_map(self, device):
        """Start the geodesic map on a side stream.  It depends only on the pool's poses, so the
        sweep that produces the embeddings can run at the same time; ``_spatial_map`` then just
        waits for the event.  Under torch.distributed the row blocks are all-gathered on that
        stream too (every rank must call this -- the sweep mixin does)."""
        device = torch.device(device)
        if device.type != "cuda" or getattr(self, "_spatial_prefetch", None) is not None:
            return
        if not (hasattr(self, "k") and hasattr(self, "distance_store_file")):
            return
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            spatial = self._spatial_map(device, self.k, self.distance_store_file,
                                        getattr(self, "logs_file", None))
            ev = torch.cuda.Event()
            ev.record(side)
        self._spatial_prefetch = (spatial, ev)

    def _spatial_map(self, device, k, distance_store_file, logs_file=None):
        """kNN-graph geodesic map on device; cached as ``.npy`` like the reference
        (spatial_temporal_selector.py:60-63,106)."""
        from .. import selector_ops as ops
        pre = getattr(self, "_spatial_prefetch", None)
        if pre is not None:                      # started on a side stream by prefetch_spatial_map()
            spatial, ev = pre
            self._spatial_prefetch = None
            cur = torch.cuda.current_stream(spatial.device)
            cur.wait_event(ev)
            spatial.record_stream(cur)
            return spatial
        if distance_store_file and os.path.exists(distance_store_file):
            self.logger.info(f"begin to load the distance map from {distance_store_file}")
            return torch.from_numpy(np.load(distance_store_file)).to(device)
        if logs_file is not None:
            # the reference resolves every frame's logfile against log.json and raises
            # KeyError on a miss (spatial_temporal_selector.py:69-81); keep that check.
            log_to_loc = {l["logfile"]: l["location"].split("-")[-1] for l in load(logs_file)}
            for info in self.infos_origin:
                log_to_loc[logfile_of(info)]
        xy = torch.from_numpy(np.ascontiguousarray(self._ego_xy(), dtype=np.float64)).to(device)
        spatial = ops.spatial_map(xy, k)
        if distance_store_file and _rank() == 0:
            save_npy_atomic(distance_store_file, spatial.cpu().numpy())
            self.logger.info(f"save the distance map as {distance_store_file}")
        return spatial
