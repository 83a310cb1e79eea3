"""Metadata-only diversity selectors: spatial / temporal / Euclidean / spatial+temporal.

Registry names, constructor kwargs and output ordering follow the reference
(det3d/selectors/{spatial,temporal,euclidean_spatial,spatial_temporal}_selector.py);
the O(N^2) maps and the greedy loop run in HIP kernels.
"""
import logging
from typing import Dict, List, Optional

import numpy as np
import torch

from ..utils.fileio import load
from .base_selector import BaseSelector, _rank, logfile_of, save_npy_atomic
from .registry import SELECTORS

_DEFAULT_LOGS = "/home/st2000/data/Datasets/nuScenes/train/v1.0-trainval/log.json"
_DEFAULT_DIJKSTRA = "/home/st2000/data/buffers/dijkstra_distance_map.npy"


@SELECTORS.register_module
class SpatialTemporalSelector(BaseSelector):
    """spatial_temporal_selector.py:16-193."""

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
            k: int = 8,
            logs_file: str = _DEFAULT_LOGS,
            normalize: str = "exp",
            distance_store_file: str = _DEFAULT_DIJKSTRA,
            cost_b: float = 0.04,
            cost_f: float = 0.12,
            lambda_t: float = 1,
            aggregate: str = "sum",
    ) -> None:
        super().__init__(budget, buffer_file, dump_file_name, infos_origin=infos_origin,
                         detector=detector, dataloader=dataloader, logger=logger, pred=pred,
                         cost_b=cost_b, cost_f=cost_f)
        self.logs_file = logs_file
        assert normalize in ["linear", "exp"]
        self.normalize = normalize
        self.k = k
        self.lambda_t = lambda_t
        self.distance_store_file = distance_store_file
        assert aggregate in ["sum", "min", "max"]
        self.aggregate = aggregate
        self.logger.info(f"lambda_t: {self.lambda_t}")

    def select_samples(self, **kwargs) -> None:
        from .. import selector_ops as ops
        device = self._device(kwargs)
        n = len(self.infos_origin)
        spatial = self._spatial_map(device, self.k, self.distance_store_file, self.logs_file)
        run_id = torch.from_numpy(self._run_ids()).to(device)
        sscale = tscale = 1.0
        if self.normalize == "linear":
            sscale = ops.max_finite(spatial)
            tscale = float(self._max_temporal_distance())
        distance_map = ops.combine_maps(
            n, spatial=spatial, temporal_id=run_id, normalize=self.normalize,
            aggregate=self.aggregate, lambda_t=float(self.lambda_t),
            spatial_scale=sscale, temporal_scale=tscale)
        del spatial
        self._greedy(distance_map, device)


@SELECTORS.register_module
class SpatialSelector(BaseSelector):
    """spatial_selector.py:15-138 (raw geodesic map, also asserts against the buffer)."""

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
            k: int = 8,
            logs_file: str = _DEFAULT_LOGS,
            distance_store_file: str = _DEFAULT_DIJKSTRA,
            cost_b: float = 0.04,
            cost_f: float = 0.12,
    ) -> None:
        super().__init__(budget, buffer_file, dump_file_name, infos_origin=infos_origin,
                         detector=detector, dataloader=dataloader, logger=logger, pred=pred,
                         cost_b=cost_b, cost_f=cost_f)
        self.logs_file = logs_file
        self.k = k
        self.distance_store_file = distance_store_file

    def select_samples(self, **kwargs) -> None:
        device = self._device(kwargs)
        distance_map = self._spatial_map(device, self.k, self.distance_store_file, self.logs_file)
        self._greedy(distance_map, device, check_seeded=True)


@SELECTORS.register_module
class TemporalSelector(BaseSelector):
    """temporal_selector.py:15-104: |i-j| within a logfile *name* group, 1e6 across."""

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
        super().__init__(budget, buffer_file, dump_file_name, infos_origin=infos_origin,
                         detector=detector, dataloader=dataloader, logger=logger, pred=pred,
                         cost_b=cost_b, cost_f=cost_f)

    def select_samples(self, **kwargs) -> None:
        from .. import selector_ops as ops
        device = self._device(kwargs)
        seen = {}
        gid = np.asarray([seen.setdefault(logfile_of(i), len(seen)) for i in self.infos_origin],
                         dtype=np.int64)
        distance_map = ops.combine_maps(len(gid), temporal_id=torch.from_numpy(gid).to(device),
                                        normalize="none", aggregate="sum", lambda_t=1.0)
        self._greedy(distance_map, device, check_seeded=True)


@SELECTORS.register_module
class EuSpatialSelector(BaseSelector):
    """euclidean_spatial_selector.py:15-143: in-city Euclidean distance, 1e6 across cities."""

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
            logs_file: str = _DEFAULT_LOGS,
            distance_store_file: str = "/home/st2000/data/buffers/euclidean_distance_map.npy",
            cost_b: float = 0.04,
            cost_f: float = 0.12,
    ) -> None:
        super().__init__(budget, buffer_file, dump_file_name, infos_origin=infos_origin,
                         detector=detector, dataloader=dataloader, logger=logger, pred=pred,
                         cost_b=cost_b, cost_f=cost_f)
        self.logs_file = logs_file
        self.distance_store_file = distance_store_file

    def select_samples(self, **kwargs) -> None:
        import os
        from .. import selector_ops as ops
        device = self._device(kwargs)
        if self.distance_store_file and os.path.exists(self.distance_store_file):
            distance_map = torch.from_numpy(np.load(self.distance_store_file)).to(device)
        else:
            log_to_loc = {l["logfile"]: l["location"].split("-")[-1] for l in load(self.logs_file)}
            seen = {}
            loc = np.asarray([seen.setdefault(log_to_loc[logfile_of(i)], len(seen))
                              for i in self.infos_origin], dtype=np.int64)
            xy = torch.from_numpy(np.ascontiguousarray(self._ego_xy(), dtype=np.float64)).to(device)
            distance_map = ops.euclid_map(xy, torch.from_numpy(loc).to(device))
            if self.distance_store_file and _rank() == 0:
                save_npy_atomic(self.distance_store_file, distance_map.cpu().numpy())
        self._greedy(distance_map, device, check_seeded=True)
