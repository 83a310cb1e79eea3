"""Embedding-based selectors that own the unlabeled-pool sweep.

Reference: det3d/selectors/feature_selector.py, spatial_feature_selector.py,
spatial_temporal_feature_selector.py.  ``buffer_pred`` runs the detector over
every frame (``detector(example, return_loss=False, estimate=True)``) and keeps
the global-average-pooled neck output as a ``[N,512]`` float32 embedding; the
pairwise L1 map, normalise/aggregate and greedy loop run in HIP kernels.
"""
import logging
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from .base_selector import BaseSelector, _rank, save_npy_atomic
from .registry import SELECTORS

_DEFAULT_LOGS = "/home/st2000/data/Datasets/nuScenes/train/v1.0-trainval/log.json"
_DEFAULT_DIJKSTRA = "/home/st2000/data/buffers/dijkstra_distance_map.npy"


class _SweepMixin:
    """The sweep loop the reference copy-pastes into seven selectors
    (e.g. feature_selector.py:51-85)."""

    def buffer_pred(self, **kwargs) -> torch.Tensor:
        from ..sweep import sweep_embeddings
        self.logger.info(
            f"begin predict all results of samples and save them as {self.buffer_path}")
        device = self._device(kwargs)
        if getattr(self, "uses_spatial_map", False):
            self.prefetch_spatial_map(device)        # overlaps the sweep (needs poses only)
        prediction = sweep_embeddings(self.detector, self.dataloader, device,
                                      num_frames=len(self.infos_origin))
        if self.buffer_path:
            from .base_selector import _rank
            if _rank() == 0:
                torch.save(prediction.cpu(), self.buffer_path)
        self.detector = None
        return prediction

    def _features(self, device, kwargs):
        if self.pred:
            feats = self.buffer_pred(**kwargs)
            self.logger.info(f"all prediction results have been saved in {self.buffer_path}")
        else:
            feats = torch.load(self.buffer_path, weights_only=True)
            self.logger.info(f"all prediction results have been load from {self.buffer_path}")
        return feats.to(device=device, dtype=torch.float32)


@SELECTORS.register_module
class FeatureSelector(_SweepMixin, BaseSelector):
    """feature_selector.py:16-172: float32 L1 map, greedy in float32, needs a
    non-empty buffer, output order ``selected + sampled``."""

    def __init__(
            self,
            budget: int,
            buffer_file: str,
            dump_file_name: Optional[str] = None,
            infos_origin: List[Dict] = [],
            buffer_path: str = "/home/st2000/data/buffers/feature_pred.pt",
            p: int = 2,
            detector: Optional[torch.nn.Module] = None,
            dataloader=None,
            logger: Optional[logging.Logger] = None,
            distance_store_file: str = "/home/st2000/data/buffers/feature_distance_map.npy",
            pred: bool = True,
            cost_b: float = 0.04,
            cost_f: float = 0.12,
    ) -> None:
        super().__init__(budget, buffer_file, dump_file_name, infos_origin=infos_origin,
                         detector=detector, dataloader=dataloader, logger=logger, pred=pred,
                         cost_b=cost_b, cost_f=cost_f)
        self.buffer_path = buffer_path
        self.distance_store_file = distance_store_file
        assert p in [1, 2]
        self.p = p

    def get_feature_distance_map(self, feats: torch.Tensor) -> torch.Tensor:
        from .. import selector_ops as ops
        if self.distance_store_file and os.path.exists(self.distance_store_file):
            return torch.from_numpy(np.load(self.distance_store_file)).to(feats.device)
        distance_map = ops.l1_distance(feats, self.p)
        if self.distance_store_file and _rank() == 0:
            save_npy_atomic(self.distance_store_file, distance_map.cpu().numpy())
        return distance_map

    def select_samples(self, **kwargs) -> None:
        device = self._device(kwargs)
        feats = self._features(device, kwargs)
        distance_map = self.get_feature_distance_map(feats)
        self._greedy(distance_map, device, order="selected+sampled")


@SELECTORS.register_module
class SpatialTemporalFeatureSelector(_SweepMixin, BaseSelector):
    """spatial_temporal_feature_selector.py:17-258:
    D = S' + lambda_t T' + lambda_f F'  with X' = 1 - exp(-X), F in float32."""
    uses_spatial_map = True

    def __init__(
            self,
            budget: int,
            buffer_file: str,
            dump_file_name: Optional[str] = None,
            infos_origin: List[Dict] = [],
            buffer_path: str = "/home/st2000/data/buffers/feature_pred.pt",
            detector: Optional[torch.nn.Module] = None,
            dataloader=None,
            logger: Optional[logging.Logger] = None,
            pred: bool = True,
            k: int = 8,
            p: int = 2,
            logs_file: str = _DEFAULT_LOGS,
            distance_store_file: str = _DEFAULT_DIJKSTRA,
            cost_b: float = 0.04,
            cost_f: float = 0.12,
            lambda_f: float = 1.0,
            lambda_t: float = 1.0,
    ) -> None:
        super().__init__(budget, buffer_file, dump_file_name, infos_origin=infos_origin,
                         detector=detector, dataloader=dataloader, logger=logger, pred=pred,
                         cost_b=cost_b, cost_f=cost_f)
        self.buffer_path = buffer_path
        self.logs_file = logs_file
        self.k = k
        self.distance_store_file = distance_store_file
        assert p in [1, 2]
        self.p = p
        self.lambda_f = lambda_f
        self.lambda_t = lambda_t
        self.logger.info(f"lambda_f: {self.lambda_f}")
        self.logger.info(f"lambda_t: {self.lambda_t}")

    def get_feature_distance_map(self, feats: torch.Tensor) -> torch.Tensor:
        from .. import selector_ops as ops
        return ops.l1_distance(feats, self.p)

    def select_samples(self, **kwargs) -> None:
        from .. import selector_ops as ops
        device = self._device(kwargs)
        n = len(self.infos_origin)
        feats = self._features(device, kwargs)
        feature_map = self.get_feature_distance_map(feats)
        spatial = self._spatial_map(device, self.k, self.distance_store_file, self.logs_file)
        run_id = torch.from_numpy(self._run_ids()).to(device)
        distance_map = ops.combine_maps(
            n, spatial=spatial, temporal_id=run_id, feat=feature_map, normalize="exp",
            aggregate="sum", lambda_t=float(self.lambda_t), lambda_f=float(self.lambda_f))
        del spatial, feature_map
        self._greedy(distance_map, device)


@SELECTORS.register_module
class SpatialFeatureSelector(_SweepMixin, BaseSelector):
    """spatial_feature_selector.py:18-234: fps seeded from the normalised *spatial*
    map, iterated on the aggregate of spatial and feature maps (A.1 quirk 9)."""
    uses_spatial_map = True

    def __init__(
            self,
            budget: int,
            buffer_file: str,
            dump_file_name: Optional[str] = None,
            infos_origin: List[Dict] = [],
            buffer_path: str = "",
            detector: Optional[torch.nn.Module] = None,
            dataloader=None,
            logger: Optional[logging.Logger] = None,
            pred: bool = False,
            k: int = 8,
            p: int = 2,
            logs_file: str = "/home/st2000/data/Datasets/nuScenes/train/train/v1.0-trainval/log.json",
            distance_store_file: str = _DEFAULT_DIJKSTRA,
            cost_b: float = 0.04,
            cost_f: float = 0.12,
            aggregate: str = "sum",
    ) -> None:
        super().__init__(budget, buffer_file, dump_file_name, infos_origin=infos_origin,
                         detector=detector, dataloader=dataloader, logger=logger, pred=pred,
                         cost_b=cost_b, cost_f=cost_f)
        self.logs_file = logs_file
        self.k = k
        self.distance_store_file = distance_store_file
        self.buffer_path = buffer_path
        assert p in [1, 2]
        self.p = p
        assert aggregate in ["sum", "min", "max"]
        self.aggregate = aggregate

    def select_samples(self, **kwargs) -> None:
        from .. import selector_ops as ops
        device = self._device(kwargs)
        n = len(self.infos_origin)
        feats = self._features(device, kwargs)
        feature_map = ops.l1_distance(feats, self.p)
        spatial = self._spatial_map(device, self.k, self.distance_store_file, self.logs_file)
        seed_map = ops.combine_maps(n, spatial=spatial, normalize="exp", aggregate="sum")
        distance_map = ops.combine_maps(n, spatial=spatial, feat=feature_map, normalize="exp",
                                        aggregate=self.aggregate, lambda_f=1.0)
        del spatial, feature_map
        self._greedy(distance_map, device, seed_map=seed_map, check_seeded=True)
