"""Uncertainty-driven selectors: Entropy, BADGE-style and UWE.

Reference: det3d/selectors/entropy_selector.py, badge_selector.py, uwe_selector.py.  They reuse
the sweep; the extra arithmetic (frame entropy, weighting, argsort) runs in
csrc/uncertainty_kernels.hip, the L1 map + greedy in the feature-selector kernels.
"""
import logging
import os
import random
from typing import Dict, List, Optional

import numpy as np
import torch

from .base_selector import BaseSelector, _rank, save_npy_atomic
from .registry import SELECTORS


@SELECTORS.register_module
class EntropySelector(BaseSelector):
    """Rank the unlabeled frames by mean box entropy and take them in that order under the cost
    budget (entropy_selector.py:14-147).  Output order ``selected + sampled``."""

    def __init__(
            self,
            budget: int,
            buffer_file: str,
            dump_file_name: Optional[str] = None,
            infos_origin: List[Dict] = [],
            buffer_path: str = "/home/st2000/data/buffers/entropy_pred.pt",
            p: int = 2,
            detector: Optional[torch.nn.Module] = None,
            dataloader=None,
            logger: Optional[logging.Logger] = None,
            pred: bool = True,
            random_sample: bool = False,
            sample_num: int = 6000,
            cost_b: float = 0.04,
            cost_f: float = 0.12,
    ) -> None:
        super().__init__(budget, buffer_file, dump_file_name, infos_origin=infos_origin,
                         detector=detector, dataloader=dataloader, logger=logger, pred=pred,
                         cost_b=cost_b, cost_f=cost_f)
        self.buffer_path = buffer_path
        assert p in [1, 2]
        self.p = p
        self.random_sample = random_sample
        self.sample_num = sample_num

    def buffer_pred(self, **kwargs) -> torch.Tensor:
        from ..sweep import sweep_embeddings
        device = self._device(kwargs)
        _, entropy = sweep_embeddings(self.detector, self.dataloader, device,
                                      num_frames=len(self.infos_origin), with_entropy=True)
        if self.buffer_path and _rank() == 0:
            torch.save(entropy.cpu(), self.buffer_path)
        self.detector = None
        return entropy

    def select_samples(self, **kwargs) -> None:
        from .. import selector_ops as ops
        device = self._device(kwargs)
        sampled = list(self.buffer[self.get_max_key()])
        left = list(range(len(self.infos_origin)))
        for x in sampled:
            left.remove(x)
        if self.pred:
            entropy = self.buffer_pred(**kwargs)
        else:
            entropy = torch.load(self.buffer_path, weights_only=True)
        if self.random_sample:
            assert self.sample_num > 0
            left = random.sample(left, self.sample_num)
        entropy = entropy.to(device=device, dtype=torch.float32)[torch.as_tensor(left, device=device)]
        order = ops.argsort_desc(entropy.contiguous()).tolist()
        picks = [left[order[0]]]
        cost = self.get_cost_amount()
        cost += self.cost_f
        # the reference charges the first pick with infos_origin[sorted position], not the frame
        # it picked (entropy_selector.py:132, quirk A.1 #7)
        cost += self.infos_origin[order[0]]["gt_names"].shape[0] * self.cost_b
        sort_id = 1
        while True:
            idx = left[order[sort_id]]          # IndexError when the pool is exhausted, as upstream
            sort_id += 1
            assert idx not in picks, f"id: {idx} has been selected"
            cost += self.cost_f
            cost += self.infos_origin[idx]["gt_names"].shape[0] * self.cost_b
            if cost > int(self.current_budget):
                break
            picks.append(idx)
        self.selected_index[self.current_budget] = picks + sampled


class _WeightedFeatureSelector(BaseSelector):
    """Shared tail of BADGE / UWE: float32 L1 map of the weighted embeddings + greedy
    (badge_selector.py:92-178, uwe_selector.py:113-197); needs a non-empty buffer."""

    def _init_common(self, weighted_feat_path, distance_store_file, p):
        self.weighted_feat_path = weighted_feat_path
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
        if self.pred:
            feats = self.buffer_pred(**kwargs)
        else:
            feats = torch.load(self.weighted_feat_path, weights_only=True)
        feats = feats.to(device=device, dtype=torch.float32)
        self._greedy(self.get_feature_distance_map(feats), device, order="selected+sampled")


@SELECTORS.register_module
class BadgeSelector(_WeightedFeatureSelector):
    """Embeddings scaled by the frame's mean box entropy (badge_selector.py:15-178)."""

    def __init__(
            self,
            budget: int,
            buffer_file: str,
            dump_file_name: Optional[str] = None,
            infos_origin: List[Dict] = [],
            weighted_feat_path: str = "/home/st2000/data/buffers/badge_feat.pt",
            distance_store_file: str = "/home/st2000/data/buffers/badge_distance_map.npy",
            p: int = 2,
            detector: Optional[torch.nn.Module] = None,
            dataloader=None,
            logger: Optional[logging.Logger] = None,
            pred: bool = True,
            cost_b: float = 0.04,
            cost_f: float = 0.12,
    ) -> None:
        super().__init__(budget, buffer_file, dump_file_name, infos_origin=infos_origin,
                         detector=detector, dataloader=dataloader, logger=logger, pred=pred,
                         cost_b=cost_b, cost_f=cost_f)
        self._init_common(weighted_feat_path, distance_store_file, p)

    def buffer_pred(self, **kwargs) -> torch.Tensor:
        from .. import selector_ops as ops
        from ..sweep import sweep_embeddings
        device = self._device(kwargs)
        feats, entropy = sweep_embeddings(self.detector, self.dataloader, device,
                                          num_frames=len(self.infos_origin), with_entropy=True)
        prediction = ops.scale_rows(feats.contiguous(), entropy.contiguous())
        if self.weighted_feat_path and _rank() == 0:
            torch.save(prediction.cpu(), self.weighted_feat_path)
        self.detector = None
        return prediction


@SELECTORS.register_module
class UWESelector(_WeightedFeatureSelector):
    """Two sweeps: frame entropies -> min-max normalised; then embeddings scaled by
    ``uncertainty_norm[b_i]`` with the *batch-local* index b_i (uwe_selector.py:51-111)."""

    def __init__(
            self,
            budget: int,
            buffer_file: str,
            dump_file_name: Optional[str] = None,
            infos_origin: List[Dict] = [],
            weighted_feat_path: str = "/home/st2000/data/buffers/uwe_feat.pt",
            distance_store_file: str = "/home/st2000/data/buffers/uwe_distance_map.npy",
            p: int = 2,
            detector: Optional[torch.nn.Module] = None,
            dataloader=None,
            logger: Optional[logging.Logger] = None,
            pred: bool = True,
            cost_b: float = 0.04,
            cost_f: float = 0.12,
    ) -> None:
        super().__init__(budget, buffer_file, dump_file_name, infos_origin=infos_origin,
                         detector=detector, dataloader=dataloader, logger=logger, pred=pred,
                         cost_b=cost_b, cost_f=cost_f)
        self._init_common(weighted_feat_path, distance_store_file, p)

    def buffer_pred(self, **kwargs) -> torch.Tensor:
        from .. import selector_ops as ops
        from ..sweep import sweep_embeddings
        device = self._device(kwargs)
        n = len(self.infos_origin)
        _, entropy = sweep_embeddings(self.detector, self.dataloader, device, num_frames=n,
                                      with_entropy=True)
        norm = ops.minmax_norm(entropy.contiguous())
        prediction = sweep_embeddings(self.detector, self.dataloader, device, num_frames=n,
                                      batch_local_weights=norm)
        if self.weighted_feat_path and _rank() == 0:
            torch.save(prediction.cpu(), self.weighted_feat_path)
        self.detector = None
        return prediction


@SELECTORS.register_module
class PPALSelector(_WeightedFeatureSelector):
    """Two-stage PPAL (ppal_selector.py:18-239): a class-weighted-entropy ranking picks a
    candidate pool worth ``delta`` times the budget, then the greedy k-center runs on the L1
    embedding map restricted to (pool + labelled) frames."""

    def __init__(
            self,
            budget: int,
            buffer_file: str,
            dump_file_name: Optional[str] = None,
            infos_origin: List[Dict] = [],
            ent_path: str = "/home/st2000/data/buffers/ppal_ent.pt",
            feat_path: str = "/home/st2000/data/buffers/ppal_feat.pt",
            distance_store_file: str = "/home/st2000/data/buffers/ppal_distance_map.npy",
            class_weight_file: str = "tools/diff_category_average.json",
            p: int = 2,
            detector: Optional[torch.nn.Module] = None,
            dataloader=None,
            logger: Optional[logging.Logger] = None,
            pred: bool = True,
            cost_b: float = 0.04,
            cost_f: float = 0.12,
            delta: int = 4,
    ) -> None:
        super().__init__(budget, buffer_file, dump_file_name, infos_origin=infos_origin,
                         detector=detector, dataloader=dataloader, logger=logger, pred=pred,
                         cost_b=cost_b, cost_f=cost_f)
        self._init_common(feat_path, distance_store_file, p)
        self.ent_path = ent_path
        self.feat_path = feat_path
        self.class_weight_file = class_weight_file
        self.delta = delta

    def buffer_pred(self, sampled_index_list=None, left_index_list=None, **kwargs):
        import json
        from ..sweep import example_to_device, gap_embedding, gather_in_dataset_order
        device = self._device(kwargs)
        with open(self.class_weight_file, "r") as f:
            class_weight = json.load(f)
        names = [c for group in self.detector.bbox_head.class_names for c in group]
        cw = torch.tensor([class_weight[c] for c in names], dtype=torch.float32, device=device)
        ents, feats, index, seen = [], [], [], 0
        sampler = list(getattr(self.dataloader, "sampler", []) or [])
        with torch.no_grad():
            for batch in self.dataloader:
                example = example_to_device(batch, device)
                preds, middle = self.detector(example, return_loss=False, estimate=True)
                emb = gap_embedding(middle[-1])
                if hasattr(preds, "frame_weighted_entropy"):
                    ents.append(preds.frame_weighted_entropy(cw))
                else:                                   # plain per-frame dicts (e.g. TransFusionHead.predict)
                    from ..sweep import _weighted_entropy_of
                    ents.append(torch.stack([_weighted_entropy_of(p["scores"], p["label_preds"], cw) for p in preds]))
                feats.append(emb)
                b = emb.shape[0]
                index.extend(sampler[seen:seen + b] if sampler else range(seen, seen + b))
                seen += b
        n = len(self.infos_origin)
        idx = torch.as_tensor(index, dtype=torch.int64, device=device)
        feat_pred = gather_in_dataset_order(torch.cat(feats), idx, n)
        entropy_pred = gather_in_dataset_order(torch.cat(ents).unsqueeze(1), idx, n).squeeze(1)
        if _rank() == 0:
            if self.feat_path:
                torch.save(feat_pred.cpu(), self.feat_path)
            if self.ent_path:
                torch.save(entropy_pred.cpu(), self.ent_path)
        self.detector = None
        return entropy_pred, feat_pred

    def select_samples(self, **kwargs) -> None:
        from .. import selector_ops as ops
        device = self._device(kwargs)
        n = len(self.infos_origin)
        sampled = list(self.buffer[self.get_max_key()])
        left = list(range(n))
        for x in sampled:
            left.remove(x)
        if self.pred:
            ents, feats = self.buffer_pred(sampled_index_list=sampled, left_index_list=left, **kwargs)
        else:
            ents = torch.load(self.ent_path, weights_only=True)
            feats = torch.load(self.feat_path, weights_only=True)
        ents = ents.to(device=device, dtype=torch.float32)
        feats = feats.to(device=device, dtype=torch.float32)
        distance_map = self.get_feature_distance_map(feats).clone()
        # stage 1: entropy-ranked candidate pool under the expanded budget (ppal_selector.py:169-189)
        order = ops.argsort_desc(ents[torch.as_tensor(left, device=device)].contiguous()).tolist()
        pool = [left[order[0]]]
        cost = self.get_cost_amount()
        cost += self.cost_f
        cost += self.infos_origin[order[0]]["gt_names"].shape[0] * self.cost_b      # quirk A.1 #7
        limit = int(self.current_budget) + self.budget * (self.delta - 1)
        sort_id = 1
        while True:
            idx = left[order[sort_id]]
            sort_id += 1
            assert idx not in pool, f"id: {idx} has been selected"
            cost += self.cost_f
            cost += self.infos_origin[idx]["gt_names"].shape[0] * self.cost_b
            if cost > limit:
                break
            pool.append(idx)
        # stage 2: greedy k-center restricted to pool + labelled frames (:191-236)
        keep = torch.zeros(n, dtype=torch.uint8, device=device)
        keep[torch.as_tensor(pool + sampled, device=device)] = 1
        ops.mask_map_(distance_map, keep)
        self._greedy(distance_map, device, order="selected+sampled")
