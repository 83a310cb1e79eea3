"""Random baseline (reference det3d/selectors/random_selector.py:14-66): pure
host-side ``random.choice`` under the cost budget -- no map, no kernel."""
import logging
import random
from typing import Dict, List, Optional

import torch

from .base_selector import BaseSelector
from .registry import SELECTORS


@SELECTORS.register_module
class RandomSelector(BaseSelector):
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
        sampled = self.buffer[self.get_max_key()]
        left = list(range(len(self.infos_origin)))
        for x in sampled:
            left.remove(x)
        cost = self.get_cost_amount()
        picks = []
        while True:
            idx = random.choice(left)
            cost += self.cost_f
            cost += self.infos_origin[idx]["gt_names"].shape[0] * self.cost_b
            if cost > int(self.current_budget):
                break
            picks.append(idx)
            left.remove(idx)
        self.selected_index[self.current_budget] = picks + sampled
