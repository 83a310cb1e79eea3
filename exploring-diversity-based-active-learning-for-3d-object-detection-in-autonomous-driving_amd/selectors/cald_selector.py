"""CALD replay selector (reference det3d/selectors/cald_selector.py:18-140).

Host-only: the reference's CALD numbers come from separate tools (``tools/cald_pred_list.py``,
``tools/cald_ent.py``) that write two rankings; this class only replays them under the cost model:

1. ``buffer_path`` -- a JSON list of frame ids sorted by entropy.  After dropping the labelled
   frames the list is consumed in order until the cost exceeds ``int(current_budget) + 0.5*budget``
   (the first entry is charged and kept unconditionally); this is the candidate pool.
2. ``jsdiv_path`` -- ``{frame id: JS divergence}`` (pickle like the reference's ``idx_to_jsdiv.pkl``
   or JSON); frames are visited by descending divergence and those inside the pool are taken
   until the cost exceeds ``int(current_budget)``.  The reference hard-codes this path
   (cald_selector.py:96); it is a constructor argument here with the same default.

Bug-compatible detail kept on purpose: the reference scans the divergence ranking with
``for i in lst: ... lst.remove(i)``, so after every miss the iterator skips the next element
(it looks at positions 0, 2, 4, ... of the shrinking list, restarting at 0 for every pick).
Output order ``selected + sampled``.
"""
import json
import logging
import pickle
from typing import Dict, List, Optional

from .base_selector import BaseSelector
from .registry import SELECTORS

_DEFAULT_SORTED = "/home/st2000/data/buffers/cald_ent_sorted_idx.json"
_DEFAULT_JSDIV = "/home/linjp/share/ActiveLearn4Detection-main/idx_to_jsdiv.pkl"


def _scan_skipping(order: list, pool: set):
    """One pass of the reference's remove-while-iterating scan over ``order`` (mutated in place):
    return the first examined id that is in ``pool`` (removed from ``order``), or None."""
    pos = 0
    while pos < len(order):
        cand = order.pop(pos)            # every examined element leaves the list ...
        if cand in pool:
            return cand
        pos += 1                         # ... and a miss makes the iterator jump over its successor
    return None


@SELECTORS.register_module
class CaldSelector(BaseSelector):
    def __init__(
            self,
            budget: int,
            buffer_file: str,
            dump_file_name: Optional[str] = None,
            infos_origin: List[Dict] = [],
            buffer_path: str = _DEFAULT_SORTED,
            detector=None,
            dataloader=None,
            logger: Optional[logging.Logger] = None,
            cost_b: float = 0.04,
            cost_f: float = 0.12,
            pred: bool = False,
            jsdiv_path: str = _DEFAULT_JSDIV,
    ) -> None:
        super().__init__(budget, buffer_file, dump_file_name, infos_origin=infos_origin,
                         detector=detector, dataloader=dataloader, logger=logger, cost_b=cost_b,
                         cost_f=cost_f, pred=pred)
        self.buffer_path = buffer_path
        self.jsdiv_path = jsdiv_path

    def _frame_cost(self, idx: int) -> float:
        return self.infos_origin[idx]["gt_names"].shape[0] * self.cost_b

    def _load_jsdiv(self) -> Dict[int, float]:
        if str(self.jsdiv_path).endswith(".json"):
            with open(self.jsdiv_path) as f:
                return {int(k): float(v) for k, v in json.load(f).items()}
        with open(self.jsdiv_path, "rb") as f:      # the user's own file, written by tools/cald_ent.py
            return pickle.load(f)

    def select_samples(self, **kwargs) -> None:
        sampled = list(self.buffer[self.get_max_key()])
        with open(self.buffer_path) as f:
            ranking = json.load(f)
        self.logger.info(f"all entropy results have been load from {self.buffer_path}")
        for x in sampled:
            ranking.remove(x)                       # ValueError if a labelled frame is not ranked, as in the reference
        # ---- stage 1: entropy-ranked candidate pool under 1.5x the round's budget
        pool = [ranking[0]]
        cost = self.get_cost_amount()
        cost += self.cost_f
        cost += self._frame_cost(ranking[0])
        limit = int(self.current_budget) + self.budget * 0.5
        at = 1
        while True:
            idx = ranking[at]                       # IndexError when the ranking runs out, as in the reference
            at += 1
            assert idx not in pool, f"id: {idx} has been selected"
            cost += self.cost_f
            cost += self._frame_cost(idx)
            if cost > limit:
                break
            pool.append(idx)
        # ---- stage 2: take pool members by descending JS divergence under the budget
        jsdiv = self._load_jsdiv()
        order = [k for k, _ in sorted(jsdiv.items(), key=lambda kv: kv[1], reverse=True)]
        members = set(pool)
        first = _scan_skipping(order, members)
        if first is None:
            raise NameError("no frame of the entropy pool appears in the divergence ranking")
        selected = [first]
        cost = self.get_cost_amount()
        cost += self.cost_f
        cost += self._frame_cost(first)
        last = first
        while True:
            nxt = _scan_skipping(order, members)
            if nxt is not None:
                last = nxt
            assert last not in selected, f"id: {last} has been selected"
            cost += self.cost_f
            cost += self._frame_cost(last)
            if cost > int(self.current_budget):
                break
            selected.append(last)
        self.selected_index[self.current_budget] = selected + sampled
