#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Golden vector for the reference's ``CaldSelector``
(det3d/selectors/cald_selector.py:18-140): a two-stage replay of precomputed rankings.

The reference class reads its second ranking from a hard-coded absolute path
(cald_selector.py:96) that does not exist here; ``builtins.open`` is redirected for that one path
to a temp file while the reference's own ``select_samples`` runs -- no reference code is edited.
"""
import builtins
import json
import logging
import os
import pickle
import random
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_import  # noqa: E402
from al3d import synthetic  # noqa: E402

HARD = "/home/linjp/share/ActiveLearn4Detection-main/idx_to_jsdiv.pkl"


def main():
    import importlib
    ref_import.import_selectors()
    mod = importlib.import_module("det3d.selectors.cald_selector")
    infos, logs = synthetic.make_pool(8, seed=21)            # N = 320
    n = len(infos)
    rng = np.random.default_rng(9)
    sampled = sorted(rng.choice(n, 12, replace=False).tolist())
    buffer = {"0": [], "40": sampled}
    sorted_idx = rng.permutation(n).tolist()                 # entropy ranking of ALL frames (labelled ones are removed)
    jsdiv_keys = rng.permutation(n).tolist()
    jsdiv_vals = rng.uniform(0, 1, n).round(6).tolist()
    budget = 30
    with tempfile.TemporaryDirectory() as td:
        ip, bp, sp, jp = (os.path.join(td, f) for f in ("infos.pkl", "buffer.json", "sorted.json", "jsdiv.pkl"))
        pickle.dump(infos, open(ip, "wb"))
        json.dump(buffer, open(bp, "w"))
        json.dump(sorted_idx, open(sp, "w"))
        pickle.dump(dict(zip(jsdiv_keys, jsdiv_vals)), open(jp, "wb"))
        real_open = builtins.open

        def redirect(path, *a, **k):
            return real_open(jp if path == HARD else path, *a, **k)
        random.seed(3407)
        sel = mod.CaldSelector(budget=budget, buffer_file=bp, infos_origin=ip, buffer_path=sp,
                               logger=logging.getLogger("golden"))
        builtins.open = redirect
        try:
            sel.select_samples()
        finally:
            builtins.open = real_open
        key = sel.current_budget
        selected = sel.selected_index[key]
    _, _, n_boxes = synthetic.pool_arrays(infos)
    out = dict(n_boxes=n_boxes, buffer_json=np.array(json.dumps(buffer)), budget=np.int64(budget),
               sorted_idx=np.array(sorted_idx, dtype=np.int64), jsdiv_keys=np.array(jsdiv_keys, dtype=np.int64),
               jsdiv_vals=np.array(jsdiv_vals, dtype=np.float64), current_budget=np.array(key),
               selected=np.array(selected, dtype=np.int64), pool_seed=np.int64(21), pool_scenes=np.int64(8))
    path = os.path.join(ROOT, "tests", "golden", "cald_seeded.npz")
    np.savez_compressed(path, **out)
    print("cald: key", key, "picked", len(selected), "first", selected[:8])


if __name__ == "__main__":
    main()
