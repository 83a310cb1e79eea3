#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Generate ``tests/golden/selector_*.npz``.

Runs the *reference's own* selector classes (imported unmodified from
/root/reference through ``ref_import``) on synthetic pools and records inputs
and outputs.  Run in the build container only:

    python oracle/gen_golden_selectors.py

Each fixture holds the flat inputs the selectors read (``car_from_global``,
logfile ids, ``n_boxes``), the buffer, the constructor arguments, and the
reference's ``selected_index`` list.  For small pools the reference's own
intermediate arrays (spatial map, combined distance map) are captured from the
live ``select_samples`` frame with a trace hook -- no reference code is edited.
"""
import json
import logging
import os
import pickle
import random
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_import  # noqa: E402
from al3d import synthetic  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED = 3407  # tools/active_select.py:76-80


def _capture_locals(func, names):
    """Call ``func`` and return the named locals of the select_samples frame."""
    captured = {}

    def tracer(frame, event, arg):
        if event == "call" and frame.f_code.co_name == "select_samples":
            def local(frame, event, arg):
                if event == "return":
                    for n in names:
                        if n in frame.f_locals:
                            captured[n] = frame.f_locals[n]
                return local
            return local
        return None

    sys.settrace(tracer)
    try:
        func()
    finally:
        sys.settrace(None)
    return captured


def run_case(name, cls_name, module, infos, logs, buffer, budget, kwargs=None,
             feats_seed=None, feats_scale=0.01, capture=False, expect_assert=False,
             entropy=None, modname=None):
    kwargs = dict(kwargs or {})
    feats = None
    if feats_seed is not None:
        feats = synthetic.make_embeddings(len(infos), seed=feats_seed, scale=feats_scale)
    mods = ref_import.import_selectors()
    if module not in mods:
        import importlib
        mods[module] = importlib.import_module("det3d.selectors." + module)
    cls = getattr(mods[module], cls_name)
    with tempfile.TemporaryDirectory() as td:
        infos_p = os.path.join(td, "infos.pkl")
        with open(infos_p, "wb") as f:
            pickle.dump(infos, f)
        logs_p = os.path.join(td, "log.json")
        with open(logs_p, "w") as f:
            json.dump(logs, f)
        buf_p = os.path.join(td, "buffer.json")
        with open(buf_p, "w") as f:
            json.dump(buffer, f)
        ctor = dict(budget=budget, buffer_file=buf_p, infos_origin=infos_p,
                    logger=logging.getLogger("golden"), pred=False)
        if "Feature" in cls_name:
            fp = os.path.join(td, "feats.pt")
            torch.save(torch.from_numpy(feats), fp)
            ctor["buffer_path"] = fp
        if cls_name in ("BadgeSelector", "UWESelector"):
            fp = os.path.join(td, "wfeats.pt")
            torch.save(torch.from_numpy(feats), fp)
            ctor["weighted_feat_path"] = fp
        if cls_name == "PPALSelector":
            fp = os.path.join(td, "pfeats.pt")
            torch.save(torch.from_numpy(feats), fp)
            ep = os.path.join(td, "pent.pt")
            torch.save(torch.from_numpy(entropy), ep)
            ctor.update(feat_path=fp, ent_path=ep, class_weight_file=os.path.join(td, "cw.json"))
        if cls_name == "EntropySelector":
            fp = os.path.join(td, "entropy.pt")
            torch.save(torch.from_numpy(entropy), fp)
            ctor["buffer_path"] = fp
        if cls_name not in ("FeatureSelector", "TemporalSelector", "RandomSelector", "EntropySelector",
                            "BadgeSelector", "UWESelector", "PPALSelector"):
            ctor["logs_file"] = logs_p
        if cls_name not in ("TemporalSelector", "RandomSelector", "EntropySelector"):
            ctor["distance_store_file"] = os.path.join(td, "dist.npy")
        ctor.update(kwargs)
        torch.manual_seed(SEED)
        np.random.seed(SEED)
        random.seed(SEED)
        sel = cls(**ctor)
        got = {}
        err = ""
        try:
            if capture:
                got = _capture_locals(sel.select_samples,
                                      ["distance_map", "spatial_distance_map",
                                       "temporal_distance_map", "feature_distance_map"])
            else:
                sel.select_samples()
        except AssertionError as e:
            if not expect_assert:
                raise
            err = "AssertionError"
        sys.stdout.write("\n")
        key = sel.current_budget
        selected = np.array(sel.selected_index.get(key, []), dtype=np.int64)
        raw_spatial = None
        dsf = ctor.get("distance_store_file")
        if capture and dsf and os.path.exists(dsf) and cls_name != "FeatureSelector":
            raw_spatial = np.load(dsf)
    cfg, run_id, n_boxes = synthetic.pool_arrays(infos)
    logfiles = np.array([i["cam_front_path"].split("/")[-1].split("__")[0] for i in infos])
    # ego XY exactly as the reference evaluates it (spatial_temporal_selector.py:83-89);
    # numpy routes the 3x3 product through BLAS, so the bits are recorded, not re-derived.
    ego_xy = np.stack([(-(i["car_from_global"][:3, 3].T @ i["car_from_global"][:3, :3]))[:2]
                       for i in infos])
    out = dict(
        car_from_global=cfg, ego_xy=ego_xy, run_id=run_id, n_boxes=n_boxes, logfiles=logfiles,
        buffer_json=np.array(json.dumps(buffer)), budget=np.int64(budget),
        cls_name=np.array(cls_name), kwargs_json=np.array(json.dumps(kwargs)),
        current_budget=np.array(key), selected=selected, error=np.array(err),
        logs_json=np.array(json.dumps(logs)),
    )
    if entropy is not None:
        out["entropy"] = entropy
    if feats is not None:
        # embeddings are regenerated from the seed by the tests
        # (al3d.synthetic.make_embeddings); the digest guards the generator.
        import hashlib
        out["feats_seed"] = np.int64(feats_seed)
        out["feats_scale"] = np.float64(feats_scale)
        out["feats_sha256"] = np.array(hashlib.sha256(feats.tobytes()).hexdigest())
    if capture:
        for k, v in got.items():
            if isinstance(v, torch.Tensor):
                v = v.numpy()
            if isinstance(v, np.ndarray):
                out["ref_" + k] = v
        if raw_spatial is not None:
            out["ref_raw_spatial_map"] = raw_spatial
    path = os.path.join(OUT, f"selector_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: N={len(infos)} key={key} picked={len(selected)} err={err!r} "
          f"first={selected[:8].tolist()} -> {os.path.getsize(path)/1024:.0f} KiB")


def main():
    os.makedirs(OUT, exist_ok=True)
    logging.basicConfig(level=logging.ERROR)
    small, small_logs = synthetic.make_pool(6, seed=1)          # N = 240
    empty = {"0": []}
    seeded = {"0": [], "50": [5, 17, 200]}
    ST = ("SpatialTemporalSelector", "spatial_temporal_selector")
    run_case("st_exp_sum_empty", *ST, small, small_logs, empty, 40, capture=True)
    run_case("st_exp_sum_seeded", *ST, small, small_logs, seeded, 30, capture=True)
    run_case("st_linear_sum", *ST, small, small_logs, empty, 40,
             kwargs=dict(normalize="linear"), capture=True)
    run_case("st_exp_min", *ST, small, small_logs, seeded, 30,
             kwargs=dict(aggregate="min"), capture=True)
    run_case("st_exp_max", *ST, small, small_logs, empty, 40,
             kwargs=dict(aggregate="max"), capture=True)
    run_case("st_lambda_half_k4", *ST, small, small_logs, empty, 40,
             kwargs=dict(lambda_t=0.5, k=4), capture=True)
    # exact straight tracks (no jitter): many near-equal geodesics
    straight, straight_logs = synthetic.make_pool(6, seed=2, jitter=0.0, yaw_jitter=0.0)
    run_case("st_straight", *ST, straight, straight_logs, empty, 40, capture=True)
    # budget larger than the whole pool -> duplicate-pick assertion (A.1 #13)
    tiny, tiny_logs = synthetic.make_pool(1, seed=3, frames_per_scene=12, max_boxes=3)
    run_case("st_overbudget", *ST, tiny, tiny_logs, empty, 1000, expect_assert=True)
    # headline configuration: 64 scenes, budget 600 (BASELINE.json configs[0..1])
    pool64, logs64 = synthetic.make_pool(64, seed=0)
    run_case("st_pool64_b600", *ST, pool64, logs64, empty, 600)
    run_case("st_pool64_b600_round2", *ST, pool64, logs64,
             {"0": [], "600": json.loads(json.dumps(
                 np.load(os.path.join(OUT, "selector_st_pool64_b600.npz"))["selected"].tolist()))},
             600)
    # single-term selectors sharing the greedy core
    run_case("spatial_empty", "SpatialSelector", "spatial_selector", small, small_logs,
             empty, 40, capture=True)
    run_case("temporal_seeded", "TemporalSelector", "temporal_selector", small, small_logs,
             seeded, 30, capture=True)
    run_case("euclid_empty", "EuSpatialSelector", "euclidean_spatial_selector", small,
             small_logs, empty, 40, capture=True)
    # feature family (float32 torch maps, selected + sampled order)
    run_case("feature_p2", "FeatureSelector", "feature_selector", small, small_logs,
             seeded, 30, kwargs=dict(p=2), feats_seed=5, capture=True)
    run_case("feature_p1", "FeatureSelector", "feature_selector", small, small_logs,
             seeded, 30, kwargs=dict(p=1), feats_seed=5, capture=True)
    STF = ("SpatialTemporalFeatureSelector", "spatial_temporal_feature_selector")
    run_case("stf_seeded", *STF, small, small_logs, seeded, 30,
             kwargs=dict(lambda_f=1.0, lambda_t=1.0), feats_seed=5, capture=True)
    run_case("stf_empty_lf2", *STF, small, small_logs, empty, 40,
             kwargs=dict(lambda_f=2.0, lambda_t=0.5, p=1), feats_seed=5, capture=True)
    run_case("stf_pool64_b600", *STF, pool64, logs64, empty, 600,
             kwargs=dict(lambda_f=1.0, lambda_t=1.0), feats_seed=6)
    # uncertainty family (pred=False: the swept quantities are loaded from .pt files)
    import importlib
    for mod in ("uwe_selector", "badge_selector"):
        ref_import.import_selectors()[mod] = importlib.import_module("det3d.selectors." + mod)
    rng = np.random.default_rng(9)
    ent = rng.uniform(0.05, 0.69, size=len(small)).astype(np.float32)
    # no exact ties: torch.argsort is not stable, the reference's tie order is unspecified
    run_case("entropy_seeded", "EntropySelector", "entropy_selector", small, small_logs, seeded, 30,
             entropy=ent)
    ent_nan = ent.copy()
    ent_nan[[10, 11, 100]] = np.nan                 # frames without detections
    run_case("entropy_empty_nan", "EntropySelector", "entropy_selector", small, small_logs, empty, 40,
             entropy=ent_nan)
    run_case("entropy_random_sample", "EntropySelector", "entropy_selector", small, small_logs, seeded, 30,
             kwargs=dict(random_sample=True, sample_num=100), entropy=ent)
    run_case("badge_seeded", "BadgeSelector", "badge_selector", small, small_logs, seeded, 30,
             kwargs=dict(p=2), feats_seed=5)
    run_case("uwe_seeded", "UWESelector", "uwe_selector", small, small_logs, seeded, 30,
             kwargs=dict(p=1), feats_seed=7)
    ent_sum = rng.uniform(0.5, 30.0, size=len(small)).astype(np.float32)     # PPAL: weighted entropy sums
    run_case("ppal_seeded", "PPALSelector", "ppal_selector", small, small_logs, seeded, 15,
             kwargs=dict(p=2, delta=4), feats_seed=8, entropy=ent_sum)
    run_case("ppal_delta2", "PPALSelector", "ppal_selector", small, small_logs, seeded, 20,
             kwargs=dict(p=1, delta=2), feats_seed=8, entropy=ent_sum)
    SF = ("SpatialFeatureSelector", "spatial_feature_selector")
    run_case("sf_seeded", *SF, small, small_logs, seeded, 30, feats_seed=5, capture=True)


if __name__ == "__main__":
    main()
