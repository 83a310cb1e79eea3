"""CPU suite: host-side plumbing that mirrors the reference's plugin API (registry, config,
fileio, selector bookkeeping, real-file point loading)."""
import json
import os
import pickle

import numpy as np
import pytest

from al3d import synthetic
from al3d.utils import Config, Registry, build_from_cfg, fileio


def test_registry_contract():
    reg = Registry("thing")

    @reg.register_module
    class A:
        def __init__(self, x, y=2):
            self.x, self.y = x, y

    assert reg.get("A") is A and "A" in repr(reg)
    obj = build_from_cfg(dict(type="A", x=1), reg, default_args=dict(y=5, x=9))
    assert (obj.x, obj.y) == (1, 5)                       # cfg wins over default_args
    assert build_from_cfg(dict(type=A, x=3), reg).x == 3  # a class works as `type`
    with pytest.raises(KeyError):
        reg.register_module(A)                            # duplicate name
    with pytest.raises(KeyError):
        build_from_cfg(dict(type="Nope"), reg)
    with pytest.raises(TypeError):
        reg.register_module(lambda: 0)
    with pytest.raises(TypeError):
        build_from_cfg(dict(type=3), reg)


def test_selector_and_model_registries_carry_reference_names():
    from al3d.models import BACKBONES, DETECTORS, HEADS, NECKS, READERS
    from al3d.selectors import SELECTORS
    for name in ("BaseSelector", "RandomSelector", "SpatialSelector", "TemporalSelector",
                 "EuSpatialSelector", "SpatialTemporalSelector", "FeatureSelector",
                 "SpatialFeatureSelector", "SpatialTemporalFeatureSelector", "EntropySelector",
                 "BadgeSelector", "UWESelector", "PPALSelector"):
        assert SELECTORS.get(name) is not None, name
    assert DETECTORS.get("FPNVoxelNet") and DETECTORS.get("VoxelNet")
    assert READERS.get("VoxelFeatureExtractorV3") and BACKBONES.get("FPNSpMiddleResNetFHD")
    assert NECKS.get("RPN") and HEADS.get("MultiGroupHead")


def test_config_py_base_inheritance(tmp_path):
    (tmp_path / "base.py").write_text("a = 1\nmodel = dict(type='X', depth=3, inner=dict(k=1))\n")
    (tmp_path / "child.py").write_text("_base_ = 'base.py'\nmodel = dict(depth=5, inner=dict(j=2))\nb = [1, dict(c=2)]\n")
    cfg = Config.fromfile(str(tmp_path / "child.py"))
    assert cfg.a == 1 and cfg.model.type == "X" and cfg.model.depth == 5
    assert cfg.model.inner.k == 1 and cfg.model.inner.j == 2 and cfg.b[1].c == 2
    cfg.model.depth = 7
    assert cfg["model"]["depth"] == 7 and "model" in cfg
    with pytest.raises(FileNotFoundError):
        Config.fromfile(str(tmp_path / "missing.py"))


def test_shipped_configs_load():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "active")
    for f in sorted(os.listdir(root)):
        if f.startswith("cbgs_"):
            cfg = Config.fromfile(os.path.join(root, f))
            assert cfg.selector.type.endswith("Selector") and cfg.model.type == "FPNVoxelNet"
            assert len(cfg.tasks) == 6 and cfg.voxel_generator.max_voxel_num == 60000


def test_fileio_roundtrip(tmp_path):
    obj = {"0": [], "600": [3, 1, 2]}
    p = str(tmp_path / "b.json")
    fileio.dump(obj, p)
    assert fileio.load(p) == obj and open(p).read() == json.dumps(obj)     # no indent by default
    p2 = str(tmp_path / "i.pkl")
    fileio.dump([{"a": np.arange(3)}], p2)
    assert pickle.load(open(p2, "rb"))[0]["a"].tolist() == [0, 1, 2]
    with pytest.raises(TypeError):
        fileio.load(str(tmp_path / "x.txt"))


def _selector(tmp_path, buffer):
    from al3d.selectors import BaseSelector
    infos, _ = synthetic.make_pool(2, seed=5, frames_per_scene=6)
    ip, bp = str(tmp_path / "infos.pkl"), str(tmp_path / "buf.json")
    pickle.dump(infos, open(ip, "wb"))
    json.dump(buffer, open(bp, "w"))
    return BaseSelector(budget=10, buffer_file=bp, infos_origin=ip), infos, ip, bp


def test_base_selector_bookkeeping(tmp_path):
    sel, infos, ip, bp = _selector(tmp_path, {"0": [], "20": [4, 1]})
    assert sel.get_max_key() == "20" and sel.current_budget == "30"
    want = 0
    want += 0.12 * 2
    for i in (4, 1):
        want += infos[i]["gt_names"].shape[0] * 0.04
    assert sel.get_cost_amount() == want                      # same float64 accumulation order
    assert sel._run_ids().tolist() == [0] * 6 + [1] * 6
    assert sel._max_temporal_distance() == 6                  # last run ignored (A.1 quirk 4)
    xy = sel._ego_xy()
    assert xy.shape == (12, 2)
    want_xy = np.stack([(-(i["car_from_global"][:3, 3].T @ i["car_from_global"][:3, :3]))[:2] for i in infos])
    assert np.array_equal(xy, want_xy)
    sel.selected_index["30"] = [4, 1, 7]
    sel.dump_file()
    assert json.load(open(bp)) == {"0": [], "20": [4, 1], "30": [4, 1, 7]}
    out = pickle.load(open(ip.replace(".pkl", "_30.pkl"), "rb"))
    assert [o["token"] for o in out] == [infos[i]["token"] for i in (4, 1, 7)]


def test_selectors_refuse_to_run_without_a_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from al3d.selectors import build_selector
    sel, infos, ip, bp = _selector(tmp_path, {"0": []})
    s = build_selector(dict(type="TemporalSelector", budget=5, buffer_file=bp, infos_origin=ip))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        s.select_samples(local_rank=0)


def test_real_file_loader_rules(tmp_path):
    from al3d.datasets.nusc_files import load_frame_points
    rng = np.random.default_rng(0)
    key = rng.uniform(-20, 20, size=(50, 5)).astype(np.float32)
    sw = rng.uniform(-20, 20, size=(40, 5)).astype(np.float32)
    sw[:5, :2] = 0.3                                         # too close: dropped in the sweep only
    key[:3, :2] = 0.2                                        # key frame keeps its close points
    key.tofile(tmp_path / "k.bin")
    sw.tofile(tmp_path / "s.bin")
    T = np.eye(4)
    T[:3, 3] = [1.0, -2.0, 0.5]
    info = {"lidar_path": "k.bin", "sweeps": [{"lidar_path": "s.bin", "transform_matrix": T, "time_lag": 0.05},
                                              {"lidar_path": "s.bin", "transform_matrix": None, "time_lag": 0.1}]}
    pts = load_frame_points(info, nsweeps=3, root=str(tmp_path))
    assert pts.dtype == np.float32 and pts.shape == (50 + 35 + 35, 5)
    assert np.array_equal(pts[:50, :4], key[:, :4]) and np.all(pts[:50, 4] == 0)
    np.testing.assert_allclose(pts[50:85, :3], sw[5:, :3] + T[:3, 3], rtol=1e-6)
    assert np.all(pts[50:85, 4] == np.float32(0.05)) and np.all(pts[85:, 4] == np.float32(0.1))
    with pytest.raises(AssertionError):
        load_frame_points(info, nsweeps=10, root=str(tmp_path))


def test_cald_selector_replays_reference_golden(tmp_path):
    """CaldSelector is host-only list logic; golden from the reference class (oracle/gen_golden_cald.py)."""
    import pickle
    from al3d import synthetic
    from al3d.selectors import build_selector
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "cald_seeded.npz"), allow_pickle=False)
    infos, _ = synthetic.make_pool(int(z["pool_scenes"]), seed=int(z["pool_seed"]))
    assert [len(i["gt_names"]) for i in infos] == z["n_boxes"].tolist()
    ip, bp, sp, jp = (str(tmp_path / f) for f in ("infos.pkl", "buffer.json", "sorted.json", "jsdiv.pkl"))
    pickle.dump(infos, open(ip, "wb"))
    open(bp, "w").write(str(z["buffer_json"]))
    json.dump(z["sorted_idx"].tolist(), open(sp, "w"))
    pickle.dump(dict(zip(z["jsdiv_keys"].tolist(), z["jsdiv_vals"].tolist())), open(jp, "wb"))
    sel = build_selector(dict(type="CaldSelector", budget=int(z["budget"]), buffer_file=bp, infos_origin=ip,
                              buffer_path=sp, jsdiv_path=jp))
    sel.select_samples(local_rank=0)
    assert sel.current_budget == str(z["current_budget"])
    assert sel.selected_index[sel.current_budget] == z["selected"].tolist()
    # the same ranking as JSON
    jj = str(tmp_path / "jsdiv.json")
    json.dump({str(k): v for k, v in zip(z["jsdiv_keys"].tolist(), z["jsdiv_vals"].tolist())}, open(jj, "w"))
    sel2 = build_selector(dict(type="CaldSelector", budget=int(z["budget"]), buffer_file=bp, infos_origin=ip,
                               buffer_path=sp, jsdiv_path=jj))
    sel2.select_samples(local_rank=0)
    assert sel2.selected_index[sel2.current_budget] == z["selected"].tolist()


def test_every_example_config_builds_its_selector_type():
    """examples/active/cbgs_*.py mirror the reference's config names; each names a registered selector."""
    import glob
    from al3d.selectors import SELECTORS
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "active")
    files = sorted(glob.glob(os.path.join(root, "cbgs_*.py")))
    assert len(files) == 12
    for f in files:
        cfg = Config.fromfile(f)
        assert cfg.selector["type"] in SELECTORS.module_dict, f
        assert cfg.model["type"] == "FPNVoxelNet"


def test_pipeline_registry_builds_the_reference_val_pipeline():
    """PIPELINES holds the stage names of the reference's test_pipeline; construction needs no GPU."""
    from al3d.datasets import PIPELINES, Compose, SweepDataset
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "active")
    cfg = Config.fromfile(os.path.join(root, "cbgs_spatial_temporal.py"))
    names = [t["type"] for t in cfg.test_pipeline]
    assert names == ["LoadPointCloudFromFile", "LoadPointCloudAnnotations", "Preprocess", "Voxelization",
                     "AssignTarget", "Reformat"]
    assert set(names) <= set(PIPELINES.module_dict)
    ds = SweepDataset([{"lidar_path": "x", "sweeps": []}], cfg.data["val"]["pipeline"], nsweeps=cfg.nsweeps)
    assert len(ds) == 1 and isinstance(ds.pipeline, Compose) and len(ds.pipeline.transforms) == 6
    with pytest.raises(NotImplementedError):
        Compose([dict(type="Preprocess", cfg=dict(mode="train"))])


def test_swin_row_maps_equal_roll_pad_partition_and_unfold():
    """Host logic of the Swin-T token path (al3d/token_ops.py): the (shifted) window row map equals torch's pad -> roll ->
    window partition of an index image, every token appears exactly once, padding is -1; the patch-merging map equals
    ``F.unfold`` of the corner-padded index image in piece-major order.  (The kernels that consume them: tests/test_swin_gpu.py.)"""
    import importlib.util
    import numpy as np
    import torch
    import torch.nn.functional as F
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = [d for d in os.listdir(root) if d.endswith("_amd") and os.path.isdir(os.path.join(root, d, "csrc"))][0]
    src = open(os.path.join(root, pkg, "token_ops.py")).read()
    # the two map builders are pure numpy: evaluate them without importing the package (which loads the HIP library)
    ns = {"np": np}
    start = src.index("def window_rowmap(")
    exec(src[start:], ns)
    for (B, H, W, shift) in ((2, 9, 11, 3), (1, 14, 21, 0), (3, 7, 7, 3), (2, 64, 176, 3)):
        m, (nwy, nwx) = ns["window_rowmap"](B, H, W, 7, shift)
        Hp, Wp = nwy * 7, nwx * 7
        assert Hp >= H and Wp >= W and Hp - H < 7 and Wp - W < 7
        assert sorted(m[m >= 0].tolist()) == list(range(B * H * W)) and int((m < 0).sum()) == B * (Hp * Wp - H * W)
        idx = torch.arange(B * H * W, dtype=torch.float32).view(B, H, W, 1) + 1.0
        pad = F.pad(idx, (0, 0, 0, Wp - W, 0, Hp - H))
        sh = torch.roll(pad, (-shift, -shift), (1, 2)).view(B, nwy, 7, nwx, 7).permute(0, 1, 3, 2, 4).reshape(-1)
        assert np.array_equal(sh.long().numpy() - 1, m.astype(np.int64))
        mm, (OH, OW) = ns["merge_rowmap"](B, H, W)
        un = F.unfold(F.pad(idx.permute(0, 3, 1, 2), (0, W % 2, 0, H % 2)), 2, stride=2)
        assert (OH, OW) == ((H + 1) // 2, (W + 1) // 2)
        assert np.array_equal(un.transpose(1, 2).reshape(-1).long().numpy() - 1, mm.astype(np.int64))
