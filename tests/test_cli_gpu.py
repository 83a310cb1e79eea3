"""GPU suite: tools/active_select.py end to end on a synthetic pool (same two-invocation flow as the
reference CLI: the first call bootstraps the empty buffer, the second sweeps + selects + dumps)."""
import json
import os
import pickle
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cli_bootstrap_then_select(tmp_path):
    cmd = [sys.executable, os.path.join(ROOT, "tools", "active_select.py"), "--config",
           os.path.join(ROOT, "examples", "active", "cbgs_spatial_temporal_feature.py"), "--budget", "20",
           "--pred", "--synthetic-scenes", "2", "--batch", "8"]
    env = dict(os.environ, PYTHONPATH=ROOT)
    r1 = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    buf = tmp_path / "data" / "buffers" / "spatial_temporal_feature.json"
    assert json.load(open(buf)) == {"0": []}
    r2 = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    out = json.load(open(buf))
    assert list(out) == ["0", "20"] and len(out["20"]) >= 5 and len(set(out["20"])) == len(out["20"])
    assert all(0 <= i < 80 for i in out["20"])
    infos = pickle.load(open(tmp_path / "data" / "nuScenes" / "infos_train_10sweeps_withvelo_20.pkl", "rb"))
    assert len(infos) == len(out["20"])
    # the picks cost at most the budget under the reference's cost model
    pool = pickle.load(open(tmp_path / "data" / "nuScenes" / "infos_train_10sweeps_withvelo.pkl", "rb"))
    cost = sum(0.12 + 0.04 * len(pool[i]["gt_names"]) for i in out["20"])
    assert cost <= 20


def test_cli_with_the_bevfusion_lidar_config(tmp_path):
    """Same flow on the head-less BEVFusion lidar-branch config (0.075 m grid, embeddings only)."""
    cmd = [sys.executable, os.path.join(ROOT, "tools", "active_select.py"), "--config",
           os.path.join(ROOT, "examples", "active", "bevfusion_lidar_spatial_temporal_feature.py"), "--budget", "20",
           "--pred", "--synthetic-scenes", "1", "--batch", "4"]
    env = dict(os.environ, PYTHONPATH=ROOT)
    for _ in range(2):
        r = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
    out = json.load(open(tmp_path / "data" / "buffers" / "bevfusion_lidar_stf.json"))
    assert list(out) == ["0", "20"] and len(out["20"]) >= 5 and all(0 <= i < 40 for i in out["20"])


@pytest.mark.parametrize("config,buffer", [("bevfusion_camera_lidar_spatial_temporal_feature.py", "bevfusion_camera_lidar_stf.json"),
                                           ("bevfusion_camera_lidar_entropy.py", "bevfusion_camera_lidar_entropy.json"),
                                           ("bevfusion_lidar_entropy.py", "bevfusion_lidar_entropy.json")])
def test_cli_with_the_bevfusion_detector_configs(tmp_path, config, buffer):
    """The same two-invocation flow on BASELINE configs[4] (the registered ``BEVFusion`` camera+lidar detector over
    ``CameraLidarSweepLoader`` batches: fused-BEV embeddings, and with the TransFusionHead under the EntropySelector) and on the
    lidar-only detector with its head (configs[3])."""
    cmd = [sys.executable, os.path.join(ROOT, "tools", "active_select.py"), "--config",
           os.path.join(ROOT, "examples", "active", config), "--budget", "20", "--pred", "--synthetic-scenes", "1",
           "--batch", "4"]
    env = dict(os.environ, PYTHONPATH=ROOT)
    for _ in range(2):
        r = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
    out = json.load(open(tmp_path / "data" / "buffers" / buffer))
    assert list(out) == ["0", "20"] and len(out["20"]) >= 5 and all(0 <= i < 40 for i in out["20"])
    assert len(set(out["20"])) == len(out["20"])
