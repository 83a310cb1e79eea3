"""CPU suite: ``bench.py --gpus N`` cannot mislabel itself (the reference's launcher contract is
env:// ranks, tools/active_select.py:94-103).  ``--launch-check`` stops after the rendezvous and the rank census, so
this runs without a GPU over gloo."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR",
                                                             "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="1", AL3D_DIST_BACKEND="gloo", **kw)
    return env


def test_gpus_2_without_a_launcher_starts_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 prints ONE line
    assert lines[0]["n_gpus"] == 2 and lines[0]["ranks_seen"] == 2 and lines[0]["ok"], lines[0]


def test_gpus_2_under_a_mismatched_world_size_refuses():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], env=_env(WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in r.stderr and not any(x.startswith("{") for x in r.stdout.splitlines())


def test_single_gpu_default_is_untouched():
    r = subprocess.run([sys.executable, BENCH, "--launch-check"], env=_env(), capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1
