"""CPU suite: the launcher path.  ``python -m torch.distributed.run`` (what the driver uses for
bench.py --gpus N and what tools/active_select.py documents) starts two gloo ranks that run the
sharded-sweep control flow with a stub detector (tests/dist_sweep_worker.py)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_torchrun_two_ranks_sharded_sweep_control_flow(tmp_path):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(HERE, "dist_sweep_worker.py"), str(tmp_path)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    recs = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(2)]
    assert [x["rows"] for x in recs] == [6, 5]
    for x in recs:
        assert x["world"] == 2 and x["ok"], x          # gathered == single-process tensor, dataset order
        assert x["raised"], x                          # the inf in rank 1's shard raised on BOTH ranks
    feats = np.load(tmp_path / "feats.npy")            # written once, by rank 0, atomically
    assert feats.shape == (11, 6) and not list(tmp_path.glob("*.tmp*"))
