"""Worker of tests/test_torchrun_gloo.py -- launched by ``python -m torch.distributed.run`` with two
ranks on CPU (gloo).  Walks the control flow of tools/active_select.py's sharded sweep
(reference tools/active_select.py:94-163 + det3d/selectors/feature_selector.py:51-85) with a stub
detector: env:// rendezvous -> contiguous shard per rank -> sweep_embeddings -> all-gather in dataset
order -> finiteness check on the gathered tensor (every rank raises together) -> rank-0-only file
writes.  The selection itself needs the GPU library (no CPU fallback) and is covered by the -m gpu
suite."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class StubDetector(torch.nn.Module):
    """detector(example, return_loss=False, estimate=True) -> (preds, middle); middle[-1] is [B,C,H,W]."""

    def forward(self, example, return_loss=True, **kwargs):
        x = example["bev"]
        return [dict(metadata=m) for m in example["metadata"]], [x]


class ShardLoader:
    def __init__(self, bev, indices, batch):
        self.bev, self.indices, self.batch = bev, list(indices), batch
        self.sampler = self.indices
        self.dataset = self.indices

    def __iter__(self):
        for s in range(0, len(self.indices), self.batch):
            ids = self.indices[s:s + self.batch]
            yield {"bev": self.bev[ids], "metadata": [{"index": i} for i in ids]}


def main():
    out_dir = sys.argv[1]
    dist.init_process_group(backend="gloo", init_method="env://")
    rank, world = dist.get_rank(), dist.get_world_size()
    from al3d.lib import Al3dError
    from al3d.selectors.base_selector import master_only, save_npy_atomic
    from al3d.sweep import sweep_embeddings
    n, c = 11, 6                                    # uneven shards: 6 + 5 frames
    g = torch.Generator().manual_seed(5)
    bev = torch.randn(n, c, 4, 3, generator=g)     # the same pool on every rank
    per = (n + world - 1) // world
    mine = list(range(rank * per, min(n, (rank + 1) * per)))
    det = StubDetector().eval()
    feats = sweep_embeddings(det, ShardLoader(bev, mine, 4), "cpu", num_frames=n)
    want = bev.mean(-1).mean(-1)
    ok = bool(torch.equal(feats, want))
    # one rank's shard holds a non-finite activation: every rank must raise, none may hang in the collective
    bad = bev.clone()
    bad[n - 1, 0, 0, 0] = float("inf")             # frame n-1 belongs to the last rank
    raised = False
    try:
        sweep_embeddings(det, ShardLoader(bad, mine, 4), "cpu", num_frames=n)
    except Al3dError as e:
        raised = "AL3D_MATH=bf16x6" in str(e)
    dist.barrier()                                  # reached by every rank: nobody was left inside all_gather

    @master_only
    def write(path):
        save_npy_atomic(path, feats.numpy())
    write(os.path.join(out_dir, "feats"))           # np.save naming rule: ".npy" appended
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump({"rank": rank, "world": world, "ok": ok, "raised": raised, "rows": len(mine)}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
