"""Active-learning selection config: RandomSelector on the CBGS FPNVoxelNet (same keys as the
reference's examples/active/cbgs_random.py; see _cbgs_common.py for the shared part)."""
_base_ = "_cbgs_common.py"

selector = dict(
    type="RandomSelector",
    budget=600,
    buffer_file="data/buffers/random.json",
    infos_origin="data/nuScenes/infos_train_10sweeps_withvelo.pkl",
)
