"""Active-learning selection config: EntropySelector on the CBGS FPNVoxelNet (same keys as the
reference's examples/active/cbgs_entropy.py; see _cbgs_common.py for the shared part)."""
_base_ = "_cbgs_common.py"

selector = dict(
    type="EntropySelector",
    budget=4800,
    buffer_file="data/buffers/entropy.json",
    infos_origin="data/nuScenes/infos_train_10sweeps_withvelo.pkl",
)
