"""Active-learning selection config: BadgeSelector on the CBGS FPNVoxelNet (same keys as the
reference's examples/active/cbgs_badge.py; see _cbgs_common.py for the shared part)."""
_base_ = "_cbgs_common.py"

selector = dict(
    type="BadgeSelector",
    budget=4800,
    buffer_file="data/buffers/badge.json",
    infos_origin="data/nuScenes/infos_train_10sweeps_withvelo.pkl",
)
