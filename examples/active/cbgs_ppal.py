"""Active-learning selection config: PPALSelector on the CBGS FPNVoxelNet (same keys as the
reference's examples/active/cbgs_ppal.py; see _cbgs_common.py for the shared part)."""
_base_ = "_cbgs_common.py"

selector = dict(
    type="PPALSelector",
    budget=4800,
    buffer_file="data/buffers/ppal.json",
    infos_origin="data/nuScenes/infos_train_10sweeps_withvelo.pkl",
)
