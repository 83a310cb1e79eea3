"""Active-learning selection config: TemporalSelector on the CBGS FPNVoxelNet (same keys as the
reference's examples/active/cbgs_temporal.py; see _cbgs_common.py for the shared part)."""
_base_ = "_cbgs_common.py"

selector = dict(
    type="TemporalSelector",
    budget=600,
    buffer_file="data/buffers/temporal.json",
    infos_origin="data/nuScenes/infos_train_10sweeps_withvelo.pkl",
)
