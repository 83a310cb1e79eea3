"""Active-learning selection config: EuSpatialSelector on the CBGS FPNVoxelNet (same keys as the
reference's examples/active/cbgs_euclidean_spatial.py; see _cbgs_common.py for the shared part)."""
_base_ = "_cbgs_common.py"

selector = dict(
    type="EuSpatialSelector",
    budget=600,
    buffer_file="data/buffers/euclidean_spatial.json",
    infos_origin="data/nuScenes/infos_train_10sweeps_withvelo.pkl",
)
