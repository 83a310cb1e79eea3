"""Active-learning selection config: SpatialTemporalFeatureSelector on the CBGS FPNVoxelNet (same keys as the
reference's examples/active/cbgs_spatial_temporal_feature.py; see _cbgs_common.py for the shared part)."""
_base_ = "_cbgs_common.py"

selector = dict(
    type="SpatialTemporalFeatureSelector",
    budget=600,
    buffer_file="data/buffers/spatial_temporal_feature.json",
    infos_origin="data/nuScenes/infos_train_10sweeps_withvelo.pkl",
)
