"""BEVFusion lidar-only branch (voxelnet_0p075) as the embedding model of the diversity selector
(BASELINE configs[3]).  Grid / voxel / channel settings are those of the reference's
bevfusion/configs/nuscenes/det/transfusion/secfpn/lidar/voxelnet_0p075.yaml and
.../secfpn/default.yaml, expressed on this build's det3d-shaped modules (see
al3d/models/bevfusion_compat.py for the layer-by-layer correspondence and the checkpoint converter).
No detection head: the sweep produces the [N,512] BEV embeddings only."""
_base_ = "_cbgs_common.py"

voxel_generator = dict(
    range=[-54.0, -54.0, -5.0, 54.0, 54.0, 3.0],
    voxel_size=[0.075, 0.075, 0.2],
    max_points_in_voxel=10,
    max_voxel_num=160000,           # max_voxels [train, test] = [120000, 160000]: the sweep is inference
)

model = dict(bbox_head=None)        # encoder + SECOND/SECONDFPN only (same modules as the CBGS model)

selector = dict(
    type="SpatialTemporalFeatureSelector",
    budget=4800,
    buffer_file="data/buffers/bevfusion_lidar_stf.json",
    infos_origin="data/nuScenes/infos_train_10sweeps_withvelo.pkl",
)
