"""BEVFusion camera+lidar detector (swint_v0p075 convfuser + TransFusionHead) under the EntropySelector (BASELINE
configs[4] with its detection half; head settings of bevfusion/configs/nuscenes/det/transfusion/default.yaml)."""
_base_ = "bevfusion_camera_lidar_spatial_temporal_feature.py"

model = dict(
    bbox_head=dict(
        _delete_=True,
        type="TransFusionHead", num_proposals=200, auxiliary=True, in_channels=512, hidden_channel=128, num_classes=10,
        num_decoder_layers=1, num_heads=8, nms_kernel_size=3, ffn_channel=256, dropout=0.1, bn_momentum=0.1,
        activation="relu", transpose_input=True,
        common_heads=dict(center=[2, 2], height=[1, 2], dim=[3, 2], rot=[2, 2], vel=[2, 2]),
        test_cfg=dict(dataset="nuScenes", grid_size=[1440, 1440, 41], out_size_factor=8, voxel_size=[0.075, 0.075],
                      pc_range=[-54.0, -54.0], nms_type=None),
        bbox_coder=dict(pc_range=[-54.0, -54.0], post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                        score_threshold=0.0, out_size_factor=8, voxel_size=[0.075, 0.075], code_size=10)),
)

selector = dict(
    type="EntropySelector",
    budget=4800,
    buffer_file="data/buffers/bevfusion_camera_lidar_entropy.json",
    infos_origin="data/nuScenes/infos_train_10sweeps_withvelo.pkl",
)
