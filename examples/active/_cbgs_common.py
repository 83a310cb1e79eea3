"""Shared model/data settings of the CBGS active-learning configs.

Values are those of the reference's ``examples/active/cbgs_*.py`` (model: FPNVoxelNet =
VoxelFeatureExtractorV3 + FPNSpMiddleResNetFHD + RPN + MultiGroupHead; nuScenes 10-sweep,
0.1 m voxels); paths are relative so the configs run out of the box on synthetic pools.
"""
import itertools

from al3d.models.box_coder import build_box_coder

norm_cfg = None

# (class, anchor size w-l-h, anchor z centre, matched / unmatched IoU) in task order
_CLASSES = [
    [("car", [1.97, 4.63, 1.74], -0.95, 0.6, 0.45)],
    [("truck", [2.51, 6.93, 2.84], -0.40, 0.55, 0.4),
     ("construction_vehicle", [2.85, 6.37, 3.19], -0.225, 0.5, 0.35)],
    [("bus", [2.94, 10.5, 3.47], -0.085, 0.55, 0.4), ("trailer", [2.90, 12.29, 3.87], 0.115, 0.5, 0.35)],
    [("barrier", [2.53, 0.50, 0.98], -1.33, 0.55, 0.4)],
    [("motorcycle", [0.77, 2.11, 1.47], -1.085, 0.5, 0.3), ("bicycle", [0.60, 1.70, 1.28], -1.18, 0.5, 0.35)],
    [("pedestrian", [0.67, 0.73, 1.77], -0.935, 0.6, 0.4), ("traffic_cone", [0.41, 0.41, 1.07], -1.285, 0.6, 0.4)],
]
tasks = [dict(num_class=len(g), class_names=[c[0] for c in g]) for g in _CLASSES]
class_names = list(itertools.chain(*[t["class_names"] for t in tasks]))

target_assigner = dict(
    type="iou",
    anchor_generators=[
        dict(type="anchor_generator_range", sizes=size,
             anchor_ranges=[-51.2, -51.2, z, 51.2, 51.2, z], rotations=[0, 1.57], velocities=[0, 0],
             matched_threshold=mt, unmatched_threshold=ut, class_name=name)
        for name, size, z, mt, ut in itertools.chain(*_CLASSES)
    ],
    sample_positive_fraction=-1, sample_size=512, region_similarity_calculator=dict(type="nearest_iou_similarity"),
    pos_area_threshold=-1, tasks=tasks,
)

box_coder = dict(type="ground_box3d_coder", n_dim=9, linear_dim=False, encode_angle_vector=True)

model = dict(
    type="FPNVoxelNet",
    pretrained=None,
    reader=dict(type="VoxelFeatureExtractorV3", num_input_features=5, norm_cfg=norm_cfg),
    backbone=dict(type="FPNSpMiddleResNetFHD", num_input_features=5, ds_factor=8, norm_cfg=norm_cfg),
    neck=dict(type="RPN", layer_nums=[5, 5], ds_layer_strides=[1, 2], ds_num_filters=[128, 256],
              us_layer_strides=[1, 2], us_num_filters=[256, 256], num_input_features=256,
              norm_cfg=norm_cfg),
    bbox_head=dict(type="MultiGroupHead", mode="3d", in_channels=sum([256, 256]), norm_cfg=norm_cfg,
                   tasks=tasks, weights=[1], box_coder=build_box_coder(box_coder),
                   encode_background_as_zeros=True, use_sigmoid_score=True,
                   encode_rad_error_by_sin=False, loss_aux=None),
)

test_cfg = dict(
    nms=dict(use_rotate_nms=True, use_multi_class_nms=False, nms_pre_max_size=1000,
             nms_post_max_size=83, nms_iou_threshold=0.2),
    score_threshold=0.1,
    post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
    max_per_img=500,
)

dataset_type = "NuScenesDataset"
nsweeps = 10
data_root = "data/nuScenes"
voxel_generator = dict(range=[-51.2, -51.2, -5.0, 51.2, 51.2, 3.0], voxel_size=[0.1, 0.1, 0.2],
                       max_points_in_voxel=10, max_voxel_num=60000)
assigner = dict(box_coder=box_coder, target_assigner=target_assigner, out_size_factor=8, debug=False)
val_preprocessor = dict(mode="val", shuffle_points=False, remove_environment=False,
                        remove_unknown_examples=False)
test_pipeline = [
    dict(type="LoadPointCloudFromFile", dataset=dataset_type),
    dict(type="LoadPointCloudAnnotations", with_bbox=True),
    dict(type="Preprocess", cfg=val_preprocessor),
    dict(type="Voxelization", cfg=voxel_generator),
    dict(type="AssignTarget", cfg=assigner),
    dict(type="Reformat"),
]
train_anno = "data/nuScenes/infos_train_10sweeps_withvelo.pkl"
val_anno = "data/nuScenes/infos_val_10sweeps_withvelo.pkl"
data = dict(samples_per_gpu=4, workers_per_gpu=4,
            val=dict(type=dataset_type, root_path=data_root, info_path=val_anno, test_mode=True,
                     nsweeps=nsweeps, class_names=class_names, pipeline=test_pipeline))
log_level = "INFO"
work_dir = "work_dirs/cbgs_active"
