"""BEVFusion camera+lidar (swint_v0p075, convfuser) as the embedding model of the spatial-temporal-feature selector
(BASELINE configs[4]): Swin-T -> GeneralizedLSSFPN -> DepthLSSTransform on six 256 x 704 cameras, the voxelnet_0p075 lidar
encoder, ConvFuser, SECOND / SECONDFPN decoder, fused-BEV embedding.  Component settings are those of the reference's
bevfusion/configs/nuscenes/det/transfusion/secfpn/camera+lidar/{default.yaml, swint_v0p075/default.yaml, swint_v0p075/convfuser.yaml}
expressed on this build's modules; ``model.lidar`` is the lidar-only embedding model of bevfusion_lidar_spatial_temporal_feature.py.
No detection head: the sweep produces the [N,512] fused-BEV embeddings only (bevfusion_camera_lidar_entropy.py adds the head)."""
_base_ = "bevfusion_lidar_spatial_temporal_feature.py"

camera = dict(image_size=[256, 704], feature_size=[32, 88], num_cameras=6)



def _lidar_model():
    """The lidar-only embedding model of the base config (encoder + SECOND / SECONDFPN, no head) as a plain dict."""
    import copy
    import os
    from al3d.utils import Config
    base = Config.fromfile(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bevfusion_lidar_spatial_temporal_feature.py"))
    return copy.deepcopy(dict(base._cfg_dict["model"]))


model = dict(
    _delete_=True,
    type="BEVFusion",
    lidar=_lidar_model(),
    image_size=[256, 704], feature_size=[32, 88],
    xbound=[-54.0, 54.0, 0.3], ybound=[-54.0, 54.0, 0.3], zbound=[-10.0, 10.0, 20.0], dbound=[1.0, 60.0, 0.5],
    camera_channels=80, lidar_channels=256,
    bbox_head=None,
)

del _lidar_model

selector = dict(
    type="SpatialTemporalFeatureSelector",
    budget=4800,
    buffer_file="data/buffers/bevfusion_camera_lidar_stf.json",
    infos_origin="data/nuScenes/infos_train_10sweeps_withvelo.pkl",
)
