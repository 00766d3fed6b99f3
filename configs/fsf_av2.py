"""Model definition of FSF on Argoverse 2 (long range: +-204.8 m, 2048 x 2048 x 32 voxel grid, 7 ring cameras, one int32
instance-id plane per camera, 26 classes, 4-d points, boxes without velocity) in the reference's config dialect — the
model part of projects/configs/Argoverse2/FSF_AV2_config.py:10-425 (which also loads unchanged through
`fullysparsefusion_amd.compat.Config`; tests/test_config_surface.py checks both files build the same parameters).
Dataset pipelines / schedules / hooks and the train-time assigners are not described here."""
CLASSES = ["Regular_vehicle", "Pedestrian", "Bicyclist", "Motorcyclist", "Wheeled_rider", "Bollard", "Construction_cone", "Sign",
           "Construction_barrel", "Stop_sign", "Mobile_pedestrian_crossing_sign", "Large_vehicle", "Bus", "Box_truck", "Truck",
           "Vehicular_trailer", "Truck_cab", "School_bus", "Articulated_bus", "Message_board_trailer", "Bicycle", "Motorcycle",
           "Wheeled_device", "Wheelchair", "Stroller", "Dog"]
GROUPS = [CLASSES[:1], CLASSES[1:5], CLASSES[5:11], CLASSES[11:20], CLASSES[20:25], CLASSES[25:]]
NUM_CLASSES = len(CLASSES)
NUM_CAMS = 7
PC_RANGE = [-204.8, -204.8, -3.2, 204.8, 204.8, 3.2]
SEG_VOXEL = (0.2, 0.2, 0.2)
SCORE_THRESH = [0.4, 0.25, 0.25, 0.25, 0.25, 0.25]
GROUP_LENS = [len(g) for g in GROUPS]
TASKS = [dict(class_names=CLASSES)]

SYNC_BN = dict(type="naiveSyncBN1d", eps=1e-3, momentum=0.01)
LN3 = dict(type="LN", eps=1e-3)
_HEAD_TEST_CFG = dict(use_rotate_nms=True, nms_pre=-1, nms_thr=0.35, score_thr=0.01, min_bbox_size=0, max_num=500)


def _sl1(w):
    return dict(type="SmoothL1Loss", loss_weight=w, beta=0.1)


def _sir(first_in):
    return dict(type="SIR", num_blocks=3, in_channels=[first_in, 132, 132], feat_channels=[[128, 128]] * 3,
                rel_mlp_hidden_dims=[[16, 32]] * 3, norm_cfg=LN3, mode="max", xyz_normalizer=[20, 20, 4], act="gelu",
                unique_once=True)


def _cluster_head(head_type, in_channel, **extra):
    cfg = dict(
        type=head_type, num_classes=NUM_CLASSES, bbox_coder=dict(type="BasePointBBoxCoder", code_size=8),
        loss_cls=dict(type="FocalLoss", use_sigmoid=True, gamma=1.0, alpha=0.25, loss_weight=4.0),
        loss_center=_sl1(0.25), loss_size=_sl1(0.25), loss_rot=_sl1(0.1), in_channel=in_channel, shared_mlp_dims=[1024, 1024],
        norm_cfg=dict(type="LN"), tasks=TASKS, class_names=CLASSES,
        common_attrs=dict(center=(3, 2, 128), dim=(3, 2, 128), rot=(2, 2, 128)), num_cls_layer=2, cls_hidden_dim=128,
        separate_head=dict(type="FSDSeparateHead", norm_cfg=dict(type="LN"), act="gelu"))
    cfg.update(extra)
    return cfg


_sample_cfg = dict(score_thresh=SCORE_THRESH, class_names=CLASSES, pre_voxelization_size=(0.1, 0.1, 0.1), group_sample=True,
                   group_names=GROUPS, offset_weight="max", group_lens=GROUP_LENS)

segmentor = dict(
    type="VoteSegmentor",
    tanh_dims=[],
    voxel_layer=dict(voxel_size=SEG_VOXEL, max_num_points=-1, point_cloud_range=PC_RANGE, max_voxels=(-1, -1)),
    voxel_encoder=dict(type="DynamicScatterVFE", in_channels=4, feat_channels=[64, 64], voxel_size=SEG_VOXEL,
                       with_cluster_center=True, with_voxel_center=True, point_cloud_range=PC_RANGE, norm_cfg=SYNC_BN),
    middle_encoder=dict(type="PseudoMiddleEncoderForSpconvFSD"),
    backbone=dict(
        type="SimpleSparseUNet", in_channels=64, sparse_shape=[32, 2048, 2048], order=("conv", "norm", "act"), norm_cfg=SYNC_BN,
        base_channels=64, output_channels=128,
        encoder_channels=((64,), (64, 64, 64), (64, 64, 64), (128, 128, 128)),
        encoder_paddings=((1,), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1)),
        decoder_channels=((128, 128, 64), (64, 64, 64), (64, 64, 64), (64, 64, 64)),
        decoder_paddings=((1, 0), (1, 0), (0, 0), (0, 1))),
    decode_neck=dict(type="Voxel2PointScatterNeck", voxel_size=SEG_VOXEL, point_cloud_range=PC_RANGE),
    segmentation_head=dict(
        type="VoteSegHead", in_channel=67, hidden_dims=[128, 128], num_classes=NUM_CLASSES, dropout_ratio=0.0,
        conv_cfg=dict(type="Conv1d"), norm_cfg=dict(type="naiveSyncBN1d"), act_cfg=dict(type="ReLU"),
        loss_decode=dict(type="CrossEntropyLoss", use_sigmoid=False, class_weight=[1.0] * NUM_CLASSES + [0.1], loss_weight=3.0),
        loss_vote=dict(type="L1Loss", loss_weight=1.0)),
    train_cfg=dict(point_loss=True, score_thresh=SCORE_THRESH, class_names=CLASSES, group_names=GROUPS, group_lens=GROUP_LENS),
    test_cfg=dict(point_loss=True, score_thresh=(0.5, 0.2, 0.2), clustering_voxel_size=(0.5, 0.5, 6)),
)

model = dict(
    type="FSF",
    num_classes=NUM_CLASSES,
    num_cams=NUM_CAMS,
    class_names=CLASSES,
    is_argo=True,
    # LiDAR query generation
    segmentor=segmentor,
    segmentor_updated_mlp=dict(in_channel=32, mlp_channel=[128, 67], norm_cfg=LN3, act="gelu"),
    backbone=_sir(243 - 64),
    bbox_head=_cluster_head("SparseClusterHeadV2", 128 * 3 * 2, train_cfg=None, test_cfg=None),
    encode_2d_mlp_cfg=dict(in_channel=32, mlp_channel=[128, 128], norm_cfg=LN3, act="gelu"),
    train_cfg=dict(sync_reg_avg_factor=True, **_sample_cfg),
    test_cfg=dict(use_rotate_nms=True, nms_pre=-1, nms_thr=0.25, score_thr=0.1, min_bbox_size=0, max_num=500, **_sample_cfg),
    cluster_assigner=dict(
        cluster_voxel_size=[(0.3, 0.3, 6.4), (0.05, 0.05, 6.4), (0.08, 0.08, 6.4), (0.5, 0.5, 6.4), (0.1, 0.1, 6.4), (0.08, 0.08, 6.4)],
        min_points=2, point_cloud_range=PC_RANGE, connected_dist=[0.6, 0.1, 0.15, 1.0, 0.2, 0.15], class_names=CLASSES),
    # camera query generation
    frustum_sir=_sir(71),
    frustum_obj_head=_cluster_head("FrustumClusterHead", 128 * 3 * 2 + 128, train_cfg=dict(), test_cfg=_HEAD_TEST_CFG, as_rpn=False),
    # query refinement
    mlp_cfg=dict(embed_dims=1024, norm_cfg=LN3, act="gelu", lidar_img_input_dim=128 * 3 * 2 + 128, lidar_input_dim=128 * 3 * 2),
    bbox_coder=dict(type="BasePointBBoxCoder", code_size=8),
    roi_extractor=dict(type="DynamicPointROIExtractor", extra_wlh=[1.0, 1.0, 1.0], max_inbox_point=512, debug=False),
    single_refine_sir_layer=dict(
        type="FullySparseBboxHead", num_classes=NUM_CLASSES, num_blocks=3, in_channels=[67 + 32 + 4 + 13, 130 + 13 + 2, 130 + 13 + 2],
        feat_channels=[[128, 128]] * 3, with_distance=False, with_cluster_center=False, with_rel_mlp=True,
        rel_mlp_hidden_dims=[[16, 32]] * 3, rel_mlp_in_channels=[13] * 3, reg_mlp=[512, 512], cls_mlp=[512, 512], mode="max",
        xyz_normalizer=[20, 20, 4], cat_voxel_feats=True, pos_fusion="mul", fusion="cat", act="gelu", geo_input=True,
        use_middle_cluster_feature=True, norm_cfg=LN3, unique_once=True),
    refined_obj_head=[_cluster_head("FrustumClusterHead", 1024, test_cfg=_HEAD_TEST_CFG, as_rpn=False)],
    refine_encode_2d_mlp_cfg=dict(in_channel=32, mlp_channel=[32, 32], norm_cfg=LN3, act="gelu"),
)
