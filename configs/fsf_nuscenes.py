"""Model definition of FSF on nuScenes for the MI355X hot path, in the reference's config dialect
(mmcv python config; same `type=` names and constructor kwargs as
projects/configs/nuScenes/FSF_nuScenes_config.py:33-411 of the reference, which loads unchanged through
`fullysparsefusion_amd.compat.Config` — tests/test_config_surface.py checks both files build the same model).

Only the model is described here; dataset pipelines, schedules and runtime hooks belong to the training
control plane, which is out of scope.
"""
CLASSES = ["car", "truck", "trailer", "bus", "construction_vehicle", "bicycle", "motorcycle", "pedestrian",
           "traffic_cone", "barrier"]
GROUPS = [["car"], ["truck", "construction_vehicle"], ["bus", "trailer"], ["barrier"], ["motorcycle", "bicycle"],
          ["pedestrian", "traffic_cone"]]
NUM_CLASSES = len(CLASSES)
PC_RANGE = [-51.2, -51.2, -5, 51.2, 51.2, 3]
SEG_VOXEL = (0.2, 0.2, 0.2)
SCORE_THRESH = [0.1] * len(GROUPS)
GROUP_LENS = [len(g) for g in GROUPS]

SYNC_BN = dict(type="naiveSyncBN1d", eps=1e-3, momentum=0.01)
LN3 = dict(type="LN", eps=1e-3)


def _l1(w):
    return dict(type="L1Loss", loss_weight=w)


def _sir(first_in):
    return dict(type="SIR", num_blocks=3, in_channels=[first_in, 133, 133], feat_channels=[[128, 128]] * 3,
                rel_mlp_hidden_dims=[[16, 32]] * 3, norm_cfg=LN3, mode="max", xyz_normalizer=[20, 20, 4], act="gelu",
                unique_once=True)


def _cluster_head(head_type, in_channel, train_cfg=None, **extra):
    cfg = dict(
        type=head_type, num_classes=NUM_CLASSES, bbox_coder=dict(type="BasePointBBoxCoder", code_size=10),
        loss_cls=dict(type="FocalLoss", use_sigmoid=True, gamma=4.0, alpha=0.25, loss_weight=1.0),
        loss_center=_l1(0.5), loss_size=_l1(0.5), loss_rot=_l1(0.2), loss_vel=_l1(0.2),
        in_channel=in_channel, shared_mlp_dims=[1024, 1024], train_cfg=train_cfg, test_cfg=None, norm_cfg=dict(type="LN"),
        tasks=[dict(num_class=NUM_CLASSES, class_names=CLASSES)], class_names=CLASSES,
        common_attrs=dict(center=(3, 2, 128), dim=(3, 2, 128), rot=(2, 2, 128), vel=(2, 2, 128)),
        num_cls_layer=2, cls_hidden_dim=128, separate_head=dict(type="FSDSeparateHead", norm_cfg=dict(type="LN"), act="gelu"))
    cfg.update(extra)
    return cfg


_HEAD_TEST_CFG = dict(use_rotate_nms=True, nms_pre=-1, nms_thr=0.35, score_thr=0.01, min_bbox_size=0, max_num=500)
_sample_cfg = dict(score_thresh=SCORE_THRESH, pre_voxelization_size=(0.1, 0.1, 0.1), group_sample=True, offset_weight="max",
                   group_lens=GROUP_LENS, class_names=CLASSES, group_names=GROUPS)

segmentor = dict(
    type="VoteSegmentor",
    tanh_dims=[],
    voxel_layer=dict(voxel_size=SEG_VOXEL, max_num_points=-1, point_cloud_range=PC_RANGE, max_voxels=(-1, -1)),
    voxel_encoder=dict(type="DynamicScatterVFE", in_channels=5, feat_channels=[64, 64], voxel_size=SEG_VOXEL,
                       with_cluster_center=True, with_voxel_center=True, point_cloud_range=PC_RANGE, norm_cfg=SYNC_BN,
                       unique_once=True),
    middle_encoder=dict(type="PseudoMiddleEncoderForSpconvFSD"),
    backbone=dict(
        type="SimpleSparseUNet", in_channels=64, sparse_shape=[40, 512, 512], order=("conv", "norm", "act"), norm_cfg=SYNC_BN,
        base_channels=64, output_channels=128,
        encoder_channels=((128,), (128, 128, 128), (128, 128, 128), (256, 256, 256), (512, 512, 512)),
        encoder_paddings=((1,), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1), (1, 1, 1)),
        decoder_channels=((512, 512, 256), (256, 256, 128), (128, 128, 128), (128, 128, 128), (128, 128, 128)),
        decoder_paddings=((1, 1), (1, 0), (1, 0), (0, 0), (0, 1))),
    decode_neck=dict(type="Voxel2PointScatterNeck", voxel_size=SEG_VOXEL, point_cloud_range=PC_RANGE),
    segmentation_head=dict(
        type="VoteSegHead", in_channel=67 + 64, hidden_dims=[128, 128], num_classes=NUM_CLASSES, dropout_ratio=0.0,
        conv_cfg=dict(type="Conv1d"), norm_cfg=dict(type="naiveSyncBN1d"), act_cfg=dict(type="ReLU"),
        loss_decode=dict(type="CrossEntropyLoss", use_sigmoid=False, class_weight=[1.0] * NUM_CLASSES + [0.1], loss_weight=10.0),
        loss_vote=_l1(1.0)),
    train_cfg=dict(point_loss=True, score_thresh=SCORE_THRESH, class_names=CLASSES, group_names=GROUPS, group_lens=GROUP_LENS),
)

model = dict(
    type="FSF",
    num_classes=NUM_CLASSES,
    num_cams=6,
    class_names=CLASSES,
    # LiDAR query generation
    segmentor=segmentor,
    backbone=_sir(116 + 64),
    bbox_head=_cluster_head("SparseClusterHeadV2", 128 * 3 * 2),
    train_cfg=dict(sync_reg_avg_factor=True, disable_pretrain=False, disable_pretrain_topks=[200] * 6, **_sample_cfg),
    test_cfg=dict(use_rotate_nms=True, nms_pre=-1, nms_thr=0.25, score_thr=0.05, min_bbox_size=0, max_num=500, **_sample_cfg),
    cluster_assigner=dict(
        cluster_voxel_size=[(0.3, 0.3, 8), (0.3, 0.3, 8), (0.3, 0.3, 8), (0.1, 0.1, 8), (0.2, 0.2, 8), (0.05, 0.05, 8)],
        min_points=2, point_cloud_range=PC_RANGE, connected_dist=[0.6, 0.6, 0.6, 0.2, 0.4, 0.1], class_names=CLASSES),
    # camera query generation
    frustum_sir=_sir(67 + 64 + 5),
    frustum_obj_head=_cluster_head("FrustumClusterHead", 128 * 3 * 2 + 128, train_cfg=dict(), test_cfg=_HEAD_TEST_CFG,
                                   as_rpn=False),
    encode_2d_mlp_cfg=dict(in_channel=16, mlp_channel=[128, 128], norm_cfg=LN3, act="gelu"),
    segmentor_updated_mlp=dict(in_channel=10, mlp_channel=[128, 67 + 64], norm_cfg=LN3, act="gelu"),
    mlp_cfg=dict(embed_dims=1024, norm_cfg=LN3, act="gelu", lidar_img_input_dim=128 * 3 * 2 + 128, lidar_input_dim=128 * 3 * 2),
    # query refinement: RoI point pooling (K17) -> 3 SIR layers per RoI -> fused query -> head -> BEV NMS (K20)
    bbox_coder=dict(type="BasePointBBoxCoder", code_size=10),
    roi_extractor=dict(type="DynamicPointROIExtractor", extra_wlh=[1.0, 1.0, 1.0], max_inbox_point=512, debug=False),
    single_refine_sir_layer=dict(
        type="FullySparseBboxHead", num_classes=NUM_CLASSES, num_blocks=3,
        in_channels=[67 + 5 + 13 + 32 + 64, 131 + 13 + 2, 131 + 13 + 2], feat_channels=[[128, 128]] * 3, with_distance=False,
        with_cluster_center=False, with_rel_mlp=True, rel_mlp_hidden_dims=[[16, 32]] * 3, rel_mlp_in_channels=[13] * 3,
        reg_mlp=[512, 512], cls_mlp=[512, 512], mode="max", xyz_normalizer=[20, 20, 4], cat_voxel_feats=True, pos_fusion="mul",
        fusion="cat", act="gelu", geo_input=True, use_middle_cluster_feature=True, norm_cfg=LN3, unique_once=True),
    refined_obj_head=[
        _cluster_head("FrustumClusterHead", 1024, test_cfg=_HEAD_TEST_CFG, as_rpn=False,
                      loss_cls=dict(type="FocalLoss", use_sigmoid=True, gamma=4.0, alpha=0.25, loss_weight=2.0)),
    ],
    refine_encode_2d_mlp_cfg=dict(in_channel=10, mlp_channel=[32, 32], norm_cfg=LN3, act="gelu"),
)
