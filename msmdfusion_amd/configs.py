"""The hot-path sections of the two target model configs, restated.

Reference: configs/MSMDFusion_nusc_voxel_LC.py:141-190 and
configs/transfusion_nusc_voxel_L.py:150-169 (values are facts; equality with the
reference dicts is pinned by tests/golden/reference_configs.json).  Only the
keys this path consumes are kept (plus the dense BEV backbone/neck of row f1);
the image backbone and the detection head are not built here.

    from msmdfusion_amd.configs import MSMDFUSION_LC, build_hot_path
    vox, vfe, enc, mm = build_hot_path(MSMDFUSION_LC)
"""
POINT_CLOUD_RANGE = [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0]
VOXEL_SIZE = [0.075, 0.075, 0.2]

_PTS_VOXEL_LAYER = dict(max_num_points=10, voxel_size=VOXEL_SIZE, max_voxels=(120000, 160000),
                        point_cloud_range=POINT_CLOUD_RANGE)
_PTS_VOXEL_ENCODER = dict(type="HardSimpleVFE", num_features=5)
_PTS_MIDDLE_ENCODER = dict(
    type="SparseEncoder", in_channels=5, sparse_shape=[41, 1440, 1440], output_channels=128,
    order=("conv", "norm", "act"),
    encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
    encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, [0, 1, 1]), (0, 0)),
    block_type="basicblock")

_BN2D = dict(type="BN", eps=0.001, momentum=0.01)
_PTS_BACKBONE = dict(type="SECOND", in_channels=256, out_channels=[128, 256], layer_nums=[5, 5],
                     layer_strides=[1, 2], norm_cfg=_BN2D,
                     conv_cfg=dict(type="Conv2d", bias=False))
_PTS_NECK = dict(type="SECONDFPN", in_channels=[128, 256], out_channels=[256, 256],
                 upsample_strides=[1, 2], norm_cfg=_BN2D,
                 upsample_cfg=dict(type="deconv", bias=False), use_conv_for_no_stride=True)

_OUT_SIZE_FACTOR = 8
_PTS_BBOX_HEAD = dict(
    type="TransFusionHead", num_proposals=200, auxiliary=True, in_channels=256 * 2,
    hidden_channel=128, num_classes=10, num_decoder_layers=1, num_heads=8,
    learnable_query_pos=False, initialize_by_heatmap=True, nms_kernel_size=3, ffn_channel=256,
    dropout=0.1, bn_momentum=0.1, activation="relu",
    common_heads=dict(center=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2)),
    bbox_coder=dict(type="TransFusionBBoxCoder", pc_range=POINT_CLOUD_RANGE[:2],
                    voxel_size=VOXEL_SIZE[:2], out_size_factor=_OUT_SIZE_FACTOR,
                    post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                    score_threshold=0.0, code_size=10),
    loss_cls=dict(type="FocalLoss", use_sigmoid=True, gamma=2, alpha=0.25, reduction="mean",
                  loss_weight=1.0),
    loss_bbox=dict(type="L1Loss", reduction="mean", loss_weight=0.25),
    loss_heatmap=dict(type="GaussianFocalLoss", reduction="mean", loss_weight=1.0))
# train_cfg.pts of configs/MSMDFusion_nusc_voxel_LC.py:242-259
_TRAIN_CFG_PTS = dict(
    dataset="nuScenes",
    assigner=dict(type="HungarianAssigner3D",
                  iou_calculator=dict(type="BboxOverlaps3D", coordinate="lidar"),
                  cls_cost=dict(type="FocalLossCost", gamma=2, alpha=0.25, weight=0.15),
                  reg_cost=dict(type="BBoxBEVL1Cost", weight=0.25),
                  iou_cost=dict(type="IoU3DCost", weight=0.25)),
    pos_weight=-1, gaussian_overlap=0.1, min_radius=2, grid_size=[1440, 1440, 40],
    voxel_size=VOXEL_SIZE, out_size_factor=_OUT_SIZE_FACTOR,
    code_weights=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2],
    point_cloud_range=POINT_CLOUD_RANGE)
_TEST_CFG_PTS = dict(dataset="nuScenes", grid_size=[1440, 1440, 40],
                     out_size_factor=_OUT_SIZE_FACTOR, pc_range=POINT_CLOUD_RANGE[0:2],
                     voxel_size=VOXEL_SIZE[:2], nms_type=None)

TRANSFUSION_L = dict(
    model=dict(type="TransFusionDetector", pts_voxel_layer=_PTS_VOXEL_LAYER,
               pts_voxel_encoder=_PTS_VOXEL_ENCODER, pts_middle_encoder=_PTS_MIDDLE_ENCODER,
               pts_backbone=_PTS_BACKBONE, pts_neck=_PTS_NECK),
    samples_per_gpu=4, point_cloud_range=POINT_CLOUD_RANGE, voxel_size=VOXEL_SIZE,
    optimizer=dict(type="AdamW", lr=0.0002, weight_decay=0.01),
    freeze_lidar_components=False)

MSMDFUSION_LC = dict(
    model=dict(
        type="MSMDFusionDetector",
        spatial_shapes=[[41, 1440, 1440], [21, 720, 720], [11, 360, 360], [5, 180, 180]],
        downscale_factors=[1, 2, 4, 8],
        fps_num_list=[2048] * 4,
        radius_list=[6, 3, 2, 1],
        max_cluster_samples_list=[200, 100, 50, 25],
        dist_thresh_list=[13.3, 6.6, 3.3, 1.6],
        pts_voxel_layer=_PTS_VOXEL_LAYER,
        pts_voxel_encoder=_PTS_VOXEL_ENCODER,
        pts_middle_encoder=_PTS_MIDDLE_ENCODER,
        multimodal_middle_encoder=dict(
            type="SparseMultiModalEncoderPaint", in_channels_3D=(16, 32, 64, 128),
            in_channels_2D=(64, 64, 64, 64), out_channels=(32, 64, 128, 128),
            padding=(1, 1, [0, 1, 1], 0), order=("conv", "norm", "act"),
            norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01)),
        pts_backbone=_PTS_BACKBONE, pts_neck=_PTS_NECK),
    samples_per_gpu=2, point_cloud_range=POINT_CLOUD_RANGE, voxel_size=VOXEL_SIZE,
    optimizer=dict(type="AdamW", lr=0.0001, betas=(0.9, 0.999), weight_decay=0.05,
                   paramwise_cfg=dict(custom_keys={
                       "absolute_pos_embed": dict(decay_mult=0.),
                       "relative_position_bias_table": dict(decay_mult=0.),
                       "norm": dict(decay_mult=0.)})),
    freeze_lidar_components=True)


def build_hot_path(cfg):
    """(Voxelization, voxel encoder, SparseEncoder, multimodal encoder | None) from one of
    the dicts above -- what MSMDFusionDetector.__init__ / MVXTwoStageDetector.__init__ build
    for this path (mmdet3d/models/detectors/mvx_two_stage.py:38-60, MSMDFusion.py:92-104)."""
    from .registry import build_middle_encoder, build_voxel_encoder
    from .voxelize import Voxelization
    m = cfg["model"]
    vox = Voxelization(**m["pts_voxel_layer"])
    vfe = build_voxel_encoder(m["pts_voxel_encoder"])
    enc = build_middle_encoder(m["pts_middle_encoder"])
    mm = build_middle_encoder(m["multimodal_middle_encoder"]) \
        if "multimodal_middle_encoder" in m else None
    return vox, vfe, enc, mm


def build_bev_tail(cfg, compute_dtype=None, rows=True):
    """bev_fusion (SPPModule, built without a config: MSMDFusion.py:130) +
    pts_backbone + pts_neck of one of the dicts above.  rows=True (default): the same
    modules, same parameters and checkpoint keys, computed on channels-last pixel rows by
    the sparse-conv kernels (grid_conv.py: fp32-equivalent, 1.7x MIOpen's fp32 on the SPP
    block); rows=False: MIOpen, optionally under bf16 autocast (compute_dtype)."""
    from .bev import BevTail, SPPModule
    from .registry import build_backbone, build_neck
    m = cfg["model"]
    if not rows:
        return BevTail(SPPModule(), build_backbone(m["pts_backbone"]),
                       build_neck(m["pts_neck"]), compute_dtype=compute_dtype)
    from .grid_conv import SECONDFPNRows, SECONDRows, SPPModuleRows

    def args(d):
        return {k: v for k, v in d.items() if k != "type"}
    return BevTail(SPPModuleRows(), SECONDRows(**args(m["pts_backbone"])),
                   SECONDFPNRows(**args(m["pts_neck"])), compute_dtype=None)


def build_head(cfg=None, rows=False):
    """pts_bbox_head of configs/MSMDFusion_nusc_voxel_LC.py:207-241 with its test_cfg
    (:260-268) and train_cfg (:242-259): TransFusionHead, LiDAR branch (msmdfusion_amd/head.py,
    head_loss.py)."""
    from .head import TransFusionHead
    args = {k: v for k, v in _PTS_BBOX_HEAD.items() if k != "type"}
    return TransFusionHead(test_cfg=dict(_TEST_CFG_PTS), train_cfg=dict(_TRAIN_CFG_PTS), rows=rows,
                           **args)
