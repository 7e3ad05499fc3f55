"""The detectors the two target configs name, assembled from the hot-path modules.

    model = build_detector(MSMDFUSION_LC["model"], train_cfg=..., test_cfg=...)

`type='MSMDFusionDetector'` (configs/MSMDFusion_nusc_voxel_LC.py:141-268) and
`type='TransFusionDetector'` (configs/transfusion_nusc_voxel_L.py:150-169) resolve through the
DETECTORS registry, as mmdet3d.models.build_detector does; child modules are built from the
unchanged config sub-dicts by the registries of msmdfusion_amd.registry and carry the
reference's attribute names, so checkpoint keys (`pts_middle_encoder.*`,
`multimodal_middle_encoder.*`, `bev_fusion.*`, `pts_backbone.*`, `pts_neck.*`,
`pts_bbox_head.*`, `conv1x1_blocks.*`, `score_net.*`) load one to one.

Reference: mmdet3d/models/detectors/MSMDFusion.py (extract_img_feat :140-167,
extract_multiscale_voxel_feat :400-419, extract_pts_feat :421-452, forward_train :494-560,
forward_pts_train :562-590, simple_test :592-640), mmdet3d/models/detectors/transfusion.py
:61-101, mmdet3d/models/detectors/mvx_two_stage.py:38-120, tools/train.py:185-219
(freeze_lidar_components).

Out of scope: the image backbone and neck (frozen ResNet-50 + FPN).  They
are INJECTED: `img_backbone` / `img_neck` may be any nn.Module / callable producing the
multi-scale feature list the reference's FPN yields; without them the detector takes the
per-scale virtual points directly (`virtual_points=`), which is what bench.py feeds it.
"""
import logging
import warnings

import torch
from torch import nn

from . import spconv
from .fusion import SparseFusionPath
from .registry import (DETECTORS, build_backbone, build_middle_encoder, build_neck,
                       build_voxel_encoder)
from .voxelize import Voxelization


def build_detector(cfg, train_cfg=None, test_cfg=None, **injected):
    """mmdet3d.models.build_detector: cfg = the config's `model` dict (its own train_cfg /
    test_cfg keys win, as in mmdet3d).  injected: img_backbone / img_neck modules."""
    cfg = dict(cfg)
    if train_cfg is not None:
        cfg.setdefault("train_cfg", train_cfg)
    if test_cfg is not None:
        cfg.setdefault("test_cfg", test_cfg)
    cfg.update(injected)
    return DETECTORS.build(cfg)


def _build_head(cfg, train_cfg, test_cfg, rows):
    from .head import TransFusionHead
    if isinstance(cfg, nn.Module):
        return cfg
    args = {k: v for k, v in cfg.items() if k != "type"}
    if cfg.get("type", "TransFusionHead") != "TransFusionHead":
        raise KeyError("pts_bbox_head type %r is not built here" % cfg.get("type"))
    pts = lambda c: (c or {}).get("pts", c) if isinstance(c, dict) else c
    return TransFusionHead(train_cfg=pts(train_cfg), test_cfg=pts(test_cfg), rows=rows, **args)


@DETECTORS.register_module()
class TransFusionDetector(nn.Module):
    """LiDAR-only detector (configs/transfusion_nusc_voxel_L.py): voxelize -> HardSimpleVFE ->
    SparseEncoder -> SECOND -> SECONDFPN -> TransFusionHead."""

    def __init__(self, pts_voxel_layer=None, pts_voxel_encoder=None, pts_middle_encoder=None,
                 pts_backbone=None, pts_neck=None, pts_bbox_head=None, img_backbone=None,
                 img_neck=None, train_cfg=None, test_cfg=None, pretrained=None, freeze_img=True,
                 rows=True, **unused):
        super().__init__()
        self.pts_voxel_layer = Voxelization(**pts_voxel_layer)
        self.pts_voxel_encoder = build_voxel_encoder(pts_voxel_encoder)
        self.pts_middle_encoder = build_middle_encoder(pts_middle_encoder)
        self.rows = bool(rows)
        self.pts_backbone = self._dense(pts_backbone, build_backbone, "SECONDRows")
        self.pts_neck = self._dense(pts_neck, build_neck, "SECONDFPNRows")
        self.pts_bbox_head = None if pts_bbox_head is None else \
            _build_head(pts_bbox_head, train_cfg, test_cfg, self.rows)
        self.img_backbone, self.img_neck = img_backbone, img_neck
        self.freeze_img = freeze_img
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.unused_cfg = dict(unused)      # keys of the config this path does not consume
        if freeze_img:
            for m in (img_backbone, img_neck):
                if isinstance(m, nn.Module):
                    for p in m.parameters():
                        p.requires_grad = False

    def _dense(self, cfg, builder, rows_cls):
        """A dense BEV module from its config dict: the torch / MIOpen form, or (rows=True)
        the same parameters and state-dict keys computed on channels-last pixel rows by the
        sparse-conv kernels (grid_conv.py)."""
        if cfg is None or isinstance(cfg, nn.Module):
            return cfg
        if not self.rows:
            return builder(cfg)
        from . import grid_conv
        return getattr(grid_conv, rows_cls)(**{k: v for k, v in cfg.items() if k != "type"})

    # ---- mvx_two_stage.py properties ---------------------------------------------------
    with_pts_bbox = property(lambda self: self.pts_bbox_head is not None)
    with_pts_backbone = property(lambda self: self.pts_backbone is not None)
    with_pts_neck = property(lambda self: self.pts_neck is not None)
    with_img_backbone = property(lambda self: self.img_backbone is not None)
    with_img_neck = property(lambda self: self.img_neck is not None)

    # ---- the path ----------------------------------------------------------------------
    @torch.no_grad()
    def voxelize(self, points):
        """transfusion.py:76-101: hard voxelization per sample, batch id prepended; the
        HardSimpleVFE mean is fused into the gather (voxels are never materialised).
        -> (mean features [M,C], coors [M,4])"""
        feats, coors = [], []
        for b, (mean, c, _) in enumerate(self.pts_voxel_layer.forward_batch(points, fused_mean=True)):
            feats.append(mean)
            coors.append(nn.functional.pad(c, (1, 0), mode="constant", value=b))
        return torch.cat(feats, 0), torch.cat(coors, 0)

    def prepare(self, points):
        """The index-only part of a step (no weights, no previous step): voxelization and
        every rulebook / tiling / pair list of the encoder -- what IndexPrefetcher runs a
        step ahead."""
        feats, coors = self.voxelize(points)
        planned, _ = self.pts_middle_encoder.plan(coors, len(points))
        return feats, coors, planned

    def extract_sparse_feat(self, points, prepared=None):
        feats, coors, planned = prepared if prepared is not None else self.prepare(points)
        bev, _ = self.pts_middle_encoder(feats, coors, len(points), planned=planned)
        return bev

    def extract_pts_feat(self, pts, img_feats=None, img_metas=None, prepared=None):
        """transfusion.py:61-74 -> list of BEV maps (the neck's output)."""
        x = self.extract_sparse_feat(pts, prepared=prepared)
        if self.with_pts_backbone:
            x = self.pts_backbone(x)
        if self.with_pts_neck:
            x = self.pts_neck(x)
        return x if isinstance(x, (list, tuple)) else [x]

    def extract_img_feat(self, img, img_metas):
        """MSMDFusion.py:140-167 around the injected image branch."""
        if self.img_backbone is None or img is None:
            return None
        input_shape = img.shape[-2:]
        for meta in img_metas:
            meta.update(input_shape=input_shape)
        if img.dim() == 5:
            img = img.view(-1, *img.shape[2:])
        feats = self.img_backbone(img.float())
        if self.img_neck is not None:
            feats = self.img_neck(feats)
        return feats

    def extract_feat(self, points, img=None, img_metas=None, **kw):
        img_feats = self.extract_img_feat(img, img_metas)
        return img_feats, self.extract_pts_feat(points, img_feats, img_metas, **kw)

    def forward_pts_train(self, pts_feats, img_feats, gt_bboxes_3d, gt_labels_3d, img_metas=None):
        """MSMDFusion.py:562-590 / transfusion.py:163-190 -> dict of losses."""
        outs = self.pts_bbox_head(pts_feats, img_feats, img_metas)
        return self.pts_bbox_head.loss(gt_bboxes_3d, gt_labels_3d, outs)

    def forward_train(self, points=None, img_metas=None, gt_bboxes_3d=None, gt_labels_3d=None,
                      img=None, **kw):
        """MSMDFusion.py:494-560: features, then the point branch's losses."""
        img_feats, pts_feats = self.extract_feat(points, img=img, img_metas=img_metas, **kw)
        return self.forward_pts_train(pts_feats, img_feats, gt_bboxes_3d, gt_labels_3d, img_metas)

    def simple_test(self, points, img_metas=None, img=None, **kw):
        """MSMDFusion.py:592-640 without NMS (nms_type=None in both configs): per sample a
        dict(boxes_3d, scores_3d, labels_3d)."""
        img_feats, pts_feats = self.extract_feat(points, img=img, img_metas=img_metas, **kw)
        outs = self.pts_bbox_head(pts_feats, img_feats, img_metas)
        # bbox3d2result's fields (mmdet3d/core/bbox/transforms.py)
        return [dict(boxes_3d=r["bboxes"], scores_3d=r["scores"], labels_3d=r["labels"])
                for r in self.pts_bbox_head.get_bboxes(outs)]

    def forward(self, points, *extra, return_loss=False, prepared=None, **kw):
        """return_loss=False (default): the BEV feature list of extract_pts_feat (what the
        throughput benchmark differentiates); True: forward_train's loss dict."""
        if return_loss:
            return self.forward_train(points=points, prepared=prepared, **kw)
        return self.extract_pts_feat(points, *extra, prepared=prepared, **kw)


_WARNED = {}


def _notice_split_mode(reference_quirks):
    """One line per process and mode, when a detector is built from a config: which
    voxel_modality_split the unchanged reference config gets here (logger `msmdfusion_amd`,
    level INFO; the load-time warning above is the loud one)."""
    key = ("mode", bool(reference_quirks))
    if _WARNED.get(key):
        return
    _WARNED[key] = True
    logging.getLogger("msmdfusion_amd").info(
        "MSMDFusionDetector: reference_quirks=%s -- voxel_modality_split uses %s",
        bool(reference_quirks),
        "the reference's float32 keys and non-cumulative batch offsets (bit for bit the "
        "reference, MSMDFusion.py:251-325)" if reference_quirks else
        "exact integer keys and cumulative batch offsets (differs from the reference where "
        "its float32 keys alias: z >= 17 or x >= 1000 on the scale-1 grid; set "
        "reference_quirks=True in the model config to reproduce the reference)")


@DETECTORS.register_module()
class MSMDFusionDetector(TransFusionDetector):
    """LiDAR + camera detector (configs/MSMDFusion_nusc_voxel_LC.py): the LiDAR encoder's
    four scales meet virtual-point voxels from the image branch in the GMA-Conv stack."""

    def __init__(self, spatial_shapes=None, downscale_factors=None, fps_num_list=None,
                 radius_list=None, max_cluster_samples_list=None, dist_thresh_list=None,
                 multimodal_middle_encoder=None, reference_quirks=False, **kwargs):
        """reference_quirks (config key of the same name, default False): True reproduces
        the reference's float32-key voxel_modality_split and its batch-offset arithmetic bit
        for bit -- the mode that matches a checkpoint trained with the reference
        (fusion.SparseFusionPath)."""
        super().__init__(**kwargs)
        self.reference_quirks = bool(reference_quirks)
        _notice_split_mode(self.reference_quirks)
        from .bev import SPPModule
        from .image_glue import DepthAwareChannelCompression, ScoreNet
        self.spatial_shapes = [list(s) for s in spatial_shapes]
        self.downscale_factors = list(downscale_factors)
        self.fps_num_list = list(fps_num_list)
        self.radius_list = list(radius_list)
        self.max_cluster_samples_list = list(max_cluster_samples_list)
        self.dist_thresh_list = list(dist_thresh_list)
        self.multimodal_middle_encoder = build_middle_encoder(multimodal_middle_encoder)
        # channel compression for ResNet-50 (MSMDFusion.py:106-128): the ModuleList itself is
        # the attribute, as in the reference (keys `conv1x1_blocks.0.0.weight` ...)
        compress = DepthAwareChannelCompression()
        self.conv1x1_blocks = compress.conv1x1_blocks
        object.__setattr__(self, "_compress", compress)      # (unregistered: one set of keys)
        self.score_net = ScoreNet()
        if self.rows:
            from .grid_conv import SPPModuleRows
            self.bev_fusion = SPPModuleRows()
        else:
            self.bev_fusion = SPPModule()
        # one object owns the sparse section's schedule (index pass first, neighbour search on
        # side streams); it holds the SAME child modules, unregistered, to keep one set of keys
        object.__setattr__(self, "_path", SparseFusionPath(
            self.pts_voxel_layer, self.pts_middle_encoder, self.multimodal_middle_encoder,
            self.spatial_shapes, self.downscale_factors, self.fps_num_list, self.radius_list,
            self.max_cluster_samples_list, self.dist_thresh_list,
            base_voxel_size=self.pts_voxel_layer.voxel_size,
            reference_quirks=self.reference_quirks))

    def load_state_dict(self, state_dict, *args, **kwargs):
        """A checkpoint trained with the reference has seen the float32-key split's false
        'mixed' voxels on every frame (MSMDFusion.py:271-272); loading weights into a detector
        built with reference_quirks=False (the exact-key split) runs them on a different voxel
        partition than they were trained with.  Said once per process, at load time."""
        if not self.reference_quirks and not _WARNED.get("load"):
            _WARNED["load"] = True
            warnings.warn(
                "MSMDFusionDetector.load_state_dict: reference_quirks=False -- "
                "voxel_modality_split keys voxels exactly.  A checkpoint trained with the "
                "reference (float32 keys: false 'mixed' voxels wherever keys alias, on every "
                "nuScenes frame at the 0.075 m scale) was trained on a different partition; "
                "build the detector with reference_quirks=True to reproduce it "
                "(INTEGRATION.md, 'Which mode reproduces a published checkpoint').",
                stacklevel=2)
        return super().load_state_dict(state_dict, *args, **kwargs)

    # ---- image side --------------------------------------------------------------------
    def virtual_points_from_images(self, img_feats, img_metas):
        """depth_aware_channel_compression + get_foreground2D per scale
        (MSMDFusion.py:400-407, 169-238) -> 4 lists of B [n, 15 + 49] tensors."""
        from .image_glue import get_foreground2D, pack_foreground, raise_if_any_bad
        pack = pack_foreground(img_metas, img_feats[0].device)
        bad = []       # out-of-map pixel counters of the five kernels: ONE host read below
        comp = self._compress(img_feats, img_metas, pack=pack, check=bad)
        per_scale = [comp[0]] + list(comp)               # img_feat_list[0] twice, :404-405
        out = [get_foreground2D(f, img_metas, self.score_net, pack=pack, check=bad,
                                reference_quirks=self.reference_quirks) for f in per_scale[:4]]
        raise_if_any_bad(bad)
        return out

    # ---- the path ----------------------------------------------------------------------
    def prepare(self, points, virtual_points, nn_side_stream=True):
        vps = virtual_points if isinstance(virtual_points[0], (list, tuple)) else [virtual_points] * 4
        return self._path.prepare(points, vps, nn_side_stream=nn_side_stream)

    def extract_sparse_feat(self, points, virtual_points, prepared=None):
        """MSMDFusion.py:421-440 up to bev_fusion's input: cat([x, x_mm], 1) as ONE
        channels-last map both sparse tensors scatter into."""
        vps = virtual_points if isinstance(virtual_points[0], (list, tuple)) else [virtual_points] * 4
        return self._path(points, vps, prepared=prepared, joint_bev=True)

    def extract_pts_feat(self, pts, img_feats=None, img_metas=None, virtual_points=None,
                         prepared=None):
        """MSMDFusion.py:421-452.  virtual_points: the per-scale foreground points when the
        caller has them already (bench.py, tests); otherwise they come from img_feats."""
        if virtual_points is None:
            if img_feats is None:
                raise ValueError("MSMDFusionDetector needs img_feats + img_metas or virtual_points")
            virtual_points = self.virtual_points_from_images(img_feats, img_metas)
        x = self.bev_fusion(self.extract_sparse_feat(pts, virtual_points, prepared=prepared))
        if self.with_pts_backbone:
            x = self.pts_backbone(x)
        if self.with_pts_neck:
            x = self.pts_neck(x)
        return x if isinstance(x, (list, tuple)) else [x]

    def forward(self, points, virtual_points=None, return_loss=False, prepared=None, **kw):
        if return_loss:
            return self.forward_train(points=points, virtual_points=virtual_points,
                                      prepared=prepared, **kw)
        return self.extract_pts_feat(points, kw.get("img_feats"), kw.get("img_metas"),
                                     virtual_points=virtual_points, prepared=prepared)


def freeze_lidar_components(model):
    """tools/train.py:185-219 (`freeze_lidar_components=True` in the LC config): the voxel
    layer, voxel encoder and LiDAR middle encoder stop training, and their BatchNorms stop
    tracking running statistics (they keep normalising with batch statistics, as the
    reference's `fix_bn` leaves them).  Also freezes the multimodal encoder's two block
    lists its forward never calls (sparse_multimodal_encoder_painting.py:142-156 vs :413-428)
    so that DDP needs no find_unused_parameters.  -> names of the parameters still trained."""
    for name, p in model.named_parameters():
        if any(k in name for k in ("pts_middle_encoder", "pts_voxel_layer", "pts_voxel_encoder")):
            p.requires_grad = False
    for part in (model.pts_voxel_layer, model.pts_voxel_encoder, model.pts_middle_encoder):
        if isinstance(part, nn.Module):
            for m in part.modules():
                if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
                    m.track_running_stats = False
    mm = getattr(model, "multimodal_middle_encoder", None)
    if mm is not None:
        from .distributed import freeze_unused_fusion_blocks
        freeze_unused_fusion_blocks(mm)
    spconv.functional.invalidate_packed_weights()
    return [n for n, p in model.named_parameters() if p.requires_grad]
