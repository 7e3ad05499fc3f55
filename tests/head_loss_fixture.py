"""Shared by the CPU and GPU tests of the head's training half (row f3): the small head of
tests/golden/make_head_loss_golden.py, its golden inputs, and oracle-backed stand-ins for the
two HIP-only pieces (3-D IoU, heat-map painting) so that the HOST logic of
msmdfusion_amd/head_loss.py can be checked on CPU tensors."""
import os

import numpy as np
import torch

from msmdfusion_amd import head_loss as HL
from msmdfusion_amd.head import TransFusionHead
from oracle import head_loss as OH

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                    "head_loss_vectors.npz")
TRAIN_CFG = dict(
    dataset="nuScenes",
    assigner=dict(type="HungarianAssigner3D",
                  iou_calculator=dict(type="BboxOverlaps3D", coordinate="lidar"),
                  cls_cost=dict(type="FocalLossCost", gamma=2, alpha=0.25, weight=0.15),
                  reg_cost=dict(type="BBoxBEVL1Cost", weight=0.25),
                  iou_cost=dict(type="IoU3DCost", weight=0.25)),
    pos_weight=-1, gaussian_overlap=0.1, min_radius=2, grid_size=[160, 160, 40],
    voxel_size=[0.075, 0.075, 0.2], out_size_factor=8,
    code_weights=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2],
    point_cloud_range=[-6.0, -6.0, -5.0, 6.0, 6.0, 3.0])
HEAD_CFG = dict(
    num_proposals=24, in_channels=32, hidden_channel=32, num_classes=10, num_decoder_layers=2,
    num_heads=4, nms_kernel_size=3, ffn_channel=48, initialize_by_heatmap=True, auxiliary=True,
    common_heads=dict(center=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2)),
    bbox_coder=dict(type="TransFusionBBoxCoder", pc_range=[-6.0, -6.0], out_size_factor=8,
                    voxel_size=[0.075, 0.075],
                    post_center_range=[-10.0, -10.0, -10.0, 10.0, 10.0, 10.0],
                    score_threshold=0.0, code_size=10),
    loss_cls=dict(type="FocalLoss", use_sigmoid=True, gamma=2, alpha=0.25, reduction="mean",
                  loss_weight=1.0),
    loss_bbox=dict(type="L1Loss", reduction="mean", loss_weight=0.25),
    loss_heatmap=dict(type="GaussianFocalLoss", reduction="mean", loss_weight=1.0),
    test_cfg=dict(dataset="nuScenes", grid_size=[160, 160, 40], out_size_factor=8, nms_type=None),
    train_cfg=TRAIN_CFG)


class OracleOverlaps:
    """BboxOverlaps3D's call surface on the oracle (any device in, same device out)."""

    def __call__(self, a, b, mode="iou", nb_valid=None):
        an, bn = a.detach().cpu().numpy(), b.detach().cpu().numpy()
        if an.ndim == 2:
            return torch.from_numpy(OH.boxes_iou3d(an, bn, mode)).to(a.device)
        out = np.zeros((an.shape[0], an.shape[1], bn.shape[1]), np.float32)
        counts = nb_valid.cpu().numpy() if nb_valid is not None else [bn.shape[1]] * an.shape[0]
        for s in range(an.shape[0]):
            g = int(counts[s])
            out[s, :, :g] = OH.boxes_iou3d(an[s], bn[s, :g], mode)
        return torch.from_numpy(out).to(a.device)


class OraclePainter:
    """HeatmapPainter's call surface on the oracle's draw_heatmap_gaussian."""

    def __call__(self, heatmap, plane, cx, cy, radius):
        h = heatmap.detach().cpu().numpy()
        flat = h.reshape(-1, h.shape[-2], h.shape[-1])
        for p, x, y, r in zip(plane.tolist(), cx.tolist(), cy.tolist(), radius.tolist()):
            if p >= 0 and r >= 0:
                OH.draw_heatmap_gaussian(flat[p], (x, y), r)
        heatmap.copy_(torch.from_numpy(h))
        return heatmap


def build_head(device="cpu", oracle_parts=False):
    head = TransFusionHead(**HEAD_CFG).to(device)
    if oracle_parts:
        head.bbox_assigner.iou_calculator = OracleOverlaps()
        head.heatmap_painter = OraclePainter()
    return head


def golden_inputs(gold, device="cpu", requires_grad=False):
    pred = {k[5:]: torch.from_numpy(gold[k]).to(device) for k in gold.files
            if k.startswith("pred_")}
    if requires_grad:
        for v in pred.values():
            if v.dtype.is_floating_point:
                v.requires_grad_(True)
    boxes = [HL.LiDARBoxes(torch.from_numpy(gold["gt_boxes_%d" % b])) for b in range(2)]
    labels = [torch.from_numpy(gold["gt_labels_%d" % b]) for b in range(2)]
    return pred, boxes, labels
