"""TransFusionHead, training half: target assignment and losses.

Reference:
  TransFusionHead.get_targets / get_targets_single / loss
      mmdet3d/models/dense_heads/transfusion_head.py:1051-1286
  HungarianAssigner3D, BBoxBEVL1Cost, BBox3DL1Cost, IoU3DCost
      mmdet3d/core/bbox/assigners/hungarian_assigner.py:14-153
  TransFusionBBoxCoder.encode          mmdet3d/core/bbox/coders/transfusion_bbox_coder.py:24-39
  BboxOverlaps3D -> bbox_overlaps_3d -> BaseInstance3DBoxes.overlaps
      mmdet3d/core/bbox/iou_calculators/iou3d_calculator.py:56-167,
      mmdet3d/core/bbox/structures/base_box3d.py:352-438 (CUDA: ops/iou3d/src/iou3d_kernel.cu)
  gaussian_radius / draw_heatmap_gaussian   mmdet3d/core/utils/gaussian.py:5-86
  clip_sigmoid                         mmdet3d/models/utils/clip_sigmoid.py
  From mmdet 2.x (a dependency that is not in the reference tree; published definitions
  restated): FocalLossCost, PseudoSampler, AssignResult, FocalLoss (sigmoid), L1Loss,
  GaussianFocalLoss, weight_reduce_loss.

What the reference does per training step, for every sample and decoder layer in a python
loop: decode the predictions, two .cuda() copies and one kernel for the BEV overlaps, three
cost matrices, a device->host copy for scipy's linear_sum_assignment, two host->device
copies back, nonzero() for the sampler, then a python loop over the ground-truth boxes that
issues ~40 tiny tensor ops each to paint the heat map, and .item() for the heat-map
normaliser.  Here the same arithmetic runs once per BATCH: one IoU launch over all samples
and layers (`msmd_boxes_iou3d_f32`), one cost tensor, ONE device->host copy, the
assignments on the host (scipy, as the reference), one host->device copy of the matches,
vectorised target assembly, one launch that paints every box of every sample
(`msmd_heatmap_gaussian_f32`), and the heat-map loss with its gradient and its normaliser
in one pass (`msmd_gaussian_focal_f32`) -- no other synchronisation.

Deviations, both documented where they occur: a sample without ground-truth boxes is all
background here (the reference's torch.cat over a None overlap tensor raises); `matched_ious`
stays a device tensor instead of a python float.
"""
import numpy as np
import torch
from torch.nn import functional as F

try:
    from scipy.optimize import linear_sum_assignment
except ImportError:                               # the reference defers the same way (:8-11)
    linear_sum_assignment = None


# ------------------------------------------------------------------ boxes
class LiDARBoxes:
    """The two members of LiDARInstance3DBoxes the head's loss reads
    (core/bbox/structures/lidar_box3d.py:36-43, base_box3d.py): `.tensor` [G, 7+] =
    (x, y, z_bottom, dx, dy, dz, yaw, ...) and `.gravity_center`."""

    def __init__(self, tensor, box_dim=None):
        tensor = torch.as_tensor(tensor, dtype=torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape(0, box_dim or 7)
        if tensor.dim() != 2 or tensor.shape[1] < 7:
            raise ValueError("boxes must be [G, >=7], got %s" % (tuple(tensor.shape),))
        self.tensor = tensor

    def __len__(self):
        return self.tensor.shape[0]

    @property
    def gravity_center(self):
        t = self.tensor
        return torch.cat([t[:, :2], (t[:, 2] + t[:, 5] * 0.5)[:, None]], dim=1)

    def to(self, device):
        return LiDARBoxes(self.tensor.to(device))


def _box_tensor(boxes):
    return boxes.tensor if hasattr(boxes, "tensor") else torch.as_tensor(boxes,
                                                                         dtype=torch.float32)


class BboxOverlaps3D:
    """iou3d_calculator.py:56-91 with coordinate='lidar'.  Runs the HIP kernel; CPU tensors
    are refused (kernels._need_cuda) -- there is no host implementation in the product."""

    def __init__(self, coordinate="lidar"):
        if coordinate != "lidar":
            raise NotImplementedError("only LiDAR-coordinate boxes are built")
        self.coordinate = coordinate

    def __call__(self, bboxes1, bboxes2, mode="iou", nb_valid=None):
        from . import kernels as K
        return K.boxes_iou3d(bboxes1, bboxes2, nb_valid=nb_valid, mode=mode)


def encode_boxes(dst_boxes, pc_range, out_size_factor, voxel_size, code_size):
    """TransFusionBBoxCoder.encode (:24-39): [n, 7+] boxes -> [n, code_size] targets."""
    t = dst_boxes.new_zeros((dst_boxes.shape[0], code_size))
    t[:, 0] = (dst_boxes[:, 0] - pc_range[0]) / (out_size_factor * voxel_size[0])
    t[:, 1] = (dst_boxes[:, 1] - pc_range[1]) / (out_size_factor * voxel_size[1])
    t[:, 3:6] = dst_boxes[:, 3:6].log()
    t[:, 2] = dst_boxes[:, 2] + dst_boxes[:, 5] * 0.5           # bottom -> gravity centre
    t[:, 6] = torch.sin(dst_boxes[:, 6])
    t[:, 7] = torch.cos(dst_boxes[:, 6])
    if code_size == 10:
        t[:, 8:10] = dst_boxes[:, 7:]
    return t


# ------------------------------------------------------------------ match costs
class FocalLossCost:
    """mmdet.core.bbox.match_costs.FocalLossCost: cls_pred [n, C] logits, gt_labels [G]."""

    def __init__(self, weight=1.0, alpha=0.25, gamma=2, eps=1e-12):
        self.weight, self.alpha, self.gamma, self.eps = weight, alpha, gamma, eps

    def table(self, cls_pred):
        """[..., C]: pos_cost - neg_cost per class, before the label gather."""
        p = cls_pred.sigmoid()
        neg = -(1 - p + self.eps).log() * (1 - self.alpha) * p.pow(self.gamma)
        pos = -(p + self.eps).log() * self.alpha * (1 - p).pow(self.gamma)
        return pos, neg

    def __call__(self, cls_pred, gt_labels):
        pos, neg = self.table(cls_pred)
        return (pos[:, gt_labels] - neg[:, gt_labels]) * self.weight


class BBoxBEVL1Cost:
    """hungarian_assigner.py:24-37: L1 distance of the BEV centres, normalised by the range."""

    def __init__(self, weight):
        self.weight = weight

    def __call__(self, bboxes, gt_bboxes, train_cfg):
        rng = train_cfg["point_cloud_range"]
        start = bboxes.new_tensor(rng[0:2])
        extent = bboxes.new_tensor(rng[3:5]) - bboxes.new_tensor(rng[0:2])
        a = (bboxes[..., :2] - start) / extent
        b = (gt_bboxes[..., :2] - start) / extent
        d = (a[..., :, None, :] - b[..., None, :, :]).abs()
        return (d[..., 0] + d[..., 1]) * self.weight              # cdist(p=1) of two columns


class BBox3DL1Cost:
    """hungarian_assigner.py:14-21."""

    def __init__(self, weight):
        self.weight = weight

    def __call__(self, bboxes, gt_bboxes, train_cfg=None):
        return torch.cdist(bboxes, gt_bboxes, p=1) * self.weight


class IoU3DCost:
    """hungarian_assigner.py:40-47."""

    def __init__(self, weight):
        self.weight = weight

    def __call__(self, iou):
        return -iou * self.weight


_MATCH_COSTS = {"FocalLossCost": FocalLossCost, "BBoxBEVL1Cost": BBoxBEVL1Cost,
                "BBox3DL1Cost": BBox3DL1Cost, "IoU3DCost": IoU3DCost}


def _build(table, cfg):
    args = dict(cfg)
    return table[args.pop("type")](**args)


class AssignResult:
    """mmdet.core.bbox.assigners.AssignResult, the four fields the head reads."""

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = (num_gts, gt_inds,
                                                                      max_overlaps, labels)


def _solve(cost_np):
    if linear_sum_assignment is None:
        raise ImportError('Please run "pip install scipy" to install scipy first.')
    return linear_sum_assignment(cost_np)


class HungarianAssigner3D:
    """hungarian_assigner.py:95-153.  `assign` is the reference's per-sample call;
    `assign_batch` is what the head uses: every sample and decoder layer from one cost
    tensor and one device->host copy."""

    def __init__(self, cls_cost=None, reg_cost=None, iou_cost=None, iou_calculator=None):
        self.cls_cost = _build(_MATCH_COSTS, cls_cost or dict(type="FocalLossCost", weight=1.0))
        self.reg_cost = _build(_MATCH_COSTS, reg_cost or dict(type="BBoxBEVL1Cost", weight=1.0))
        self.iou_cost = _build(_MATCH_COSTS, iou_cost or dict(type="IoU3DCost", weight=1.0))
        calc = dict(iou_calculator or dict(type="BboxOverlaps3D", coordinate="lidar"))
        if calc.pop("type") != "BboxOverlaps3D":
            raise NotImplementedError("iou_calculator must be BboxOverlaps3D")
        self.iou_calculator = BboxOverlaps3D(**calc)

    def assign(self, bboxes, gt_bboxes, gt_labels, cls_pred, train_cfg):
        num_gts, num_bboxes = gt_bboxes.size(0), bboxes.size(0)
        gt_inds = bboxes.new_full((num_bboxes,), -1, dtype=torch.long)
        labels = bboxes.new_full((num_bboxes,), -1, dtype=torch.long)
        if num_gts == 0 or num_bboxes == 0:
            if num_gts == 0:
                gt_inds[:] = 0
            return AssignResult(num_gts, gt_inds, None, labels=labels)
        cls_cost = self.cls_cost(cls_pred[0].T, gt_labels)
        reg_cost = self.reg_cost(bboxes, gt_bboxes, train_cfg)
        iou = self.iou_calculator(bboxes, gt_bboxes)
        cost = cls_cost + reg_cost + self.iou_cost(iou)
        rows, cols = _solve(cost.detach().cpu().numpy())
        rows = torch.from_numpy(rows).to(bboxes.device)
        cols = torch.from_numpy(cols).to(bboxes.device)
        gt_inds[:] = 0
        gt_inds[rows] = cols + 1
        labels[rows] = gt_labels[cols]
        max_overlaps = torch.zeros_like(iou.max(1).values)
        max_overlaps[rows] = iou[rows, cols]
        return AssignResult(num_gts, gt_inds, max_overlaps, labels=labels)

    def assign_batch(self, bboxes, gt_bboxes, gt_labels, gt_counts, cls_pred, train_cfg,
                     num_layers=1):
        """bboxes [B, L*P, >=7]; gt_bboxes [B, Gmax, >=7] (rows past gt_counts[b] are
        padding), gt_labels [B, Gmax] long, gt_counts: python ints; cls_pred [B, C, L*P]
        logits.  Every layer's P proposals are matched separately.
        -> (sample, row, col) index arrays (numpy int64, host) and the IoU tensor
        [B, L*P, Gmax] (device)."""
        B, n, gmax = bboxes.shape[0], bboxes.shape[1], gt_bboxes.shape[1]
        dev = bboxes.device
        if gmax == 0 or n == 0 or sum(gt_counts) == 0:
            empty = np.zeros((0,), np.int64)
            return empty, empty, empty, bboxes.new_zeros((B, n, gmax))
        counts = torch.as_tensor(list(gt_counts), dtype=torch.int32).to(dev, non_blocking=True)
        iou = self.iou_calculator(bboxes, gt_bboxes, nb_valid=counts)
        pos, neg = self.cls_cost.table(cls_pred.permute(0, 2, 1))            # [B, n, C]
        pick = gt_labels[:, None, :].expand(B, n, gmax)
        cls_cost = (pos.gather(2, pick) - neg.gather(2, pick)) * self.cls_cost.weight
        cost = cls_cost + self.reg_cost(bboxes, gt_bboxes, train_cfg) + self.iou_cost(iou)
        cost_np = cost.detach().cpu().numpy()                 # the step's one host read
        per = n // num_layers
        out = ([], [], [])
        for b in range(B):
            g = int(gt_counts[b])
            if g == 0:
                continue
            for layer in range(num_layers):
                r, c = _solve(cost_np[b, layer * per:(layer + 1) * per, :g])
                out[0].append(np.full(r.shape, b, np.int64))
                out[1].append(r.astype(np.int64) + layer * per)
                out[2].append(c.astype(np.int64))
        return tuple(np.concatenate(v) for v in out) + (iou,)


# ------------------------------------------------------------------ heat map targets
def gaussian_radius(height, width, min_overlap):
    """core/utils/gaussian.py:56-86 on whole tensors (same float32 operations in the same
    order, so the integer radius agrees with the per-box evaluation)."""
    # a division by a python scalar runs as a multiplication by its reciprocal on the GPU
    # and as a division on the CPU; a tensor divisor is a true division on both
    b1 = height + width
    c1 = width * height * (1 - min_overlap) / height.new_tensor(1 + min_overlap)
    r1 = (b1 + torch.sqrt(b1 ** 2 - 4 * c1)) / 2
    b2 = 2 * (height + width)
    c2 = (1 - min_overlap) * width * height
    r2 = (b2 + torch.sqrt(b2 ** 2 - 16 * c2)) / 2
    a3 = 4 * min_overlap
    b3 = -2 * min_overlap * (height + width)
    c3 = (min_overlap - 1) * width * height
    r3 = (b3 + torch.sqrt(b3 ** 2 - 4 * a3 * c3)) / 2
    return torch.minimum(torch.minimum(r1, r2), r3)


def heatmap_boxes(gt_xy, gt_wl, train_cfg):
    """The per-box numbers of transfusion_head.py:1193-1209 for all boxes at once:
    -> (center_x, center_y, radius) int32; radius -1 where the reference skips the box."""
    osf, rng = train_cfg["out_size_factor"], train_cfg["point_cloud_range"]
    vs = gt_wl.new_tensor(train_cfg["voxel_size"][:2])     # tensor divisors: see gaussian_radius
    width = gt_wl[:, 0] / vs[0] / osf
    length = gt_wl[:, 1] / vs[1] / osf
    radius = gaussian_radius(length, width, train_cfg["gaussian_overlap"])
    radius = radius.to(torch.int32).clamp(min=int(train_cfg["min_radius"]))
    radius = torch.where((width > 0) & (length > 0), radius, torch.full_like(radius, -1))
    cx = ((gt_xy[:, 0] - rng[0]) / vs[0] / osf).to(torch.int32)
    cy = ((gt_xy[:, 1] - rng[1]) / vs[1] / osf).to(torch.int32)
    return cx, cy, radius


class HeatmapPainter:
    """draw_heatmap_gaussian for all boxes of the batch in one launch (HIP only)."""

    def __call__(self, heatmap, plane, cx, cy, radius):
        from . import kernels as K
        return K.heatmap_gaussian(heatmap, plane, cx, cy, radius)


# ------------------------------------------------------------------ losses (mmdet 2.x)
def clip_sigmoid(x, eps=1e-4):
    """mmdet3d/models/utils/clip_sigmoid.py (out of place: the input is kept)."""
    return torch.clamp(x.sigmoid(), min=eps, max=1 - eps)


def weight_reduce_loss(loss, weight=None, reduction="mean", avg_factor=None):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        if reduction == "mean":
            return loss.mean()
        return loss.sum() if reduction == "sum" else loss
    if reduction == "mean":
        return loss.sum() / avg_factor
    if reduction != "none":
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


class FocalLoss(torch.nn.Module):
    """mmdet FocalLoss(use_sigmoid=True): pred [n, C] logits, target [n] in 0..C (C =
    background), weight [n]."""

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction="mean",
                 loss_weight=1.0):
        super().__init__()
        if not use_sigmoid:
            raise NotImplementedError("Only sigmoid focal loss supported now.")
        self.gamma, self.alpha, self.reduction, self.loss_weight = gamma, alpha, reduction, \
            loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None):
        c = pred.size(1)
        t = F.one_hot(target, num_classes=c + 1)[:, :c].type_as(pred)
        p = pred.sigmoid()
        pt = (1 - p) * t + p * (1 - t)
        focal = (self.alpha * t + (1 - self.alpha) * (1 - t)) * pt.pow(self.gamma)
        loss = F.binary_cross_entropy_with_logits(pred, t, reduction="none") * focal
        if weight is not None and weight.shape != loss.shape:
            weight = weight.view(-1, 1) if weight.size(0) == loss.size(0) \
                else weight.view(loss.size(0), -1)
        return self.loss_weight * weight_reduce_loss(loss, weight, self.reduction, avg_factor)


class L1Loss(torch.nn.Module):
    """mmdet L1Loss."""

    def __init__(self, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None):
        loss = (pred - target).abs()
        return self.loss_weight * weight_reduce_loss(loss, weight, self.reduction, avg_factor)


class _FusedGaussianFocal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, clip):
        from . import kernels as K
        sums, grad = K.gaussian_focal(logits, target, clip, want_grad=logits.requires_grad)
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(sums)
        return sums[0], sums

    @staticmethod
    def backward(ctx, grad_loss, _grad_sums):
        (grad,) = ctx.saved_tensors
        return grad * grad_loss, None, None


class GaussianFocalLoss(torch.nn.Module):
    """mmdet GaussianFocalLoss (alpha=2, gamma=4) on the LOGITS of the dense heat map: the
    reference calls it as loss_heatmap(clip_sigmoid(logits), target, avg_factor=
    max(target.eq(1).sum().item(), 1)) (transfusion_head.py:1247-1249); `from_logits` is that
    whole expression -- on a CUDA tensor one fused pass that also counts the positives, so
    nothing is read back; on a CPU tensor the same formula in torch ops."""

    def __init__(self, alpha=2.0, gamma=4.0, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.alpha, self.gamma, self.reduction, self.loss_weight = alpha, gamma, reduction, \
            loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None):
        eps = 1e-12
        pos_w = target.eq(1)
        neg_w = (1 - target).pow(self.gamma)
        pos = -(pred + eps).log() * (1 - pred).pow(self.alpha) * pos_w
        neg = -(1 - pred + eps).log() * pred.pow(self.alpha) * neg_w
        return self.loss_weight * weight_reduce_loss(pos + neg, weight, self.reduction,
                                                     avg_factor)

    def from_logits(self, logits, target, clip=1e-4):
        if logits.is_cuda:
            if (self.alpha, self.gamma, self.reduction) != (2.0, 4.0, "mean"):
                raise NotImplementedError("fused heat-map loss: alpha=2, gamma=4, mean only")
            total, sums = _FusedGaussianFocal.apply(logits, target, clip)
            return self.loss_weight * total / sums[1].clamp(min=1)
        avg = target.eq(1).float().sum().clamp(min=1)
        return self.forward(clip_sigmoid(logits, clip), target, avg_factor=avg)


_LOSSES = {"FocalLoss": FocalLoss, "L1Loss": L1Loss, "GaussianFocalLoss": GaussianFocalLoss}


def build_loss(cfg):
    return _build(_LOSSES, cfg)


def build_assigner(cfg):
    args = dict(cfg)
    kind = args.pop("type")
    if kind != "HungarianAssigner3D":
        raise NotImplementedError("assigner %r is not built (the configs use "
                                  "HungarianAssigner3D)" % kind)
    return HungarianAssigner3D(**args)


# ------------------------------------------------------------------ the head's methods
def _pad_ground_truth(gt_bboxes_3d, gt_labels_3d, device):
    """list of per-sample boxes / labels -> padded device tensors + python counts.
    Built on the host when the inputs live there (one copy each), with torch ops otherwise."""
    tensors = [_box_tensor(b) for b in gt_bboxes_3d]
    counts = [int(t.shape[0]) for t in tensors]
    gmax, width = max(counts + [0]), max([t.shape[1] for t in tensors] + [7])
    B = len(tensors)
    boxes = torch.zeros((B, gmax, width), dtype=torch.float32, device=tensors[0].device)
    boxes[..., 3:6] = 1.0                       # padding rows: unit boxes (finite arithmetic)
    labels = torch.zeros((B, gmax), dtype=torch.long, device=gt_labels_3d[0].device)
    for b in range(B):
        if counts[b]:
            boxes[b, :counts[b]] = tensors[b]
            labels[b, :counts[b]] = gt_labels_3d[b].long()
    return boxes.to(device, non_blocking=True), labels.to(device, non_blocking=True), counts


def get_targets(head, gt_bboxes_3d, gt_labels_3d, preds_dict):
    """TransFusionHead.get_targets (:1051-1090) for the whole batch at once.
    preds_dict: [dict] (first index = level).  -> labels [B, L*P] long, label_weights
    [B, L*P] long, bbox_targets [B, L*P, code], bbox_weights [B, L*P, code], ious [B, L*P],
    num_pos (python int), matched_ious (0-dim tensor), heatmap [B, C, H, W] (only with
    initialize_by_heatmap)."""
    pred = preds_dict[0]
    cfg, coder = head.train_cfg, head.bbox_coder
    score = pred["heatmap"].detach()
    dev = score.device
    B, n = score.shape[0], score.shape[-1]
    vel = pred["vel"].detach() if "vel" in pred else None
    decoded = coder.decode(score, pred["rot"].detach(), pred["dim"].detach(),
                           pred["center"].detach(), pred["height"].detach(), vel)
    bboxes = torch.stack([d["bboxes"] for d in decoded])                    # [B, n, code]
    gt, gt_labels, counts = _pad_ground_truth(gt_bboxes_3d, gt_labels_3d, dev)
    layers = head.num_decoder_layers if head.auxiliary else 1
    sample, row, col, iou = head.bbox_assigner.assign_batch(
        bboxes, gt, gt_labels, counts, score, cfg, num_layers=layers)
    num_pos = int(row.shape[0])
    idx = torch.from_numpy(np.stack([sample, row, col])).to(dev, non_blocking=True)
    s_i, r_i, c_i = idx[0], idx[1], idx[2]

    code = coder.code_size
    bbox_targets = bboxes.new_zeros((B, n, code))
    bbox_weights = bboxes.new_zeros((B, n, code))
    labels = torch.full((B, n), head.num_classes, dtype=torch.long, device=dev)
    label_weights = torch.ones((B, n), dtype=torch.long, device=dev)       # pos_weight <= 0
    ious = bboxes.new_zeros((B, n))
    if num_pos:
        matched = gt[s_i, c_i]
        bbox_targets[s_i, r_i] = encode_boxes(matched, coder.pc_range, coder.out_size_factor,
                                              coder.voxel_size, code)
        bbox_weights[s_i, r_i] = 1.0
        labels[s_i, r_i] = gt_labels[s_i, c_i]
        if cfg["pos_weight"] > 0:
            label_weights[s_i, r_i] = cfg["pos_weight"]
        ious[s_i, r_i] = iou[s_i, r_i, c_i].clamp(min=0.0, max=1.0)
    # mean over samples of (sum of matched IoUs / max(matches of the sample, 1)) (:1212, :1084)
    per_sample = np.bincount(sample, minlength=B).astype(np.float32)
    denom = torch.from_numpy(np.maximum(per_sample, 1.0)).to(dev, non_blocking=True)
    matched_ious = (ious.sum(dim=1) / denom).mean()
    out = (labels, label_weights, bbox_targets, bbox_weights, ious, num_pos, matched_ious)
    if not head.initialize_by_heatmap:
        return out

    grid = cfg["grid_size"]
    w_map, h_map = grid[0] // cfg["out_size_factor"], grid[1] // cfg["out_size_factor"]
    heatmap = bboxes.new_zeros((B, head.num_classes, h_map, w_map))
    if sum(counts):
        valid = torch.arange(gt.shape[1], device=dev)[None, :] < \
            torch.as_tensor(counts, device=dev)[:, None]                     # [B, Gmax]
        flat = gt.reshape(-1, gt.shape[-1])
        cx, cy, radius = heatmap_boxes(flat[:, 0:2], flat[:, 3:5], cfg)
        plane = torch.arange(B, device=dev)[:, None] * head.num_classes + gt_labels
        plane = torch.where(valid, plane, torch.full_like(plane, -1)).reshape(-1)
        head.heatmap_painter(heatmap, plane, cx, cy, radius)
    return out + (heatmap,)


def loss(head, gt_bboxes_3d, gt_labels_3d, preds_dicts):
    """TransFusionHead.loss (:1221-1286)."""
    targets = get_targets(head, gt_bboxes_3d, gt_labels_3d, preds_dicts[0])
    labels, label_weights, bbox_targets, bbox_weights, ious, num_pos, matched_ious = targets[:7]
    pred = preds_dicts[0][0]
    losses = {}
    if head.initialize_by_heatmap:
        losses["loss_heatmap"] = head.loss_heatmap.from_logits(pred["dense_heatmap"], targets[7])
    P = head.num_proposals
    layers = head.num_decoder_layers if head.auxiliary else 1
    code_weights = head.train_cfg.get("code_weights", None)
    for i in range(layers):
        last = i == head.num_decoder_layers - 1 or (i == 0 and not head.auxiliary)
        prefix = "layer_-1" if last else "layer_%d" % i
        cut = slice(i * P, (i + 1) * P)
        cls_score = pred["heatmap"][..., cut].permute(0, 2, 1).reshape(-1, head.num_classes)
        losses[prefix + "_loss_cls"] = head.loss_cls(
            cls_score, labels[..., cut].reshape(-1), label_weights[..., cut].reshape(-1),
            avg_factor=max(num_pos, 1))
        parts = [pred[k][..., cut] for k in ("center", "height", "dim", "rot")]
        if "vel" in pred:
            parts.append(pred["vel"][..., cut])
        preds = torch.cat(parts, dim=1).permute(0, 2, 1)                    # [B, P, code]
        reg_w = bbox_weights[:, cut, :] * bbox_weights.new_tensor(code_weights)
        losses[prefix + "_loss_bbox"] = head.loss_bbox(preds, bbox_targets[:, cut, :], reg_w,
                                                       avg_factor=max(num_pos, 1))
    losses["matched_ious"] = matched_ious
    return losses
