"""CPU restatement of the box / heat-map arithmetic under TransFusionHead.loss (row f3).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

  boxes_overlap_bev   iou3d_cuda.boxes_overlap_bev_gpu (ops/iou3d/src/iou3d.cpp:70-98 ->
                      iou3d_kernel.cu:36-264): C restatement in msmd_oracle.c
  boxes_iou3d         BaseInstance3DBoxes.overlaps (core/bbox/structures/base_box3d.py:352-438)
                      with LiDARInstance3DBoxes.bev (lidar_box3d.py:87-90) and xywhr2xyxyr
                      (structures/utils.py:62-82)
  gaussian_2d / draw_heatmap_gaussian / gaussian_radius   core/utils/gaussian.py:5-86
  heatmap_targets     the dense-heat-map loop of TransFusionHead.get_targets_single
                      (models/dense_heads/transfusion_head.py:1186-1210)

Pinned by the reference's own known answers (tests/test_utils/test_box3d.py:897-936,
tests/test_utils/test_utils.py:6-11) and by goldens made by running gaussian.py itself
(tests/golden/make_head_loss_golden.py).
"""
import ctypes as C

import numpy as np

from . import oracle as _o


def boxes_overlap_bev(boxes_a, boxes_b):
    a = np.ascontiguousarray(boxes_a, np.float32).reshape(-1, 5)
    b = np.ascontiguousarray(boxes_b, np.float32).reshape(-1, 5)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    fn = _o._lib.orc_boxes_overlap_bev
    fn.restype = None
    fn(_o._f(a), C.c_int(a.shape[0]), _o._f(b), C.c_int(b.shape[0]), _o._f(out))
    return out


def xywhr2xyxyr(bev):
    bev = np.asarray(bev, np.float32)
    out = np.zeros_like(bev)
    half_w, half_h = bev[:, 2] / np.float32(2), bev[:, 3] / np.float32(2)
    out[:, 0], out[:, 1] = bev[:, 0] - half_w, bev[:, 1] - half_h
    out[:, 2], out[:, 3] = bev[:, 0] + half_w, bev[:, 1] + half_h
    out[:, 4] = bev[:, 4]
    return out


def boxes_iou3d(boxes1, boxes2, mode="iou"):
    """boxes [n, >=7] (x, y, z_bottom, dx, dy, dz, yaw) float32 -> [n1, n2]"""
    b1, b2 = np.asarray(boxes1, np.float32), np.asarray(boxes2, np.float32)
    if b1.shape[0] * b2.shape[0] == 0:
        return np.zeros((b1.shape[0], b2.shape[0]), np.float32)
    top1, bot1 = (b1[:, 2] + b1[:, 5])[:, None], b1[:, 2][:, None]
    top2, bot2 = (b2[:, 2] + b2[:, 5])[None], b2[:, 2][None]
    overlap_h = np.maximum(np.minimum(top1, top2) - np.maximum(bot1, bot2), np.float32(0))
    bev = boxes_overlap_bev(xywhr2xyxyr(b1[:, [0, 1, 3, 4, 6]]), xywhr2xyxyr(b2[:, [0, 1, 3, 4, 6]]))
    inter = bev * overlap_h
    vol1 = (b1[:, 3] * b1[:, 4] * b1[:, 5])[:, None]
    vol2 = (b2[:, 3] * b2[:, 4] * b2[:, 5])[None]
    if mode == "iou":
        return inter / np.maximum(vol1 + vol2 - inter, np.float32(1e-8))
    return inter / np.maximum(vol1, np.float32(1e-8))


def gaussian_2d(shape, sigma=1.0):
    m, n = [(s - 1.0) / 2.0 for s in shape]
    y, x = np.ogrid[-m:m + 1, -n:n + 1]
    h = np.exp(-(x * x + y * y) / (2 * sigma * sigma))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    return h


def draw_heatmap_gaussian(heatmap, center, radius, k=1):
    """heatmap [H, W] float32, modified in place (maximum with the clipped bump)."""
    diameter = 2 * radius + 1
    g = gaussian_2d((diameter, diameter), sigma=diameter / 6)
    x, y = int(center[0]), int(center[1])
    height, width = heatmap.shape
    left, right = min(x, radius), min(width - x, radius + 1)
    top, bottom = min(y, radius), min(height - y, radius + 1)
    target = heatmap[y - top:y + bottom, x - left:x + right]
    bump = g[radius - top:radius + bottom, radius - left:radius + right].astype(np.float32)
    if min(bump.shape) > 0 and min(target.shape) > 0:
        np.maximum(target, bump * np.float32(k), out=target)
    return heatmap


def gaussian_radius(height, width, min_overlap):
    """float32 arithmetic in the reference's operation order (python scalars enter a
    float32 tensor op as float32)."""
    f = np.float32
    h, w = f(height), f(width)
    b1 = h + w
    c1 = w * h * f(1 - min_overlap) / f(1 + min_overlap)
    r1 = (b1 + np.sqrt(b1 * b1 - f(4) * c1)) / f(2)
    b2 = f(2) * (h + w)
    c2 = f(1 - min_overlap) * w * h
    r2 = (b2 + np.sqrt(b2 * b2 - f(16) * c2)) / f(2)
    a3 = 4 * min_overlap
    b3 = f(-2 * min_overlap) * (h + w)
    c3 = f(min_overlap - 1) * w * h
    r3 = (b3 + np.sqrt(b3 * b3 - f(4 * a3) * c3)) / f(2)
    return min(r1, r2, r3)


def heatmap_targets(gt_boxes, gt_labels, num_classes, grid_size, pc_range, voxel_size,
                    out_size_factor, gaussian_overlap, min_radius):
    """gt_boxes [G, >=7] with the GRAVITY centre in columns 0..2 -> [num_classes, H, W]"""
    f = np.float32
    gt = np.asarray(gt_boxes, f)
    wmap, hmap = grid_size[0] // out_size_factor, grid_size[1] // out_size_factor
    heat = np.zeros((num_classes, hmap, wmap), f)
    for i in range(gt.shape[0]):
        width = gt[i, 3] / f(voxel_size[0]) / f(out_size_factor)
        length = gt[i, 4] / f(voxel_size[1]) / f(out_size_factor)
        if not (width > 0 and length > 0):
            continue
        radius = max(min_radius, int(gaussian_radius(length, width, gaussian_overlap)))
        cx = (gt[i, 0] - f(pc_range[0])) / f(voxel_size[0]) / f(out_size_factor)
        cy = (gt[i, 1] - f(pc_range[1])) / f(voxel_size[1]) / f(out_size_factor)
        draw_heatmap_gaussian(heat[int(gt_labels[i])], (int(cx), int(cy)), radius)
    return heat
