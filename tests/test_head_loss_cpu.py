"""Row f3, training half, without a GPU.

1. The oracle's restatements against the reference's own known answers
   (tests/test_utils/test_box3d.py:897-936, tests/test_utils/test_utils.py:6-11), an
   independent fp64 polygon clip, and heat maps painted by the reference's gaussian.py.
2. The host logic of msmdfusion_amd/head_loss.py (costs, batched Hungarian assignment, target
   assembly, losses and their gradients) against tests/golden/head_loss_vectors.npz -- outputs
   of the reference's get_targets_single / get_targets / loss / HungarianAssigner3D.assign
   (tests/golden/make_head_loss_golden.py) -- with the two HIP-only pieces (IoU kernel,
   heat-map painter) replaced by oracle stand-ins.
"""
import numpy as np
import pytest
import torch

import head_loss_fixture as FX
from msmdfusion_amd import head_loss as HL
from oracle import head_loss as OH


@pytest.fixture(scope="module")
def gold():
    return np.load(FX.GOLD)


# ------------------------------------------------------------------ the oracle is pinned
def test_oracle_iou3d_reference_known_answers():
    """The literals of the reference's test_boxes3d_overlaps, at its tolerances."""
    b1 = np.array([[1.8, -2.5, -1.8, 1.75, 3.39, 1.65, 1.6615927],
                   [8.9, -2.5, -1.6, 1.54, 4.01, 1.57, 1.5215927],
                   [28.3, 0.5, -1.3, 1.47, 2.23, 1.48, 4.7115927],
                   [31.3, -8.2, -1.6, 1.74, 3.77, 1.48, 0.35]], np.float32)
    b2 = np.array([[1.2, -3.0, -1.9, 1.8, 3.4, 1.7, 1.9], [8.1, -2.9, -1.8, 1.5, 4.1, 1.6, 1.8],
                   [31.3, -8.2, -1.6, 1.74, 3.77, 1.48, 0.35],
                   [20.1, -28.5, -1.9, 1.6, 3.5, 1.4, 5.1]], np.float32)
    iou = np.array([[0.3710, 0, 0, 0], [0, 0.3322, 0, 0], [0, 0, 0, 0], [0, 0, 1.0, 0]], np.float32)
    iof = np.array([[0.5582, 0, 0, 0], [0, 0.5025, 0, 0], [0, 0, 0, 0], [0, 0, 1.0, 0]], np.float32)
    np.testing.assert_allclose(OH.boxes_iou3d(b1, b2), iou, rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(OH.boxes_iou3d(b1, b2, "iof"), iof, rtol=1e-4, atol=5e-5)
    assert OH.boxes_iou3d(b1[:0], b2).shape == (0, 4)


def _clip_area(a, b):
    """Independent check: Sutherland-Hodgman clip of two convex quads in float64."""
    def corners(box):
        cx, cy = (box[0] + box[2]) / 2, (box[1] + box[3]) / 2
        c, s = np.cos(box[4]), np.sin(box[4])
        pts = []
        for x, y in ((box[0], box[1]), (box[2], box[1]), (box[2], box[3]), (box[0], box[3])):
            dx, dy = x - cx, y - cy
            pts.append((dx * c + dy * s + cx, -dx * s + dy * c + cy))   # the kernel's R(-angle)
        return pts
    poly, clip = corners(np.asarray(a, np.float64)), corners(np.asarray(b, np.float64))
    def area2(p):
        return sum(p[i][0] * p[(i + 1) % len(p)][1] - p[(i + 1) % len(p)][0] * p[i][1]
                   for i in range(len(p)))
    if area2(clip) < 0:
        clip = clip[::-1]
    for i in range(4):
        p0, p1 = clip[i], clip[(i + 1) % 4]
        side = lambda q: (p1[0] - p0[0]) * (q[1] - p0[1]) - (p1[1] - p0[1]) * (q[0] - p0[0])
        out = []
        for j in range(len(poly)):
            q0, q1 = poly[j], poly[(j + 1) % len(poly)]
            s0, s1 = side(q0), side(q1)
            if s0 >= 0:
                out.append(q0)
            if (s0 >= 0) != (s1 >= 0):
                t = s0 / (s0 - s1)
                out.append((q0[0] + t * (q1[0] - q0[0]), q0[1] + t * (q1[1] - q0[1])))
        poly = out
        if not poly:
            return 0.0
    return abs(area2(poly)) / 2


def test_oracle_overlap_bev_against_a_polygon_clip():
    rs = np.random.RandomState(3)
    n = 60
    ctr, size = rs.uniform(-4, 4, (n, 2)), rs.uniform(0.5, 5, (n, 2))
    boxes = np.concatenate([ctr - size / 2, ctr + size / 2, rs.uniform(-3.2, 3.2, (n, 1))],
                           axis=1).astype(np.float32)
    got = OH.boxes_overlap_bev(boxes[:30], boxes[30:])
    want = np.array([[_clip_area(a, b) for b in boxes[30:]] for a in boxes[:30]])
    assert (want > 0.1).sum() > 100
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-4)
    own = OH.boxes_overlap_bev(boxes, boxes)
    np.testing.assert_allclose(np.diag(own), size[:, 0] * size[:, 1], rtol=1e-4)
    np.testing.assert_allclose(own, own.T, rtol=1e-4, atol=1e-4)


def test_oracle_overlap_bev_invariances():
    """Properties any rotated-rectangle overlap has, at the float tolerance of the algorithm:
    a rigid motion of the pair changes nothing, a contained box overlaps by its own area,
    separated boxes by zero, and the 3-D IoU of a box with its vertically shifted copy is the
    height overlap's share."""
    rs = np.random.RandomState(11)
    n = 40
    ctr, size, ang = rs.uniform(-3, 3, (n, 2)), rs.uniform(0.5, 4, (n, 2)), rs.uniform(-3, 3, n)

    def boxes(c, a):
        return np.concatenate([c - size / 2, c + size / 2, a[:, None]], 1).astype(np.float32)
    base = OH.boxes_overlap_bev(boxes(ctr, ang)[:20], boxes(ctr, ang)[20:])
    # the kernel turns corners by R(-angle): turning the scene by +phi moves centres by R(-phi)
    phi = 0.7
    c, s_ = np.cos(phi), np.sin(phi)
    moved = np.stack([ctr[:, 0] * c + ctr[:, 1] * s_, -ctr[:, 0] * s_ + ctr[:, 1] * c], 1) + [5.0, -2.0]
    turned = OH.boxes_overlap_bev(boxes(moved, ang + phi)[:20], boxes(moved, ang + phi)[20:])
    np.testing.assert_allclose(turned, base, rtol=2e-4, atol=2e-4)
    big = np.array([[-4, -3, 4, 3, 0.4]], np.float32)
    small = np.array([[-0.5, -0.25, 0.5, 0.25, 1.1]], np.float32)
    assert abs(float(OH.boxes_overlap_bev(big, small)[0, 0]) - 0.5) < 1e-5
    far = np.array([[20, 20, 21, 21, 0.3]], np.float32)
    assert float(OH.boxes_overlap_bev(big, far)[0, 0]) == 0.0
    box = np.array([[1.0, 2.0, -1.0, 2.0, 4.0, 1.5, 0.6]], np.float32)
    up = box.copy()
    up[0, 2] += 0.5                                  # 1.0 of 1.5 in common: IoU = 1 / 2
    np.testing.assert_allclose(OH.boxes_iou3d(box, up), [[0.5]], rtol=1e-5)
    np.testing.assert_allclose(OH.boxes_iou3d(box, up, "iof"), [[2.0 / 3.0]], rtol=1e-5)


def test_oracle_gaussian_reference_known_answer():
    heat = np.zeros((128, 128), np.float32)
    OH.draw_heatmap_gaussian(heat, (64, 64), 2)
    assert abs(float(heat.sum()) - 4.3505) < 1e-3           # tests/test_utils/test_utils.py


def test_oracle_heatmap_against_the_reference_painter(gold):
    cfg = FX.TRAIN_CFG
    boxes, labels = gold["hm_boxes"], gold["hm_labels"]
    heat = OH.heatmap_targets(boxes, labels, 10, cfg["grid_size"], cfg["point_cloud_range"],
                              cfg["voxel_size"], cfg["out_size_factor"], cfg["gaussian_overlap"],
                              cfg["min_radius"])
    np.testing.assert_array_equal(heat, gold["hm_heatmap"])
    assert heat[3].max() == 0 and heat[5].max() == 0         # outside the map / zero width
    assert heat[0].max() == 1 and 0 < heat[4].max() < 1      # centre on / beyond the border
    for b, r in zip(boxes, gold["hm_radius"]):
        if r >= 0:
            w, l = b[3] / np.float32(0.075) / np.float32(8), b[4] / np.float32(0.075) / np.float32(8)
            assert OH.gaussian_radius(l, w, 0.1) == r


# ------------------------------------------------------------------ host logic vs the reference
def test_single_sample_assignment_matches_the_reference(gold):
    head = FX.build_head(oracle_parts=True)
    pred, boxes, labels = FX.golden_inputs(gold)
    dec = head.bbox_coder.decode(pred["heatmap"], pred["rot"], pred["dim"], pred["center"],
                                 pred["height"], pred["vel"])
    np.testing.assert_allclose(dec[0]["bboxes"].numpy(), gold["decoded_0"], rtol=1e-6, atol=1e-6)
    P = head.num_proposals
    ar = head.bbox_assigner.assign(dec[0]["bboxes"][:P], boxes[0].tensor, labels[0],
                                   pred["heatmap"][0:1, :, :P], head.train_cfg)
    np.testing.assert_array_equal(ar.gt_inds.numpy(), gold["assign_gt_inds"])
    np.testing.assert_array_equal(ar.labels.numpy(), gold["assign_labels"])
    np.testing.assert_allclose(ar.max_overlaps.numpy(), gold["assign_max_overlaps"], rtol=1e-6)
    assert (ar.gt_inds > 0).sum() == 7 and ar.max_overlaps.max() > 0.3
    c = head.bbox_coder
    enc = HL.encode_boxes(boxes[0].tensor, c.pc_range, c.out_size_factor, c.voxel_size, c.code_size)
    np.testing.assert_allclose(enc.numpy(), gold["encoded_0"], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(boxes[0].gravity_center.numpy(), gold["gravity_center_0"])
    empty = head.bbox_assigner.assign(dec[0]["bboxes"][:P], boxes[0].tensor[:0], labels[0][:0],
                                      pred["heatmap"][0:1, :, :P], head.train_cfg)
    assert empty.max_overlaps is None and (empty.gt_inds == 0).all()


def test_batched_targets_match_the_reference(gold):
    head = FX.build_head(oracle_parts=True)
    pred, boxes, labels = FX.golden_inputs(gold)
    keep = {k: v.clone() for k, v in pred.items()}
    tg = head.get_targets(boxes, labels, [pred])
    names = ("labels", "label_weights", "bbox_targets", "bbox_weights", "ious")
    for k, v in zip(names, tg):
        want = gold["target_" + k]
        assert v.dtype == torch.from_numpy(want).dtype and tuple(v.shape) == want.shape, k
        if v.dtype == torch.long:
            np.testing.assert_array_equal(v.numpy(), want, err_msg=k)
        else:
            np.testing.assert_allclose(v.numpy(), want, rtol=1e-6, atol=1e-6, err_msg=k)
    assert tg[5] == int(gold["target_num_pos"]) == 22         # 2 layers x (7 + 4) boxes
    np.testing.assert_allclose(float(tg[6]), float(gold["target_matched_ious"]), rtol=1e-6)
    np.testing.assert_array_equal(tg[7].numpy(), gold["target_heatmap"])
    for k, v in pred.items():                                  # "donot change the network outputs"
        assert torch.equal(v, keep[k]), k


def test_loss_and_gradients_match_the_reference(gold):
    head = FX.build_head(oracle_parts=True)
    pred, boxes, labels = FX.golden_inputs(gold, requires_grad=True)
    losses = head.loss(boxes, labels, ([pred],))
    want = {k[5:]: gold[k] for k in gold.files if k.startswith("loss_")}
    assert sorted(losses) == sorted(want)
    for k, v in losses.items():
        np.testing.assert_allclose(float(v.detach()), float(want[k]), rtol=2e-6, err_msg=k)
    sum(v for k, v in losses.items() if "loss" in k).backward()
    for k, v in pred.items():
        if "grad_" + k in gold.files:
            np.testing.assert_allclose(v.grad.numpy(), gold["grad_" + k], rtol=1e-4, atol=1e-7,
                                       err_msg=k)
    assert pred["query_heatmap_score"].grad is None


def test_sample_without_boxes_is_all_background(gold):
    """The reference raises here (torch.cat over a None overlap tensor, :1148); a batch with
    an empty sample trains on its classification / heat-map terms instead."""
    head = FX.build_head(oracle_parts=True)
    pred, boxes, labels = FX.golden_inputs(gold)
    boxes[1], labels[1] = HL.LiDARBoxes(boxes[1].tensor[:0], box_dim=9), labels[1][:0]
    tg = head.get_targets(boxes, labels, [pred])
    assert tg[5] == 14 and (tg[0][1] == head.num_classes).all() and tg[3][1].abs().sum() == 0
    assert tg[7][1].abs().sum() == 0 and tg[7][0].max() == 1
    np.testing.assert_array_equal(tg[0][0].numpy(), gold["target_labels"][0])
    both_empty = head.get_targets([b.__class__(b.tensor[:0], box_dim=9) for b in boxes],
                                  [l[:0] for l in labels], [pred])
    assert both_empty[5] == 0 and float(both_empty[6]) == 0


def test_gpu_only_parts_refuse_cpu_tensors(gold):
    head = FX.build_head()
    pred, boxes, labels = FX.golden_inputs(gold)
    with pytest.raises((RuntimeError, ValueError)):
        head.get_targets(boxes, labels, [pred])
    bare = FX.build_head()
    bare.train_cfg = None
    with pytest.raises(RuntimeError):
        bare.loss(boxes, labels, ([pred],))
