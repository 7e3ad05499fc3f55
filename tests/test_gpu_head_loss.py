"""Row f3, training half, on the GPU: the three HIP kernels under TransFusionHead.loss against
the oracle and the goldens made by the reference's own code, and the whole loss through the
C ABI against the reference's outputs (tests/golden/head_loss_vectors.npz)."""
import numpy as np
import pytest
import torch

import head_loss_fixture as FX
from msmdfusion_amd import head_loss as HL
from msmdfusion_amd import kernels as K
from oracle import head_loss as OH

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def gold():
    return np.load(FX.GOLD)


def _random_boxes(rs, n, extent=20.0):
    b = np.zeros((n, 9), np.float32)
    b[:, 0:2] = rs.uniform(-extent, extent, (n, 2))
    b[:, 2] = rs.uniform(-2, 0, n)
    b[:, 3:6] = rs.uniform(0.5, 6.0, (n, 3))
    b[:, 6] = rs.uniform(-6.3, 6.3, n)
    return b


def test_reference_known_answers_through_the_kernel(dev):
    """tests/test_utils/test_box3d.py:897-936 of the reference, its own tolerances."""
    b1 = torch.tensor([[1.8, -2.5, -1.8, 1.75, 3.39, 1.65, 1.6615927],
                       [8.9, -2.5, -1.6, 1.54, 4.01, 1.57, 1.5215927],
                       [28.3, 0.5, -1.3, 1.47, 2.23, 1.48, 4.7115927],
                       [31.3, -8.2, -1.6, 1.74, 3.77, 1.48, 0.35]], device=dev)
    b2 = torch.tensor([[1.2, -3.0, -1.9, 1.8, 3.4, 1.7, 1.9], [8.1, -2.9, -1.8, 1.5, 4.1, 1.6, 1.8],
                       [31.3, -8.2, -1.6, 1.74, 3.77, 1.48, 0.35],
                       [20.1, -28.5, -1.9, 1.6, 3.5, 1.4, 5.1]], device=dev)
    iou = torch.tensor([[0.3710, 0, 0, 0], [0, 0.3322, 0, 0], [0, 0, 0, 0], [0, 0, 1.0, 0]])
    iof = torch.tensor([[0.5582, 0, 0, 0], [0, 0.5025, 0, 0], [0, 0, 0, 0], [0, 0, 1.0, 0]])
    assert torch.allclose(K.boxes_iou3d(b1, b2).cpu(), iou, rtol=1e-4, atol=5e-5)
    assert torch.allclose(K.boxes_iou3d(b1, b2, mode="iof").cpu(), iof, rtol=1e-4, atol=5e-5)
    assert K.boxes_iou3d(b1[:0], b2).shape == (0, 4)
    heat = torch.zeros((1, 128, 128), device=dev)
    one = torch.tensor([0], dtype=torch.int32, device=dev)
    K.heatmap_gaussian(heat, one, one + 64, one + 64, one + 2)
    assert abs(float(heat.sum()) - 4.3505) < 1e-3           # tests/test_utils/test_utils.py:6-11


def test_overlap_bev_against_the_oracle(dev):
    """Float arithmetic with device cosf / sinf / atan2f: 1e-5 of the box scale."""
    rs = np.random.RandomState(5)
    a, b = _random_boxes(rs, 400, 12.0), _random_boxes(rs, 90, 12.0)
    xy = lambda t: OH.xywhr2xyxyr(t[:, [0, 1, 3, 4, 6]])
    want = OH.boxes_overlap_bev(xy(a), xy(b))
    got = K.boxes_overlap_bev(torch.from_numpy(xy(a)).to(dev), torch.from_numpy(xy(b)).to(dev))
    assert (want > 0.5).sum() > 500
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=2e-4)
    # a box with itself, and with itself turned by a quarter: exact areas
    sq = torch.tensor([[-1.0, -2.0, 3.0, 2.0, 0.3]], device=dev)
    assert abs(float(K.boxes_overlap_bev(sq, sq)) - 16.0) < 1e-4


def test_iou3d_batched_against_the_oracle(dev):
    rs = np.random.RandomState(6)
    B, na, nb = 3, 200, 64
    a = np.stack([_random_boxes(rs, na) for _ in range(B)])
    b = np.stack([_random_boxes(rs, nb) for _ in range(B)])
    counts = [64, 17, 0]
    got = K.boxes_iou3d(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev),
                        nb_valid=torch.tensor(counts, dtype=torch.int32, device=dev)).cpu().numpy()
    for s in range(B):
        want = OH.boxes_iou3d(a[s], b[s, :counts[s]])
        np.testing.assert_allclose(got[s, :, :counts[s]], want, rtol=1e-4, atol=2e-5)
        assert (got[s, :, counts[s]:] == 0).all()
    assert (got > 0.05).sum() > 30 and got.max() <= 1.0
    single = K.boxes_iou3d(torch.from_numpy(a[0]).to(dev), torch.from_numpy(b[0]).to(dev), mode="iof")
    np.testing.assert_allclose(single.cpu().numpy(), OH.boxes_iou3d(a[0], b[0], "iof"), rtol=1e-4,
                               atol=2e-5)
    with pytest.raises(ValueError):
        K.boxes_iou3d(torch.zeros(4, 6, device=dev), torch.zeros(4, 7, device=dev))
    with pytest.raises(RuntimeError):
        K.boxes_iou3d(torch.zeros(4, 7), torch.zeros(4, 7))             # host tensors


def test_heatmap_painter_against_the_reference_and_the_oracle(gold, dev):
    cfg = FX.TRAIN_CFG
    boxes = torch.from_numpy(gold["hm_boxes"]).to(dev)
    labels = torch.from_numpy(gold["hm_labels"]).to(dev)
    cx, cy, radius = HL.heatmap_boxes(boxes[:, 0:2], boxes[:, 3:5], cfg)
    want_r = np.where(gold["hm_radius"] < 0, -1, np.maximum(gold["hm_radius"].astype(np.int32), 2))
    np.testing.assert_array_equal(radius.cpu().numpy(), want_r)
    heat = torch.zeros((1, 10, 20, 20), device=dev)
    K.heatmap_gaussian(heat, labels.int(), cx, cy, radius)
    np.testing.assert_allclose(heat[0].cpu().numpy(), gold["hm_heatmap"], rtol=1.2e-7, atol=0)
    # LC-sized maps, 300 boxes over two samples, against the oracle's painter
    rs = np.random.RandomState(8)
    n = 300
    plane = rs.randint(-1, 20, n).astype(np.int32)
    px, py = rs.randint(-6, 186, n).astype(np.int32), rs.randint(-6, 186, n).astype(np.int32)
    rad = rs.randint(-1, 9, n).astype(np.int32)
    big = torch.zeros((2, 10, 180, 180), device=dev)
    K.heatmap_gaussian(big, *[torch.from_numpy(v).to(dev) for v in (plane, px, py, rad)])
    want = torch.zeros((2, 10, 180, 180))
    FX.OraclePainter()(want, torch.from_numpy(plane), torch.from_numpy(px), torch.from_numpy(py),
                       torch.from_numpy(rad))
    np.testing.assert_allclose(big.cpu().numpy(), want.numpy(), rtol=1.2e-7, atol=0)
    assert (want == 1).sum() > 100


def test_fused_heatmap_loss_against_torch(dev):
    """clip_sigmoid + GaussianFocalLoss + the positive count, value and gradient, against the
    torch fp32 composition of the same formula (LC map size)."""
    rs = np.random.RandomState(9)
    logits = torch.from_numpy(rs.standard_normal((2, 10, 180, 180)).astype(np.float32) * 4).to(dev)
    target = torch.from_numpy(rs.uniform(0, 1, (2, 10, 180, 180)).astype(np.float32) ** 6).to(dev)
    target.view(-1)[torch.from_numpy(rs.choice(target.numel(), 70, replace=False)).to(dev)] = 1.0
    loss_mod = HL.GaussianFocalLoss(reduction="mean", loss_weight=1.0)
    x1 = logits.clone().requires_grad_(True)
    got = loss_mod.from_logits(x1, target)
    got.backward()
    x2 = logits.clone().requires_grad_(True)
    want = loss_mod(HL.clip_sigmoid(x2), target, avg_factor=max(target.eq(1).float().sum().item(), 1))
    want.backward()
    np.testing.assert_allclose(float(got.detach()), float(want.detach()), rtol=2e-6)
    scale = float(x2.grad.abs().max())
    np.testing.assert_allclose(x1.grad.cpu().numpy(), x2.grad.cpu().numpy(), rtol=2e-4,
                               atol=2e-6 * scale)
    sums, _ = K.gaussian_focal(logits, target)
    assert float(sums[1]) == 70.0
    # saturated logits: the clamp stops the gradient exactly as torch's does
    sat = torch.tensor([-30.0, 30.0, 0.0], device=dev, requires_grad=True)
    tgt = torch.tensor([0.0, 1.0, 0.5], device=dev)
    loss_mod.from_logits(sat, tgt).backward()
    assert sat.grad[0] == 0 and sat.grad[1] == 0 and sat.grad[2] != 0


def test_targets_and_loss_match_the_reference(gold, dev):
    """The whole training half on the GPU (IoU kernel, painter, fused heat-map loss) against
    the reference's get_targets / loss outputs."""
    head = FX.build_head(dev)
    pred, boxes, labels = FX.golden_inputs(gold, dev, requires_grad=True)
    tg = head.get_targets(boxes, labels, [{k: v.detach() for k, v in pred.items()}])
    for k, v in zip(("labels", "label_weights", "bbox_targets", "bbox_weights", "ious"), tg):
        want = gold["target_" + k]
        if v.dtype == torch.long:
            np.testing.assert_array_equal(v.cpu().numpy(), want, err_msg=k)
        else:
            np.testing.assert_allclose(v.cpu().numpy(), want, rtol=1e-4, atol=2e-5, err_msg=k)
    assert tg[5] == int(gold["target_num_pos"])
    np.testing.assert_allclose(float(tg[6]), float(gold["target_matched_ious"]), rtol=1e-4)
    np.testing.assert_allclose(tg[7].cpu().numpy(), gold["target_heatmap"], rtol=1.2e-7, atol=0)
    losses = head.loss(boxes, labels, ([pred],))
    for k, v in losses.items():
        np.testing.assert_allclose(float(v.detach()), float(gold["loss_" + k]), rtol=1e-4,
                                   err_msg=k)
    sum(v for k, v in losses.items() if "loss" in k).backward()
    for k, v in pred.items():
        if "grad_" + k in gold.files:
            want = gold["grad_" + k]
            np.testing.assert_allclose(v.grad.cpu().numpy(), want, rtol=1e-3,
                                       atol=1e-5 * float(np.abs(want).max()), err_msg=k)


def test_lc_sized_loss_against_the_host_logic_on_the_oracle(dev):
    """configs[2]'s head (200 proposals, 180 x 180 map, 10 classes), 2 samples with 45 and 30
    boxes: the GPU path against the same module on CPU tensors with the oracle's IoU and
    painter -- same matches, same targets, same losses."""
    from msmdfusion_amd import configs as C
    rs = np.random.RandomState(12)
    head = C.build_head(C.MSMDFUSION_LC)
    P = head.num_proposals
    gts, labs = [], []
    for g in (45, 30):
        b = _random_boxes(rs, g, 50.0)
        b[:, 3:6] = rs.uniform(0.5, 5.0, (g, 3))
        b[:, 7:9] = rs.uniform(-3, 3, (g, 2))
        gts.append(b)
        labs.append(rs.randint(0, 10, g).astype(np.int64))
    pred = dict(heatmap=rs.standard_normal((2, 10, P)) * 2, center=rs.uniform(0, 180, (2, 2, P)),
                height=rs.uniform(-2, 1, (2, 1, P)), dim=rs.uniform(-0.5, 1.6, (2, 3, P)),
                rot=rs.uniform(-1, 1, (2, 2, P)), vel=rs.standard_normal((2, 2, P)),
                dense_heatmap=rs.standard_normal((2, 10, 180, 180)) * 3,
                query_heatmap_score=rs.uniform(0, 1, (2, 10, P)))
    for s in range(2):                       # some proposals on top of the boxes
        for k in range(len(gts[s])):
            j, g = (k * 4 + s) % P, gts[s][k]
            pred["center"][s, :, j] = (g[0:2] + 54.0) / 0.6 + rs.uniform(-0.5, 0.5, 2)
            pred["dim"][s, :, j] = np.log(g[3:6]) + rs.uniform(-0.15, 0.15, 3)
            pred["height"][s, 0, j] = g[2] + g[5] / 2
            pred["rot"][s, :, j] = (np.sin(g[6]), np.cos(g[6]))
    pred = {k: torch.from_numpy(v.astype(np.float32)) for k, v in pred.items()}
    boxes = [HL.LiDARBoxes(torch.from_numpy(b)) for b in gts]
    labels = [torch.from_numpy(l) for l in labs]

    cpu_head = C.build_head(C.MSMDFUSION_LC)
    cpu_head.bbox_assigner.iou_calculator = FX.OracleOverlaps()
    cpu_head.heatmap_painter = FX.OraclePainter()
    want_t = cpu_head.get_targets(boxes, labels, [pred])
    want_l = cpu_head.loss(boxes, labels, ([pred],))

    gpu_pred = {k: v.to(dev) for k, v in pred.items()}
    got_t = head.to(dev).get_targets(boxes, labels, [gpu_pred])
    got_l = head.loss(boxes, labels, ([gpu_pred],))
    assert got_t[5] == want_t[5] == 75
    assert float(want_t[6]) > 0.1                                  # the IoU term is live
    for k, a, b in zip(("labels", "label_weights", "bbox_targets", "bbox_weights", "ious"),
                       got_t, want_t):
        if a.dtype == torch.long:
            assert torch.equal(a.cpu(), b), k
        else:
            np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=1e-4, atol=2e-5, err_msg=k)
    np.testing.assert_allclose(got_t[7].cpu().numpy(), want_t[7].numpy(), rtol=1.2e-7, atol=0)
    for k in want_l:
        np.testing.assert_allclose(float(got_l[k]), float(want_l[k]), rtol=1e-4, err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [False, True])
def test_forward_single_on_the_gpu_matches_the_reference(rows):
    """TransFusionHead.forward_single on the GPU -- the torch / MIOpen form and the row-kernel
    form (shared_conv and heat-map convs on the sparse-conv kernels) -- against the outputs of
    the REFERENCE's own forward_single (tests/golden/head_vectors.npz,
    transfusion_head.py:755-1027), the same vectors tests/test_head_cpu.py pins the CPU path
    with.  (Round 2 compared the row form with torch on the same GPU only.)"""
    import test_head_cpu as T
    from msmdfusion_amd import synthetic as S
    from msmdfusion_amd.head import TransFusionHead
    dev = torch.device("cuda:0")
    gold = np.load(T.GOLD)
    head = S.seeded_parameters(TransFusionHead(rows=rows, **T.CFG), seed=21).eval().to(dev)
    x = torch.from_numpy(np.random.RandomState(22).standard_normal((2, 32, 20, 20))
                         .astype(np.float32)).to(dev)
    with torch.no_grad():
        res = head(x)
    (pred,) = res[0]
    np.testing.assert_array_equal(head.query_labels.cpu().numpy(), gold["query_labels"])
    for k, v in pred.items():
        np.testing.assert_allclose(v.cpu().numpy(), gold["fs_" + k], rtol=2e-4, atol=5e-5,
                                   err_msg="%s rows=%s" % (k, rows))
    for i, d in enumerate(head.get_bboxes(res)):
        np.testing.assert_array_equal(d["labels"].cpu().numpy(), gold["dec_%d_labels" % i])
        np.testing.assert_allclose(d["scores"].cpu().numpy(), gold["dec_%d_scores" % i],
                                   rtol=2e-4, atol=2e-6)
        np.testing.assert_allclose(d["bboxes"].cpu().numpy(), gold["dec_%d_bboxes" % i],
                                   rtol=2e-4, atol=2e-4)
