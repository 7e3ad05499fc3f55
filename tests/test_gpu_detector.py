"""The in-package detectors on the GPU: config dict -> build_detector -> a training step,
with the virtual points coming (a) from ready per-scale tensors, as bench.py feeds them, and
(b) from files on disk through loaders.py -> image_glue.pack_foreground -> get_foreground2D
(row f4 connected to the step)."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture()
def dev():
    return torch.device("cuda:0")


def _sparse_cfg():
    from msmdfusion_amd import configs as C
    return {k: v for k, v in C.MSMDFUSION_LC["model"].items()
            if k not in ("pts_backbone", "pts_neck")}


def test_detector_equals_the_bare_sparse_path(dev):
    """MSMDFusionDetector.extract_sparse_feat == SparseFusionPath on the same modules (the
    object every earlier parity test drives), prepared ahead or inline."""
    import proc_prefetch_helper as H
    from msmdfusion_amd.detector import build_detector
    torch.manual_seed(0)
    det = build_detector(_sparse_cfg()).to(dev).train()
    det.multimodal_middle_encoder.dummy_embedding_fn = H.fixed_dummy
    clouds, virt = H.make_batch(dev)
    with torch.no_grad():
        want = det._path(clouds, [virt] * 4, joint_bev=True)
        got = det.extract_sparse_feat(clouds, virt, prepared=det.prepare(clouds, virt))
        assert torch.equal(got, want)
        x = det.extract_pts_feat(clouds, virtual_points=virt)[0]     # + bev_fusion (SPP)
    assert x.shape[0] == 2 and x.shape[1] == 256 and torch.isfinite(x).all()


def test_files_to_training_step(dev, tmp_path, monkeypatch):
    """Virtual-point files on disk -> LoadForeground2D(+MultiSweeps) -> img_metas ->
    pack_foreground / depth-aware compression / get_foreground2D -> voxels -> GMA-Conv stack ->
    head loss -> backward: every trained parameter gets a finite gradient."""
    import foreground_files as FF
    from msmdfusion_amd import configs as C
    from msmdfusion_amd import loaders as L
    from msmdfusion_amd.detector import build_detector, freeze_lidar_components
    from msmdfusion_amd.head_loss import LiDARBoxes
    from msmdfusion_amd import synthetic as S
    monkeypatch.chdir(tmp_path)
    metas = []
    for b in range(2):
        res = FF.make_tree("data%d" % b, seed=20 + b)
        res = L.LoadForeground2DFromMultiSweeps()(L.LoadForeground2D()(copy.deepcopy(res)))
        info = res["foreground2D_info"]
        # pixels inside the padded image; points inside the detection range
        for cam in range(FF.CAMS):
            info["fg_pixels"][cam][:, :2] = np.clip(info["fg_pixels"][cam][:, :2], 0, [799, 447])
            info["fg_real_pixels"][cam][:, :2] = np.clip(info["fg_real_pixels"][cam][:, :2], 0,
                                                         [799, 447])
        metas.append(dict(foreground2D_info=info, pad_shape=(448, 800, 3), input_shape=(448, 800),
                          lidar2img=[np.eye(4, dtype=np.float32)] * FF.CAMS))
    torch.manual_seed(1)
    det = build_detector(dict(C.MSMDFUSION_LC["model"], pts_bbox_head=dict(C._PTS_BBOX_HEAD)),
                         train_cfg=dict(pts=dict(C._TRAIN_CFG_PTS)),
                         test_cfg=dict(pts=dict(C._TEST_CFG_PTS))).to(dev).train()
    trained = freeze_lidar_components(det)
    clouds = [torch.from_numpy(S.lidar_sweep(i, n_az=300)).to(dev) for i in range(2)]
    # stand-in for the (out-of-scope, injected) image backbone + FPN: 4 scales, 256 channels
    img_feats = [torch.randn(2 * FF.CAMS, 256, 448 // s, 800 // s, device=dev) * 0.1
                 for s in (4, 8, 16, 32)]
    rs = np.random.RandomState(3)
    gt_boxes, gt_labels = [], []
    for _ in range(2):
        g = 7
        box = np.zeros((g, 9), np.float32)
        box[:, 0:2] = rs.uniform(-40, 40, (g, 2))
        box[:, 2] = rs.uniform(-2, 0, g)
        box[:, 3:6] = rs.uniform((0.5, 0.5, 1.0), (2.5, 6.0, 3.0), (g, 3))
        box[:, 6] = rs.uniform(-3, 3, g)
        gt_boxes.append(LiDARBoxes(torch.from_numpy(box).to(dev)))
        gt_labels.append(torch.from_numpy(rs.randint(0, 10, g).astype(np.int64)).to(dev))
    vp = det.virtual_points_from_images(img_feats[:3], metas)
    assert len(vp) == 4 and len(vp[0]) == 2 and vp[0][0].shape[1] == 15 + 49
    losses = det.forward_pts_train(det.extract_pts_feat(clouds, img_feats[:3], metas), None,
                                   gt_boxes, gt_labels)
    total = sum(v for k, v in losses.items() if "loss" in k)
    assert torch.isfinite(total)
    total.backward()
    named = dict(det.named_parameters())
    missing = [n for n in trained if named[n].grad is None
               and not n.startswith(("conv1x1_blocks", "score_net"))]   # behind @no_grad voxelize
    assert not missing, missing[:5]
    assert all(torch.isfinite(named[n].grad).all() for n in trained if named[n].grad is not None)
    out = det.eval().simple_test(clouds, img_metas=metas, virtual_points=vp)
    assert len(out) == 2 and out[0]["boxes_3d"].shape[-1] >= 7
