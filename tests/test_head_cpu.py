"""Row f3: TransFusionHead's LiDAR branch against outputs of the reference's own classes
(tests/golden/head_vectors.npz, made by tests/golden/make_head_golden.py: the reference's
decoder layer, attention, FFN, forward_single and box coder executed on seeded inputs)."""
import os

import numpy as np
import pytest
import torch

from msmdfusion_amd import synthetic as S
from msmdfusion_amd.head import TransFusionBBoxCoder, TransFusionHead

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "head_vectors.npz")
CFG = dict(num_proposals=24, in_channels=32, hidden_channel=32, num_classes=10,
           num_decoder_layers=2, num_heads=4, nms_kernel_size=3, ffn_channel=48,
           initialize_by_heatmap=True,
           common_heads=dict(center=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2)),
           bbox_coder=dict(type="TransFusionBBoxCoder", pc_range=[-54.0, -54.0], out_size_factor=8,
                           voxel_size=[0.075, 0.075],
                           post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                           score_threshold=0.0, code_size=10),
           test_cfg=dict(dataset="nuScenes", grid_size=[160, 160, 40], out_size_factor=8,
                         nms_type=None))


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def head():
    return S.seeded_parameters(TransFusionHead(**CFG), seed=21).eval()


def test_state_dict_keys_are_the_reference_heads(gold, head):
    assert sorted(head.state_dict().keys()) == list(gold["state_dict_keys"])


def test_forward_single_matches_the_reference(gold, head):
    x = torch.from_numpy(np.random.RandomState(22).standard_normal((2, 32, 20, 20))
                         .astype(np.float32))
    np.testing.assert_array_equal(head.bev_pos.numpy(), gold["bev_pos"])
    with torch.no_grad():
        res = head(x)
    assert isinstance(res, tuple) and len(res) == 1 and len(res[0]) == 1
    (pred,) = res[0]
    np.testing.assert_array_equal(head.query_labels.numpy(), gold["query_labels"])
    assert sorted(pred) == sorted(k[3:] for k in gold.files if k.startswith("fs_"))
    for k, v in pred.items():
        np.testing.assert_allclose(v.numpy(), gold["fs_" + k], rtol=2e-4, atol=2e-5, err_msg=k)
    # two decoder layers, auxiliary: per-layer results concatenated on the proposal axis
    assert pred["center"].shape == (2, 2, 48) and pred["query_heatmap_score"].shape == (2, 10, 24)
    boxes = head.get_bboxes(res)
    for i, d in enumerate(boxes):
        np.testing.assert_array_equal(d["labels"].numpy(), gold["dec_%d_labels" % i])
        np.testing.assert_allclose(d["scores"].numpy(), gold["dec_%d_scores" % i], rtol=2e-4,
                                   atol=1e-6)
        np.testing.assert_allclose(d["bboxes"].numpy(), gold["dec_%d_bboxes" % i], rtol=2e-4,
                                   atol=2e-4)
        assert d["bboxes"].shape[1] == 9        # x, y, z, w, l, h, yaw, vx, vy


def test_bbox_coder_leaves_its_inputs_alone():
    coder = TransFusionBBoxCoder(**{k: v for k, v in CFG["bbox_coder"].items() if k != "type"})
    rng = torch.Generator().manual_seed(0)
    hm, rot, dim = torch.rand(1, 10, 7, generator=rng), torch.randn(1, 2, 7, generator=rng), \
        torch.randn(1, 3, 7, generator=rng)
    center, height, vel = torch.rand(1, 2, 7, generator=rng) * 20, torch.randn(1, 1, 7, generator=rng), \
        torch.randn(1, 2, 7, generator=rng)
    keep = [t.clone() for t in (dim, center)]
    out = coder.decode(hm, rot, dim, center, height, vel, filter=False)
    assert torch.equal(dim, keep[0]) and torch.equal(center, keep[1])
    assert out[0]["bboxes"].shape == (7, 9)
    np.testing.assert_allclose(out[0]["bboxes"][:, 0].numpy(),
                               (center[0, 0] * 8 * 0.075 - 54.0).numpy(), rtol=1e-6)


def test_lc_config_head_builds():
    """configs/MSMDFusion_nusc_voxel_LC.py:207-241 (values restated in configs.py)."""
    from msmdfusion_amd import configs as C
    head = C.build_head(C.MSMDFUSION_LC)
    sd = head.state_dict()
    assert tuple(sd["shared_conv.weight"].shape) == (128, 512, 3, 3)
    assert tuple(sd["heatmap_head.1.weight"].shape) == (10, 128, 3, 3)
    assert tuple(sd["decoder.0.multihead_attn.in_proj_weight"].shape) == (384, 128)
    assert tuple(sd["prediction_heads.0.vel.1.weight"].shape) == (2, 64, 1)
    assert len(head.decoder) == 1 and head.num_proposals == 200
    assert tuple(head.bev_pos.shape) == (1, 180 * 180, 2)
    with pytest.raises(NotImplementedError):
        TransFusionHead(fuse_img=True, **CFG)
