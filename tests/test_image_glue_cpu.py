"""CPU tests of rows a13 (image half), f2 and f1: the numpy restatement
(oracle/image_glue.py) against outputs of the reference's OWN functions
(tests/golden/image_glue_vectors.npz, made by make_image_glue_golden.py), and the
dense BEV tail modules against the reference's SPPModule output, its checkpoint
key layout and plain PyTorch."""
import numpy as np
import pytest
import torch
from torch import nn
from torch.nn import functional as F

from msmdfusion_amd import synthetic as S
from oracle import image_glue as OI

from image_glue_fixture import Fixture


@pytest.fixture(scope="module")
def fx():
    return Fixture()


def test_oracle_get_foreground2d_matches_the_reference(fx):
    sd = fx.state_dict("score_net.")
    w, b = sd["0.weight"].numpy(), sd["0.bias"].numpy()
    for i, feat in enumerate(fx.fg_inputs):
        got = OI.get_foreground2d(feat, fx.metas, w, b)
        for bi in range(fx.B):
            want = fx.fg[i][bi]
            assert got[bi].shape == want.shape
            np.testing.assert_array_equal(got[bi][:, :15], want[:, :15])      # points: copied
            np.testing.assert_allclose(got[bi][:, 15:], want[:, 15:], rtol=2e-5, atol=2e-6)
    # the fixture exercises what it should: an empty camera, scores that are not all 0
    assert any(p.shape[0] == 0 for p in fx.metas[1]["foreground2D_info"]["fg_pixels"])
    assert np.abs(fx.fg[0][0][:, 15:]).max() > 0


def test_oracle_bilinear_resize_is_interpolate():
    rng = np.random.RandomState(0)
    x = rng.rand(3, 1, 64, 112).astype(np.float32)
    for h, w in [(8, 14), (4, 7), (2, 4), (16, 28), (64, 112), (100, 130)]:
        want = F.interpolate(torch.from_numpy(x), (h, w), mode="bilinear").numpy()
        np.testing.assert_allclose(OI.bilinear_resize(x, h, w), want, rtol=1e-5, atol=1e-6)


def _blocks(fx):
    blocks = nn.ModuleList([
        nn.Sequential(nn.Conv2d(fx.c_img + 1, fx.c_out, k, 1, k // 2, bias=False),
                      nn.BatchNorm2d(fx.c_out, eps=0.001, momentum=0.01), nn.ReLU())
        for k in (5, 5, 3)])
    blocks.load_state_dict(fx.state_dict("conv1x1_blocks."))
    return blocks.train()


def test_oracle_depth_canvas_matches_the_reference(fx):
    """canvas (last row wins on a shared pixel) -> bilinear -> the conv blocks with
    the fixture's weights == the reference's depth_aware_channel_compression."""
    canvas = OI.depth_canvas(fx.metas, fx.H, fx.W, fx.cams)
    # duplicates exist in the fixture, so the overwrite rule is exercised
    dup = 0
    for m in fx.metas:
        for r in m["foreground2D_info"]["fg_real_pixels"]:
            xy = np.trunc(r[:, :2]).astype(np.int64)
            dup += r.shape[0] - np.unique(xy, axis=0).shape[0]
    assert dup > 0
    blocks = _blocks(fx)
    with torch.no_grad():
        for i, block in enumerate(blocks):
            feat = torch.from_numpy(fx.feats[i])
            depth = torch.from_numpy(OI.bilinear_resize(canvas, *feat.shape[-2:]))
            got = block(torch.cat([feat, depth], 1)).numpy()
            np.testing.assert_allclose(got, fx.comp[i], rtol=1e-4, atol=1e-5)


def test_oracle_bev_concat_is_dense_view_cat():
    rng = np.random.RandomState(3)
    B, H, W = 2, 9, 7
    tensors = []
    for c, D in [(4, 2), (6, 1)]:
        idx = S.random_voxel_indices(40, B, [D, H, W], seed=c, clustered=False)
        tensors.append((rng.randn(idx.shape[0], c).astype(np.float32), idx, (D, H, W)))
    got = OI.bev_concat(tensors, B)
    maps = []
    for feat, idx, (D, H, W) in tensors:          # structure.py:55-64 with torch
        dense = torch.zeros(B, D, H, W, feat.shape[1])
        ii = torch.from_numpy(idx).long()
        dense[ii[:, 0], ii[:, 1], ii[:, 2], ii[:, 3]] = torch.from_numpy(feat)
        dense = dense.permute(0, 4, 1, 2, 3).contiguous()
        maps.append(dense.view(B, -1, H, W))
    np.testing.assert_array_equal(got, torch.cat(maps, 1).numpy())


# ------------------------------------------------------------------ f1: dense BEV tail
def test_spp_module_matches_the_reference_output(fx):
    """Our SPPModule with the name-seeded weights the reference class was given
    reproduces the reference's forward (train-mode BN) and running statistics."""
    from msmdfusion_amd.bev import SPPModule
    spp = S.seeded_parameters(SPPModule(), seed=13).train()
    x = torch.from_numpy(np.random.RandomState(14).standard_normal((2, 640, 12, 12))
                         .astype(np.float32))
    with torch.no_grad():
        y = spp(x)
    np.testing.assert_allclose(y.numpy(), fx.g["spp_y"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(spp.fuse[1].running_mean.numpy(), fx.g["spp_running_mean_fuse"],
                               rtol=1e-4, atol=1e-6)
    # channels-last weights + input: same numbers
    spp2 = S.seeded_parameters(SPPModule(), seed=13).train().to(memory_format=torch.channels_last)
    with torch.no_grad():
        y2 = spp2(x.contiguous(memory_format=torch.channels_last))
    np.testing.assert_allclose(y2.numpy(), fx.g["spp_y"], rtol=1e-4, atol=1e-5)


def _bn_keys(prefix):
    return [prefix + s for s in ("weight", "bias", "running_mean", "running_var",
                                 "num_batches_tracked")]


def test_bev_tail_checkpoint_keys_and_shapes():
    """Key layout of the reference modules (backbones/second.py:32-61: blocks.i =
    Sequential of conv, BN, ReLU triples; necks/second_fpn.py:44-67: deblocks.i =
    Sequential(up, BN, ReLU); MSMDFusion.py:51-78 for SPPModule), built from the LC
    config's own dicts."""
    from msmdfusion_amd import configs as C
    tail = C.build_bev_tail(C.MSMDFUSION_LC, rows=False)
    sd = tail.state_dict()
    # the row-kernel variants are the same modules with another forward: same keys
    assert sorted(C.build_bev_tail(C.MSMDFUSION_LC).state_dict()) == sorted(sd)
    want = []
    for br in ("conv1x1", "conv3x3", "dilated_conv3x3_rate6", "dilated_conv3x3_rate12", "fuse"):
        want += [f"bev_fusion.{br}.0.weight"] + _bn_keys(f"bev_fusion.{br}.1.")
    for i in range(2):
        for j in range(6):
            want += [f"pts_backbone.blocks.{i}.{3 * j}.weight"] + \
                _bn_keys(f"pts_backbone.blocks.{i}.{3 * j + 1}.")
    for i in range(2):
        want += [f"pts_neck.deblocks.{i}.0.weight"] + _bn_keys(f"pts_neck.deblocks.{i}.1.")
    assert sorted(sd.keys()) == sorted(want)
    assert tuple(sd["bev_fusion.dilated_conv3x3_rate12.0.weight"].shape) == (256, 640, 3, 3)
    assert tuple(sd["bev_fusion.fuse.0.weight"].shape) == (256, 1024, 1, 1)
    assert tuple(sd["pts_backbone.blocks.1.0.weight"].shape) == (256, 128, 3, 3)
    # use_conv_for_no_stride: level 0 is a 1x1 Conv2d, level 1 a 2x2 stride-2 deconv
    assert isinstance(tail.pts_neck.deblocks[0][0], nn.Conv2d)
    assert tuple(sd["pts_neck.deblocks.0.0.weight"].shape) == (256, 128, 1, 1)
    assert isinstance(tail.pts_neck.deblocks[1][0], nn.ConvTranspose2d)
    assert tuple(sd["pts_neck.deblocks.1.0.weight"].shape) == (256, 256, 2, 2)
    bn = tail.pts_backbone.blocks[0][1]
    assert (bn.eps, bn.momentum) == (1e-3, 0.01)
    assert tail.pts_backbone.blocks[1][0].stride == (2, 2)
    assert tail.bev_fusion.dilated_conv3x3_rate6[0].dilation == (6, 6)


def test_bev_tail_forward_backward_equals_plain_torch():
    """BevTail (channels-last) == the same layers applied one by one in NCHW."""
    from msmdfusion_amd.bev import SECOND, SECONDFPN, BevTail, SPPModule
    torch.manual_seed(0)
    tail = BevTail(SPPModule(in_channels=24, channels=16),
                   SECOND(16, [8, 16], [1, 2], [1, 2]),
                   SECONDFPN([8, 16], [12, 12], [1, 2], use_conv_for_no_stride=True),
                   channels_last=True).train()
    x = torch.randn(2, 10, 12, 12)
    xm = torch.randn(2, 14, 12, 12)
    xa, xb = x.clone().requires_grad_(True), xm.clone().requires_grad_(True)
    out = tail(xa, xb)
    assert isinstance(out, list) and len(out) == 1 and tuple(out[0].shape) == (2, 24, 12, 12)
    out[0].square().sum().backward()

    import copy
    ref = copy.deepcopy(tail).to(memory_format=torch.contiguous_format)
    for m in ref.modules():           # fresh running stats are irrelevant in train mode
        if isinstance(m, nn.BatchNorm2d):
            m.reset_running_stats()
    ra, rb = x.clone().requires_grad_(True), xm.clone().requires_grad_(True)
    z = torch.cat([ra, rb], 1)
    sp = ref.bev_fusion
    z = sp.fuse(torch.cat([sp.conv1x1(z), sp.conv3x3(z), sp.dilated_conv3x3_rate6(z),
                           sp.dilated_conv3x3_rate12(z)], 1))
    lv = []
    for blk in ref.pts_backbone.blocks:
        z = blk(z)
        lv.append(z)
    y = torch.cat([ref.pts_neck.deblocks[i](lv[i]) for i in range(2)], 1)
    y.square().sum().backward()
    torch.testing.assert_close(out[0], y, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(xa.grad, ra.grad, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(xb.grad, rb.grad, rtol=1e-3, atol=1e-4)


def test_registry_builds_second_and_fpn_from_reference_dicts():
    from msmdfusion_amd.registry import build_backbone, build_neck
    bb = build_backbone(dict(type="SECOND", in_channels=256, out_channels=[128, 256],
                             layer_nums=[5, 5], layer_strides=[1, 2],
                             norm_cfg=dict(type="BN", eps=0.001, momentum=0.01),
                             conv_cfg=dict(type="Conv2d", bias=False)))
    assert len(bb.blocks) == 2 and len(bb.blocks[0]) == 18
    nk = build_neck(dict(type="SECONDFPN", in_channels=[128, 128, 256], out_channels=[256] * 3,
                         upsample_strides=[0.5, 1, 2]))
    # fractional stride: a strided conv (second_fpn.py:55-61); stride 1 without
    # use_conv_for_no_stride: a 1x1 deconv
    assert isinstance(nk.deblocks[0][0], nn.Conv2d) and nk.deblocks[0][0].stride == (2, 2)
    assert isinstance(nk.deblocks[1][0], nn.ConvTranspose2d)
    with pytest.raises(ValueError):
        build_neck(dict(type="SECONDFPN", in_channels=[128], out_channels=[256, 256],
                        upsample_strides=[1, 2]))


def test_pack_foreground_layout_on_cpu(fx):
    """The host-side packing (concatenation order, plane ids, dtype rule) needs no GPU."""
    from msmdfusion_amd.image_glue import pack_foreground
    pack = pack_foreground(fx.metas, "cpu")
    n = sum(p.shape[0] for m in fx.metas for p in m["foreground2D_info"]["fg_pixels"])
    assert pack.pixels.shape == (n, 3) and pack.pixels.dtype == torch.float32
    assert pack.points.shape == (n, 15) and pack.lidar2img.shape == (fx.B * fx.cams, 16)
    assert pack.sample_counts == [sum(p.shape[0] for p in m["foreground2D_info"]["fg_pixels"])
                                  for m in fx.metas]
    first = fx.metas[0]["foreground2D_info"]["fg_pixels"][0].shape[0]
    assert int(pack.plane[first - 1]) == 0 and int(pack.plane[first]) == 1
    assert int(pack.plane[-1]) == fx.B * fx.cams - 1
    np.testing.assert_array_equal(pack.pixels[:first].numpy(),
                                  fx.metas[0]["foreground2D_info"]["fg_pixels"][0])
    # float64 pixel arrays stay float64 (numpy would scale them in double)
    metas64 = [dict(m, foreground2D_info=dict(
        m["foreground2D_info"],
        fg_pixels=[p.astype(np.float64) for p in m["foreground2D_info"]["fg_pixels"]]))
        for m in fx.metas]
    assert pack_foreground(metas64, "cpu").pixels.dtype == torch.float64
