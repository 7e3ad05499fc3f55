"""GPU parity of rows a13 (image half), f2, the channels-last BEV hand-over and
the dense tail f1: the HIP path against the reference's own outputs
(tests/golden/image_glue_vectors.npz) and, at production sizes, against
oracle/image_glue.py."""
import numpy as np
import pytest
import torch

from msmdfusion_amd import synthetic as S
from oracle import image_glue as OI

from image_glue_fixture import Fixture

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fx():
    return Fixture()


def _score_net(fx, dev):
    from msmdfusion_amd.image_glue import ScoreNet
    net = ScoreNet(fx.c_out + 17)
    net.load_state_dict(fx.state_dict("score_net."))
    return net.to(dev)


@pytest.mark.parametrize("channels_last", [False, True])
def test_get_foreground2d_matches_the_reference(dev, fx, channels_last):
    from msmdfusion_amd.image_glue import get_foreground2D, pack_foreground
    net = _score_net(fx, dev)
    pack = pack_foreground(fx.metas, dev)           # packed once, used by all four scales
    for i, feat in enumerate(fx.fg_inputs):
        f = torch.from_numpy(feat).to(dev)
        if channels_last:
            f = f.contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            got = get_foreground2D(f, fx.metas, net, pack=pack)
        assert len(got) == fx.B
        for b in range(fx.B):
            want = fx.fg[i][b]
            g = got[b].cpu().numpy()
            assert g.shape == want.shape
            np.testing.assert_array_equal(g[:, :15], want[:, :15])
            np.testing.assert_allclose(g[:, 15:], want[:, 15:], rtol=1e-4, atol=1e-5)


def _random_metas(rng, B, cams, H, W, n_per_cam, dtype=np.float32, n_real=None):
    metas = []
    for b in range(B):
        pix, pts, real, l2i = [], [], [], []
        for j in range(cams):
            n = int(n_per_cam * (0.5 + rng.rand()))
            p = np.stack([rng.rand(n) * (W - 1e-3), rng.rand(n) * (H - 1e-3),
                          1 + rng.rand(n) * 50], 1)
            pix.append(p.astype(dtype))
            pts.append(rng.randn(n, 15).astype(np.float32))
            m = n_real if n_real is not None else n // 4
            r = np.stack([rng.randint(0, W, m), rng.randint(0, H, m), 1 + rng.rand(m) * 50], 1)
            real.append(r.astype(dtype))
            l2i.append(rng.randn(4, 4).astype(np.float32))
        metas.append(dict(foreground2D_info=dict(fg_pixels=pix, fg_points=pts, fg_real_pixels=real),
                          lidar2img=l2i, input_shape=(H, W), pad_shape=(H, W, 3)))
    return metas


@pytest.mark.parametrize("B", [1, 2, 3, 4])
def test_get_foreground2d_reference_write_back(dev, B):
    """reference_quirks=True: the score-scaled channels reach sample 0 always and sample 1
    only when B == 2 (MSMDFusion.py:226-234 scales a torch.cat copy and writes back
    `[0]` and `if B == 2: [1]`); every other sample keeps the unscaled gather.  Against the
    oracle's transcription of those lines; the default mode scales every sample."""
    from msmdfusion_amd.image_glue import ScoreNet, get_foreground2D
    rng = np.random.RandomState(40 + B)
    cams, H, W = 3, 64, 96
    metas = _random_metas(rng, B, cams, H, W, 200)
    feat = rng.randn(B * cams, 49, H // 4, W // 4).astype(np.float32)
    torch.manual_seed(2)
    net = ScoreNet().to(dev)
    with torch.no_grad():
        net[0].bias.fill_(0.7)
        f = torch.from_numpy(feat).to(dev)
        quirk = get_foreground2D(f, metas, net, reference_quirks=True)
        plain = get_foreground2D(f, metas, net)
    w, b = net[0].weight.detach().cpu().numpy(), net[0].bias.detach().cpu().numpy()
    want_q = OI.get_foreground2d(feat, metas, w, b, reference_write_back=True)
    want_p = OI.get_foreground2d(feat, metas, w, b)
    raw = OI.get_foreground2d(feat, metas, np.zeros_like(w), np.ones_like(b))
    for bi in range(B):
        np.testing.assert_allclose(plain[bi].cpu().numpy(), want_p[bi], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(quirk[bi].cpu().numpy(), want_q[bi], rtol=1e-4, atol=1e-5)
        if bi == 0 or (bi == 1 and B == 2):
            assert not np.array_equal(want_q[bi], raw[bi])
        else:       # the unscaled gather, bit for bit
            np.testing.assert_array_equal(quirk[bi].cpu().numpy(), raw[bi])


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_fused_scored_gather_equals_the_op_chain(dev, layout, monkeypatch):
    """msmd_fg_gather_scored_f32 (no-gradient path: gather + score_net + scaling in one launch)
    == the differentiable chain gather -> Linear + ReLU -> cat / multiply: point block and
    channels of a zero score bit for bit, scaled channels to fp32 rounding of the 66-term dot
    product; bad pixels counted the same; n_scaled leaves the tail unscaled."""
    from msmdfusion_amd import image_glue as G
    from msmdfusion_amd import kernels as K
    rng = np.random.RandomState(77)
    B, cams, H, W = 3, 4, 96, 160
    metas = _random_metas(rng, B, cams, H, W, 700)
    feat = torch.from_numpy(rng.randn(B * cams, 49, H // 4, W // 4).astype(np.float32)).to(dev)
    if layout == "nhwc":
        feat = feat.contiguous(memory_format=torch.channels_last)
    torch.manual_seed(4)
    net = G.ScoreNet().to(dev)
    with torch.no_grad():
        net[0].bias.fill_(0.3)
        pack = G.pack_foreground(metas, dev)
        fused = G.get_foreground2D(feat, metas, net, pack=pack)
        monkeypatch.setattr(G, "_FG_FUSED", False)
        chain = G.get_foreground2D(feat, metas, net, pack=pack)
        monkeypatch.setattr(G, "_FG_FUSED", True)
        for a, b in zip(fused, chain):
            assert a.shape == b.shape and torch.equal(a[:, :15], b[:, :15])
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
            zero = (b[:, 15:] == 0).all(1)                   # ReLU cut the score to 0
            assert torch.equal(a[zero], b[zero])
        assert sum(int(((b[:, 15:] == 0).all(1)).sum()) for b in chain) > 10
        # n_scaled: the tail keeps the plain gather
        n0 = pack.sample_counts[0]
        part, bad = K.fg_gather_scored(feat, pack.pixels, pack.plane, 0.25, pack.points,
                                       pack.lidar2img, net[0].weight, net[0].bias, n_scaled=n0)
        raw = K.fg_gather(feat, pack.pixels, pack.plane, 0.25, pack.points, pack.lidar2img)[0]
        assert int(bad) == 0
        assert torch.equal(part[n0:], raw[n0:])
        torch.testing.assert_close(part[:n0], torch.cat(chain)[:n0], rtol=1e-5, atol=1e-6)
        # an out-of-map pixel is counted (and raises through the module path)
        metas[1]["foreground2D_info"]["fg_pixels"][2][5, 0] = W + 40.0
        with pytest.raises(IndexError):
            G.get_foreground2D(feat, metas, net)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_get_foreground2d_production_size_vs_oracle(dev, dtype):
    """nuScenes LC shape: 2 samples x 6 cameras, 448x800 input, stride-8 map
    (56x100), 49 channels, ~50k points per sample.  Gathered channels and the
    point block are bit-exact (copies); the scaled block is fp32 arithmetic."""
    from msmdfusion_amd.image_glue import ScoreNet, get_foreground2D
    rng = np.random.RandomState(5)
    B, cams, H, W = 2, 6, 448, 800
    metas = _random_metas(rng, B, cams, H, W, 8300, dtype)
    feat = rng.randn(B * cams, 49, H // 8, W // 8).astype(np.float32)
    torch.manual_seed(1)
    net = ScoreNet().to(dev)
    with torch.no_grad():
        net[0].bias.fill_(1.0)
        got = get_foreground2D(torch.from_numpy(feat).to(dev), metas, net)
    w, b = net[0].weight.detach().cpu().numpy(), net[0].bias.detach().cpu().numpy()
    want = OI.get_foreground2d(feat, metas, w, b)
    # the unscaled gather, bit for bit (score == 1 via a zero weight, bias 1)
    with torch.no_grad():
        net[0].weight.zero_()
        raw = get_foreground2D(torch.from_numpy(feat).to(dev), metas, net)
    raw_want = OI.get_foreground2d(feat, metas, np.zeros_like(w), np.ones_like(b))
    for bi in range(B):
        assert got[bi].shape[0] > 40000
        np.testing.assert_array_equal(raw[bi].cpu().numpy(), raw_want[bi])
        np.testing.assert_allclose(got[bi].cpu().numpy(), want[bi], rtol=1e-4, atol=1e-5)


def test_get_foreground2d_backward_is_index_backward(dev, fx):
    """d/d(feature map) and d/d(score_net) == autograd through plain torch
    indexing, as the reference writes it (MSMDFusion.py:213-228)."""
    from msmdfusion_amd.image_glue import get_foreground2D, pack_foreground
    net = _score_net(fx, dev)
    feat = torch.from_numpy(fx.fg_inputs[2]).to(dev)
    fa = feat.clone().requires_grad_(True)
    out = get_foreground2D(fa, fx.metas, net)
    wts = [torch.randn_like(o) for o in out]
    sum((o * w).sum() for o, w in zip(out, wts)).backward()
    g_net = [p.grad.clone() for p in net.parameters()]
    net.zero_grad()

    fb = feat.clone().requires_grad_(True)
    pack = pack_foreground(fx.metas, dev)
    scale = feat.shape[-1] / fx.W
    cell = (pack.pixels * scale).long()
    f = fb[pack.plane.long(), :, cell[:, 1], cell[:, 0]]
    score_in = torch.cat([f, pack.pixels[:, 2:3], pack.lidar2img[pack.plane.long()]], 1)
    full = torch.cat([pack.points, f * net(score_in)], 1)
    ref = torch.split(full, pack.sample_counts, 0)
    sum((o * w).sum() for o, w in zip(ref, wts)).backward()
    for o, r in zip(out, ref):
        torch.testing.assert_close(o, r, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(fa.grad, fb.grad, rtol=1e-4, atol=1e-5)
    for a, p in zip(g_net, net.parameters()):
        torch.testing.assert_close(a, p.grad, rtol=1e-4, atol=1e-5)


def test_out_of_range_pixels_raise(dev, fx):
    from msmdfusion_amd.image_glue import get_foreground2D, sparse_depth_canvas
    import copy
    metas = copy.deepcopy(fx.metas)
    metas[0]["foreground2D_info"]["fg_pixels"][2][0, 0] = 10.0 * fx.W     # way off the map
    with pytest.raises(IndexError):
        get_foreground2D(torch.from_numpy(fx.fg_inputs[0]).to(dev), metas, _score_net(fx, dev))
    metas = copy.deepcopy(fx.metas)
    metas[1]["foreground2D_info"]["fg_real_pixels"][0][0, 1] = float(fx.H)
    with pytest.raises(IndexError):
        sparse_depth_canvas(metas, fx.H, fx.W, dev)
    # a negative cell wraps like torch indexing does
    metas = copy.deepcopy(fx.metas)
    metas[1]["foreground2D_info"]["fg_real_pixels"][0][0, :2] = (-1.0, -2.0)
    canvas = sparse_depth_canvas(metas, fx.H, fx.W, dev)
    np.testing.assert_array_equal(canvas.cpu().numpy(), OI.depth_canvas(metas, fx.H, fx.W, fx.cams))


def test_depth_aware_channel_compression_matches_the_reference(dev, fx):
    from msmdfusion_amd.image_glue import DepthAwareChannelCompression, sparse_depth_canvas
    canvas = sparse_depth_canvas(fx.metas, fx.H, fx.W, dev)
    np.testing.assert_array_equal(canvas.cpu().numpy(),
                                  OI.depth_canvas(fx.metas, fx.H, fx.W, fx.cams))
    mod = DepthAwareChannelCompression(in_channels=fx.c_img, out_channels=fx.c_out)
    mod.load_state_dict({"conv1x1_blocks." + k: v
                         for k, v in fx.state_dict("conv1x1_blocks.").items()})
    mod = mod.to(dev).train()
    with torch.no_grad():
        out = mod([torch.from_numpy(f).to(dev) for f in fx.feats], fx.metas)
    for i, o in enumerate(out):
        np.testing.assert_allclose(o.cpu().numpy(), fx.comp[i], rtol=1e-4, atol=2e-5)
    # the detector's own layer sizes and keys
    full = DepthAwareChannelCompression()
    assert tuple(full.state_dict()["conv1x1_blocks.2.0.weight"].shape) == (49, 257, 3, 3)
    assert tuple(full.state_dict()["conv1x1_blocks.0.0.weight"].shape) == (49, 257, 5, 5)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_depth_canvas_production_size_bit_exact(dev, dtype):
    """2 x 6 cameras, 448 x 800, ~9k real pixels per camera squeezed into a
    quarter of the image so that thousands of pixels collide: the highest row
    wins on every one of them."""
    from msmdfusion_amd.image_glue import sparse_depth_canvas
    rng = np.random.RandomState(9)
    metas = _random_metas(rng, 2, 6, 224, 400, 10, dtype, n_real=9000)
    for m in metas:
        m["pad_shape"] = (448, 800, 3)
    got = sparse_depth_canvas(metas, 448, 800, dev).cpu().numpy()
    want = OI.depth_canvas(metas, 448, 800, 6)
    np.testing.assert_array_equal(got, want)
    assert (want != 0).sum() < 12 * 9000 - 1000          # collisions did happen


def test_bev_concat_matches_oracle_and_dense(dev):
    """LC hand-over shapes: conv_out [n,128] on (2,180,180) and the fusion stack's
    [n,192] on (2,180,180) -> [B,640,180,180], bit-exact; backward = gather."""
    from msmdfusion_amd import spconv
    from msmdfusion_amd.spconv import functional as Fsp
    rng = np.random.RandomState(2)
    B, shape = 2, [2, 180, 180]
    parts, sp = [], []
    for c, n in [(128, 21000), (192, 38000)]:
        idx = S.random_voxel_indices(n, B, shape, seed=c)
        feat = rng.randn(idx.shape[0], c).astype(np.float32)
        parts.append((feat, idx, tuple(shape)))
        sp.append(spconv.SparseConvTensor(torch.from_numpy(feat).to(dev).requires_grad_(True),
                                          torch.from_numpy(idx).to(dev), shape, B))
    joint = Fsp.bev_concat(sp)
    assert tuple(joint.shape) == (B, 640, 180, 180)
    assert joint.is_contiguous(memory_format=torch.channels_last)
    want = OI.bev_concat(parts, B)
    np.testing.assert_array_equal(joint.detach().cpu().numpy(), want)
    ref = torch.cat([t.dense().view(B, -1, 180, 180) for t in sp], 1)
    assert torch.equal(ref, joint)
    g = torch.randn_like(ref)
    (joint * g).sum().backward()
    got = [t.features.grad.clone() for t in sp]
    for t in sp:
        t.features.grad = None
    (ref * g).sum().backward()
    for a, t in zip(got, sp):
        assert torch.equal(a, t.features.grad)


def test_bev_tail_on_gpu_matches_cpu_fp32(dev):
    """SPP + SECOND + SECONDFPN with the LC widths on a 36x36 map: the GPU
    (MIOpen, channels-last) result against PyTorch CPU fp32; bf16 autocast within
    bf16 tolerance."""
    from msmdfusion_amd import configs as C
    tail = S.seeded_parameters(C.build_bev_tail(C.MSMDFUSION_LC, rows=False), seed=3).train()
    x = torch.from_numpy(np.random.RandomState(4).standard_normal((2, 640, 36, 36))
                         .astype(np.float32))
    import copy
    cpu = copy.deepcopy(tail)
    xc = x.clone().requires_grad_(True)
    yc = cpu(xc)[0]
    yc.square().mean().backward()
    gpu = copy.deepcopy(tail).to(dev)
    nhwc = copy.deepcopy(tail).to(dev)
    nhwc.channels_last = True
    nhwc = nhwc.to(memory_format=torch.channels_last)
    with torch.no_grad():
        yn = nhwc(x.to(dev))[0]
    xg = x.to(dev).requires_grad_(True)
    yg = gpu(xg)[0]
    assert tuple(yg.shape) == (2, 512, 36, 36)
    yg.square().mean().backward()
    scale = float(yc.abs().max())
    assert float((yg.cpu() - yc).abs().max()) <= 2e-4 * scale + 1e-5
    assert float((yn.cpu() - yc).abs().max()) <= 2e-4 * scale + 1e-5
    # gradients through 13 train-mode BN layers cancel heavily (|dx| ~ 1e-5 here): compare
    # in the L2 sense
    def rel(a, b):
        return float((a - b).norm() / b.norm())
    assert rel(xg.grad.cpu(), xc.grad) <= 2e-2
    assert rel(gpu.bev_fusion.conv3x3[0].weight.grad.cpu(),
               cpu.bev_fusion.conv3x3[0].weight.grad) <= 2e-2
    assert rel(gpu.pts_neck.deblocks[1][0].weight.grad.cpu(),
               cpu.pts_neck.deblocks[1][0].weight.grad) <= 1e-3
    half = copy.deepcopy(tail).to(dev)
    half.compute_dtype = torch.bfloat16
    with torch.no_grad():
        yh = half(x.to(dev))[0].float().cpu()
    # (torch + MIOpen in bf16 through 13 conv / BN layers, not a kernel of this repo: a sanity
    # bound.  MIOpen picks its algorithms per box; 0.064 * scale has been seen.)
    assert float((yh - yc).abs().max()) <= 0.1 * scale


def test_sparse_path_joint_bev_equals_cat(dev):
    """SparseFusionPath(joint_bev=True) == cat([x, x_mm], 1) of the default
    outputs, bit for bit, and feeds BevTail."""
    from msmdfusion_amd import configs as C
    from msmdfusion_amd.fusion import SparseFusionPath
    torch.manual_seed(0)
    vox, _, enc, mm = C.build_hot_path(C.MSMDFUSION_LC)
    path = SparseFusionPath(vox, enc, mm).to(dev).train()
    fixed = {c: torch.rand(1, c) for c in (16, 32, 64, 128)}
    mm.dummy_embedding_fn = lambda c, device: fixed[c].to(device)
    pts = [torch.from_numpy(S.lidar_sweep(i, n_az=300)).to(dev) for i in range(2)]
    virt = [torch.from_numpy(S.virtual_points(i, n=12000)).to(dev) for i in range(2)]
    with torch.no_grad():
        x, x_mm = path(pts, [virt] * 4)
        joint = path(pts, [virt] * 4, joint_bev=True)
    assert torch.equal(torch.cat([x, x_mm], 1), joint)
    tail = C.build_bev_tail(C.MSMDFUSION_LC).to(dev).train()      # (the row kernels)
    joint = path(pts, [virt] * 4, joint_bev=True)
    out = tail(joint)[0]
    assert tuple(out.shape) == (2, 512, 180, 180)
    out.float().mean().backward()
    assert any(p.grad is not None and p.grad.abs().sum() > 0
               for n, p in path.named_parameters() if "gate_control" in n)
