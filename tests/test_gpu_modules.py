"""GPU parity of the module layer (the mirror of the reference's mmdet3d /
spconv API) against the oracle composed the way the reference composes it."""
import numpy as np
import pytest
import torch
from torch import nn

from msmdfusion_amd import synthetic as S
from oracle import oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _np(t):
    return t.detach().cpu().numpy()


class OracleSparse:
    def __init__(self, feat, idx, shape, batch):
        self.feat, self.idx, self.shape, self.batch = feat, idx, list(shape), batch


def oracle_forward(module, x):
    """Walk a module tree with oracle ops: SparseSequential (modules.py:125-137),
    SparseBasicBlock (sparse_block.py:103-126), conv (spconv_ops.h:260-361),
    BatchNorm1d with batch statistics, ReLU."""
    from msmdfusion_amd import spconv
    from msmdfusion_amd.sparse_block import SparseBasicBlock
    if isinstance(module, spconv.SparseSequential):
        for m in module._modules.values():
            x = oracle_forward(m, x)
        return x
    if isinstance(module, SparseBasicBlock):
        identity = x.feat
        out = oracle_forward(module.conv1, x)
        out = oracle_forward(module.norm1, out)
        out = oracle_forward(module.relu, out)
        out = oracle_forward(module.conv2, out)
        out = oracle_forward(module.norm2, out)
        out.feat = np.maximum(out.feat + identity, 0)
        return out
    if isinstance(module, spconv.SparseConvolution):
        w = _np(module.weight_kio()).copy()
        oi, pr, nm, osz = O.get_indice_pairs(x.idx, x.batch, x.shape, module.kernel_size,
                                             module.stride, module.padding, 1, module.subm)
        out = O.indice_conv_fwd(x.feat, w, pr, nm, oi.shape[0], subm=module.subm)
        if not module.subm:   # HIP path orders outputs by linear id
            coi, _, perm = O.canonical_rulebook(oi, pr, nm, osz)
            oi, out = coi, out[perm]
        return OracleSparse(out, oi, osz, x.batch)
    if isinstance(module, nn.BatchNorm1d):
        f = x.feat.astype(np.float64)
        if module.training:
            mean, var = f.mean(0), f.var(0)
        else:
            mean, var = _np(module.running_mean), _np(module.running_var)
        y = (f - mean) / np.sqrt(var + module.eps) * _np(module.weight) + _np(module.bias)
        return OracleSparse(y.astype(np.float32), x.idx, x.shape, x.batch)
    if isinstance(module, nn.ReLU):
        return OracleSparse(np.maximum(x.feat, 0), x.idx, x.shape, x.batch)
    raise TypeError(type(module))


def test_registry_builds_reference_cfg(dev):
    from msmdfusion_amd import spconv
    from msmdfusion_amd.registry import build_conv_layer, build_middle_encoder, build_norm_layer
    conv = build_conv_layer(dict(type="SubMConv3d", indice_key="subm1"), 5, 16, 3, stride=1,
                            padding=1, bias=False)
    assert isinstance(conv, spconv.SubMConv3d) and conv.weight.shape == (16, 3, 3, 3, 5)
    assert conv.bias is None and conv.indice_key == "subm1"
    name, bn = build_norm_layer(dict(type="BN1d", eps=1e-3, momentum=0.01), 16, postfix=1)
    assert name == "bn1" and bn.eps == 1e-3 and bn.momentum == 0.01
    enc = build_middle_encoder(dict(
        type="SparseEncoder", in_channels=5, sparse_shape=[41, 1440, 1440], output_channels=128,
        order=("conv", "norm", "act"),
        encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
        encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, [0, 1, 1]), (0, 0)),
        block_type="basicblock"))
    keys = set(enc.state_dict().keys())
    for k in ["conv_input.0.weight", "conv_input.1.running_mean",
              "encoder_layers.encoder_layer1.0.conv1.weight",
              "encoder_layers.encoder_layer1.0.bn2.weight",
              "encoder_layers.encoder_layer1.2.0.weight",
              "encoder_layers.encoder_layer4.1.conv2.weight", "conv_out.0.weight"]:
        assert k in keys, k
    assert enc.encoder_layers.encoder_layer3[2][0].padding == [0, 1, 1]
    assert enc.conv_out[0].kernel_size == [3, 1, 1] and enc.conv_out[0].stride == [2, 1, 1]
    n_convs = sum(isinstance(m, spconv.SparseConvolution) for m in enc.modules())
    assert n_convs == 21     # 17 SubM + 4 strided


def test_conv_module_grads(dev):
    """SubMConv3d / SparseConv3d modules: forward + autograd against the oracle
    (covers the KRSC <-> [K,Cin,Cout] mapping)."""
    from msmdfusion_amd import spconv
    shape = [9, 40, 40]
    idx = S.random_voxel_indices(800, 2, shape, seed=0)
    rng = np.random.RandomState(0)
    f = rng.randn(idx.shape[0], 16).astype(np.float32)
    for cls, kw in [(spconv.SubMConv3d, dict(kernel_size=3, padding=1)),
                    (spconv.SparseConv3d, dict(kernel_size=3, stride=2, padding=1))]:
        torch.manual_seed(0)
        conv = cls(16, 32, bias=True, indice_key="k", **kw).to(dev)
        x = torch.from_numpy(f).to(dev).requires_grad_(True)
        st = spconv.SparseConvTensor(x, torch.from_numpy(idx).to(dev), shape, 2)
        out = conv(st)
        g = rng.randn(out.features.shape[0], 32).astype(np.float32)
        out.features.backward(torch.from_numpy(g).to(dev))
        w = _np(conv.weight_kio()).copy()
        oi, pr, nm, osz = O.get_indice_pairs(idx, 2, shape, conv.kernel_size, conv.stride,
                                             conv.padding, 1, conv.subm)
        exp = O.indice_conv_fwd(f, w, pr, nm, oi.shape[0], subm=conv.subm)
        perm = np.arange(oi.shape[0])
        if not conv.subm:
            coi, _, perm = O.canonical_rulebook(oi, pr, nm, osz)
            assert np.array_equal(_np(out.indices), coi)
        np.testing.assert_allclose(_np(out.features), exp[perm] + _np(conv.bias), rtol=TOL,
                                   atol=TOL)
        g_or = np.zeros_like(g)
        g_or[perm] = g
        edin, edw = O.indice_conv_bwd(f, w, g_or, pr, nm, subm=conv.subm)
        np.testing.assert_allclose(_np(x.grad), edin, rtol=TOL, atol=TOL)
        dw_krsc = edw.transpose(2, 0, 1).reshape(conv.weight.shape)
        np.testing.assert_allclose(_np(conv.weight.grad), dw_krsc, rtol=TOL, atol=5 * TOL)
        np.testing.assert_allclose(_np(conv.bias.grad), g.sum(0), rtol=TOL, atol=5 * TOL)
        assert out.indice_dict["k"].is_subm == conv.subm


def test_sparse_encoder_forward_matches_oracle(dev):
    from msmdfusion_amd.sparse_encoder import SparseEncoder
    shape = [17, 64, 64]
    torch.manual_seed(1)
    enc = SparseEncoder(5, shape, output_channels=32, base_channels=16,
                        encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64)),
                        encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0)),
                        block_type="basicblock").to(dev).train()
    idx = S.random_voxel_indices(3000, 2, shape, seed=4)
    f = np.random.RandomState(4).randn(idx.shape[0], 5).astype(np.float32)
    bev, feats = enc(torch.from_numpy(f).to(dev), torch.from_numpy(idx).to(dev), 2)
    x = OracleSparse(f, idx, shape, 2)
    x = oracle_forward(enc.conv_input, x)
    exp_feats = [x]
    for layer in enc.encoder_layers:
        x = oracle_forward(layer, x)
        exp_feats.append(x)
    out = oracle_forward(enc.conv_out, exp_feats[-1])
    assert len(feats) == len(exp_feats)
    for got, exp in zip(feats, exp_feats):
        assert np.array_equal(_np(got.indices), exp.idx)
        assert got.spatial_shape == list(exp.shape)
        np.testing.assert_allclose(_np(got.features), exp.feat, rtol=1e-3, atol=2e-4)
    dense = O.dense(out.feat, out.idx, 2, out.shape)
    np.testing.assert_allclose(_np(bev), dense.reshape(2, -1, out.shape[1], out.shape[2]),
                               rtol=1e-3, atol=2e-4)
    # the 12 SubM convs of 3 stages share one rulebook per voxel set
    (bev.sum()).backward()
    assert all(p.grad is not None for p in enc.parameters())


@pytest.mark.parametrize("c,relu,res", [(16, True, False), (80, True, True), (128, False, False),
                                        (192, True, True)])
def test_fused_bn_act_matches_torch(dev, c, relu, res):
    """bn_act == relu(bn(x) + residual) of torch (float64 reference): outputs,
    all gradients, running statistics; training and eval."""
    from msmdfusion_amd.spconv.functional import bn_act
    torch.manual_seed(c)
    n = 3000
    x = (torch.randn(n, c, device=dev) * 2 + 0.5)
    r = torch.randn(n, c, device=dev) if res else None
    g = torch.randn(n, c, device=dev)
    for training in (True, False):
        bn = nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev)
        ref = nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev).double()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
            bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2)
            ref.load_state_dict(bn.state_dict())
        bn.train(training); ref.train(training)
        xa = x.clone().requires_grad_(True)
        ra = r.clone().requires_grad_(True) if res else None
        y = bn_act(xa, bn, relu=relu, residual=ra)
        y.backward(g)
        xb = x.double().requires_grad_(True)
        rb = r.double().requires_grad_(True) if res else None
        yb = ref(xb)
        if res:
            yb = yb + rb
        if relu:
            yb = torch.relu(yb)
        yb.backward(g.double())
        np.testing.assert_allclose(_np(y), _np(yb), rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(_np(xa.grad), _np(xb.grad), rtol=1e-4, atol=2e-5)
        if res:
            np.testing.assert_allclose(_np(ra.grad), _np(rb.grad), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(_np(bn.weight.grad), _np(ref.weight.grad), rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(_np(bn.bias.grad), _np(ref.bias.grad), rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(_np(bn.running_mean), _np(ref.running_mean), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(_np(bn.running_var), _np(ref.running_var), rtol=1e-5, atol=1e-6)
        assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked)
    # frozen layer (tools/train.py:205-211): batch statistics, buffers untouched
    bn = nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev).train()
    bn.track_running_stats = False
    before = bn.running_mean.clone()
    y = bn_act(x, bn, relu=False)
    assert torch.equal(bn.running_mean, before)
    np.testing.assert_allclose(_np(y.mean(0)), _np(bn.bias), atol=1e-4)


@pytest.mark.parametrize("n,c", [(3000, 16), (70001, 128), (1, 64), (130, 96)])
def test_bn_relu_backward_without_y_equals_the_masked_one(dev, n, c):
    """msmd_bn_relu_bwd_f32 (BatchNorm + ReLU, no residual: the ReLU mask recomputed from x with
    the forward pass's arithmetic instead of read from y) gives bit for bit what
    msmd_bn_act_bwd_f32 gives with the forward's y -- training and frozen statistics, values
    around the ReLU threshold included."""
    from msmdfusion_amd import kernels as K
    torch.manual_seed(n + c)
    x = torch.randn(n, c, device=dev) * 2 + 0.3
    dy = torch.randn(n, c, device=dev)
    gamma = torch.empty(c, device=dev).uniform_(-1.5, 1.5)     # (negative scales too)
    beta = torch.empty(c, device=dev).uniform_(-0.5, 0.5)
    rm, rv = torch.randn(c, device=dev), torch.empty(c, device=dev).uniform_(0.5, 2)
    for training in (True, False):
        y, mean, invstd = K.bn_act_forward(x, None, gamma, beta, rm.clone(), rv.clone(), training,
                                           0.01, 1e-3, True)
        assert (y == 0).any() and (y > 0).any()
        dx0, _, dg0, db0 = K.bn_act_backward(x, y, dy, gamma, mean, invstd, training, True, False)
        dx1, dg1, db1 = K.bn_relu_backward(x, dy, gamma, beta, mean, invstd, training)
        assert torch.equal(dx0, dx1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)


def test_voxelization_module_and_vfe(dev):
    from msmdfusion_amd.voxel_encoder import HardSimpleVFE
    from msmdfusion_amd.voxelize import Voxelization
    pts = S.lidar_sweep(2, n_az=400)
    layer = Voxelization(S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, 10, (120000, 160000)).to(dev)
    layer.voxel_size = [v * 2 for v in S.VOXEL_SIZE]   # MSMDFusion.py:475-478 mutates it
    v, c, n = layer(torch.from_numpy(pts).to(dev))
    ev, ec, en = O.hard_voxelize(pts, layer.voxel_size, S.POINT_CLOUD_RANGE, 10, 120000)
    assert np.array_equal(_np(v), ev) and np.array_equal(_np(c), ec) and np.array_equal(_np(n), en)
    vfe = HardSimpleVFE(num_features=5)
    np.testing.assert_allclose(_np(vfe(v, n, c)), O.voxel_mean(ev, en, 5), rtol=1e-6, atol=1e-6)
    m, c2, n2 = layer.forward_mean(torch.from_numpy(pts).to(dev))
    np.testing.assert_allclose(_np(m), O.voxel_mean(ev, en, 5), rtol=1e-6, atol=1e-6)


def test_sparse_add_functional(dev):
    from msmdfusion_amd import spconv
    from msmdfusion_amd.spconv import functional as Fsp
    shape = [11, 50, 50]
    a = S.random_voxel_indices(700, 2, shape, seed=8)
    b = np.concatenate([a[::2], S.random_voxel_indices(500, 2, shape, seed=9)])
    b = b[np.sort(np.unique(b, axis=0, return_index=True)[1])]
    rng = np.random.RandomState(3)
    fa, fb = rng.randn(a.shape[0], 96).astype(np.float32), rng.randn(b.shape[0], 96).astype(np.float32)
    ta = spconv.SparseConvTensor(torch.from_numpy(fa).to(dev).requires_grad_(True),
                                 torch.from_numpy(a).to(dev), shape, 2)
    tb = spconv.SparseConvTensor(torch.from_numpy(fb).to(dev).requires_grad_(True),
                                 torch.from_numpy(b).to(dev), shape, 2)
    out = Fsp.sparse_add(ta, tb)
    eoi, eof, ma, mb = O.sparse_add(fa, a, fb, b, shape)
    assert np.array_equal(_np(out.indices), eoi)
    np.testing.assert_allclose(_np(out.features), eof, rtol=1e-6, atol=1e-6)
    # dense cross-check (independent pin): dense(a)+dense(b) == dense(a+b)
    d = O.dense(fa, a, 2, shape) + O.dense(fb, b, 2, shape)
    np.testing.assert_allclose(_np(out.dense()), d, rtol=1e-6, atol=1e-6)
    g = rng.randn(*eof.shape).astype(np.float32)
    out.features.backward(torch.from_numpy(g).to(dev))
    np.testing.assert_allclose(_np(ta.features.grad), g[ma])
    np.testing.assert_allclose(_np(tb.features.grad), g[mb])


def test_pack_weight_split_many_equals_single_packs(dev):
    """msmd_spconv_pack_weight_split_many: several weights (KRSC and [K,Cin,Cout] layouts,
    with and without the transposed image, ragged channel counts) in one launch == the images
    of msmd_spconv_pack_weight_split[_pair], byte for byte."""
    from msmdfusion_amd import kernels as K
    g = torch.Generator(device=dev).manual_seed(3)
    shapes = [((96, 3, 3, 3, 80), True, True), ((27, 64, 128), False, True),
              ((192, 3, 1, 1, 192), True, False), ((27, 40, 36), False, True),
              ((32, 3, 3, 3, 32), True, True)]
    for planes in (3, 2, 1):
        jobs, want = [], []
        for shape, krsc, pair in shapes:
            w = torch.randn(*shape, device=dev, generator=g)
            if krsc:
                cout, cin = shape[0], shape[-1]
                kvol = w.numel() // (cout * cin)
            else:
                kvol, cin, cout = shape
            a = torch.full((K.lib.msmd_spconv_packed_split_bytes(kvol, cin, cout, planes),), 7,
                           dtype=torch.uint8, device=dev)
            b = torch.full((K.lib.msmd_spconv_packed_split_bytes(kvol, cout, cin, planes),), 7,
                           dtype=torch.uint8, device=dev) if pair else None
            jobs.append((w, krsc, a, b))
            want.append((K.pack_weight_split(w, planes, krsc=krsc),
                         K.pack_weight_split(w, planes, transpose=True, krsc=krsc) if pair else None))
        K.pack_weight_split_many(jobs, planes)
        for (_, _, a, b), (ea, eb) in zip(jobs, want):
            assert torch.equal(a, ea)
            if b is not None:
                assert torch.equal(b, eb)


def test_packed_parameter_images_follow_the_optimizer(dev):
    """The packed images of module parameters are kept between forwards and refreshed --
    all stale ones in one launch -- when a parameter's version has moved: a training loop
    with an in-place update between steps sees every update, a frozen weight is packed once,
    and invalidate_packed_weights() covers writes through .data."""
    from msmdfusion_amd import spconv
    from msmdfusion_amd.spconv import functional as Fsp
    shape = [9, 40, 40]
    idx = S.random_voxel_indices(900, 2, shape, seed=4)
    torch.manual_seed(0)
    convs = [spconv.SubMConv3d(32, 32, 3, padding=1, bias=False).to(dev) for _ in range(3)]
    convs[2].weight.requires_grad = False            # frozen
    x0 = torch.randn(idx.shape[0], 32, device=dev)
    t_idx = torch.from_numpy(idx).to(dev)

    def forward():
        x = spconv.SparseConvTensor(x0.clone().requires_grad_(True), t_idx, shape, 2)
        for c in convs:
            x = c(x)
        return x.features

    def reference():     # the same convs with nothing cached
        Fsp.invalidate_packed_weights()
        out = forward()
        Fsp.invalidate_packed_weights()
        return out

    assert torch.equal(forward(), reference())
    y0 = forward()                                   # (reference() left the cache empty)
    assert torch.equal(forward(), y0)                # served from the cache: same result
    frozen_entry = Fsp._PACKS[id(convs[2].weight)]
    frozen_version = frozen_entry["version"]
    y0.square().mean().backward()
    with torch.no_grad():                            # "optimizer step": in-place, bumps _version
        for c in convs[:2]:
            c.weight.add_(0.05 * torch.randn_like(c.weight))
    y1 = forward()
    assert not torch.equal(y1, y0)
    # (the frozen weight's image was not touched by the refresh: same entry, same version)
    assert Fsp._PACKS[id(convs[2].weight)] is frozen_entry
    assert frozen_entry["version"] == frozen_version
    assert torch.equal(y1, reference())
    y1b = forward()
    assert torch.equal(y1b, y1)
    # gradients of the refreshed weights agree with a cache-free evaluation
    for c in convs:
        c.weight.grad = None
    y1b.square().mean().backward()
    got = [c.weight.grad.clone() for c in convs[:2]]
    for c in convs:
        c.weight.grad = None
    reference().square().mean().backward()
    for g_, c in zip(got, convs[:2]):
        assert torch.equal(g_, c.weight.grad)
    # a write through .data does not bump the version: the documented invalidate call does
    convs[0].weight.data.mul_(1.5)
    Fsp.invalidate_packed_weights()
    assert torch.equal(forward(), reference())
    # ... for one parameter, and a module's train() / eval() does it for its own weight
    # (EMA / fp16 hooks swap parameters through .data at those boundaries)
    y2 = forward()
    convs[1].weight.data.mul_(0.5)
    assert torch.equal(forward(), y2)                # stale image: the documented exposure
    Fsp.invalidate_packed_weights(convs[1].weight)
    y3 = forward()
    assert not torch.equal(y3, y2) and torch.equal(y3, reference())
    forward()
    entry = Fsp._PACKS[id(convs[0].weight)]
    buf = entry["packed"].data_ptr()
    convs[0].weight.data.mul_(2.0)
    convs[0].eval()
    convs[0].train()
    # the toggle marks the image stale; it is repacked into the allocation it already has
    assert entry["version"] is None
    y4 = forward()
    assert Fsp._PACKS[id(convs[0].weight)] is entry and entry["packed"].data_ptr() == buf
    assert torch.equal(y4, reference())


def test_plan_batch_equals_table_by_table_planning(dev, monkeypatch):
    """SparseEncoder.plan() inside plan_batch (every table's tiling / prefixes / pair lists /
    segment table computed by ONE msmd_rulebook_plan_many call when the context closes) leaves
    on every rulebook exactly what IndiceData.prepare() computes table by table
    (MSMD_PLAN_BATCH=0)."""
    from msmdfusion_amd import configs as C
    from msmdfusion_amd.spconv import core
    _, _, enc, _ = C.build_hot_path(C.MSMDFUSION_LC)
    enc = enc.to(dev).train()
    idx = torch.from_numpy(S.random_voxel_indices(9000, 2, enc.sparse_shape, seed=3)).to(dev)
    assert core.PLAN_BATCHING
    monkeypatch.setattr(core, "PLAN_SCOPE", "stage")         # the whole encoder = one launch set
    batched, stages_b = enc.plan(idx, 2)
    assert getattr(core._PLAN, "batch", None) is None        # context closed, all flushed
    monkeypatch.setattr(core, "PLAN_BATCHING", False)
    single, stages_s = enc.plan(idx, 2)
    for (ia, sa), (ib, sb) in zip(stages_b, stages_s):
        assert sa == sb and torch.equal(ia, ib)
    rbs_b, rbs_s = list(batched._rb_cache.values()), list(single._rb_cache.values())
    assert len(rbs_b) == len(rbs_s) >= 8
    planned = 0
    for got, rb in zip(rbs_b, rbs_s):
        assert got is not rb and torch.equal(got.nbr_fwd, rb.nbr_fwd)
        for f in ("_order_fwd", "_tiled_fwd", "_order_bwd", "_tiled_bwd"):
            a, b = getattr(got, f), getattr(rb, f)
            assert (a is None) == (b is None), f
            if a is not None:
                assert torch.equal(a[0], b[0]), f
                planned += 1
        for pa, pb in ((got._prefix_fwd, rb._prefix_fwd), (got._prefix_bwd, rb._prefix_bwd)):
            assert set(pa) == set(pb)
            for h in pa:
                assert torch.equal(pa[h], pb[h])
        assert (got._pairs is None) == (rb._pairs is None)
        if rb._pairs is not None:
            assert torch.equal(got._pairs[0], rb._pairs[0])
            assert torch.equal(got._pairs[1], rb._pairs[1])
        sa, sb = got._pair_segments, rb._pair_segments
        assert type(sa) == type(sb), (sa, sb)
        if isinstance(sb, tuple):
            assert sa[1] == sb[1] and torch.equal(sa[0], sb[0])
    assert planned >= 16
