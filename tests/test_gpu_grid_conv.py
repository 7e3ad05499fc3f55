"""Dense 2-D convolutions on the sparse-conv kernels (msmdfusion_amd/grid_conv.py) against
plain PyTorch fp32 convolutions, and SPPModuleRows against the reference's own SPPModule
output (tests/golden/image_glue_vectors.npz)."""
import numpy as np
import pytest
import torch
from torch.nn import functional as F

from msmdfusion_amd import synthetic as S

pytestmark = pytest.mark.gpu

GEOMS = [  # cin, cout, k, stride, pad, dil, H, W
    (64, 64, 3, 1, 1, 1, 20, 24),
    (64, 96, 3, 1, 6, 6, 20, 24),        # dilated, 'same'
    (128, 64, 1, 1, 0, 1, 12, 12),
    (64, 128, 3, 2, 1, 1, 20, 24),       # strided: separate backward table
    (32, 32, 5, 1, 2, 1, 9, 11),
    (64, 64, 3, 1, 0, 1, 10, 10),        # 'valid': output smaller than input
    (640, 256, 3, 1, 12, 12, 36, 36),    # an SPP branch
]


@pytest.mark.parametrize("cin,cout,k,stride,pad,dil,H,W", GEOMS)
def test_grid_conv2d_matches_torch(dev, cin, cout, k, stride, pad, dil, H, W):
    from msmdfusion_amd.grid_conv import grid_conv2d, map_of, rows_of
    rng = np.random.RandomState(cin + k + dil)
    B = 2
    x = torch.from_numpy(rng.randn(B, cin, H, W).astype(np.float32)).to(dev)
    w = torch.from_numpy((rng.randn(cout, cin, k, k) / np.sqrt(cin * k * k)).astype(np.float32)).to(dev)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    want = F.conv2d(xr, wr, None, stride, pad, dil)
    g = torch.randn_like(want)
    (want * g).sum().backward()

    xa, wa = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    rows, grid = rows_of(xa.contiguous(memory_format=torch.channels_last))
    y, ogrid = grid_conv2d(rows, grid, wa, stride, pad, dil)
    got = map_of(y, ogrid)
    assert tuple(got.shape) == tuple(want.shape)
    (got * g).sum().backward()

    def close(a, b, tol):
        scale = float(b.detach().abs().max())
        assert float((a - b).abs().max()) <= tol * scale + 1e-7, \
            (float((a - b).abs().max()), scale)
    close(got, want, 1e-5)            # fp32-equivalent (three bf16 planes)
    close(xa.grad, xr.grad, 1e-5)
    close(wa.grad, wr.grad, 2e-5)


def test_spp_module_rows_is_the_reference_spp(dev):
    from image_glue_fixture import Fixture
    from msmdfusion_amd.bev import SPPModule
    from msmdfusion_amd.grid_conv import SPPModuleRows
    fx = Fixture()
    spp = S.seeded_parameters(SPPModuleRows(), seed=13).to(dev).train()
    assert sorted(spp.state_dict()) == sorted(SPPModule().state_dict())
    x = torch.from_numpy(np.random.RandomState(14).standard_normal((2, 640, 12, 12))
                         .astype(np.float32)).to(dev)
    with torch.no_grad():
        y = spp(x)
    np.testing.assert_allclose(y.cpu().numpy(), fx.g["spp_y"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(spp.fuse[1].running_mean.cpu().numpy(),
                               fx.g["spp_running_mean_fuse"], rtol=1e-4, atol=1e-6)


def test_spp_module_rows_gradients(dev):
    """Same parameters through MIOpen (bev.SPPModule) and through the row kernels:
    outputs and gradients agree, on a channels-last input as bev_concat hands it over."""
    import copy
    from msmdfusion_amd.bev import SPPModule
    from msmdfusion_amd.grid_conv import SPPModuleRows
    rows_mod = S.seeded_parameters(SPPModuleRows(), seed=3).to(dev).train()
    ref_mod = SPPModule().to(dev).train()
    ref_mod.load_state_dict(copy.deepcopy(rows_mod.state_dict()))
    x = torch.randn(2, 640, 30, 28, device=dev)
    xa = x.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    ya, yb = rows_mod(xa), ref_mod(xb)
    g = torch.randn_like(yb)
    (ya * g).sum().backward()
    (yb * g).sum().backward()

    def rel(a, b):
        return float((a.detach() - b.detach()).norm() / b.detach().norm())
    assert rel(ya, yb) <= 1e-4
    assert rel(xa.grad, xb.grad) <= 1e-2     # (both sides are fp32 roundings of a BN-conditioned sum)
    for (na, pa), (nb, pb) in zip(rows_mod.named_parameters(), ref_mod.named_parameters()):
        assert na == nb
        assert rel(pa.grad, pb.grad) <= 2e-2, na


def test_second_and_fpn_rows_match_miopen(dev):
    """SECONDRows + SECONDFPNRows (LC config: stride-2 stage, 1x1 conv and 2x2 stride-2
    transposed conv in the neck) against the MIOpen modules with the same parameters."""
    import copy
    from msmdfusion_amd import configs as C
    tail_rows = S.seeded_parameters(C.build_bev_tail(C.MSMDFUSION_LC), seed=5).to(dev).train()
    tail_ref = C.build_bev_tail(C.MSMDFUSION_LC, rows=False).to(dev).train()
    assert sorted(tail_rows.state_dict()) == sorted(tail_ref.state_dict())
    tail_ref.load_state_dict(copy.deepcopy(tail_rows.state_dict()))
    x = torch.randn(2, 256, 36, 40, device=dev)
    xa = x.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    ya = tail_rows.pts_neck(tail_rows.pts_backbone(xa))[0]
    yb = tail_ref.pts_neck(tail_ref.pts_backbone(xb))[0]
    assert tuple(ya.shape) == tuple(yb.shape) == (2, 512, 36, 40)
    g = torch.randn_like(yb)
    (ya * g).sum().backward()
    (yb * g).sum().backward()

    def rel(a, b):
        return float((a.detach() - b.detach()).norm() / b.detach().norm())
    assert rel(ya, yb) <= 1e-4
    assert rel(xa.grad, xb.grad) <= 2e-2
    pa = dict(tail_rows.named_parameters())
    for name, p in tail_ref.named_parameters():
        if name.startswith("pts_neck") or name.endswith("blocks.1.0.weight"):
            assert rel(pa[name].grad, p.grad) <= 2e-2, name


def test_bev_tail_rows_end_to_end(dev):
    """The whole tail on rows from the joint channels-last BEV map == MIOpen fp32."""
    import copy
    from msmdfusion_amd import configs as C
    rows = S.seeded_parameters(C.build_bev_tail(C.MSMDFUSION_LC), seed=7).to(dev).train()
    ref = C.build_bev_tail(C.MSMDFUSION_LC, rows=False).to(dev).train()
    ref.load_state_dict(copy.deepcopy(rows.state_dict()))
    x = torch.randn(2, 640, 36, 36, device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        ya, yb = rows(x)[0], ref(x)[0]
    assert tuple(ya.shape) == (2, 512, 36, 36)
    assert float((ya - yb).norm() / yb.norm()) <= 1e-3      # 25 layers, train-mode BN
    # (no-grad forwards take the conv's bypass, where weights that do not require grad may
    # come from the frozen-weight pack cache: it must never serve one layer's temporary
    # KRSC copy to the next layer)
    with torch.no_grad():
        assert torch.equal(rows(x)[0], ya)


def test_head_on_gpu_after_the_row_tail(dev):
    """LC config end of the chain: joint BEV map -> tail on rows -> TransFusionHead (its
    512 -> 128 shared_conv on the row kernels too): same predictions as the torch path,
    boxes decoded."""
    import copy
    from msmdfusion_amd import configs as C
    torch.manual_seed(0)
    tail = C.build_bev_tail(C.MSMDFUSION_LC).to(dev).eval()
    head_rows = C.build_head(rows=True).to(dev).eval()
    head_ref = C.build_head(rows=False).to(dev).eval()
    head_ref.load_state_dict(copy.deepcopy(head_rows.state_dict()))
    x = torch.randn(2, 640, 180, 180, device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        feat = tail(x)[0]
        a, b = head_rows(feat), head_ref(feat.contiguous())
    (pa,), (pb,) = a[0], b[0]
    assert tuple(pa["center"].shape) == (2, 2, 200) and tuple(pa["heatmap"].shape) == (2, 10, 200)
    # the proposals are an argsort of the heatmap: identical inputs up to rounding give the
    # same top-200 except at near-ties; compare the dense heatmap and, where the query sets
    # agree, the predictions
    assert float((pa["dense_heatmap"] - pb["dense_heatmap"]).abs().max()) <= \
        1e-4 * float(pb["dense_heatmap"].abs().max())
    same = head_rows.query_labels == head_ref.query_labels
    assert float(same.float().mean()) > 0.9
    boxes = head_rows.get_bboxes(a)
    assert len(boxes) == 2 and boxes[0]["bboxes"].shape[1] == 9
    assert torch.isfinite(boxes[0]["bboxes"]).all()


def test_head_heatmap_rows_train_mode_gradients(dev):
    """shared_conv + heatmap_head (ConvModule with batch statistics + class conv with its 10
    outputs padded to 32) on the row kernels against the torch / MIOpen path: logits and the
    gradients of every parameter involved."""
    import copy
    from msmdfusion_amd import configs as C
    torch.manual_seed(1)
    head_rows = C.build_head(rows=True).to(dev).train()
    head_ref = C.build_head(rows=False).to(dev).train()
    head_ref.load_state_dict(copy.deepcopy(head_rows.state_dict()))
    x = torch.randn(2, 512, 60, 60, device=dev)
    w = torch.randn(2, 10, 60, 60, device=dev)
    outs = []
    for head in (head_rows, head_ref):
        feat, rows, grid = head._shared_conv(x)
        assert (rows is None) == (head is head_ref)
        hm = head._heatmap(feat, rows, grid)
        assert hm.is_contiguous() and tuple(hm.shape) == (2, 10, 60, 60)
        (hm * w).mean().backward()
        outs.append(hm.detach())
    scale = float(outs[1].abs().max())
    assert float((outs[0] - outs[1]).abs().max()) <= 2e-4 * scale
    for name in ("shared_conv.weight", "shared_conv.bias", "heatmap_head.0.conv.weight",
                 "heatmap_head.0.bn.weight", "heatmap_head.0.bn.bias", "heatmap_head.1.weight",
                 "heatmap_head.1.bias"):
        ga = dict(head_rows.named_parameters())[name].grad
        gb = dict(head_ref.named_parameters())[name].grad
        rel = float((ga - gb).norm() / gb.norm().clamp(min=1e-12))
        assert rel < 2e-3, (name, rel)
    assert torch.allclose(head_rows.heatmap_head[0].bn.running_var,
                          head_ref.heatmap_head[0].bn.running_var, rtol=1e-4)
