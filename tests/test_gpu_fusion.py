"""GPU parity of the GMA-Conv / multimodal path
against the oracle composed step by step as the reference composes it
(sparse_multimodal_encoder_painting.py:276-459, MSMDFusion.py:251-325)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from msmdfusion_amd import synthetic as S
from oracle import oracle as O
from test_gpu_modules import OracleSparse, _np, oracle_forward

pytestmark = pytest.mark.gpu


def _oracle_fps_nn(query, key, fps_num, radius, max_cluster, thresh):
    """fps_NN_fast (:276-323) with oracle pieces."""
    nq = query.shape[0]
    if nq <= fps_num:
        return O.nn_search(query[:, 1:], key[:, 1:], thresh)
    q = query[:, 1:].astype(np.float32)[None]
    rep_idx = O.furthest_point_sample(q, fps_num)[0]
    rep = query[rep_idx, 1:]
    rep_nn = O.nn_search(rep, key[:, 1:], thresh)
    grp = O.ball_query(0, radius, max_cluster, q, rep.astype(np.float32)[None])[0]
    return O.nn_assign(grp, rep_nn, nq)


def test_fps_nn_fast(dev):
    from msmdfusion_amd.multimodal_encoder import fps_nn_fast
    rng = np.random.RandomState(0)
    shape = [41, 400, 400]
    key = S.random_voxel_indices(9000, 1, shape, seed=1)
    for nq, fps_num in [(500, 2048), (6000, 512)]:
        query = S.random_voxel_indices(nq, 1, shape, seed=2 + nq)
        exp = _oracle_fps_nn(query, key, fps_num, 6, 50, 13.3)
        got = fps_nn_fast(torch.from_numpy(query).to(dev), torch.from_numpy(key).to(dev),
                          fps_num, 6, 50, 13.3)
        assert np.array_equal(_np(got), exp)
        assert (exp >= 0).any() and (exp < 0).any() or nq <= fps_num


def _make_inputs(dev, shape, n3, n2, c3, batch, seed):
    rng = np.random.RandomState(seed)
    i3 = S.random_voxel_indices(n3, batch, shape, seed=seed)
    extra = S.random_voxel_indices(n2, batch, shape, seed=seed + 50)
    i2 = np.concatenate([i3[::5], extra])
    i2 = i2[np.sort(np.unique(i2, axis=0, return_index=True)[1])]
    # group rows by sample (the reference's tensors always are)
    i3 = i3[np.argsort(i3[:, 0], kind="stable")]
    i2 = i2[np.argsort(i2[:, 0], kind="stable")]
    f3 = rng.randn(i3.shape[0], c3).astype(np.float32)
    f2 = rng.randn(i2.shape[0], 64).astype(np.float32)
    return i3, f3, i2, f2


def test_modality_split_function(dev):
    from msmdfusion_amd import spconv
    from msmdfusion_amd.fusion import voxel_modality_split
    shape = [21, 100, 100]
    i3, f3, i2, f2 = _make_inputs(dev, shape, 2500, 1500, 32, 2, 3)
    a = spconv.SparseConvTensor(torch.from_numpy(f3).to(dev), torch.from_numpy(i3).to(dev), shape, 2)
    b = spconv.SparseConvTensor(torch.from_numpy(f2).to(dev), torch.from_numpy(i2).to(dev), shape, 2)
    a, b, s3, s2 = voxel_modality_split(a, b, 2)
    assert a.indices.shape[1] == 5 and b.indices.shape[1] == 5
    e3, e2, p3, p2 = [], [], [], []
    for bi in range(2):
        r3, r2 = np.flatnonzero(i3[:, 0] == bi), np.flatnonzero(i2[:, 0] == bi)
        m3, m2, q3, q2 = O.modality_split(i3[r3, 1:], i2[r2, 1:], shape)
        e3.append(m3); e2.append(m2); p3.append(r3[q3]); p2.append(r2[q2])
    assert np.array_equal(_np(a.indices)[:, 1], np.concatenate(e3))
    assert np.array_equal(_np(b.indices)[:, 1], np.concatenate(e2))
    assert np.array_equal(_np(a.indices)[:, [0, 2, 3, 4]], i3)
    assert np.array_equal(_np(s3), np.concatenate(p3)) and np.array_equal(_np(s2), np.concatenate(p2))
    # matched rows really are the same voxel
    assert np.array_equal(i3[_np(s3)], i2[_np(s2)])
    # the reference's float32 keys agree with exact keys where they cannot alias (z<=15, x<1000)
    lo3, lo2 = i3[(i3[:, 0] == 0) & (i3[:, 1] <= 15)], i2[(i2[:, 0] == 0) & (i2[:, 1] <= 15)]
    fm = O.modality_split(lo3[:, 1:], lo2[:, 1:], shape, float_keys=True)
    xm = O.modality_split(lo3[:, 1:], lo2[:, 1:], shape, float_keys=False)
    assert all(np.array_equal(x, y) for x, y in zip(fm, xm))


@pytest.mark.gpu
def test_modality_split_float_keys_reproduce_the_reference_aliasing(dev):
    """voxel_modality_split(float_keys=True) == the oracle's float-key restatement of
    MSMDFusion.py:271-272 + type_assign (:27-45), bit for bit, on the stage-0 grid where the
    keys DO alias (z up to 40 >= 17, x up to 1439 >= 1000): the false "mixed" matches a
    reference-trained checkpoint has seen -- and they differ from the exact-key default."""
    from msmdfusion_amd import spconv
    from msmdfusion_amd.fusion import voxel_modality_split
    shape = [41, 1440, 1440]
    rng = np.random.RandomState(11)

    def cloud(n, b):      # dense enough around a few (z, y) lines for aliases and true matches
        z = rng.randint(17, 41, n)
        y = rng.randint(100, 104, n)
        x = rng.randint(990, 1440, n)
        u = np.unique(np.stack([np.full(n, b), z, y, x], 1), axis=0)
        return u[rng.permutation(u.shape[0])].astype(np.int32)
    i3 = np.concatenate([cloud(3000, 0), cloud(2500, 1)])
    i2 = np.concatenate([cloud(2000, 0), cloud(3500, 1)])
    f3 = rng.randn(i3.shape[0], 16).astype(np.float32)
    f2 = rng.randn(i2.shape[0], 64).astype(np.float32)

    def run(float_keys):
        a = spconv.SparseConvTensor(torch.from_numpy(f3).to(dev), torch.from_numpy(i3).to(dev), shape, 2)
        b = spconv.SparseConvTensor(torch.from_numpy(f2).to(dev), torch.from_numpy(i2).to(dev), shape, 2)
        a, b, s3, s2 = voxel_modality_split(a, b, 2, float_keys=float_keys)
        return _np(a.indices)[:, 1], _np(b.indices)[:, 1], _np(s3), _np(s2)
    got = run(True)
    e3, e2, p3, p2 = [], [], [], []
    for bi in range(2):
        r3, r2 = np.flatnonzero(i3[:, 0] == bi), np.flatnonzero(i2[:, 0] == bi)
        m3, m2, q3, q2 = O.modality_split(i3[r3, 1:], i2[r2, 1:], shape, float_keys=True)
        e3.append(m3); e2.append(m2); p3.append(r3[q3]); p2.append(r2[q2])
    assert np.array_equal(got[0], np.concatenate(e3)) and np.array_equal(got[1], np.concatenate(e2))
    assert np.array_equal(got[2], np.concatenate(p3)) and np.array_equal(got[3], np.concatenate(p2))
    exact = run(False)
    false_matches = int((i3[got[2]] != i2[got[3]]).any(1).sum())
    assert false_matches > 0, "the test data must alias"
    assert not np.array_equal(got[0], exact[0])          # the documented deviation B.3
    assert (i3[exact[2]] == i2[exact[3]]).all()


def _oracle_stage(enc, stage, i3, f3, i2, f2, shape, batch, dummy, fps_num, radius, mcs, thresh,
                  float_keys=False):
    """grouped_sparse_conv (:325-430) with numpy + oracle ops.  float_keys: the split with
    the reference's float32 keys (MSMDFusion.py:271-272) instead of exact ones."""
    e3, e2, p3, p2 = [], [], [], []
    for bi in range(batch):
        r3, r2 = np.flatnonzero(i3[:, 0] == bi), np.flatnonzero(i2[:, 0] == bi)
        m3, m2, q3, q2 = O.modality_split(i3[r3, 1:], i2[r2, 1:], shape, float_keys=float_keys)
        e3.append(m3); e2.append(m2); p3.append(r3[q3]); p2.append(r2[q2])
    mix3, mix2 = np.concatenate(e3), np.concatenate(e2)
    s3, s2 = np.concatenate(p3), np.concatenate(p2)
    o2_idx, o2_feat = i2[mix2 == 0], f2[mix2 == 0]
    nn3 = np.full(o2_idx.shape[0], -1, np.int64)
    base = 0
    for bi in range(batch):
        m2, m3 = o2_idx[:, 0] == bi, i3[:, 0] == bi
        r = _oracle_fps_nn(o2_idx[m2], i3[m3], fps_num, radius, mcs, thresh).astype(np.int64)
        nn3[m2] = np.where(r >= 0, r + base, r)
        base += int(m3.sum())
    lin = enc.cross_gate_control[stage][0]
    cg = np.maximum(np.concatenate([f3, dummy]) @ _np(lin.weight).T + _np(lin.bias), 0)
    o2_feat = cg[nn3] * o2_feat            # -1 -> last row, like the reference
    only3 = OracleSparse(f3[mix3 == 0], i3[mix3 == 0], shape, batch)
    lin = enc.gate_control[stage][0]
    m3f, m2f = f3[s3], f2[s2]
    m2f = np.maximum(m3f @ _np(lin.weight).T + _np(lin.bias), 0) * m2f
    mixed_feat, mixed_idx = np.concatenate([m3f, m2f], 1), i2[s2]
    name = f"stage_{stage + 1}"
    only3 = oracle_forward(getattr(enc.grouped_sp_conv_blocks_3D, name), only3)
    c3 = f3.shape[1]
    uf = np.concatenate([np.pad(only3.feat, ((0, 0), (0, 64))), np.pad(o2_feat, ((0, 0), (c3, 0))),
                         mixed_feat]).astype(np.float32)
    ui = np.concatenate([only3.idx, o2_idx, mixed_idx]).astype(np.int32)
    return oracle_forward(getattr(enc.aggregation_blocks, name), OracleSparse(uf, ui, shape, batch))


@pytest.mark.parametrize("stage,n2,fps_num", [(0, 1500, 2048), (1, 4000, 512), (2, 3500, 512),
                                              (3, 1200, 2048)])
def test_gma_conv_stage_matches_oracle(dev, stage, n2, fps_num):
    from msmdfusion_amd import spconv
    from msmdfusion_amd.fusion import voxel_modality_split
    from msmdfusion_amd.multimodal_encoder import SparseMultiModalEncoderPaint
    torch.manual_seed(0)
    enc = SparseMultiModalEncoderPaint(in_channels_2D=(64,) * 4, padding=(1, 1, [0, 1, 1], 0)) \
        .to(dev).train()
    c3 = enc.in_channels_3D[stage]
    shape, batch = [21, 120, 120], 2
    i3, f3, i2, f2 = _make_inputs(dev, shape, 3000, n2, c3, batch, 10 + stage)
    dummy = np.random.RandomState(5).rand(1, c3).astype(np.float32)
    enc.dummy_embedding_fn = lambda c, device: torch.from_numpy(dummy).to(device)
    a = spconv.SparseConvTensor(torch.from_numpy(f3).to(dev), torch.from_numpy(i3).to(dev), shape, batch)
    b = spconv.SparseConvTensor(torch.from_numpy(f2).to(dev), torch.from_numpy(i2).to(dev), shape, batch)
    a, b, s3, s2 = voxel_modality_split(a, b, batch)
    out = enc.grouped_sparse_conv(a, b, s3, s2, stage, fps_num, 6, 50, 13.3)
    exp = _oracle_stage(enc, stage, i3, f3, i2, f2, shape, batch, dummy, fps_num, 6, 50, 13.3)
    assert np.array_equal(_np(out.indices), exp.idx)
    # two SubM convs + BN on top of the gated features: 2e-4 (multi-layer composition)
    np.testing.assert_allclose(_np(out.features), exp.feat, rtol=2e-4, atol=2e-4)


def test_gma_stage0_on_the_real_lc_batch_matches_oracle(dev):
    """GMA-Conv stage 0 at the size bench.py runs it: the LC headline batch (2 x (28.7 k LiDAR
    + 50 k virtual points)), the stage's REAL inputs -- the frozen LiDAR encoder's first scale
    and the virtual-point voxels of the product path -- walked by the oracle (exact-key split,
    FPS / ball query / nearest voxel, gates, three SubM blocks) against grouped_sparse_conv
    with the one-launch assembly.  (Round 2 covered this size by properties only.)"""
    import proc_prefetch_helper as H
    from msmdfusion_amd import spconv
    from msmdfusion_amd.fusion import voxel_modality_split
    model = H.build_model(dev)
    path, B = model.path, 2
    enc, mm = path.pts_middle_encoder, path.multimodal_middle_encoder
    clouds = [torch.from_numpy(S.lidar_sweep(i)).to(dev) for i in range(B)]
    virt = [torch.from_numpy(S.virtual_points(i)).to(dev) for i in range(B)]
    dummy = np.full((1, 16), 0.25, np.float32)           # H.fixed_dummy
    with torch.no_grad():
        feats, coors, v2 = path._voxelize_all(clouds, [virt] * 4, B)
        _, encode_features = enc(feats, coors, B)
        v3 = encode_features[0]
        shape = list(v3.spatial_shape)
        i3, f3 = _np(v3.indices), _np(v3.features)
        i2, f2 = _np(v2[0].indices), _np(v2[0].features)
        assert i3.shape[0] > 30000 and i2.shape[0] > 40000 and f3.shape[1] == 16
        a = spconv.SparseConvTensor(v3.features, v3.indices, shape, B)
        b = spconv.SparseConvTensor(v2[0].features, v2[0].indices, shape, B)
        a, b, s3, s2 = voxel_modality_split(a, b, B)
        out = mm.grouped_sparse_conv(a, b, s3, s2, 0, path.fps_num_list[0], path.radius_list[0],
                                     path.max_cluster_samples_list[0], path.dist_thresh_list[0])
    exp = _oracle_stage(mm, 0, i3, f3, i2, f2, shape, B, dummy, path.fps_num_list[0],
                        path.radius_list[0], path.max_cluster_samples_list[0],
                        path.dist_thresh_list[0])
    assert np.array_equal(_np(out.indices), exp.idx)
    # gates + 16->16 block + two 80->80 blocks with BN on ~80 k rows: 2e-4 (composition of
    # five layers; each conv alone is pinned at 1e-4 in test_gpu_kernels / test_gpu_production)
    np.testing.assert_allclose(_np(out.features), exp.feat, rtol=2e-4, atol=2e-4)


def test_sparse_fusion_path_end_to_end(dev):
    """LC sparse path on small synthetic inputs: shapes, finiteness, autograd,
    determinism with a pinned dummy embedding, checkpoint key names."""
    from msmdfusion_amd.fusion import SparseFusionPath
    from msmdfusion_amd.registry import build_middle_encoder
    from msmdfusion_amd.voxelize import Voxelization
    torch.manual_seed(0)
    vox = Voxelization(S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, 10, (120000, 160000))
    enc = build_middle_encoder(dict(
        type="SparseEncoder", in_channels=5, sparse_shape=[41, 1440, 1440], output_channels=128,
        order=("conv", "norm", "act"),
        encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
        encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, [0, 1, 1]), (0, 0)), block_type="basicblock"))
    mm = build_middle_encoder(dict(          # configs/MSMDFusion_nusc_voxel_LC.py:182-190
        type="SparseMultiModalEncoderPaint", in_channels_3D=(16, 32, 64, 128),
        in_channels_2D=(64, 64, 64, 64), out_channels=(32, 64, 128, 128),
        padding=(1, 1, [0, 1, 1], 0), order=("conv", "norm", "act"),
        norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01)))
    path = SparseFusionPath(vox, enc, mm).to(dev).train()
    keys = set(path.state_dict().keys())
    for k in ["multimodal_middle_encoder.aggregation_blocks.stage_1.conv1.weight",
              "multimodal_middle_encoder.grouped_sp_conv_blocks_3D.stage_2.0.weight",
              "multimodal_middle_encoder.grouped_sp_conv_blocks_2D.stage_1.0.weight",
              "multimodal_middle_encoder.grouped_sp_conv_blocks_mix.stage_4.bn2.weight",
              "multimodal_middle_encoder.gate_control.0.0.weight",
              "multimodal_middle_encoder.cross_gate_control.3.0.bias",
              "multimodal_middle_encoder.downscale_blocks.stage_4.0.weight",
              "pts_middle_encoder.conv_input.0.weight"]:
        assert k in keys, k
    assert path.state_dict()["multimodal_middle_encoder.downscale_blocks.stage_4.0.weight"].shape \
        == (192, 3, 1, 1, 192)
    fixed = {c: torch.rand(1, c) for c in (16, 32, 64, 128)}
    mm.dummy_embedding_fn = lambda c, device: fixed[c].to(device)
    pts = [torch.from_numpy(S.lidar_sweep(i, n_az=300)).to(dev) for i in range(2)]
    virt = [torch.from_numpy(S.virtual_points(i, n=12000)).to(dev) for i in range(2)]
    x, x_mm = path(pts, [virt] * 4)
    assert x.shape == (2, 256, 180, 180) and x_mm.shape == (2, 384, 180, 180)
    assert torch.isfinite(x).all() and torch.isfinite(x_mm).all() and x_mm.abs().sum() > 0
    x2, x_mm2 = path(pts, [virt] * 4)
    assert torch.equal(x_mm, x_mm2)
    # the planned / side-stream schedule computes exactly what the reference's
    # order of calls does (extract_pts_feat, MSMDFusion.py:421-443)
    from msmdfusion_amd.fusion import (virtual_points_to_voxels, voxel_modality_split,
                                       voxelize_batch)
    with torch.no_grad():
        feats, _, coors = voxelize_batch(vox, pts, 1.0, fused_mean=True)
        x_ref, enc_feats = enc(feats, coors, 2)
        v3, v2, s3, s2 = [], [], [], []
        for i in range(4):
            voxel_2D = virtual_points_to_voxels(vox, virt, path.spatial_shapes[i],
                                                path.downscale_factors[i], 2)
            a, b, pa, pb = voxel_modality_split(enc_feats[i].shadow_copy(), voxel_2D, 2)
            v3.append(a); v2.append(b); s3.append(pa); s2.append(pb)
        outs = mm(v3, v2, s3, s2, path.fps_num_list, path.radius_list,
                  path.max_cluster_samples_list, path.dist_thresh_list)
        mm_ref = outs[-1].dense()
        x_p, x_mm_p = path(pts, [virt] * 4)
    assert torch.equal(x_ref, x_p)
    assert torch.equal(mm_ref.view(2, -1, 180, 180), x_mm_p)
    (x.mean() + x_mm.mean()).backward()
    used = [n for n, p in path.named_parameters() if p.grad is not None]
    dead = [n for n, p in path.named_parameters() if p.grad is None]
    assert all(("blocks_2D" in n or "blocks_mix" in n) for n in dead), dead   # built, never called
    assert any("gate_control" in n for n in used)


@pytest.mark.gpu
@pytest.mark.parametrize("threaded", [False, True])
def test_index_prefetch_matches_inline(dev, threaded):
    """bench.py's step pipelining: voxelization + rulebooks of batch i+1 built on
    a side stream while batch i's feature pass runs (msmdfusion_amd/prefetch.py).
    Same BEV map and the same weight gradients, bit for bit, as the inline order,
    over several steps with a different batch each (lifetime of the handed-over
    tables across streams)."""
    import bench
    from msmdfusion_amd import synthetic as S
    from msmdfusion_amd.prefetch import IndexPrefetcher
    torch.manual_seed(0)
    model = bench.Backbone().to(dev).train()
    batches = [[torch.from_numpy(S.lidar_sweep(2 * i + j)).to(dev) for j in range(2)]
               for i in range(4)]

    def run(bev):
        model.zero_grad(set_to_none=True)
        bev.square().mean().backward()
        return bev.detach().clone(), [p.grad.clone() for p in model.parameters()
                                      if p.grad is not None]

    want = [run(model(b)) for b in batches]
    pf = IndexPrefetcher(model.prepare, dev, threaded=threaded)
    pending = [pf.submit(batches[0])]
    for i, b in enumerate(batches):
        if i + 1 < len(batches):
            pending.append(pf.submit(batches[i + 1]))
        ticket = pending.pop(0)
        bev, grads = run(model(b, prepared=pf.take(ticket)))
        pf.retire(ticket)
        assert torch.equal(bev, want[i][0])
        assert len(grads) == len(want[i][1])
        for g, w in zip(grads, want[i][1]):
            assert torch.equal(g, w)
    assert len(pf._retired) <= pf.max_behind + 1


@pytest.mark.gpu
def test_index_prefetch_matches_inline_lc(dev):
    """Same for the LC fusion path: SparseFusionPath.prepare (voxelization of the
    LiDAR and the virtual points, modality split, FPS / ball query / nearest voxel
    of all four stages) a step ahead on the prefetcher's stream == inline."""
    import bench
    from msmdfusion_amd.prefetch import IndexPrefetcher
    torch.manual_seed(0)
    model = bench.FusionBackbone().to(dev).train()
    mm = model.path.multimodal_middle_encoder
    fixed = {c: torch.rand(1, c) for c in mm.in_channels_3D}
    mm.dummy_embedding_fn = lambda c, device: fixed[c].to(device)
    batches = [([torch.from_numpy(S.lidar_sweep(3 * i + j)).to(dev) for j in range(2)],
                [torch.from_numpy(S.virtual_points(3 * i + j)).to(dev) for j in range(2)])
               for i in range(3)]

    def run(bev):
        model.zero_grad(set_to_none=True)
        bev.square().mean().backward()
        return bev.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters()
                                      if p.grad is not None}

    want = [run(model(*b)) for b in batches]
    pf = IndexPrefetcher(model.prepare, dev)
    pending = [pf.submit(*batches[0])]
    for i, b in enumerate(batches):
        if i + 1 < len(batches):
            pending.append(pf.submit(*batches[i + 1]))
        ticket = pending.pop(0)
        bev, grads = run(model(*b, prepared=pf.take(ticket)))
        pf.retire(ticket)
        assert torch.equal(bev, want[i][0])
        assert grads.keys() == want[i][1].keys()
        # forward is bit-exact; the gate tables' index_select backward is torch's
        # index_add_ (fp32 atomics, order not fixed), so gradients agree to rounding
        for n in grads:
            w = want[i][1][n]
            assert (grads[n] - w).abs().max().item() <= 2e-5 * w.abs().max().item() + 1e-12, n


@pytest.mark.gpu
def test_inline_forward_next_to_a_running_prepare(dev):
    """Two prepare() calls may overlap -- a prefetcher of depth 2, or a forward pass run inline
    (evaluation, bench.py's sanity step) while the worker prepares the next batch.  The four
    neighbour-search streams and their scratch are process-wide, so their enqueue is a
    critical section (fusion._NN_STREAMS_LOCK): without it the inline pass now and then read
    the other batch's neighbour indices (round 3: one bench run in ~10 ended with gradients
    missing on the stage 1-3 3-D convs).  Here: the inline result must equal the sequential
    one while a worker thread prepares OTHER batches in a loop."""
    import threading
    import bench
    torch.manual_seed(0)
    model = bench.FusionBackbone().to(dev).train()
    mm = model.path.multimodal_middle_encoder
    fixed = {c: torch.rand(1, c) for c in mm.in_channels_3D}
    mm.dummy_embedding_fn = lambda c, device: fixed[c].to(device)
    batches = [([torch.from_numpy(S.lidar_sweep(2 * i + j)).to(dev) for j in range(2)],
                [torch.from_numpy(S.virtual_points(2 * i + j)).to(dev) for j in range(2)])
               for i in range(3)]
    with torch.no_grad():
        want = model(*batches[0]).clone()
    torch.cuda.synchronize()
    stop = threading.Event()
    errors = []

    def worker():
        try:
            torch.cuda.set_device(dev)
            side = torch.cuda.Stream(device=dev)
            with torch.no_grad(), torch.cuda.stream(side):
                i = 1
                while not stop.is_set():
                    model.prepare(*batches[i])
                    i = 3 - i
            side.synchronize()
        except Exception as e:      # noqa: BLE001
            errors.append(e)
    th = threading.Thread(target=worker)
    th.start()
    try:
        for _ in range(8):
            with torch.no_grad():
                got = model(*batches[0])
            assert torch.equal(got, want)
    finally:
        stop.set()
        th.join()
    assert not errors, errors


@pytest.mark.gpu
def test_prefetched_step_through_ddp(dev):
    """bench.py's N>1 step on one rank: the prepared batch travels through
    DistributedDataParallel's forward as a keyword argument (RCCL backend, world
    size 1 here; the 2-rank gradient averaging is covered on gloo in
    test_dist_cpu.py).  Same BEV map and gradients as the bare module."""
    import socket
    import torch.distributed as dist
    import bench
    from msmdfusion_amd.prefetch import IndexPrefetcher
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0,
                            world_size=1, device_id=dev)
    try:
        torch.manual_seed(0)
        model = bench.Backbone().to(dev).train()
        clouds = [torch.from_numpy(S.lidar_sweep(j)).to(dev) for j in range(2)]
        bev = model(clouds)
        bev.square().mean().backward()
        want = bev.detach().clone(), [p.grad.clone() for p in model.parameters()]
        model.zero_grad(set_to_none=True)
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index],
                                                        gradient_as_bucket_view=True)
        pf = IndexPrefetcher(model.prepare, dev, threaded=False)
        ticket = pf.submit(clouds)
        bev = net(clouds, prepared=pf.take(ticket))
        bev.square().mean().backward()
        pf.retire(ticket)
        torch.cuda.synchronize()
        assert torch.equal(bev, want[0])
        for p, w in zip(model.parameters(), want[1]):
            assert torch.equal(p.grad, w)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_prefetched_lc_step_through_ddp(dev):
    """The LC path the way `bench.py --gpus N` runs a rank: FusionBackbone under
    DistributedDataParallel (RCCL backend, world size 1), the index pass on the threaded
    IndexPrefetcher with its four search streams, distributed.TrainStep (prepared batch as a
    forward keyword, clip, fused AdamW) for several steps.  First-step BEV map equals the bare
    module's; every step's loss and every trained parameter stay finite."""
    import socket
    import torch.distributed as dist
    import proc_prefetch_helper as H
    from msmdfusion_amd import distributed as D
    from msmdfusion_amd.prefetch import IndexPrefetcher
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0,
                            world_size=1, device_id=dev)
    try:
        model = H.build_model(dev)
        clouds, virt = H.make_batch(dev)
        with torch.no_grad():
            want = model(clouds, virt).clone()
        params = [p for p in model.parameters() if p.requires_grad]
        # wrap_data_parallel() hands a lone rank the bare module; the point here is the
        # wrapper's hooks and buckets around the LC step, so build it directly
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index],
                                                        gradient_as_bucket_view=True)
        assert D.rccl_ranks() == 1
        opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True)
        pf = IndexPrefetcher(model.prepare, dev, threaded=True)
        seen = []

        def loss_fn(bev):
            seen.append(bev.detach())
            return bev.float().square().mean()
        step = D.TrainStep(net, params, opt, loss_fn, pf, 10.0)
        step.prime((clouds, virt))
        losses = [step((clouds, virt)) for _ in range(4)]
        torch.cuda.synchronize()
        assert torch.equal(seen[0], want)
        assert all(torch.isfinite(l).item() for l in losses)
        assert float(losses[-1].detach()) != float(losses[0].detach())          # the optimizer moved the weights
        assert all(torch.isfinite(p).all() for p in params)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_process_prefetch_matches_inline_lc(dev):
    """prepare() in a worker PROCESS (flat IPC buffers + pickled skeleton,
    msmdfusion_amd/prefetch_proc.py) feeds the same feature pass as prepare() inline:
    identical BEV map, gradients flow, several batches in a row."""
    import proc_prefetch_helper as H
    from msmdfusion_amd.prefetch_proc import ProcessPrefetcher
    model = H.build_model(dev)
    clouds, virt = H.make_batch(dev)
    with torch.no_grad():
        want = model(clouds, virt, prepared=model.prepare(clouds, virt))
    pf = ProcessPrefetcher(H.init, (), dev)
    try:
        for it in range(3):
            ticket = pf.submit()
            prepared = pf.take(ticket)
            if it < 2:
                with torch.no_grad():
                    got = model(clouds, virt, prepared=prepared)
                assert torch.equal(got, want), it
            else:
                out = model(clouds, virt, prepared=prepared)
                out.mean().backward()
                assert any(p.grad is not None and p.grad.abs().sum() > 0
                           for n, p in model.named_parameters() if "gate_control" in n)
            pf.retire(ticket)
    finally:
        pf.close()


@pytest.mark.gpu
def test_lc_path_bf16_operands(dev, monkeypatch):
    """BASELINE configs[2] names bf16: MSMD_CONV_PLANES=1 runs every split conv of the LC
    path (encoder + fusion stack, forward, dgrad, wgrad) on plain bf16 operands with fp32
    accumulation.  Same voxel sets and row order as the fp32-equivalent default (the index
    pass does not depend on it); features within bf16 rounding of it; finite gradients."""
    import proc_prefetch_helper as H
    model = H.build_model(dev)
    clouds, virt = H.make_batch(dev)
    prepared = model.prepare(clouds, virt)
    with torch.no_grad():
        want = model(clouds, virt, prepared=prepared)
    monkeypatch.setenv("MSMD_CONV_PLANES", "1")
    with torch.no_grad():
        got = model(clouds, virt, prepared=model.prepare(clouds, virt))
    assert got.shape == want.shape
    assert torch.equal(got != 0, want != 0) or \
        float(((got != 0) != (want != 0)).float().mean()) < 1e-3     # same occupied cells
    rel = float((got - want).norm() / want.norm())
    assert 1e-5 < rel < 3e-2, rel          # bf16 operands: ~2^-9 per product, 30 layers deep
    out = model(clouds, virt, prepared=model.prepare(clouds, virt))
    out.mean().backward()
    grads = [p.grad for n, p in model.named_parameters() if p.grad is not None]
    assert grads and all(torch.isfinite(g).all() for g in grads)


@pytest.mark.gpu
def test_gma_assemble_kernel_matches_torch_ops(dev):
    """kernels.gma_assemble (one launch each way) against the index / cat / pad / mul chain
    of the reference's grouped_sparse_conv (:349-421), ragged sizes incl. pad rows and
    rows without a neighbour: forward bit for bit; d conv3 / d gate bit for bit, d cross_gate
    to float-atomic order (the index_add_ it replaces is atomic too)."""
    from msmdfusion_amd import kernels as K
    g = torch.Generator(device=dev).manual_seed(5)
    for (n3, n2, n_o3, n_o2, n_mix, c3, c2, p2, pm) in [(700, 900, 300, 450, 200, 16, 64, 0, 0),
                                                         (50, 40, 17, 23, 9, 32, 64, 2, 1),
                                                         (10, 10, 0, 5, 0, 128, 64, 0, 3),
                                                         (64, 64, 20, 0, 33, 64, 64, 1, 0)]:
        r = lambda *s: torch.randn(*s, device=dev, generator=g)
        ri = lambda hi, n: torch.randint(0, hi, (n,), device=dev, generator=g)
        feat3, feat2 = r(n3, c3), r(n2, c2)
        conv3 = r(n_o3, c3).requires_grad_()
        cross = r(n3 + 1, c2).requires_grad_()
        gate = r(n_mix, c2).requires_grad_()
        nn3 = ri(n3, n_o2)
        nn3[::3] = -1
        rows_o2, rows_m3, rows_m2 = ri(n2, n_o2), ri(n3, n_mix), ri(n2, n_mix)
        out = K.gma_assemble(conv3, cross, gate, feat3, feat2, nn3, rows_o2, rows_m3, rows_m2,
                             p2, pm)
        w = r(*out.shape)
        (out * w).sum().backward()
        got = [t.grad.clone() for t in (conv3, cross, gate)]
        for t in (conv3, cross, gate):
            t.grad = None
        # with the rows sorted by nearest voxel: fixed summation order, run to run identical
        seg = K.gma_nn_segments(nn3, n3)
        det = []
        for _ in range(2):
            out2 = K.gma_assemble(conv3, cross, gate, feat3, feat2, nn3, rows_o2, rows_m3,
                                  rows_m2, p2, pm, segments=seg)
            assert torch.equal(out2, out)
            (out2 * w).sum().backward()
            det.append(cross.grad.clone())
            assert torch.equal(conv3.grad, got[0]) and torch.equal(gate.grad, got[2])
            for t in (conv3, cross, gate):
                t.grad = None
        assert torch.equal(det[0], det[1])
        assert torch.allclose(det[0], got[1], rtol=1e-5, atol=1e-5)
        o2 = cross.index_select(0, torch.where(nn3 >= 0, nn3, torch.full_like(nn3, n3))) * \
            feat2.index_select(0, rows_o2)
        o2 = torch.cat([o2, o2.new_zeros((p2, c2))], 0)
        mixed = torch.cat([feat3.index_select(0, rows_m3), gate * feat2.index_select(0, rows_m2)], -1)
        mixed = torch.cat([mixed, mixed.new_zeros((pm, c3 + c2))], 0)
        ref = torch.cat([F.pad(conv3, (0, c2)), F.pad(o2, (c3, 0)), mixed], 0)
        assert torch.equal(out, ref)
        (ref * w).sum().backward()
        assert torch.equal(got[0], conv3.grad) and torch.equal(got[2], gate.grad)
        assert torch.allclose(got[1], cross.grad, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("c_in,c_out,n", [(16, 64, 5000), (32, 64, 70001), (64, 64, 1000),
                                           (128, 64, 33333), (64, 128, 777), (32, 32, 3)])
def test_rows_linear_matches_torch(dev, c_in, c_out, n):
    """kernels.rows_linear (the gate tables: Linear + ReLU over voxel rows, the dummy row
    appended without a concatenation) == torch's Linear + ReLU forward, and its weight / bias
    / input gradients (fp32 sums in another order: 1e-5 of the largest entry)."""
    from msmdfusion_amd import kernels as K
    assert K.rows_linear_supported(c_in, c_out) and not K.rows_linear_supported(10, 64)
    g = torch.Generator(device=dev).manual_seed(c_in + c_out + n)
    x = torch.randn(n, c_in, device=dev, generator=g)
    tail = torch.randn(1, c_in, device=dev, generator=g)
    lin = torch.nn.Linear(c_in, c_out).to(dev)
    dy = torch.randn(n + 1, c_out, device=dev, generator=g)
    for use_tail, req_x in ((True, False), (False, True)):
        xr = x.clone().requires_grad_(req_x)
        xin = torch.cat([xr, tail], 0) if use_tail else xr
        want = torch.relu(lin(xin))
        lin.zero_grad(set_to_none=True)
        want.backward(dy[:want.shape[0]])
        gw, gb = lin.weight.grad.clone(), lin.bias.grad.clone()
        gx = xr.grad.clone() if req_x else None
        xr2 = x.clone().requires_grad_(req_x)
        lin.zero_grad(set_to_none=True)
        got = K.rows_linear(xr2, lin.weight, lin.bias, relu=True, x_tail=tail if use_tail else None)
        assert got.shape == want.shape
        assert (got - want).abs().max().item() <= 1e-5 * max(want.abs().max().item(), 1.0)
        got.backward(dy[:got.shape[0]])
        for a, b in ((lin.weight.grad, gw), (lin.bias.grad, gb)):
            assert (a - b).abs().max().item() <= 1e-5 * max(b.abs().max().item(), 1.0)
        if req_x:
            assert (xr2.grad - gx).abs().max().item() <= 1e-5 * max(gx.abs().max().item(), 1.0)
    # no ReLU, no bias; run to run identical (fixed summation order)
    y0 = K.rows_linear(x, lin.weight, None, relu=False)
    assert (y0 - x @ lin.weight.t()).abs().max().item() <= 1e-5 * y0.abs().max().item()
    w2 = lin.weight.detach().clone().requires_grad_(True)
    grads = []
    for _ in range(2):
        w2.grad = None
        K.rows_linear(x, w2, lin.bias, relu=True).backward(dy[:n])
        grads.append(w2.grad.clone())
    assert torch.equal(grads[0], grads[1])


@pytest.mark.gpu
def test_fused_stage_assembly_matches_the_op_chain(dev):
    """The LC path with the one-launch stage assembly == with the reference's op-by-op
    chain: identical BEV map; gate / conv gradients equal up to the order of float atomics."""
    import proc_prefetch_helper as H
    model = H.build_model(dev)
    clouds, virt = H.make_batch(dev)
    mm = model.path.multimodal_middle_encoder
    res = {}
    for fused in (True, False):
        mm.fused_assembly = fused
        model.zero_grad(set_to_none=True)
        out = model(clouds, virt, prepared=model.prepare(clouds, virt))
        out.square().mean().backward()
        res[fused] = (out.detach().clone(),
                      {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    mm.fused_assembly = True
    assert torch.equal(res[True][0], res[False][0])
    assert res[True][1].keys() == res[False][1].keys() and \
        any("cross_gate_control" in n for n in res[True][1])
    for n, g in res[True][1].items():
        w = res[False][1][n]
        assert torch.allclose(g, w, rtol=2e-4, atol=1e-6 + 2e-5 * float(w.abs().max())), n
