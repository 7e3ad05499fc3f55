"""GPU parity tests: every C-ABI entry point (through msmdfusion_amd.kernels)
against the CPU oracle on the same seeded inputs.  Integer results (voxel
indices, rulebooks, set operations, FPS / ball query) must match bit for bit;
fp32 features within 1e-4 (BASELINE.json north_star)."""
import os

import numpy as np
import pytest
import torch

from msmdfusion_amd import synthetic as S
from oracle import oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4


def t(a, dev, dtype=None):
    x = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return x if dtype is None else x.to(dtype)


# ------------------------------------------------------------------ voxelization
@pytest.mark.parametrize("max_points,max_voxels,scale", [(10, 120000, 1), (10, 5000, 1),
                                                         (3, 120000, 2), (1, 700, 8)])
def test_hard_voxelize_lidar(dev, max_points, max_voxels, scale):
    from msmdfusion_amd import kernels as K
    pts = S.lidar_sweep(1)
    vs = [v * scale for v in S.VOXEL_SIZE]
    ev, ec, en = O.hard_voxelize(pts, vs, S.POINT_CLOUD_RANGE, max_points, max_voxels)
    v, c, n, mean = K.hard_voxelize(t(pts, dev), vs, S.POINT_CLOUD_RANGE, max_points, max_voxels,
                                    want_voxels=True, want_mean=True)
    assert c.shape[0] == ec.shape[0]
    assert np.array_equal(c.cpu().numpy(), ec)
    assert np.array_equal(n.cpu().numpy(), en)
    assert np.array_equal(v.cpu().numpy(), ev)          # features are copies: bit exact
    em = O.voxel_mean(ev, en)
    np.testing.assert_allclose(mean.cpu().numpy(), em, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(K.voxel_mean(v, n).cpu().numpy(), em, rtol=1e-6, atol=1e-6)


def test_hard_voxelize_virtual_64ch(dev):
    from msmdfusion_amd import kernels as K
    pts = S.virtual_points(0, n=20000)
    for f in (1, 4):
        vs = [v * f for v in S.VOXEL_SIZE]
        ev, ec, en = O.hard_voxelize(pts, vs, S.POINT_CLOUD_RANGE, 10, 160000)
        v, c, n, _ = K.hard_voxelize(t(pts, dev), vs, S.POINT_CLOUD_RANGE, 10, 160000)
        assert np.array_equal(c.cpu().numpy(), ec)
        assert np.array_equal(n.cpu().numpy(), en)
        assert np.array_equal(v.cpu().numpy(), ev)


def test_hard_voxelize_edge_cases(dev):
    from msmdfusion_amd import kernels as K
    rng = np.random.RandomState(0)
    # all points out of range -> zero voxels
    far = (rng.rand(100, 5) + 100).astype(np.float32)
    v, c, n, _ = K.hard_voxelize(t(far, dev), S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, 10, 1000)
    assert c.shape[0] == 0 and v.shape[0] == 0
    # every point in one voxel, more points than slots
    one = np.tile(np.array([[0.01, 0.01, 0.01, 0.5, 0]], np.float32), (50, 1))
    one[:, 3] = np.arange(50)
    ev, ec, en = O.hard_voxelize(one, S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, 10, 1000)
    v, c, n, _ = K.hard_voxelize(t(one, dev), S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, 10, 1000)
    assert np.array_equal(v.cpu().numpy(), ev) and np.array_equal(n.cpu().numpy(), en)
    # points exactly on cell / range boundaries (float32 floor semantics)
    edge = np.array([[-54.0, -54.0, -5.0, 0, 0], [54.0, 0, 0, 0, 0], [53.999996, 0, 0, 0, 0],
                     [0.075, 0.15, 0.2, 0, 0], [0.0749999, 0.1499999, 0.1999999, 0, 0],
                     [-0.0, -0.0, -0.0, 0, 0]], np.float32)
    ev, ec, en = O.hard_voxelize(edge, S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, 10, 1000)
    v, c, n, _ = K.hard_voxelize(t(edge, dev), S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, 10, 1000)
    assert np.array_equal(c.cpu().numpy(), ec) and np.array_equal(n.cpu().numpy(), en)


# ------------------------------------------------------------------ rulebooks
GEOMS = [  # (subm, ksize, stride, padding) : the geometries of SURVEY Appendix A
    (True, 3, 1, 1),
    (False, 3, 2, 1),
    (False, 3, 2, [0, 1, 1]),
    (False, [3, 1, 1], [2, 1, 1], 0),
]


def _check_rulebook(dev, idx, batch, shape, subm, ks, st, pd):
    from msmdfusion_amd import kernels as K
    oi, pr, nm, osz = O.get_indice_pairs(idx, batch, shape, ks, st, pd, 1, subm)
    d_idx = t(idx, dev)
    if subm:
        nbr = K.rulebook_subm(d_idx, batch, shape, ks, method="hash")
        # the occupancy-bitmap index builds the identical table
        assert torch.equal(K.rulebook_subm(d_idx, batch, shape, ks, method="bitmap"), nbr)
        # SubM keeps input order: compare raw rows
        exp = np.full((nm.shape[0], idx.shape[0]), -1, np.int32)
        for k in range(nm.shape[0]):
            p = int(nm[k])
            exp[k, pr[k, 1, :p]] = pr[k, 0, :p]
        assert np.array_equal(nbr.cpu().numpy(), exp)
        pairs, num = K.rulebook_pairs(nbr)
        assert np.array_equal(num.cpu().numpy(), nm)
        _, can, _ = O.canonical_rulebook(oi, pr, nm, osz, keep_rows=True)
        got = pairs.cpu().numpy()
        for k in range(nm.shape[0]):
            p = int(nm[k])
            assert np.array_equal(got[k, :, :p].T, can[k])
            assert (got[k, :, p:] == -1).all()
        return
    out_idx, nbr_fwd, nbr_bwd, out_shape = K.rulebook_conv(d_idx, batch, shape, ks, st, pd)
    assert list(out_shape) == list(osz)
    coi, can, perm = O.canonical_rulebook(oi, pr, nm, osz)
    assert np.array_equal(out_idx.cpu().numpy(), coi)           # ascending linear id
    exp_fwd = O.nbr_table_from_pairs(can, coi.shape[0])
    assert np.array_equal(nbr_fwd.cpu().numpy(), exp_fwd)
    exp_bwd = np.full((nm.shape[0], idx.shape[0]), -1, np.int32)
    for k, po in enumerate(can):
        exp_bwd[k, po[:, 0]] = po[:, 1]
    assert np.array_equal(nbr_bwd.cpu().numpy(), exp_bwd)
    pairs, num = K.rulebook_pairs(nbr_fwd, ld=max(idx.shape[0], coi.shape[0]))
    assert np.array_equal(num.cpu().numpy(), nm)
    got = pairs.cpu().numpy()
    for k in range(nm.shape[0]):
        assert np.array_equal(got[k, :, :int(nm[k])].T, can[k])


@pytest.mark.parametrize("subm,ks,st,pd", GEOMS)
def test_rulebook_random(dev, subm, ks, st, pd):
    idx = S.random_voxel_indices(6000, 2, [21, 200, 200], seed=3)
    _check_rulebook(dev, idx, 2, [21, 200, 200], subm, ks, st, pd)


@pytest.mark.parametrize("subm,ks,st,pd", GEOMS)
def test_rulebook_lidar_full_grid(dev, subm, ks, st, pd):
    _, c, _ = O.hard_voxelize(S.lidar_sweep(0), S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, 10, 120000)
    idx = np.concatenate([np.zeros((c.shape[0], 1), np.int32), c], 1)
    _check_rulebook(dev, idx, 1, S.SPARSE_SHAPE, subm, ks, st, pd)


def test_rulebook_edges(dev):
    from msmdfusion_amd import kernels as K
    shape = [5, 8, 8]
    # corners and a full dense block: boundary handling on every side
    dense_blk = np.array([[0, z, y, x] for z in range(5) for y in range(8) for x in range(8)],
                         np.int32)
    for subm, ks, st, pd in GEOMS:
        _check_rulebook(dev, dense_blk, 1, shape, subm, ks, st, pd)
    single = np.array([[1, 4, 7, 7]], np.int32)
    for subm, ks, st, pd in GEOMS:
        _check_rulebook(dev, single, 2, shape, subm, ks, st, pd)
    empty = torch.zeros((0, 4), dtype=torch.int32, device=dev)
    assert K.rulebook_subm(empty, 1, shape, 3).shape == (27, 0)
    oi, f, b, _ = K.rulebook_conv(empty, 1, shape, 3, 2, 1)
    assert oi.shape[0] == 0 and f.shape == (27, 0)


def test_rulebook_conv_chain_equals_level_by_level(dev):
    """msmd_rulebook_conv3d_count_chain: SparseEncoder's four strided convs (k3 s2 p1 twice,
    k3 s2 p(0,1,1), k(3,1,1) s(2,1,1) p0) counted back to back from each other's bitmaps, one
    host read -- tensor for tensor the level-by-level rulebooks."""
    from msmdfusion_amd import kernels as K
    geoms = [(3, 2, 1), (3, 2, 1), (3, 2, [0, 1, 1]), ([3, 1, 1], [2, 1, 1], 0)]
    for seed, batch, shape, n in ((1, 2, [41, 200, 176], 20000), (2, 3, [25, 64, 64], 3000),
                                  (3, 1, [41, 16, 16], 1)):
        idx = t(S.random_voxel_indices(n, batch, shape, seed=seed), dev)
        chain = K.rulebook_conv_chain(idx, batch, shape, geoms)
        cur, cur_shape = idx, shape
        for (ks, st, pd), got in zip(geoms, chain):
            want = K.rulebook_conv(cur, batch, cur_shape, ks, st, pd)
            assert list(got[3]) == list(want[3])
            for a, b in zip(got[:3], want[:3]):
                assert torch.equal(a, b)
            cur, cur_shape = want[0], want[3]
        assert chain[-1][0].shape[0] > 0
    empty = torch.zeros((0, 4), dtype=torch.int32, device=dev)
    for out_idx, fwd, bwd, _ in K.rulebook_conv_chain(empty, 1, [41, 16, 16], geoms):
        assert out_idx.shape[0] == 0 and fwd.shape[1] == 0


def test_rulebook_subm_index_methods(dev):
    """hash and bitmap SubM indices: duplicate coordinates (both keep the LAST row, as the
    CPU reference's grid does, geometry.h:277-282), cells at the word boundaries of the
    bitmap, a line kernel, and the automatic choice by size."""
    from msmdfusion_amd import kernels as K
    shape = [3, 5, 70]
    rows = [[0, 1, 2, x] for x in (0, 30, 31, 32, 33, 63, 64, 65, 69)]
    rows += [[1, 1, 2, 31], [1, 1, 2, 32], [0, 1, 2, 31], [1, 2, 4, 69], [1, 2, 4, 0], [0, 1, 2, 31]]
    idx = np.array(rows, np.int32)
    d = t(idx, dev)
    for ks in (3, [1, 1, 3], [3, 1, 1], 5):
        a = K.rulebook_subm(d, 2, shape, ks, method="hash")
        b = K.rulebook_subm(d, 2, shape, ks, method="bitmap")
        assert torch.equal(a, b), ks
    centre = K.rulebook_subm(d, 2, shape, 3, method="bitmap")[13].cpu().numpy()
    assert centre[2] == 14 and centre[11] == 14 and centre[14] == 14    # last duplicate wins
    assert K.subm_index_method(720000, 1, [41, 1440, 1440]) == "bitmap"
    assert K.subm_index_method(38000, 2, [41, 1440, 1440]) == "hash"
    big = S.random_voxel_indices(90000, 2, [21, 300, 300], seed=5)
    assert K.subm_index_method(big.shape[0], 2, [21, 300, 300]) == "bitmap"
    assert torch.equal(K.rulebook_subm(t(big, dev), 2, [21, 300, 300], 3),
                       K.rulebook_subm(t(big, dev), 2, [21, 300, 300], 3, method="hash"))


# ------------------------------------------------------------------ convolution
CHANNELS = [(5, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128),
            (80, 80), (80, 96), (96, 128), (128, 192), (192, 192), (7, 20)]


@pytest.mark.parametrize("cin,cout", CHANNELS)
def test_subm_conv_fwd_bwd(dev, cin, cout):
    from msmdfusion_amd import kernels as K
    shape = [11, 64, 64]
    idx = S.random_voxel_indices(1500, 2, shape, seed=cin + cout)
    n = idx.shape[0]
    rng = np.random.RandomState(cin * 1000 + cout)
    f = rng.randn(n, cin).astype(np.float32)
    w = (rng.randn(27, cin, cout) / np.sqrt(27 * cin)).astype(np.float32)
    g = rng.randn(n, cout).astype(np.float32)
    oi, pr, nm, _ = O.get_indice_pairs(idx, 2, shape, 3, 1, 1, 1, True)
    exp = O.indice_conv_fwd(f, w, pr, nm, n, subm=True)
    edin, edw = O.indice_conv_bwd(f, w, g, pr, nm, subm=True)

    nbr = K.rulebook_subm(t(idx, dev), 2, shape, 3)
    wd = t(w, dev)
    out = K.conv_forward(t(f, dev), K.pack_weight(wd), nbr, n, cout)
    np.testing.assert_allclose(out.cpu().numpy(), exp, rtol=TOL, atol=TOL)
    # any tiling order gives bit-identical results (mask-sorted and random)
    order = K.row_mask_order(nbr)
    assert sorted(order.cpu().tolist()) == list(range(n))
    out_o = K.conv_forward(t(f, dev), K.pack_weight(wd), nbr, n, cout, row_order=order)
    assert torch.equal(out, out_o)
    perm = torch.randperm(n, device=dev).int()
    assert torch.equal(out, K.conv_forward(t(f, dev), K.pack_weight(wd), nbr, n, cout,
                                           row_order=perm))
    # dgrad: forward table read with flipped weights and W^T
    din = K.conv_forward(t(g, dev), K.pack_weight(wd, transpose=True), nbr, n, cin,
                         weight_flip=True)
    np.testing.assert_allclose(din.cpu().numpy(), edin, rtol=TOL, atol=TOL)
    pairs, num = K.rulebook_pairs(nbr)
    dw = K.conv_wgrad(t(f, dev), t(g, dev), pairs, num)
    np.testing.assert_allclose(dw.cpu().numpy(), edw, rtol=TOL, atol=TOL * 5)


@pytest.mark.parametrize("ks,st,pd", [(3, 2, 1), (3, 2, [0, 1, 1]), ([3, 1, 1], [2, 1, 1], 0)])
@pytest.mark.parametrize("cin,cout", [(16, 32), (64, 128), (128, 128), (80, 96)])
def test_strided_conv_fwd_bwd(dev, ks, st, pd, cin, cout):
    from msmdfusion_amd import kernels as K
    shape = [11, 64, 64]
    idx = S.random_voxel_indices(1500, 2, shape, seed=11)
    n = idx.shape[0]
    rng = np.random.RandomState(7)
    f = rng.randn(n, cin).astype(np.float32)
    kvol = int(np.prod(O.expand3(ks)))
    w = (rng.randn(kvol, cin, cout) / np.sqrt(kvol * cin)).astype(np.float32)
    oi, pr, nm, osz = O.get_indice_pairs(idx, 2, shape, ks, st, pd, 1, False)
    m = oi.shape[0]
    g = rng.randn(m, cout).astype(np.float32)
    exp = O.indice_conv_fwd(f, w, pr, nm, m)
    edin, edw = O.indice_conv_bwd(f, w, g, pr, nm)
    _, _, perm = O.canonical_rulebook(oi, pr, nm, osz)   # oracle rows -> sorted rows

    out_idx, nbr_fwd, nbr_bwd, _ = K.rulebook_conv(t(idx, dev), 2, shape, ks, st, pd)
    wd = t(w, dev)
    out = K.conv_forward(t(f, dev), K.pack_weight(wd), nbr_fwd, m, cout)
    np.testing.assert_allclose(out.cpu().numpy(), exp[perm], rtol=TOL, atol=TOL)
    gs = t(g[perm], dev)
    din = K.conv_forward(gs, K.pack_weight(wd, transpose=True), nbr_bwd, n, cin)
    np.testing.assert_allclose(din.cpu().numpy(), edin, rtol=TOL, atol=TOL)
    pairs, num = K.rulebook_pairs(nbr_fwd, ld=max(n, m))
    dw = K.conv_wgrad(t(f, dev), gs, pairs, num)
    np.testing.assert_allclose(dw.cpu().numpy(), edw, rtol=TOL, atol=TOL * 5)


@pytest.mark.parametrize("n_vox", [1500, 4097, 130])
def test_rulebook_plan_equals_its_parts(dev, n_vox):
    """msmd_rulebook_plan (order + tiled table + 128-row stream-K prefix out of the tiling's
    last kernel + pair lists, one call) == rulebook_tiling, permute_cols, tile_prefix and
    rulebook_pairs called one by one (odd tile counts, a partial last tile, fewer rows than
    one block)."""
    from msmdfusion_amd import kernels as K
    shape = [11, 64, 64]
    idx = S.random_voxel_indices(n_vox, 2, shape, seed=n_vox)
    nbr = K.rulebook_subm(t(idx, dev), 2, shape, 3)
    plan = K.rulebook_plan(nbr, tile_rows=(128,), want_pairs=True)
    order, tiled = K.rulebook_tiling(nbr)
    assert torch.equal(plan["order"], order) and torch.equal(plan["tiled"], tiled)
    assert torch.equal(tiled, K.permute_cols(nbr, order))
    assert torch.equal(plan["prefix"][128], K.tile_prefix(tiled, 128))
    pairs, num = K.rulebook_pairs(nbr)
    assert torch.equal(plan["pairs"][0], pairs) and torch.equal(plan["pairs"][1], num)
    again = K.rulebook_plan(nbr, tile_rows=(128,))          # (the block counter re-arms itself)
    assert torch.equal(again["prefix"][128], plan["prefix"][128])


def test_rulebook_subm_many_equals_the_single_calls(dev):
    """msmd_rulebook_subm3d_many -- the SubM tables of several voxel sets from one launch set,
    hash-indexed and bitmap-indexed sets mixed, different grids, kernel sizes and batch
    sizes -- equals msmd_rulebook_subm3d / _bitmap set by set (duplicate coordinates keep the
    last row in both); 19 sets = two launch sets."""
    from msmdfusion_amd import kernels as K
    cases = [(5000, 2, [11, 64, 64], 3, "hash"), (5000, 2, [11, 64, 64], 3, "bitmap"),
             (1, 1, [5, 8, 8], 3, "bitmap"), (130, 3, [7, 30, 30], [3, 3, 1], "hash"),
             (2600, 2, [11, 64, 64], [1, 3, 3], "bitmap"), (9000, 4, [21, 90, 90], 3, None),
             (700, 1, [41, 200, 200], 3, None), (3000, 2, [3, 100, 100], [3, 5, 3], "bitmap")]
    jobs = []
    for seed, (n_vox, batch, shape, ks, method) in enumerate(cases):
        idx = S.random_voxel_indices(n_vox, batch, shape, seed=seed)
        if seed == 0:                       # duplicate coordinates: the last row wins
            idx = np.concatenate([idx, idx[:37]])
        ti = t(idx, dev)
        jobs.append(dict(indices=ti, batch_size=batch, spatial_shape=shape, ksize=ks,
                         method=method, nbr=K.subm_table(ti, ks)))
    for round_ in range(2):                 # (workspace reuse)
        for j in jobs:
            j["nbr"].fill_(-7)
        K.rulebook_subm_many(jobs)
        for j in jobs:
            ref = K.rulebook_subm(j["indices"], j["batch_size"], j["spatial_shape"], j["ksize"],
                                  method=j["method"])
            assert torch.equal(j["nbr"], ref), (j["indices"].shape, j["ksize"], j["method"])
    many = [dict(jobs[i % 5], nbr=K.subm_table(jobs[i % 5]["indices"], jobs[i % 5]["ksize"]))
            for i in range(19)]
    K.rulebook_subm_many(many)
    for i, j in enumerate(many):
        assert torch.equal(j["nbr"], jobs[i % 5]["nbr"])
    K.rulebook_subm_many([])
    empty = t(np.zeros((0, 4), np.int32), dev)
    K.rulebook_subm_many([dict(indices=empty, batch_size=1, spatial_shape=[4, 4, 4], ksize=3,
                               nbr=K.subm_table(empty, 3))])


def test_add_conv_chain_equals_stage_by_stage(dev):
    """msmd_rulebook_add_conv_count_chain + the level-by-level fills (kernels.add_conv_chain:
    one host read for the fusion stack's whole stage chain) == sparse_add_index +
    rulebook_conv called stage by stage: union indices, both row maps, output indices and
    both neighbour tables of every level.  Mixed strides / kernels, an empty extra set at a
    middle level, voxels shared between a stage's own set and the previous output."""
    from msmdfusion_amd import kernels as K
    batch, shape = 2, [17, 96, 96]
    geoms = [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)),
             ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 1, 1), (2, 1, 1), (0, 0, 0))]
    shapes, sh = [], list(shape)
    for ks, st, pd in geoms:
        shapes.append(list(sh))
        sh = K.conv_output_size(sh, list(ks), list(st), list(pd))
    sizes = [6000, 2500, 0, 300]
    extras = [t(S.random_voxel_indices(n, batch, shapes[l], seed=40 + l), dev) if n else
              torch.zeros((0, 4), dtype=torch.int32, device=dev) for l, n in enumerate(sizes)]
    for need_bwd in (True, False):
        got = K.add_conv_chain(extras, batch, shape, geoms, need_bwd=need_bwd)
        prev = None
        for l, (ks, st, pd) in enumerate(geoms):
            if l == 0:
                total = extras[0]
                assert got[0]["map_a"] is None and got[0]["total_indices"] is extras[0]
            else:
                total, ma, mb = K.sparse_add_index(extras[l], prev, batch, shapes[l])
                assert torch.equal(got[l]["total_indices"], total), l
                assert torch.equal(got[l]["map_a"], ma) and torch.equal(got[l]["map_b"], mb), l
            oi, nf, nb, osz = K.rulebook_conv(total, batch, shapes[l], list(ks), list(st), list(pd),
                                              need_bwd=need_bwd)
            assert got[l]["out_shape"] == list(osz) and got[l]["in_shape"] == shapes[l]
            assert torch.equal(got[l]["out_indices"], oi), l
            assert torch.equal(got[l]["nbr_fwd"], nf), l
            if need_bwd:
                assert torch.equal(got[l]["nbr_bwd"], nb), l
            else:
                assert got[l]["nbr_bwd"] is None
            prev = oi
        assert prev.shape[0] > 0


def test_rows_where_eq(dev):
    """msmd_rows_where_eq == (flags == value).nonzero(): contiguous and strided (a column of an
    index tensor) flags, several scan tiles, no hit / all hits, a capacity below the count."""
    from msmdfusion_amd import kernels as K
    rng = np.random.RandomState(5)
    for n in (1, 77, 2048, 2049, 50001):
        idx = t(rng.randint(0, 3, size=(n, 5)).astype(np.int32), dev)
        for flags in (idx[:, 1], idx[:, 1].contiguous()):
            for value in (0, 1, 7):
                want = (flags == value).nonzero().flatten()
                got = K.rows_where_eq(flags, value, want.shape[0])
                assert got.dtype == torch.long and torch.equal(got, want), (n, value)
        ones = torch.ones((n,), dtype=torch.int32, device=dev)
        assert torch.equal(K.rows_where_eq(ones, 1, n), torch.arange(n, device=dev))
        if n > 10:      # (a smaller capacity: the first rows, nothing written past them)
            assert torch.equal(K.rows_where_eq(ones, 1, 10), torch.arange(10, device=dev))
        # a host-side count LARGER than the real one (stale statistics): the tail is -1, the
        # padding torch.nonzero_static uses -- never whatever the allocation held
        half = (torch.arange(n, device=dev) % 2).int()
        real = int((half == 1).sum())
        got = K.rows_where_eq(half, 1, real + 5)
        assert torch.equal(got[:real], (half == 1).nonzero().flatten())
        assert bool((got[real:] == -1).all())


def test_rows_where_eq_many(dev):
    """msmd_rows_where_eq_many == msmd_rows_where_eq list by list: contiguous and strided
    flags, lists below / across scan tiles, an empty flag vector, a count of zero (skipped),
    a capacity above the real count (-1 tail), 20 lists = two launch sets."""
    from msmdfusion_amd import kernels as K
    rng = np.random.RandomState(6)
    jobs, want = [], []
    for n in (1, 77, 2048, 2049, 50001, 0, 4100, 300, 9000, 12):
        idx = t(rng.randint(0, 3, size=(max(n, 1), 5)).astype(np.int32), dev)[:n]
        for flags in (idx[:, 1], idx[:, 1].contiguous()):
            value = int(rng.randint(0, 3))
            real = int((flags == value).sum()) if n else 0
            extra = 3 if n == 300 else 0
            jobs.append((flags, value, real + extra))
            w = (flags == value).nonzero().flatten()
            want.append(torch.cat([w, torch.full((extra,), -1, dtype=torch.long, device=dev)]))
    got = K.rows_where_eq_many(jobs)
    assert len(got) == len(jobs) == 20
    for g, w, (f, v, c) in zip(got, want, jobs):
        assert g.dtype == torch.long and g.shape[0] == c and torch.equal(g, w)
        if c:
            assert torch.equal(g, K.rows_where_eq(f, v, c))
    assert K.rows_where_eq_many([]) == []


def _plan_tables(dev):
    """Tables of an index pass in miniature: SubM 3x3x3 of three sizes (one below a block, one
    a single row), both sides of a stride-2 conv (ld > rows: the output side is the shorter
    one), a (3,1,1) conv (K = 3), a 2x2x2 conv (K = 8), a 3x3x2 SubM (K = 18: its key does
    not fit under a table id -> the single-call path inside plan_many)."""
    from msmdfusion_amd import kernels as K
    shape = [11, 64, 64]
    tabs = []
    for n_vox, seed in ((5000, 1), (130, 2), (1, 3), (2600, 4)):
        idx = t(S.random_voxel_indices(n_vox, 2, shape, seed=seed), dev)
        tabs.append(dict(nbr=K.rulebook_subm(idx, 2, shape, 3), ld=idx.shape[0]))
    idx = t(S.random_voxel_indices(6000, 2, shape, seed=5), dev)
    out, fwd, bwd, _ = K.rulebook_conv(idx, 2, shape, 3, 2, 1)
    ld = max(idx.shape[0], out.shape[0])
    tabs += [dict(nbr=fwd, ld=ld), dict(nbr=bwd, ld=ld)]
    out, fwd, bwd, _ = K.rulebook_conv(idx, 2, shape, [3, 1, 1], [2, 1, 1], 0)
    tabs.append(dict(nbr=fwd, ld=max(idx.shape[0], out.shape[0])))
    out, fwd, bwd, _ = K.rulebook_conv(idx, 2, shape, 2, 2, 0)
    tabs.append(dict(nbr=fwd, ld=max(idx.shape[0], out.shape[0])))
    tabs.append(dict(nbr=K.rulebook_subm(idx, 2, shape, [3, 3, 2]), ld=idx.shape[0]))
    return tabs


def test_rulebook_plan_many_equals_the_single_plans(dev):
    """msmd_rulebook_plan_many -- every table of an index pass planned by one launch set (one
    radix sort with the table id above the mask key) -- gives each table exactly what
    msmd_rulebook_plan / msmd_rulebook_pair_segments give it alone: order, tile-ordered table,
    both prefixes, pair lists with their -1 tails (ld past the padded table), counts, the
    one-chunk segment table.  Mixed requests: order only, table without pairs, everything."""
    from msmdfusion_amd import kernels as K
    tabs = _plan_tables(dev)
    wants = [dict(tile_rows={128, 256}, want_pairs=True, want_segments=True),
             dict(tile_rows={128}, want_pairs=True),
             dict(tile_rows={256}, want_segments=True),
             dict(),                                         # order only
             dict(tile_rows={128}, want_segments=True),
             dict(tile_rows={256}),
             dict(tile_rows={128}, want_pairs=True),
             dict(want_table=True, want_pairs=True, want_segments=True),
             dict(tile_rows={128, 256}, want_pairs=True, want_segments=True)]
    assert len(wants) == len(tabs)
    jobs = [dict(tab, **w) for tab, w in zip(tabs, wants)]
    for round_ in range(2):                                  # (workspace reuse)
        res = K.rulebook_plan_many(jobs)
        assert len(res) == len(jobs)
        for j, r in zip(jobs, res):
            nbr = j["nbr"]
            order, tiled = K.rulebook_tiling(nbr)
            assert torch.equal(r["order"], order), nbr.shape
            rows = set(j.get("tile_rows") or ())
            if rows or j.get("want_table"):
                assert torch.equal(r["tiled"], tiled)
            else:
                assert r["tiled"] is None
            assert set(r["prefix"]) == rows
            for h in rows:
                assert torch.equal(r["prefix"][h], K.tile_prefix(tiled, h)), (nbr.shape, h)
            if j.get("want_pairs") or j.get("want_segments"):
                pairs, num = K.rulebook_pairs(nbr, ld=j["ld"])
                assert torch.equal(r["pairs"][1], num)
                assert torch.equal(r["pairs"][0], pairs), nbr.shape
            else:
                assert r["pairs"] is None
            if j.get("want_segments"):
                table, n_chunks = K.pair_segments(*K.rulebook_pairs(nbr, ld=j["ld"]), chunk_rows=0)
                assert n_chunks == 1 and r["segments"][1] == 1
                assert torch.equal(r["segments"][0], table)
            else:
                assert r["segments"] is None
    assert K.rulebook_plan_many([]) == []
    # more tables than one launch set holds (32): split inside, same results
    many = [dict(tabs[i % 4], tile_rows={128}, want_pairs=True) for i in range(37)]
    res = K.rulebook_plan_many(many)
    for j, r in zip(many, res):
        one = K.rulebook_plan(j["nbr"], tile_rows=(128,), want_pairs=True, ld=j["ld"])
        assert torch.equal(r["order"], one["order"]) and torch.equal(r["tiled"], one["tiled"])
        assert torch.equal(r["prefix"][128], one["prefix"][128])
        assert torch.equal(r["pairs"][0], one["pairs"][0])
        assert torch.equal(r["pairs"][1], one["pairs"][1])


@pytest.mark.parametrize("cin,cout", [(64, 128), (32, 64), (80, 80), (128, 192), (40, 72),
                                      (96, 176), (128, 160)])
def test_conv_epilogue_leaves_the_batchnorm_partials(dev, cin, cout):
    """msmd_spconv_fwd_split_stats: per row tile (128 rows, 256 in the ping-pong form from 161
    output channels up) the column sums and sums of squares of the rows the conv wrote (every
    instantiation width: 2 / 4 / 6 / 8 / 12 column tiles, a partial last
    channel tile, a partial last row tile; stream-K pieces summed by the owner first), and
    msmd_bn_act_fwd_from_partials_f32 == the BatchNorm with its own statistics pass."""
    from msmdfusion_amd import kernels as K
    shape = [11, 64, 64]
    idx = S.random_voxel_indices(2100, 2, shape, seed=cin + cout)
    n = idx.shape[0]
    nbr = K.rulebook_subm(t(idx, dev), 2, shape, 3)
    tr = K.split_tile_rows(cout)
    plan = K.rulebook_plan(nbr, tile_rows=(tr,))
    g = torch.Generator(device=dev).manual_seed(cin * 7 + cout)
    f = torch.randn(n, cin, device=dev, generator=g)
    w = torch.randn(27, cin, cout, device=dev, generator=g) / (27 * cin) ** 0.5
    ws = K.pack_weight_split(w, 3)
    for pre in (plan["prefix"][tr], None):      # stream-K and whole tiles
        out, part = K.conv_forward_split(f, ws, plan["tiled"], n, cout, 3, row_order=plan["order"],
                                         tile_prefix=pre, bn_stats=True)
        assert torch.equal(out, K.conv_forward_split(f, ws, plan["tiled"], n, cout, 3,
                                                     row_order=plan["order"], tile_prefix=pre))
        assert part.shape == ((n + tr - 1) // tr, 2, cout)
        rows = out[plan["order"].long()].double()          # tile t = positions tr * t ..
        for ti in (0, part.shape[0] // 2, part.shape[0] - 1):
            blk = rows[tr * ti:tr * ti + tr]
            scale = max(blk.abs().max().item(), 1.0)
            assert (part[ti, 0].double() - blk.sum(0)).abs().max().item() <= 1e-4 * scale
            assert (part[ti, 1].double() - (blk * blk).sum(0)).abs().max().item() <= 1e-4 * scale ** 2
        tot = part.double().sum(0)
        assert (tot[0] - rows.sum(0)).abs().max().item() <= 1e-3
    if cout % 4 == 0:
        bn = torch.nn.BatchNorm1d(cout).to(dev).train()
        bn2 = torch.nn.BatchNorm1d(cout).to(dev).train()
        y1, m1, i1 = K.bn_act_forward(out, None, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                      True, 0.1, 1e-5, True)
        y2, m2, i2 = K.bn_act_forward(out, None, bn2.weight, bn2.bias, bn2.running_mean,
                                      bn2.running_var, True, 0.1, 1e-5, True, partials=part)
        assert (m1 - m2).abs().max().item() <= 1e-6 and (i1 / i2 - 1).abs().max().item() <= 1e-5
        assert (y1 - y2).abs().max().item() <= 1e-4
        assert (bn.running_var - bn2.running_var).abs().max().item() <= 1e-6


def test_scans_do_not_depend_on_what_else_runs(dev):
    """The scan-based index kernels (pair-list compaction, strided rulebook ranks, sparse_add
    maps) give the same tables while another stream keeps the chip busy with persistent conv
    kernels.  Round 3: block_range_sum left its LDS scratch unguarded and a block now and then
    took a wrong carry -- only when its waves were descheduled in between, i.e. under load.
    (An invariance check, not a reproducer: the library without the barrier passes this too;
    what exposed it was the training step itself, one bench leg in ~6 --
    a round-3 script (pruned; git history) repeated legs in one process.)"""
    from msmdfusion_amd import kernels as K
    shape = [21, 160, 160]
    idx = t(S.random_voxel_indices(60000, 2, shape, seed=5), dev)
    nbr = K.rulebook_subm(idx, 2, shape, 3)
    n = nbr.shape[1]
    want_pairs, want_num = K.rulebook_pairs(nbr)
    want_conv = K.rulebook_conv(idx, 2, shape, 3, 2, 1)
    idx_b = t(S.random_voxel_indices(50000, 2, shape, seed=6), dev)
    want_add = K.sparse_add_index(idx, idx_b, 2, shape)
    order = K.rulebook_tiling(nbr, want_table=False)[0]
    tab = K.permute_cols(nbr, order)
    pre = K.tile_prefix(tab, K.split_tile_rows(64))
    f = torch.randn(n, 64, device=dev)
    g = torch.randn(n, 64, device=dev)
    ws = K.pack_weight_split(torch.randn(27, 64, 64, device=dev) * 0.05, 3)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for _ in range(120):
        with torch.cuda.stream(side):    # one whole-CU workgroup per CU, 144 KB of LDS each
            K.conv_wgrad_split(f, g, want_pairs, want_num, 3)
            K.conv_forward_split(f, ws, tab, n, 64, 3, row_order=order, tile_prefix=pre)
        pairs, num = K.rulebook_pairs(nbr)
        assert torch.equal(num, want_num) and torch.equal(pairs, want_pairs)
        got = K.rulebook_conv(idx, 2, shape, 3, 2, 1)
        assert all(torch.equal(a, b) for a, b in zip(got[:3], want_conv[:3]))
        add = K.sparse_add_index(idx, idx_b, 2, shape)
        assert all(torch.equal(a, b) for a, b in zip(add, want_add))
    torch.cuda.synchronize()


# --------------------------------------------- split-bf16 ("fp32-equivalent") convolution
SPLIT_CHANNELS = [(32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (96, 96), (64, 32),
                  (128, 64), (128, 96),
                  # the fusion stack's widths: partial last k-block (c_in % 32 != 0), odd
                  # tile counts, c_out > 128 (two column passes)
                  (80, 80), (80, 96), (96, 128), (128, 192), (192, 192), (40, 72),
                  # round 5: 11 tiles in the 12-tile single pass, 10 tiles as two 5-tile
                  # passes of the 6-tile ping-pong instantiation, 13 tiles as 7 + 6
                  (96, 176), (64, 160), (32, 208)]


@pytest.mark.parametrize("planes", [3, 2])
@pytest.mark.parametrize("cin,cout", SPLIT_CHANNELS)
def test_split_conv_subm(dev, cin, cout, planes):
    """msmd_spconv_fwd_split (forward and dgrad) against the oracle; planes=3 must
    be as close to an fp64 evaluation as the fp32 MFMA kernel is."""
    from msmdfusion_amd import kernels as K
    assert K.split_supported(cin, cout) and K.split_supported(cout, cin)
    shape = [11, 64, 64]
    idx = S.random_voxel_indices(1500, 2, shape, seed=cin + cout)
    n = idx.shape[0]
    rng = np.random.RandomState(cin * 1000 + cout)
    f = rng.randn(n, cin).astype(np.float32)
    w = (rng.randn(27, cin, cout) / np.sqrt(27 * cin)).astype(np.float32)
    g = rng.randn(n, cout).astype(np.float32)
    oi, pr, nm, _ = O.get_indice_pairs(idx, 2, shape, 3, 1, 1, 1, True)
    exp = O.indice_conv_fwd(f, w, pr, nm, n, subm=True)
    edin, _ = O.indice_conv_bwd(f, w, g, pr, nm, subm=True)
    # three planes: north_star's 1e-4 element by element.  Two planes (three products, the
    # opt-in mode): 2e-4 element by element -- entries that are small through cancellation
    # see the dropped 2^-16 terms -- and, against the LARGEST output, what DESIGN.md 3.1
    # claims for it: <= 2e-5 (measured 4.3e-6; three planes 2.2e-6)
    tol = TOL if planes == 3 else 2 * TOL
    rel = 6e-6 if planes == 3 else 2e-5

    nbr = K.rulebook_subm(t(idx, dev), 2, shape, 3)
    wd, fd = t(w, dev), t(f, dev)
    ws = K.pack_weight_split(wd, planes)
    out = K.conv_forward_split(fd, ws, nbr, n, cout, planes)
    np.testing.assert_allclose(out.cpu().numpy(), exp, rtol=tol, atol=tol)
    assert np.abs(out.cpu().numpy() - exp).max() <= rel * np.abs(exp).max(), planes
    # one scheduling unit per tile: any tiling order gives bit-identical results (the
    # table travels in tile order).  With split tiles (the default) a heavy tile's sum
    # is (lower offsets) + (upper offsets): equal to rounding, and deterministic
    whole = K.conv_forward_split(fd, ws, nbr, n, cout, planes, split_tiles=False)
    np.testing.assert_allclose(whole.cpu().numpy(), exp, rtol=tol, atol=tol)
    scale = whole.abs().max().item()
    assert (out - whole).abs().max().item() <= 2e-6 * scale
    for order in (K.row_mask_order(nbr), torch.randperm(n, device=dev).int()):
        tab = K.permute_cols(nbr, order)
        out_o = K.conv_forward_split(fd, ws, tab, n, cout, planes, row_order=order,
                                     split_tiles=False)
        assert torch.equal(whole, out_o)
        out_s = K.conv_forward_split(fd, ws, tab, n, cout, planes, row_order=order)
        assert (out_s - whole).abs().max().item() <= 2e-6 * scale
        assert torch.equal(out_s, K.conv_forward_split(fd, ws, tab, n, cout, planes,
                                                       row_order=order))
        # stream-K: tiles cut at segment boundaries are summed in pieces, in ticket order
        pre = K.tile_prefix(tab, K.split_tile_rows(cout))
        out_k = K.conv_forward_split(fd, ws, tab, n, cout, planes, row_order=order,
                                     tile_prefix=pre)
        # (on this small input a segment is ~10 ranks: a tile is summed in 3-4 pieces;
        # reassociating ~5000 products of a row costs a few 1e-6 of the largest output)
        assert (out_k - whole).abs().max().item() <= 5e-6 * scale
        assert torch.equal(out_k, K.conv_forward_split(fd, ws, tab, n, cout, planes,
                                                       row_order=order, tile_prefix=pre))
    # KRSC weights pack to the same image
    w_krsc = wd.permute(2, 0, 1).contiguous().view(cout, 3, 3, 3, cin)
    assert torch.equal(ws, K.pack_weight_split(w_krsc, planes, krsc=True))
    # dgrad: forward table read with flipped weights and W^T
    din = K.conv_forward_split(t(g, dev), K.pack_weight_split(wd, planes, transpose=True), nbr,
                               n, cin, planes, weight_flip=True)
    np.testing.assert_allclose(din.cpu().numpy(), edin, rtol=tol, atol=tol)
    if planes == 3:   # fp32-equivalent: error against fp64 no worse than the fp32 kernel's
        ref = np.zeros((n, cout))
        nb = nbr.cpu().numpy()
        for k in range(27):
            m = nb[k] >= 0
            ref[m] += f[nb[k][m]].astype(np.float64) @ w[k].astype(np.float64)
        e_split = np.abs(out.cpu().numpy() - ref).max()
        if (cout + 15) // 16 in (12, 8, 6, 5, 4, 3, 2, 1):     # widths msmd_spconv_fwd_f32 takes
            e_fp32 = np.abs(K.conv_forward(fd, K.pack_weight(wd), nbr, n, cout).cpu().numpy()
                            - ref).max()
            assert e_split <= 2.0 * e_fp32 + 1e-7, (e_split, e_fp32)
        else:                           # (10, 11, 13 tiles: against the fp64 sum alone)
            assert e_split <= 4e-6 * np.abs(ref).max(), e_split


@pytest.mark.parametrize("ks,st,pd", [(3, 2, 1), (3, 2, [0, 1, 1]), ([3, 1, 1], [2, 1, 1], 0)])
@pytest.mark.parametrize("cin,cout", [(32, 64), (64, 128), (128, 128)])
def test_split_conv_strided(dev, ks, st, pd, cin, cout):
    from msmdfusion_amd import kernels as K
    shape = [11, 64, 64]
    idx = S.random_voxel_indices(1500, 2, shape, seed=11)
    n = idx.shape[0]
    rng = np.random.RandomState(7)
    f = rng.randn(n, cin).astype(np.float32)
    kvol = int(np.prod(O.expand3(ks)))
    w = (rng.randn(kvol, cin, cout) / np.sqrt(kvol * cin)).astype(np.float32)
    oi, pr, nm, osz = O.get_indice_pairs(idx, 2, shape, ks, st, pd, 1, False)
    m = oi.shape[0]
    g = rng.randn(m, cout).astype(np.float32)
    exp = O.indice_conv_fwd(f, w, pr, nm, m)
    edin, _ = O.indice_conv_bwd(f, w, g, pr, nm)
    _, _, perm = O.canonical_rulebook(oi, pr, nm, osz)

    _, nbr_fwd, nbr_bwd, _ = K.rulebook_conv(t(idx, dev), 2, shape, ks, st, pd)
    wd = t(w, dev)
    of = K.row_mask_order(nbr_fwd)
    out = K.conv_forward_split(t(f, dev), K.pack_weight_split(wd, 3), K.permute_cols(nbr_fwd, of),
                               m, cout, 3, row_order=of)
    np.testing.assert_allclose(out.cpu().numpy(), exp[perm], rtol=TOL, atol=TOL)
    ob = K.row_mask_order(nbr_bwd)
    din = K.conv_forward_split(t(g[perm], dev), K.pack_weight_split(wd, 3, transpose=True),
                               K.permute_cols(nbr_bwd, ob), n, cin, 3, row_order=ob)
    np.testing.assert_allclose(din.cpu().numpy(), edin, rtol=TOL, atol=TOL)


@pytest.mark.parametrize("planes", [3, 2])
@pytest.mark.parametrize("cin,cout", [(64, 64), (64, 128), (128, 64), (128, 128), (192, 64),
                                      (80, 80), (80, 96), (96, 128), (68, 100), (96, 96),
                                      (144, 80), (112, 160)])
def test_split_wgrad(dev, cin, cout, planes):
    """msmd_spconv_wgrad_split against the oracle (SubM and strided pair lists,
    ragged chunk tails, [K,Cin,Cout] and KRSC outputs)."""
    from msmdfusion_amd import kernels as K
    assert K.wgrad_split_supported(cin, cout) and not K.wgrad_split_supported(32, 64)
    shape = [11, 64, 64]
    idx = S.random_voxel_indices(2500, 2, shape, seed=cin + cout)
    n = idx.shape[0]
    rng = np.random.RandomState(cin + 7 * cout)
    f = rng.randn(n, cin).astype(np.float32)
    g = rng.randn(n, cout).astype(np.float32)
    w = np.zeros((27, cin, cout), np.float32)
    oi, pr, nm, _ = O.get_indice_pairs(idx, 2, shape, 3, 1, 1, 1, True)
    _, edw = O.indice_conv_bwd(f, w, g, pr, nm, subm=True)
    tol = TOL if planes == 3 else 2 * TOL      # (see test_split_conv_subm)
    nbr = K.rulebook_subm(t(idx, dev), 2, shape, 3)
    pairs, num = K.rulebook_pairs(nbr)
    dw = K.conv_wgrad_split(t(f, dev), t(g, dev), pairs, num, planes)
    np.testing.assert_allclose(dw.cpu().numpy(), edw, rtol=tol, atol=5 * tol)
    # relative to the largest entry: ~1e-6 with three planes, ~5e-6 with two
    assert np.abs(dw.cpu().numpy() - edw).max() <= (6e-6 if planes == 3 else 2e-5) * np.abs(edw).max()
    dwk = K.conv_wgrad_split(t(f, dev), t(g, dev), pairs, num, planes,
                             krsc_shape=(cout, 3, 3, 3, cin))
    assert torch.equal(dwk.view(cout, 27, cin).permute(1, 2, 0), dw)
    if planes == 3:   # as close to the oracle as the fp32 MFMA kernel
        d32 = K.conv_wgrad(t(f, dev), t(g, dev), pairs, num).cpu().numpy()
        assert np.abs(dw.cpu().numpy() - edw).max() <= 2 * np.abs(d32 - edw).max() + 1e-6
    # strided conv: compact lists of very different lengths, ld > pairs
    oi, pr, nm, osz = O.get_indice_pairs(idx, 2, shape, 3, 2, 1, 1, False)
    m = oi.shape[0]
    g2 = rng.randn(m, cout).astype(np.float32)
    _, edw2 = O.indice_conv_bwd(f, w, g2, pr, nm)
    _, _, perm = O.canonical_rulebook(oi, pr, nm, osz)
    _, nbr_fwd, _, _ = K.rulebook_conv(t(idx, dev), 2, shape, 3, 2, 1)
    pairs2, num2 = K.rulebook_pairs(nbr_fwd, ld=max(n, m))
    dw2 = K.conv_wgrad_split(t(f, dev), t(g2[perm], dev), pairs2, num2, planes)
    np.testing.assert_allclose(dw2.cpu().numpy(), edw2, rtol=tol, atol=5 * tol)


def test_wgrad_kernel_volume_above_64(dev):
    """5x5x5 SubM (K = 125), 64 -> 64: the whole-block wgrad kernel takes K <= 64, so no
    segment table exists for this rulebook (pair_segments -> None) and the weight gradient
    comes from the 64 x 64 slab kernel -- through IndiceData / autograd as a module would."""
    from msmdfusion_amd import kernels as K
    from msmdfusion_amd import spconv
    shape = [9, 30, 30]
    idx = S.random_voxel_indices(1200, 2, shape, seed=11)
    n = idx.shape[0]
    rng = np.random.RandomState(3)
    f = rng.randn(n, 64).astype(np.float32)
    g = rng.randn(n, 64).astype(np.float32)
    w = (rng.randn(125, 64, 64) / np.sqrt(125 * 64)).astype(np.float32)
    oi, pr, nm, _ = O.get_indice_pairs(idx, 2, shape, 5, 1, 2, 1, True)
    exp = O.indice_conv_fwd(f, w, pr, nm, n, subm=True)
    edin, edw = O.indice_conv_bwd(f, w, g, pr, nm, subm=True)
    conv = spconv.SubMConv3d(64, 64, 5, padding=2, bias=False).to(dev)
    with torch.no_grad():      # KRSC [c_out, kd, kh, kw, c_in]
        conv.weight.copy_(t(w, dev).permute(2, 0, 1).reshape(64, 5, 5, 5, 64))
    x = spconv.SparseConvTensor(t(f, dev).requires_grad_(True), t(idx, dev), shape, 2)
    rb = x.cached_rulebook([5, 5, 5], [1, 1, 1], [2, 2, 2], [1, 1, 1], True)
    rb.prepare(True, 64, 64)
    assert rb.pair_segments() is None
    y = conv(x).features
    np.testing.assert_allclose(y.detach().cpu().numpy(), exp, rtol=TOL, atol=TOL)
    y.backward(t(g, dev))
    np.testing.assert_allclose(x.features.grad.cpu().numpy(), edin, rtol=TOL, atol=TOL)
    dw = conv.weight.grad.reshape(64, 125, 64).permute(1, 2, 0)
    np.testing.assert_allclose(dw.cpu().numpy(), edw, rtol=TOL, atol=TOL * 5)
    pairs, num = rb.pairs()
    assert K.pair_segments(pairs, num) is None


@pytest.mark.parametrize("cin,cout,chunk_rows", [(64, 64, 256), (128, 96, 100), (192, 192, 512),
                                                 (96, 128, 5000)])
def test_split_wgrad_row_chunk_segments(dev, cin, cout, chunk_rows):
    """The whole-block wgrad kernel walking the pairs row chunk by row chunk
    (msmd_rulebook_pair_segments + msmd_spconv_wgrad_split_segments): the segment table cuts
    every offset's pair list where the output row crosses a chunk boundary; the result equals
    the oracle and the offset-major result to fp32 rounding, and is run-to-run identical.
    SubM and strided pair lists (n_in != n_out, ld > pairs, chunks without a pair)."""
    from msmdfusion_amd import kernels as K
    shape = [11, 64, 64]
    idx = S.random_voxel_indices(2500, 2, shape, seed=cin + cout + chunk_rows)
    n = idx.shape[0]
    rng = np.random.RandomState(cin + 11 * cout)
    f = rng.randn(n, cin).astype(np.float32)
    w = np.zeros((27, cin, cout), np.float32)
    for subm in (True, False):
        if subm:
            oi, pr, nm, _ = O.get_indice_pairs(idx, 2, shape, 3, 1, 1, 1, True)
            nbr = K.rulebook_subm(t(idx, dev), 2, shape, 3)
            m, perm = n, None
        else:
            oi, pr, nm, osz = O.get_indice_pairs(idx, 2, shape, 3, 2, 1, 1, False)
            m = oi.shape[0]
            _, _, perm = O.canonical_rulebook(oi, pr, nm, osz)
            _, nbr, _, _ = K.rulebook_conv(t(idx, dev), 2, shape, 3, 2, 1)
        g = rng.randn(m, cout).astype(np.float32)
        _, edw = O.indice_conv_bwd(f, w, g, pr, nm, subm=subm)
        pairs, num = K.rulebook_pairs(nbr, ld=max(n, m))
        seg = K.pair_segments(pairs, num, chunk_rows)
        table, n_chunks = seg
        ld = pairs.shape[2]
        assert n_chunks == (ld + chunk_rows - 1) // chunk_rows
        tb = table.cpu().numpy()
        NS = n_chunks * 27
        prefix, p0, cnt = tb[:NS + 1], tb[NS + 1:2 * NS + 1], tb[2 * NS + 1:]
        host_num = num.cpu().numpy()
        host_out = pairs.cpu().numpy()[:, 1, :]
        assert np.array_equal(cnt.reshape(n_chunks, 27).sum(0), host_num)
        assert np.array_equal(np.diff(prefix), (cnt + 31) // 32) and prefix[0] == 0
        for c in range(n_chunks):
            for k in range(27):
                rows = host_out[k, p0[c * 27 + k]:p0[c * 27 + k] + cnt[c * 27 + k]]
                assert ((rows >= c * chunk_rows) & (rows < (c + 1) * chunk_rows)).all()
        gg = t(g if perm is None else g[perm], dev)
        dw_seg = K.conv_wgrad_split(t(f, dev), gg, pairs, num, 3, segments=seg)
        dw_one = K.conv_wgrad_split(t(f, dev), gg, pairs, num, 3)
        np.testing.assert_allclose(dw_seg.cpu().numpy(), edw, rtol=TOL, atol=5 * TOL)
        scale = np.abs(edw).max()
        assert np.abs(dw_seg.cpu().numpy() - edw).max() <= 6e-6 * scale
        assert (dw_seg - dw_one).abs().max().item() <= 4e-6 * scale
        assert torch.equal(dw_seg, K.conv_wgrad_split(t(f, dev), gg, pairs, num, 3, segments=seg))
        dwk = K.conv_wgrad_split(t(f, dev), gg, pairs, num, 3, krsc_shape=(cout, 3, 3, 3, cin),
                                 segments=seg)
        assert torch.equal(dwk.view(cout, 27, cin).permute(1, 2, 0), dw_seg)


def test_split_conv_bf16_operands(dev):
    """planes=1 (MSMD_CONV_PLANES=1): plain bf16 operands, fp32 accumulate -- the
    arithmetic of configs[2]'s "bf16".  Equal to an fp64 evaluation on operands
    rounded to bf16 (up to fp32 accumulation error), and within bf16's 2^-9 relative
    operand rounding of the fp32 result."""
    from msmdfusion_amd import kernels as K
    shape = [11, 64, 64]
    idx = S.random_voxel_indices(1500, 2, shape, seed=5)
    n = idx.shape[0]
    rng = np.random.RandomState(5)
    f = rng.randn(n, 64).astype(np.float32)
    w = (rng.randn(27, 64, 128) / np.sqrt(27 * 64)).astype(np.float32)
    nbr = K.rulebook_subm(t(idx, dev), 2, shape, 3)
    fd, wd = t(f, dev), t(w, dev)
    out = K.conv_forward_split(fd, K.pack_weight_split(wd, 1), nbr, n, 128, 1).double()
    fb, wb = fd.bfloat16().double(), wd.bfloat16().double()
    ref_b = torch.zeros(n, 128, dtype=torch.float64, device=dev)
    ref = torch.zeros(n, 128, dtype=torch.float64, device=dev)
    for k in range(27):
        m = nbr[k] >= 0
        rows = nbr[k][m].long()
        ref_b[m] += fb[rows] @ wb[k]
        ref[m] += fd.double()[rows] @ wd.double()[k]
    assert (out - ref_b).abs().max().item() < 1e-5
    assert (out - ref).abs().max().item() < 2e-2 * ref.abs().max().item()


@pytest.mark.parametrize("c", [64, 192])
def test_split_conv_under_cu_contention(dev, c):
    """The persistent kernels' tile counter must be back at 0 after every launch
    even when workgroups become resident late because another stream holds the
    CUs (wgrad on its side stream, RCCL under DDP): a late workgroup can draw the
    launch's last ticket with its very first draw.  c = 192: the ping-pong kernel, whose
    weight buffers are guarded by COUNTED waits (LDS-DMA pieces waited for with the row
    gathers still in flight) -- bit-identical results while another stream's GEMMs compete
    for the memory system is what shows that those waits cover what they must."""
    from msmdfusion_amd import kernels as K
    shape = [21, 128, 128]
    idx = S.random_voxel_indices(40000, 2, shape, seed=3)
    n = idx.shape[0]
    nbr = K.rulebook_subm(t(idx, dev), 2, shape, 3)
    order = K.row_mask_order(nbr)
    nbr_t = K.permute_cols(nbr, order)
    f = torch.randn(n, c, device=dev)
    ws = K.pack_weight_split(torch.randn(27, c, c, device=dev) * 0.05, 3)
    ref = K.conv_forward_split(f, ws, nbr_t, n, c, 3, row_order=order)
    torch.cuda.synchronize()
    hog = torch.cuda.Stream()
    a = torch.randn(8192, 8192, device=dev)
    for rep in range(6):
        with torch.cuda.stream(hog):
            for _ in range(4):
                a = torch.mm(a, a) * 1e-4        # chip-filling GEMMs on the other stream
        outs = [K.conv_forward_split(f, ws, nbr_t, n, c, 3, row_order=order) for _ in range(4)]
        for o in outs:
            assert torch.equal(o, ref)
        assert int(K._tile_counter(f.device).abs().sum().item()) == 0    # counter and flags
    torch.cuda.synchronize()
    # the same under stream-K scheduling: an owner waits only for lower tickets, which are
    # resident by construction -- late workgroups cannot deadlock it, and the counter and
    # every exchange flag are back at 0 after each launch
    pre = K.tile_prefix(nbr_t, K.split_tile_rows(c))
    ref_k = K.conv_forward_split(f, ws, nbr_t, n, c, 3, row_order=order, tile_prefix=pre)
    assert (ref_k - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    torch.cuda.synchronize()
    for rep in range(6):
        with torch.cuda.stream(hog):
            for _ in range(4):
                a = torch.mm(a, a) * 1e-4
        outs = [K.conv_forward_split(f, ws, nbr_t, n, c, 3, row_order=order, tile_prefix=pre)
                for _ in range(4)]
        for o in outs:
            assert torch.equal(o, ref_k)
        assert int(K._tile_counter(f.device).abs().sum().item()) == 0
    torch.cuda.synchronize()


def test_tile_prefix(dev):
    """msmd_rulebook_tile_prefix == prefix sums of the tiles' stream-K cost (per active
    offset 12 + the number of 32-row groups it keeps busy; at least 1 per tile), including a
    partial last tile and an all-empty tile."""
    from msmdfusion_amd import kernels as K
    shape = [11, 64, 64]
    idx = S.random_voxel_indices(1500, 2, shape, seed=8)
    nbr = K.rulebook_subm(t(idx, dev), 2, shape, 3)
    order = K.row_mask_order(nbr)
    tab = K.permute_cols(nbr, order)
    tab[:, 256:384] = -1                   # one tile without any neighbour
    tn = tab.cpu().numpy()
    n = tn.shape[1]
    for rows in (128, 256):
        got = K.tile_prefix(tab, rows).cpu().numpy()
        w = []
        for t0 in range(0, n, rows):
            pos = np.minimum(np.arange(t0, t0 + rows), n - 1)       # past the end: the last row
            act = (tn[:, pos] >= 0).reshape(tn.shape[0], rows // 32, 32).any(2)   # [K, groups]
            groups = act.sum(1)
            w.append(max(int((12 + groups[groups > 0]).sum()), 1))    # kSkC1Default = 12
        assert np.array_equal(got, np.concatenate([[0], np.cumsum(w)]))
        if rows == 128:
            assert w[2] == 1
    # such a tile's rows still come out as zeros under stream-K
    f = torch.randn(n, 32, device=dev)
    ws = K.pack_weight_split(torch.randn(27, 32, 32, device=dev) * 0.1, 3)
    out = K.conv_forward_split(f, ws, tab, n, 32, 3, row_order=order, tile_prefix=K.tile_prefix(tab, K.split_tile_rows(32)))
    whole = K.conv_forward_split(f, ws, tab, n, 32, 3, row_order=order, split_tiles=False)
    assert (out - whole).abs().max().item() <= 2e-6 * whole.abs().max().item()
    assert not out[order[256:384].long()].abs().any()


def test_split_conv_edges(dev):
    """Empty and tiny inputs, a single tile with padding rows, unsupported shapes."""
    from msmdfusion_amd import kernels as K
    from msmdfusion_amd._lib import MsmdError
    assert not K.split_supported(16, 16) and not K.split_supported(5, 16)
    assert not K.split_supported(36, 64) and not K.split_supported(32, 30)
    assert K.split_supported(48, 64) and K.split_supported(192, 192)
    w = torch.randn(27, 32, 64, device=dev)
    ws = K.pack_weight_split(w, 3)
    for n in (0, 1, 3, 129):
        f = torch.randn(n, 32, device=dev)
        nbr = torch.full((27, n), -1, dtype=torch.int32, device=dev)
        if n:
            nbr[13] = torch.arange(n, dtype=torch.int32, device=dev)
        out = K.conv_forward_split(f, ws, nbr, n, 64, 3)
        ref = f.double() @ w[13].double()
        assert out.shape == (n, 64)
        if n:
            np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-5)
    with pytest.raises(MsmdError):
        K.conv_forward_split(torch.randn(4, 16, device=dev), ws,
                             torch.zeros((27, 4), dtype=torch.int32, device=dev), 4, 16, 3)
    # split tiles whose second half is empty (one or no active offset per tile) and
    # tiles with exactly two offsets; 128 -> 128 is a layer the halves are used on
    w = torch.randn(27, 128, 128, device=dev) * 0.1
    ws = K.pack_weight_split(w, 3)
    for n, ks in [(1, [13]), (200, [13]), (1000, [4, 13]), (300, [])]:
        f = torch.randn(n, 128, device=dev)
        nbr = torch.full((27, n), -1, dtype=torch.int32, device=dev)
        ref = torch.zeros(n, 128, dtype=torch.float64, device=dev)
        for k in ks:
            src = torch.randperm(n, device=dev).int()
            nbr[k] = src
            ref += f.double()[src.long()] @ w[k].double()
        for _ in range(2):      # twice: the flags must be back at 0 after the first launch
            out = K.conv_forward_split(f, ws, nbr, n, 128, 3)
            np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-5)
        assert int(K._tile_counter(dev).abs().sum().item()) == 0


# ------------------------------------------------------------------ dense / sets
@pytest.mark.parametrize("c", [1, 5, 128, 192])
def test_dense_scatter_gather(dev, c):
    from msmdfusion_amd import kernels as K
    shape = [2, 45, 45]
    idx = S.random_voxel_indices(900, 3, shape, seed=c)
    f = np.random.RandomState(c).randn(idx.shape[0], c).astype(np.float32)
    exp = O.dense(f, idx, 3, shape)
    out = K.dense_scatter(t(f, dev), t(idx, dev), 3, shape)
    assert np.array_equal(out.cpu().numpy(), exp)
    back = K.dense_gather(out, t(idx, dev), shape)
    assert np.array_equal(back.cpu().numpy(), f)


def test_sparse_add(dev):
    from msmdfusion_amd import kernels as K
    shape = [11, 90, 90]
    a = S.random_voxel_indices(3000, 2, shape, seed=1)
    b = np.concatenate([a[::3], S.random_voxel_indices(2000, 2, shape, seed=2)])
    b = b[np.sort(np.unique(b, axis=0, return_index=True)[1])]
    rng = np.random.RandomState(0)
    fa, fb = rng.randn(a.shape[0], 96).astype(np.float32), rng.randn(b.shape[0], 96).astype(np.float32)
    eoi, eof, ema, emb = O.sparse_add(fa, a, fb, b, shape)
    oi, of, ma, mb = K.sparse_add(t(fa, dev), t(a, dev), t(fb, dev), t(b, dev), 2, shape)
    assert np.array_equal(oi.cpu().numpy(), eoi)
    assert np.array_equal(ma.cpu().numpy(), ema) and np.array_equal(mb.cpu().numpy(), emb)
    np.testing.assert_allclose(of.cpu().numpy(), eof, rtol=1e-6, atol=1e-6)
    # empty operand
    z = torch.zeros((0, 4), dtype=torch.int32, device=dev)
    zf = torch.zeros((0, 96), device=dev)
    oi2, of2, _, _ = K.sparse_add(t(fa, dev), t(a, dev), zf, z, 2, shape)
    assert oi2.shape[0] == a.shape[0]


def test_modality_split(dev):
    from msmdfusion_amd import kernels as K
    shape = [41, 300, 300]
    a = S.random_voxel_indices(5000, 2, shape, seed=5)
    b = np.concatenate([a[1::4], S.random_voxel_indices(3000, 2, shape, seed=6)])
    b = b[np.sort(np.unique(b, axis=0, return_index=True)[1])]
    m3, m2, p3, p2 = K.modality_split(t(a, dev), t(b, dev), 2, shape)
    e3, e2, ep3, ep2 = [], [], [], []
    off3 = off2 = 0
    for bi in range(2):   # the reference walks sample by sample (MSMDFusion.py:262)
        sa, sb = a[a[:, 0] == bi], b[b[:, 0] == bi]
        x3, x2, q3, q2 = O.modality_split(sa[:, 1:], sb[:, 1:], shape)
        e3.append(x3); e2.append(x2)
        ep3.append(np.flatnonzero(a[:, 0] == bi)[q3]); ep2.append(np.flatnonzero(b[:, 0] == bi)[q2])
    assert np.array_equal(m3.cpu().numpy()[np.argsort(a[:, 0], kind="stable")], np.concatenate(e3))
    assert np.array_equal(m2.cpu().numpy()[np.argsort(b[:, 0], kind="stable")], np.concatenate(e2))
    assert np.array_equal(p3.cpu().numpy(), np.concatenate(ep3))
    assert np.array_equal(p2.cpu().numpy(), np.concatenate(ep2))


# ------------------------------------------------------------------ GMA-Conv helpers
@pytest.mark.parametrize("n,m", [(5, 3), (1000, 64), (3000, 512), (20000, 256), (30000, 64)])
def test_fps(dev, n, m):
    from msmdfusion_amd import kernels as K
    rng = np.random.RandomState(n)
    # integer voxel coordinates: ties are the common case
    xyz = np.stack([rng.randint(0, 41, (2, n)), rng.randint(0, 200, (2, n)),
                    rng.randint(0, 200, (2, n))], -1).astype(np.float32)
    exp = O.furthest_point_sample(xyz, m)
    got = K.furthest_point_sample(t(xyz, dev), m)
    assert np.array_equal(got.cpu().numpy(), exp)


@pytest.mark.parametrize("order", ["first-touch", "shuffled"])
def test_fps_pruned_lc_shape(dev, order):
    """The bucket-pruned kernel at the LC stage-0 shape (2 x ~22k voxel coordinates,
    2048 samples), on spatially coherent input (buckets get skipped) and on the
    same points shuffled (nothing can be skipped): bit-identical to the oracle
    both ways."""
    from msmdfusion_amd import kernels as K
    clouds = []
    for seed in range(2):
        p = S.virtual_points(seed, n=50000)[:, :3]
        c = np.floor((p - np.array(S.POINT_CLOUD_RANGE[:3])) / np.array(S.VOXEL_SIZE))
        c = c.astype(np.int64)[:, ::-1]
        _, first = np.unique(c, axis=0, return_index=True)
        clouds.append(np.ascontiguousarray(c[np.sort(first)], dtype=np.float32))
    n = min(c.shape[0] for c in clouds)
    assert 15000 < n <= 24576
    xyz = np.stack([c[:n] for c in clouds])
    if order == "shuffled":
        rng = np.random.RandomState(0)
        xyz = np.stack([x[rng.permutation(n)] for x in xyz])
    exp = O.furthest_point_sample(xyz, 2048)
    got = K.furthest_point_sample(t(xyz, dev), 2048)
    assert np.array_equal(got.cpu().numpy(), exp)
    # non-integer coordinates (no ties, rounding matters for the box bound)
    rng = np.random.RandomState(1)
    fxyz = (xyz * np.float32(0.075) + rng.rand(*xyz.shape).astype(np.float32) * 0.01)
    exp = O.furthest_point_sample(fxyz, 512)
    got = K.furthest_point_sample(t(fxyz, dev), 512)
    assert np.array_equal(got.cpu().numpy(), exp)


def test_fps_ragged_batch(dev):
    """One launch over elements of different sizes == per-element FPS."""
    from msmdfusion_amd import kernels as K
    rng = np.random.RandomState(0)
    sizes = [700, 5000, 22000, 1, 3000]
    parts = [np.stack([rng.randint(0, 41, n), rng.randint(0, 300, n), rng.randint(0, 300, n)],
                      -1).astype(np.float32) for n in sizes]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    got = K.furthest_point_sample_ragged(t(np.concatenate(parts), dev), t(offs, dev), max(sizes),
                                         64).cpu().numpy()
    for i, p in enumerate(parts):
        m = min(64, sizes[i]) if sizes[i] > 1 else 64
        exp = O.furthest_point_sample(p[None], 64)[0]
        if sizes[i] >= 64:
            assert np.array_equal(got[i], exp), i
        else:   # fewer points than samples: the reference keeps re-selecting; compare prefix
            assert np.array_equal(got[i][:1], exp[:1])


def test_ball_query_and_assign(dev):
    from msmdfusion_amd import kernels as K
    rng = np.random.RandomState(1)
    n, m, ns = 6000, 300, 50
    xyz = np.stack([rng.randint(0, 11, (1, n)), rng.randint(0, 120, (1, n)),
                    rng.randint(0, 120, (1, n))], -1).astype(np.float32)
    centers = xyz[:, rng.choice(n, m, replace=False)]
    exp = O.ball_query(0, 6.0, ns, xyz, centers)
    got = K.ball_query(0, 6.0, ns, t(xyz, dev), t(centers, dev))
    assert np.array_equal(got.cpu().numpy(), exp)
    rep_nn = rng.randint(-1, 500, m).astype(np.int32)
    ea = O.nn_assign(exp[0], rep_nn, n)
    ga = K.nn_assign(got[0], t(rep_nn, dev), n)
    assert np.array_equal(ga.cpu().numpy(), ea)


def test_nn_search(dev):
    from msmdfusion_amd import kernels as K
    rng = np.random.RandomState(2)
    q = np.stack([rng.randint(0, 41, 700), rng.randint(0, 300, 700), rng.randint(0, 300, 700)], -1)
    k = np.stack([rng.randint(0, 41, 9000), rng.randint(0, 300, 9000), rng.randint(0, 300, 9000)], -1)
    exp = O.nn_search(q, k, 13.3)
    got = K.nn_search(t(q.astype(np.int32), dev), t(k.astype(np.int32), dev), 13.3)
    assert np.array_equal(got.cpu().numpy(), exp)


@pytest.mark.gpu
def test_sparse_add_index_then_rows_equals_fused(dev):
    """The split sparse_add (index pass ahead of time, feature pass with its maps)
    returns exactly what the fused call does -- including empty operands."""
    from msmdfusion_amd import kernels as K
    g = torch.Generator().manual_seed(3)
    shape = [11, 60, 60]
    for na, nb in [(5000, 7000), (0, 300), (300, 0), (1, 1)]:
        def coords(n):
            lin = torch.randperm(2 * shape[0] * shape[1] * shape[2], generator=g)[:n]
            b, r = lin // (shape[0] * shape[1] * shape[2]), lin % (shape[0] * shape[1] * shape[2])
            return torch.stack([b, r // (shape[1] * shape[2]), (r // shape[2]) % shape[1],
                                r % shape[2]], 1).int().to(dev)
        ia, ib = coords(na), coords(nb)
        fa = torch.randn(na, 48, generator=g).to(dev)
        fb = torch.randn(nb, 48, generator=g).to(dev)
        oi, of, ma, mb = K.sparse_add(fa, ia, fb, ib, 2, shape)
        oi2, ma2, mb2 = K.sparse_add_index(ia, ib, 2, shape)
        of2 = K.sparse_add_rows(fa, ma2, fb, mb2, oi2.shape[0])
        assert torch.equal(oi, oi2) and torch.equal(ma, ma2) and torch.equal(mb, mb2)
        assert torch.equal(of, of2)


def test_sparse_add_rows_gather(dev):
    """The gather form of sparse_add's feature half (msmd_rows_inverse +
    msmd_sparse_add_rows_gather) == the atomic one: bit for bit when no tensor repeats a
    coordinate (every output row is one a-row + one b-row, in that order, or one of them),
    to rounding when one does (the repeated rows go through the fix-up pass); empty operands;
    every output row written (the buffer starts as NaN)."""
    from msmdfusion_amd import kernels as K
    g = torch.Generator().manual_seed(11)
    shape = [11, 60, 60]
    cells = 2 * shape[0] * shape[1] * shape[2]

    def coords(n, dup=0):
        lin = torch.randperm(cells, generator=g)[:n]
        if dup:
            lin = torch.cat([lin, lin[:dup]])
        b, r = lin // (cells // 2), lin % (cells // 2)
        return torch.stack([b, r // (shape[1] * shape[2]), (r // shape[2]) % shape[1],
                            r % shape[2]], 1).int().to(dev)
    for na, nb, dup, c in [(5000, 7000, 0, 192), (0, 300, 0, 16), (300, 0, 0, 64), (1, 1, 0, 4),
                           (3000, 2000, 57, 80)]:
        ia, ib = coords(na, dup), coords(nb, dup // 2)
        fa = torch.randn(ia.shape[0], c, generator=g).to(dev)
        fb = torch.randn(ib.shape[0], c, generator=g).to(dev)
        oi, ma, mb = K.sparse_add_index(ia, ib, 2, shape)
        n_out = oi.shape[0]
        want = K.sparse_add_rows(fa, ma, fb, mb, n_out)
        inv_a, inv_b = K.rows_inverse(ma, n_out), K.rows_inverse(mb, n_out)
        # inv: the last row that maps to j (or -1)
        ref = torch.full((n_out,), -1, dtype=torch.int64, device=dev)
        if ma.shape[0]:
            ref.scatter_reduce_(0, ma.long(), torch.arange(ma.shape[0], device=dev), "amax",
                                include_self=True)
        assert torch.equal(inv_a.long(), ref)
        got = K.sparse_add_rows_gather(fa, ma, inv_a, fb, mb, inv_b, n_out)
        assert not torch.isnan(got).any()
        if dup == 0:
            assert torch.equal(got, want), (na, nb, c)
        else:
            np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-6, atol=1e-6)
            assert int((inv_a >= 0).sum()) < ma.shape[0]      # (some rows did repeat)


@pytest.mark.parametrize("n", [1, 127, 128, 129, 5000])
@pytest.mark.parametrize("kvol", [27, 3])
def test_rulebook_tiling_one_call(dev, n, kvol):
    """msmd_rulebook_tiling against row_mask_order + permute_cols (the torch-side
    route): a permutation of the rows, the table in that order, and tiles that cost
    the same (ties between equal keys may be broken differently).  The one-call tiling
    keeps the tiles in mask order (stream-K balances them wherever they lie); the
    heaviest-first re-sequencing of whole tiles is MSMD_TILE_LPT=1 / tile_lpt=True."""
    from msmdfusion_amd import kernels as K
    g = torch.Generator().manual_seed(n + kvol)
    nbr = torch.where(torch.rand(kvol, n, generator=g) < 0.35,
                      torch.randint(0, max(n, 1), (kvol, n), generator=g), -1).int().to(dev)
    order, tiled = K.rulebook_tiling(nbr)
    assert sorted(order.cpu().tolist()) == list(range(n))
    assert torch.equal(tiled, K.permute_cols(nbr, order))
    ref = K.row_mask_order(nbr, tile_lpt=os.environ.get("MSMD_TILE_LPT", "0") == "1")

    def unions(o):      # offsets each 128-row tile walks
        m = (nbr[:, o.long()] >= 0)
        pad = (-n) % 128
        m = torch.cat([m, torch.zeros(kvol, pad, dtype=torch.bool, device=dev)], 1)
        return m.view(kvol, -1, 128).any(2).sum(0)
    if kvol == 27:
        assert torch.equal(unions(order), unions(ref))
    only_order, none = K.rulebook_tiling(nbr, want_table=False)
    assert none is None and torch.equal(only_order, order)


@pytest.mark.parametrize("cin,cout,krsc", [(64, 128, False), (128, 128, True), (96, 32, True)])
def test_pack_weight_split_pair(dev, cin, cout, krsc):
    from msmdfusion_amd import kernels as K
    w = torch.randn(27, cin, cout, device=dev)
    if krsc:
        w = w.permute(2, 0, 1).contiguous().view(cout, 3, 3, 3, cin)
    for planes in (3, 1):
        a, b = K.pack_weight_split_pair(w, planes, krsc=krsc)
        assert torch.equal(a, K.pack_weight_split(w, planes, krsc=krsc))
        assert torch.equal(b, K.pack_weight_split(w, planes, transpose=True, krsc=krsc))


def test_hard_voxelize_many_equals_cloud_by_cloud(dev, monkeypatch):
    """msmd_hard_voxelize_many (every cloud of a batch in one launch set: blocks find their
    cloud, the scan restarts per cloud, one fill for all tables) == msmd_hard_voxelize called
    cloud by cloud, bit for bit: mixed channel counts and voxel sizes (the LiDAR sweeps + the
    four virtual-point scales of an LC step), a cloud below one scan tile, an EMPTY cloud, a
    cloud that hits max_voxels (the reference loop's `break`), 14 clouds = two launch sets;
    and against the oracle for one of them."""
    from msmdfusion_amd import kernels as K
    rng = np.random.RandomState(12)
    rg = S.POINT_CLOUD_RANGE
    base = S.VOXEL_SIZE

    def cloud(n, c, spread=1.0):
        p = rng.randn(n, c).astype(np.float32)
        p[:, 0] = rng.uniform(rg[0] * spread, rg[3] * spread, n)
        p[:, 1] = rng.uniform(rg[1] * spread, rg[4] * spread, n)
        p[:, 2] = rng.uniform(rg[2], rg[5], n)
        return p
    clouds = [S.lidar_sweep(0, n_az=400), S.lidar_sweep(1, n_az=300), cloud(5000, 64),
              cloud(1500, 64, 0.2), cloud(100, 5), np.zeros((0, 5), np.float32),
              cloud(30000, 5, 0.05)]
    sizes = [list(base), list(base), [v * 2 for v in base], [v * 4 for v in base],
             [v * 8 for v in base], list(base), list(base)]
    # 14 clouds: a second launch set (12 per set)
    clouds = clouds + [c[::-1].copy() for c in clouds]
    sizes = sizes + sizes
    d = [t(c, dev) for c in clouds]
    for max_voxels in (20000, 700):        # 700: several clouds run into the cap
        for want_voxels, want_mean in ((True, False), (False, True), (True, True)):
            monkeypatch.setattr(K, "VOXELIZE_MANY", True)
            got = K.hard_voxelize_batch(d, sizes, rg, 10, max_voxels, want_voxels, want_mean)
            monkeypatch.setattr(K, "VOXELIZE_MANY", False)
            ref = K.hard_voxelize_batch(d, sizes, rg, 10, max_voxels, want_voxels, want_mean)
            assert len(got) == len(ref) == len(clouds)
            for ci, (a, b) in enumerate(zip(got, ref)):
                for x, y in zip(a, b):
                    assert (x is None) == (y is None), ci
                    if x is not None:
                        assert x.shape == y.shape and torch.equal(x, y), (ci, max_voxels)
            assert got[5][1].shape[0] == 0                       # the empty cloud
        if max_voxels == 700:
            assert sum(int(g[1].shape[0] == 700) for g in got) >= 4
    monkeypatch.setattr(K, "VOXELIZE_MANY", True)
    got = K.hard_voxelize_batch(d, sizes, rg, 10, 20000, True, False)
    ev, ec, en = O.hard_voxelize(clouds[2], sizes[2], rg, 10, 20000)
    assert np.array_equal(got[2][1].cpu().numpy(), ec) and np.array_equal(got[2][2].cpu().numpy(), en)
    assert np.array_equal(got[2][0].cpu().numpy(), ev)
