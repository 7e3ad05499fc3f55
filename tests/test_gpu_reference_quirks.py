"""`reference_quirks=True`: the reference's float32-key voxel_modality_split
(mmdet3d/models/detectors/MSMDFusion.py:271-272 + type_assign :27-45) and its batch-offset
arithmetic (:288-289,313-314; sparse_multimodal_encoder_painting.py:355-369), reproduced bit
for bit by csrc/modality_float.hip and the modules above it -- against the oracle's float-key
restatement, on grids where the keys DO alias (z >= 17, x >= 1000 at the 0.075 m scale)."""
import numpy as np
import pytest
import torch

from msmdfusion_amd import synthetic as S
from oracle import oracle as O
from test_gpu_fusion import _oracle_stage
from test_gpu_modules import _np

pytestmark = pytest.mark.gpu

SHAPE0 = [41, 1440, 1440]


def _aliasing_cloud(rng, n, b, ymin=100, ymax=104):
    """Voxels packed around a few (z, y) lines with z >= 17 and x around 1000: float keys
    alias (spacing 2 above 2^24, 4 above 2^25; x >= 1000 runs into the next y) and true
    matches exist too."""
    z = rng.randint(17, 41, n)
    y = rng.randint(ymin, ymax, n)
    x = rng.randint(990, 1440, n)
    u = np.unique(np.stack([np.full(n, b), z, y, x], 1), axis=0)
    return u[rng.permutation(u.shape[0])].astype(np.int32)


def _oracle_split(i3, i2, shape, batch, float_keys, reference_offsets=False):
    """Per-sample oracle split, rows numbered globally or as the reference numbers them."""
    e3, e2, p3, p2 = [], [], [], []
    last3 = last2 = 0
    for b in range(batch):
        r3, r2 = np.flatnonzero(i3[:, 0] == b), np.flatnonzero(i2[:, 0] == b)
        m3, m2, q3, q2 = O.modality_split(i3[r3, 1:], i2[r2, 1:], shape, float_keys=float_keys)
        e3.append(m3)
        e2.append(m2)
        if reference_offsets:      # position in the sample + the PREVIOUS sample's count
            p3.append(q3 + last3)
            p2.append(q2 + last2)
        else:
            p3.append(r3[q3])
            p2.append(r2[q2])
        last3, last2 = len(r3), len(r2)
    return np.concatenate(e3), np.concatenate(e2), np.concatenate(p3), np.concatenate(p2)


@pytest.mark.parametrize("batch", [1, 2, 3])
def test_float_key_split_kernel_matches_the_oracle(dev, batch):
    from msmdfusion_amd import kernels as K
    rng = np.random.RandomState(20 + batch)
    i3 = np.concatenate([_aliasing_cloud(rng, 3000 - 400 * b, b) for b in range(batch)])
    i2 = np.concatenate([_aliasing_cloud(rng, 2000 + 700 * b, b) for b in range(batch)])
    d3, d2 = torch.from_numpy(i3).to(dev), torch.from_numpy(i2).to(dev)
    for ref_off in (False, True):
        m3, m2, p3, p2 = K.modality_split(d3, d2, batch, SHAPE0, float_keys=True,
                                          reference_offsets=ref_off)
        e3, e2, q3, q2 = _oracle_split(i3, i2, SHAPE0, batch, True, ref_off)
        assert np.array_equal(_np(m3), e3) and np.array_equal(_np(m2), e2)
        assert np.array_equal(_np(p3), q3) and np.array_equal(_np(p2), q2)
        if batch <= 2:      # the reference's offsets ARE the global rows there
            g3, g2 = _oracle_split(i3, i2, SHAPE0, batch, True, False)[2:]
            assert np.array_equal(q3, g3) and np.array_equal(q2, g2)
    # the aliasing is real on this data, and differs from the exact-key default
    e3, _, q3, q2 = _oracle_split(i3, i2, SHAPE0, batch, True)
    assert int((i3[q3] != i2[q2]).any(1).sum()) > 0
    x3, x2, xp3, xp2 = K.modality_split(d3, d2, batch, SHAPE0)
    assert not np.array_equal(_np(x3), e3)
    assert (i3[_np(xp3)] == i2[_np(xp2)]).all()
    # the per-sample statistics that come back with the split (one host read for all jobs)
    (f3, f2, fp3, fp2, stats), = K.modality_split_many([(d3, d2, SHAPE0)], batch, float_keys=True,
                                                       reference_offsets=True)
    e3, e2, q3, q2 = _oracle_split(i3, i2, SHAPE0, batch, True, True)
    assert np.array_equal(_np(fp3), q3) and np.array_equal(_np(fp2), q2)
    for b in range(batch):
        s3, s2 = i3[:, 0] == b, i2[:, 0] == b
        assert stats["c3_mixed"][b] == int(e3[s3].sum()) and stats["c3_plain"][b] == int((1 - e3[s3]).sum())
        assert stats["c2_mixed"][b] == int(e2[s2].sum()) and stats["c2_plain"][b] == int((1 - e2[s2]).sum())
        assert stats["c3_mixed"][b] == stats["c2_mixed"][b]


@pytest.mark.parametrize("tag,batch", [("b1", 1), ("b2", 2), ("b3", 3)])
def test_float_key_split_kernel_vs_the_references_own_function(dev, tag, batch):
    """tests/golden/modality_split_vectors.npz holds what the REFERENCE's voxel_modality_split +
    type_assign returned (make_modality_split_golden.py executes MSMDFusion.py:251-325,27-45
    as they stand) for voxel sets whose float32 keys collide across the sets but never inside
    one (so the reference's unspecified tie order plays no part): msmd_modality_split_float_keys
    with the reference's batch offsets == those flags and syn_mix lists, bit for bit -- at
    batch 3 the reference's own (non-cumulative, i.e. wrong) rows included."""
    import os
    from msmdfusion_amd import kernels as K
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                             "modality_split_vectors.npz"))
    i3, i2 = g[tag + "_idx3"], g[tag + "_idx2"]
    m3, m2, p3, p2 = K.modality_split(torch.from_numpy(i3).to(dev), torch.from_numpy(i2).to(dev),
                                      batch, SHAPE0, float_keys=True, reference_offsets=True)
    assert np.array_equal(_np(m3), g[tag + "_mix3"]) and np.array_equal(_np(m2), g[tag + "_mix2"])
    assert np.array_equal(_np(p3).astype(np.int64), g[tag + "_syn3"])
    assert np.array_equal(_np(p2).astype(np.int64), g[tag + "_syn2"])
    if batch <= 2:      # global rows there: the false matches are visible as unequal coordinates
        assert int((i3[_np(p3)] != i2[_np(p2)]).any(1).sum()) > 500
    x3 = K.modality_split(torch.from_numpy(i3).to(dev), torch.from_numpy(i2).to(dev), batch,
                          SHAPE0)[0]
    assert int(x3.sum()) < int(g[tag + "_mix3"].sum())


def test_float_key_split_kernel_edges(dev):
    from msmdfusion_amd import kernels as K
    from msmdfusion_amd._lib import MsmdError
    rng = np.random.RandomState(1)
    i3 = torch.from_numpy(_aliasing_cloud(rng, 500, 0)).to(dev)
    none = torch.zeros((0, 4), dtype=torch.int32, device=dev)
    m3, m2, p3, p2 = K.modality_split(i3, none, 1, SHAPE0, float_keys=True)
    assert m3.shape[0] == i3.shape[0] and int(m3.sum()) == 0 and p3.shape[0] == 0 == p2.shape[0]
    m3, m2, p3, p2 = K.modality_split(none, i3, 1, SHAPE0, float_keys=True)
    assert m2.shape[0] == i3.shape[0] and int(m2.sum()) == 0 and p3.shape[0] == 0
    # a set against itself: everything matches itself in key order... unless keys repeat inside
    # the set, where the r-th occurrence pairs with the r-th (row order): still the identity
    m3, m2, p3, p2 = K.modality_split(i3, i3, 1, SHAPE0, float_keys=True)
    assert int(m3.sum()) == i3.shape[0] and torch.equal(p3, p2)
    # a grid whose largest key does not fit 26 bits is refused, not mangled
    with pytest.raises(MsmdError):
        K.modality_split(i3, i3, 1, [80, 1440, 1440], float_keys=True)
    # ... judged in the kernel's own float32 arithmetic: [68, 109, 863] has its largest key at
    # 2^26 - 2 in double but exactly 2^26 in float32 (spacing 4 up there, ties to even)
    with pytest.raises(MsmdError):
        K.modality_split(i3[:1] * 0, i3[:1] * 0, 1, [68, 109, 863], float_keys=True)
    # rows outside the grid (negative ones included) are refused on the device
    for bad_row in ([0, 41, 0, 0], [0, 0, -1, 0], [1, 0, 0, 0]):
        bad = torch.cat([i3, torch.tensor([bad_row], dtype=torch.int32, device=dev)])
        with pytest.raises(ValueError):
            K.modality_split(bad, i3, 1, SHAPE0, float_keys=True)
        with pytest.raises(ValueError):
            K.modality_split_many([(i3, bad, SHAPE0)], 1, float_keys=True)
    # reference_offsets numbers rows inside their sample: ungrouped rows are refused
    mixed_up = torch.from_numpy(np.concatenate([_aliasing_cloud(rng, 50, 1),
                                                _aliasing_cloud(rng, 50, 0)])).to(dev)
    with pytest.raises(ValueError):
        K.modality_split(mixed_up, mixed_up, 2, SHAPE0, float_keys=True, reference_offsets=True)
    K.modality_split(mixed_up, mixed_up, 2, SHAPE0, float_keys=True)     # fine without them
    # where the keys cannot alias (z <= 15, x < 1000) both modes agree
    a = S.random_voxel_indices(3000, 2, [16, 300, 300], seed=3)
    b = np.concatenate([a[::3], S.random_voxel_indices(2000, 2, [16, 300, 300], seed=4)])
    b = b[np.sort(np.unique(b, axis=0, return_index=True)[1])]
    a, b = a[np.argsort(a[:, 0], kind="stable")], b[np.argsort(b[:, 0], kind="stable")]
    da, db = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    f = K.modality_split(da, db, 2, [16, 300, 300], float_keys=True, reference_offsets=True)
    x = K.modality_split(da, db, 2, [16, 300, 300])
    assert all(torch.equal(u, v) for u, v in zip(f, x))


def _spacing2_cloud(rng, n, b, odd_share):
    """z in 17..32 (keys in [2^24, 2^25): float spacing 2), x < 1000: a voxel at odd x shares
    its key with the even neighbour the rounding picks (ties to even: the multiple of 4),
    voxels at even x are exact.  odd_share: fraction of odd-x voxels."""
    z = rng.randint(17, 33, n)
    y = rng.randint(100, 105, n)
    x = 2 * rng.randint(200, 420, n) + (rng.rand(n) < odd_share)
    u = np.unique(np.stack([np.full(n, b), z, y, x], 1), axis=0)
    return u[rng.permutation(u.shape[0])].astype(np.int32)


def test_gma_stage_with_reference_quirks_matches_the_oracle(dev):
    """GMA-Conv stage 0 on the scale-1 grid with aliasing voxel sets: the module path in
    reference mode == the oracle walk with float keys, false "mixed" voxels included.
    LiDAR voxels at even x only, virtual-point voxels at even AND odd x: an odd-x 2D voxel
    is paired with the even-x 3D voxel its rounded key lands on (a false match), but no
    mixed voxel can land on an only-3D voxel's coordinate -- the unified set has no repeated
    coordinate.  (Where it does -- real data -- the reference's result depends on which of
    two racing hash inserts of spconv-2.x wins, a package that is not in the tree; the
    product resolves it deterministically: the last row wins every look-up, every row gets
    its own output.  INTEGRATION.md records this.)"""
    from msmdfusion_amd import spconv
    from msmdfusion_amd.fusion import voxel_modality_split
    from msmdfusion_amd.multimodal_encoder import SparseMultiModalEncoderPaint
    torch.manual_seed(0)
    enc = SparseMultiModalEncoderPaint(in_channels_2D=(64,) * 4, padding=(1, 1, [0, 1, 1], 0)) \
        .to(dev).train()
    enc.reference_quirks = True
    rng = np.random.RandomState(9)
    batch, c3 = 2, 16
    i3 = np.concatenate([_spacing2_cloud(rng, 2600, b, 0.0) for b in range(batch)])
    i2 = np.concatenate([_spacing2_cloud(rng, 2200, b, 0.5) for b in range(batch)])
    f3 = rng.randn(i3.shape[0], c3).astype(np.float32)
    f2 = rng.randn(i2.shape[0], 64).astype(np.float32)
    dummy = np.random.RandomState(5).rand(1, c3).astype(np.float32)
    enc.dummy_embedding_fn = lambda c, device: torch.from_numpy(dummy).to(device)
    a = spconv.SparseConvTensor(torch.from_numpy(f3).to(dev), torch.from_numpy(i3).to(dev), SHAPE0, batch)
    b = spconv.SparseConvTensor(torch.from_numpy(f2).to(dev), torch.from_numpy(i2).to(dev), SHAPE0, batch)
    a, b, s3, s2 = voxel_modality_split(a, b, batch, float_keys=True)
    false_matches = int((i3[_np(s3)] != i2[_np(s2)]).any(1).sum())
    true_matches = int((i3[_np(s3)] == i2[_np(s2)]).all(1).sum())
    assert false_matches > 100 and true_matches > 100, (false_matches, true_matches)
    out = enc.grouped_sparse_conv(a, b, s3, s2, 0, 2048, 6, 50, 13.3)
    exp = _oracle_stage(enc, 0, i3, f3, i2, f2, SHAPE0, batch, dummy, 2048, 6, 50, 13.3,
                        float_keys=True)
    assert np.array_equal(_np(out.indices), exp.idx)
    assert np.unique(exp.idx, axis=0).shape[0] == exp.idx.shape[0]     # no repeated coordinate
    np.testing.assert_allclose(_np(out.features), exp.feat, rtol=2e-4, atol=2e-4)
    # ... and it is not what the exact mode computes
    a2 = spconv.SparseConvTensor(torch.from_numpy(f3).to(dev), torch.from_numpy(i3).to(dev), SHAPE0, batch)
    b2 = spconv.SparseConvTensor(torch.from_numpy(f2).to(dev), torch.from_numpy(i2).to(dev), SHAPE0, batch)
    a2, b2, t3, t2 = voxel_modality_split(a2, b2, batch)
    # (fewer pairs, all of them true matches; a few true matches of the exact mode are taken
    # by an earlier row with the same rounded key in the reference's)
    assert true_matches <= t3.shape[0] < s3.shape[0]
    assert (i3[_np(t3)] == i2[_np(t2)]).all()


def test_detector_switch_selects_the_reference_split(dev):
    """MSMDFusionDetector(reference_quirks=True) (the config key): prepare() splits every
    scale with float keys == the oracle's float-key split of the same voxel sets; at the two
    fine scales (z up to 40 / 20 >= 17) that differs from the default detector's exact split,
    at the two coarse ones (keys < 2^24, x < 1000) it cannot.  The full forward runs and is
    deterministic in both modes."""
    from msmdfusion_amd import configs as C
    from msmdfusion_amd.detector import build_detector
    cfg = {k: v for k, v in C.MSMDFUSION_LC["model"].items()
           if k not in ("pts_backbone", "pts_neck", "pts_bbox_head")}
    B = 2
    pts = [torch.from_numpy(S.lidar_sweep(i, n_az=500)).to(dev) for i in range(B)]
    virt = [torch.from_numpy(S.virtual_points(i, n=20000)).to(dev) for i in range(B)]
    outs = {}
    for quirks in (False, True):
        torch.manual_seed(0)
        model = build_detector(dict(cfg, reference_quirks=quirks)).to(dev).train()
        assert model.reference_quirks is quirks and model._path.reference_quirks is quirks
        assert model.multimodal_middle_encoder.reference_quirks is quirks
        fixed = {c: torch.full((1, c), 0.25) for c in (16, 32, 64, 128)}
        model.multimodal_middle_encoder.dummy_embedding_fn = lambda c, device: fixed[c].to(device)
        with torch.no_grad():
            prep = model.prepare(pts, virt, nn_side_stream=False)
            bev = model.extract_sparse_feat(pts, virt, prepared=prep)
            bev2 = model.extract_sparse_feat(pts, virt)
        torch.cuda.synchronize()
        # (reference mode: a unified set may repeat a coordinate, and three rows summed into
        # one by sparse_add's atomics have no fixed order -- equal to rounding, not to the bit)
        assert torch.isfinite(bev).all()
        assert torch.equal(bev, bev2) if not quirks else torch.allclose(bev, bev2, atol=1e-5)
        outs[quirks] = (prep, bev)
        for i in range(4):
            i3 = _np(prep["stages"][i][0])
            i2 = _np(prep["v2"][i].indices)[:, [0, 2, 3, 4]]
            shape = [max(a, b) for a, b in zip(prep["stages"][i][1], model.spatial_shapes[i])]
            e3, e2, q3, q2 = _oracle_split(i3, i2, shape, B, quirks, quirks)
            assert np.array_equal(_np(prep["idx3_5"][i])[:, 1], e3), (quirks, i)
            assert np.array_equal(_np(prep["v2"][i].indices)[:, 1], e2), (quirks, i)
            assert np.array_equal(_np(prep["s3"][i]), q3) and np.array_equal(_np(prep["s2"][i]), q2)
    exact, ref = outs[False][0], outs[True][0]
    assert not torch.equal(exact["idx3_5"][0], ref["idx3_5"][0]), "scale 1 must alias on LiDAR data"
    for i in (2, 3):        # 11 x 360 x 360 and 5 x 180 x 180: float keys are exact
        assert torch.equal(exact["idx3_5"][i], ref["idx3_5"][i])
        assert torch.equal(exact["s3"][i], ref["s3"][i]) and torch.equal(exact["s2"][i], ref["s2"][i])
    assert not torch.equal(outs[False][1], outs[True][1])


def test_reference_batch_offsets_of_the_nearest_voxel_rows(dev):
    """sparse_multimodal_encoder_painting.py:355-369 adds `base = this_batch_mask_3D.sum()` --
    the PREVIOUS sample's count, not the running total: the same rows for batch <= 2, the
    reference's own (wrong) rows beyond; reference_quirks reproduces them."""
    from msmdfusion_amd.multimodal_encoder import SparseMultiModalEncoderPaint
    enc = SparseMultiModalEncoderPaint(in_channels_2D=(64,) * 4, padding=(1, 1, [0, 1, 1], 0)).to(dev)
    shape, B = [21, 200, 200], 3
    k = S.random_voxel_indices(6000, B, shape, seed=1)
    q = S.random_voxel_indices(900, B, shape, seed=2)
    k, q = k[np.argsort(k[:, 0], kind="stable")], q[np.argsort(q[:, 0], kind="stable")]
    dk, dq = torch.from_numpy(k).to(dev), torch.from_numpy(q).to(dev)
    cum = _np(enc.nearest_3d_of_only_2d(dq, dk, B, 2048, 6, 50, 13.3))
    enc.reference_quirks = True
    ref = _np(enc.nearest_3d_of_only_2d(dq, dk, B, 2048, 6, 50, 13.3))
    c3 = [int((k[:, 0] == b).sum()) for b in range(B)]
    o3 = np.cumsum([0] + c3)
    for b in range(B):
        rows = q[:, 0] == b
        hit = cum[rows] >= 0
        local = cum[rows][hit] - o3[b]
        assert np.array_equal(ref[rows][hit], local + (c3[b - 1] if b else 0))
        assert np.array_equal(ref[rows][~hit], cum[rows][~hit])
    assert np.array_equal(ref[q[:, 0] < 2], cum[q[:, 0] < 2]) and not np.array_equal(ref, cum)


def test_repeated_coordinate_through_subm_sparse_add_and_downscale(dev):
    """What reference mode can produce on real frames and the stage test above avoids: a false
    match puts a mixed voxel ON an only-3D voxel's coordinate, so the unified set of a GMA-Conv
    stage REPEATS a coordinate.  (spconv-2.x's result then depends on which of two racing hash
    inserts wins; the package is not in the tree.  The in-tree spconv-1.x CPU code,
    geometry.h:247-297, is input-stationary: both rows scatter into the last row's output and
    the earlier row's output keeps its centre product only -- yet another answer.)  The
    documented, deterministic behaviour of this product -- the conv tables are
    output-stationary, one input row per (output row, offset), and every look-up of a
    coordinate finds its LAST row -- pinned through the ops such a set meets on its way down
    the stack, each against the oracle on the equivalent duplicate-free problem:

      * SubM conv == the conv over the set with only the last row of every coordinate kept,
        its outputs handed to EVERY row of the coordinate (each row gets its own, equal,
        output; the earlier rows' features are never read);
      * sparse_add with the previous stage: rows sharing a coordinate are summed (COO add +
        coalesce, as torch's and spconv's) -- three rows on one coordinate included;
      * the strided down-scaling conv straight on the repeating set (stage 0 has no
        sparse_add in front) == the conv over the last rows; after sparse_add the set is
        duplicate-free and the plain oracle walk applies."""
    from msmdfusion_amd import spconv
    from msmdfusion_amd.spconv import functional as Fsp
    from test_gpu_modules import OracleSparse, oracle_forward
    rng = np.random.RandomState(21)
    batch, shape, c = 2, [21, 64, 64], 32
    base = S.random_voxel_indices(3000, batch, shape, seed=7)
    twice = rng.choice(base.shape[0], 60, replace=False)       # "only-3D" rows hit again
    idx = np.concatenate([base, base[twice]]).astype(np.int32)  # ... by later ("mixed") rows
    n = idx.shape[0]
    feat = rng.randn(n, c).astype(np.float32)
    first, last = twice, np.arange(base.shape[0], n)
    keep = np.setdiff1d(np.arange(n), first)                    # the last row of every coordinate
    where = {tuple(r): j for j, r in enumerate(idx[keep])}
    to_kept = np.array([where[tuple(r)] for r in idx])           # row -> its coordinate's kept row
    torch.manual_seed(3)
    subm = spconv.SubMConv3d(c, c, 3, padding=1, bias=False).to(dev)
    down = spconv.SparseConv3d(c, 64, 3, stride=2, padding=1, bias=False).to(dev)

    def tensor(f, i):
        return spconv.SparseConvTensor(torch.from_numpy(np.ascontiguousarray(f)).to(dev),
                                       torch.from_numpy(np.ascontiguousarray(i)).to(dev),
                                       shape, batch)
    with torch.no_grad():
        y = subm(tensor(feat, idx))
        exp_kept = oracle_forward(subm, OracleSparse(feat[keep], idx[keep], shape, batch))
        assert np.array_equal(_np(y.indices), idx)
        yf = _np(y.features)
        np.testing.assert_allclose(yf, exp_kept.feat[to_kept], rtol=1e-4, atol=1e-4)
        # own output each, the same sums (in another tile: equal to the last bit or two)
        np.testing.assert_allclose(yf[first], yf[last], rtol=1e-5, atol=1e-6)
        other = feat.copy()
        other[first] = rng.randn(first.shape[0], c)             # the earlier rows are never read
        assert torch.equal(subm(tensor(other, idx)).features, y.features)
        later = feat.copy()
        later[last] += 1.0                                      # ... the last rows are
        assert not torch.equal(subm(tensor(later, idx)).features, y.features)

        # sparse_add with a previous-stage tensor that shares some of those coordinates
        prev_idx = np.concatenate([base[twice[:20]], base[rng.choice(base.shape[0], 500, False)],
                                   S.random_voxel_indices(800, batch, shape, seed=8)])
        prev_idx = prev_idx[np.sort(np.unique(prev_idx, axis=0, return_index=True)[1])].astype(np.int32)
        prev = rng.randn(prev_idx.shape[0], c).astype(np.float32)
        z = Fsp.sparse_add(y, tensor(prev, prev_idx))
        ei, ef, _, _ = O.sparse_add(yf, idx, prev, prev_idx, shape)
        assert np.array_equal(_np(z.indices), ei)
        assert np.unique(ei, axis=0).shape[0] == ei.shape[0]     # the union repeats nothing
        np.testing.assert_allclose(_np(z.features), ef, rtol=1e-5, atol=1e-5)
        key = {tuple(r): j for j, r in enumerate(ei)}
        for a, b_ in zip(first[:20], last[:20]):                # three rows on one coordinate
            p = np.where((prev_idx == idx[a]).all(1))[0][0]
            np.testing.assert_allclose(_np(z.features)[key[tuple(idx[a])]],
                                       yf[a] + yf[b_] + prev[p], rtol=1e-5, atol=1e-5)

        # the down-scaling conv after the union: plain oracle walk
        got = down(z)
        exp = oracle_forward(down, OracleSparse(ef, ei, shape, batch))
        assert np.array_equal(_np(got.indices), exp.idx)
        np.testing.assert_allclose(_np(got.features), exp.feat, rtol=1e-4, atol=1e-4)
        # ... and straight on the repeating set: the last row of a coordinate feeds its outputs
        got = down(tensor(yf, idx))
        exp = oracle_forward(down, OracleSparse(yf[keep], idx[keep], shape, batch))
        assert np.array_equal(_np(got.indices), exp.idx)
        np.testing.assert_allclose(_np(got.features), exp.feat, rtol=1e-4, atol=1e-4)
        moved = yf.copy()
        moved[first] += 1.0        # (yf[first] == yf[last] so far: now the two rows differ)
        for _ in range(3):         # the earlier rows are not read -- whichever thread fills last
            assert torch.equal(down(tensor(moved, idx)).features, got.features)
