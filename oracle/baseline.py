"""Host-side timing of the reference algorithm on whole samples (bench.py's
`cpu_baseline` leg).  TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

Both functions push ONE synthetic sample through the oracle's restatement of the
reference CPU path (oracle/msmd_oracle.c: voxelization_cpu.cpp, geometry.h
rulebooks, the gather -> GEMM -> scatter-add loop of spconv_ops.h:260-456) with
random weights of the configured layer shapes, and return the wall time plus a
description of what ran.  They are timing harnesses: parity lives in tests/.

  transfusion_l_sample -- BASELINE.json configs[1]: voxelize + 8 rulebooks + 21
      sparse convs forward AND backward (dgrad + wgrad).
  lc_sample -- configs[2] (MSMDFusion.py:421-443 with the LC config): the LiDAR
      encoder forward only (frozen: tools/train.py:185-219), virtual-point
      voxelization + mean VFE at 4 scales (MSMDFusion.py:371-393), modality
      split (:251-325), fps_NN_fast (sparse_multimodal_encoder_painting.py:
      276-323), the gates, and the fusion stack's 16 sparse convs forward +
      backward with sparse_add in between (:325-459).

Skipped in both (elementwise, < 1 % of the CPU time): BatchNorm, ReLU, the
residual adds and the optimizer; in lc_sample also the backward of the four
gate Linears.  reference_conv_leg() times the conv part alone through the
reference's own compiled loop (oracle/_ref: gather / torch::mm / scatter-add) beside
this port on the same layers -- on the build container's cores torch::mm's BLAS makes
the reference loop ~4x faster than the port's plain OpenMP GEMM on the wide layers.
"""
import time

import numpy as np

from msmdfusion_amd import synthetic as S

from . import oracle as O

ENCODER_CHANNELS = ((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128))   # Appendix A.1
DOWN_PADS = {0: 1, 1: 1, 2: [0, 1, 1]}
C3 = (16, 32, 64, 128)                    # Appendix A.2
MM_OUT = (32, 64, 128, 128)
MM_PAD = (1, 1, [0, 1, 1], 0)
MM_KS = (3, 3, 3, [3, 1, 1])
MM_ST = (2, 2, 2, [2, 1, 1])
FPS_NUM, RADIUS, MAX_CLUSTER, DIST = 2048, (6, 3, 2, 1), (200, 100, 50, 25), (13.3, 6.6, 3.3, 1.6)
SPATIAL = ([41, 1440, 1440], [21, 720, 720], [11, 360, 360], [5, 180, 180])


def _encoder_layers():
    layers = [("subm", 5, 16)]
    for i, blocks in enumerate(ENCODER_CHANNELS):
        cin = layers[-1][2]
        for j, cout in enumerate(blocks):
            if j == len(blocks) - 1 and i != 3:
                layers.append(("down%d" % i, cin, cout))
            else:
                layers += [("subm", cout, cout), ("subm", cout, cout)]
            cin = cout
    layers.append(("out", 128, 128))
    return layers


def _lidar_encoder(pts, rng, backward, keep_stages=False):
    """-> (macs, stage outputs [(feat, idx, shape)] if keep_stages)."""
    v, c, n = O.hard_voxelize(pts, S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, 10, 120000)
    feat = O.voxel_mean(v, n)
    idx = np.concatenate([np.zeros((c.shape[0], 1), np.int32), c], 1)
    shape = list(S.SPARSE_SHAPE)
    macs, cache, stages = 0, {}, []
    for li, (kind, cin, cout) in enumerate(_encoder_layers()):
        if kind == "subm":
            key = (idx.shape[0], tuple(shape))
            if key not in cache:
                cache[key] = O.get_indice_pairs(idx, 1, shape, 3, 1, 1, 1, True)
            oi, pr, nm, osz = cache[key]
            w = rng.randn(27, cin, cout).astype(np.float32) * 0.05
            out = O.indice_conv_fwd(feat, w, pr, nm, oi.shape[0], subm=True)
            if backward:
                O.indice_conv_bwd(feat, w, out, pr, nm, subm=True)
        else:
            ks, st, pd = (3, 2, DOWN_PADS[int(kind[4])]) if kind != "out" else \
                ([3, 1, 1], [2, 1, 1], 0)
            oi, pr, nm, osz = O.get_indice_pairs(idx, 1, shape, ks, st, pd, 1, False)
            w = rng.randn(pr.shape[0], cin, cout).astype(np.float32) * 0.05
            out = O.indice_conv_fwd(feat, w, pr, nm, oi.shape[0])
            if backward:
                O.indice_conv_bwd(feat, w, out, pr, nm)
            idx, shape = oi, osz
        macs += int(nm.sum()) * cin * cout
        feat = np.maximum(out, 0)
        # encode_features[0..3] (sparse_encoder.py:117-133): conv_input's output, then
        # the output of each stage = of its closing stride-2 conv
        if keep_stages and (li == 0 or kind.startswith("down")):
            stages.append((feat, idx, list(shape)))
    return macs, stages, c.shape[0]


def transfusion_l_sample(seed):
    pts = S.lidar_sweep(seed)
    t0 = time.perf_counter()
    macs, _, nvox = _lidar_encoder(pts, np.random.RandomState(0), backward=True)
    return dict(seconds=time.perf_counter() - t0, gmac_fwd=macs / 1e9, points=pts.shape[0],
                voxels=nvox)


def _conv_fwd_bwd(feat, idx, shape, rng, cin, cout, ks=3, st=1, pd=1, subm=True, rb=None):
    if rb is None:
        rb = O.get_indice_pairs(idx, 1, shape, ks, st, pd, 1, subm)
    oi, pr, nm, osz = rb
    w = rng.randn(pr.shape[0], cin, cout).astype(np.float32) * 0.05
    out = O.indice_conv_fwd(feat, w, pr, nm, oi.shape[0], subm=subm)
    O.indice_conv_bwd(feat, w, out, pr, nm, subm=subm)
    return np.maximum(out, 0), oi, osz, int(nm.sum()) * cin * cout, rb


def _fps_nn(query, key, radius, max_cluster, thresh):
    nq = query.shape[0]
    if nq == 0 or key.shape[0] == 0:
        return np.full((nq,), -1, np.int32)
    if nq <= FPS_NUM:
        return O.nn_search(query, key, thresh)
    q = query.astype(np.float32)[None]
    rep_idx = O.furthest_point_sample(q, FPS_NUM)[0]
    rep = query[rep_idx]
    rep_nn = O.nn_search(rep, key, thresh)
    grp = O.ball_query(0, radius, max_cluster, q, rep.astype(np.float32)[None])[0]
    return O.nn_assign(grp, rep_nn, nq)


def lc_sample(seed):
    pts, virt = S.lidar_sweep(seed), S.virtual_points(seed)
    rng = np.random.RandomState(0)
    t0 = time.perf_counter()
    macs_enc, stages, nvox = _lidar_encoder(pts, rng, backward=False, keep_stages=True)
    macs = 0
    prev = None
    n2_total = 0
    for i in range(4):
        f3, i3, shape3 = stages[i]
        vs = [v * (2 ** i) for v in S.VOXEL_SIZE]
        v, c, n = O.hard_voxelize(virt, vs, S.POINT_CLOUD_RANGE, 10, 120000)
        f2 = O.voxel_mean(v, n)
        f2[:, :3] /= np.array([13.5, 13.5, 2.0], np.float32)
        i2 = np.concatenate([np.zeros((c.shape[0], 1), np.int32), c], 1)
        n2_total += i2.shape[0]
        shape = [max(a, b) for a, b in zip(shape3, SPATIAL[i])]
        mix3, mix2, p3, p2 = O.modality_split(i3[:, 1:], i2[:, 1:], shape)
        o2 = i2[mix2 == 0]
        nn3 = _fps_nn(o2[:, 1:], i3[:, 1:], RADIUS[i], MAX_CLUSTER[i], DIST[i])
        c3 = C3[i]
        wg = rng.randn(c3, 64).astype(np.float32) * 0.1
        cross = np.maximum(np.concatenate([f3, rng.rand(1, c3).astype(np.float32)]) @ wg, 0)
        o2_feat = cross[nn3] * f2[mix2 == 0]
        m3f = f3[p3]
        m2f = np.maximum(m3f @ wg, 0) * f2[p2]
        only3_f, only3_i = f3[mix3 == 0], i3[mix3 == 0]
        only3_f, _, _, m, _ = _conv_fwd_bwd(only3_f, only3_i, shape3, rng, c3, c3)
        macs += m
        cm = c3 + 64
        uf = np.concatenate([np.pad(only3_f, ((0, 0), (0, 64))), np.pad(o2_feat, ((0, 0), (c3, 0))),
                             np.concatenate([m3f, m2f], 1)]).astype(np.float32)
        ui = np.concatenate([only3_i, o2, i2[p2]]).astype(np.int32)
        uf, _, _, m, rb = _conv_fwd_bwd(uf, ui, SPATIAL[i], rng, cm, cm)
        macs += m
        uf, _, _, m, _ = _conv_fwd_bwd(uf, ui, SPATIAL[i], rng, cm, cm, rb=rb)
        macs += m
        if prev is not None:
            ui, uf, _, _ = O.sparse_add(uf, ui, prev[0], prev[1], SPATIAL[i])
        df, di, dshape, m, _ = _conv_fwd_bwd(uf, ui, SPATIAL[i], rng, cm, MM_OUT[i] + 64,
                                             MM_KS[i], MM_ST[i], MM_PAD[i], subm=False)
        macs += m
        prev = (df, di)
    return dict(seconds=time.perf_counter() - t0, gmac_fwd=(macs_enc + macs) / 1e9,
                gmac_fwd_fusion=macs / 1e9, points=pts.shape[0], voxels=nvox,
                virtual_points=virt.shape[0], virtual_voxels=n2_total)


def reference_conv_leg(seed=0, budget_s=6.0):
    """The conv part of the path through the REFERENCE's own compiled CPU code
    (oracle/_ref/libmsmd_ref.so: spconv's SparseGatherFunctor / SparseScatterAddFunctor around
    torch::mm, the loop of spconv_ops.h:260-456) beside the OpenMP port on the SAME layers:
    SubM 3x3x3 forward + backward (dgrad + wgrad) on the LiDAR voxels of one synthetic sample
    at the encoder's widths, widest layers first until ~budget_s of reference time is spent.
    -> dict with both GMAC/s figures, or None when the reference build is not on this box."""
    if not O.have_ref():
        return None
    import torch
    pts = S.lidar_sweep(seed)
    v, c, n = O.hard_voxelize(pts, S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, 10, 120000)
    idx = np.concatenate([np.zeros((c.shape[0], 1), np.int32), c], 1)
    shape = list(S.SPARSE_SHAPE)
    rng = np.random.RandomState(seed)
    layers, t_ref, t_port, macs = [], 0.0, 0.0, 0
    # stage s: the voxel set after s stride-2 convs, SubM at the stage's width
    for s, width in enumerate((16, 32, 64, 128)):
        if s:
            idx, _, _, shape = O.get_indice_pairs(idx, 1, shape, 3, 2, DOWN_PADS[s - 1], 1, False)
            shape = list(shape)
        oi, pr, nm, _ = O.get_indice_pairs(idx, 1, shape, 3, 1, 1, 1, True)
        layers.append((width, idx.shape[0], pr, nm))
    data = [(width, rows, pr, nm, rng.randn(rows, width).astype(np.float32),
             rng.randn(27, width, width).astype(np.float32) * 0.05,
             rng.randn(rows, width).astype(np.float32)) for width, rows, pr, nm in reversed(layers)]
    rounds = 0
    # whole rounds over the four layers (the first is the warm-up of both libraries' thread
    # pools and is not counted when more follow) until ~budget_s of reference + port time
    while True:
        tr = tp = 0.0
        m = 0
        for width, rows, pr, nm, f, w, g in data:
            for use_ref in (False, True):
                t0 = time.perf_counter()
                O.indice_conv_fwd(f, w, pr, nm, rows, subm=True, use_ref=use_ref)
                O.indice_conv_bwd(f, w, g, pr, nm, subm=True, use_ref=use_ref)
                dt = time.perf_counter() - t0
                if use_ref:
                    tr += dt
                else:
                    tp += dt
            m += 3 * int(nm.sum()) * width * width          # fwd + dgrad + wgrad
        rounds += 1
        if rounds == 1:
            first = (tr, tp, m)
            spent = tr + tp
            continue
        t_ref += tr
        t_port += tp
        macs += m
        spent += tr + tp
        if spent > budget_s or rounds >= 40:
            break
    if macs == 0:
        t_ref, t_port, macs = first
    return dict(reference_gmac_per_s=round(macs / 1e9 / t_ref, 2),
                port_gmac_per_s=round(macs / 1e9 / t_port, 2),
                reference_seconds=round(t_ref, 2), port_seconds=round(t_port, 2),
                torch_threads=torch.get_num_threads(),
                rounds=rounds - 1 if rounds > 1 else 1,
                layers="SubM 3x3x3 fwd+dgrad+wgrad on one synthetic sample's LiDAR voxel sets, "
                       "widths 128, 64, 32, 16; whole rounds after an untimed first one")


def reference_index_leg(seed=0, rounds=2):
    """The INDEX part of the path -- hard voxelization (voxelization_cpu.cpp:68-96) of the
    LiDAR cloud and of the virtual points at the four scales, and the rulebooks
    (geometry.h getIndicePairsConv / getIndicePairsSubM via spconv's CPU entry) of the four
    SubM voxel sets and three stride-2 convs of one synthetic sample -- through the
    REFERENCE's own compiled CPU code (oracle/_ref: ref_hard_voxelize, ref_get_indice_pairs)
    beside the port, same inputs, both single-threaded as the reference's loops are.
    -> dict of seconds per sample for both, or None when the reference build is absent."""
    if not O.have_ref():
        return None
    pts, virt = S.lidar_sweep(seed), S.virtual_points(seed)
    out = {}
    for use_ref in (False, True):
        t_vox = t_rb = 0.0
        for r in range(rounds + 1):
            t0 = time.perf_counter()
            v, c, n = O.hard_voxelize(pts, S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, 10, 120000,
                                      use_ref=use_ref)
            for i in range(4):
                vs = [x * (2 ** i) for x in S.VOXEL_SIZE]
                O.hard_voxelize(virt, vs, S.POINT_CLOUD_RANGE, 10, 120000, use_ref=use_ref)
            t1 = time.perf_counter()
            idx = np.concatenate([np.zeros((c.shape[0], 1), np.int32), c], 1)
            shape = list(S.SPARSE_SHAPE)
            pairs = 0
            for s in range(4):
                if s:
                    idx, _, nm, shape = O.get_indice_pairs(idx, 1, shape, 3, 2, DOWN_PADS[s - 1], 1,
                                                           False, use_ref=use_ref)
                    shape = list(shape)
                    pairs += int(nm.sum())
                _, _, nm, _ = O.get_indice_pairs(idx, 1, shape, 3, 1, 1, 1, True, use_ref=use_ref)
                pairs += int(nm.sum())
            t2 = time.perf_counter()
            if r:                       # the first round warms the page cache
                t_vox += t1 - t0
                t_rb += t2 - t1
        key = "reference" if use_ref else "port"
        out[key + "_voxelize_s"] = round(t_vox / rounds, 4)
        out[key + "_rulebooks_s"] = round(t_rb / rounds, 4)
    out["points"] = int(pts.shape[0] + 4 * virt.shape[0])
    out["rulebook_pairs"] = pairs
    out["what"] = ("one synthetic sample: hard_voxelize of the LiDAR cloud + the virtual points at "
                   "4 scales; SubM 3x3x3 rulebooks of the 4 encoder voxel sets + the 3 stride-2 "
                   "rulebooks between them; reference = oracle/_ref (voxelization_cpu.cpp, "
                   "geometry.h compiled where they lie), port = oracle/msmd_oracle.c; 1 thread each")
    return out
