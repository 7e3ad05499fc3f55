#!/usr/bin/env python
"""HBM-side roofline of the integer kernels: hard voxelization and the SubM /
strided rulebook builders, at the nominal size (BASELINE configs[1]: 4 clouds,
0.075 m) and at the stress size (configs[4]: 10-sweep ~290k-pt clouds, 0.05 m
voxels, ~1M active voxels at batch 2-4).

Prints one JSON line per case: time per call (HIP events on the launch stream),
algorithmic bytes and GB/s against the 8 TB/s HBM3E peak.
Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes)
for the measured traffic; see profiles/.
"""
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from msmdfusion_amd import kernels as K  # noqa: E402
from msmdfusion_amd import synthetic as S  # noqa: E402

HBM_PEAK = 8000.0  # GB/s


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        out = fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3, out   # us


def case(name, clouds, voxel_size, shape, max_voxels):
    dev = clouds[0].device
    b = len(clouds)
    n_pts = sum(c.shape[0] for c in clouds)
    c_feat = clouds[0].shape[1]

    def vox():
        return K.hard_voxelize_batch(clouds, voxel_size, S.POINT_CLOUD_RANGE, 10, max_voxels,
                                     want_voxels=True, want_mean=False)
    us, res = timed(vox, 10)
    m = sum(r[1].shape[0] for r in res)
    t_slots = sum(1 << max((2 * c.shape[0] - 1).bit_length(), 6) for c in clouds)
    vox_bytes = 4 * c_feat * n_pts + 4 * (10 * c_feat + 4) * m + 8 * t_slots
    out = [dict(kernel="hard_voxelize (batch, incl. one host read)", case=name, points=n_pts,
                voxels=m, us=round(us, 1), algo_MB=round(vox_bytes / 1e6, 2),
                GBps=round(vox_bytes / us / 1e3, 1), frac_hbm=round(vox_bytes / us / 1e3 / HBM_PEAK, 4))]
    idx = torch.cat([F.pad(r[1], (1, 0), value=i) for i, r in enumerate(res)]).contiguous()
    n = idx.shape[0]
    t = 1 << max((2 * n - 1).bit_length(), 6)
    sub_bytes = 16 * n + 8 * t + 4 * 27 * n            # index rows + hash slots + the table
    auto = K.subm_index_method(n, b, shape)
    for method in ("hash", "bitmap"):
        us, nbr = timed(lambda: K.rulebook_subm(idx, b, shape, 3, method=method))
        out.append(dict(kernel="rulebook_subm3d 3x3x3 (%s index%s)"
                               % (method, ", the automatic choice" if method == auto else ""),
                        case=name, voxels=n, pairs=int((nbr >= 0).sum()), us=round(us, 1),
                        algo_MB=round(sub_bytes / 1e6, 2), GBps=round(sub_bytes / us / 1e3, 1),
                        frac_hbm=round(sub_bytes / us / 1e3 / HBM_PEAK, 4)))
    # the deeper stages' grids (stride-2 outputs): same voxels-per-sample order, 8x fewer cells
    idx2, _, _, shape2 = K.rulebook_conv(idx, b, shape, 3, 2, 1, need_bwd=False)
    n2 = idx2.shape[0]
    for method in ("hash", "bitmap"):
        us, _ = timed(lambda: K.rulebook_subm(idx2, b, list(shape2), 3, method=method))
        out.append(dict(kernel="rulebook_subm3d 3x3x3, stage-1 grid %s (%s index%s)"
                               % ("x".join(map(str, shape2)), method, ", the automatic choice"
                                  if method == K.subm_index_method(n2, b, shape2) else ""),
                        case=name, voxels=n2, us=round(us, 1)))
    us, (oi, nf, nb_, osz) = timed(lambda: K.rulebook_conv(idx, b, shape, 3, 2, 1))
    words = (b * osz[0] * osz[1] * osz[2] + 31) // 32
    cv_bytes = 32 * n + 8 * words + 4 * 27 * (oi.shape[0] + n) + 16 * oi.shape[0]
    out.append(dict(kernel="rulebook_conv3d k3 s2 p1 (incl. one host read)", case=name, voxels=n,
                    out_voxels=oi.shape[0], us=round(us, 1), algo_MB=round(cv_bytes / 1e6, 2),
                    GBps=round(cv_bytes / us / 1e3, 1),
                    frac_hbm=round(cv_bytes / us / 1e3 / HBM_PEAK, 4)))
    us, _ = timed(lambda: K.rulebook_pairs(nbr))
    pr_bytes = 2 * 4 * 27 * n + 8 * int((nbr >= 0).sum())
    out.append(dict(kernel="rulebook_pairs (table -> indicePairs)", case=name, us=round(us, 1),
                    algo_MB=round(pr_bytes / 1e6, 2), GBps=round(pr_bytes / us / 1e3, 1),
                    frac_hbm=round(pr_bytes / us / 1e3 / HBM_PEAK, 4)))
    return out


def main():
    dev = torch.device("cuda:0")
    rows = []
    nominal = [torch.from_numpy(S.lidar_sweep(i)).to(dev) for i in range(4)]
    rows += case("nominal: 4 x 28.7k pts, 0.075 m, grid 41x1440x1440", nominal, S.VOXEL_SIZE,
                 S.SPARSE_SHAPE, 120000)
    stress = [torch.from_numpy(S.lidar_sweep(10 + i, sweeps=10)).to(dev) for i in range(4)]
    rows += case("stress: 4 x 10-sweep ~290k pts, 0.05 m, grid 41x2160x2160", stress,
                 [0.05, 0.05, 0.2], [41, 2160, 2160], 1200000)
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
