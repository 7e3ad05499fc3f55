#!/usr/bin/env python
"""wgrad: the plane-tensor kernel (csrc/spconv_planes.hip) against the r01 kernel that
re-splits fp32 operands in registers, on the bench workloads' voxel sets and widths
(SubM on the 64-/128-channel LiDAR stages, the fusion stack's 80/96/128/192 widths)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from msmdfusion_amd import kernels as K
from msmdfusion_amd import synthetic as S
from msmdfusion_amd.voxelize import Voxelization
from tools.split_bench import timed

dev = torch.device("cuda:0")
vox = Voxelization(S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, 10, (120000, 160000))
clouds = [torch.from_numpy(S.lidar_sweep(i)).to(dev) for i in range(4)]
coors = [F.pad(c, (1, 0), value=b) for b, (_, c, _) in
         enumerate(vox.forward_batch(clouds, fused_mean=True))]
idx = torch.cat(coors).contiguous()
shape = list(S.SPARSE_SHAPE)
stages = []
for i, pad in enumerate([1, 1, [0, 1, 1]]):
    stages.append((idx, shape))
    oidx, nf, nb, oshape = K.rulebook_conv(idx, 4, shape, 3, 2, pad)
    idx, shape = oidx, oshape
stages.append((idx, shape))
for si, cin, cout in [(2, 64, 64), (3, 128, 128), (0, 80, 80), (1, 96, 96), (2, 128, 128),
                      (3, 192, 192), (1, 96, 128), (2, 128, 192)]:
    idx, shape = stages[si]
    n = idx.shape[0]
    nbr = K.rulebook_subm(idx, 4, shape, 3)
    pairs, num = K.rulebook_pairs(nbr)
    P = int(num.sum())
    f, g = torch.randn(n, cin, device=dev), torch.randn(n, cout, device=dev)
    t_old = timed(lambda: K.conv_wgrad_split(f, g, pairs, num, 3))
    t_sp = timed(lambda: K.split_planes(f, 3))
    fp, gp = K.split_planes(f, 3), K.split_planes(g, 3)
    t_new = timed(lambda: K.conv_wgrad_planes(fp, gp, pairs, num))
    a, b = K.conv_wgrad_split(f, g, pairs, num, 3), K.conv_wgrad_planes(fp, gp, pairs, num)
    fl = 2.0 * P * cin * cout
    print("stage %d %3d->%3d n=%6d pairs=%7d | r01 split %5.0f us %5.1f TF | planes %5.0f us %5.1f TF "
          "(frac %.3f) + split pass %4.0f us (%.2f TB/s) | maxdiff/max %.1e" % (
              si, cin, cout, n, P, t_old, fl / t_old / 1e6, t_new, fl / t_new / 1e6,
              fl / t_new / 1e6 / 419.4, t_sp, n * cin * 10 / t_sp / 1e6,
              (a - b).abs().max().item() / a.abs().max().item()), flush=True)
